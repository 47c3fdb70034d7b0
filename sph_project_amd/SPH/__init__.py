"""Drop-in mirror of the reference's `SPH` package (same sub-modules and class names), backed by
libsph_hip.so instead of Taichi.  Put the directory that contains this package first on
sys.path (or run sph_project_amd/run_simulation.py) and the reference's driver imports resolve here."""
