"""Drop-in mirror of the reference's `SPH` package (same sub-modules and class names), backed by
libsph_hip.so instead of Taichi.  Put the directory that contains this package first on
sys.path (or run sph_project_amd/run_simulation.py) and the reference's driver imports resolve here."""
import os as _os
import sys as _sys

# `import SPH` with sph_project_amd/ on sys.path (the reference driver's own import line, INTEGRATION.md option A) must still find
# the engine binding, which lives one package up
_pkg_parent = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _pkg_parent not in _sys.path:
    _sys.path.insert(0, _pkg_parent)
