"""sph_project_amd -- MI355X (gfx950) native SPH hot path behind the reference's SPH package API.

Layout:
  csrc/            hand-written HIP kernels + the C-ABI (include/sph_hip.h) -> libsph_hip.so
  _lib.py          thin ctypes binding of that C-ABI (no torch, no taichi)
  scene.py         host-side scene arithmetic of BaseContainer/BaseSolver.__init__
  SPH/             drop-in mirror of the reference's `SPH` package (same module/class names)
  run_simulation.py  GGUI-free driver with the reference's loop arithmetic and PLY export
"""
__version__ = "0.1.0"
