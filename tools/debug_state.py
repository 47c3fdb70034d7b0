"""Manual check: run a built-in scene for some steps and dump the state by particle id, so that two builds / switches can be compared
bit by bit.   python tools/debug_state.py <c1|c2|c1box> <steps> <fast_math> <tag>   ->  gpurun_out/state_<scene>_<tag>.npz"""
import os, sys
import numpy as np
sys.path.insert(0, '.')
from sph_project_amd import _lib as L, product as P
from tests import helpers as H
scene, steps, fast, tag = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
cfg = {"c1": lambda: P.dam_break_scene(), "c2": lambda: P.c2_scene("wcsph"),
       "c1box": lambda: P.dam_break_scene(add_domain_box=True) if "add_domain_box" in P.dam_break_scene.__code__.co_varnames else P.dam_break_scene()}[scene]()
container, solver = P.build_product(cfg, fast_math=fast)
solver.prepare()
e = container.engine
import time
e.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    solver.step()
e.synchronize(); dt = time.perf_counter() - t0
ids = e.download(L.F_PARTICLE_ID)
out = {k: H.by_id(ids, e.download(f)) for k, f in (("x", L.F_POSITION), ("v", L.F_VELOCITY), ("rho", L.F_DENSITY), ("p", L.F_PRESSURE))}
st = solver.stats()
print(scene, tag, "steps", steps, "ms/step %.4f" % (1e3 * dt / steps), "lds_fallback", st.get("lds_fallback_blocks"), "pairs", st.get("pair_interactions", st.get("pairs")))
os.makedirs("gpurun_out", exist_ok=True)
np.savez(f"gpurun_out/state_{scene}_{tag}.npz", **out)
