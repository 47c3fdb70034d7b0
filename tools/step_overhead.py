"""Per-step cost of the drop-in Python path at C2: solver.step() (step_begin / host hook / step_end per step), solver.advance(n)
(one sph_step(h, n)), engine.step_async(n); interleaved so that all three see the same flow states."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph_project_amd import product as P  # noqa: E402

container, solver = P.build_product(P.c2_scene(), fast_math=1)
solver.prepare()
eng = container.engine
for _ in range(20):
    solver.step()
eng.synchronize()
acc = {"solver.step()": 0.0, "solver.advance(n)": 0.0, "engine.step_async(n)": 0.0}
n, rounds = 50, 4
for _ in range(rounds):
    t0 = time.perf_counter()
    for _ in range(n):
        solver.step()
    eng.synchronize()
    t1 = time.perf_counter()
    solver.advance(n)
    eng.synchronize()
    t2 = time.perf_counter()
    eng.step_async(n)
    eng.synchronize()
    t3 = time.perf_counter()
    acc["solver.step()"] += t1 - t0
    acc["solver.advance(n)"] += t2 - t1
    acc["engine.step_async(n)"] += t3 - t2
print("; ".join("%s: %.3f ms/step" % (k, 1e3 * v / (n * rounds)) for k, v in acc.items()))
