import sys, json, numpy as np
sys.path.insert(0, '.')
from sph_project_amd import _lib as L
from tests import helpers as H
name = sys.argv[1]
z = np.load(f'tests/golden/{name}.npz'); cfg = json.loads(bytes(z["scene_json"]).decode())
container, solver = H.build_product(cfg, fast_math=0)
container.insert_object(); e = container.engine
e.upload(L.F_POSITION, z["init_positions"]); solver.prepare()
ref = None
step = 0
for cp in z["checkpoints"]:
    while step < cp:
        solver.step(); step += 1
        st = solver.stats(); print("step", step, "iters hip", st["iter_density"], st["iter_divergence"], "err", st["err_density"], st["err_divergence"])
    pre = f"s{cp}_"
    print(" fixture iters", z[pre+"iter_d"], z[pre+"iter_v"])
    k = e.download(L.F_DFSPH_KAPPA); kr = z[pre+"kappa"]
    dif = np.abs(k-kr); i = dif.argmax(); print(" kappa maxdiff", dif.max(), "at slot", i, k[i], kr[i], "scale", np.abs(kr).max(), "n nonzero", (kr!=0).sum(), (k!=0).sum())
    ids = e.download(L.F_PARTICLE_ID); print(" ids equal", np.array_equal(ids, z[pre+"ids"]))
