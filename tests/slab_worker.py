"""One rank of a multi-process slab-sharded run (spawned by tests/test_hip_slab.py and usable by hand):
python tests/slab_worker.py <rank> <nranks> <id_hex> <scene.json> <steps> <out.npz> [jitter seed]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# test-hook library only (SPH_HIP_LIB=libsph_hip_testhooks.so): the consumer stall is applied on ONE rank, so that its neighbour runs ahead
if os.environ.get("SPH_WORKER_DELAY_RANK") not in (None, "", "all", sys.argv[1]):
    os.environ.pop("SPH_TEST_HALO_DELAY_US", None)

from sph_project_amd import _lib as L  # noqa: E402
from sph_project_amd import scene, slab  # noqa: E402
from sph_project_amd.SPH.utils import SimConfig  # noqa: E402
from tests import helpers as H  # noqa: E402


def main():
    rank, nranks = int(sys.argv[1]), int(sys.argv[2])
    uid = bytes.fromhex(sys.argv[3])
    cfg = json.load(open(sys.argv[4]))
    steps = int(sys.argv[5])
    out = sys.argv[6]
    jitter, seed = (float(sys.argv[7]), int(sys.argv[8])) if len(sys.argv) > 8 else (0.0, 0)
    # slab cuts from the z histogram of the initial lattice (every rank computes the same plan)
    c, geo, batches = H.scene_particles(cfg)
    pos = np.concatenate([b["pos"] for b in batches])
    mat = np.concatenate([b["material"] for b in batches])
    if jitter > 0:
        fl = mat == 1
        pos[fl] = H.perturb(pos[fl], jitter, seed)
    # SPH_WORKER_STATE=<npz with x, v>: the scene's one fluid block is replaced by these particles (a state the undecomposed product reached
    # by itself, e.g. the collapsed column of step 2500), global id = row of the arrays; slab membership and cuts from THEIR positions
    state = np.load(os.environ["SPH_WORKER_STATE"]) if os.environ.get("SPH_WORKER_STATE") else None
    if state is not None:
        pos, mat = np.ascontiguousarray(state["x"], np.float32), np.ones(len(state["x"]), np.int32)
    nz = int(geo.grid_num[2])
    hist = np.bincount(slab.cell_layer(pos[:, 2], geo.dh, nz), minlength=nz)
    cuts = slab.plan_slabs(hist, nranks)
    extra = {}
    if int(os.environ.get("SPH_FIXED_ITERATIONS", "0")) > 0:
        extra["fixed_iterations"] = int(os.environ["SPH_FIXED_ITERATIONS"])
    container, solver = H.build_product(cfg, slab=dict(rank=rank, nranks=nranks, unique_id=uid, cuts=cuts),
                                        fast_math=int(os.environ.get("SPH_FAST", "0")), **extra)
    if state is not None:
        blk = container.fluid_blocks[0]
        assert len(container.fluid_blocks) == 1 and not container.fluid_bodies and not container.rigid_bodies
        container.fluid_blocks = []
        container.fluid_bodies = [dict(objectId=blk["objectId"], entryTime=-1.0, voxelizedPoints=pos, particleNum=len(pos), velocity=[0.0, 0.0, 0.0],
                                       density=blk["density"], color=blk["color"])]   # (the explicit-points form mesh bodies take: slab selection by these positions)
        container.insert_object()
        ids = np.concatenate(container._global_ids)
        container.engine.upload(L.F_VELOCITY, np.ascontiguousarray(state["v"], np.float32)[ids])
    if jitter > 0:  # same perturbed lattice on every rank: overwrite the positions of the particles kept here
        container.insert_object()
        ids = np.concatenate(container._global_ids)
        container.engine.upload(L.F_POSITION, pos[ids])
    solver.prepare()
    if os.environ.get("SPH_WORKER_DIE_RANK") == str(rank):   # test hook: this rank disappears while its neighbours wait for its halo
        os._exit(3)
    hist = []
    if os.environ.get("SPH_WORKER_ADVANCE") == "1":
        solver.advance(steps)   # one call: WCSPH over the push transport runs them without a host read-back
        st = solver.stats()
        hist.append((st["iter_pcisph"], st["err_pcisph"], st["iter_cg"], st["err_cg"]))
    else:
        for _ in range(steps):
            solver.step()
            st = solver.stats()
            hist.append((st["iter_pcisph"], st["err_pcisph"], st["iter_cg"], st["err_cg"]))
    e = container.engine
    g = e.download(L.F_GHOST) == 1
    info = e.comm_get_slab()
    np.savez(out, ids=e.download(L.F_PARTICLE_ID)[~g], pos=e.download(L.F_POSITION)[~g], vel=e.download(L.F_VELOCITY)[~g],
             rho=e.download(L.F_DENSITY)[~g], prs=e.download(L.F_PRESSURE)[~g], n_ghost=info["n_ghost"], cuts=np.array(cuts),
             pairs=solver.stats()["pair_interactions"], iter_density=solver.stats()["iter_density"],
             iter_divergence=solver.stats()["iter_divergence"], iter_pcisph=solver.stats()["iter_pcisph"], hist=np.array(hist, np.float64),
             err_pcisph=solver.stats()["err_pcisph"], z_lo=info["z_lo"], z_hi=info["z_hi"], transport=e.comm_transport())
    print(f"rank {rank}: slab {info['z_lo']}..{info['z_hi']} owned {info['n_owned']} ghosts {info['n_ghost']}")


if __name__ == "__main__":
    main()
