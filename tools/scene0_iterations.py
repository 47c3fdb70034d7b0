"""Why does the reference's own 1.23 M scene (data/scenes/final_scene0.json: the C2 fluid block + sampled domain box + two static
dragons, DFSPH with its stop tests) cost ~68 ms/step here when C3 (the same block, no box, 2+2 fixed iterations) costs ~1.2 ms?


    python tools/scene0_iterations.py [--steps 60] [--scale-steps 12] [--scale 0.25] > gpurun_out/scene0_iterations.json

1. Full size, the scene file's numbers without the two mesh bodies (they need the reference's data/models, which do not travel):
   per step the DFSPH iteration counts (divergence loop DFSPH.py:139-159, density loop :225-243), the wall time and the time per
   solver iteration.
2. A scaled copy (every length x `scale`) through BOTH the HIP path and the CPU oracle with the reference's stop tests: the
   iteration histories must agree (+-1, reduction order: SURVEY 8c).  If they do, the cost is the reference's algorithm on this
   scene -- a block whose outer lattice planes start inside the support of the boundary particles is compressed from step 0 and
   the density loop needs hundreds of iterations to push the MEAN density error under 1e-4 -- not a defect of a path no fixture
   covers.  (The oracle is the checker here; nothing of it is on the product path.)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def scene0(scale=1.0):
    """final_scene0.json (Configuration + FluidBlocks verbatim; RigidBodies left out).  scale < 1: the block's extents and the free
    space beside and above it shrink, its distances to the walls it starts next to (0.09 in x, 0.2 in y, 0.2 on both sides in z) do
    not -- those are what decides how hard the boundary particles compress it."""
    s = scale
    return {
        "Configuration": {
            "domainStart": [0.0, 0.0, 0.0], "domainEnd": [0.09 + (1.61 + 6.8) * s, 0.2 + (3.8 + 4.0) * s, 0.2 + 1.6 * s + 0.2], "addDomainBox": True, "particleRadius": 0.01,
            "density0": 1000, "gravitation": [0.0, -9.81, 0.0], "simulationMethod": "dfsph", "viscosityMethod": "standard",
            "timeStepSize": 0.0006, "viscosity": 10.0, "viscosity_b": 0.3,
        },
        "FluidBlocks": [{
            "objectId": 0, "start": [0.09, 0.2, 0.2], "end": [0.09 + 1.61 * s, 0.2 + 3.8 * s, 0.2 + 1.6 * s], "translation": [0.0, 0.0, 0.0],
            "scale": [1, 1, 1], "velocity": [0.0, -0.5, 0.0], "density": 1000.0, "color": [50, 100, 200], "entryTime": -1.0,
        }],
    }


def main():
    sys.stdout = sys.stderr   # the containers print ("No rigid body in the scene ..."); stdout carries the JSON only
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--scale", type=float, default=0.25)
    ap.add_argument("--scale-steps", type=int, default=12)
    ap.add_argument("--skip-full", action="store_true")
    ap.add_argument("--scene-file", default=None,
                    help="manual check: a scene file of the reference (needs an untracked copy of its data/ directory; run from its "
                         "parent): per-step iteration counts and the per-kernel HIP-event table, nothing else")
    args = ap.parse_args()
    from sph_project_amd import product as P
    out = {}
    if args.scene_file:
        cfg = json.load(open(args.scene_file))
        container, solver = P.build_product(cfg, fast_math=1)
        solver.prepare()
        eng = container.engine
        names = [eng.lib.sph_kernel_name(k).decode() for k in range(19)]
        rows = []
        eng.profile_enable(-1, True); eng.profile_reset()
        for _ in range(args.steps):
            eng.synchronize(); t0 = time.perf_counter()
            solver.step()
            st = solver.stats()
            rows.append({"iter_divergence": int(st["iter_divergence"]), "iter_density": int(st["iter_density"]), "ms": 1e3 * (time.perf_counter() - t0),
                         "err_density": float(st["err_density"]), "err_divergence": float(st["err_divergence"]),
                         "lds_fallback_blocks": int(st["lds_fallback_blocks"])})
        table = {names[k]: eng.profile_read(k) for k in range(19)}
        n_it = sum(r["iter_divergence"] + r["iter_density"] for r in rows)
        print(json.dumps({"scene": args.scene_file, "particles": int(container.particle_num[None]), "fluid_particles": int(container.fluid_particle_num[None]),
                          "steps": args.steps, "ms_per_step": sum(r["ms"] for r in rows) / args.steps, "solver_iterations_per_step": n_it / args.steps,
                          "kernels_ms_per_step": {k: [v[0] / args.steps, round(v[1] / args.steps, 4)] for k, v in table.items() if v[0]},
                          "per_step": rows}), file=sys.__stdout__)
        return
    if not args.skip_full:
        container, solver = P.build_product(scene0(1.0), fast_math=1)
        solver.prepare()
        eng = container.engine
        rows = []
        for _ in range(args.steps):
            eng.synchronize(); t0 = time.perf_counter()
            solver.step()
            st = solver.stats()
            dt = time.perf_counter() - t0
            rows.append({"iter_divergence": int(st["iter_divergence"]), "iter_density": int(st["iter_density"]), "ms": 1e3 * dt,
                         "err_density": float(st["err_density"])})
        n_it = sum(r["iter_divergence"] + r["iter_density"] for r in rows)
        out["full_size"] = {"particles": int(container.particle_num[None]), "fluid_particles": int(container.fluid_particle_num[None]),
                            "steps": args.steps, "ms_per_step": sum(r["ms"] for r in rows) / args.steps,
                            "solver_iterations_per_step": n_it / args.steps, "ms_per_solver_iteration": sum(r["ms"] for r in rows) / max(n_it, 1),
                            "per_step": rows}
        eng.close()
    # scaled copy: HIP vs the oracle, the reference's stop tests on both sides
    from tests import helpers as H   # the checker (test infrastructure)
    cfg = scene0(args.scale)
    container, solver = P.build_product(cfg, fast_math=0)
    solver.prepare()
    ref = H.build_oracle(cfg)
    ref.prepare()
    import numpy as np
    from sph_project_amd import _lib as L
    hip_hist, ref_hist, drift_hist, err_hist = [], [], [], []
    t_ref = 0.0
    for _ in range(args.scale_steps):
        solver.step()
        st = solver.stats()
        hip_hist.append((int(st["iter_divergence"]), int(st["iter_density"])))
        t0 = time.perf_counter()
        ref.step(1)
        t_ref += time.perf_counter() - t0
        ref_hist.append((int(ref.scalar("last_iter_div")), int(ref.scalar("last_iter_den"))))
        # how far apart are the two states?  (an iteration count is a threshold crossing of a mean error that creeps towards the
        # threshold over tens of iterations: a last-bit difference in the error can move the crossing by several iterations)
        e = container.engine
        ids = e.download(L.F_PARTICLE_ID)
        x = H.by_id(ids, e.download(L.F_POSITION))
        xr = H.by_id(H.oracle_ids(ref), ref.field("particle_positions").copy())
        drift_hist.append(float(H.drift(x, xr, container.dh).max()))
        err_hist.append((float(st["err_divergence"]), float(ref.scalar("last_err_div")), float(st["err_density"]), float(ref.scalar("last_err_den"))))
    worst = max(max(abs(a[0] - b[0]), abs(a[1] - b[1])) for a, b in zip(hip_hist, ref_hist))
    out["scaled_copy"] = {"scale": args.scale, "particles": int(container.particle_num[None]), "fluid_particles": int(container.fluid_particle_num[None]),
                          "steps": args.scale_steps, "hip_iterations_div_den": hip_hist, "oracle_iterations_div_den": ref_hist,
                          "max_difference": worst, "drift_vs_oracle_per_step": drift_hist,
                          "final_errors_hip_oracle_div_den": err_hist, "oracle_seconds": t_ref}
    print(json.dumps(out), file=sys.__stdout__)


if __name__ == "__main__":
    main()
