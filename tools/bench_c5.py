"""SURVEY 8(d) configuration C5: the reference's buckling scene (data/scenes/final_scene3.json: DFSPH + implicit
viscosity, mu = mu_b = 1800, dt 1e-3, emitter above y = 2.5, 38 x 560 x 5 fluid sheet inside a 4 x 20 x 8 domain box of
~2.07 M static boundary particles, G = 10 M cells) without its mesh rigid body (needs trimesh).  True convergence
criteria (host read-back per iteration, like the reference); reports ms/step, CG iterations/step, time per CG iteration.
    python tools/bench_c5.py [--steps 20] [--warmup 3]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-events", action="store_true", help="no per-kernel HIP events (they serialise the stream a little)")
    args = ap.parse_args()
    from sph_project_amd import product as P
    cfg = P.c5_scene()
    container, solver = P.build_product(cfg, fast_math=1)
    eng = container.engine
    t0 = time.perf_counter()
    solver.prepare()
    t_prep = time.perf_counter() - t0
    names = [eng.lib.sph_kernel_name(k).decode() for k in range(19)]
    for _ in range(args.warmup):
        solver.step()
    eng.profile_enable(-1, not args.no_events)
    eng.profile_reset()
    eng.synchronize()
    iters = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        solver.step()
        st = solver.stats()
        iters.append((st["iter_cg"], st["iter_density"], st["iter_divergence"]))
    eng.synchronize()
    elapsed = time.perf_counter() - t0
    table = {names[k]: eng.profile_read(k) for k in range(19)}
    table = {k: v for k, v in table.items() if v[0] > 0}
    n_cg = sum(i[0] for i in iters)
    cg_ms = table.get("cg_ap", (0, 0.0))[1] + table.get("cg_vector", (0, 0.0))[1]
    out = {
        "config": "C5 buckling sheet (final_scene3 without its mesh body): dfsph + implicit viscosity",
        "particles": int(container.particle_num[None]), "fluid_particles": int(container.fluid_particle_num[None]),
        "grid_cells": int(container.grid_num.prod()), "steps": args.steps, "ms_per_step": 1e3 * elapsed / args.steps,
        "prepare_s": t_prep,
        "cg_iterations_per_step": n_cg / args.steps, "ms_per_cg_iteration": cg_ms / max(n_cg, 1),
        "dfsph_density_iterations_per_step": sum(i[1] for i in iters) / args.steps,
        "dfsph_divergence_iterations_per_step": sum(i[2] for i in iters) / args.steps,
        "kernels_ms_per_step": {k: round(v[1] / args.steps, 4) for k, v in sorted(table.items(), key=lambda kv: -kv[1][1])},
        "math": "fast", "note": "synchronous steps with the reference's convergence tests (one 4-byte read-back per solver iteration)",
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
