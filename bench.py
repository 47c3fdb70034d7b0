#!/usr/bin/env python
"""bench.py -- headline benchmark of the SPH hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one solver.step() of the reference (neighbour search + density/EOS + non-pressure
forces + pressure forces + symplectic integration + boundary) over the whole particle set.
N=1 workload = BASELINE.json configs[1] (SURVEY 8d "C2"): 1,231,200-particle dam break, WCSPH,
f32, synthetic lattice (no RNG).  Prints ONE JSON line on rank 0.

Timing: W untimed warm-up steps, then exactly K steps enqueued on the library's HIP stream between
two (barrier + device synchronise) fences; max over ranks.  `value` = fluid particles advanced per
second by the whole job with all state resident in HBM (the scene upload is outside the region).
Roofline leg: the dominant kernel (picked by a short all-kernel HIP-event pre-pass) is timed with
HIP events on the library's own stream *inside* the timed region; achieved GB/s = algorithmic
bytes per launch / average launch duration, against 8 TB/s HBM3E peak.
CPU baseline leg (rank 0, N=1 only): the oracle (this repo's C restatement of the reference
algorithm, OpenMP) timed on the host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy rate)

# Algorithmic HBM bytes per particle per launch (DESIGN.md "Kernels").
ALG_BYTES = {
    "hash_count": 16, "scatter": 60, "density": 24, "non_pressure": 44, "pressure_integrate": 60, "wcsph_forces": 96,
    "dfsph_density_alpha": 24, "dfsph_rho_adv": 36, "dfsph_correct": 48,
    "pcisph_rho_star": 40, "pcisph_pressure_accel": 64,
}


def c2_scene(method="wcsph", scale_z=1):
    """block identical to data/scenes/final_scene0.json:55-59 of the reference.
    scale_z = N (weak scaling over N GPUs): the block and the domain are N times as deep in z, i.e. N x 80 lattice
    planes = N x 1,231,200 particles, so every z-slab holds one C2's worth of work."""
    dt = 6e-4 if method == "dfsph" else 4e-4
    return {
        "Configuration": {
            "domainStart": [0.0, 0.0, 0.0], "domainEnd": [8.5, 8.0, 0.4 + 1.6 * scale_z], "addDomainBox": False,
            "particleRadius": 0.01, "density0": 1000, "simulationMethod": method, "viscosityMethod": "standard",
            "gravitation": [0.0, -9.81, 0.0], "timeStepSize": dt, "viscosity": 10.0,
        },
        "FluidBlocks": [{
            "objectId": 0, "start": [0.09, 0.2, 0.2], "end": [1.7, 4.0, 0.2 + 1.6 * scale_z],
            "translation": [0.0, 0.0, 0.0], "scale": [1, 1, 1], "velocity": [0.0, -0.5, 0.0], "density": 1000.0,
            "color": [50, 100, 200], "entryTime": -1.0,
        }],
    }


def c4_scene(method="wcsph"):
    """100 x 250 x 160 = 4,000,000 particles; the block spans the full z extent, so z-slabs stay
    balanced while the dam breaks along x.  One fixed scene: sharding it over N GPUs is strong scaling."""
    return {
        "Configuration": {
            "domainStart": [0.0, 0.0, 0.0], "domainEnd": [6.0, 6.0, 3.36], "addDomainBox": False,
            "particleRadius": 0.01, "density0": 1000, "simulationMethod": method, "viscosityMethod": "standard",
            "gravitation": [0.0, -9.81, 0.0], "timeStepSize": 4e-4, "viscosity": 10.0,
        },
        "FluidBlocks": [{
            "objectId": 0, "start": [0.1, 0.1, 0.08], "end": [2.1, 5.1, 3.28], "translation": [0.0, 0.0, 0.0],
            "scale": [1, 1, 1], "velocity": [0.0, -0.5, 0.0], "density": 1000.0, "color": [50, 100, 200],
            "entryTime": -1.0,
        }],
    }


def c1_scene(method="wcsph"):
    from tests import helpers as H
    return H.dam_break_scene(method=method)


def cpu_baseline(cfg, steps, threads):
    """Oracle (kind "port") timed on the host.  Test infrastructure used as the *measured baseline
    only*; nothing of it is on the product path."""
    os.environ["OMP_NUM_THREADS"] = str(threads)
    from tests import helpers as H
    sim = H.build_oracle(cfg, fixed_iterations=2 if cfg["Configuration"]["simulationMethod"] != "wcsph" else 0)
    sim.prepare()
    sim.step(1)  # warm-up (page faults, thread pool)
    t0 = time.perf_counter()
    sim.step(steps)
    dt = time.perf_counter() - t0
    n = sim.fluid_particle_num
    pairs = sim.last_pairs
    sim.close()
    return n * steps / dt, pairs / (dt / steps), dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c2", choices=["c1", "c2", "c3", "c4"])
    ap.add_argument("--method", default=None, choices=["wcsph", "dfsph", "pcisph"],
                    help="override the solver of the chosen scene (pcisph: no configuration of its own in BASELINE.json)")
    ap.add_argument("--strict-math", action="store_true", help="IEEE div/sqrt build instead of the fast build")
    ap.add_argument("--no-deterministic", action="store_true")
    ap.add_argument("--force-global", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--measured-iterations", action="store_true",
                    help="c3: the solver's own convergence tests (DFSPH.py:150/:239) instead of 2+2 fixed iterations; "
                         "iteration counts of the last step are reported in config")
    ap.add_argument("--presteps", type=int, default=0,
                    help="untimed steps before the warm-up (tuning aid: time the passes on a dam break in motion instead of "
                         "the rest lattice; the headline number is always quoted with 0)")
    ap.add_argument("--all-kernels", action="store_true", help="also print the per-kernel HIP-event table (stderr)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N>1: weak = every z-slab gets one C2 block (N x 1.23 M particles); strong = the 1.23 M scene is split")
    ap.add_argument("--replicas", action="store_true", help="N>1: independent copies instead of z-slab sharding (debug)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    use_gloo = bool(os.environ.get("SPH_BENCH_GLOO"))  # test rig: several ranks on ONE GPU (with SPH_COMM_TRANSPORT=shm)
    if world > 1:
        import torch
        import torch.distributed as dist
        if use_gloo:
            local_rank = 0
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    def _reduce(x, op, dtype):
        import torch
        t = torch.tensor([x], device="cpu" if use_gloo else "cuda", dtype=dtype)
        dist.all_reduce(t, op=op)
        return t.item()

    method = args.method or ("dfsph" if args.config == "c3" else "wcsph")
    sharded = world > 1 and not args.replicas and method == "wcsph"
    scale_z = world if (sharded and args.scaling == "weak" and args.config == "c2") else 1
    cfg = c1_scene(method) if args.config == "c1" else (c4_scene(method) if args.config == "c4" else c2_scene(method, scale_z=scale_z))
    from tests import helpers as H  # scene -> container/solver exactly like run_simulation.py
    slab_opt = None
    n_global = None
    if sharded:
        import numpy as np
        from sph_project_amd import _lib as L, slab
        import ctypes
        uid = [None]
        if rank == 0:
            buf = ctypes.create_string_buffer(128)
            assert L.load().sph_comm_unique_id(buf) == 0
            uid[0] = buf.raw
        dist.broadcast_object_list(uid, src=0)
        _, geo, batches = H.scene_particles(cfg)
        z = np.concatenate([b["pos"][:, 2] for b in batches])
        nz = int(geo.grid_num[2])
        cuts = slab.plan_slabs(np.bincount(slab.cell_layer(z, geo.dh, nz), minlength=nz), world)
        slab_opt = dict(rank=rank, nranks=world, unique_id=uid[0], cuts=cuts)
        n_global = int(sum((b["material"] == 1).sum() for b in batches))
        del batches, z
    if world == 1 and os.environ.get("SPH_BENCH_FORCE_SLAB") and method == "wcsph":
        # tuning aid: ONE rank in slab mode (classify / tables / ghost-aware kernels, no neighbour to talk to)
        import numpy as np
        _, geo, _b = H.scene_particles(cfg)
        os.environ.setdefault("SPH_COMM_TRANSPORT", "shm")   # SPH_COMM_TRANSPORT=rccl: a one-rank RCCL communicator instead
        uid = b"\0" * 128
        if os.environ["SPH_COMM_TRANSPORT"] != "shm":
            import ctypes
            from sph_project_amd import _lib as L
            buf = ctypes.create_string_buffer(128)
            assert L.load().sph_comm_unique_id(buf) == 0
            uid = buf.raw
        slab_opt = dict(rank=0, nranks=1, unique_id=uid, cuts=[0, int(geo.grid_num[2])])
        n_global = int(sum((b["material"] == 1).sum() for b in _b))
        del _b
    opts = dict(fast_math=0 if args.strict_math else 1, deterministic=0 if args.no_deterministic else 1,
                force_global=int(os.environ.get('SPH_DEBUG_MODE', int(args.force_global))), device=local_rank if world > 1 else -1)
    if method != "wcsph" and not args.measured_iterations:
        opts["fixed_iterations"] = 2
    container = solver = None
    if slab_opt and world == 1:
        container, solver = H.build_product(cfg, slab=slab_opt, **opts)
    elif slab_opt:
        # every rank must agree on the mode: if the communicator cannot be set up anywhere, all fall back to replicas
        import torch
        err = ""
        try:
            container, solver = H.build_product(cfg, slab=slab_opt, **opts)
        except Exception as exc:  # noqa: BLE001
            err = f"{type(exc).__name__}: {exc}"
        if _reduce(0 if err else 1, dist.ReduceOp.MIN, torch.int64) == 0:
            if rank == 0 or err:
                print(f"[bench] rank {rank}: slab sharding unavailable ({err or 'failed on another rank'}); running replicas", file=sys.stderr)
            container = solver = None
            sharded = False
            scale_z = 1
            cfg = c1_scene(method) if args.config == "c1" else (c4_scene(method) if args.config == "c4" else c2_scene(method))
    if container is None:
        container, solver = H.build_product(cfg, **opts)
    eng = container.engine
    solver.prepare()
    n_fluid = container.fluid_particle_num[None]
    names = [eng.lib.sph_kernel_name(k).decode() for k in range(19)]

    def fence():
        eng.synchronize()
        if dist is not None:
            import torch
            dist.barrier()
            if not use_gloo:
                torch.cuda.synchronize()

    run = eng.step if args.measured_iterations else eng.step_async   # convergence tests need the (batched) flag read-back
    if args.presteps:
        run(args.presteps)
    run(args.warmup)
    fence()

    # pre-pass: which kernel dominates?  (all-kernel events perturb the stream slightly -> untimed)
    eng.profile_enable(-1, True)
    eng.profile_reset()
    run(5)
    eng.synchronize()
    table = {names[k]: eng.profile_read(k) for k in range(19)}
    table = {k: v for k, v in table.items() if v[0] > 0}
    dom = max((k for k in table if k in ALG_BYTES), key=lambda k: table[k][1])
    if args.all_kernels and rank == 0:
        for k, (n, ms) in sorted(table.items(), key=lambda kv: -kv[1][1]):
            print(f"  {k:24s} launches {n:5d}  avg {1e3 * ms / n:9.1f} us", file=sys.stderr)
    eng.profile_enable(-1, False)
    eng.profile_enable(names.index(dom), not os.environ.get('SPH_BENCH_NO_EVENTS'))
    eng.profile_reset()

    fence()
    t0 = time.perf_counter()
    run(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        elapsed = float(_reduce(elapsed, dist.ReduceOp.MAX, torch.float64))

    launches, ms = eng.profile_read(names.index(dom))
    stats = solver.stats()
    pairs = stats["pair_interactions"]
    if sharded:
        n_total = n_global  # every fluid particle is owned by exactly one rank
        import torch
        pairs = int(_reduce(pairs, dist.ReduceOp.SUM, torch.int64))
        n_fluid = n_global // world  # per-launch share for the roofline leg
    else:
        n_total = n_fluid * world  # replicas: every rank advances its own copy
        pairs = pairs * world
    value = n_total * args.steps / elapsed
    avg_s = (ms / max(launches, 1)) * 1e-3
    achieved = ALG_BYTES[dom] * n_fluid / avg_s / 1e9 if launches else None
    traffic = None
    tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tf):
        try:
            traffic = json.load(open(tf)).get(args.config, {}).get(dom)
        except Exception:
            traffic = None

    out = {
        "metric": "particle-updates/sec", "value": value, "unit": "particle-updates/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": ("strong" if args.config == "c4" else args.scaling) if sharded else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": {"c1": "C1 8,000-particle cube dam break", "c2": "C2 1,231,200-particle dam break" + (f" x{scale_z} in z" if scale_z > 1 else ""),
                         "c4": "C4 4,000,000-particle dam break",
                         "c3": "C3 1,231,200-particle dam break, " + ("DFSPH iterations as measured" if args.measured_iterations else "2+2 fixed DFSPH iterations")}[args.config],
            "method": method, "particles": int(n_total), "grid_cells": int(container.grid_num.prod()),
            "dt": cfg["Configuration"]["timeStepSize"], "math": "strict" if args.strict_math else "fast",
            "deterministic_sort": not args.no_deterministic,
            "parallelism": "single-gpu" if world == 1 else
                           (f"z-slab x{world}, RCCL halo exchange ({'strong' if args.config == 'c4' else args.scaling} scaling)" if sharded else f"replicas x{world}"),
            "pair_interactions_per_step": int(pairs),
            "pair_interactions_per_s": pairs * args.steps / elapsed,
            "lds_fallback_blocks_last_step": int(stats["lds_fallback_blocks"]),
            **({"solver_iterations_last_step": {"density": int(stats["iter_density"]), "divergence": int(stats["iter_divergence"])}} if method == "dfsph" else {}),
            "device": eng.device_info()["name"],
        },
        "roofline": {
            "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
            "launches": int(launches), "avg_launch_us": 1e6 * avg_s,
            "alg_bytes_per_launch": ALG_BYTES[dom] * n_fluid,
            "step_achieved": (204 * n_fluid + 12 * int(container.grid_num.prod())) / (elapsed / args.steps) / 1e9,
            "note": "neighbour passes are VALU/LDS-bound, not HBM-bound (DESIGN.md)",
        },
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        threads = os.cpu_count() or 1
        v, pairs_s, secs = cpu_baseline(cfg, args.cpu_steps, threads)
        out["cpu_baseline"] = {
            "value": v, "unit": "particle-updates/s", "cores": threads, "kind": "port",
            "sample": f"{args.cpu_steps} steps of the same workload ({secs:.1f} s), oracle/sph_ref.c "
                      f"(this repo's C restatement of the reference algorithm, OpenMP), not Taichi",
            "pair_interactions_per_s": pairs_s,
        }
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
