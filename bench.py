#!/usr/bin/env python
"""bench.py -- headline benchmark of the SPH hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one solver.step() of the reference (neighbour search + density/EOS + non-pressure forces + pressure
forces + symplectic integration + boundary) over the whole particle set.  N=1 workload = BASELINE.json configs[1]
(SURVEY 8d "C2"): 1,231,200-particle dam break, WCSPH, f32, synthetic lattice (no RNG).  Rank 0 prints ONE JSON line.

N > 1: one process per GPU, no torch anywhere.  `python bench.py --gpus N` spawns its own N ranks; under a launcher
that already did (torch.distributed.run exports RANK / LOCAL_RANK / WORLD_SIZE) every process is one rank.  The RCCL
unique id travels through a file in /dev/shm, barriers and the max / sum reductions are RCCL all-reduces behind the
C-ABI (sph_comm_barrier / sph_comm_allreduce).  Scenes are z-slab sharded (RCCL halo exchange; DFSPH / PCISPH with their
per-iteration ghost refreshes and all-reduced residuals); --replicas runs independent copies instead and says so.  Halo payload travels
by device stores into the neighbour's hipIpc-mapped inbox ("push", self-tested at set-up) with RCCL send/recv as the all-ranks fall-back;
SPH_COMM_TRANSPORT=shm+ipc / shm lets several ranks share one GPU (test rig).

Timing: W untimed warm-up steps, then 3 repetitions of exactly K steps, each enqueued on the library's HIP stream
between two (device synchronise + barrier) fences, max over ranks; `ms_per_step` is the MEDIAN repetition
(`repeat_ms_per_step` lists all three; from rest the lattice relaxes, so later repetitions are 1-2 % faster).  The
device synchronise is sph_synchronize: a one-wave kernel behind the queued steps publishes a number into pinned host
memory and the host spins on it (it returns ~30 us sooner than hipStreamSynchronize's interrupt; SPH_SLOW_SYNC=1).  `value` = fluid particles advanced per second by the whole job with all state
resident in HBM (scene upload outside the region).  The same scene is then advanced to step 2500 and timed again
(`in_motion`): the rest lattice holds 29 neighbours per particle, the collapsing column ~39.
Roofline leg: the dominant kernel (picked by a short all-kernel HIP-event pre-pass) is timed with HIP events on the
library's own stream inside the timed region; achieved GB/s = algorithmic bytes per launch /
average launch duration, against the 8 TB/s HBM3E spec AND against the device-to-device copy rate measured in this run
(`measured_copy_gbs`).  `traffic` and `secondary` (what actually bounds the kernel: VALU issue, parked wave-cycles, LDS,
effective clock) come from the committed PMC passes of this command under profiles/ and say so (`traffic_source`).
CPU baseline leg (rank 0, N=1 only): the oracle (this repo's C restatement of the reference algorithm, OpenMP) on
the host cores, a bounded sample of the same workload, swept over thread counts; the best is reported.
Extra objects of the line (never `value`): N = 1: `extras.c3` (BASELINE configs[2]: DFSPH, 2+2 iterations, per-walk
microseconds, 92 B/particle/iteration roofline) and `extras.c5` (configs[4]: the implicit-viscosity buckling scene with the
solvers' stop tests, CG iterations per step, microseconds per CG iteration, 224 B/particle/iteration roofline); N > 1:
the headline itself is the metric as BASELINE.json words it -- the 1.23 M scene split over the N ranks (`scaling: strong`);
`c2_weak_scaling` (one C2 block per rank) and `c4_strong_scaling` (configs[3]) follow on communicators of their own, with the
halo transport in effect.
"""
import argparse
import json
import os
import subprocess
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
N_KERNEL_IDS = 19


def parse_args(argv=None):
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
    ap.add_argument("--cpu-steps", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=3, help="repetitions of the K timed steps (median reported)")
    ap.add_argument("--measured-iterations", action="store_true",
                    help="c3: the solver's own convergence tests (DFSPH.py:150/:239) instead of 2+2 fixed iterations; "
                         "iteration counts of the last step are reported in config")
    ap.add_argument("--presteps", type=int, default=0,
                    help="untimed steps before the warm-up (the headline `value` is always quoted with 0)")
    ap.add_argument("--motion-step", type=int, default=2500,
                    help="after the headline measurement advance the scene to this step and time again (`in_motion`); 0: skip")
    ap.add_argument("--all-kernels", action="store_true", help="also print the per-kernel HIP-event table (stderr)")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="N>1: strong (default) = BASELINE.json's metric as written: the 1.23 M scene itself split over the N ranks; "
                         "weak = every z-slab gets one C2 block (N x 1.23 M particles) -- without this flag weak scaling is the extra "
                         "object `c2_weak_scaling` of the line")
    ap.add_argument("--replicas", action="store_true", help="N>1: independent copies instead of z-slab sharding")
    ap.add_argument("--no-c4", action="store_true", help="N>1: skip the extra C4 (4 M particles, strong scaling) measurement")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra objects of the line: N=1: C3 (DFSPH) and C5 (implicit-viscosity buckling scene); "
                         "N>1: the 1.23 M scene and C4 under strong scaling")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------- launcher
def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script, relay rank 0's line."""
    rdv = f"/dev/shm/sph_bench_{os.getpid()}_{int(time.time() * 1e6)}.id"
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), SPH_BENCH_RDV=rdv)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # a rank that dies (capacity error, HIP error) leaves its neighbours blocked in a receive: stop them (these exact
    # children, by pid) as soon as any rank exits non-zero, instead of waiting for ever
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    while True:
        rcs = [p.poll() for p in procs]
        if all(rc is not None for rc in rcs):
            break
        if any(rc not in (None, 0) for rc in rcs):
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.05)
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10)
    out = b"".join(c for c in chunks if c)
    try:
        os.unlink(rdv)
    except OSError:
        pass
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    if any(rcs):
        print(f"[bench] rank exit codes {rcs}", file=sys.stderr)
        sys.exit(1)


def rendezvous_path():
    p = os.environ.get("SPH_BENCH_RDV")
    if p:
        return p
    # ranks started by one launcher share their parent; its start time makes the name unique across runs
    ppid = os.getppid()
    try:
        start = open(f"/proc/{ppid}/stat").read().rsplit(")", 1)[1].split()[19]
    except Exception:  # noqa: BLE001
        start = "0"
    return f"/dev/shm/sph_bench_{ppid}_{start}_{os.environ.get('MASTER_PORT', '0')}.id"


def exchange_unique_id(lib, rank, suffix=""):
    import ctypes
    path = rendezvous_path() + suffix
    if rank == 0:
        buf = ctypes.create_string_buffer(128)
        if lib.sph_comm_unique_id(buf) != 0:
            raise RuntimeError("sph_comm_unique_id failed")
        tmp = path + ".tmp"
        with open(tmp, "wb") as f:
            f.write(buf.raw)
        os.rename(tmp, path)
        return buf.raw
    t0 = time.time()
    while True:
        try:
            data = open(path, "rb").read()
            if len(data) == 128:
                return data
        except OSError:
            pass
        if time.time() - t0 > 120:
            raise RuntimeError(f"rank {rank}: no unique id at {path} after 120 s")
        time.sleep(0.01)


# ----------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(cfg, steps):
    """Oracle (kind "port") timed on the host.  Test infrastructure used as the *measured baseline only*; nothing of
    it is on the product path.  Thread counts are swept and the best one is reported, with the speed-up over one thread.
    Threads are bound (OMP_PROC_BIND=spread over OMP_PLACES=cores, set before libgomp is loaded) and the oracle first-touches
    its arrays page-interleaved over the threads, so that a two-socket box is not limited by the memory of one node."""
    import ctypes
    # hardware threads this PROCESS may run on, read before libgomp exists: OMP_PROC_BIND binds the calling thread to its own place
    # (one core = 2 hardware threads) as soon as the runtime starts, after which sched_getaffinity(0) of this thread says 2 -- the
    # figure rounds 4-5 printed as `affinity_threads` beside a 14x speed-up
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = None
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    os.environ.setdefault("OMP_PLACES", "cores")
    from tests import helpers as H
    ncpu = os.cpu_count() or 1
    sim = H.build_oracle(cfg, fixed_iterations=2 if cfg["Configuration"]["simulationMethod"] != "wcsph" else 0)
    omp = ctypes.CDLL("libgomp.so.1")
    sim.prepare()
    sim.step(1)  # warm-up (page faults, thread pool)
    sweep = sorted({t for t in (1, 8, 16, 32, 64, ncpu // 2, ncpu) if 1 <= t <= ncpu})
    best = None
    table = {}
    for t in sweep:
        omp.omp_set_num_threads(int(t))
        k = 1 if t == 1 else steps   # (one thread: one step, ~2 s at C2)
        t0 = time.perf_counter()
        sim.step(k)
        dt = time.perf_counter() - t0
        table[t] = sim.fluid_particle_num * k / dt
        if best is None or table[t] > table[best]:
            best = t
        best_dt = dt if best == t else best_dt
    steps = 1 if best == 1 else steps
    n, pairs = sim.fluid_particle_num, sim.last_pairs
    sim.close()
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    # how many of those hardware threads may this process actually use?  (a container with a CPU quota reports all of the host's threads
    # in os.cpu_count(); a sweep that peaks at 16-32 threads and collapses beyond is the quota's throttling, not the code's scaling)
    quota = None
    try:
        a, b_ = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if a == "max" else float(a) / float(b_)
    except (OSError, ValueError):
        pass
    return dict(value=table[best], cores=best, secs=best_dt, pairs_per_s=pairs * steps / best_dt, sweep=table, model=model, ncpu=ncpu,
                cpu_quota_cores=quota, affinity_threads=affinity,
                speedup_over_1_thread=table[best] / table[1] if 1 in table else None)


# ----------------------------------------------------------------------------------------------- fixed scenes on N GPUs
def sharded_extra(args, rank, world, device, lib, cfg, workload, suffix, scaling="strong"):
    """One more scene z-slab sharded over the N ranks of this job, measured after the headline with a communicator of its
    own and reported as an extra object of the JSON line, never as `value`:
      * `c4_strong_scaling`: BASELINE configs[3], the 4,000,000-particle WCSPH dam break;
      * `c2_weak_scaling`: one C2 block per rank (N x 1,231,200 particles) -- a workload BASELINE.json does not name;
      * `c2_strong_scaling` (only with --scaling weak, where the headline is the weak one): the C2 scene itself."""
    import numpy as np
    from sph_project_amd import product as P, slab
    uid = exchange_unique_id(lib, rank, suffix=suffix)
    _, geo, batches = P.scene_particles(cfg)
    z = np.concatenate([b["pos"][:, 2] for b in batches])
    nz = int(geo.grid_num[2])
    cuts = slab.plan_slabs(np.bincount(slab.cell_layer(z, geo.dh, nz), minlength=nz), world)
    n_global = int(sum((b["material"] == 1).sum() for b in batches))
    del batches, z
    container, solver = P.build_product(cfg, fast_math=0 if args.strict_math else 1, deterministic=0 if args.no_deterministic else 1,
                                        device=device, slab=dict(rank=rank, nranks=world, unique_id=uid, cuts=cuts))
    eng = container.engine
    solver.prepare()
    eng.step_async(args.warmup)
    times = []
    for _ in range(args.repeats):
        eng.synchronize(); eng.comm_barrier()
        t0 = time.perf_counter()
        eng.step_async(args.steps)
        eng.synchronize(); eng.comm_barrier()
        times.append(eng.comm_allreduce([time.perf_counter() - t0], "max")[0])
    el = sorted(times)[len(times) // 2]
    pairs = int(eng.comm_allreduce([solver.stats()["pair_interactions"]], "sum")[0])
    info = eng.comm_get_slab()
    owned = [int(v) for v in eng.comm_allreduce([info["n_owned"] if r == rank else 0 for r in range(min(world, 16))], "sum")]
    transport = eng.comm_transport() if hasattr(eng, "comm_transport") else os.environ.get("SPH_COMM_TRANSPORT", "rccl")
    eng.comm_barrier()
    if rank == 0 and not os.environ.get("SPH_BENCH_RDV"):
        try:
            os.unlink(rendezvous_path() + suffix)
        except OSError:
            pass
    eng.close()
    return {"workload": workload, "scaling": scaling, "particles": n_global, "n_gpus": world,
            "ms_per_step": 1e3 * el / args.steps, "value": n_global * args.steps / el, "unit": "particle-updates/s",
            "pair_interactions_per_s": pairs * args.steps / el, "slab_cuts": [int(c) for c in cuts], "owned_per_rank": owned,
            "halo_transport": transport, "steps": args.steps, "warmup": args.warmup}


# ----------------------------------------------------------------------------------------------- C3 / C5 at N = 1
def pmc_figures(config, kernel):
    """Counter-derived figures of one kernel from profiles/pmc_derived.json (separate rocprofv3 --pmc passes, tools/prof.sh; NOT measured
    in the running process): HBM traffic per launch (2 FETCH_SIZE + WRITE_SIZE) and what actually bounds the kernel."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_derived.json")))
        e = d.get(config, {}).get(kernel)
        if not e:
            return None
        src = d.get(config, {}).get("_source") or d.get("_source")
        return {"traffic": e.get("hbm_bytes_per_launch"), "avg_us_in_that_run": e.get("avg_us"),
                "secondary": {k: e.get(k) for k in ("valu_issue_frac", "waves_parked_frac", "lds_active_frac", "lds_conflict_frac", "eff_clock_ghz")},
                "traffic_source": f"{src}: 2*FETCH_SIZE + WRITE_SIZE per launch from separate rocprofv3 --pmc passes; NOT measured in this run"}
    except Exception:  # noqa: BLE001
        return None


def median(v):
    s_ = sorted(v)
    return s_[len(s_) // 2]


def kernel_table(eng, names):
    t = {names[k]: eng.profile_read(k) for k in range(N_KERNEL_IDS)}
    return {k: v for k, v in t.items() if v[0] > 0}


def extra_c3(args, names):
    """BASELINE configs[2]: the C2 scene with DFSPH (divergence + density solver), 2 + 2 fixed iterations per step (SURVEY 8d
    C3), enqueued without host read-backs; per-pass microseconds from an all-kernel HIP-event pre-pass."""
    from sph_project_amd import product as P
    cfg = P.c2_scene("dfsph")
    container, solver = P.build_product(cfg, fast_math=0 if args.strict_math else 1, fixed_iterations=2,
                                        deterministic=0 if args.no_deterministic else 1)
    eng = container.engine
    solver.prepare()
    n = int(container.fluid_particle_num[None])
    eng.step_async(args.warmup); eng.synchronize()
    eng.profile_enable(-1, True); eng.profile_reset()
    eng.step_async(5); eng.synchronize()
    table = kernel_table(eng, names)
    eng.profile_enable(-1, False)
    reps = []
    for _ in range(args.repeats):
        eng.synchronize(); t0 = time.perf_counter()
        eng.step_async(args.steps); eng.synchronize()
        reps.append(time.perf_counter() - t0)
    el = median(reps)
    st = solver.stats()
    it_div, it_den = int(st["iter_divergence"]), int(st["iter_density"])
    per_pass = {k: {"launches_per_step": v[0] / 5.0, "avg_us": 1e3 * v[1] / v[0]} for k, v in sorted(table.items(), key=lambda kv: -kv[1][1])}
    dom = max((k for k in table if k in ALG_BYTES), key=lambda k: table[k][1])
    avg_s = table[dom][1] / table[dom][0] * 1e-3
    ach = ALG_BYTES[dom] * n / avg_s / 1e9
    # one solver iteration = one correction pass + one rho* / D rho pass (+ their reduction): 92 B/particle
    it_us = sum(1e3 * table[k][1] / table[k][0] for k in ("dfsph_correct", "dfsph_rho_adv") if k in table)
    step_bytes = (16 + 60 + 20 + 20 + 44 + 36 + 92 * (it_div + it_den)) * n + 12 * int(container.grid_num.prod())
    out = {"workload": "C3 1,231,200-particle dam break, DFSPH, 2+2 fixed iterations", "particles": n, "dt": cfg["Configuration"]["timeStepSize"],
           "ms_per_step": 1e3 * el / args.steps, "repeat_ms_per_step": [1e3 * r / args.steps for r in reps],
           "value": n * args.steps / el, "unit": "particle-updates/s", "steps": args.steps, "warmup": args.warmup,
           "solver_iterations_per_step": {"divergence": it_div, "density": it_den},
           "pair_interactions_per_s": st["pair_interactions"] * args.steps / el, "kernels": per_pass,
           "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                        "avg_launch_us": 1e6 * avg_s, "alg_bytes_per_launch": ALG_BYTES[dom] * n,
                        "solver_iteration": {"alg_bytes_per_particle": 92, "us": it_us,
                                             "achieved": 92 * n / (it_us * 1e-6) / 1e9 if it_us else None,
                                             "frac": 92 * n / (it_us * 1e-6) / 1e9 / HBM_PEAK_GBS if it_us else None},
                        "step_achieved": step_bytes / (el / args.steps) / 1e9,
                        "step_alg_bytes": step_bytes,
                        **(pmc_figures("c3", dom) or {"traffic": None, "secondary": None}),
                        "solver_walks": {k: pmc_figures("c3", k) for k in ("dfsph_rho_adv", "dfsph_correct")}}}
    # the same regime once the column collapses (the state a user spends most of a run in)
    c3_motion = int(os.environ.get("SPH_BENCH_C3_MOTION_STEP", "1000"))
    done = args.warmup + 5 + args.repeats * args.steps
    if c3_motion > done:
        eng.step_async(c3_motion - done); eng.synchronize()
        reps = []
        for _ in range(args.repeats):
            eng.synchronize(); t0 = time.perf_counter()
            eng.step_async(args.steps); eng.synchronize()
            reps.append(time.perf_counter() - t0)
        m_el = median(reps)
        st2 = solver.stats()
        out["in_motion"] = {"from_step": c3_motion, "ms_per_step": 1e3 * m_el / args.steps, "value": n * args.steps / m_el,
                            "pair_interactions_per_s": st2["pair_interactions"] * args.steps / m_el}
    eng.close()
    # SURVEY 8d C3 "with iteration counts as measured": the reference's own stop tests (DFSPH.py:139-159, :225-243), steps synchronous
    # like the reference's (one flag read-back per batch of iterations), from rest and from the same in-motion state
    try:
        container, solver = P.build_product(cfg, fast_math=0 if args.strict_math else 1, deterministic=0 if args.no_deterministic else 1)
        eng = container.engine
        solver.prepare()
        k_steps = min(args.steps, 20)

        def measured():
            iters = []
            eng.synchronize(); t0 = time.perf_counter()
            for _ in range(k_steps):
                eng.step(1)
                st = solver.stats()
                iters.append((int(st["iter_divergence"]), int(st["iter_density"])))
            eng.synchronize()
            el = time.perf_counter() - t0
            return {"ms_per_step": 1e3 * el / k_steps, "value": n * k_steps / el, "steps": k_steps,
                    "iterations_per_step": {"divergence": sum(i[0] for i in iters) / k_steps, "density": sum(i[1] for i in iters) / k_steps},
                    "max_iterations_in_a_step": {"divergence": max(i[0] for i in iters), "density": max(i[1] for i in iters)}}
        eng.step(args.warmup)
        m = {"from_rest": measured()}
        done = args.warmup + k_steps
        if c3_motion > done:
            eng.step(c3_motion - done)
            m["in_motion"] = dict(measured(), from_step=c3_motion)
        out["measured"] = dict(m, note="the solvers' own convergence tests instead of 2 + 2 fixed iterations; iteration counts as measured")
        eng.close()
    except Exception as e:  # noqa: BLE001
        out["measured"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def extra_c5(args, names):
    """BASELINE configs[4]: the reference's buckling scene (final_scene3.json without its mesh body): DFSPH + implicit
    viscosity (matrix-free CG over the neighbour graph), 2.17 M particles of which 106,400 fluid, G = 10 M cells; the
    solvers' own stop tests (DFSPH.py:150/:239, base_solver.py:445-449), so steps are synchronous like the reference's."""
    from sph_project_amd import product as P
    cfg = P.c5_scene()
    container, solver = P.build_product(cfg, fast_math=0 if args.strict_math else 1, deterministic=0 if args.no_deterministic else 1)
    eng = container.engine
    solver.prepare()
    n, nf = int(container.particle_num[None]), int(container.fluid_particle_num[None])
    for _ in range(3):
        solver.step()
    eng.profile_enable(-1, True); eng.profile_reset()
    it_pre = 0
    for _ in range(5):
        solver.step(); it_pre += int(solver.stats()["iter_cg"])
    eng.synchronize()
    table = kernel_table(eng, names)
    eng.profile_enable(-1, False)
    k_steps = min(args.steps, 20)
    iters = []
    eng.synchronize(); t0 = time.perf_counter()
    for _ in range(k_steps):
        solver.step()
        st = solver.stats()
        iters.append((int(st["iter_cg"]), int(st["iter_density"]), int(st["iter_divergence"])))
    eng.synchronize()
    el = time.perf_counter() - t0
    n_cg = sum(i[0] for i in iters)
    cg_ms = sum(table[k][1] for k in ("cg_ap", "cg_vector") if k in table)
    it_us = 1e3 * cg_ms / max(it_pre, 1)          # device time of one CG iteration (A p pass + vector updates), event pre-pass
    wall_it_us = None
    per_pass = {k: {"launches_per_step": v[0] / 5.0, "avg_us": 1e3 * v[1] / v[0]} for k, v in sorted(table.items(), key=lambda kv: -kv[1][1])}
    ach = 224 * nf / (it_us * 1e-6) / 1e9 if it_us else None
    out = {"workload": "C5 buckling sheet (final_scene3.json without its mesh body): DFSPH + implicit viscosity, solver stop tests",
           "particles": n, "fluid_particles": nf, "grid_cells": int(container.grid_num.prod()), "dt": cfg["Configuration"]["timeStepSize"],
           "ms_per_step": 1e3 * el / k_steps, "value": nf * k_steps / el, "unit": "fluid particle-updates/s", "steps": k_steps, "warmup": 8,
           "cg_iterations_per_step": n_cg / k_steps, "us_per_cg_iteration": it_us,
           "dfsph_iterations_per_step": {"density": sum(i[1] for i in iters) / k_steps, "divergence": sum(i[2] for i in iters) / k_steps},
           "kernels": per_pass,
           "roofline": {"bound": "hbm", "kernel": "cg iteration (cg_ap + cg_vector)", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS if ach else None, "alg_bytes_per_launch": 224 * nf, "avg_launch_us": it_us,
                        **(pmc_figures("c5", "cg_ap") or {"traffic": None, "secondary": None}),
                        "note": "224 B per fluid particle per CG iteration (SURVEY 8d); 106 k rows: launch/dependency latency, not bandwidth; "
                                "traffic / secondary are the A p walk's (cg_ap), one of the iteration's two launches"}}
    eng.close()
    return out


# ----------------------------------------------------------------------------------------------- one rank
def run_rank(args, rank, world, local_rank):
    # stdout carries ONE JSON line and nothing else: librccl prints a version banner through C stdio (flushed at exit,
    # i.e. after the line), so everything but the line is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    from sph_project_amd import _lib as L, product as P
    lib = L.load()
    method = args.method or ("dfsph" if args.config == "c3" else "wcsph")
    sharded = world > 1 and not args.replicas   # every solver shards (implicit viscosity, which no bench config uses, would not)
    scale_z = world if (sharded and args.scaling == "weak" and args.config in ("c2", "c3")) else 1
    cfg = (P.dam_break_scene(method=method) if args.config == "c1" else
           P.c4_scene(method) if args.config == "c4" else P.c2_scene(method, scale_z=scale_z))
    ndev = lib.sph_device_count()
    if ndev < 1:
        raise RuntimeError("no HIP device visible (the product path has no CPU fallback)")
    device = local_rank % ndev if world > 1 else -1   # several ranks per GPU only happen with SPH_COMM_TRANSPORT=shm
    slab_opt = comm_opt = None
    n_global = None
    force_slab = world == 1 and bool(os.environ.get("SPH_BENCH_FORCE_SLAB"))
    if world > 1 or force_slab:
        uid = exchange_unique_id(lib, rank) if world > 1 else None
        if force_slab:   # tuning aid: ONE rank in slab mode (no neighbour to talk to): the compute-side cost of sharding
            import ctypes
            buf = ctypes.create_string_buffer(128)
            assert lib.sph_comm_unique_id(buf) == 0
            uid = buf.raw
        if sharded or force_slab:
            import numpy as np
            from sph_project_amd import slab
            _, geo, batches = P.scene_particles(cfg)
            z = np.concatenate([b["pos"][:, 2] for b in batches])
            nz = int(geo.grid_num[2])
            cuts = slab.plan_slabs(np.bincount(slab.cell_layer(z, geo.dh, nz), minlength=nz), world)
            slab_opt = dict(rank=rank, nranks=world, unique_id=uid, cuts=cuts)
            n_global = int(sum((b["material"] == 1).sum() for b in batches))
            del batches, z
        else:
            comm_opt = dict(rank=rank, nranks=world, unique_id=uid)
    opts = dict(fast_math=0 if args.strict_math else 1, deterministic=0 if args.no_deterministic else 1,
                force_global=int(os.environ.get("SPH_DEBUG_MODE", int(args.force_global))), device=device)
    if method != "wcsph" and not args.measured_iterations:
        opts["fixed_iterations"] = 2
    if slab_opt:
        opts["slab"] = slab_opt      # a failure here is fatal on every rank: no silent fall-back to replicas
    if comm_opt:
        opts["comm"] = comm_opt
    container, solver = P.build_product(cfg, **opts)
    eng = container.engine
    if os.environ.get("SPH_BENCH_SELFTEST") and (slab_opt or comm_opt):
        eng.comm_selftest(1 << 16)
    solver.prepare()
    if os.environ.get("SPH_BENCH_FAIL_RANK") == str(rank):   # test hook: this rank dies while the others wait for its halo
        raise SystemExit(3)
    n_fluid = container.fluid_particle_num[None]
    names = [lib.sph_kernel_name(k).decode() for k in range(N_KERNEL_IDS)]
    multi = world > 1

    def fence():
        eng.synchronize()
        if multi:
            eng.comm_barrier()

    def allmax(x):
        return eng.comm_allreduce([x], "max")[0] if multi else x

    def allsum(x):
        return eng.comm_allreduce([x], "sum")[0] if multi else x

    run = eng.step if args.measured_iterations else eng.step_async   # convergence tests need the (batched) flag read-back
    steps_done = 0
    if args.presteps:
        run(args.presteps); steps_done += args.presteps
    run(args.warmup); steps_done += args.warmup
    fence()

    # pre-pass: which kernel dominates?  (all-kernel events perturb the stream slightly -> untimed)
    eng.profile_enable(-1, True)
    eng.profile_reset()
    run(5); steps_done += 5
    eng.synchronize()
    table = {names[k]: eng.profile_read(k) for k in range(N_KERNEL_IDS)}
    table = {k: v for k, v in table.items() if v[0] > 0}
    dom = max((k for k in table if k in ALG_BYTES), key=lambda k: table[k][1])
    # The two neighbour walks of a WCSPH step take the same time to within 1-3 %, so "the kernel with the most time" flipped between runs --
    # and `roofline.frac` with it (24 vs 96 algorithmic bytes per particle: 3.4 % vs 13.6 %).  Among the kernels within 3 % of the longest the
    # one that moves the most algorithmic bytes is named; the other one is `runner_up`, and `step_frac` does not depend on the choice.
    near = [k for k in table if k in ALG_BYTES and table[k][1] >= 0.97 * table[dom][1]]
    dom = max(near, key=lambda k: ALG_BYTES[k])
    if args.all_kernels and rank == 0:
        for k, (n, ms) in sorted(table.items(), key=lambda kv: -kv[1][1]):
            print(f"  {k:24s} launches {n:5d}  avg {1e3 * ms / n:9.1f} us", file=sys.stderr)
    eng.profile_enable(-1, False)
    # ... and every kernel that carries algorithmic bytes once more ALONE (one kernel id's events at a time): with all ids enabled the
    # event pairs of neighbouring launches serialise the stream and inflate a walk by 6-8 %.  `all_kernels` / `runner_up`
    # quote these solo figures; the all-event one is kept beside them.
    solo = {}
    if not os.environ.get("SPH_BENCH_NO_EVENTS"):
        for k in [k for k in table if k in ALG_BYTES]:
            eng.profile_enable(names.index(k), True); eng.profile_reset()
            run(3); steps_done += 3
            eng.synchronize()
            solo[k] = eng.profile_read(names.index(k))
            eng.profile_enable(-1, False)
    table_all = dict(table)
    table = {k: (solo[k] if k in solo and solo[k][0] > 0 else v) for k, v in table.items()}

    def timed(reps):
        """reps x (fence, K steps, fence); returns the per-repetition max-over-ranks seconds"""
        out = []
        for _ in range(reps):
            fence()
            t0 = time.perf_counter()
            run(args.steps)
            fence()
            out.append(allmax(time.perf_counter() - t0))
        return out

    eng.profile_enable(names.index(dom), not os.environ.get("SPH_BENCH_NO_EVENTS"))
    eng.profile_reset()
    timed_from = steps_done   # steps of the scene behind it when the timed region starts
    st_before = solver.stats()
    reps = timed(args.repeats); steps_done += args.repeats * args.steps
    elapsed = median(reps)
    launches, ms = eng.profile_read(names.index(dom))
    eng.profile_enable(-1, False)
    stats = solver.stats()
    # Which sort path did the timed region take?  Inside one sph_step_async(K) of an unsharded all-fluid WCSPH scene the force pass of
    # every step but the last is also the next step's init_grid (NextHash): it then moves the hash's 16 B per particle as well, and
    # its algorithmic bytes are 96 + 16 = 112 (the step's 204 B are unchanged: the hash kernel's 16 B moved, they did not vanish).
    prehashed = int(stats.get("prehashed_sorts", 0) - st_before.get("prehashed_sorts", 0))
    hashed = int(stats.get("hash_launches", 0) - st_before.get("hash_launches", 0))
    alg_bytes = dict(ALG_BYTES)
    if prehashed > 0 and prehashed + hashed > 0:
        alg_bytes["wcsph_forces"] = ALG_BYTES["wcsph_forces"] + ALG_BYTES["hash_count"] * prehashed / (prehashed + hashed)
    pairs, evals = stats["pair_interactions"], stats["pair_evaluations"]
    if sharded or force_slab:
        n_total = n_global  # every fluid particle is owned by exactly one rank
        pairs, evals = int(allsum(pairs)), int(allsum(evals))
        n_fluid = n_global // world  # per-launch share for the roofline leg
    else:
        n_total = n_fluid * world  # replicas: every rank advances its own copy
        pairs, evals = pairs * world, evals * world
    value = n_total * args.steps / elapsed
    # (a sharded step may run a pass as two launches -- boundary tiles, then interior tiles: per-launch figures are then per PASS)
    passes = args.repeats * args.steps if launches > 1.5 * args.repeats * args.steps else launches
    avg_s = (ms / max(passes, 1)) * 1e-3
    achieved = alg_bytes[dom] * n_fluid / avg_s / 1e9 if launches else None

    in_motion = None
    fixed_work = method == "wcsph" or not args.measured_iterations   # (measured-iteration loops: ms/step is the iteration count's)
    if args.motion_step and not args.presteps and args.config in ("c2", "c3", "c4") and fixed_work and steps_done < args.motion_step:
        run(args.motion_step - steps_done); steps_done = args.motion_step
        m_el = median(timed(args.repeats))
        st2 = solver.stats()
        p2 = int(allsum(st2["pair_interactions"])) if (sharded or force_slab) else st2["pair_interactions"] * world
        e2 = int(allsum(st2["pair_evaluations"])) if (sharded or force_slab) else st2["pair_evaluations"] * world
        in_motion = {"from_step": args.motion_step, "ms_per_step": 1e3 * m_el / args.steps,
                     "value": n_total * args.steps / m_el, "pair_interactions_per_step": p2,
                     "pair_interactions_per_s": p2 * args.steps / m_el, "pair_evaluations_per_s": e2 * args.steps / m_el,
                     "lds_fallback_blocks_last_step": int(st2["lds_fallback_blocks"])}
        if method == "wcsph":   # two neighbour walks per step (density, fused forces)
            in_motion["neighbours_per_particle"] = e2 / max(n_total, 1) / 2.0

    # PMC-derived figures of the dominant kernel: NOT measured in this run -- read from profiles/pmc_derived.json, which
    # tools/prof_summary.py --json writes from the committed rocprofv3 PMC passes of this same command (source named there)
    traffic = secondary = pmc_source = None
    tf = os.path.join(ROOT, "profiles", "pmc_derived.json")
    if os.path.exists(tf):
        try:
            d = json.load(open(tf))
            e = d.get(args.config, {}).get(dom)
            if e:
                traffic = e.get("hbm_bytes_per_launch")
                secondary = {k: e.get(k) for k in ("valu_issue_frac", "waves_parked_frac", "lds_active_frac", "lds_conflict_frac", "eff_clock_ghz")}
                pmc_source = d.get(args.config, {}).get("_source") or d.get("_source")
        except Exception:  # noqa: BLE001
            pass
    copy_gbs = None
    if rank == 0:
        try:
            copy_gbs = eng.measure_copy_rate(1 << 30, 10)   # measured now, on this GPU: the second denominator
        except Exception as ex:  # noqa: BLE001
            print(f"[bench] copy-rate measurement failed: {ex}", file=sys.stderr)

    step_bytes = 204 * n_fluid + 12 * int(container.grid_num.prod()) // (world if sharded else 1)   # SURVEY 8d, per rank
    scaling = ("strong" if args.config == "c4" else args.scaling) if sharded else "weak"
    transport = (eng.comm_transport() if hasattr(eng, "comm_transport") and (sharded or force_slab)
                 else os.environ.get("SPH_COMM_TRANSPORT", "rccl"))
    out = {
        "metric": "particle-updates/sec", "value": value, "unit": "particle-updates/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "repeat_ms_per_step": [1e3 * r / args.steps for r in reps],
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": {"c1": "C1 8,000-particle cube dam break", "c2": "C2 1,231,200-particle dam break" + (f" x{scale_z} in z" if scale_z > 1 else ""),
                         "c4": "C4 4,000,000-particle dam break",
                         "c3": "C3 1,231,200-particle dam break, " + ("DFSPH iterations as measured" if args.measured_iterations else "2+2 fixed DFSPH iterations")}[args.config],
            "method": method, "particles": int(n_total), "grid_cells": int(container.grid_num.prod()),
            "dt": cfg["Configuration"]["timeStepSize"], "math": "strict" if args.strict_math else "fast",
            "deterministic_sort": not args.no_deterministic,
            "parallelism": "single-gpu" if world == 1 else
                           (f"z-slab x{world}, {transport} halo exchange ({scaling} scaling)" if sharded else f"replicas x{world}"),
            "state": "steps %d..%d from the initial lattice" % (timed_from, timed_from + args.repeats * args.steps),
            "pair_interactions_per_step": int(pairs),
            "pair_interactions_per_s": pairs * args.steps / elapsed,      # every accepted pair once per REFERENCE pass
            "pair_evaluations_per_step": int(evals),
            "pair_evaluations_per_s": evals * args.steps / elapsed,       # as evaluated: once per neighbour walk of this library
            "neighbours_per_particle": evals / max(n_total, 1) / 2.0 if method == "wcsph" else None,   # density walk + force walk
            "lds_fallback_blocks_last_step": int(stats["lds_fallback_blocks"]),
            **({"solver_iterations_last_step": {"density": int(stats["iter_density"]), "divergence": int(stats["iter_divergence"])}} if method == "dfsph" else {}),
            "device": eng.device_info()["name"],
        },
        "in_motion": in_motion,
        "value_state": "from rest (steps %d.. of the initial lattice, %s neighbours per particle); `value_in_motion` is the same scene from step %s on"
                       % (timed_from, ("%.0f" % (evals / max(n_total, 1) / 2.0)) if method == "wcsph" else "n/a", args.motion_step),
        "value_in_motion": in_motion["value"] if in_motion else None,
        "roofline": {
            "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
            "traffic_source": (f"{pmc_source}: 2*FETCH_SIZE + WRITE_SIZE per launch from separate rocprofv3 --pmc passes; NOT measured in this run"
                               if traffic else None),
            "launches": int(launches), "avg_launch_us": 1e6 * avg_s,
            "alg_bytes_per_launch": alg_bytes[dom] * n_fluid,
            "alg_bytes_per_particle": alg_bytes[dom],
            "sort_path": {"steps_timed": args.repeats * args.steps, "hash_kernel_launches": hashed, "hashed_by_the_force_pass": prehashed,
                          "note": "NextHash: inside one sph_step_async(K) the force pass also files cell id / histogram / arrival rank of the next "
                                  "sort (16 B per particle on top of its 96); tests/test_hip_wcsph.py::test_c2_full_size_20_steps runs this "
                                  "path against the oracle and asserts these counters"},
            "step_achieved": step_bytes / (elapsed / args.steps) / 1e9, "step_alg_bytes": step_bytes,
            "step_frac": step_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,   # whole step: does not depend on which kernel is "dominant"
            "measured_copy_gbs": copy_gbs,
            "frac_of_measured_copy": (achieved / copy_gbs) if (achieved and copy_gbs) else None,
            "step_frac_of_measured_copy": (step_bytes / (elapsed / args.steps) / 1e9 / copy_gbs) if copy_gbs else None,
            "secondary": secondary,
            # the two neighbour walks of a WCSPH step are within 2-3 % of each other, so which one is "dominant" can flip between runs --
            # and with it `frac` (24 vs 96 algorithmic bytes per particle).  Both, each from an event pass of its own in this run (one kernel id
            # enabled at a time; `avg_us_with_all_events_on` is the 6-8 % higher figure of the pass with every id enabled):
            "all_kernels": {k: {"avg_us": 1e3 * v[1] / v[0], "avg_us_with_all_events_on": 1e3 * table_all[k][1] / table_all[k][0],
                                "alg_bytes_per_launch": alg_bytes[k] * n_fluid,
                                "frac": alg_bytes[k] * n_fluid / (v[1] / v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS}
                            for k, v in table.items() if k in ALG_BYTES},
            "runner_up": (lambda ru: None if ru is None else {"kernel": ru, "avg_launch_us": 1e3 * table[ru][1] / table[ru][0],
                                                               "frac": alg_bytes[ru] * n_fluid / (table[ru][1] / table[ru][0] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                               "note": "second kernel by time; when it is within a few per cent of `kernel`, which of the two "
                                                                       "is 'dominant' -- and with it `frac` (their algorithmic bytes differ 4x) -- can flip between runs"})(
                max((k for k in table if k in ALG_BYTES and k != dom), key=lambda k: table[k][1], default=None)),
            "note": "the neighbour passes are not HBM-bound: per the PMC passes they run at ~2.4 GHz with about half of the VALU issue "
                    "slots used and 40-50 % of the wave-cycles parked (4-5 resident waves per SIMD, about half of them runnable: occupancy, "
                    "not serialised loads -- halving a workgroup's memory round trips moved them 2 %, profiles/r03w_ab_round_trips.txt); "
                    "`secondary` (from profiles/, not this run) carries those figures; DESIGN.md 5",
        },
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline(cfg, args.cpu_steps)
        out["cpu_baseline"] = {
            "value": cb["value"], "unit": "particle-updates/s", "cores": cb["cores"], "kind": "port",
            "sample": f"{args.cpu_steps} steps of the same workload per thread count ({cb['secs']:.1f} s at the best one), "
                      f"oracle/sph_ref.c (this repo's C restatement of the reference algorithm, OpenMP), not Taichi",
            "pair_interactions_per_s": cb["pairs_per_s"], "cpu": cb["model"], "hardware_threads": cb["ncpu"],
            "threads_sweep": {str(k): v for k, v in cb["sweep"].items()},
            "speedup_over_1_thread": cb["speedup_over_1_thread"],
            "cgroup_cpu_quota_cores": cb["cpu_quota_cores"],
            "affinity_threads": cb["affinity_threads"],   # of the process, before OpenMP bound the calling thread to its own core
            "thread_binding": "OMP_PROC_BIND=%s OMP_PLACES=%s, arrays first-touched page-interleaved over the threads" % (
                os.environ.get("OMP_PROC_BIND"), os.environ.get("OMP_PLACES")),
        }
    elif rank == 0:
        out["cpu_baseline"] = None
    extras_ok = not args.no_extras and args.config == "c2" and method == "wcsph" and not args.presteps
    if multi and sharded and extras_ok:
        eng.comm_barrier()
        eng.close()
        if args.scaling == "weak":   # (the headline was the weak one: BASELINE.json's metric as written rides along)
            out["c2_strong_scaling"] = sharded_extra(args, rank, world, device, lib, P.c2_scene("wcsph"),
                                                     "C2 1,231,200-particle dam break, WCSPH", ".c2s")
        else:                        # the headline IS the metric as written; one C2 block per rank as an extra
            out["c2_weak_scaling"] = sharded_extra(args, rank, world, device, lib, P.c2_scene("wcsph", scale_z=world),
                                                   "C2 1,231,200-particle dam break x%d in z (one block per rank), WCSPH" % world, ".c2w",
                                                   scaling="weak")
        if not args.no_c4:
            out["c4_strong_scaling"] = sharded_extra(args, rank, world, device, lib, P.c4_scene("wcsph"),
                                                     "C4 4,000,000-particle dam break, WCSPH", ".c4")
    elif multi:
        eng.comm_barrier()
    elif extras_ok and rank == 0:
        eng.close()
        ex = {}
        for key, fn in (("c3", extra_c3), ("c5", extra_c5)):
            try:
                ex[key] = fn(args, names)
            except Exception as e:  # noqa: BLE001  (an extra never takes the headline down)
                ex[key] = {"error": f"{type(e).__name__}: {e}"}
        out["extras"] = ex
    if multi:
        if rank == 0 and not os.environ.get("SPH_BENCH_RDV"):
            try:
                os.unlink(rendezvous_path())
            except OSError:
                pass
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        return spawn_ranks(args.gpus)
    world = int(env_world) if env_world is not None else 1
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    run_rank(args, int(os.environ.get("RANK", "0")), world, int(os.environ.get("LOCAL_RANK", "0")))


if __name__ == "__main__":
    main()
