"""GPU: what round 1 left untested -- late entryTime insertion (base_container.py:218-221) for all three solvers incl. a
late static rigid body, the full-size BASELINE configs C3 / C4 and a scaled C5 against the oracle, the torch-free
multi-rank bench launcher, the RCCL transport (one-rank communicator: ncclCommInitRank + a self send/recv group + an
all-reduce execute on hardware), dump(), and the run_simulation.py driver on a scene file."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from sph_project_amd import _lib as L
from sph_project_amd import product as P
from tests import helpers as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cube(lo, n, spacing=0.02):
    ax = [lo[k] + spacing * np.arange(n[k]) for k in range(3)]
    return np.ascontiguousarray(np.stack(np.meshgrid(*ax, indexing="ij"), -1).reshape(-1, 3), dtype=np.float32)


def _state(container):
    e = container.engine
    ids = e.download(L.F_PARTICLE_ID)
    return ids, {k: H.by_id(ids, e.download(f)) for k, f in (("x", L.F_POSITION), ("v", L.F_VELOCITY), ("rho", L.F_DENSITY),
                                                            ("V", L.F_REST_VOLUME), ("m", L.F_MASS), ("mat", L.F_MATERIAL))}


def _oracle_state(ref):
    ids = H.oracle_ids(ref)
    return ids, {k: H.by_id(ids, ref.field(f).copy()) for k, f in (("x", "particle_positions"), ("v", "particle_velocities"),
                                                                   ("rho", "particle_densities"), ("V", "particle_rest_volumes"),
                                                                   ("m", "particle_masses"), ("mat", "particle_materials"))}


@pytest.mark.parametrize("method", ["wcsph", "dfsph", "pcisph"])
def test_late_fluid_block_and_late_static_rigid_body(gpu, method):
    """A fluid block (entryTime in step 3) and a static rigid body (entryTime in step 5) enter a running simulation.
    The reference inserts both in the middle of _step() (WCSPH.py:41 / DFSPH.py:307 / PCISPH.py:181); its end-of-step
    compute_rigid_particle_volume (base_solver.py:696) then runs on a grid that does not contain the new body (WCSPH /
    PCISPH: V = 1 / W(0) for one step; DFSPH: V0 during the post-insertion passes).  Oracle = literal restatement."""
    dt = 6e-4 if method == "dfsph" else 4e-4
    cfg = H.dam_break_scene(method=method, end=(0.16, 0.12, 0.16), dt=dt, velocity=(0.0, -0.5, 0.0), viscosity_b=0.3)
    cfg["FluidBlocks"].append({"objectId": 1, "start": [0.0, 0.0, 0.0], "end": [0.09, 0.07, 0.09], "translation": [0.12, 0.22, 0.13],
                               "scale": [1, 1, 1], "velocity": [0.0, -1.0, 0.0], "density": 1000.0, "color": [1, 2, 3],
                               "entryTime": 2.5 * dt})
    pts = _cube((0.12, 0.04, 0.12), (6, 2, 6))   # a slab under the first block, within its support radius
    cfg["RigidBodies"] = [{"objectId": 2, "geometryFile": "unused.obj", "voxelizedPoints": pts.tolist(), "isDynamic": False,
                           "entryTime": 4.5 * dt, "density": 2000.0, "color": [9, 9, 9], "velocity": [0.0, 0.0, 0.0],
                           "translation": [0, 0, 0], "scale": [1, 1, 1], "rotationAngle": 0.0, "rotationAxis": [0, 1, 0]}]
    container, solver = H.build_product(cfg, fixed_iterations=3 if method != "wcsph" else 0)
    solver.prepare()
    # the oracle wrapper knows fluid blocks only: budget for the body and queue it by hand (ids continue in entry order)
    n_r = pts.shape[0]
    cfg_o, geo, batches = H.scene_particles({k: v for k, v in cfg.items() if k != "RigidBodies"})
    from sph_project_amd import scene
    pd = scene.params_dict(geo, scene.derive_solver_constants(cfg_o), method, sum(b["pos"].shape[0] for b in batches) + n_r,
                           fixed_iterations=3 if method != "wcsph" else 0)
    ref = H.oracle_ref.RefSim(pd)
    ref._next_id, ref._pending, ref._time, ref._dt = 0, [], 0.0, float(np.float32(dt))
    for b in batches:
        (ref._pending.append(b) if b["entry_time"] > 0 else H._oracle_insert(ref, b))
    ref._pending.append(dict(object_id=2, pos=pts, vel=np.zeros((n_r, 3), np.float32), density=np.full(n_r, 2000.0, np.float32),
                             material=np.full(n_r, 2, np.int32), is_dynamic=np.zeros(n_r, np.int32), entry_time=4.5 * dt))
    ref.prepare()
    n0 = container.particle_num[None]
    seen = set()
    first_rigid_V = None
    for step in range(1, 10):
        solver.step()
        H.oracle_step(ref, 1)
        assert container.particle_num[None] == ref.particle_num, (step, container.particle_num[None], ref.particle_num)
        seen.add(container.particle_num[None])
        ids, mine = _state(container)
        ido, theirs = _oracle_state(ref)
        assert np.array_equal(np.sort(ids), np.arange(len(ids))) and np.array_equal(np.sort(ido), np.arange(len(ido)))
        assert np.array_equal(mine["mat"], theirs["mat"])
        d = H.drift(mine["x"], theirs["x"], container.dh).max()
        assert d <= 1e-5, (method, step, d)
        np.testing.assert_allclose(mine["V"], theirs["V"], rtol=2e-6, err_msg=f"{method} rest volumes after step {step}")
        np.testing.assert_allclose(mine["m"], theirs["m"], rtol=2e-6)
        fl = mine["mat"] == 1
        np.testing.assert_allclose(mine["rho"][fl], theirs["rho"][fl], rtol=2e-5, err_msg=f"{method} density after step {step}")
        np.testing.assert_allclose(mine["v"], theirs["v"], rtol=0, atol=2e-5 * max(1.0, float(np.abs(theirs["v"]).max())))
        assert solver.stats()["pair_interactions"] == ref.last_pairs, (method, step)
        if first_rigid_V is None and (mine["mat"] == 2).any():
            first_rigid_V = mine["V"][mine["mat"] == 2]
    assert len(seen) == 3 and min(seen) == n0, seen   # 2 insertions happened, at different steps
    # the late body's volumes went through the stale-grid value (one value for all its particles) and ended at the
    # proper, position-dependent ones
    assert first_rigid_V is not None and (len(np.unique(first_rigid_V)) == 1) == (method != "dfsph"), np.unique(first_rigid_V)
    rigid = mine["mat"] == 2
    assert rigid.sum() == n_r and len(np.unique(mine["V"][rigid])) > 1


def _two_block_scene():
    cfg = H.dam_break_scene(end=(0.12, 0.12, 0.12))
    cfg["FluidBlocks"].append({"objectId": 3, "start": [0.0, 0.0, 0.0], "end": [0.07, 0.07, 0.07], "translation": [0.3, 0.1, 0.3],
                               "scale": [1, 1, 1], "velocity": [0.1, 0.0, 0.0], "density": 1000.0, "color": [1, 2, 3], "entryTime": -1.0})
    return cfg


def _oracle_dump(ref, oid):
    """BaseContainer.dump (base_container.py:599-609) on the oracle's state: positions / velocities of the particles whose object
    id is `oid`, in the current sorted order (the oracle's stable counting sort = serial execution of :510-515)."""
    sel = ref.field("particle_object_ids") == oid
    return ref.field("particle_positions")[sel].copy(), ref.field("particle_velocities")[sel].copy(), H.oracle_ids(ref)[sel]


def test_dump_matches_the_oracle(gpu):
    """BaseContainer.dump (base_container.py:599): positions / velocities of one object id in the current sorted order -- against the
    ORACLE's dump of the same scene after the same steps (same particles in the same order, values to the parity limit), not only
    against the product's own download."""
    cfg = _two_block_scene()
    container, solver = H.build_product(cfg)
    solver.prepare()
    ref = H.build_oracle(cfg)
    ref.prepare()
    for _ in range(3):
        solver.step()
    H.oracle_step(ref, 3)
    e = container.engine
    obj, pid = e.download(L.F_OBJECT_ID), e.download(L.F_PARTICLE_ID)
    for oid, n_expect in ((0, 216), (3, 64)):
        d = container.dump(obj_id=oid)
        assert d["position"].shape == (n_expect, 3) and d["velocity"].shape == (n_expect, 3)
        assert d["position"].dtype == np.float32 and d["velocity"].dtype == np.float32
        xr, vr, idr = _oracle_dump(ref, oid)
        np.testing.assert_array_equal(pid[obj == oid], idr)                      # same particles, same (sorted) order
        assert H.drift(d["position"], xr, container.dh).max() <= 1e-5
        assert np.abs(d["velocity"] - vr).max() <= 5e-5 * max(float(np.abs(vr).max()), 1e-30)
        np.testing.assert_array_equal(d["position"], e.download(L.F_POSITION)[obj == oid])
        np.testing.assert_array_equal(d["velocity"], e.download(L.F_VELOCITY)[obj == oid])
    assert container.object_id_fluid_body == {0, 3}


def test_run_simulation_driver_writes_the_reference_frames(gpu, tmp_path):
    """The drop-in driver on a JSON scene in the reference's format: frame directories and one ASCII PLY per fluid object
    ({scene}_output/{cnt:06}/particle_object_{id}.ply, run_simulation.py:137-144) with the loop arithmetic of :28-39 -- and the
    NUMBERS in every file: a frame written at count c holds dump(obj_id) after c + 1 steps (the reference steps, then exports, then
    counts: :126-151); every PLY of every frame is parsed and compared vertex for vertex, in file order, with the oracle's dump at
    that step.  Header and number format are Taichi's PLYWriter.export_ascii as restated in tests/test_host_cpu.py::test_ply_writer_layout."""
    from sph_project_amd.run_simulation import read_ply_ascii
    cfg = _two_block_scene()
    cfg["Configuration"].update(exportPly=True, fps=500, totalTime=0.0064)   # output_interval = int((1/500)/4e-4) = 5, 16 rounds
    scene_file = tmp_path / "tiny_dam.json"
    scene_file.write_text(json.dumps(cfg))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "sph_project_amd", "run_simulation.py"), "--scene_file", str(scene_file)],
                       cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Simulation Finished: 16 steps" in r.stdout, r.stdout
    assert "writing 4 frame(s)" in r.stdout, r.stdout
    out = tmp_path / "tiny_dam_output"
    frames = sorted(os.listdir(out))
    assert frames == ["000000", "000005", "000010", "000015"], frames
    ref = H.build_oracle(cfg)
    ref.prepare()
    done = 0
    for fr in frames:
        H.oracle_step(ref, int(fr) + 1 - done)
        done = int(fr) + 1
        assert sorted(os.listdir(out / fr)) == ["particle_object_0.ply", "particle_object_3.ply"]
        for oid, n_expect in ((0, 216), (3, 64)):
            path = out / fr / f"particle_object_{oid}.ply"
            head = path.read_text().split("end_header\n")[0]
            assert head == (f"ply\nformat ascii 1.0\ncomment created by PLYWriter\nelement vertex {n_expect}\n"
                            "property float x\nproperty float y\nproperty float z\n"), head
            x = read_ply_ascii(str(path))
            xr, _, _ = _oracle_dump(ref, oid)
            assert x.shape == xr.shape == (n_expect, 3)
            d = H.drift(x, xr, 0.04)
            assert d.max() <= 1e-5, (fr, oid, d.max())
    # a scene that exports nothing counts no frames
    cfg["Configuration"].update(exportPly=False)
    scene_file.write_text(json.dumps(cfg))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "sph_project_amd", "run_simulation.py"), "--scene_file", str(scene_file),
                        "--output_dir", str(tmp_path / "none")], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "writing 0 frame(s)" in r.stdout, r.stdout + r.stderr


# --------------------------------------------------------------------------------------------- BASELINE configs
def _full_size_vs_oracle(cfg, steps, fixed, tol, pair_rtol=0.0, advance=False):
    container, solver = H.build_product(cfg, fast_math=1, **({"fixed_iterations": fixed} if fixed else {}))
    solver.prepare()
    ref = H.build_oracle(cfg, fixed_iterations=fixed)
    ref.prepare()
    if advance:   # ONE sph_step_async(steps), the call bench.py times (WCSPH: the force pass is the next step's init_grid, NextHash)
        p0 = solver.stats()["prehashed_sorts"]
        solver.advance(steps)
        assert solver.stats()["prehashed_sorts"] - p0 == steps - 1, solver.stats()
    else:
        for _ in range(steps):
            solver.step()
    ref.step(steps)
    e = container.engine
    ids = e.download(L.F_PARTICLE_ID)
    n = len(ids)
    assert np.array_equal(np.sort(ids), np.arange(n))
    x = H.by_id(ids, e.download(L.F_POSITION))
    xr = H.by_id(H.oracle_ids(ref), ref.field("particle_positions").copy())
    assert np.isfinite(x).all()
    d = H.drift(x, xr, container.dh)
    st = solver.stats()
    print("full size n=%d: drift max %.3e p99 %.3e; pairs/step %d" % (n, d.max(), np.percentile(d, 99), st["pair_interactions"]))
    assert d.max() <= tol, d.max()
    assert abs(st["pair_interactions"] - ref.last_pairs) <= pair_rtol * ref.last_pairs, (st["pair_interactions"], ref.last_pairs)
    return container, solver, ref


def test_c3_full_size_dfsph(gpu):
    """BASELINE configs[2]: 1,231,200 particles, DFSPH, 2 + 2 fixed solver iterations (bench.py --config c3), 5 steps."""
    # fast-math DFSPH positions differ from the oracle's in the last bit, which flips a few of the lattice's exactly-2d-apart
    # pairs (|x_ij| = h up to rounding; W and grad W vanish there): pair counts agree to ~1e-6, not exactly as for WCSPH
    container, solver, ref = _full_size_vs_oracle(P.c2_scene("dfsph"), 5, 2, 1e-4, pair_rtol=2e-6)
    ids = container.engine.download(L.F_PARTICLE_ID)
    rho = H.by_id(ids, container.engine.download(L.F_DENSITY))
    rho_r = H.by_id(H.oracle_ids(ref), ref.field("particle_densities").copy())
    np.testing.assert_allclose(rho, rho_r, rtol=5e-5)


def test_c3_full_size_dfsph_in_motion(gpu, from_step=1000):
    """BASELINE configs[2] in the state a DFSPH user spends a run in: the 1.23 M column collapsed (step 1000 of bench.py --config c3,
    2 + 2 fixed iterations; from rest the solvers have nothing to correct and the walks see the lattice's 29 neighbours).  The oracle is
    seeded with the product's positions / velocities at that step, in the product's order (H.oracle_from_state; its prepare() recomputes
    the density and alpha the product's last step_end left, DFSPH.py:321), both advance 3 steps with 2 + 2 iterations."""
    cfg = P.c2_scene("dfsph")
    container, solver = H.build_product(cfg, fast_math=1, fixed_iterations=2)
    solver.prepare()
    solver.advance(from_step)
    e = container.engine
    ids0, x0, v0 = (e.download(f) for f in (L.F_PARTICLE_ID, L.F_POSITION, L.F_VELOCITY))
    n = len(ids0)
    assert np.array_equal(np.sort(ids0), np.arange(n)) and np.isfinite(x0).all() and np.isfinite(v0).all()
    ref = H.oracle_from_state(cfg, x0, v0, ids0, fixed_iterations=2)
    ref.prepare()
    solver.advance(3)
    ref.step(3)
    ids, oid = e.download(L.F_PARTICLE_ID), H.oracle_ids(ref)
    assert np.array_equal(np.sort(ids), np.arange(n))
    x, xr = H.by_id(ids, e.download(L.F_POSITION)), H.by_id(oid, ref.field("particle_positions").copy())
    d = H.drift(x, xr, container.dh)
    rho, rho_r = H.by_id(ids, e.download(L.F_DENSITY)), H.by_id(oid, ref.field("particle_densities").copy())
    st = solver.stats()
    print("C3 in motion (steps %d..%d): drift max %.3e p99 %.3e, max relative density difference %.3e, pairs/step %d (oracle %+d), "
          "slots in another order %d" % (from_step, from_step + 3, d.max(), np.percentile(d, 99), np.abs(rho / rho_r - 1).max(),
                                         st["pair_interactions"], ref.last_pairs - st["pair_interactions"], int((ids != oid).sum())))
    assert d.max() <= 1e-5
    assert abs(st["pair_interactions"] - ref.last_pairs) <= 2e-6 * ref.last_pairs, (st["pair_interactions"], ref.last_pairs)
    np.testing.assert_allclose(rho, rho_r, rtol=5e-5)
    ref.close()


def test_c3_full_size_dfsph_in_motion_own_stop_tests(gpu, from_step=1000):
    """The regime `extras.c3.measured.in_motion` of the bench line quotes (what a DFSPH user gets): the collapsed 1.23 M column with the
    solvers' OWN stop tests (DFSPH.py:139-159, :225-243) -- ~26 density iterations per step, each ending in a mean over 1.23 M particles
    compared with a threshold (a22: a sum whose order can flip the stop iteration).  The product runs there by itself with its stop
    tests, the oracle is seeded with that state; two more steps each: the iteration counts of every step agree within one (measured: equal,
    26 and 27), drift <= 1e-4."""
    cfg = P.c2_scene("dfsph")
    container, solver = H.build_product(cfg, fast_math=1)
    solver.prepare()
    solver.advance(from_step)
    e = container.engine
    ids0, x0, v0 = (e.download(f) for f in (L.F_PARTICLE_ID, L.F_POSITION, L.F_VELOCITY))
    n = len(ids0)
    assert np.array_equal(np.sort(ids0), np.arange(n)) and np.isfinite(x0).all() and np.isfinite(v0).all()
    ref = H.oracle_from_state(cfg, x0, v0, ids0)
    ref.prepare()
    for step in (1, 2):
        solver.step()
        ref.step(1)
        st = solver.stats()
        it = (int(st["iter_divergence"]), int(st["iter_density"]))
        it_ref = (int(ref.scalar("last_iter_div")), int(ref.scalar("last_iter_den")))
        print("C3 own stop tests, step %d: iterations (divergence, density) hip %s oracle %s; residuals hip %.3e %.3e" % (
            from_step + step, it, it_ref, st["err_divergence"], st["err_density"]))
        assert abs(it[0] - it_ref[0]) <= 1 and abs(it[1] - it_ref[1]) <= 1, (it, it_ref)
        assert it_ref[1] >= 10, it_ref   # (this IS the many-iterations regime)
    ids, oid = e.download(L.F_PARTICLE_ID), H.oracle_ids(ref)
    x, xr = H.by_id(ids, e.download(L.F_POSITION)), H.by_id(oid, ref.field("particle_positions").copy())
    d = H.drift(x, xr, container.dh)
    print("C3 own stop tests in motion: drift max %.3e p99 %.3e" % (d.max(), np.percentile(d, 99)))
    # 2 x 27 Jacobi iterations of the density solver on a compressed column amplify the fast build's last-bit differences: measured
    # 1.3e-5 max (p99 5e-7) after two steps, against 2.3e-7 with 2 + 2 fixed iterations (the test above); the bound is the north star's
    assert d.max() <= 1e-4
    ref.close()


def test_c4_full_size_wcsph_one_gpu(gpu):
    """BASELINE configs[3]'s scene (4,000,000 particles, WCSPH) on ONE GPU, 5 steps in one advance(5) -- the timed path, with the
    force pass hashing for the next sort (its 8-GPU sharding: tests/test_hip_slab.py, 8 ranks on one GPU; real devices: the driver's)."""
    _full_size_vs_oracle(P.c4_scene(), 5, 0, 1e-4, advance=True)


def _large_block(side):
    d = 0.02
    ext = side * d + 0.4
    cfg = P.dam_break_scene(domain_end=(ext, ext, ext), start=(0.0, 0.0, 0.0), end=(side * d - 0.01, side * d - 0.01, side * d - 0.01),
                            translation=(0.2, 0.2, 0.2), velocity=(0.3, -0.5, 0.2))
    container, solver = H.build_product(cfg, fast_math=1)
    solver.prepare()
    e = container.engine
    n = e.particle_num
    assert n == side ** 3
    steps = 3
    v0 = e.download(L.F_VELOCITY).astype(np.float64).sum(0)
    t0 = time.perf_counter()
    e.step_async(steps); e.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    st = solver.stats()
    x, v = e.download(L.F_POSITION), e.download(L.F_VELOCITY)
    ids = e.download(L.F_PARTICLE_ID)
    assert np.isfinite(x).all() and np.isfinite(v).all()
    pad = np.float32(container.padding)
    assert (x >= pad).all() and (x <= (container.domain_size - container.padding).astype(np.float32)).all()
    seen = np.zeros(n, bool); seen[ids] = True
    assert seen.all(), "ids are a permutation"
    per_particle = st["pair_interactions"] / 4.0 / n      # 4 reference passes per WCSPH step
    # 26 lattice neighbours within 2 d, plus some of the 6 at exactly 2 d = h (decided by f32 rounding of the positions)
    assert 25.0 < per_particle < 32.5, per_particle
    dt, g = 4e-4, np.array([0.0, -9.81, 0.0])
    dv = v.astype(np.float64).sum(0) - v0
    expect = n * g * steps * dt
    print("%d particles: %.2f ms/step (first steps), %.1f neighbours/particle, momentum change / (M g t) = %s" % (
        n, ms, per_particle, dv / expect[1]))
    assert abs(dv[1] / expect[1] - 1.0) < 2e-5 and abs(dv[0] / expect[1]) < 2e-5 and abs(dv[2] / expect[1]) < 2e-5


@pytest.mark.skipif(bool(os.environ.get("SPH_SKIP_LARGE")), reason="SPH_SKIP_LARGE set")
def test_64_million_particles_on_one_gpu(gpu):
    """Maximum sizes: a 400^3 = 64,000,000-particle WCSPH block (52 x C2; ~16 GB of the 288 GB) -- 32-bit indices, byte offsets,
    grid dimensions and the counting sort at a size the oracle cannot reach in test time.  Checked through size-independent
    properties: the persistent ids stay a permutation, everything stays finite and inside the clamped domain, interior
    particles keep the lattice's neighbour count, and the pair forces are antisymmetric -- pressure, viscosity and surface
    tension cancel pairwise, so the total momentum changes by gravity alone (M g t)."""
    _large_block(400)


@pytest.mark.skipif(bool(os.environ.get("SPH_SKIP_LARGE")), reason="SPH_SKIP_LARGE set")
def test_250_million_particles_on_one_gpu(gpu):
    """630^3 = 250,047,000 particles (~65 GB of the 288 GB): just under the 2^28 - 1 a handle takes (32-bit byte offsets into float4
    arrays).  54 ms/step = the per-particle throughput of C2 at 203 x its size."""
    _large_block(630)


def test_particle_capacity_limit_is_an_error(gpu):
    from sph_project_amd import scene
    c = H.SimConfig(config=P.dam_break_scene())
    pd = scene.params_dict(scene.derive_geometry(c), scene.derive_solver_constants(c), "wcsph", 0x10000000)
    p = L.SphParams()
    for k in ("particle_radius", "support_radius", "V0", "padding", "g_upper", "viscosity", "viscosity_b", "density_0",
              "surface_tension", "dt", "particle_max_num", "viscosity_implicit"):
        setattr(p, k, pd[k])
    p.domain_size[:] = pd["domain_size"]; p.grid_num[:] = pd["grid_num"]; p.gravity[:] = pd["gravity"]
    p.method = L.METHOD["wcsph"]; p.device = -1
    with pytest.raises(L.SphError, match="268435455"):
        L.Engine(p)


def test_c5_scaled_buckling_scene(gpu):
    """BASELINE configs[4] (tools/bench_c5.py's scene: DFSPH + implicit viscosity mu = mu_b = 1800, emitter above
    gravitationUpper, sampled domain box) shrunk to a 10 x 60 x 3 sheet in a 1.2 x 2.4 x 1.2 box so that the oracle
    finishes in seconds.  Both sides run the reference's own stop tests (CG: |r| < 1e-6, base_solver.py:449): with mu = 1800
    the system is stiff, and a CG iterate cut off after a fixed handful of iterations amplifies the rounding of the dot
    products (4 fixed iterations: drift 5e-5 after 6 steps), a converged one does not."""
    cfg = P.c5_scene(domain_end=(1.2, 2.4, 1.2), start=(0.5, 0.4, 0.56), end=(0.7, 1.6, 0.62), g_upper=1.0)
    fixed = 0
    container, solver = H.build_product(cfg, fast_math=0, fixed_iterations=fixed)
    solver.prepare()
    ref = H.build_oracle(cfg, fixed_iterations=fixed)
    ref.prepare()
    n_fluid0 = container.fluid_particle_num[None]
    for step in range(1, 7):
        solver.step()
        ref.step(1)
        print("C5 scaled step %d: pairs hip %d oracle %d, cg iterations %d" % (step, solver.stats()["pair_interactions"], ref.last_pairs, solver.stats()["iter_cg"]))
        # (no pair-count assertion here: one CG iteration more or less is one neighbour pass more or less; and the CG dot
        # products are reduced in another order than the oracle's serial sums, so last-bit position differences flip a few
        # of the lattice's exactly-one-support-radius pairs, which carry W = grad W = 0)
    e = container.engine
    ids = e.download(L.F_PARTICLE_ID)
    mat = H.by_id(ids, e.download(L.F_MATERIAL))
    mat_r = H.by_id(H.oracle_ids(ref), ref.field("particle_materials").copy())
    assert np.array_equal(mat, mat_r)
    assert (mat == 1).sum() > 0 and (mat == 2).sum() > 30000   # emitter feeds fluid into a box of boundary particles
    x = H.by_id(ids, e.download(L.F_POSITION))
    xr = H.by_id(H.oracle_ids(ref), ref.field("particle_positions").copy())
    v = H.by_id(ids, e.download(L.F_VELOCITY))
    vr = H.by_id(H.oracle_ids(ref), ref.field("particle_velocities").copy())
    d = H.drift(x, xr, container.dh).max()
    print("C5 scaled: n=%d fluid=%d (at start %d) drift %.3e, |v|max %.3f" % (len(ids), (mat == 1).sum(), n_fluid0, d, np.abs(vr).max()))
    assert d <= 1e-5
    dv = float(np.abs(v.astype(np.float64) - vr).max())
    print("C5 scaled: max |v - v_oracle| = %.3e of |v|max %.3f" % (dv, np.abs(vr).max()))
    # both CG solves stop at |r| < 1e-6 (not at the same iterate): velocities agree to the solve's own accuracy
    assert dv <= 5e-3 * float(np.abs(vr).max())
    it_ref = int(ref.scalar("last_iter_cg"))
    print("C5 scaled: CG iterations hip %d oracle %d" % (solver.stats()["iter_cg"], it_ref))
    assert abs(solver.stats()["iter_cg"] - it_ref) <= 2 and it_ref >= 5


def test_measured_copy_rate_is_plausible(gpu):
    """sph_measure_copy_rate (the second denominator of the bench's roofline): between 1 and 8 TB/s on an MI355X, and repeatable."""
    container, solver = H.build_product(P.dam_break_scene(end=(0.1, 0.1, 0.1)))
    a = container.engine.measure_copy_rate(1 << 28, 8)
    b = container.engine.measure_copy_rate(1 << 28, 8)
    print("device copy rate: %.0f / %.0f GB/s" % (a, b))
    assert 1000.0 < a < 8000.0 and 1000.0 < b < 8000.0 and abs(a - b) < 0.35 * max(a, b)
    with pytest.raises(L.SphError):
        container.engine.measure_copy_rate(16, 1)


def test_implicit_viscosity_before_the_emitter_releases_anything(gpu):
    """The reference's final_scene4.json starts with every fluid particle above gravitationUpper: for its first 14 steps there is
    NO active fluid particle, the CG system is empty and the reference's loop leaves after one pass (|r| = 0, base_solver.py:445-461).
    Round 2 kept the loop's books in a workgroup that an empty fluid-workgroup list never ran: 1000 empty iterations per step
    (44 ms instead of 1 ms; found in round 3 by tools/scene0_iterations.py).  Same count as the oracle now, and no fluid moves."""
    cfg = P.c5_scene(domain_end=(1.2, 2.4, 1.2), start=(0.5, 1.3, 0.56), end=(0.7, 1.9, 0.62), g_upper=1.0, velocity=(0.0, -2.2, 0.0))
    container, solver = H.build_product(cfg, fast_math=1)
    solver.prepare()
    ref = H.build_oracle(cfg)
    ref.prepare()
    e = container.engine
    assert (e.download(L.F_MATERIAL) == 1).sum() == 0   # prepare_emitter froze the whole sheet
    for step in range(1, 4):
        t0 = time.perf_counter()
        solver.step()
        ms = 1e3 * (time.perf_counter() - t0)
        ref.step(1)
        it, it_ref = solver.stats()["iter_cg"], int(ref.scalar("last_iter_cg"))
        print("empty CG system, step %d: iterations hip %d oracle %d, %.2f ms" % (step, it, it_ref, ms))
        assert it_ref <= 1 and abs(it - it_ref) <= 1, (it, it_ref)
    ids = e.download(L.F_PARTICLE_ID)
    x = H.by_id(ids, e.download(L.F_POSITION))
    xr = H.by_id(H.oracle_ids(ref), ref.field("particle_positions").copy())
    assert H.drift(x, xr, container.dh).max() <= 1e-6


def test_c5_full_size(gpu):
    """BASELINE configs[4] at its full size: the reference's buckling scene (data/scenes/final_scene3.json without its mesh
    body): DFSPH + implicit viscosity, 2,171,495 particles of which 106,400 fluid, G = 10,000,000 cells, the solvers' own stop
    tests.  3 steps against the oracle (drift, CG iteration count of every step within +-2 of the oracle's and of the counts
    kept from the MI355X run of round 3: the solve starts at 16 iterations from rest and settles near 41-42 per step, the number
    bench.py reports for steps 8..28), plus size-independent properties."""
    cfg = P.c5_scene()
    container, solver = H.build_product(cfg, fast_math=1)
    solver.prepare()
    ref = H.build_oracle(cfg)
    ref.prepare()
    e = container.engine
    n, nf0 = e.particle_num, container.fluid_particle_num[None]
    assert n == 2171495 and nf0 == 106400 and int(container.grid_num.prod()) == 10_000_000
    x0 = H.by_id(e.download(L.F_PARTICLE_ID), e.download(L.F_POSITION))
    KEPT_CG = {1: 16}   # gpurun_out r03a: first step from rest (later steps: printed, checked against the oracle only)
    for step in range(1, 4):
        solver.step()
        ref.step(1)
        st = solver.stats()
        it_ref = int(ref.scalar("last_iter_cg"))
        print("C5 full size step %d: cg iterations hip %d oracle %d; dfsph iterations density %d divergence %d" % (
            step, st["iter_cg"], it_ref, st["iter_density"], st["iter_divergence"]))
        assert abs(st["iter_cg"] - it_ref) <= 2, (step, st["iter_cg"], it_ref)
        assert abs(st["iter_cg"] - KEPT_CG.get(step, st["iter_cg"])) <= 2 and 5 <= st["iter_cg"] <= 80, (step, st["iter_cg"])
    ids = e.download(L.F_PARTICLE_ID)
    assert np.array_equal(np.sort(ids), np.arange(n))
    mat = H.by_id(ids, e.download(L.F_MATERIAL))
    mat_r = H.by_id(H.oracle_ids(ref), ref.field("particle_materials").copy())
    assert np.array_equal(mat, mat_r)
    x = H.by_id(ids, e.download(L.F_POSITION))
    xr = H.by_id(H.oracle_ids(ref), ref.field("particle_positions").copy())
    assert np.isfinite(x).all()
    obj = H.by_id(ids, e.download(L.F_OBJECT_ID))
    box = obj < 0                                   # the sampled domain box: static boundary particles never move
    assert box.sum() == n - nf0 and np.array_equal(x[box], x0[box])
    d = H.drift(x, xr, container.dh)
    print("C5 full size: n=%d fluid now %d drift max %.3e p99 %.3e" % (n, (mat == 1).sum(), d.max(), np.percentile(d, 99)))
    assert d.max() <= 1e-4, d.max()


# --------------------------------------------------------------------------------------------- multi-rank launcher / RCCL
def _bench(args, env_extra, timeout=600):
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), "stdout must be the one JSON line and nothing else:\n" + r.stdout
    return json.loads(lines[0])


def test_advance_equals_repeated_step(gpu):
    """solver.advance(n) = n x solver.step(): one sph_step(h, n) where the host has nothing to do inside the steps, the
    step-by-step loop while an object is still waiting for its entryTime."""
    for cfg in (P.dam_break_scene(method="dfsph", dt=6e-4), None):
        if cfg is None:   # a second block enters during the 4th step
            cfg = P.dam_break_scene(end=(0.12, 0.1, 0.12), velocity=(0.0, -0.5, 0.0))
            cfg["FluidBlocks"].append({"objectId": 1, "start": [0.0, 0.0, 0.0], "end": [0.07, 0.07, 0.09], "translation": [0.12, 0.2, 0.11],
                                       "scale": [1, 1, 1], "velocity": [0.0, -1.0, 0.0], "density": 1000.0, "color": [1, 2, 3],
                                       "entryTime": 2.5 * 4e-4})
        a_c, a_s = H.build_product(cfg); a_s.prepare()
        b_c, b_s = H.build_product(cfg); b_s.prepare()
        a_s.advance(4); a_s.advance(5)
        for _ in range(9):
            b_s.step()
        assert a_c.total_time == b_c.total_time and a_c.engine.particle_num == b_c.engine.particle_num
        for f in (L.F_PARTICLE_ID, L.F_POSITION, L.F_VELOCITY, L.F_DENSITY):
            np.testing.assert_array_equal(a_c.engine.download(f), b_c.engine.download(f))
        del a_c, a_s, b_c, b_s


def test_handles_release_their_memory(gpu):
    """sph_destroy gives back what sph_create / sph_comm_* took: 12 create-run-destroy cycles (every second one in slab mode
    with a communicator) do not eat device memory."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")

    def free_bytes():
        f, t = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
        return f.value

    cfg = P.dam_break_scene(end=(1.2, 1.2, 1.2), domain_end=(1.6, 1.6, 1.6))   # 216,000 particles, ~130 MB per handle
    lib = L.load()

    def cycle(k):
        opts = {}
        if k % 2:
            buf = ctypes.create_string_buffer(128)
            assert lib.sph_comm_unique_id(buf) == 0
            opts["slab"] = dict(rank=0, nranks=1, unique_id=buf.raw, cuts=[0, 40])
        c, s = H.build_product(cfg, **opts)
        s.prepare(); s.step(); s.step()
        c.engine.close()

    os.environ.setdefault("SPH_COMM_TRANSPORT", "shm+ipc")   # (one rank: inbox, mirror and tickets are allocated and freed too)
    cycle(0); cycle(1)                 # first use: runtime pools, code objects
    before = free_bytes()
    for k in range(12):
        cycle(k)
    lost = before - free_bytes()
    print("device memory lost over 12 create/destroy cycles: %.1f MB" % (lost / 1e6))
    assert lost < 48e6, lost


def test_bench_spawns_two_ranks_without_torch(gpu):
    """`python bench.py --gpus 2` with no launcher: two ranks, z-slab sharded, halo exchange through the shared-memory
    transport (two ranks on this box's one GPU), barriers / reductions through sph_comm_*; no torch import."""
    out = _bench(["--gpus", "2", "--config", "c1", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--motion-step", "0"],
                 {"SPH_COMM_TRANSPORT": "shm+ipc", "SPH_BENCH_SELFTEST": "1"})
    assert out["n_gpus"] == 2 and out["config"]["parallelism"].startswith("z-slab x2, ipc-push+shm"), out["config"]
    assert out["config"]["particles"] == 8000 and out["value"] > 0 and len(out["repeat_ms_per_step"]) == 3
    one = _bench(["--gpus", "1", "--config", "c1", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--motion-step", "0"], {})
    assert one["n_gpus"] == 1
    # the same steps of the same scene: the sharded run finds exactly the pairs the single-GPU run finds
    assert out["config"]["pair_interactions_per_step"] == one["config"]["pair_interactions_per_step"]
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "import torch" not in src and "torch.distributed" not in src.replace("torch.distributed.run", "")


def test_bench_eight_ranks_share_one_gpu(gpu):
    """The driver's 8-rank job -- BASELINE.json's metric as written: the 1,231,200-particle scene itself split over the 8 ranks (strong
    scaling; every rank topology: two edge ranks, six with both neighbours; 5-7 cell layers per rank) -- and the same job with one C2 block
    per rank (--scaling weak, 8 x 1,231,200 particles), with all ranks on this box's ONE GPU -- push transport, asynchronous steps, 8-way control plane.  Round 3 found here that workgroups spinning on a
    neighbour's message hold their CU slots: with every workgroup of the consuming kernels polling, 8 ranks filled the chip with pollers and
    the kernels that had to produce the awaited messages were never scheduled (all waits timed out); the waiting kernels are capped at 64
    workgroups now.  Oversubscription only -- one rank per GPU cannot starve itself -- but a hang is a hang."""
    common = ["--gpus", "8", "--steps", "5", "--warmup", "2", "--repeats", "1", "--no-cpu-baseline", "--motion-step", "0", "--no-extras"]
    env = {"SPH_COMM_TRANSPORT": "shm+ipc", "SPH_COMM_TIMEOUT_S": "60"}
    out = _bench(common, env, timeout=900)
    assert out["n_gpus"] == 8 and out["config"]["particles"] == 1231200 and out["scaling"] == "strong"
    assert out["config"]["workload"] == "C2 1,231,200-particle dam break"   # the N = 1 line's workload, byte for byte
    assert out["config"]["parallelism"].startswith("z-slab x8, ipc-push+shm") and out["value"] > 0
    # ~26 lattice neighbours (+ some of the 6 at exactly h, decided by rounding; fewer at the block's faces) x 4 reference passes
    assert 4 * 24 * 1231200 < out["config"]["pair_interactions_per_step"] < 4 * 33 * 1231200
    out = _bench(common + ["--scaling", "weak"], env, timeout=900)
    assert out["n_gpus"] == 8 and out["config"]["particles"] == 8 * 1231200 and out["scaling"] == "weak"
    assert out["config"]["workload"] == "C2 1,231,200-particle dam break x8 in z"
    assert out["config"]["parallelism"].startswith("z-slab x8, ipc-push+shm") and out["value"] > 0


def test_bench_stops_all_ranks_when_one_dies(gpu):
    """A rank that dies leaves its neighbour blocked in a halo receive; the launcher stops the survivors (its own children, by
    pid) and exits non-zero instead of hanging."""
    env = dict(os.environ, SPH_COMM_TRANSPORT="shm+ipc", SPH_BENCH_FAIL_RANK="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "c1", "--steps", "5", "--warmup", "2",
                        "--no-cpu-baseline", "--motion-step", "0"], env=env, capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert r.returncode != 0 and "rank exit codes" in r.stderr, r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")], "no result line from a run that lost a rank"
    assert time.time() - t0 < 200


def test_bench_rejects_world_size_mismatch(gpu):
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stderr + r.stdout)


def test_rccl_transport_executes_on_hardware(gpu):
    """One-rank RCCL communicator: ncclCommInitRank, ncclGroupStart / ncclSend / ncclRecv (to self) / ncclGroupEnd with
    every word checked, ncclAllReduce (sph_comm_selftest, sph_comm_allreduce, sph_comm_barrier), then a slab-mode WCSPH
    run over that communicator (bench.py one-rank slab mode) that must agree with the plain single-GPU run."""
    env = dict(os.environ)
    env.pop("SPH_COMM_TRANSPORT", None)
    code = (
        "import ctypes, numpy as np\n"
        "from sph_project_amd import _lib as L, product as P\n"
        "lib = L.load(); buf = ctypes.create_string_buffer(128); assert lib.sph_comm_unique_id(buf) == 0\n"
        "c, s = P.build_product(P.dam_break_scene(end=(0.1, 0.1, 0.1)), comm=dict(rank=0, nranks=1, unique_id=buf.raw))\n"
        "e = c.engine; e.comm_selftest(1 << 18)\n"
        "assert e.comm_allreduce([1.5, -2.0, 7.0], 'sum') == [1.5, -2.0, 7.0]\n"
        "assert e.comm_allreduce([3.0], 'max') == [3.0]; e.comm_barrier(); print('RCCL_OK')\n")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    args = ["--config", "c1", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--motion-step", "0"]
    slab = _bench(args, {"SPH_BENCH_FORCE_SLAB": "1", "SPH_COMM_TRANSPORT": "rccl", "SPH_BENCH_SELFTEST": "1"})
    plain = _bench(args, {})
    assert slab["config"]["pair_interactions_per_step"] == plain["config"]["pair_interactions_per_step"]
    # the default data plane on top of the RCCL control plane: push transport (no neighbour to map on one rank, but inbox, device-resident
    # counts, asynchronous steps and launch bounds are all in play)
    push = _bench(args, {"SPH_BENCH_FORCE_SLAB": "1", "SPH_BENCH_SELFTEST": "1", "SPH_COMM_VERBOSE": "1"})
    assert push["config"]["pair_interactions_per_step"] == plain["config"]["pair_interactions_per_step"]


# --------------------------------------------------------------------------------------------- dynamic rigid body, end to end
_CUBE_OBJ = """v -0.05 -0.05 -0.05
v 0.05 -0.05 -0.05
v 0.05 0.05 -0.05
v -0.05 0.05 -0.05
v -0.05 -0.05 0.05
v 0.05 -0.05 0.05
v 0.05 0.05 0.05
v -0.05 0.05 0.05
f 1 3 2
f 1 4 3
f 5 6 7
f 5 7 8
f 1 2 6
f 1 6 5
f 2 3 7
f 2 7 6
f 3 4 8
f 3 8 7
f 4 1 5
f 4 5 8
"""


def test_dynamic_rigid_body_scene_runs_end_to_end(gpu, tmp_path):
    """A scene with `isDynamic: true` (every reference scene with a floating / falling body): mesh -> particles on the
    host, fluid -> rigid wrench on the device, the host rigid solver integrates the body (native backend; PyBullet is not
    in this image), pose -> particles on the device, mesh_object_{id}.obj per frame with exportObj."""
    (tmp_path / "cube.obj").write_text(_CUBE_OBJ)
    cfg = H.dam_break_scene(domain_end=(0.6, 0.8, 0.6), end=(0.3, 0.16, 0.3), translation=(0.12, 0.06, 0.12), dt=4e-4, viscosity_b=0.5)
    cfg["Configuration"].update(exportPly=True, exportObj=True, fps=250, totalTime=0.024)   # 60 rounds, a frame every 10
    cfg["RigidBodies"] = [{"objectId": 1, "geometryFile": str(tmp_path / "cube.obj"), "isDynamic": True, "entryTime": -1.0,
                           "density": 600.0, "color": [200, 50, 50], "velocity": [0.0, -1.0, 0.0], "translation": [0.27, 0.29, 0.27],
                           "scale": [1, 1, 1], "rotationAngle": 0.0, "rotationAxis": [0, 1, 0]}]
    container, solver = H.build_product(cfg)
    solver.prepare()
    e = container.engine
    rigid = e.download(L.F_MATERIAL) == 2
    assert 64 <= rigid.sum() <= 400
    for _ in range(250):
        solver.step()
    b = solver.rigid_solver.bodies[1]
    ids = e.download(L.F_PARTICLE_ID)
    x = H.by_id(ids, e.download(L.F_POSITION))
    mat = H.by_id(ids, e.download(L.F_MATERIAL))
    r = mat == 2
    assert np.isfinite(x).all()
    # the body fell into the fluid (it started 0.04 above the surface at 1 m/s), was decelerated by it, and its
    # particles moved rigidly with the pose the host pushed
    assert b.com[1] < 0.29 - 0.05, b.com
    assert b.vel[1] > -1.0 - 9.81 * 250 * 4e-4 + 0.2, ("the fluid never pushed back", b.vel)
    pts = np.asarray(container.rigid_bodies[0]["voxelizedPoints"], dtype=np.float64)   # body frame = insertion order
    np.testing.assert_allclose(x[r], b.com + pts @ b.rot.T, atol=2e-6, err_msg="particles = com + R (rest position)")
    del container, solver
    # the driver writes the OBJ / PLY tree
    scene_file = tmp_path / "float_cube.json"
    scene_file.write_text(json.dumps(cfg))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "sph_project_amd", "run_simulation.py"), "--scene_file", str(scene_file)],
                       cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = tmp_path / "float_cube_output"
    assert sorted(os.listdir(out)) == ["%06d" % k for k in range(0, 60, 10)]
    from sph_project_amd import meshgen
    m0 = meshgen.load_obj(str(out / "000000" / "mesh_object_1.obj"))
    m5 = meshgen.load_obj(str(out / "000050" / "mesh_object_1.obj"))
    assert m0.vertices.shape == (8, 3) and m5.vertices[:, 1].mean() < m0.vertices[:, 1].mean() - 0.01
    assert (out / "000050" / "particle_object_0.ply").exists()


def test_bench_under_torch_distributed_run(gpu):
    """The driver's launch line for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...`.  torch only starts the processes (RANK / LOCAL_RANK / WORLD_SIZE in
    the environment); the ranks find each other through the /dev/shm rendezvous file, talk through sph_comm_*, and the
    extra C4 strong-scaling measurement runs on a communicator of its own."""
    env = dict(os.environ, SPH_COMM_TRANSPORT="shm+ipc")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "SPH_BENCH_RDV"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--repeats", "1",
           "--motion-step", "0", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    out = json.loads(lines[0])
    # the headline is BASELINE.json's metric as written: the 1.23 M scene itself over the N ranks
    assert out["n_gpus"] == 2 and out["config"]["particles"] == 1231200 and out["scaling"] == "strong"
    assert out["config"]["workload"] == "C2 1,231,200-particle dam break" and out["config"]["parallelism"].startswith("z-slab x2")
    c4 = out["c4_strong_scaling"]
    assert c4["particles"] == 4000000 and c4["n_gpus"] == 2 and c4["scaling"] == "strong" and c4["value"] > 0
    assert sum(c4["owned_per_rank"]) == 4000000 and c4["halo_transport"] == "ipc-push+shm"
    c2w = out["c2_weak_scaling"]     # one C2 block per rank: an extra, never `value`
    assert c2w["particles"] == 2 * 1231200 and c2w["scaling"] == "weak" and sum(c2w["owned_per_rank"]) == 2 * 1231200 and c2w["value"] > 0
    assert "c2_strong_scaling" not in out
