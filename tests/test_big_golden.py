"""Fixtures at BASELINE configs[0] size (tests/golden/big/*.npz, generator: oracle/gen_golden.py BIG_SCENES): the
reference's own, unmodified source files run under oracle/taichi_shim (a serial f32 interpreter; NOT a Taichi run) on

  * c1_wcsph            : configs[0] exactly -- 20^3 = 8,000-particle cube, WCSPH, dt 4e-4, 40 steps, checkpoints 1/5/10/20/40;
  * c1_wcsph_jitter     : the same block perturbed and moving, 20 steps;
  * dfsph_4k, pcisph_4k : 16^3 = 4,096 particles, 10 steps, the solvers' own stop tests, iteration history of every step;
  * *_4k_compressed     : the same block packed tighter than the rest spacing, so that the solver loops iterate;
  * visc_4k             : the 16^3 block under DFSPH + implicit viscosity (the path of configs[4]), CG iteration history of every step;
  * wcsph_box_4k        : the 16^3 block falling onto the floor of a sampled domain box (3,429 boundary particles, 20 steps): the rigid-aware
                          instantiations of every pass at size.  (An impact: densities follow the drifted positions, measured drho ~1e-5.)
  * dfsph_box_4k        : the same impact under DFSPH, the first TWO steps (6 + 4 divergence and 19 + 8 density iterations with the rigid terms
                          of alpha and of D rho / Dt).  Not more: from the third step on a hard DFSPH impact is chaotic -- the C oracle and
                          the interpreter agree in every iteration count of ten steps and are 2e-5 apart after 5, 1e-2 after 10.

Checked here: the CPU oracle (every run of the CPU suite) and the HIP path, strict and fast build (GPU suite), with the SURVEY 8(c)
metric -- per-particle position drift relative to max(|x|, dh), matched by particle id -- at EVERY checkpoint, the velocities, and the
iteration histories (+-1: reduction order).

Limits and where they come from (u = 2^-24, one f32 rounding):
  * drift <= 1e-4 is the north-star bar (BASELINE.json); what is asserted is 2e-5, an order below it.  The measured values are printed;
    after 40 steps of C1 they are ~1e-6: a position is x0 + dt * sum of k velocities, each carrying O(10 u) relative rounding from a
    ~60-term neighbour sum, so drift ~ k * dt * |v| * 10 u * sqrt(60) / dh = 40 * 4e-4 * 1 * 5e-6 / 0.04 ~ 2e-6.
  * velocities: |dv| <= 1e-4 * max|v|: one step adds dt * a with a = a sum of ~60 pair terms of either sign whose absolute values add
    up to ~50x the result (pressure + viscosity cancel against gravity), each with ~20 roundings: 50 * 20 u * sqrt(60) ~ 5e-4 of |a|,
    times dt |a| / |v| ~ 4e-4 * 300 / 0.5 ~ 0.25 per step, accumulating like a random walk over k <= 40 steps: ~ 1e-4 * 0.25 * 6 ~ 1e-4.
  * densities: a sum of ~60 positive terms: 20 u * 1 = 1.2e-6 per particle; asserted 2e-5 (the cubic (1 - q)^3 is evaluated as t*t*t by
    both C paths and as pow() by the reference: 2 ulp more on the few terms near q = 1).
Pressures are REPORTED, not asserted: p = 50000 ((rho / rho0)^7 - 1) amplifies a density difference by 7 * 50000 / max|p| (hundreds
when the block is barely compressed) -- any limit would be a number fitted to pass.
"""
import glob
import json
import os

import numpy as np
import pytest

from oracle import ref as oracle_ref
from sph_project_amd import _lib as L, scene
from sph_project_amd.SPH.utils import SimConfig
from tests import helpers as H

BIG = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "big", "*.npz")))
IDS = [os.path.basename(p)[:-4] for p in BIG]
DRIFT_LIMIT, VEL_LIMIT, RHO_LIMIT = 2e-5, 1e-4, 2e-5
HIST = (("hist_iter_v", "iter_divergence", "last_iter_div"), ("hist_iter_d", "iter_density", "last_iter_den"),
        ("hist_iter_pci", "iter_pcisph", "last_iter_pci"), ("hist_iter_cg", "iter_cg", "last_iter_cg"))
# +-1 for the DFSPH / PCISPH loops (a mean error against a threshold: reduction order); +-2 for CG, whose stop test is on a
# residual that falls by a factor per iteration and whose three dot products per iteration are each summed in another order here
HIST_TOL = {"hist_iter_cg": 2}


def _load(path):
    z = np.load(path)
    return z, json.loads(bytes(z["scene_json"]).decode())


def _check(z, pre, ids, x, v, rho, prs, dh, tag):
    fl = H.by_id(z[pre + "ids"], z[pre + "materials"]) == 1
    xr, vr = H.by_id(z[pre + "ids"], z[pre + "positions"]), H.by_id(z[pre + "ids"], z[pre + "velocities"])
    rr, pr = H.by_id(z[pre + "ids"], z[pre + "densities"]), H.by_id(z[pre + "ids"], z[pre + "pressures"])
    assert np.array_equal(np.sort(ids), np.sort(z[pre + "ids"]))
    d = H.drift(H.by_id(ids, x), xr, dh)
    dv = float(np.abs(H.by_id(ids, v).astype(np.float64) - vr).max() / max(float(np.abs(vr).max()), 1e-30))
    drho = float(np.abs(H.by_id(ids, rho).astype(np.float64) - rr)[fl].max() / float(np.abs(rr[fl]).max()))
    pscale = float(np.abs(pr[fl]).max())
    dp = float(np.abs(H.by_id(ids, prs).astype(np.float64) - pr)[fl].max() / pscale) if pscale > 0 else 0.0
    print("%s %s drift max %.2e p99 %.2e | dv/vmax %.2e | drho/rho %.2e | dp/pmax %.2e (reported; p amplification %.0f)" % (
        tag, pre, d.max(), np.percentile(d, 99), dv, drho, dp, 7 * 50000.0 / pscale if pscale > 0 else 0.0))
    assert d.max() <= DRIFT_LIMIT, (tag, pre, d.max())
    assert dv <= VEL_LIMIT, (tag, pre, dv)
    assert drho <= RHO_LIMIT, (tag, pre, drho)
    return float(d.max())


@pytest.mark.parametrize("path", BIG, ids=IDS)
def test_oracle_matches_reference_source_at_config0_size(path):
    """CPU: oracle/sph_ref.c against the big fixtures -- the pin of the checker itself at the size SURVEY 8(c) asked for."""
    z, cfg = _load(path)
    from tests.test_oracle_golden import _oracle_from_fixture   # objects in insertion order (fluid blocks, then the sampled domain box)
    sim, geo = _oracle_from_fixture(z, cfg)
    sim.prepare()
    np.testing.assert_array_equal(H.oracle_ids(sim), z["prep_ids"])            # the sort is exact
    np.testing.assert_array_equal(sim.field("particle_positions"), z["prep_positions"])
    step, hist = 0, {k: [] for k, _, _ in HIST}
    for cp in z["checkpoints"]:
        while step < cp:
            sim.step(1)
            step += 1
            for k, _, name in HIST:
                hist[k].append(int(sim.scalar(name)))
        _check(z, f"s{cp}_", H.oracle_ids(sim), sim.field("particle_positions").copy(), sim.field("particle_velocities").copy(),
               sim.field("particle_densities").copy(), sim.field("particle_pressures").copy(), geo.dh, "oracle")
    for k, _, _ in HIST:
        if k in z.files:
            print("oracle", k, hist[k], "reference", list(z[k]))
            assert np.abs(np.array(hist[k]) - z[k]).max() <= HIST_TOL.get(k, 1), (k, hist[k], list(z[k]))


@pytest.mark.gpu
@pytest.mark.parametrize("fast_math", [0, 1], ids=["strict", "fast"])
@pytest.mark.parametrize("path", BIG, ids=IDS)
def test_hip_matches_reference_source_at_config0_size(gpu, path, fast_math):
    """GPU: the product path (through the C-ABI, driven like run_simulation.py drives the reference) against the big fixtures."""
    z, cfg = _load(path)
    container, solver = H.build_product(cfg, fast_math=fast_math)
    container.insert_object()
    solver.rigid_solver.insert_rigid_object()
    e = container.engine
    assert e.particle_num == z["init_positions"].shape[0]
    e.upload(L.F_POSITION, z["init_positions"])   # the generator's seeded perturbation
    solver.prepare()
    np.testing.assert_array_equal(e.download(L.F_PARTICLE_ID), z["prep_ids"])
    np.testing.assert_array_equal(e.download(L.F_POSITION), z["prep_positions"])
    step, hist = 0, {k: [] for k, _, _ in HIST}
    for cp in z["checkpoints"]:
        while step < cp:
            solver.step()
            step += 1
            st = solver.stats()
            for k, name, _ in HIST:
                hist[k].append(int(st[name]))
        _check(z, f"s{cp}_", e.download(L.F_PARTICLE_ID), e.download(L.F_POSITION), e.download(L.F_VELOCITY), e.download(L.F_DENSITY),
               e.download(L.F_PRESSURE), container.dh, "hip " + ("fast" if fast_math else "strict"))
    for k, _, _ in HIST:
        if k in z.files:
            print("hip", k, hist[k], "reference", list(z[k]))
            assert np.abs(np.array(hist[k]) - z[k]).max() <= HIST_TOL.get(k, 1), (k, hist[k], list(z[k]))
