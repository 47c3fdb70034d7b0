"""GPU: the HIP path against the committed golden fixtures (tests/golden/*.npz: the reference's own
source run under a serial f32 interpreter, see oracle/gen_golden.py) for every solver that is built."""
import glob
import json
import os

import numpy as np
import pytest

from sph_project_amd import _lib as L
from tests import helpers as H

pytestmark = pytest.mark.gpu

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
FIELDS = {"positions": L.F_POSITION, "velocities": L.F_VELOCITY, "densities": L.F_DENSITY, "pressures": L.F_PRESSURE,
          "accelerations": L.F_ACCELERATION, "rest_volumes": L.F_REST_VOLUME, "masses": L.F_MASS,
          "alphas": L.F_DFSPH_ALPHA, "kappa": L.F_DFSPH_KAPPA, "kappa_v": L.F_DFSPH_KAPPA_V,
          "densities_star": L.F_DENSITY_STAR, "densities_derivatives": L.F_DENSITY_DERIV}
# accelerations are only comparable where the reference leaves the same thing in the field
ACC_METHODS = ("wcsph", "pcisph")


def slot_fluid_now(z, pre):
    return z[pre + "materials"] == 1


@pytest.mark.parametrize("fast_math", [0, 1])
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_hip_matches_golden(gpu, path, fast_math):
    z = np.load(path)
    cfg = json.loads(bytes(z["scene_json"]).decode())
    method = cfg["Configuration"]["simulationMethod"]
    container, solver = H.build_product(cfg, fast_math=fast_math)
    container.insert_object()
    solver.rigid_solver.insert_rigid_object()
    e = container.engine
    if "inject_count" in z.files:
        # the generator's dynamic rigid body (oracle/gen_golden.py inject_rigid): same particles, identity pose about the centroid
        k = int(z["inject_count"])
        obj = int(z["init_object_ids"][-1])
        e.set_object(obj, 2, 1)
        e.append_particles(obj, z["init_positions"][-k:], z["init_velocities"][-k:], z["init_densities"][-k:], np.zeros(k, np.float32),
                           z["init_materials"][-k:], z["init_is_dynamic"][-k:], np.zeros((k, 3), np.int32))
        zero3 = np.zeros(3, np.float32)
        e.set_rigid_pose(obj, z["inject_com"], np.eye(3, dtype=np.float32), zero3, zero3, com0=z["inject_com"])
    assert e.particle_num == z["init_positions"].shape[0]
    np.testing.assert_array_equal(e.download(L.F_MATERIAL), z["init_materials"])
    e.upload(L.F_POSITION, z["init_positions"])  # jittered scenes: same seeded perturbation as the fixture
    solver.prepare()
    np.testing.assert_array_equal(e.download(L.F_PARTICLE_ID), z["prep_ids"])
    np.testing.assert_array_equal(e.download(L.F_POSITION), z["prep_positions"])
    step = 0
    slot_fluid, slot_fluid_step = z["prep_materials"] == 1, 0   # who occupied each slot before the last sort
    for cp in z["checkpoints"]:
        while step < cp:
            if method == "wcsph" and step == cp - 1:   # the positions the checkpoint's forces are evaluated at
                ids_before, x_before, mat_before = e.download(L.F_PARTICLE_ID), e.download(L.F_POSITION), e.download(L.F_MATERIAL)
            solver.step()
            step += 1
            if "pose_step" in z.files and step == int(z["pose_step"]):   # the pose a rigid solver would have written
                e.set_rigid_pose(int(z["init_object_ids"][-1]), z["pose_com"], z["pose_rot"], z["pose_vel"], z["pose_angvel"],
                                 com0=z["inject_com"])
        pre = f"s{cp}_"
        ids = e.download(L.F_PARTICLE_ID)
        fluid = H.by_id(z[pre + "ids"], z[pre + "materials"]) == 1
        x = H.by_id(ids, e.download(L.F_POSITION))
        xr = H.by_id(z[pre + "ids"], z[pre + "positions"])
        d = H.drift(x, xr, container.dh).max()
        assert d <= 1e-5, (cp, d)
        worst = {}
        for key, fid in FIELDS.items():
            if pre + key not in z.files or (key == "accelerations" and method not in ACC_METHODS):
                continue
            try:
                raw = e.download(fid)
            except L.SphError:
                continue
            if method == "dfsph" and key in ("kappa", "densities_star"):
                # computed before the end-of-step sort and not reordered by it (base_container.py:506 list):
                # slot-indexed in the reference and here alike, so compare slot by slot
                # ... and only where the slot held a fluid particle when the value was written (slots that held a
                # boundary particle keep stale values of earlier steps in the reference)
                if slot_fluid_step != cp - 1 and not slot_fluid.all():
                    continue
                if slot_fluid.shape[0] != raw.shape[0]:   # a late entryTime block came in since the last checkpoint
                    continue
                scale = max(float(np.abs(z[pre + key]).max()), 1e-30)
                worst[key] = float(np.abs(raw.astype(np.float64) - z[pre + key].astype(np.float64))[slot_fluid].max()) / scale
                continue
            mine = H.by_id(ids, raw)
            ref = H.by_id(z[pre + "ids"], z[pre + key])
            if key not in ("positions", "velocities", "rest_volumes", "masses"):
                mine, ref = mine[fluid], ref[fluid]
            scale = max(float(np.abs(ref).max()), 1e-30)
            err = np.abs(mine.astype(np.float64) - ref.astype(np.float64))
            if key == "alphas" and pre + "densities" in z.files:
                # alpha_i = -1 / (sum_j |V_j grad W_ij|^2 + |sum_j V_j grad W_ij|^2), grad W ~ (1 - q)^2 for q > 1/2: a particle
                # whose few neighbours all sit near the edge of its support (rho_i well below rho0: spray, the first particles out of
                # an emitter) has every term carry the cancellation 1 - q, amplification 4 / (1 - q) per term -- the fast build's
                # v_rcp / v_rsq (1 ulp in q) then show up as ~1e-4 in alpha of that ONE particle (tools/debug_alpha.py:
                # dfsph_implicit_box, particle 1019 at rho = 272: 4.6e-5 or 8.8e-5 depending on how the compiler contracts the
                # surrounding code into FMAs; median over the fluid 7e-7, p99 5e-6; the strict build 1e-6 everywhere).  Parity is
                # asserted on the particles with a populated neighbourhood; the sparse ones get a guard.
                rho = H.by_id(z[pre + "ids"], z[pre + "densities"])[fluid]
                dense = rho >= 0.5 * float(cfg["Configuration"]["density0"])
                worst["alphas_sparse"] = float(err[~dense].max()) / scale if (~dense).any() else 0.0
                err = err[dense] if dense.any() else err[:0]
            worst[key] = (float(err.max()) / scale) if err.size else 0.0
        if pre + "cg_x" in z.files:
            # CG warm start: slot-indexed in the reference (not reordered by the sort) and here; it holds x - v
            cgx = e.download(L.F_CG_X)
            vs = max(float(np.abs(z[pre + "velocities"]).max()), 1e-30)
            worst["cg_x"] = float(np.abs(cgx.astype(np.float64) - z[pre + "cg_x"].astype(np.float64))[slot_fluid_now(z, pre)].max()) / vs
            assert abs(solver.stats()["iter_cg"] - int(z[pre + "iter_cg"])) <= 2, (solver.stats()["iter_cg"], int(z[pre + "iter_cg"]))
        if "inject_count" in z.files:
            # running sums of force / torque on the dynamic rigid body (nothing resets them in the fixture run)
            force, torque = e.get_rigid_wrench(reset=False)
            for key, mine in (("rigid_forces", force), ("rigid_torques", torque)):
                ref = z[pre + key].astype(np.float64)
                assert np.abs(ref).max() > 0
                worst[key] = float(np.abs(mine[:ref.shape[0]].astype(np.float64) - ref).max() / np.abs(ref).max())
        # PARITY limits (u = 2^-24): well-conditioned quantities -- sums of <= ~60 same-signed pair terms with ~20 roundings each
        # (20 u * 1 ~ 1e-6 per particle; 2e-5 leaves room for pow() vs t*t*t near q = 1 and for the fast build's v_rcp / v_rsq),
        # velocities / positions integrated from them over <= 30 steps (derivation: tests/test_big_golden.py).
        parity = {"positions": 1e-5, "velocities": 5e-5, "densities": 2e-5, "rest_volumes": 5e-6, "masses": 5e-6, "alphas": 5e-5,
                  "densities_star": 1e-5, "cg_x": 2e-5, "rigid_forces": 2e-5, "rigid_torques": 5e-5, "acc_backward": 1.0, "drho_backward": 5e-6}
        # REGRESSION GUARDS, not parity claims: quantities that cancel (D rho / Dt, kappa, accelerations: |sum| << sum |terms|) or
        # amplify (p = 50000 ((rho/rho0)^7 - 1): a density difference times 7 * 50000 / max|p|).  Their error relative to the
        # field's maximum is the conditioning of the scene, not of the code; the numbers below are a few times the worst error the
        # fixtures show in either build ("limits fitted to pass") and are kept only to catch a formula that breaks.
        # Where the stored state allows it they are REPLACED below by per-term checks against a float64 re-evaluation with a derived
        # bound: WCSPH pressures (EOS) and accelerations (pressure force), DFSPH D rho / Dt.  kappa / kappa_v (they belong to the velocities
        # of the last solver iteration, which no field of the reference keeps) and PCISPH's accumulated pressure keep a fitted number HERE,
        # but since round 6 only as regression guards: their formulas are pinned from the product's own inputs in tests/test_hip_solvers.py
        # (test_dfsph_kappa_per_term: kappa = (rho* - 1) alpha / dt to 3 u against the pass's own rho* and alpha, kappa_v = D rho / Dt x alpha
        # bit for bit; test_pcisph_pressure_update_per_term: p = max(0, p_prev + k (rho0 - rho*)) bit for bit), their inputs by the parity
        # limits above.  What is left fitted with nothing behind it: alpha of particles with a near-empty neighbourhood (conditioning, see above).
        guard = {"kappa": 1e-3, "kappa_v": 1e-3, "densities_derivatives": 4e-3, "pressures": 2e-3, "accelerations": 2e-3, "alphas_sparse": 1e-3}
        if method == "wcsph" and pre + "pressures" in z.files:
            # PER-TERM check instead of the fitted guard: the EOS p = 50000 ((rho / rho0)^7 - 1) (WCSPH.py:17-24) on the
            # product's OWN stored (clamped) density -- the density is pinned to the fixture by the parity limit above, the formula here:
            # rounding of pow / x^7 by squaring is a few u of 50000 x^7.  The same relation must hold inside the fixture (sanity of the check).
            rho0 = float(cfg["Configuration"]["density0"])
            for tag, rr, pp in (("fixture", z[pre + "densities"][z[pre + "materials"] == 1], z[pre + "pressures"][z[pre + "materials"] == 1]),
                                ("hip", H.by_id(ids, e.download(L.F_DENSITY))[fluid], H.by_id(ids, e.download(L.F_PRESSURE))[fluid])):
                x7 = (rr.astype(np.float64) / rho0) ** 7
                err = np.abs(pp.astype(np.float64) - 50000.0 * (x7 - 1.0)) / (50000.0 * x7)
                assert rr.min() >= rho0 and err.max() <= 4e-6, (cp, tag, float(err.max()))
            worst.pop("pressures", None)
            # ... and the pressure acceleration, the field the step leaves in `accelerations` (WCSPH.py:63-70): every pair term recomputed
            # in float64 from the product's own density / pressure and the positions the step started from.  Bound per component:
            # ~12 roundings per term (difference, r, q, the polynomial, p / rho^2 twice, three products; v_rsq / v_rcp 1 ulp each in the
            # fast build) + a sum of <= 64 terms in any order: (12 + 64) u = 4.5e-6 of sum_j |term_j|; 1e-5 asserted -- plus the ~3
            # roundings that reach q = r / h times the conditioning of the kernel gradient in q (amp_j, see the helper: a neighbour
            # near the edge of the support contributes (1 - q)^2, i.e. 2 q / (1 - q) times the error of q): 5e-7 of sum_j |term_j| amp_j.
            # This is a backward-error statement (relative to sum |terms|, not to |sum|), so it holds however badly the scene cancels.
            if len(ids_before) == len(ids):   # no emitter / late block came in during the step
                get = lambda fid: H.by_id(ids, e.download(fid))
                mat_now = get(L.F_MATERIAL)
                if np.array_equal(mat_now, H.by_id(ids_before, mat_before)) and not (set(np.unique(mat_now)) - {1, 2}):
                    a64, mag, mag_amp = H.wcsph_pressure_accel_f64(H.by_id(ids_before, x_before), get(L.F_DENSITY), get(L.F_PRESSURE), get(L.F_MASS),
                                                          get(L.F_REST_VOLUME), mat_now, container.dh, rho0)
                    err = np.abs(get(L.F_ACCELERATION).astype(np.float64) - a64)[fluid]
                    bound = 1e-5 * mag[fluid] + 5e-7 * mag_amp[fluid] + 1e-30
                    assert (err <= bound).all(), (cp, "pressure acceleration", float((err / bound).max()))
                    worst.pop("accelerations", None)
                    worst["acc_backward"] = float((err / bound).max())   # fraction of the derived bound
        if method == "dfsph" and "densities_derivatives" in worst:
            # PER-TERM check instead of the fitted guard: the divergence solve is the last thing a DFSPH step does (DFSPH.py:316-319), so
            # the stored D rho / Dt belongs to the stored positions and velocities: max(sum_j V_j (v_i - v_j) . grad W_ij, 0), zero below
            # 20 neighbours (DFSPH.py:66-98), recomputed in float64 from the product's OWN state.  ~10 roundings per term + a sum of
            # <= 64 terms: 74 u = 4.4e-6 of sum |terms|; 5e-6 asserted (the fixtures themselves sit at <= 3.3e-7 in this measure).
            get = lambda fid: H.by_id(ids, e.download(fid))
            mat_now = get(L.F_MATERIAL)
            val, mag, n_lo, n_hi = H.dfsph_density_derivative_f64(get(L.F_POSITION), get(L.F_VELOCITY), get(L.F_REST_VOLUME), mat_now, container.dh)
            got = get(L.F_DENSITY_DERIV).astype(np.float64)
            fl = mat_now == 1
            assert (got[fl & (n_hi < 20)] == 0).all(), (cp, "D rho / Dt below 20 neighbours")
            sure = fl & (n_lo >= 20)
            err = np.abs(got - np.maximum(val, 0.0))[sure]
            assert sure.sum() > 0.2 * fl.sum() and (err <= 5e-6 * mag[sure] + 1e-30).all(), (cp, "D rho / Dt", float((err / (mag[sure] + 1e-30)).max()))
            worst["drho_backward"] = float((err / (mag[sure] + 1e-30)).max()) if err.size else 0.0
            worst.pop("densities_derivatives")
        for k, v in worst.items():
            assert v < parity.get(k, guard.get(k)), (cp, k, v, worst)
        slot_fluid, slot_fluid_step = z[pre + "materials"] == 1, cp
    print(os.path.basename(path), "fast" if fast_math else "strict", "final drift %.2e" % d, worst)
