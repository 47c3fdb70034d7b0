#!/usr/bin/env python
"""Generate tests/golden/*.npz by executing the reference's UNMODIFIED source files
(/root/reference/SPH/**/*.py) under oracle/taichi_shim -- a serial f32 interpreter of the Taichi
constructs they use.  Runs only in the build container (the reference never travels); the
fixtures it writes are plain data (inputs + expected outputs) and are committed.

Label of every fixture: "reference source under a serial f32 interpreter" -- NOT a Taichi run.

    python oracle/gen_golden.py [scene ...]        # all scenes if none named

Per scene the reference objects are driven exactly like run_simulation.py drives them
(XContainer(config) -> XSolver(container) -> prepare() -> step() ...), with two generator-side
additions between insert_object() and the first sort, both through the reference's own fields:
a persistent particle id is written into particle_colors (reordered with the particles and unused
by the physics: base_container.py:528/:541), and -- for the "jitter" scenes -- a seeded
perturbation is added to the fluid lattice so that symmetric cancellations do not hide errors.
"""
import contextlib
import io
import json
import os
import re
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "taichi_shim"))
sys.path.insert(0, "/root/reference")
sys.path.insert(1, ROOT)

from tests.helpers import dam_break_scene  # noqa: E402  (scene dicts shared with the tests)

OUT = os.path.join(ROOT, "tests", "golden")
LATE_COLOR = [100000, 7, 7]   # sentinel colour of particles that enter late (replaced by persistent ids once they are in)


def late_scene(method, dt):
    """Block 0 present from the start, block 1 enters right on top of it (one lattice spacing away, so the two interact
    at once) when total_time reaches 2.5 dt (= during the 4th step).  Its extents are not multiples of the spacing:
    the reference budgets particle_max_num with the untranslated corners (base_container.py:89) and np.arange's
    end-point rounding must not make the translated cube any larger."""
    cfg = dam_break_scene(method=method, end=(0.12, 0.1, 0.12), dt=dt, velocity=(0.0, -0.5, 0.0))
    cfg["FluidBlocks"].append({
        "objectId": 1, "start": [0.0, 0.0, 0.0], "end": [0.07, 0.07, 0.09], "translation": [0.12, 0.2, 0.11],
        "scale": [1, 1, 1], "velocity": [0.0, -1.0, 0.0], "density": 1000.0, "color": LATE_COLOR, "entryTime": 2.5 * dt,
    })
    return cfg

def rigid_scene(method, dt):
    """A dynamic rigid body in the fluid's support, for the wrench terms (base_solver.py:240-278 viscosity, :147-186 pressure;
    DFSPH.py:173-203, :255-283).  The reference builds rigid bodies from meshes (trimesh) and moves them with PyBullet, neither
    of which exists here, so the generator puts the body's particles in through the reference's own add_particles and
    rigid_body_* fields (see inject_rigid).  With no RigidBodies in the scene file the reference's PyBulletSolver.step() returns
    at once (bullet_solver.py:145): the body stays where it is and rigid_body_forces / _torques accumulate over the steps --
    that running sum is what the fixture records.  Room for the body in particle_max_num comes from a fluid block that never
    enters (entryTime far in the future) and lends the body its object id."""
    cfg = dam_break_scene(method=method, end=(0.14, 0.14, 0.14), particleSpacing=0.019, viscosity_b=0.4, dt=dt,
                          velocity=(0.1, -0.4, 0.05))
    cfg["FluidBlocks"].append({
        "objectId": 1, "start": [0.0, 0.0, 0.0], "end": [0.09, 0.09, 0.09], "translation": [0.6, 0.6, 0.6],
        "scale": [1, 1, 1], "velocity": [0.0, 0.0, 0.0], "density": 1000.0, "color": LATE_COLOR, "entryTime": 1.0e6,
    })
    return cfg


# body of the rigid scenes: 4 x 4 x 4 particles at the lattice spacing, under the fluid block and inside its support
RIGID_BODY = {"objectId": 1, "corner": (0.13, 0.035, 0.13), "side": 4, "density": 2200.0}


def rigid_points(spec, diameter):
    k = np.arange(spec["side"], dtype=np.float32) * np.float32(diameter)
    g = np.stack(np.meshgrid(k, k, k, indexing="ij"), -1).reshape(-1, 3)
    return (g + np.array(spec["corner"], np.float32)).astype(np.float32)


def rigid_pose_after_step1(com0):
    """The pose PyBullet would have written (bullet_solver.py:160-167) -- here a fixed, made-up one, written into the
    reference's rigid_body_* fields after the first step so that _renew_rigid_particle_state (base_solver.py:616) moves and
    spins the body's particles in the second."""
    axis = np.array([1.0, 2.0, 3.0]) / np.sqrt(14.0)
    a = 0.1
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    R = np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)
    return {"com": (com0.astype(np.float64) + np.array([0.004, 0.003, -0.002])).astype(np.float32), "rot": R.astype(np.float32),
            "vel": np.array([0.3, 0.1, -0.2], np.float32), "angvel": np.array([1.0, -2.0, 0.5], np.float32)}


def inject_rigid(container, spec):
    """What insert_object() does for a dynamic entry of cfg.get_rigid_bodies() (base_container.py:301-340), without the mesh."""
    obj = spec["objectId"]
    pts = rigid_points(spec, container.particle_diameter)
    n = pts.shape[0]
    zeros3 = np.zeros((n, 3), np.float32)
    container.object_id_rigid_body.add(obj)
    container.rigid_body_particle_num[obj] = n
    container.object_visibility[obj] = 1
    container.object_materials[obj] = container.material_rigid
    container.object_collection[obj] = {"particleNum": n}
    container.add_particles(obj, n, pts, zeros3, spec["density"] * np.ones(n, np.float32), np.zeros(n, np.float32),
                            np.array([container.material_rigid] * n, dtype=np.int32), np.ones(n, dtype=np.int32),
                            np.zeros((n, 3), np.int32))
    container.rigid_body_is_dynamic[obj] = 1
    container.rigid_body_velocities[obj] = np.zeros(3, np.float32)
    container.rigid_body_masses[obj] = container.compute_rigid_body_mass(obj)
    com = pts.astype(np.float64).mean(0).astype(np.float32)   # bullet_solver.py:12: "center of mass = base position"
    container.rigid_body_original_centers_of_mass[obj] = com
    container.rigid_body_centers_of_mass[obj] = com
    container.rigid_body_rotations[obj] = np.eye(3, dtype=np.float32)
    container.present_object.append(obj)
    return n, com


SCENES = {
    # name: (scene dict, jitter amplitude, seed, checkpoints (steps))
    "wcsph_cube": (dam_break_scene(end=(0.16, 0.16, 0.16)), 0.0, 0, [1, 2, 5, 10]),
    "wcsph_jitter": (dam_break_scene(end=(0.14, 0.16, 0.12), velocity=(0.3, -0.5, 0.1)), 0.004, 11, [1, 3, 6]),
    "wcsph_box": (dam_break_scene(domain_end=(0.32, 0.32, 0.32), start=(0.06, 0.06, 0.06), end=(0.14, 0.16, 0.14),
                                  translation=(0.0, 0.0, 0.0), add_domain_box=True, viscosity_b=0.3), 0.003, 5, [1, 3, 5]),
    "dfsph_cube": (dam_break_scene(method="dfsph", end=(0.14, 0.14, 0.14), dt=6e-4, velocity=(0.0, -0.5, 0.0)), 0.003, 2, [1, 2, 4]),
    "pcisph_cube": (dam_break_scene(method="pcisph", end=(0.14, 0.14, 0.14), dt=4e-4), 0.003, 3, [1, 2, 3]),
    "dfsph_implicit": (dam_break_scene(method="dfsph", end=(0.12, 0.14, 0.12), dt=6e-4, viscosity=50.0,
                                       viscosity_method="implicit", velocity=(0.2, -0.5, 0.0)), 0.003, 4, [1, 2]),
    # lattices packed tighter than the rest spacing: rho > rho0 from step 0, so the pressure terms are live
    "wcsph_compressed": (dam_break_scene(end=(0.126, 0.126, 0.126), particleSpacing=0.018, velocity=(0.1, -0.3, 0.0)), 0.002, 21, [1, 2, 4, 8]),
    "pcisph_compressed": (dam_break_scene(method="pcisph", end=(0.1, 0.116, 0.1), particleSpacing=0.0165), 0.0015, 22, [1, 2, 3]),
    "dfsph_compressed": (dam_break_scene(method="dfsph", end=(0.108, 0.126, 0.108), particleSpacing=0.0185, dt=6e-4), 0.002, 23, [1, 2, 3]),
    # emitter hack (base_solver.py:18-23, :660-677): fluid above gravitationUpper is frozen as "rigid" and released
    # when it crosses the threshold
    "wcsph_emitter": (dam_break_scene(end=(0.12, 0.24, 0.12), translation=(0.1, 0.2, 0.1), velocity=(0.0, -2.5, 0.0),
                                      gravitationUpper=0.33), 0.0, 0, [1, 10, 30]),
    "dfsph_emitter": (dam_break_scene(method="dfsph", end=(0.12, 0.24, 0.12), translation=(0.1, 0.2, 0.1), dt=6e-4,
                                      velocity=(0.0, -2.5, 0.0), gravitationUpper=0.33), 0.0, 0, [1, 10, 20]),
    "dfsph_box": (dam_break_scene(method="dfsph", domain_end=(0.32, 0.32, 0.32), start=(0.06, 0.06, 0.06),
                                  end=(0.14, 0.16, 0.14), translation=(0.0, 0.0, 0.0), add_domain_box=True, dt=6e-4,
                                  viscosity_b=0.3),
                  0.003, 6, [1, 2, 3]),
    # implicit viscosity with boundary particles: the rigid-neighbour terms of D_ii and b (base_solver.py:326-349) --
    # the branch BASELINE config 5 (buckling sheet inside a sampled domain box) runs
    "dfsph_implicit_box": (dam_break_scene(method="dfsph", domain_end=(0.32, 0.32, 0.32), start=(0.06, 0.06, 0.06),
                                           end=(0.14, 0.14, 0.14), translation=(0.0, 0.0, 0.0), add_domain_box=True, dt=6e-4,
                                           viscosity=50.0, viscosity_b=20.0, viscosity_method="implicit",
                                           velocity=(0.2, -0.5, 0.1)), 0.003, 31, [1, 2]),
    # implicit viscosity + emitter hack (gravitationUpper): frozen "rigid" fluid above the threshold enters D_ii / b as a
    # boundary and is released step by step (base_solver.py:660-677)
    "dfsph_implicit_emitter": (dam_break_scene(method="dfsph", end=(0.1, 0.2, 0.1), translation=(0.1, 0.2, 0.1), dt=6e-4,
                                               viscosity=50.0, viscosity_method="implicit", velocity=(0.0, -2.5, 0.0),
                                               gravitationUpper=0.31), 0.0, 0, [1, 6, 12]),
    # implicit viscosity under WCSPH: the solve runs BEFORE compute_pressure clamps particle_densities (WCSPH.py:29-33), i.e. on
    # the unclamped densities -- a free-surface block, so that rho < rho0 on most particles
    "wcsph_implicit": (dam_break_scene(end=(0.12, 0.14, 0.12), viscosity=50.0, viscosity_method="implicit",
                                       velocity=(0.2, -0.5, 0.1)), 0.003, 41, [1, 2, 3]),
    # PCISPH next to boundary particles: rho* uses the CURRENT position of a rigid neighbour and the predicted one of a fluid
    # neighbour (PCISPH.py:33-63), the pressure acceleration has its own rigid branch (:85-107)
    "pcisph_box": (dam_break_scene(method="pcisph", domain_end=(0.32, 0.32, 0.32), start=(0.075, 0.075, 0.075),
                                   end=(0.17, 0.19, 0.17), translation=(0.0, 0.0, 0.0), add_domain_box=True,
                                   particleSpacing=0.017, viscosity_b=0.3, velocity=(0.1, -0.4, 0.0)), 0.002, 51, [1, 2, 3]),
    "pcisph_emitter": (dam_break_scene(method="pcisph", end=(0.1, 0.2, 0.1), translation=(0.1, 0.2, 0.1),
                                       velocity=(0.0, -2.5, 0.0), gravitationUpper=0.31), 0.0, 0, [1, 6, 12]),
    "pcisph_implicit": (dam_break_scene(method="pcisph", end=(0.1, 0.12, 0.1), particleSpacing=0.0175, viscosity=50.0,
                                        viscosity_method="implicit", velocity=(0.2, -0.5, 0.1)), 0.002, 52, [1, 2, 3]),
    "wcsph_implicit_box": (dam_break_scene(domain_end=(0.32, 0.32, 0.32), start=(0.06, 0.06, 0.06), end=(0.14, 0.14, 0.14),
                                           translation=(0.0, 0.0, 0.0), add_domain_box=True, viscosity=50.0, viscosity_b=20.0,
                                           viscosity_method="implicit", velocity=(0.2, -0.5, 0.1)), 0.003, 53, [1, 2]),
    # dynamic rigid body: force / torque accumulators of every solver (see rigid_scene)
    "rigid_wcsph": (rigid_scene("wcsph", 4e-4), 0.002, 61, [1, 2, 4]),
    "rigid_dfsph": (rigid_scene("dfsph", 6e-4), 0.002, 62, [1, 2, 3]),
    "rigid_pcisph": (rigid_scene("pcisph", 4e-4), 0.002, 63, [1, 2, 3]),
    # late entry (base_container.py:218-221): a second block whose entryTime falls into the 4th step; inserted by
    # _step() itself (WCSPH.py:41, DFSPH.py:307, PCISPH.py:181)
    "wcsph_late": (late_scene("wcsph", 4e-4), 0.0, 0, [2, 4, 6]),
    "dfsph_late": (late_scene("dfsph", 6e-4), 0.0, 0, [2, 4, 6]),
    "pcisph_late": (late_scene("pcisph", 4e-4), 0.0, 0, [2, 4, 5]),
}

# Round 3: fixtures at BASELINE configs[0] size (SURVEY 8c asked for 512-8000 particles x 5-100 steps).  Written to
# tests/golden/big/ with LEAN snapshots (ids, positions, velocities, densities, pressures, materials) plus the whole
# iteration history the reference's python loops print, one entry per step.  Hours of interpreter time: run them one
# per process (`python oracle/gen_golden.py c1_wcsph &` ...).
BIG_SCENES = {
    # C1 exactly: domain [1,1,1], block [0,0.4]^3 translated by 0.1 -> 20^3 = 8000, WCSPH, dt 4e-4, mu 10, no jitter
    "c1_wcsph": (dam_break_scene(), 0.0, 0, [1, 5, 10, 20, 40]),
    # the same block, perturbed and moving, so that lattice symmetries do not hide errors
    "c1_wcsph_jitter": (dam_break_scene(velocity=(0.2, -0.5, 0.1)), 0.003, 71, [1, 5, 10, 20]),
    # 16^3 = 4096 particles, solver loops with the reference's own stop tests
    "dfsph_4k": (dam_break_scene(method="dfsph", end=(0.31, 0.31, 0.31), dt=6e-4, velocity=(0.1, -0.5, 0.0)), 0.003, 72, [1, 2, 5, 10]),
    "pcisph_4k": (dam_break_scene(method="pcisph", end=(0.31, 0.31, 0.31), dt=4e-4, velocity=(0.1, -0.5, 0.0)), 0.003, 73, [1, 2, 5, 10]),
    # the same 16^3 block packed tighter than the rest spacing (rho > rho0 from step 0): the solver loops iterate
    "dfsph_4k_compressed": (dam_break_scene(method="dfsph", end=(0.287, 0.287, 0.287), particleSpacing=0.0185, dt=6e-4,
                                            velocity=(0.1, -0.5, 0.0)), 0.002, 74, [1, 2, 4, 6]),
    "pcisph_4k_compressed": (dam_break_scene(method="pcisph", end=(0.256, 0.256, 0.256), particleSpacing=0.0165, dt=4e-4,
                                             velocity=(0.1, -0.5, 0.0)), 0.0015, 75, [1, 2, 4, 6]),
    # boundary particles at size: the 16^3 block inside a sampled domain box, falling onto its floor (the rigid-aware instantiations of
    # every pass: fluid-rigid density / viscosity / pressure terms, base_solver.py:136-178, :232-262)
    "wcsph_box_4k": (dam_break_scene(domain_end=(0.5, 0.5, 0.5), start=(0.06, 0.045, 0.06), end=(0.37, 0.355, 0.37),
                                     translation=(0.0, 0.0, 0.0), add_domain_box=True, viscosity_b=0.3, velocity=(0.1, -1.0, 0.05)),
                     0.003, 77, [1, 5, 10, 20]),
    # the same impact under DFSPH (rigid terms of the factor alpha and of the density derivative: DFSPH.py:60-110), solver loops with their
    # own stop tests.  (The PCISPH version of this scene did not finish its first step in 28 minutes of interpreter time: not kept.)
    "dfsph_box_4k": (dam_break_scene(method="dfsph", domain_end=(0.5, 0.5, 0.5), start=(0.06, 0.045, 0.06), end=(0.37, 0.355, 0.37),
                                     translation=(0.0, 0.0, 0.0), add_domain_box=True, dt=6e-4, viscosity_b=0.3, velocity=(0.1, -1.0, 0.05)),
                     0.003, 78, [1, 2]),   # two steps (6 + 4 divergence, 19 + 8 density iterations): from the third on a hard DFSPH impact is
                                           # chaotic -- this repo's C oracle and the interpreter, identical in every iteration count of ten steps,
                                           # are 2e-5 apart in position after 5 steps and 1e-2 (a handful of particles at the floor) after 10
    # round 4: a SOFT impact under DFSPH that stays well-conditioned for ten steps (a 2e-7 perturbation of the initial lattice grows to 4e-6 in
    # the C oracle; the hard impact above reaches 1e-2).  What made the scenes above violent is not their speed but their start: a block that
    # begins at 0.06 overlaps the inner wall layer of the sampled box (layers at padding = 0.04 and 0.06, base_container.py:62-64, :832).  Here
    # the block starts one spacing clear of the walls, packed tighter than the rest spacing so that both loops iterate (2 + 16, 1 + 5, ... ).
    "dfsph_box_soft_4k": (dam_break_scene(method="dfsph", domain_end=(0.5, 0.5, 0.5), start=(0.1, 0.079, 0.1), end=(0.387, 0.366, 0.387),
                                          particleSpacing=0.018, translation=(0.0, 0.0, 0.0), add_domain_box=True, dt=6e-4, viscosity_b=0.3,
                                          velocity=(0.05, -1.0, 0.02)), 0.002, 82, [1, 2, 5, 10]),
    # round 4: PCISPH next to boundary particles at a size the interpreter finishes (1,584 fluid + 2,763 box particles, 11 / 3 / 3 / 2 / 2
    # iterations): the rigid branches of rho* and of the pressure acceleration (PCISPH.py:33-63, :85-107) with the loop iterating
    "pcisph_box_1k5": (dam_break_scene(method="pcisph", domain_end=(0.4, 0.4, 0.4), start=(0.1, 0.076, 0.1), end=(0.303, 0.262, 0.303),
                                       particleSpacing=0.017, translation=(0.0, 0.0, 0.0), add_domain_box=True, dt=4e-4, viscosity_b=0.3,
                                       velocity=(0.1, -1.0, 0.05)), 0.0015, 81, [1, 2, 3, 5]),
    # the path of configs[4] (DFSPH + implicit viscosity: matrix-free CG, base_solver.py:509) at the same 16^3 size, CG history kept
    "visc_4k": (dam_break_scene(method="dfsph", end=(0.31, 0.31, 0.31), dt=6e-4, viscosity=50.0, viscosity_method="implicit",
                                velocity=(0.1, -0.5, 0.0)), 0.003, 76, [1, 2, 5, 10]),
}
LEAN_KEYS = ("ids", "positions", "velocities", "densities", "pressures", "materials", "iter_v", "iter_d", "iter_pci", "iter_cg")
ITER_PATTERNS = (("iter_v", r"DFSPH - iteration V: (\d+)"), ("iter_d", r"DFSPH - iterations: (\d+)"),
                 ("iter_pci", r"PCISPH - iteration: (\d+)"), ("iter_cg", r"CG iteration:\s+(\d+)"))


def _np(field):
    return field.to_numpy()


def snapshot(container, solver, method, log):
    d = {
        "ids": _np(container.particle_colors)[:, 0].copy(),
        "positions": _np(container.particle_positions), "velocities": _np(container.particle_velocities),
        "accelerations": _np(container.particle_accelerations), "densities": _np(container.particle_densities),
        "pressures": _np(container.particle_pressures), "rest_volumes": _np(container.particle_rest_volumes),
        "masses": _np(container.particle_masses), "materials": _np(container.particle_materials),
        "object_ids": _np(container.particle_object_ids), "grid_ids": _np(container.grid_ids),
    }
    if method == "dfsph":
        d.update(alphas=_np(container.particle_dfsph_alphas), kappa=_np(container.particle_dfsph_kappa),
                 kappa_v=_np(container.particle_dfsph_kappa_v), densities_star=_np(container.particle_densities_star),
                 densities_derivatives=_np(container.particle_densities_derivatives))
    if method == "pcisph":
        d.update(densities_star=_np(container.particle_densities_star),
                 pressure_accelerations=_np(container.particle_pressure_accelerations),
                 pcisph_k=np.float32(container.pcisph_k[None]), density_error=np.float32(container.density_error[None]))
    if hasattr(solver, "cg_x"):
        d.update(cg_x=_np(solver.cg_x))
    wrench = {"rigid_forces": _np(container.rigid_body_forces).copy(), "rigid_torques": _np(container.rigid_body_torques).copy(),
              "rigid_masses": _np(container.rigid_body_masses).copy()}
    n = container.particle_num[None]
    out = {k: (v[:n] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] >= n else v) for k, v in d.items()}
    out.update(wrench)
    # iteration counts printed by the reference's python loops (DFSPH.py:159,:243; PCISPH.py:125; base_solver.py:461)
    text = log.getvalue()
    for key, pat in (("iter_v", r"DFSPH - iteration V: (\d+)"), ("iter_d", r"DFSPH - iterations: (\d+)"),
                     ("iter_pci", r"PCISPH - iteration: (\d+)"), ("iter_cg", r"CG iteration:\s+(\d+)")):
        m = re.findall(pat, text)
        out[key] = np.int32(int(m[-1]) if m else -1)
    return out


def run_scene(name):
    big = name in BIG_SCENES
    cfg, jitter, seed, checkpoints = (BIG_SCENES if big else SCENES)[name]
    method = cfg["Configuration"]["simulationMethod"]
    tmp = tempfile.NamedTemporaryFile("w", suffix=".json", delete=False)
    json.dump(cfg, tmp)
    tmp.close()
    log = io.StringIO()
    t0 = time.time()
    with contextlib.redirect_stdout(log):
        from SPH.utils import SimConfig
        from SPH.containers import DFSPHContainer, PCISPHContainer, WCSPHContainer
        from SPH.fluid_solvers import DFSPHSolver, PCISPHSolver, WCSPHSolver
        ccls, scls = {"wcsph": (WCSPHContainer, WCSPHSolver), "dfsph": (DFSPHContainer, DFSPHSolver),
                      "pcisph": (PCISPHContainer, PCISPHSolver)}[method]
        container = ccls(SimConfig(scene_file_path=tmp.name))
        solver = scls(container)
        # ---- solver.prepare() (base_solver.py:683-690), spelled out so ids / jitter can be set after insertion
        solver.init_object_id()
        container.insert_object()
        inject = None
        if name.startswith("rigid_"):
            inject = inject_rigid(container, RIGID_BODY)
        n = container.particle_num[None]
        colors = container.particle_colors._data
        colors[:n, 0] = np.arange(n)
        colors[:n, 1:] = 0
        if jitter > 0:
            rng = np.random.default_rng(seed)
            pos = container.particle_positions._data
            mat = container.particle_materials._data[:n]
            fl = np.nonzero(mat == 1)[0]
            pos[fl] = (pos[fl] + rng.uniform(-jitter, jitter, (len(fl), 3))).astype(np.float32)
        init = {"positions": container.particle_positions._data[:n].copy(),
                "velocities": container.particle_velocities._data[:n].copy(),
                "densities": container.particle_densities._data[:n].copy(),
                "materials": container.particle_materials._data[:n].copy(),
                "object_ids": container.particle_object_ids._data[:n].copy(),
                "is_dynamic": container.particle_is_dynamic._data[:n].copy()}
        solver.prepare_emitter()
        solver.rigid_solver.insert_rigid_object()
        solver.renew_rigid_particle_state()
        container.prepare_neighborhood_search()
        solver.compute_rigid_particle_volume()
        if method == "dfsph":  # DFSPH.py:321
            solver.compute_density()
            solver.compute_alpha()
        if method == "pcisph":  # PCISPH.py:188
            solver.compute_pcisph_k()
    out = {"scene_json": np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8), "jitter": np.float64(jitter),
           "seed": np.int64(seed), "checkpoints": np.array(checkpoints),
           # what BaseContainer.__init__ derived (pins sph_project_amd/scene.py)
           "geo_dx": np.float64(container.dx), "geo_dh": np.float64(container.dh), "geo_V0": np.float64(container.V0),
           "geo_grid_num": np.array(container.grid_num), "geo_padding": np.float64(container.padding),
           "geo_particle_max_num": np.int64(container.particle_max_num)}
    if inject:
        out["inject_count"], out["inject_com"] = np.int64(inject[0]), inject[1]
    for k, v in init.items():
        out["init_" + k] = v
    for k, v in snapshot(container, solver, method, log).items():
        if not big or k in LEAN_KEYS:
            out["prep_" + k] = v
    step = 0
    for cp in checkpoints:
        while step < cp:
            with contextlib.redirect_stdout(log):
                solver.step()
            step += 1
            if inject and step == 1:
                pose = rigid_pose_after_step1(inject[1])
                obj = RIGID_BODY["objectId"]
                container.rigid_body_centers_of_mass[obj] = pose["com"]
                container.rigid_body_rotations[obj] = pose["rot"]
                container.rigid_body_velocities[obj] = pose["vel"]
                container.rigid_body_angular_velocities[obj] = pose["angvel"]
                out["pose_step"] = np.int64(1)
                for k, v in pose.items():
                    out["pose_" + k] = v
            # late entrants: still carrying the sentinel colour; their persistent id = insertion index, i.e. the count so
            # far + their rank in lattice order (x slowest, z fastest: base_container.py:769-777), whatever a sort did since
            n_now = container.particle_num[None]
            colors = container.particle_colors._data
            late = np.nonzero((colors[:n_now, 0] == LATE_COLOR[0]) & (colors[:n_now, 1] == LATE_COLOR[1]))[0]
            if len(late):
                pos = container.particle_positions._data[late]
                order = np.lexsort((pos[:, 2], pos[:, 1], pos[:, 0]))
                first = n_now - len(late)
                colors[late[order], 0] = first + np.arange(len(late))
                colors[late, 1:] = 0
                out[f"late_step"] = np.int64(step)   # the step during which they were inserted (1-based)
                out[f"late_first_id"] = np.int64(first)
        for k, v in snapshot(container, solver, method, log).items():
            if not big or k in LEAN_KEYS:
                out[f"s{cp}_" + k] = v
        print(f"  {name}: step {step} done ({time.time() - t0:.0f} s)", flush=True)
    outdir = os.path.join(OUT, "big") if big else OUT
    if big:
        # one entry per step (prepare() prints nothing): the loops' iteration counts over the whole run
        for key, pat in ITER_PATTERNS:
            m = re.findall(pat, log.getvalue())
            if m:
                out["hist_" + key] = np.array([int(v) for v in m], np.int32)
    os.makedirs(outdir, exist_ok=True)
    np.savez_compressed(os.path.join(outdir, name + ".npz"), **out)
    os.unlink(tmp.name)
    print(f"{name}: n={n} written ({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    names = sys.argv[1:] or list(SCENES)   # the BIG_SCENES only by name
    for nm in names:
        run_scene(nm)
