"""Scene dicts and builders of the product path (no oracle, no test code): what bench.py, __graft_entry__.smoke()
and the tests use to go from a JSON-like scene in the reference's format to a container + solver pair driven exactly
like the reference's run_simulation.py drives its own (XContainer(config) -> XSolver(container) -> prepare() -> step()).
"""
from __future__ import annotations

import copy

import numpy as np

from . import scene
from .SPH.utils import SimConfig


def dam_break_scene(method="wcsph", domain_end=(1.0, 1.0, 1.0), start=(0.0, 0.0, 0.0), end=(0.4, 0.4, 0.4),
                    translation=(0.1, 0.1, 0.1), velocity=(0.0, 0.0, 0.0), dt=4e-4, viscosity=10.0,
                    add_domain_box=False, viscosity_method="standard", radius=0.01, **extra):
    """SURVEY 8(d) config C1 by default: 20^3 = 8000-particle cube, WCSPH."""
    cfg = {
        "Configuration": {
            "domainStart": [0.0, 0.0, 0.0], "domainEnd": list(domain_end), "addDomainBox": add_domain_box,
            "particleRadius": radius, "density0": 1000, "simulationMethod": method,
            "viscosityMethod": viscosity_method, "gravitation": [0.0, -9.81, 0.0], "timeStepSize": dt,
            "viscosity": viscosity,
        },
        "FluidBlocks": [{
            "objectId": 0, "start": list(start), "end": list(end), "translation": list(translation),
            "scale": [1, 1, 1], "velocity": list(velocity), "density": 1000.0, "color": [50, 100, 200],
            "entryTime": -1.0,
        }],
    }
    cfg["Configuration"].update(extra)
    return cfg


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


def c5_scene(domain_end=(4.0, 20.0, 8.0), start=(1.12, 1.0, 1.0), end=(1.88, 12.2, 1.08), g_upper=2.5,
             velocity=(0.0, -2.2, 0.75)):
    """the reference's buckling scene data/scenes/final_scene3.json (DFSPH + implicit viscosity,
    mu = mu_b = 1800, dt 1e-3, emitter above y = 2.5, a 38 x 560 x 5 fluid sheet inside a 4 x 20 x 8 domain box of
    ~2.07 M static boundary particles, G = 10 M cells) without its mesh rigid body (cookie_bar_small.obj needs trimesh).
    The arguments shrink the box and the sheet for tests; the defaults are the reference's numbers."""
    return {
        "Configuration": {
            "domainStart": [0.0, 0.0, 0.0], "domainEnd": list(domain_end), "addDomainBox": True,
            "particleRadius": 0.01, "density0": 1000, "simulationMethod": "dfsph", "viscosityMethod": "implicit",
            "gravitation": [0.0, -9.81, 0.0], "gravitationUpper": g_upper, "timeStepSize": 0.001,
            "viscosity": 1800.0, "viscosity_b": 1800.0,
        },
        "FluidBlocks": [{
            "objectId": 0, "start": list(start), "end": list(end), "translation": [0.0, 0.0, 0.0],
            "scale": [1, 1, 1], "velocity": list(velocity), "density": 1000.0, "color": [50, 100, 200],
            "entryTime": -1.0,
        }],
    }


def scene_particles(cfg_dict):
    """Host lattice of every object present at prepare(), in the reference's insertion order
    (domain box first: base_container.py:192, then FluidBlocks: :215).  Blocks with entryTime > 0 are listed with
    their entry time (`entry_time`); callers that cannot insert late skip or reject them."""
    cfg = SimConfig(config=copy.deepcopy(cfg_dict))
    geo = scene.derive_geometry(cfg)
    batches = []
    blocks = cfg.get_fluid_blocks()
    if geo.add_domain_box:
        pos = scene.box_lattice(geo.domain_box_start, geo.domain_box_size, geo.domain_box_thickness, geo.particle_spacing)
        n = pos.shape[0]
        # BaseSolver.prepare() -> init_object_id() (base_solver.py:680) runs after the box was added in
        # BaseContainer.__init__, so the reference's box particles carry object id -1
        batches.append(dict(object_id=-1, pos=pos, vel=np.zeros((n, 3), np.float32),
                            density=np.full(n, 1000.0, np.float32), material=np.full(n, 2, np.int32),
                            is_dynamic=np.zeros(n, np.int32), entry_time=-1.0))
    for blk in blocks:
        off = np.array(blk["translation"])
        s, e = np.array(blk["start"]) + off, np.array(blk["end"]) + off
        pos = scene.cube_lattice(s, (e - s) * np.array(blk["scale"]), geo.particle_spacing)
        n = pos.shape[0]
        batches.append(dict(object_id=blk["objectId"], pos=pos, vel=np.tile(np.asarray(blk["velocity"], np.float32), (n, 1)),
                            density=np.full(n, blk["density"], np.float32), material=np.full(n, 1, np.int32),
                            is_dynamic=np.ones(n, np.int32), entry_time=float(blk.get("entryTime", -1.0))))
    return cfg, geo, batches


def build_product(cfg_dict, **engine_opts):
    """Containers + solver of the product path for a scene dict (what run_simulation.py:46-63 does for a scene file)."""
    from .SPH import containers, fluid_solvers
    cfg = SimConfig(config=copy.deepcopy(cfg_dict))
    method = cfg.get_cfg("simulationMethod")
    ccls = {"wcsph": containers.WCSPHContainer, "dfsph": containers.DFSPHContainer, "pcisph": containers.PCISPHContainer}[method]
    scls = {"wcsph": fluid_solvers.WCSPHSolver, "dfsph": fluid_solvers.DFSPHSolver, "pcisph": fluid_solvers.PCISPHSolver}[method]
    container = ccls(cfg, GGUI=False, **engine_opts)
    solver = scls(container)
    return container, solver
