"""ctypes binding of libsph_hip.so (C-ABI declared in include/sph_hip.h).

The product path has no CPU fallback: if the HIP library is missing or no gfx950 device is
visible, loading / creating a handle raises.  Nothing in here imports the oracle.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPH_HIP_LIB", os.path.join(_HERE, "libsph_hip.so"))  # override: A/B builds only

MAX_OBJECTS = 20
MAT_FLUID, MAT_RIGID = 1, 2
METHOD = {"wcsph": 0, "dfsph": 1, "pcisph": 2}

# enum SphField
(F_POSITION, F_VELOCITY, F_ACCELERATION, F_DENSITY, F_PRESSURE, F_REST_VOLUME, F_MASS, F_MATERIAL,
 F_OBJECT_ID, F_IS_DYNAMIC, F_COLOR, F_PARTICLE_ID, F_GRID_ID, F_DFSPH_ALPHA, F_DFSPH_KAPPA,
 F_DFSPH_KAPPA_V, F_DENSITY_STAR, F_DENSITY_DERIV, F_PRESSURE_ACCEL, F_PREDICTED_VEL, F_PREDICTED_POS,
 F_CG_X, F_ORIG_POSITION, F_GHOST, F_DFSPH_KAPPA_NEXT, F_DFSPH_KAPPA_V_NEXT, F_DEBUG_CAPTURE) = range(27)

_FIELD_SPEC = {  # field -> (dtype, components)
    F_POSITION: (np.float32, 3), F_VELOCITY: (np.float32, 3), F_ACCELERATION: (np.float32, 3),
    F_DENSITY: (np.float32, 1), F_PRESSURE: (np.float32, 1), F_REST_VOLUME: (np.float32, 1),
    F_MASS: (np.float32, 1), F_MATERIAL: (np.int32, 1), F_OBJECT_ID: (np.int32, 1),
    F_IS_DYNAMIC: (np.int32, 1), F_COLOR: (np.int32, 3), F_PARTICLE_ID: (np.int32, 1),
    F_GRID_ID: (np.int32, 1), F_DFSPH_ALPHA: (np.float32, 1), F_DFSPH_KAPPA: (np.float32, 1),
    F_DFSPH_KAPPA_V: (np.float32, 1), F_DENSITY_STAR: (np.float32, 1), F_DENSITY_DERIV: (np.float32, 1),
    F_PRESSURE_ACCEL: (np.float32, 3), F_PREDICTED_VEL: (np.float32, 3), F_PREDICTED_POS: (np.float32, 3),
    F_CG_X: (np.float32, 3), F_ORIG_POSITION: (np.float32, 3), F_GHOST: (np.int32, 1),
    F_DFSPH_KAPPA_NEXT: (np.float32, 1), F_DFSPH_KAPPA_V_NEXT: (np.float32, 1), F_DEBUG_CAPTURE: (np.float32, 1),
}

# enum SphPhase
(PH_NEIGHBOR_SEARCH, PH_RIGID_VOLUME, PH_DENSITY, PH_NON_PRESSURE, PH_PRESSURE_INTEGRATE, PH_DFSPH_ALPHA,
 PH_DFSPH_DIVERGENCE, PH_DFSPH_DENSITY) = range(8)

# enum SphKernelId
KERNEL_IDS = ["hash_count", "scan", "scatter", "density", "non_pressure", "pressure_integrate",
              "rigid_volume", "dfsph_density_alpha", "dfsph_rho_adv", "dfsph_correct", "reduce",
              "pcisph_rho_star", "pcisph_pressure_accel", "cg_prepare", "cg_ap", "cg_vector", "misc", "halo"]


class SphParams(C.Structure):
    _fields_ = [
        ("domain_size", C.c_double * 3), ("particle_radius", C.c_double), ("support_radius", C.c_double),
        ("V0", C.c_double), ("padding", C.c_double), ("grid_num", C.c_int32 * 3),
        ("gravity", C.c_double * 3), ("g_upper", C.c_double), ("viscosity", C.c_double),
        ("viscosity_b", C.c_double), ("density_0", C.c_double), ("surface_tension", C.c_double),
        ("dt", C.c_double), ("particle_max_num", C.c_int32), ("viscosity_implicit", C.c_int32),
        ("method", C.c_int32), ("fixed_iterations", C.c_int32), ("fast_math", C.c_int32),
        ("device", C.c_int32), ("force_global", C.c_int32), ("deterministic", C.c_int32),
    ]


class SphStats(C.Structure):
    _fields_ = [
        ("steps", C.c_int64), ("pair_interactions", C.c_int64), ("particle_num", C.c_int32),
        ("fluid_particle_num", C.c_int32), ("iter_divergence", C.c_int32), ("iter_density", C.c_int32),
        ("iter_pcisph", C.c_int32), ("iter_cg", C.c_int32), ("err_divergence", C.c_float),
        ("err_density", C.c_float), ("err_pcisph", C.c_float), ("err_cg", C.c_float),
        ("lds_fallback_blocks", C.c_int64), ("total_time", C.c_double), ("pair_evaluations", C.c_int64),
        ("hash_launches", C.c_int64), ("prehashed_sorts", C.c_int64), ("list_sorts", C.c_int64),
    ]


class SphError(RuntimeError):
    pass


_lib = None

# every symbol include/sph_hip.h declares: (name, restype, argtypes)
_VP = C.c_void_p
_SIGNATURES = [
    ("sph_create", C.c_int, [C.POINTER(SphParams), C.POINTER(_VP)]),
    ("sph_destroy", None, [_VP]),
    ("sph_last_error", C.c_char_p, [_VP]),
    ("sph_append_particles", C.c_int, [_VP, C.c_int, C.c_int] + [_VP] * 7),
    ("sph_set_appended_ids", C.c_int, [_VP, C.c_int, _VP]),
    ("sph_set_object", C.c_int, [_VP, C.c_int, C.c_int, C.c_int]),
    ("sph_set_rigid_pose", C.c_int, [_VP, C.c_int] + [_VP] * 5),
    ("sph_get_rigid_wrench", C.c_int, [_VP, _VP, _VP, C.c_int]),
    ("sph_prepare", C.c_int, [_VP]),
    ("sph_step", C.c_int, [_VP, C.c_int]),
    ("sph_step_async", C.c_int, [_VP, C.c_int]),
    ("sph_synchronize", C.c_int, [_VP]),
    ("sph_step_begin", C.c_int, [_VP]),
    ("sph_step_end", C.c_int, [_VP]),
    ("sph_run_phase", C.c_int, [_VP, C.c_int]),
    ("sph_download", C.c_int, [_VP, C.c_int, _VP, C.c_size_t]),
    ("sph_upload", C.c_int, [_VP, C.c_int, _VP, C.c_size_t]),
    ("sph_particle_num", C.c_int, [_VP]),
    ("sph_fluid_particle_num", C.c_int, [_VP]),
    ("sph_get_stats", C.c_int, [_VP, C.POINTER(SphStats)]),
    ("sph_profile_enable", C.c_int, [_VP, C.c_int, C.c_int]),
    ("sph_profile_reset", C.c_int, [_VP]),
    ("sph_profile_read", C.c_int, [_VP, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    ("sph_kernel_name", C.c_char_p, [C.c_int]),
    ("sph_device_info", C.c_int, [_VP, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    ("sph_measure_copy_rate", C.c_int, [_VP, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    ("sph_comm_unique_id", C.c_int, [_VP]),
    ("sph_comm_init", C.c_int, [_VP, C.c_int, C.c_int, _VP]),
    ("sph_comm_transport", C.c_char_p, [_VP]),
    ("sph_comm_set_slab", C.c_int, [_VP, C.c_int, C.c_int]),
    ("sph_comm_get_slab", C.c_int, [_VP] + [C.POINTER(C.c_int)] * 4),
    ("sph_comm_set_rebalance", C.c_int, [_VP, C.c_int]),
    ("sph_comm_add_global_count", C.c_int, [_VP, C.c_int, C.c_int]),
    ("sph_device_count", C.c_int, []),
    ("sph_voxelize_mesh", C.c_int, [_VP, C.c_int, _VP, C.c_int, C.c_double, _VP, C.c_int64, C.POINTER(C.c_int64)]),
    ("sph_points_in_mesh", C.c_int, [_VP, C.c_int, _VP, C.c_int] + [_VP, C.c_int] * 3 + [_VP]),
    ("sph_write_ply_ascii", C.c_int, [C.c_char_p, _VP, C.c_int64]),
    ("sph_format_f32", C.c_int, [C.c_float, C.c_char_p]),
    ("sph_comm_allreduce", C.c_int, [_VP, C.POINTER(C.c_double), C.c_int, C.c_int]),
    ("sph_comm_barrier", C.c_int, [_VP]),
    ("sph_comm_selftest", C.c_int, [_VP, C.c_int]),
]
EXPORTED_SYMBOLS = [s[0] for s in _SIGNATURES]


def load():
    """Load libsph_hip.so; raises SphError if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SphError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(or `make -C sph_project_amd/csrc`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, res, args in _SIGNATURES:
        fn = getattr(lib, name)  # AttributeError => header / library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Engine:
    """Owns one SphHandle.  All arrays crossing the boundary are C-contiguous numpy f32 / i32."""

    def __init__(self, params: SphParams):
        self.lib = load()
        self.params = params
        h = C.c_void_p()
        rc = self.lib.sph_create(C.byref(params), C.byref(h))
        if rc != 0:
            msg = self.lib.sph_last_error(None)
            raise SphError(f"sph_create failed ({rc}): {msg.decode() if msg else ''}")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.sph_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            msg = self.lib.sph_last_error(self.h)
            raise SphError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    # -- scene upload
    def append_particles(self, object_id, pos, vel, density, pressure, material, is_dynamic, color):
        pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
        n = pos.shape[0]
        vel = np.ascontiguousarray(vel, dtype=np.float32).reshape(n, 3)
        density = np.ascontiguousarray(density, dtype=np.float32).reshape(n)
        pressure = np.ascontiguousarray(pressure, dtype=np.float32).reshape(n)
        material = np.ascontiguousarray(material, dtype=np.int32).reshape(n)
        is_dynamic = np.ascontiguousarray(is_dynamic, dtype=np.int32).reshape(n)
        color = np.ascontiguousarray(color, dtype=np.int32).reshape(n, 3)
        self._chk(self.lib.sph_append_particles(self.h, int(object_id), n, _ptr(pos), _ptr(vel), _ptr(density),
                                                _ptr(pressure), _ptr(material), _ptr(is_dynamic), _ptr(color)),
                  "sph_append_particles")

    def set_appended_ids(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int32).reshape(-1)
        self._chk(self.lib.sph_set_appended_ids(self.h, ids.shape[0], _ptr(ids)), "sph_set_appended_ids")

    def set_object(self, object_id, material, is_dynamic):
        self._chk(self.lib.sph_set_object(self.h, int(object_id), int(material), int(bool(is_dynamic))), "sph_set_object")

    def set_rigid_pose(self, object_id, com, rot, vel, angvel, com0=None):
        f = lambda a: np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
        com, rot, vel, angvel = f(com), f(rot), f(vel), f(angvel)
        com0 = None if com0 is None else f(com0)
        self._chk(self.lib.sph_set_rigid_pose(self.h, int(object_id), _ptr(com), _ptr(rot), _ptr(vel), _ptr(angvel),
                                              _ptr(com0)), "sph_set_rigid_pose")

    def get_rigid_wrench(self, reset=True):
        force = np.zeros((MAX_OBJECTS, 3), np.float32)
        torque = np.zeros((MAX_OBJECTS, 3), np.float32)
        self._chk(self.lib.sph_get_rigid_wrench(self.h, _ptr(force), _ptr(torque), int(reset)), "sph_get_rigid_wrench")
        return force, torque

    # -- stepping
    def prepare(self):
        self._chk(self.lib.sph_prepare(self.h), "sph_prepare")

    def step(self, nsteps=1):
        self._chk(self.lib.sph_step(self.h, int(nsteps)), "sph_step")

    def step_async(self, nsteps=1):
        self._chk(self.lib.sph_step_async(self.h, int(nsteps)), "sph_step_async")

    def synchronize(self):
        self._chk(self.lib.sph_synchronize(self.h), "sph_synchronize")

    def step_begin(self):
        self._chk(self.lib.sph_step_begin(self.h), "sph_step_begin")

    def step_end(self):
        self._chk(self.lib.sph_step_end(self.h), "sph_step_end")

    def run_phase(self, phase):
        self._chk(self.lib.sph_run_phase(self.h, int(phase)), "sph_run_phase")

    # -- state access
    @property
    def particle_num(self):
        return self.lib.sph_particle_num(self.h)

    @property
    def fluid_particle_num(self):
        return self.lib.sph_fluid_particle_num(self.h)

    def download(self, field):
        dtype, comp = _FIELD_SPEC[field]
        n = self.particle_num
        out = np.empty((n, comp) if comp > 1 else (n,), dtype=dtype)
        self._chk(self.lib.sph_download(self.h, int(field), _ptr(out), out.nbytes), f"sph_download({field})")
        return out

    def upload(self, field, arr):
        dtype, comp = _FIELD_SPEC[field]
        arr = np.ascontiguousarray(arr, dtype=dtype)
        self._chk(self.lib.sph_upload(self.h, int(field), _ptr(arr), arr.nbytes), f"sph_upload({field})")

    def stats(self):
        st = SphStats()
        self._chk(self.lib.sph_get_stats(self.h, C.byref(st)), "sph_get_stats")
        return {k: getattr(st, k) for k, _ in SphStats._fields_}

    # -- profiling
    def profile_enable(self, kernel_id=-1, on=True):
        self._chk(self.lib.sph_profile_enable(self.h, int(kernel_id), int(on)), "sph_profile_enable")

    def profile_reset(self):
        self._chk(self.lib.sph_profile_reset(self.h), "sph_profile_reset")

    def profile_read(self, kernel_id):
        n, ms = C.c_int64(), C.c_double()
        self._chk(self.lib.sph_profile_read(self.h, int(kernel_id), C.byref(n), C.byref(ms)), "sph_profile_read")
        return n.value, ms.value

    # -- multi-GPU (z-slab sharding)
    def comm_unique_id(self):
        buf = C.create_string_buffer(128)
        self._chk(self.lib.sph_comm_unique_id(buf), "sph_comm_unique_id")
        return buf.raw

    def comm_init(self, rank, nranks, unique_id: bytes):
        assert len(unique_id) == 128
        self._chk(self.lib.sph_comm_init(self.h, int(rank), int(nranks), C.c_char_p(unique_id)), "sph_comm_init")

    def comm_set_slab(self, z_lo, z_hi):
        self._chk(self.lib.sph_comm_set_slab(self.h, int(z_lo), int(z_hi)), "sph_comm_set_slab")

    def comm_add_global_count(self, n, n_fluid):
        self._chk(self.lib.sph_comm_add_global_count(self.h, int(n), int(n_fluid)), "sph_comm_add_global_count")

    def comm_set_rebalance(self, every_steps):
        self._chk(self.lib.sph_comm_set_rebalance(self.h, int(every_steps)), "sph_comm_set_rebalance")

    def comm_get_slab(self, counts=True):
        v = [C.c_int() for _ in range(4)]
        args = [C.byref(v[0]), C.byref(v[1])] + ([C.byref(v[2]), C.byref(v[3])] if counts else [None, None])
        self._chk(self.lib.sph_comm_get_slab(self.h, *args), "sph_comm_get_slab")
        out = dict(z_lo=v[0].value, z_hi=v[1].value)
        if counts:
            out.update(n_owned=v[2].value, n_ghost=v[3].value)
        return out

    def comm_allreduce(self, values, op="sum"):
        """In-place all-reduce of up to 16 doubles over the communicator; returns the list of reduced values."""
        vals = [float(v) for v in (values if isinstance(values, (list, tuple)) else [values])]
        buf = (C.c_double * len(vals))(*vals)
        self._chk(self.lib.sph_comm_allreduce(self.h, buf, len(vals), {"sum": 0, "max": 1, "min": 2}[op]), "sph_comm_allreduce")
        return list(buf)

    def comm_barrier(self):
        self._chk(self.lib.sph_comm_barrier(self.h), "sph_comm_barrier")

    def comm_selftest(self, n=4096):
        self._chk(self.lib.sph_comm_selftest(self.h, int(n)), "sph_comm_selftest")

    def measure_copy_rate(self, nbytes=1 << 30, reps=10):
        """GB/s (read + written) of device-to-device copies on this GPU, measured now."""
        v = C.c_double()
        self._chk(self.lib.sph_measure_copy_rate(self.h, int(nbytes), int(reps), C.byref(v)), "sph_measure_copy_rate")
        return v.value

    def comm_transport(self):
        return self.lib.sph_comm_transport(self.h).decode()

    def device_info(self):
        name = C.create_string_buffer(256)
        cu, mem = C.c_int(), C.c_int64()
        self._chk(self.lib.sph_device_info(self.h, name, C.byref(cu), C.byref(mem)), "sph_device_info")
        return {"name": name.value.decode(), "cu_count": cu.value, "hbm_bytes": mem.value}
