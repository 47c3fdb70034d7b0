"""CPU: the native backend of the host rigid-body solver (sph_project_amd/SPH/rigid_solver/host_rigid_solver.py), the
seam where the reference calls PyBullet (bullet_solver.py:144-167), against closed-form rigid-body motion; and the OBJ
writer behind exportObj (run_simulation.py:146-150)."""
import types

import numpy as np
import pytest

from sph_project_amd import meshgen
from sph_project_amd.SPH.rigid_solver import host_rigid_solver as R
from sph_project_amd.SPH.utils import SimConfig


class _Engine:
    def __init__(self):
        self.poses, self.force, self.torque = [], np.zeros((20, 3), np.float32), np.zeros((20, 3), np.float32)

    def set_rigid_pose(self, oid, com, rot, vel, angvel, com0=None):
        self.poses.append((oid, np.array(com), np.array(rot), np.array(vel), np.array(angvel), com0))

    def get_rigid_wrench(self, reset=True):
        f, t = self.force.copy(), self.torque.copy()
        if reset:
            self.force[:] = 0
            self.torque[:] = 0
        return f, t


def _container(bodies, domain_end=(4.0, 4.0, 4.0)):
    cfg = SimConfig(config={"Configuration": {}, "RigidBodies": bodies})
    c = types.SimpleNamespace(dim=3, cfg=cfg, padding=0.04, particle_diameter=0.02, domain_box_thickness=0.03,
                              domain_start=np.zeros(3), domain_end=np.array(domain_end), V0=0.8 * 0.02 ** 3,
                              rigid_body_masses=np.zeros(20, np.float32), rigid_body_velocities=np.zeros((20, 3), np.float32),
                              engine=_Engine())
    return c


def _cube_points(n=4, d=0.02):
    ax = (np.arange(n) - (n - 1) / 2) * d
    return np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)


def _body(oid=1, entry=-1.0, dynamic=True, velocity=(0.0, 0.0, 0.0), translation=(2.0, 2.0, 2.0), angle=0.0):
    return {"objectId": oid, "geometryFile": "x.obj", "voxelizedPoints": _cube_points(), "isDynamic": dynamic, "entryTime": entry,
            "density": 1000.0, "velocity": list(velocity), "translation": list(translation), "scale": [1, 1, 1],
            "rotationAngle": angle, "rotationAxis": [0, 0, 1], "color": [0, 0, 0]}


def test_free_fall_and_pose_push():
    c = _container([_body(velocity=(0.5, 0.0, 0.0))])
    s = R.HostRigidSolver(c, gravity=(0.0, -9.81, 0.0), dt=1e-3)
    s.insert_rigid_object()
    assert s.present_rigid_object == [1] and len(c.engine.poses) == 1
    oid, com, rot, vel, ang, com0 = c.engine.poses[0]
    np.testing.assert_allclose(com, [2, 2, 2]); np.testing.assert_allclose(rot, np.eye(3)); np.testing.assert_allclose(com0, 0)
    for _ in range(100):
        s.step()
    b = s.bodies[1]
    t = 0.1
    np.testing.assert_allclose(b.vel, [0.5, -9.81 * t, 0.0], rtol=1e-12)
    # semi-implicit Euler: x_n = x_0 + v_0 t - g dt^2 n (n + 1) / 2
    np.testing.assert_allclose(b.com, [2 + 0.5 * t, 2 - 9.81 * 1e-6 * 100 * 101 / 2, 2.0], rtol=1e-12)
    assert len(c.engine.poses) == 101 and np.allclose(c.rigid_body_velocities[1], b.vel)


def test_constant_torque_about_a_principal_axis():
    c = _container([_body()])
    s = R.HostRigidSolver(c, gravity=(0.0, 0.0, 0.0), dt=1e-3)
    s.insert_rigid_object()
    b = s.bodies[1]
    I = b.I_body
    assert np.allclose(I, np.diag(np.diag(I))) and np.allclose(np.diag(I), I[0, 0])   # a cube: isotropic tensor
    m_p = 1000.0 * c.V0
    pts = _cube_points()
    assert np.isclose(I[2, 2], m_p * (pts[:, 0] ** 2 + pts[:, 1] ** 2).sum())
    for _ in range(200):
        c.engine.torque[1] = [0.0, 0.0, 1e-4]
        c.engine.force[1] = [0.0, 0.0, 0.0]
        s.step()
    w = 1e-4 / I[2, 2] * 0.2
    np.testing.assert_allclose(b.angvel, [0, 0, w], rtol=1e-6, atol=1e-12)   # the wrench crosses the boundary as f32
    ang = 1e-4 / I[2, 2] * 1e-6 * 200 * 201 / 2   # accumulated angle
    np.testing.assert_allclose(b.rot, [[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], atol=1e-7)
    np.testing.assert_allclose(b.rot @ b.rot.T, np.eye(3), atol=1e-12)


def test_torque_free_precession_conserves_angular_momentum():
    body = _body()
    body["voxelizedPoints"] = _cube_points() * np.array([1.0, 2.0, 3.0])   # a brick: three distinct principal moments
    c = _container([body])
    s = R.HostRigidSolver(c, gravity=(0.0, 0.0, 0.0), dt=1e-4)
    s.insert_rigid_object()
    b = s.bodies[1]
    b.angvel = np.array([3.0, 0.2, 0.1])
    L0 = b.rot @ b.I_body @ b.rot.T @ b.angvel
    E0 = 0.5 * b.angvel @ L0
    for _ in range(2000):
        s.step()
    L1 = b.rot @ b.I_body @ b.rot.T @ b.angvel
    assert np.linalg.norm(L1 - L0) / np.linalg.norm(L0) < 2e-3
    assert abs(0.5 * b.angvel @ L1 - E0) / E0 < 5e-3


def test_walls_late_entry_and_static_bodies():
    c = _container([_body(oid=1, velocity=(0.0, -3.0, 0.0), translation=(2.0, 0.2, 2.0)), _body(oid=2, entry=0.05),
                    _body(oid=3, dynamic=False)], domain_end=(4.0, 4.0, 4.0))
    s = R.HostRigidSolver(c, gravity=(0.0, -9.81, 0.0), dt=1e-3)
    s.insert_rigid_object()
    assert sorted(s.present_rigid_object) == [1, 3] and list(s.bodies) == [1]   # static: registered, never integrated
    for k in range(100):
        s.step()
        s.total_time += s.dt
        s.insert_rigid_object()
    assert 2 in s.bodies
    eps = 0.04 + 0.02 + 0.03
    # resting ON the floor wall: the lowest particle of the body touches it (bullet_solver.py:53-71 collides the mesh with
    # the wall boxes), the centre of mass sits the body's half extent above
    b = s.bodies[1]
    low = (b.points @ b.rot.T)[:, 1].min()
    assert low < 0 and abs(b.com[1] + low - eps) < 1e-12 and b.vel[1] == 0.0
    st = s.get_rigid_body_states(1)
    assert set(st) == {"position", "rotation_matrix", "linear_velocity", "angular_velocity"}


def test_native_backend_warns_once_and_pybullet_request_is_never_silently_replaced(capsys, monkeypatch):
    monkeypatch.delenv("SPH_RIGID_NATIVE_OK", raising=False)
    R._WARNED[0] = False
    for _ in range(2):
        s = R.HostRigidSolver(_container([_body()]), dt=1e-3)
        s.insert_rigid_object()
    err = capsys.readouterr().err
    assert err.count("WARNING: dynamic rigid body") == 1 and "NO body-body contacts" in err
    monkeypatch.setenv("SPH_RIGID_BACKEND", "pybullet")
    try:
        import pybullet  # noqa: F401
    except ImportError:
        with pytest.raises(NotImplementedError):
            R.HostRigidSolver(_container([_body()]), dt=1e-3)
    monkeypatch.setenv("SPH_RIGID_BACKEND", "bogus")
    with pytest.raises(ValueError):
        R.HostRigidSolver(_container([_body()]), dt=1e-3)


def test_initial_orientation_matches_bullet_euler_convention():
    # bullet_solver.py:97-101: euler = axis * angle, quaternion from (roll X, pitch Y, yaw Z), R = Rz Ry Rx
    Rm = R._rotation(np.pi / 2, [0, 0, 1])
    np.testing.assert_allclose(Rm @ [1, 0, 0], [0, 1, 0], atol=1e-12)
    Rm = R._rotation(np.pi / 2, [1, 0, 0])
    np.testing.assert_allclose(Rm @ [0, 1, 0], [0, 0, 1], atol=1e-12)


def test_obj_export_round_trip(tmp_path):
    m = meshgen.Mesh([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], [[0, 1, 2], [0, 1, 3], [0, 2, 3], [1, 2, 3]])
    text = m.export(file_type="obj")
    p = tmp_path / "t.obj"
    p.write_text(text)
    back = meshgen.load_obj(str(p))
    np.testing.assert_allclose(back.vertices, m.vertices)
    np.testing.assert_array_equal(back.faces, m.faces)


def test_solver_step_dispatch_without_a_device():
    """BaseSolver.step() (base_solver.py:692) with a recording stand-in for the engine: without anything for the host to do inside the
    step it is ONE enqueue (WCSPH / fixed iterations: asynchronous; solver loops with their own stop tests: sph_step), with a dynamic rigid
    body or an object still waiting for its entryTime it is the begin / host hook / end sequence in the reference's order
    (WCSPH.py:39-43), and advance(n) hands n steps over as one call exactly when step() would not need the host."""
    from sph_project_amd.SPH.fluid_solvers.base_solver import BaseSolver

    calls = []

    class Eng:
        def step(self, n=1): calls.append(("step", n))
        def step_async(self, n=1): calls.append(("step_async", n))
        def step_begin(self): calls.append(("begin",))
        def step_end(self): calls.append(("end",))

    def make(method, fixed, bodies, pending):
        s = BaseSolver.__new__(BaseSolver)
        cfg = types.SimpleNamespace(get_cfg=lambda k: None)
        s.container = types.SimpleNamespace(METHOD=method, params_dict={"fixed_iterations": fixed}, total_time=0.0,
                                            objects_pending=lambda: pending, insert_object=lambda: calls.append(("insert",)))
        s.cfg, s.engine, s.dt = cfg, Eng(), {None: 1e-3}
        s.rigid_solver = types.SimpleNamespace(bodies=bodies, total_time=0.0, step=lambda: calls.append(("rigid",)),
                                               insert_rigid_object=lambda: calls.append(("insert_rigid",)))
        return s

    s = make("wcsph", 0, {}, False); s.step(); s.advance(7)
    assert calls == [("step_async", 1), ("step_async", 7)] and abs(s.container.total_time - 8e-3) < 1e-12
    calls.clear(); s = make("dfsph", 0, {}, False); s.step(); s.advance(3)
    assert calls == [("step", 1), ("step", 3)]          # the solver loops read their stop flag back
    calls.clear(); s = make("dfsph", 2, {}, False); s.step()
    assert calls == [("step_async", 1)]
    calls.clear(); s = make("wcsph", 0, {1: object()}, False); s.step()
    assert calls == [("begin",), ("rigid",), ("insert",), ("insert_rigid",), ("end",)]
    calls.clear(); s = make("wcsph", 0, {}, True); s.advance(2)
    assert calls == [("begin",), ("rigid",), ("insert",), ("insert_rigid",), ("end",)] * 2
