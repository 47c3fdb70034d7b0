"""Host-side rigid-body hook (the reference's SPH/rigid_solver/bullet_solver.py).

Rigid-body dynamics stays on the CPU (<= 20 bodies); the device side only accumulates the
fluid->rigid force / torque per object and consumes the pose.  This class is the seam: `step()`
pulls the wrench from the engine (sph_get_rigid_wrench), lets a backend integrate the bodies and
pushes the poses back (sph_set_rigid_pose).  With PyBullet installed the backend reproduces
bullet_solver.py:144-167; without it (this image) scenes with dynamic bodies raise, static
bodies need no backend at all (bullet_solver.py:31-42 makes the same distinction).
"""
import numpy as np


class HostRigidSolver:
    def __init__(self, container, gravity=(0, -9.8, 0), dt=1e-3):
        self.container = container
        self.total_time = 0.0
        self.present_rigid_object = []
        self.gravity, self.dt = gravity, dt
        self.rigid_bodies = container.cfg.get_rigid_bodies()
        self.dynamic_ids = [b["objectId"] for b in self.rigid_bodies if b["isDynamic"]]
        self.backend = None
        if self.dynamic_ids:
            try:
                import pybullet  # noqa: F401
            except ImportError as exc:
                raise NotImplementedError("dynamic rigid bodies need pybullet on the host (absent in this image); "
                                          "the device side (wrench out / pose in) is in place") from exc
        else:
            if not self.rigid_bodies:
                print("No rigid body in the scene, skip bullet solver initialization.")

    def insert_rigid_object(self):
        for body in self.rigid_bodies:
            oid = body["objectId"]
            if oid in self.present_rigid_object or body["entryTime"] > self.total_time:
                continue
            self.present_rigid_object.append(oid)

    def step(self):
        if not self.dynamic_ids:
            return
        force, torque = self.container.engine.get_rigid_wrench(reset=True)
        self._integrate(force, torque)

    def _integrate(self, force, torque):  # pragma: no cover - needs pybullet
        raise NotImplementedError


PyBulletSolver = HostRigidSolver
