"""Host-side rigid-body solver behind the reference's interface (SPH/rigid_solver/bullet_solver.py: PyBulletSolver with
insert_rigid_object() / step() / total_time).

Rigid-body dynamics stays on the CPU (<= 20 bodies); the device side only accumulates the fluid->rigid force / torque
per object and consumes the pose.  `step()` is bullet_solver.py:144-167: pull the wrench (sph_get_rigid_wrench =
rigid_body_forces / rigid_body_torques read + reset), integrate, push centre of mass, rotation, linear and angular
velocity back (sph_set_rigid_pose); the device turns the pose into particle positions / velocities
(_renew_rigid_particle_state, base_solver.py:616) at sph_step_end.

Two backends:
  * "native" (default): free rigid bodies under gravity and the fluid wrench, semi-implicit Euler, inertia tensor of the
    body's own particle set, inelastic contact of the body's particle-set EXTENT (its axis-aligned bounds in the current
    orientation) with the reference's boundary walls (bullet_solver.py:53-71) -- the reference collides the body's mesh
    with those wall boxes, so a body comes to rest ON the floor, not with its centre of mass in it.  No body-body
    contacts, no friction, no contact torque: that is Bullet's job, and a scene that needs it diverges from the
    reference -- the first dynamic body integrated by this backend therefore prints a one-time warning to stderr
    (silence it with SPH_RIGID_NATIVE_OK=1).  Unit-tested (tests/test_rigid_host.py).
  * "pybullet": the reference's calls (URDF from the mesh file, applyExternalForce / Torque at the base,
    stepSimulation, base pose read-back) when the package is importable.  It is not installed in this image, so this
    backend has never run here; select it with SPH_RIGID_BACKEND=pybullet.
Static bodies need no backend at all (bullet_solver.py:31-42 makes the same distinction).
"""
import math
import os
import sys

import numpy as np

_WARNED = [False]


def _rotation(angle, axis):
    """Orientation bullet_solver.py:97-101 builds: Euler angles (axis * angle) -> quaternion -> matrix (XYZ order)."""
    ex, ey, ez = [float(a) * angle for a in axis]
    cx, sx, cy, sy, cz, sz = math.cos(ex), math.sin(ex), math.cos(ey), math.sin(ey), math.cos(ez), math.sin(ez)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return rz @ ry @ rx


def _skew_exp(w):
    """exp([w]x): rotation by |w| about w (Rodrigues)."""
    th = float(np.linalg.norm(w))
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * (K @ K)


class _Body:
    def __init__(self, oid, mass, inertia_body, com, rot, vel):
        self.oid, self.mass = oid, float(mass)
        self.I_body = np.asarray(inertia_body, dtype=np.float64)
        self.I_body_inv = np.linalg.inv(self.I_body)
        self.com = np.asarray(com, dtype=np.float64).copy()
        self.rot = np.asarray(rot, dtype=np.float64).copy()
        self.vel = np.asarray(vel, dtype=np.float64).copy()
        self.angvel = np.zeros(3)
        self.points = None   # body-frame particle positions (wall contact uses their extent)


class HostRigidSolver:
    def __init__(self, container, gravity=(0, -9.8, 0), dt=1e-3):
        self.container = container
        self.total_time = 0.0
        self.present_rigid_object = []
        assert container.dim == 3, "the rigid solver only supports 3-D scenes (bullet_solver.py:19)"
        self.gravity, self.dt = np.asarray(gravity, dtype=np.float64), float(dt)
        self.rigid_bodies = container.cfg.get_rigid_bodies()
        self.rigid_blocks = container.cfg.get_rigid_blocks()
        self.bodies = {}   # object id -> _Body (dynamic ones)
        self.backend = os.environ.get("SPH_RIGID_BACKEND", "native")
        self._bullet = None
        if not self.rigid_bodies and not self.rigid_blocks:
            print("No rigid body in the scene, skip bullet solver initialization.")
        elif self.backend == "pybullet":
            try:
                self._bullet = _BulletBackend(container, self.gravity, self.dt)
            except ImportError as e:   # asked for explicitly: never fall back silently
                raise NotImplementedError("SPH_RIGID_BACKEND=pybullet, but pybullet is not importable") from e
        elif self.backend != "native":
            raise ValueError(f"SPH_RIGID_BACKEND={self.backend!r}: expected 'native' or 'pybullet'")
        # walls no part of a body may cross (bullet_solver.py:57-61)
        eps = container.padding + container.particle_diameter + container.domain_box_thickness
        self.wall_lo = np.asarray(container.domain_start, dtype=np.float64) + eps
        self.wall_hi = np.asarray(container.domain_end, dtype=np.float64) - eps

    # ------------------------------------------------------------------ bullet_solver.py:46-51, :75-131
    def insert_rigid_object(self):
        for body in self.rigid_bodies:
            oid = body["objectId"]
            if oid in self.present_rigid_object or body["entryTime"] > self.total_time:
                continue
            self.present_rigid_object.append(oid)
            if not body["isDynamic"]:
                continue
            c = self.container
            translation = np.asarray(body["translation"], dtype=np.float64)
            angle = body["rotationAngle"] / 360 * (2 * math.pi)
            rot = _rotation(angle, body["rotationAxis"])
            vel = np.asarray(body["velocity"], dtype=np.float64)
            # the body's particles were voxelised from the scaled, untransformed mesh (base_container.py:616-626): their
            # positions ARE the body-frame coordinates, with the base frame origin taken as centre of mass (:12-13)
            pts = np.asarray(body["voxelizedPoints"], dtype=np.float64)
            m_p = float(body["density"]) * float(c.V0)
            r2 = (pts * pts).sum(1)
            inertia = m_p * (np.eye(3) * r2.sum() - pts.T @ pts)
            if np.linalg.matrix_rank(inertia) < 3:   # degenerate (a single row of particles): regularise
                inertia = inertia + np.eye(3) * m_p * c.particle_diameter ** 2
            mass = float(c.rigid_body_masses[oid]) or m_p * len(pts)
            self.bodies[oid] = _Body(oid, mass, inertia, translation, rot, vel)
            self.bodies[oid].points = pts
            if self._bullet is None and not _WARNED[0] and not os.environ.get("SPH_RIGID_NATIVE_OK"):
                _WARNED[0] = True
                print("WARNING: dynamic rigid body %d is integrated by the built-in 'native' rigid backend: free-body motion under "
                      "gravity + the fluid wrench, wall contact by the body's axis-aligned extent, NO body-body contacts, friction or "
                      "contact torque (the reference uses PyBullet, bullet_solver.py).  Trajectories of bodies in contact diverge "
                      "from the reference; install pybullet and set SPH_RIGID_BACKEND=pybullet for its behaviour "
                      "(SPH_RIGID_NATIVE_OK=1 silences this)." % oid, file=sys.stderr, flush=True)
            if self._bullet is not None:
                self._bullet.add(body, mass, translation, angle, vel)
            self._push(self.bodies[oid], com0=np.zeros(3))
        for _ in self.rigid_blocks:
            raise NotImplementedError  # bullet_solver.py:133-135

    def _push(self, b, com0=None):
        self.container.engine.set_rigid_pose(b.oid, b.com, b.rot, b.vel, b.angvel, com0=com0)
        self.container.rigid_body_velocities[b.oid] = b.vel

    # ------------------------------------------------------------------ bullet_solver.py:144-167
    def step(self):
        if not self.bodies:
            return
        force, torque = self.container.engine.get_rigid_wrench(reset=True)
        if self._bullet is not None:
            self._bullet.step(self.bodies, force, torque)
        else:
            for b in self.bodies.values():
                self.integrate(b, force[b.oid].astype(np.float64), torque[b.oid].astype(np.float64))
        for b in self.bodies.values():
            self._push(b)

    def integrate(self, b, force, torque):
        """One semi-implicit Euler step of a free rigid body: external force at the centre of mass + gravity, external
        torque, gyroscopic term; then the walls."""
        dt = self.dt
        b.vel = b.vel + dt * (force / b.mass + self.gravity)
        I_inv = b.rot @ b.I_body_inv @ b.rot.T
        I_w = b.rot @ b.I_body @ b.rot.T
        b.angvel = b.angvel + dt * (I_inv @ (torque - np.cross(b.angvel, I_w @ b.angvel)))
        b.com = b.com + dt * b.vel
        b.rot = _skew_exp(dt * b.angvel) @ b.rot
        u, _, vt = np.linalg.svd(b.rot)   # keep it a rotation
        b.rot = u @ vt
        # inelastic wall contact of the body's extent: bounds of its particle set in the current orientation, about the
        # centre of mass (a body without a particle set -- unit tests -- is a point)
        if b.points is not None and len(b.points):
            w = b.points @ b.rot.T
            ext_lo, ext_hi = w.min(0), w.max(0)
        else:
            ext_lo = ext_hi = np.zeros(3)
        for k in range(3):
            if b.com[k] + ext_lo[k] < self.wall_lo[k] and self.wall_lo[k] - ext_lo[k] <= self.wall_hi[k] - ext_hi[k]:
                b.com[k] = self.wall_lo[k] - ext_lo[k]
                b.vel[k] = max(b.vel[k], 0.0)
            elif b.com[k] + ext_hi[k] > self.wall_hi[k]:
                b.com[k] = max(self.wall_hi[k] - ext_hi[k], self.wall_lo[k] - ext_lo[k])
                b.vel[k] = min(b.vel[k], 0.0)

    def get_rigid_body_states(self, container_idx):
        b = self.bodies[container_idx]
        return {"position": b.com.copy(), "rotation_matrix": b.rot.copy(), "linear_velocity": b.vel.copy(),
                "angular_velocity": b.angvel.copy()}


class _BulletBackend:   # pragma: no cover - needs pybullet (absent in this image)
    """The reference's PyBullet calls (bullet_solver.py:30-40, :86-127, :137-176)."""

    def __init__(self, container, gravity, dt):
        import pybullet as p
        import pybullet_data
        self.p, self.container, self.ids = p, container, {}
        self.client = p.connect(p.DIRECT)
        p.setAdditionalSearchPath(pybullet_data.getDataPath())
        p.setTimeStep(dt)
        p.setGravity(*[float(g) for g in gravity])

    def add(self, body, mass, translation, angle, vel):
        p = self.p
        mesh = body["geometryFile"]
        urdf = mesh[:-4] + ".urdf"
        s = body["scale"]
        with open(urdf, "w") as f:   # what SPH/utils create_urdf writes: one link, the mesh as visual + collision geometry
            f.write(f'<?xml version="1.0"?><robot name="b"><link name="base"><inertial><mass value="{mass}"/>'
                    f'<inertia ixx="1" ixy="0" ixz="0" iyy="1" iyz="0" izz="1"/></inertial>'
                    f'<collision><geometry><mesh filename="{mesh}" scale="{s[0]} {s[1]} {s[2]}"/></geometry></collision>'
                    f'</link></robot>')
        d = body["rotationAxis"]
        quat = p.getQuaternionFromEuler([d[0] * angle, d[1] * angle, d[2] * angle])
        self.ids[body["objectId"]] = p.loadURDF(urdf, basePosition=list(translation), baseOrientation=quat)
        os.remove(urdf)
        p.resetBaseVelocity(self.ids[body["objectId"]], list(vel))

    def step(self, bodies, force, torque):
        p = self.p
        for oid, b in bodies.items():
            pos, _ = p.getBasePositionAndOrientation(self.ids[oid])
            p.applyExternalForce(self.ids[oid], -1, forceObj=[float(x) for x in force[oid]], posObj=pos, flags=p.WORLD_FRAME)
            p.applyExternalTorque(self.ids[oid], -1, torqueObj=[float(x) for x in torque[oid]], flags=p.WORLD_FRAME)
        p.stepSimulation()
        for oid, b in bodies.items():
            lin, ang = p.getBaseVelocity(self.ids[oid])
            pos, orn = p.getBasePositionAndOrientation(self.ids[oid])
            b.com, b.vel, b.angvel = np.array(pos), np.array(lin), np.array(ang)
            b.rot = np.array(p.getMatrixFromQuaternion(orn)).reshape(3, 3)


PyBulletSolver = HostRigidSolver
