from .host_rigid_solver import HostRigidSolver, PyBulletSolver
