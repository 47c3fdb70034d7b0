"""Placeholder so the reference's `import pybullet as p` succeeds (fixture generation only).
PyBulletSolver never connects when the scene has no rigid bodies (bullet_solver.py:31-42)."""
DIRECT = 0
def __getattr__(name):
    raise RuntimeError("pybullet is not available: dynamic rigid bodies are outside the fixture scenes")
