"""Placeholder so the reference's `import trimesh as tm` succeeds (fixture generation only).
Mesh scenes are out of scope; any use raises."""
def __getattr__(name):
    raise RuntimeError("trimesh is not available: mesh bodies are outside the fixture scenes")
