"""Mesh -> particles without trimesh: what base_container.py:611 `load_rigid_body`
(`mesh.voxelized(pitch).fill().points`) and :676 `load_fluid_body` (lattice points with `mesh.contains`) need.
`voxel_points` / `fluid_points` call the dependency-free C++ behind the C-ABI (`sph_voxelize_mesh`, `sph_points_in_mesh`:
sph_project_amd/csrc/sph_voxel.hpp); the numpy + scipy.ndimage restatement they replaced stays below as
`voxel_points_numpy` / `fluid_points_numpy`, the cross-check of tests/test_meshgen.py (same point sets, bit for bit).
Host-side preprocessing, not part of the accelerated path.

Parity: UNPINNED -- trimesh is not installed here, so these follow its published algorithms (subdivision voxeliser:
surface samples rounded to the lattice of integer multiples of `pitch`, then `binary_fill_holes`; containment by
ray-crossing parity), not its bit patterns: particle sets may differ from the reference's in voxels the surface
merely grazes.  A scene can always pin its particles with a "voxelizedPoints" entry instead.
"""
from __future__ import annotations

import numpy as np


class Mesh:
    """Minimal stand-in for the trimesh object the reference keeps in `object_collection[id]["mesh"]`."""

    def __init__(self, vertices, faces):
        self.vertices = np.asarray(vertices, dtype=np.float64).reshape(-1, 3)
        self.faces = np.asarray(faces, dtype=np.int64).reshape(-1, 3)

    def copy(self):
        return Mesh(self.vertices.copy(), self.faces.copy())

    @property
    def bounds(self):
        return np.stack([self.vertices.min(axis=0), self.vertices.max(axis=0)])

    def export(self, file_type="obj"):
        """Wavefront OBJ text (`v` / `f` records, 1-based indices): what run_simulation.py:146-150 writes per rigid body."""
        if file_type != "obj":
            raise NotImplementedError(file_type)
        lines = ["# sph_project_amd"]
        lines += ["v %.8f %.8f %.8f" % tuple(v) for v in self.vertices]
        lines += ["f %d %d %d" % tuple(f + 1) for f in self.faces]
        return "\n".join(lines) + "\n"


def load_obj(path):
    """Wavefront OBJ: `v` and `f` records (polygons are fan-triangulated, `v/vt/vn` and negative indices accepted)."""
    verts, faces = [], []
    with open(path) as fh:
        for line in fh:
            if line.startswith("v "):
                verts.append([float(t) for t in line.split()[1:4]])
            elif line.startswith("f "):
                idx = []
                for tok in line.split()[1:]:
                    k = int(tok.split("/")[0])
                    idx.append(k - 1 if k > 0 else len(verts) + k)
                for a in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[a], idx[a + 1]])
    if not verts or not faces:
        raise ValueError(f"{path}: no vertices / faces found")
    return Mesh(verts, faces)


def rotation_about(angle, axis, point):
    """4x4 rotation by `angle` (radians) about the line through `point` along `axis` (Rodrigues)."""
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    c, s = np.cos(angle), np.sin(angle)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    R = c * np.eye(3) + s * K + (1 - c) * np.outer(axis, axis)
    M = np.eye(4)
    M[:3, :3] = R
    M[:3, 3] = np.asarray(point) - R @ np.asarray(point)
    return M


def place(mesh, scale, angle, axis, translation):
    """scale -> rotate about the vertex centroid -> translate (base_container.py:613-626 / :679-685)."""
    v = mesh.vertices * np.asarray(scale, dtype=np.float64)
    M = rotation_about(angle, axis, v.mean(axis=0))
    v = v @ M[:3, :3].T + M[:3, 3]
    return Mesh(v + np.asarray(translation, dtype=np.float64), mesh.faces)


def _column_crossings(mesh, xs, ys):
    """z of every intersection of the vertical lines (xs[i], ys[j]) with the triangles: dict (i, j) -> list of z."""
    tri = mesh.vertices[mesh.faces]                      # (m, 3, 3)
    out = {}
    for t in tri:
        (x0, y0, z0), (x1, y1, z1), (x2, y2, z2) = t
        det = (y1 - y2) * (x0 - x2) + (x2 - x1) * (y0 - y2)
        if abs(det) < 1e-300:
            continue                                       # vertical triangle: no transversal crossing
        i0, i1 = np.searchsorted(xs, min(x0, x1, x2), "left"), np.searchsorted(xs, max(x0, x1, x2), "right")
        j0, j1 = np.searchsorted(ys, min(y0, y1, y2), "left"), np.searchsorted(ys, max(y0, y1, y2), "right")
        if i0 >= i1 or j0 >= j1:
            continue
        X, Y = np.meshgrid(xs[i0:i1], ys[j0:j1], indexing="ij")
        a = ((y1 - y2) * (X - x2) + (x2 - x1) * (Y - y2)) / det
        b = ((y2 - y0) * (X - x2) + (x0 - x2) * (Y - y2)) / det
        c = 1.0 - a - b
        inside = (a >= 0) & (b >= 0) & (c >= 0)
        for ii, jj in zip(*np.nonzero(inside)):
            out.setdefault((i0 + ii, j0 + jj), []).append(a[ii, jj] * z0 + b[ii, jj] * z1 + c[ii, jj] * z2)
    return out


def contains_lattice(mesh, axes):
    """Boolean (nx, ny, nz) array: lattice point inside the closed mesh (crossing parity along z).  The columns are
    shifted by an irrational fraction of 1e-7 of the mesh size so that no line runs exactly through an edge."""
    xs, ys, zs = [np.asarray(a, dtype=np.float64) for a in axes]
    size = float(np.ptp(mesh.vertices, axis=0).max())
    hits = _column_crossings(mesh, xs + size * 1.2345e-7, ys + size * 0.7071e-7)
    inside = np.zeros((len(xs), len(ys), len(zs)), dtype=bool)
    for (i, j), zc in hits.items():
        zc = np.sort(np.asarray(zc))
        if len(zc) % 2:                                     # grazing contact: drop the closest pair member
            zc = zc[:-1]
        below = np.searchsorted(zc, zs + size * 1e-9, side="right")   # crossings at or below every lattice z (tolerance: sph_voxel.hpp)
        inside[i, j, :] = (below % 2) == 1
    return inside


def _mesh_arrays(mesh):
    v = np.ascontiguousarray(mesh.vertices, dtype=np.float64).reshape(-1, 3)
    f = np.ascontiguousarray(mesh.faces, dtype=np.int32).reshape(-1, 3)
    return v, f


def fluid_points(mesh, pitch):
    """base_container.py:686-694: np.arange lattice over the bounding box, points the mesh contains, (n, 3) f32 in
    meshgrid 'ij' order.  Containment through sph_points_in_mesh (C++)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    lo, hi = mesh.bounds
    axes = [np.ascontiguousarray(np.arange(lo[k], hi[k], pitch), dtype=np.float64) for k in range(3)]
    v, f = _mesh_arrays(mesh)
    inside = np.zeros(len(axes[0]) * len(axes[1]) * len(axes[2]), dtype=np.uint8)
    rc = lib.sph_points_in_mesh(v.ctypes.data_as(C.c_void_p), v.shape[0], f.ctypes.data_as(C.c_void_p), f.shape[0],
                                axes[0].ctypes.data_as(C.c_void_p), len(axes[0]), axes[1].ctypes.data_as(C.c_void_p), len(axes[1]),
                                axes[2].ctypes.data_as(C.c_void_p), len(axes[2]), inside.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError(f"sph_points_in_mesh failed ({rc})")
    pts = np.array(np.meshgrid(*axes, sparse=False, indexing="ij"), dtype=np.float32).reshape(3, -1).T
    return np.ascontiguousarray(pts[inside.astype(bool)])


def voxel_points(mesh, pitch):
    """base_container.py:641-642 `mesh.voxelized(pitch).fill().points` through sph_voxelize_mesh (C++)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    v, f = _mesh_arrays(mesh)
    n = C.c_int64(0)
    args = (v.ctypes.data_as(C.c_void_p), v.shape[0], f.ctypes.data_as(C.c_void_p), f.shape[0], float(pitch))
    rc = lib.sph_voxelize_mesh(*args, None, 0, C.byref(n))
    if rc != 0:
        raise ValueError(f"sph_voxelize_mesh failed ({rc})")
    out = np.empty((n.value, 3), dtype=np.float32)
    rc = lib.sph_voxelize_mesh(*args, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n))
    if rc != 0:
        raise ValueError(f"sph_voxelize_mesh failed ({rc})")
    return out


def fluid_points_numpy(mesh, pitch):
    """numpy restatement of fluid_points (cross-check of the C++)."""
    lo, hi = mesh.bounds
    axes = [np.arange(lo[k], hi[k], pitch) for k in range(3)]
    inside = contains_lattice(mesh, axes)
    pts = np.array(np.meshgrid(*axes, sparse=False, indexing="ij"), dtype=np.float32).reshape(3, -1).T
    return np.ascontiguousarray(pts[inside.reshape(-1)])


def voxel_points_numpy(mesh, pitch):
    """numpy + scipy.ndimage restatement of voxel_points (cross-check of the C++): centres (integer multiples of pitch)
    of the voxels the surface touches plus the region they enclose."""
    from scipy import ndimage
    tri = mesh.vertices[mesh.faces]
    step = 0.5 * pitch
    samples = []
    for a, b, c in tri:
        n = int(np.ceil(max(np.linalg.norm(b - a), np.linalg.norm(c - a), np.linalg.norm(c - b)) / step)) + 1
        u, v = np.meshgrid(np.arange(n + 1), np.arange(n + 1), indexing="ij")
        keep = (u + v) <= n
        u, v = u[keep] / n, v[keep] / n
        samples.append(a + np.outer(u, b - a) + np.outer(v, c - a))
    idx = np.unique(np.round(np.concatenate(samples) / pitch).astype(np.int64), axis=0)
    lo = idx.min(axis=0)
    dense = np.zeros(idx.max(axis=0) - lo + 1, dtype=bool)
    dense[tuple((idx - lo).T)] = True
    dense = ndimage.binary_fill_holes(dense)
    return np.ascontiguousarray(((np.argwhere(dense) + lo) * pitch).astype(np.float32))
