"""Mesh -> particles without trimesh (sph_project_amd/meshgen.py): restates base_container.py:611 / :676 of the
reference.  No golden vectors exist for this (trimesh is absent, parity unpinned): these are known-answer tests on
shapes whose particle sets can be written down."""
import numpy as np
import pytest

from sph_project_amd import meshgen as M

CUBE_V = [[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]]
CUBE_Q = [[1, 4, 3, 2], [5, 6, 7, 8], [1, 2, 6, 5], [3, 4, 8, 7], [2, 3, 7, 6], [4, 1, 5, 8]]   # quads, 1-based


def write_cube_obj(path, with_normals=False, negative=False):
    with open(path, "w") as fh:
        for v in CUBE_V:
            fh.write("v %g %g %g\n" % tuple(v))
        fh.write("vn 0 0 1\nvt 0 0\n")
        for q in CUBE_Q:
            idx = [k - 9 for k in q] if negative else q
            fh.write("f " + " ".join(("%d/1/1" % k) if with_normals else str(k) for k in idx) + "\n")


def icosphere(level=3):
    v = [[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]]
    f = [[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]]
    v = [np.array(p, float) for p in v]
    for _ in range(level):
        cache, nf = {}, []
        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        f = nf
    return M.Mesh(np.array(v), np.array(f))


@pytest.mark.parametrize("with_normals,negative", [(False, False), (True, False), (False, True)])
def test_obj_reader_and_cube_particles(tmp_path, with_normals, negative):
    path = tmp_path / "cube.obj"
    write_cube_obj(path, with_normals, negative)
    mesh = M.load_obj(str(path))
    assert mesh.vertices.shape == (8, 3) and mesh.faces.shape == (12, 3)
    pts = M.fluid_points(mesh, 0.1)                     # np.arange(0, 1, 0.1)^3, all inside the closed cube
    assert pts.shape == (1000, 3) and pts.dtype == np.float32
    assert np.allclose(pts[1] - pts[0], [0, 0, 0.1])     # meshgrid 'ij' order: z fastest
    vox = M.voxel_points(mesh, 0.1)                      # surface voxels + filled interior: lattice 0, 0.1, ..., 1.0
    assert vox.shape == (1331, 3)
    assert np.allclose(vox / 0.1, np.round(vox / 0.1), atol=1e-4)   # centres sit on integer multiples of the pitch
    assert np.all(np.lexsort((vox[:, 2], vox[:, 1], vox[:, 0])) == np.arange(len(vox)))   # x-major order


def test_place_rotates_about_the_centroid():
    mesh = M.Mesh(CUBE_V, [[0, 1, 2]])
    out = M.place(mesh, [2, 1, 1], np.pi / 2, [0, 1, 0], [1, 2, 3])
    c0 = (mesh.vertices * [2, 1, 1]).mean(axis=0)
    assert np.allclose(out.vertices.mean(axis=0), c0 + [1, 2, 3])
    # a quarter turn about y maps the x extent (2) onto z
    ext = out.bounds[1] - out.bounds[0]
    assert np.allclose(ext, [1, 1, 2])


def test_sphere_counts_match_its_volume():
    mesh = icosphere(3)
    mesh = M.place(mesh, [0.3, 0.3, 0.3], 0.0, [0, 1, 0], [1.0, 1.0, 1.0])
    pitch = 0.02
    pts = M.fluid_points(mesh, pitch)
    tri = mesh.vertices[mesh.faces]
    vol = abs(np.einsum("ij,ij->i", tri[:, 0], np.cross(tri[:, 1], tri[:, 2])).sum()) / 6.0
    assert abs(len(pts) * pitch ** 3 - vol) / vol < 0.03
    assert np.all(np.linalg.norm(pts - 1.0, axis=1) <= 0.3 + 1e-6)
    vox = M.voxel_points(mesh, pitch)
    # the filled voxel set covers the interior lattice and adds a shell of surface voxels
    assert len(vox) > len(pts) and len(vox) * pitch ** 3 < 1.25 * vol
    assert np.all(np.linalg.norm(vox - 1.0, axis=1) <= 0.3 + pitch)
