// sph_voxel.hpp -- mesh -> particles on the host, dependency-free: what the reference gets from
// trimesh in base_container.py:611 load_rigid_body (`mesh.voxelized(pitch).fill().points`) and :676 load_fluid_body
// (`mesh.contains(lattice)`).  Plain C++ behind the C-ABI (included by sph_api.hip; host code only, not part of the
// accelerated path).  Parity with trimesh itself is UNPINNED (the package is not installed here): the algorithms are
// trimesh's published ones -- subdivision voxeliser (surface samples at half-pitch spacing rounded to the lattice of
// integer multiples of `pitch`, then hole filling = everything not 6-connected to the outside) and containment by
// ray-crossing parity -- and may differ from it in voxels the surface merely grazes.  A scene can always pin its
// particle set with a "voxelizedPoints" entry.
#pragma once
#include <algorithm>
#include <cfenv>
#include <cmath>
#include <cstdint>
#include <vector>

namespace sphvox {

struct V3 { double x, y, z; };

static inline long long round_half_even(double v) { return (long long)std::nearbyint(v); }   // FE_TONEAREST: ties to even, like np.round

// surface voxels: every triangle is sampled on a barycentric grid fine enough that consecutive samples are <= pitch / 2
// apart; samples are rounded to integer voxel indices
static void surface_voxels(const double *vert, const int32_t *faces, int nf, double pitch, std::vector<std::array<long long, 3>> &out) {
    const double step = 0.5 * pitch;
    for (int f = 0; f < nf; ++f) {
        const double *a = vert + 3 * (size_t)faces[3 * f], *b = vert + 3 * (size_t)faces[3 * f + 1], *c = vert + 3 * (size_t)faces[3 * f + 2];
        const double ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, ac[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
        const double cb[3] = {c[0] - b[0], c[1] - b[1], c[2] - b[2]};
        const double lab = std::sqrt(ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2]);
        const double lac = std::sqrt(ac[0] * ac[0] + ac[1] * ac[1] + ac[2] * ac[2]);
        const double lcb = std::sqrt(cb[0] * cb[0] + cb[1] * cb[1] + cb[2] * cb[2]);
        const int n = (int)std::ceil(std::max(lab, std::max(lac, lcb)) / step) + 1;
        for (int u = 0; u <= n; ++u)
            for (int v = 0; u + v <= n; ++v) {
                const double fu = (double)u / n, fv = (double)v / n;
                std::array<long long, 3> id;
                for (int k = 0; k < 3; ++k) id[k] = round_half_even(((a[k] + fu * ab[k]) + fv * ac[k]) / pitch);
                out.push_back(id);
            }
    }
}

// voxelise + fill: returns voxel centres (index * pitch) as float32 xyz, x slowest / z fastest
static int voxelize_fill(const double *vert, int nv, const int32_t *faces, int nf, double pitch, std::vector<float> &pts) {
    if (nv <= 0 || nf <= 0 || !(pitch > 0)) return -1;
    std::fesetround(FE_TONEAREST);
    std::vector<std::array<long long, 3>> sv;
    surface_voxels(vert, faces, nf, pitch, sv);
    long long lo[3] = {sv[0][0], sv[0][1], sv[0][2]}, hi[3] = {sv[0][0], sv[0][1], sv[0][2]};
    for (auto &p : sv) for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); }
    const long long nx = hi[0] - lo[0] + 1, ny = hi[1] - lo[1] + 1, nz = hi[2] - lo[2] + 1;
    if (nx * ny * nz > (1ll << 31)) return -2;
    // state: 0 free, 1 surface, 2 outside
    std::vector<uint8_t> g((size_t)(nx * ny * nz), 0);
    auto at = [&](long long x, long long y, long long z) -> uint8_t & { return g[(size_t)((x * ny + y) * nz + z)]; };
    for (auto &p : sv) at(p[0] - lo[0], p[1] - lo[1], p[2] - lo[2]) = 1;
    // everything 6-connected to the faces of the bounding box is outside (scipy.ndimage.binary_fill_holes' definition)
    std::vector<std::array<int, 3>> stack;
    auto push = [&](long long x, long long y, long long z) {
        if (x < 0 || y < 0 || z < 0 || x >= nx || y >= ny || z >= nz) return;
        uint8_t &c = at(x, y, z);
        if (c == 0) { c = 2; stack.push_back({(int)x, (int)y, (int)z}); }
    };
    for (long long x = 0; x < nx; ++x) for (long long y = 0; y < ny; ++y) { push(x, y, 0); push(x, y, nz - 1); }
    for (long long x = 0; x < nx; ++x) for (long long z = 0; z < nz; ++z) { push(x, 0, z); push(x, ny - 1, z); }
    for (long long y = 0; y < ny; ++y) for (long long z = 0; z < nz; ++z) { push(0, y, z); push(nx - 1, y, z); }
    while (!stack.empty()) {
        const auto p = stack.back(); stack.pop_back();
        push(p[0] - 1, p[1], p[2]); push(p[0] + 1, p[1], p[2]); push(p[0], p[1] - 1, p[2]);
        push(p[0], p[1] + 1, p[2]); push(p[0], p[1], p[2] - 1); push(p[0], p[1], p[2] + 1);
    }
    pts.clear();
    for (long long x = 0; x < nx; ++x) for (long long y = 0; y < ny; ++y) for (long long z = 0; z < nz; ++z)
        if (at(x, y, z) != 2) {
            pts.push_back((float)((double)(x + lo[0]) * pitch)); pts.push_back((float)((double)(y + lo[1]) * pitch));
            pts.push_back((float)((double)(z + lo[2]) * pitch));
        }
    return 0;
}

// lattice points (xs x ys x zs) inside the closed mesh: crossing parity along z.  The columns are shifted by an irrational
// fraction of 1e-7 of the mesh size so that no line runs exactly through an edge.  inside[(i * ny + j) * nz + k].
static int contains_lattice(const double *vert, int nv, const int32_t *faces, int nf, const double *xs, int nx, const double *ys,
                            int ny, const double *zs, int nz, uint8_t *inside) {
    if (nv <= 0 || nf <= 0 || nx < 0 || ny < 0 || nz < 0) return -1;
    double lo[3] = {vert[0], vert[1], vert[2]}, hi[3] = {vert[0], vert[1], vert[2]};
    for (int v = 0; v < nv; ++v) for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], vert[3 * v + k]); hi[k] = std::max(hi[k], vert[3 * v + k]); }
    const double size = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
    std::vector<double> X(nx), Y(ny);
    for (int i = 0; i < nx; ++i) X[i] = xs[i] + size * 1.2345e-7;
    for (int j = 0; j < ny; ++j) Y[j] = ys[j] + size * 0.7071e-7;
    std::vector<std::vector<double>> hits((size_t)nx * ny);
    for (int f = 0; f < nf; ++f) {
        const double *p0 = vert + 3 * (size_t)faces[3 * f], *p1 = vert + 3 * (size_t)faces[3 * f + 1], *p2 = vert + 3 * (size_t)faces[3 * f + 2];
        const double x0 = p0[0], y0 = p0[1], z0 = p0[2], x1 = p1[0], y1 = p1[1], z1 = p1[2], x2 = p2[0], y2 = p2[1], z2 = p2[2];
        const double det = (y1 - y2) * (x0 - x2) + (x2 - x1) * (y0 - y2);
        if (std::fabs(det) < 1e-300) continue;   // vertical triangle: no transversal crossing
        const int i0 = (int)(std::lower_bound(X.begin(), X.end(), std::min(x0, std::min(x1, x2))) - X.begin());
        const int i1 = (int)(std::upper_bound(X.begin(), X.end(), std::max(x0, std::max(x1, x2))) - X.begin());
        const int j0 = (int)(std::lower_bound(Y.begin(), Y.end(), std::min(y0, std::min(y1, y2))) - Y.begin());
        const int j1 = (int)(std::upper_bound(Y.begin(), Y.end(), std::max(y0, std::max(y1, y2))) - Y.begin());
        for (int i = i0; i < i1; ++i)
            for (int j = j0; j < j1; ++j) {
                const double a = ((y1 - y2) * (X[i] - x2) + (x2 - x1) * (Y[j] - y2)) / det;
                const double b = ((y2 - y0) * (X[i] - x2) + (x0 - x2) * (Y[j] - y2)) / det;
                const double c = 1.0 - a - b;
                if (a >= 0 && b >= 0 && c >= 0) hits[(size_t)i * ny + j].push_back(a * z0 + b * z1 + c * z2);
            }
    }
    for (int i = 0; i < nx; ++i)
        for (int j = 0; j < ny; ++j) {
            std::vector<double> &zc = hits[(size_t)i * ny + j];
            std::sort(zc.begin(), zc.end());
            if (zc.size() % 2) zc.pop_back();   // grazing contact: drop the odd one out
            for (int k = 0; k < nz; ++k) {
                // crossings at or below z, "at" with a tolerance of 1e-9 of the mesh size: a lattice that starts ON a face (np.arange from the
                // bounds' minimum, base_container.py:686-690: the lower faces of a box-shaped body) meets crossings computed 1 ulp above or
                // below its own z -- without the tolerance 14 of the 225 points on the lower z face of the reference's cube.obj were dropped.
                // Together with the positive column shift this makes containment half-open, [lo, hi), on all three axes, like np.arange.
                const size_t below = (size_t)(std::upper_bound(zc.begin(), zc.end(), zs[k] + size * 1e-9) - zc.begin());
                inside[((size_t)i * ny + j) * nz + k] = (below % 2) == 1;
            }
        }
    return 0;
}

}  // namespace sphvox
