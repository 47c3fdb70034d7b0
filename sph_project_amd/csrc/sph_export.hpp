// sph_export.hpp -- frame export on the host: the ASCII PLY the reference writes per fluid object and frame
// (run_simulation.py:137-144: ti.tools.PLYWriter(num_vertices).add_vertex_pos(x, y, z).export_ascii(path)).  Plain C++ behind the C-ABI
// (included by sph_api.hip; host code only).  Layout restated from Taichi's python/taichi/tools/ply.py (not compared with a Taichi run: the
// package is absent): header "ply / format ascii 1.0 / comment created by PLYWriter / element vertex N / property float x|y|z /
// end_header", then per vertex every value as str(np.float32) followed by ONE blank, "\n" at the end of the line.
// Why native: numpy needs ~5 s to turn the 1.23 M positions of a frame into those strings (it was the 68 ms/step "all in" of the
// reference's final_scene0 in round 3); this writes the same bytes in ~0.2 s.
#pragma once
#include <charconv>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>

namespace sphexp {

// str(np.float32(v)): the shortest digits that round-trip (std::to_chars, like numpy's dragon4 in unique mode), positional with at least
// one digit behind the point for 0 and 1e-4 <= |v| < 1e16 ("0.1", "123456790.0", "-0.0"), else scientific with a two-digit exponent
// ("1e-05", "1.2345679e+16"); "inf" / "nan" as numpy prints them.  Returns the number of characters written (no terminator).
static inline int format_f32(float v, char *out) {
    char *p = out;
    if (std::isnan(v)) { memcpy(p, "nan", 3); return 3; }
    if (std::signbit(v)) { *p++ = '-'; v = -v; }
    if (std::isinf(v)) { memcpy(p, "inf", 3); return (int)(p - out) + 3; }
    if (v == 0.0f) { memcpy(p, "0.0", 3); return (int)(p - out) + 3; }
    char sci[48];
    const auto r = std::to_chars(sci, sci + sizeof(sci), v, std::chars_format::scientific);   // d[.ddd]e[+-]XX
    const int len = (int)(r.ptr - sci);
    *r.ptr = 0;   // (atoi below)
    const double a = (double)v;
    if (!(a < 1.e16 && a >= 1.e-4)) { memcpy(p, sci, (size_t)len); return (int)(p - out) + len; }
    int epos = 0;
    while (sci[epos] != 'e') ++epos;
    const int exp10 = atoi(sci + epos + 1);
    char dig[16]; int nd = 0;
    for (int k = 0; k < epos; ++k) if (sci[k] != '.') dig[nd++] = sci[k];
    if (exp10 >= 0) {
        for (int k = 0; k <= exp10; ++k) *p++ = k < nd ? dig[k] : '0';
        *p++ = '.';
        if (nd > exp10 + 1) for (int k = exp10 + 1; k < nd; ++k) *p++ = dig[k];
        else *p++ = '0';
    } else {
        *p++ = '0'; *p++ = '.';
        for (int k = 0; k < -exp10 - 1; ++k) *p++ = '0';
        for (int k = 0; k < nd; ++k) *p++ = dig[k];
    }
    return (int)(p - out);
}

static int write_ply_ascii(const char *path, const float *xyz, int64_t n) {
    if (!path || n < 0 || (n > 0 && !xyz)) return -1;
    FILE *f = fopen(path, "wb");
    if (!f) return -2;
    int rc = 0;
    if (fprintf(f, "ply\nformat ascii 1.0\ncomment created by PLYWriter\nelement vertex %lld\nproperty float x\nproperty float y\nproperty float z\nend_header\n", (long long)n) < 0) rc = -3;
    std::vector<char> buf(1 << 20);
    size_t used = 0;
    for (int64_t i = 0; i < n && rc == 0; ++i) {
        if (used + 3 * 48 + 4 > buf.size()) { if (fwrite(buf.data(), 1, used, f) != used) rc = -3; used = 0; }
        for (int c = 0; c < 3; ++c) { used += (size_t)format_f32(xyz[3 * i + c], buf.data() + used); buf[used++] = ' '; }
        buf[used++] = '\n';
    }
    if (rc == 0 && used && fwrite(buf.data(), 1, used, f) != used) rc = -3;
    if (fclose(f) != 0 && rc == 0) rc = -3;
    return rc;
}

}  // namespace sphexp
