// mb_issue.hip -- issue-rate microbenchmarks behind DESIGN.md's roofline argument for k_nbr_pass.
//
//   hipcc --offload-arch=gfx950 -O3 mb_issue.hip -o mb_issue && ./mb_issue > profiles/r02_mb_issue.txt
//
// Every kernel runs hand-written instruction blocks (inline asm, so the instruction mix is exactly what is named) with
// W = 1, 2, 4, 6, 8 waves per SIMD: 256-thread workgroups (one wave per SIMD), W workgroups per CU forced by a dynamic
// LDS allocation of 160 KiB / W, grid = 256 CUs x W.  Reported per op:
//   cyc/inst/wave   shader cycles (s_memtime) one wave needs per instruction = latency-ish view of ONE wave
//   cyc/inst/SIMD   the same divided by W = issue cost of a wave-instruction on the SIMD when W waves interleave
//   wall cyc/inst   wall time x 2.4 GHz / (instructions per SIMD): cross-check without s_memtime (DVFS lowers it)
// The guide (MI355X_MICROARCH.md) quotes 2 cycles per wave64 VALU instruction on the SIMD-32.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

enum { OP_FMA = 0, OP_SUB, OP_MUL, OP_CMP_ADDC, OP_PK_FMA, OP_DIST8, OP_LDS_B64, OP_LDS_B32, OP_LDS_B128,
       OP_P1_CUR, OP_P1_SOA, OP_P1_TWO, OP_P1_SOA_TWO, OP_P1_SOA_PK, OP_COUNT };
static const char *op_name[OP_COUNT] = {
    "v_fma_f32 x8 independent", "v_sub_f32 x8 independent", "v_mul_f32 x8 independent", "v_cmp_gt_f32+v_addc_co_u32",
    "v_pk_fma_f32 x8 independent", "distance test, 8 VALU/candidate (regs)", "ds_read_b64 x16 + wait", "ds_read_b32 x16 + wait",
    "ds_read_b128 x8 + wait", "phase-1 chunk: 8x(b64+b32) + 64 VALU", "phase-1 chunk SoA: 6x b128 + 64 VALU",
    "phase-1 chunk, 2 particles/lane: 16 LDS + 128 VALU", "phase-1 SoA, 2 particles/lane: 6x b128 + 128 VALU",
    "phase-1 SoA packed: 6x b128 + 24 v_pk + 16 VALU"};
// instructions per block (VALU or LDS, what the op is about), candidates tested per block
static const int op_inst[OP_COUNT] = {8, 8, 8, 2, 8, 64, 16, 16, 8, 80, 70, 144, 134, 46};
static const int op_valu[OP_COUNT] = {8, 8, 8, 2, 8, 64, 0, 0, 0, 64, 64, 128, 128, 40};
static const int op_cand[OP_COUNT] = {0, 0, 0, 0, 0, 8, 0, 0, 0, 8, 8, 16, 16, 8};

#define DIST1(xj, yj, zj, XI, YI, ZI, M)                                                                       \
    "v_sub_f32 %[dx], " XI ", " xj "\n\tv_sub_f32 %[dy], " YI ", " yj "\n\tv_sub_f32 %[dz], " ZI ", " zj "\n\t" \
    "v_mul_f32 %[r2], %[dx], %[dx]\n\tv_fmac_f32 %[r2], %[dy], %[dy]\n\tv_fmac_f32 %[r2], %[dz], %[dz]\n\t"    \
    "v_cmp_gt_f32 vcc, %[h2], %[r2]\n\tv_addc_co_u32 " M ", vcc, " M ", " M ", vcc\n\t"

template <int OP>
__global__ void __launch_bounds__(256) k(const float4 *__restrict__ in, unsigned *__restrict__ out, long long *__restrict__ cyc,
                                         int iters, float h2) {
    extern __shared__ float4 lds[];   // 160 KiB / W: caps the workgroups per CU at W
    const int tid = threadIdx.x;
    float *ldsf = (float *)lds;
    for (int t = tid; t < 4096; t += 256) ldsf[t] = in[t & 1023].x + (float)t * 1e-3f;
    __syncthreads();
    const float4 me = in[(blockIdx.x * 256 + tid) & 1023];
    float a0 = me.x, a1 = me.y, a2 = me.z, a3 = me.w, a4 = me.x + 1.f, a5 = me.y + 1.f, a6 = me.z + 1.f, a7 = me.w + 1.f;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float b = me.y * 0.999f, c = me.z * 1e-3f;
    const v2f pb = {b, b}, pc = {c, c};
    unsigned mask = 0, mask2 = 0;
    // 8 lanes share a "cell": same LDS address (broadcast), neighbouring cells 64 B apart, like the real tile
    const unsigned base = (unsigned)((tid >> 3) * 64);
    float dx, dy, dz, r2;
    const float xi = me.x, yi = me.y, zi = me.z, xk = me.x + 0.01f, yk = me.y + 0.01f, zk = me.z + 0.01f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (OP == OP_FMA)
            asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                         "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        else if (OP == OP_SUB)
            asm volatile("v_sub_f32 %0, %0, %8\n\tv_sub_f32 %1, %1, %8\n\tv_sub_f32 %2, %2, %8\n\tv_sub_f32 %3, %3, %8\n\t"
                         "v_sub_f32 %4, %4, %8\n\tv_sub_f32 %5, %5, %8\n\tv_sub_f32 %6, %6, %8\n\tv_sub_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        else if (OP == OP_MUL)
            asm volatile("v_mul_f32 %0, %0, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\t"
                         "v_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        else if (OP == OP_CMP_ADDC)
            asm volatile("v_cmp_gt_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(a0), "v"(h2) : "vcc");
        else if (OP == OP_PK_FMA)
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n\tv_pk_fma_f32 %1, %1, %8, %9\n\tv_pk_fma_f32 %2, %2, %8, %9\n\tv_pk_fma_f32 %3, %3, %8, %9\n\t"
                         "v_pk_fma_f32 %4, %4, %8, %9\n\tv_pk_fma_f32 %5, %5, %8, %9\n\tv_pk_fma_f32 %6, %6, %8, %9\n\tv_pk_fma_f32 %7, %7, %8, %9"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));
        else if (OP == OP_DIST8) {
            asm volatile(DIST1("%[c0]", "%[c1]", "%[c2]", "%[xi]", "%[yi]", "%[zi]", "%[m]") DIST1("%[c1]", "%[c2]", "%[c3]", "%[xi]", "%[yi]", "%[zi]", "%[m]")
                         DIST1("%[c2]", "%[c3]", "%[c4]", "%[xi]", "%[yi]", "%[zi]", "%[m]") DIST1("%[c3]", "%[c4]", "%[c5]", "%[xi]", "%[yi]", "%[zi]", "%[m]")
                         DIST1("%[c4]", "%[c5]", "%[c6]", "%[xi]", "%[yi]", "%[zi]", "%[m]") DIST1("%[c5]", "%[c6]", "%[c7]", "%[xi]", "%[yi]", "%[zi]", "%[m]")
                         DIST1("%[c6]", "%[c7]", "%[c0]", "%[xi]", "%[yi]", "%[zi]", "%[m]") DIST1("%[c7]", "%[c0]", "%[c1]", "%[xi]", "%[yi]", "%[zi]", "%[m]")
                         : [m] "+v"(mask), [dx] "=&v"(dx), [dy] "=&v"(dy), [dz] "=&v"(dz), [r2] "=&v"(r2)
                         : [c0] "v"(a0), [c1] "v"(a1), [c2] "v"(a2), [c3] "v"(a3), [c4] "v"(a4), [c5] "v"(a5), [c6] "v"(a6), [c7] "v"(a7),
                           [xi] "v"(xi), [yi] "v"(yi), [zi] "v"(zi), [h2] "v"(h2) : "vcc");
        } else if (OP == OP_LDS_B64) {
            v2f q0, q1, q2, q3, q4, q5, q6, q7;
            asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:8\n\tds_read_b64 %2, %8 offset:16\n\tds_read_b64 %3, %8 offset:24\n\t"
                         "ds_read_b64 %4, %8 offset:32\n\tds_read_b64 %5, %8 offset:40\n\tds_read_b64 %6, %8 offset:48\n\tds_read_b64 %7, %8 offset:56\n\t"
                         "ds_read_b64 %0, %8 offset:4096\n\tds_read_b64 %1, %8 offset:4104\n\tds_read_b64 %2, %8 offset:4112\n\tds_read_b64 %3, %8 offset:4120\n\t"
                         "ds_read_b64 %4, %8 offset:4128\n\tds_read_b64 %5, %8 offset:4136\n\tds_read_b64 %6, %8 offset:4144\n\tds_read_b64 %7, %8 offset:4152\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(q6), "=&v"(q7) : "v"(base) : "memory");
            a0 += q0.x + q7.y;
        } else if (OP == OP_LDS_B32) {
            float q0, q1, q2, q3, q4, q5, q6, q7;
            asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:8\n\tds_read_b32 %2, %8 offset:16\n\tds_read_b32 %3, %8 offset:24\n\t"
                         "ds_read_b32 %4, %8 offset:32\n\tds_read_b32 %5, %8 offset:40\n\tds_read_b32 %6, %8 offset:48\n\tds_read_b32 %7, %8 offset:56\n\t"
                         "ds_read_b32 %0, %8 offset:4096\n\tds_read_b32 %1, %8 offset:4104\n\tds_read_b32 %2, %8 offset:4112\n\tds_read_b32 %3, %8 offset:4120\n\t"
                         "ds_read_b32 %4, %8 offset:4128\n\tds_read_b32 %5, %8 offset:4136\n\tds_read_b32 %6, %8 offset:4144\n\tds_read_b32 %7, %8 offset:4152\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(q6), "=&v"(q7) : "v"(base) : "memory");
            a0 += q0 + q7;
        } else if (OP == OP_LDS_B128) {
            v4f q0, q1, q2, q3, q4, q5, q6, q7;
            asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\tds_read_b128 %3, %8 offset:48\n\t"
                         "ds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:4112\n\tds_read_b128 %6, %8 offset:4128\n\tds_read_b128 %7, %8 offset:4144\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(q6), "=&v"(q7) : "v"(base) : "memory");
            a0 += q0.x + q7.w;
        } else if (OP == OP_P1_CUR || OP == OP_P1_TWO) {
            v2f xy0, xy1, xy2, xy3, xy4, xy5, xy6, xy7;
            float z0, z1, z2, z3, z4, z5, z6, z7;
            asm volatile("ds_read_b64 %0, %16\n\tds_read_b32 %8, %16 offset:8192\n\tds_read_b64 %1, %16 offset:8\n\tds_read_b32 %9, %16 offset:8200\n\t"
                         "ds_read_b64 %2, %16 offset:16\n\tds_read_b32 %10, %16 offset:8208\n\tds_read_b64 %3, %16 offset:24\n\tds_read_b32 %11, %16 offset:8216\n\t"
                         "ds_read_b64 %4, %16 offset:32\n\tds_read_b32 %12, %16 offset:8224\n\tds_read_b64 %5, %16 offset:40\n\tds_read_b32 %13, %16 offset:8232\n\t"
                         "ds_read_b64 %6, %16 offset:48\n\tds_read_b32 %14, %16 offset:8240\n\tds_read_b64 %7, %16 offset:56\n\tds_read_b32 %15, %16 offset:8248\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(xy0), "=&v"(xy1), "=&v"(xy2), "=&v"(xy3), "=&v"(xy4), "=&v"(xy5), "=&v"(xy6), "=&v"(xy7),
                           "=&v"(z0), "=&v"(z1), "=&v"(z2), "=&v"(z3), "=&v"(z4), "=&v"(z5), "=&v"(z6), "=&v"(z7) : "v"(base) : "memory");
#define T1(xy, z, XI, YI, ZI, M) { const float ddx = XI - xy.x, ddy = YI - xy.y, ddz = ZI - z; const float rr = ddx * ddx + ddy * ddy + ddz * ddz; \
            asm("v_cmp_gt_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(M) : "v"(rr), "v"(h2) : "vcc"); }
            T1(xy0, z0, xi, yi, zi, mask) T1(xy1, z1, xi, yi, zi, mask) T1(xy2, z2, xi, yi, zi, mask) T1(xy3, z3, xi, yi, zi, mask)
            T1(xy4, z4, xi, yi, zi, mask) T1(xy5, z5, xi, yi, zi, mask) T1(xy6, z6, xi, yi, zi, mask) T1(xy7, z7, xi, yi, zi, mask)
            if (OP == OP_P1_TWO) {
                T1(xy0, z0, xk, yk, zk, mask2) T1(xy1, z1, xk, yk, zk, mask2) T1(xy2, z2, xk, yk, zk, mask2) T1(xy3, z3, xk, yk, zk, mask2)
                T1(xy4, z4, xk, yk, zk, mask2) T1(xy5, z5, xk, yk, zk, mask2) T1(xy6, z6, xk, yk, zk, mask2) T1(xy7, z7, xk, yk, zk, mask2)
            }
        } else if (OP == OP_P1_SOA || OP == OP_P1_SOA_TWO) {
            v4f X0, X1, Y0, Y1, Z0, Z1;   // x[8], y[8], z[8] of 8 consecutive candidates
            asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\tds_read_b128 %2, %6 offset:4096\n\tds_read_b128 %3, %6 offset:4112\n\t"
                         "ds_read_b128 %4, %6 offset:8192\n\tds_read_b128 %5, %6 offset:8208\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(X0), "=&v"(X1), "=&v"(Y0), "=&v"(Y1), "=&v"(Z0), "=&v"(Z1) : "v"(base >> 1) : "memory");
#define T2(x, y, z, XI, YI, ZI, M) { const float ddx = XI - x, ddy = YI - y, ddz = ZI - z; const float rr = ddx * ddx + ddy * ddy + ddz * ddz; \
            asm("v_cmp_gt_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(M) : "v"(rr), "v"(h2) : "vcc"); }
            T2(X0.x, Y0.x, Z0.x, xi, yi, zi, mask) T2(X0.y, Y0.y, Z0.y, xi, yi, zi, mask) T2(X0.z, Y0.z, Z0.z, xi, yi, zi, mask) T2(X0.w, Y0.w, Z0.w, xi, yi, zi, mask)
            T2(X1.x, Y1.x, Z1.x, xi, yi, zi, mask) T2(X1.y, Y1.y, Z1.y, xi, yi, zi, mask) T2(X1.z, Y1.z, Z1.z, xi, yi, zi, mask) T2(X1.w, Y1.w, Z1.w, xi, yi, zi, mask)
            if (OP == OP_P1_SOA_TWO) {
                T2(X0.x, Y0.x, Z0.x, xk, yk, zk, mask2) T2(X0.y, Y0.y, Z0.y, xk, yk, zk, mask2) T2(X0.z, Y0.z, Z0.z, xk, yk, zk, mask2) T2(X0.w, Y0.w, Z0.w, xk, yk, zk, mask2)
                T2(X1.x, Y1.x, Z1.x, xk, yk, zk, mask2) T2(X1.y, Y1.y, Z1.y, xk, yk, zk, mask2) T2(X1.z, Y1.z, Z1.z, xk, yk, zk, mask2) T2(X1.w, Y1.w, Z1.w, xk, yk, zk, mask2)
            }
        } else if (OP == OP_P1_SOA_PK) {
            // candidates two at a time: (x0,x1) ... straight out of the b128 registers; 6 v_pk per pair + 2 x (cmp + addc)
            v4f X0, X1, Y0, Y1, Z0, Z1;
            const v2f xi2 = {xi, xi}, yi2 = {yi, yi}, zi2 = {zi, zi};
            asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\tds_read_b128 %2, %6 offset:4096\n\tds_read_b128 %3, %6 offset:4112\n\t"
                         "ds_read_b128 %4, %6 offset:8192\n\tds_read_b128 %5, %6 offset:8208\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(X0), "=&v"(X1), "=&v"(Y0), "=&v"(Y1), "=&v"(Z0), "=&v"(Z1) : "v"(base >> 1) : "memory");
#define PK2(XX, YY, ZZ, lo, hi) { v2f xx = {XX.lo, XX.hi}, yy = {YY.lo, YY.hi}, zz = {ZZ.lo, ZZ.hi}; v2f ddx, ddy, ddz, rr; \
            asm("v_pk_add_f32 %0, %4, %5 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %1, %6, %7 neg_lo:[0,1] neg_hi:[0,1]\n\t" \
                "v_pk_add_f32 %2, %8, %9 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_mul_f32 %3, %0, %0\n\tv_pk_fma_f32 %3, %1, %1, %3\n\tv_pk_fma_f32 %3, %2, %2, %3" \
                : "=&v"(ddx), "=&v"(ddy), "=&v"(ddz), "=&v"(rr) : "v"(xi2), "v"(xx), "v"(yi2), "v"(yy), "v"(zi2), "v"(zz)); \
            asm("v_cmp_gt_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(rr.x), "v"(h2) : "vcc"); \
            asm("v_cmp_gt_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(rr.y), "v"(h2) : "vcc"); }
            PK2(X0, Y0, Z0, x, y) PK2(X0, Y0, Z0, z, w) PK2(X1, Y1, Z1, x, y) PK2(X1, Y1, Z1, z, w)
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * 256 + tid] = mask + mask2 + __float_as_uint(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y);
    if ((tid & 63) == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int OP> void run(int W, const float4 *d_in, unsigned *d_out, long long *d_cyc, int iters) {
    const int blocks = 256 * W;
    const size_t lds = (size_t)(160 * 1024 / W) / 256 * 256 - (W == 1 ? 0 : 512);   // W of these fit a CU, W + 1 do not
    CHK(hipFuncSetAttribute((const void *)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), lds, 0, d_in, d_out, d_cyc, iters / 8 + 1, 0.0016f);   // warm-up
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), lds, 0, d_in, d_out, d_cyc, iters, 0.0016f);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h((size_t)blocks * 4);
    CHK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double mean = 0; for (long long v : h) mean += (double)v; mean /= (double)h.size();
    const double ninst = (double)iters * op_inst[OP];
    const double per_wave = mean / ninst, per_simd = per_wave / W, wall = ms * 1e-3 * 2.4e9 / (ninst * W);
    printf("%-52s W=%d  cyc/inst/wave %7.2f  cyc/inst/SIMD %6.2f  wall@2.4GHz %6.2f", op_name[OP], W, per_wave, per_simd, wall);
    if (op_cand[OP]) printf("  | cyc/candidate/SIMD %6.2f  => 216 cand x 18.8 waves/SIMD = %6.1f us @2.4GHz", per_simd * op_inst[OP] / op_cand[OP],
                            per_simd * op_inst[OP] / op_cand[OP] * 216 * 18.79 / 2400.0);
    printf("\n");
    (void)op_valu;
}

template <int OP> void sweep(const float4 *d_in, unsigned *d_out, long long *d_cyc, int iters) {
    const int Ws[5] = {1, 2, 4, 6, 8};
    for (int W : Ws) run<OP>(W, d_in, d_out, d_cyc, iters);
}

int main() {
    std::vector<float4> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = make_float4((i % 17) * 0.011f + 0.5f, (i % 13) * 0.013f + 0.5f, (i % 11) * 0.012f + 0.5f, 1.0f);
    float4 *d_in; unsigned *d_out; long long *d_cyc;
    CHK(hipMalloc(&d_in, 1024 * sizeof(float4))); CHK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(unsigned))); CHK(hipMalloc(&d_cyc, 256 * 8 * 4 * 8));
    CHK(hipMemcpy(d_in, h.data(), 1024 * sizeof(float4), hipMemcpyHostToDevice));
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    printf("# device %s, %d CUs, clockRate %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    sweep<OP_FMA>(d_in, d_out, d_cyc, 20000); sweep<OP_SUB>(d_in, d_out, d_cyc, 20000); sweep<OP_MUL>(d_in, d_out, d_cyc, 20000);
    sweep<OP_CMP_ADDC>(d_in, d_out, d_cyc, 40000); sweep<OP_PK_FMA>(d_in, d_out, d_cyc, 20000); sweep<OP_DIST8>(d_in, d_out, d_cyc, 4000);
    sweep<OP_LDS_B64>(d_in, d_out, d_cyc, 4000); sweep<OP_LDS_B32>(d_in, d_out, d_cyc, 4000); sweep<OP_LDS_B128>(d_in, d_out, d_cyc, 4000);
    sweep<OP_P1_CUR>(d_in, d_out, d_cyc, 4000); sweep<OP_P1_SOA>(d_in, d_out, d_cyc, 4000); sweep<OP_P1_TWO>(d_in, d_out, d_cyc, 4000);
    sweep<OP_P1_SOA_TWO>(d_in, d_out, d_cyc, 4000);
    sweep<OP_P1_SOA_PK>(d_in, d_out, d_cyc, 4000);
    return 0;
}
