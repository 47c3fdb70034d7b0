// Microbenchmark of the phase-1 inner loop (distance test of LDS-staged candidates).
// hipcc --offload-arch=gfx950 -O3 mb_phase1.hip -o mb_phase1 && ./mb_phase1
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CAP 1192
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ float2 lds_ld2v(const float2 *p) {
    typedef const volatile __attribute__((address_space(3))) unsigned long long *lds_u64;
    const unsigned long long v = *(lds_u64)(p);
    return make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
}

// VAR 0: two float2 arrays, plain loads (compiler may fuse to ds_read2_b64)
// VAR 1: two float2 arrays, volatile loads (ds_read_b64 x2)
// VAR 2: float4 array (ds_read_b128 / b96)
// VAR 3: no LDS in the loop (candidate from registers): VALU only
// VAR 4: LDS only (float2 x2 volatile), trivial VALU
// VAR 5: float2 xy + float z arrays (b64 + b32)
// VAR 6: two targets per lane, volatile float2 x2
template <int VAR, int EXTRA_LDS>
__global__ void __launch_bounds__(256) k(const float4 *__restrict__ in, unsigned *__restrict__ out, int runs, float h2) {
    __shared__ float2 sXY[CAP];
    __shared__ float2 sZW[CAP];
    __shared__ float4 sA[VAR == 2 ? CAP : 1];
    __shared__ float sZ[VAR == 5 ? CAP : 1];
    __shared__ char pad[EXTRA_LDS > 0 ? EXTRA_LDS : 1];
    const int tid = threadIdx.x;
    for (int t = tid; t < CAP; t += 256) {
        float4 a = in[(blockIdx.x * 37 + t) % 4096];
        sXY[t] = make_float2(a.x, a.y); sZW[t] = make_float2(a.z, a.w);
        if (VAR == 2) sA[t] = a;
        if (VAR == 5) sZ[t] = a.z;
    }
    if (EXTRA_LDS > 0 && tid == 0) pad[0] = 1;
    __syncthreads();
    const float4 me = in[(blockIdx.x * 256 + tid) % 4096];
    const float4 me2 = in[(blockIdx.x * 256 + tid + 1) % 4096];
    const float xi = me.x, yi = me.y, zi = me.z;
    unsigned acc = 0, acc2 = 0;
    for (int r = 0; r < runs; ++r) {
        int base = ((tid >> 3) * 8 + r * 24) % (CAP - 40);   // 8 lanes per cell, runs of 24..32 slots
        unsigned mask = 0, mask2 = 0;
        for (int t0 = 0; t0 < 32; t0 += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float x, y, z;
                if (VAR == 0) { float2 xy = sXY[base + t0 + u]; float2 zw = sZW[base + t0 + u]; x = xy.x; y = xy.y; z = zw.x; }
                else if (VAR == 1 || VAR == 4 || VAR == 6) { float2 xy = lds_ld2v(&sXY[base + t0 + u]); float2 zw = lds_ld2v(&sZW[base + t0 + u]); x = xy.x; y = xy.y; z = zw.x; }
                else if (VAR == 2) { float4 a = sA[base + t0 + u]; x = a.x; y = a.y; z = a.z; }
                else if (VAR == 5) { float2 xy = lds_ld2v(&sXY[base + t0 + u]); x = xy.x; y = xy.y; z = sZ[base + t0 + u]; }
                else { x = me.w + (float)(t0 + u); y = me.y * (float)(r + u); z = me.x + (float)u; }
                if (VAR == 4) { mask |= (__float_as_uint(x) ^ __float_as_uint(z)) >> 31 << (t0 + u); }
                else {
                    const float dx = xi - x, dy = yi - y, dz = zi - z;
                    const float r2 = dx * dx + dy * dy + dz * dz;
                    mask |= (r2 < h2 ? 1u : 0u) << (t0 + u);
                    if (VAR == 6) {
                        const float ex = me2.x - x, ey = me2.y - y, ez = me2.z - z;
                        const float q2 = ex * ex + ey * ey + ez * ez;
                        mask2 |= (q2 < h2 ? 1u : 0u) << (t0 + u);
                    }
                }
            }
        }
        acc += __popc(mask); acc2 += __popc(mask2);
    }
    out[blockIdx.x * 256 + tid] = acc + acc2;
}

// VAR 7/8: closer to the real kernel: per-lane run length m, wave-uniform __any trip counts, post-masking;
// 8 additionally builds the mask with v_cmp + v_addc_co (asm volatile)
template <int VAR>
__global__ void __launch_bounds__(256) k2(const float4 *__restrict__ in, unsigned *__restrict__ out, int runs, float h2, const int *__restrict__ cs) {
    __shared__ float2 sXY[CAP];
    __shared__ float2 sZW[CAP];
    __shared__ int s_cs[9][32];
    const int tid = threadIdx.x;
    for (int t = tid; t < CAP; t += 256) {
        float4 a = in[(blockIdx.x * 37 + t) % 4096];
        sXY[t] = make_float2(a.x, a.y); sZW[t] = make_float2(a.z, a.w);
    }
    for (int t = tid; t < 9 * 32; t += 256) s_cs[t / 32][t % 32] = cs[t] ;
    __syncthreads();
    const float4 me = in[(blockIdx.x * 256 + tid) % 4096];
    const float xi = me.x, yi = me.y, zi = me.z;
    const int i = blockIdx.x * 256 + tid;
    unsigned acc = 0;
    for (int r = 0; r < runs; ++r) {
        const int e = (tid >> 3) & 15;
        const int js = s_cs[r][e], je = s_cs[r][e + 3];      // 3 cells of 8..9 particles
        const int loff = 0;
        for (int j0 = js; __any(j0 < je); j0 += 32) {
            int m = je - j0; m = m < 0 ? 0 : (m > 32 ? 32 : m);
            int base = j0 + loff; base = base > CAP - 40 ? CAP - 40 : base;
            unsigned mask = 0; int S = 0;
            for (int t0 = 0; __any(t0 < m); t0 += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    float2 xy = lds_ld2v(&sXY[base + t0 + u]); float2 zw = lds_ld2v(&sZW[base + t0 + u]);
                    const float dx = xi - xy.x, dy = yi - xy.y, dz = zi - zw.x;
                    const float r2 = dx * dx + dy * dy + dz * dz;
                    if (VAR == 8) asm volatile("v_cmp_gt_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(r2), "v"(h2) : "vcc");
                    else mask |= (r2 < h2 ? 1u : 0u) << (t0 + u);
                }
                S += 8;
            }
            if (VAR == 8) { const int drop = S - m; mask = drop >= 32 ? 0u : (mask >> drop) << drop; }
            else mask &= m >= 32 ? 0xffffffffu : ((1u << m) - 1u);
            const unsigned self = (unsigned)(i - j0);
            if (self < 32u) mask &= ~(1u << self);
            acc += __popc(mask);
        }
    }
    out[blockIdx.x * 256 + tid] = acc;
}

template <int VAR> int run2(const char *name, const float4 *d_in, unsigned *d_out, int blocks, const int *d_cs) {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int runs = 9;
    hipLaunchKernelGGL((k2<VAR>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, runs, 0.0016f, d_cs);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int it = 0; it < 10; ++it) hipLaunchKernelGGL((k2<VAR>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, runs, 0.0016f, d_cs);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.1f us/launch\n", name, ms * 100.0);
    return 0;
}

template <int VAR, int EXTRA> int run(const char *name, const float4 *d_in, unsigned *d_out, int blocks) {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int runs = 9;
    hipLaunchKernelGGL((k<VAR, EXTRA>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, runs, 0.0016f);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int it = 0; it < 10; ++it) hipLaunchKernelGGL((k<VAR, EXTRA>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, runs, 0.0016f);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 100.0;  // per launch
    const double slots = (double)blocks * 4 * runs * 32;  // wave-slots
    printf("%-44s %8.1f us/launch  %6.2f CU-cycles per wave-slot (256 CUs @2.4GHz)\n", name, us, us * 2400.0 * 256 / slots);
    return 0;
}

int main() {
    std::vector<float4> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = make_float4((i % 17) * 0.011f, (i % 13) * 0.013f, (i % 11) * 0.012f, 1.0f);
    float4 *d_in; unsigned *d_out; const int blocks = 4810;
    CHK(hipMalloc(&d_in, 4096 * sizeof(float4))); CHK(hipMalloc(&d_out, blocks * 256 * sizeof(unsigned)));
    CHK(hipMemcpy(d_in, h.data(), 4096 * sizeof(float4), hipMemcpyHostToDevice));
    run<0, 0>("float2x2 plain", d_in, d_out, blocks);
    run<1, 0>("float2x2 volatile (ds_read_b64 x2)", d_in, d_out, blocks);
    run<2, 0>("float4 (b128/b96)", d_in, d_out, blocks);
    run<3, 0>("VALU only (no LDS in loop)", d_in, d_out, blocks);
    run<4, 0>("LDS only (b64 x2 volatile)", d_in, d_out, blocks);
    run<5, 0>("float2 + float (b64 + b32)", d_in, d_out, blocks);
    run<6, 0>("2 targets per lane, b64 x2 volatile", d_in, d_out, blocks);
    { std::vector<int> hcs(9 * 32); for (int r = 0; r < 9; ++r) for (int e = 0; e < 32; ++e) hcs[r * 32 + e] = r * 90 + e * 8 + (e % 3 == 0);
      int *d_cs; CHK(hipMalloc(&d_cs, hcs.size() * 4)); CHK(hipMemcpy(d_cs, hcs.data(), hcs.size() * 4, hipMemcpyHostToDevice));
      run2<7>("realistic loops, C++ mask", d_in, d_out, blocks, d_cs);
      run2<8>("realistic loops, v_cmp+v_addc mask", d_in, d_out, blocks, d_cs); }
    run<1, 24000>("float2x2 volatile, +24KB LDS (3 blocks/CU)", d_in, d_out, blocks);
    run<1, 60000>("float2x2 volatile, +60KB LDS (2 blocks/CU)", d_in, d_out, blocks);
    return 0;
}
