// Microbenchmark: is v_pk_{add,mul,fma}_f32 full rate on gfx950?  Distance test of 2 candidates per step,
// scalar (8 VALU per candidate) vs packed over candidate pairs (x1,x2),(y1,y2),(z1,z2): 6 pk + 2x(cmp+addc) per pair.
// hipcc --offload-arch=gfx950 -O3 mb_pk.hip -o mb_pk && ./mb_pk
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int VAR>
__global__ void __launch_bounds__(256) k(const float4 *__restrict__ in, unsigned *__restrict__ out, int iters, float h2) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    const float4 me = in[tid & 4095];
    float4 c0 = in[(tid + 1) & 4095], c1 = in[(tid + 2) & 4095];
    unsigned mask = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            // candidates drift so nothing is loop-invariant
            c0.x += 0.001f; c1.y -= 0.002f;
            if (VAR == 0) {
                {
                    const float dx = me.x - c0.x, dy = me.y - c0.y, dz = me.z - c0.z;
                    const float r2 = dx * dx + dy * dy + dz * dz;
                    asm("v_cmp_gt_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(r2), "v"(h2) : "vcc");
                }
                {
                    const float dx = me.x - c1.x, dy = me.y - c1.y, dz = me.z - c1.z;
                    const float r2 = dx * dx + dy * dy + dz * dz;
                    asm("v_cmp_gt_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(r2), "v"(h2) : "vcc");
                }
            } else {
                const v2f X = {c0.x, c1.x}, Y = {c0.y, c1.y}, Z = {c0.z, c1.z};
                const v2f mx = {me.x, me.x}, my = {me.y, me.y}, mz = {me.z, me.z};
                const v2f dx = mx - X, dy = my - Y, dz = mz - Z;
                v2f r2 = dx * dx;
                r2 = __builtin_elementwise_fma(dy, dy, r2);
                r2 = __builtin_elementwise_fma(dz, dz, r2);
                asm("v_cmp_gt_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(r2.x), "v"(h2) : "vcc");
                asm("v_cmp_gt_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(r2.y), "v"(h2) : "vcc");
            }
        }
    }
    out[tid] = mask;
}

int main() {
    float4 *in; unsigned *out;
    const int nb = 256 * 8 * 4, iters = 2000;
    CHK(hipMalloc(&in, 4096 * sizeof(float4))); CHK(hipMalloc(&out, nb * 256 * sizeof(unsigned)));
    CHK(hipMemset(in, 0x3c, 4096 * sizeof(float4)));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int var = 0; var < 2; ++var) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (var == 0) hipLaunchKernelGGL(k<0>, dim3(nb), dim3(256), 0, 0, in, out, iters, 0.5f);
            else hipLaunchKernelGGL(k<1>, dim3(nb), dim3(256), 0, 0, in, out, iters, 0.5f);
            hipEventRecord(e1); CHK(hipEventSynchronize(e1));
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double cand = (double)nb * 256 * iters * 16;
            if (rep) printf("var %d (%s): %.3f ms, %.2f G candidate-tests/s per GPU, %.2f cycles/wave/candidate @2.4GHz\n", var, var ? "packed pairs" : "scalar",
                            ms, cand / ms * 1e-6, ms * 1e-3 * 2.4e9 / ((double)nb * 4 / 1024 * iters * 16));
        }
    }
    return 0;
}
