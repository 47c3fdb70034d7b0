// mb_gridbar.hip -- what a device-wide barrier costs on this MI355X, by itself (a persistent CG / sort kernel
// stands or falls with it).
//
//   hipcc --offload-arch=gfx950 -O3 tools/mb/mb_gridbar.hip -o tools/mb/mb_gridbar && tools/mb/mb_gridbar > profiles/r05_mb_gridbar.txt
//
// B workgroups of 256 threads, all resident (B <= 256 CUs x WG/CU by the occupancy query), run K barriers back to back; per-barrier
// cost = (t(K) - t(0)) / K from HIP events.  Variants:
//   flat : one monotonic counter, thread 0 of every workgroup: release fence, atomic add, relaxed poll + s_sleep, acquire fence
//   xcd  : hierarchical -- per-XCC counter (workgroups of one XCD meet on their own line), the last arriver of an XCC goes to the top
//          counter, the last XCC publishes the generation to eight per-XCC words everybody else polls
// and, for comparison, the dependent kernel boundary it would replace: 1000 empty launches of the same grid on one stream.
// Every spin is bounded (a barrier that cannot complete sets a flag and the kernel leaves).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Bar {
    unsigned flat;            unsigned pad0[31];
    unsigned top;             unsigned pad1[31];
    unsigned xcc_cnt[8][32];  // one 128-byte line per XCC
    unsigned xcc_gen[8][32];
    unsigned xcc_pop[8][32];  // workgroups per XCC (census of this launch)
    unsigned n_xcc;           unsigned timeout; unsigned pad2[30];
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 7u;
}
__device__ __forceinline__ unsigned ld(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool spin_until(const unsigned *p, unsigned target, unsigned *timeout) {
    for (unsigned spins = 0; (int)(ld(p) - target) < 0; ++spins) {
        __builtin_amdgcn_s_sleep(1);
        if (spins > (1u << 22)) { atomicOr(timeout, 1u); return false; }
    }
    return true;
}

// flat barrier number `epoch` (1-based) of B workgroups
__device__ __forceinline__ bool bar_flat(Bar *b, unsigned B, unsigned epoch) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(&b->flat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = spin_until(&b->flat, epoch * B, &b->timeout);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}
__device__ __forceinline__ bool bar_xcd(Bar *b, unsigned x, unsigned pop, unsigned nx, unsigned epoch) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned a = __hip_atomic_fetch_add(&b->xcc_cnt[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a + 1 == epoch * pop) {   // last of this XCC
            const unsigned t = __hip_atomic_fetch_add(&b->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t + 1 == epoch * nx) {   // last XCC: publish
                for (unsigned k = 0; k < 8; ++k) __hip_atomic_store(&b->xcc_gen[k][0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        ok = spin_until(&b->xcc_gen[x][0], epoch, &b->timeout);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}

template <int MODE>
__global__ void __launch_bounds__(256) k_bar(Bar *b, int K, unsigned *sink, int work) {
    const unsigned B = gridDim.x;
    const unsigned x = xcc_id();
    unsigned pop = 0, nx = 0;
    unsigned epoch = 0;
    if (MODE == 1) {
        // census: who sits on which XCC (dispatch placement is not a contract), then one flat barrier so that everybody reads the result
        if (threadIdx.x == 0) atomicAdd(&b->xcc_pop[x][0], 1u);
        if (!bar_flat(b, B, 1)) return;
        pop = ld(&b->xcc_pop[x][0]);
        for (unsigned k = 0; k < 8; ++k) nx += ld(&b->xcc_pop[k][0]) ? 1u : 0u;
    }
    float acc = (float)threadIdx.x;
    for (int k = 0; k < K; ++k) {
        for (int w = 0; w < work; ++w) acc = acc * 1.0001f + 0.5f;   // a little work between barriers
        ++epoch;
        if (MODE == 0) { if (!bar_flat(b, B, epoch)) return; }
        else { if (!bar_xcd(b, x, pop, nx, epoch)) return; }
    }
    if (acc == 12345.678f) sink[0] = 1;
}
__global__ void __launch_bounds__(256) k_empty(unsigned *sink) { if (threadIdx.x == 999) sink[0] = 1; }

static float run(int mode, int B, int K, int work, Bar *bar, unsigned *sink) {
    CHK(hipMemset(bar, 0, sizeof(Bar)));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0));
    if (mode == 0) hipLaunchKernelGGL(k_bar<0>, dim3(B), dim3(256), 0, 0, bar, K, sink, work);
    else hipLaunchKernelGGL(k_bar<1>, dim3(B), dim3(256), 0, 0, bar, K, sink, work);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    Bar hb; CHK(hipMemcpy(&hb, bar, sizeof(Bar), hipMemcpyDeviceToHost));
    if (hb.timeout) { printf("  TIMEOUT mode %d B %d\n", mode, B); return -1.f; }
    return ms;
}

int main() {
    Bar *bar; unsigned *sink;
    CHK(hipMalloc(&bar, sizeof(Bar))); CHK(hipMalloc(&sink, 64));
    int per_cu = 0;
    CHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_bar<1>, 256, 0));
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    printf("# %s, %d CUs, occupancy query: %d workgroups of 256 per CU\n", prop.gcnArchName, prop.multiProcessorCount, per_cu);
    const int K = 2000;
    printf("# per-barrier cost in us = (t(K = %d) - t(0)) / K; 'work' = dependent FMAs between barriers per thread\n", K);
    printf("%8s %6s %10s %10s\n", "WGs", "work", "flat us", "xcd us");
    const int Bs[] = {104, 256, 416, 512, 832, 1024, 1248};
    for (int work : {0, 2000}) {
        for (int B : Bs) {
            if (B > prop.multiProcessorCount * (per_cu < 6 ? per_cu : 6)) continue;   // (stay well inside what is resident)
            float r[2];
            for (int mode = 0; mode < 2; ++mode) {
                run(mode, B, 10, work, bar, sink);   // warm
                const float t0 = run(mode, B, 0, work, bar, sink), t1 = run(mode, B, K, work, bar, sink);
                r[mode] = (t0 < 0 || t1 < 0) ? -1.f : 1e3f * (t1 - t0) / K;
            }
            float base = 0.f;
            if (work) {   // one workgroup's own loop time
                const float a = run(0, 1, 0, work, bar, sink), c = run(0, 1, K, work, bar, sink);
                base = 1e3f * (c - a) / K;
            }
            printf("%8d %6d %10.2f %10.2f   (loop body + barrier with ONE workgroup: %.2f us)\n", B, work, r[0], r[1], base);
        }
    }
    for (int B : {416, 1248, 4810}) {
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        for (int k = 0; k < 50; ++k) hipLaunchKernelGGL(k_empty, dim3(B), dim3(256), 0, 0, sink);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        for (int k = 0; k < 1000; ++k) hipLaunchKernelGGL(k_empty, dim3(B), dim3(256), 0, 0, sink);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("empty kernel of %d workgroups, 1000 dependent launches on one stream: %.2f us per launch\n", B, ms);
    }
    return 0;
}
