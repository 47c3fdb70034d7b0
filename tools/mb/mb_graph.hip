// mb_graph.hip -- does a hipGraph shorten the boundary between two dependent kernels on this MI355X?  (The solver loops -- 41 CG iterations of
// two launches each in the buckling scene, 27 DFSPH iterations of three -- are chains of short dependent kernels; the host is always far
// ahead of the GPU, so what a graph could save is GPU-side: packet processing between the end of one kernel and the start of the next.)
//
//   hipcc --offload-arch=gfx950 -O3 tools/mb/mb_graph.hip -o tools/mb/mb_graph && tools/mb/mb_graph > profiles/r06_mb_graph.txt
//
// A "CG iteration" here = kernel A (1248 workgroups x 256 threads, one dependent load + store per thread: the size of the three-way split
// A p walk of C5) followed by kernel B (416 workgroups).  ITER iterations are (1) launched one by one on a stream, (2) captured once into a
// graph of 2 x ITER kernel nodes and launched as one graph, (3) the same graph launched REP times back to back.  Time per iteration from HIP
// events around the whole sequence; the kernels' own durations are measured with single launches between events.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_a(const float *__restrict__ in, float *__restrict__ out, int n, int spin) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float v = i < n ? in[i] : 0.0f;
    for (int k = 0; k < spin; ++k) v = v * 1.0001f + 0.5f;   // (a dependent chain: `spin` x 4 cycles of work per wave)
    if (i < n) out[i] = v;
}

int main() {
    const int NA = 1248 * 256, NB = 416 * 256, ITER = 100, REP = 20;
    float *x, *y;
    CHK(hipMalloc(&x, NA * sizeof(float))); CHK(hipMalloc(&y, NA * sizeof(float)));
    CHK(hipMemset(x, 0, NA * sizeof(float))); CHK(hipMemset(y, 0, NA * sizeof(float)));
    hipStream_t st; CHK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int spin : {0, 60, 250, 700}) {   // empty kernels, and kernels of ~1.5, 5 and 15 us
        auto iteration = [&]() {
            hipLaunchKernelGGL(k_a, dim3(NA / 256), dim3(256), 0, st, x, y, NA, spin);
            hipLaunchKernelGGL(k_a, dim3(NB / 256), dim3(256), 0, st, y, x, NB, spin);
        };
        for (int k = 0; k < 10; ++k) iteration();
        CHK(hipStreamSynchronize(st));
        float ms;
        // the kernels by themselves (event to event around ONE launch includes the event packets: an upper bound)
        CHK(hipEventRecord(e0, st)); hipLaunchKernelGGL(k_a, dim3(NA / 256), dim3(256), 0, st, x, y, NA, spin); CHK(hipEventRecord(e1, st));
        CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("spin %4d: one launch of A between events %.2f us\n", spin, ms * 1e3);
        // (1) stream launches
        double best_s = 1e30;
        for (int r = 0; r < 5; ++r) {
            CHK(hipEventRecord(e0, st));
            for (int k = 0; k < ITER * REP; ++k) iteration();
            CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best_s) best_s = ms;
        }
        printf("spin %4d: stream launches          %.3f us per iteration (2 kernels)\n", spin, best_s * 1e3 / (ITER * REP));
        // (2) + (3) a graph of ITER iterations
        hipGraph_t g; hipGraphExec_t ge;
        CHK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int k = 0; k < ITER; ++k) iteration();
        CHK(hipStreamEndCapture(st, &g));
        CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CHK(hipGraphLaunch(ge, st)); CHK(hipStreamSynchronize(st));
        double best_g1 = 1e30, best_g = 1e30;
        for (int r = 0; r < 5; ++r) {
            CHK(hipEventRecord(e0, st)); CHK(hipGraphLaunch(ge, st)); CHK(hipEventRecord(e1, st));
            CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best_g1) best_g1 = ms;
            CHK(hipEventRecord(e0, st));
            for (int k = 0; k < REP; ++k) CHK(hipGraphLaunch(ge, st));
            CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best_g) best_g = ms;
        }
        printf("spin %4d: one graph of %d iterations %.3f us per iteration\n", spin, ITER, best_g1 * 1e3 / ITER);
        printf("spin %4d: %d graph launches          %.3f us per iteration\n", spin, REP, best_g * 1e3 / (ITER * REP));
        CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g));
    }
    return 0;
}
