// sph_device.hpp -- device-side building blocks (compiled twice: SPH_FAST=0 strict, =1 fast).
//
// The central piece is k_nbr_pass<P>: one workgroup owns BLOCK consecutive particles of the
// cell-sorted arrays.  Because cells are linearised z-fastest (reference order,
// base_container.py:473), the 27 neighbour cells of a particle are 9 contiguous particle runs
// (one per (ox, oy) offset, 3 z-cells each), and the union over the workgroup's particles is 9
// contiguous runs as well.  Those 9 runs are staged once into LDS (coalesced float4 loads from
// HBM/L2), then every lane walks its own 9 sub-ranges out of LDS in two phases per <=32
// candidates: phase 1 = distance test -> 32-bit acceptance mask in a VGPR (no LDS writes),
// phase 2 = iterate the set bits and evaluate the pair physics.  Phase 2 therefore runs
// ~max-neighbour-count iterations per wave instead of one iteration per candidate.
// Order of accumulation: runs in (ox outer, oy inner) order, particles ascending -- the same
// order as the serial reference semantics (base_container.py:552-560), so with the stable sort
// (deterministic=1) the strict build reproduces the oracle's summation order.
#pragma once
#include "sph_common.hpp"

#ifndef SPH_FAST
#define SPH_FAST 0
#endif

// ------------------------------------------------------------------ math
__device__ __forceinline__ float fdiv(float a, float b) {
#if SPH_FAST
    return a * __builtin_amdgcn_rcpf(b);
#else
    return a / b;
#endif
}
__device__ __forceinline__ float fsqrt(float a) {
#if SPH_FAST
    return __builtin_amdgcn_sqrtf(a);
#else
    return __builtin_sqrtf(a);
#endif
}

// base_container.py:468 pos_to_index, one axis (IEEE division in both builds: cell assignment
// must agree with the hash kernel), clamped into the grid.
__device__ __forceinline__ int cell_coord(float x, float gs, int n) {
    int c = (int)(x / gs);
    c = c < 0 ? 0 : c;
    c = c > n - 1 ? n - 1 : c;
    return c;
}

// base_solver.py:57 kernel_W.  pow(1-q, 3.0) is evaluated as t*t*t (<= 2 ulp from powf).
__device__ __forceinline__ float kernW(const Consts &c, float r) {
    float res = 0.0f;
    float q = fdiv(r, c.h);
    if (q <= 1.0f) {
        if (q <= 0.5f) {
            float q2 = q * q;
            float q3 = q2 * q;
            res = c.kW * (6.0f * q3 - 6.0f * q2 + 1.0f);
        } else {
            float t = 1.0f - q;
            res = c.kW * 2.0f * (t * t * t);
        }
    }
    return res;
}

// base_solver.py:81 kernel_gradient; R = x_i - x_j, rn = |R|
__device__ __forceinline__ void kernGrad(const Consts &c, float dx, float dy, float dz, float rn,
                                         float &gx, float &gy, float &gz) {
    gx = gy = gz = 0.0f;
    float q = fdiv(rn, c.h);
    if (rn > 1e-5f && q <= 1.0f) {
        float s;
        if (q <= 0.5f) s = c.kG * q * (3.0f * q - 2.0f);
        else { float f = 1.0f - q; s = c.kG * (-f * f); }
#if SPH_FAST
        float inv = __builtin_amdgcn_rcpf(rn * c.h);
        gx = s * (dx * inv); gy = s * (dy * inv); gz = s * (dz * inv);
#else
        float den = rn * c.h;
        gx = s * (dx / den); gy = s * (dy / den); gz = s * (dz / den);
#endif
    }
}

// XCD-aware, bijective workgroup remap: consecutive tiles share neighbour runs, so keep them on
// one XCD's L2 (dispatch places workgroup b on XCD b % 8).
__device__ __forceinline__ int xcd_remap(int b, int nb) {
    int xcd = b & 7, q = nb >> 3, r = nb & 7;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (b >> 3);
}

__device__ __forceinline__ int wave_incl_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// ------------------------------------------------------------------ grid build
// base_container.py:496 init_grid: cell id + histogram.  The atomic's return value is the
// particle's arrival rank inside its cell, which replaces the second atomic pass of :515.
__global__ void __launch_bounds__(256)
k_hash_count(const Consts c, const float4 *__restrict__ posv, int *__restrict__ cellid,
             int *__restrict__ rank, int *__restrict__ cell_count) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= c.n) return;
    float4 p = posv[i];
    int cx = cell_coord(p.x, c.grid_size, c.nx);
    int cy = cell_coord(p.y, c.grid_size, c.ny);
    int cz = cell_coord(p.z, c.grid_size, c.nz);
    int lin = (cx * c.ny + cy) * c.nz + cz;
    cellid[i] = lin;
    rank[i] = atomicAdd(&cell_count[lin], 1);
}

// base_container.py:546 PrefixSumExecutor.run -- here an exclusive scan into cell_start[0..G],
// three launches: tile sums, scan of tile sums, tile rescans.
#define SCAN_TPB 256
#define SCAN_IPT 8
#define SCAN_TILE (SCAN_TPB * SCAN_IPT)

__device__ __forceinline__ int block_excl_scan_256(int v, int *s_w, int &total) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int inc = wave_incl_scan(v);
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < SCAN_TPB / 64; ++k) {
        int t = s_w[k];
        if (k < w) off += t;
        tot += t;
    }
    total = tot;
    __syncthreads();
    return off + inc - v;
}

__global__ void __launch_bounds__(SCAN_TPB)
k_scan_reduce(const int *__restrict__ in, int n, int *__restrict__ partial) {
    __shared__ int s_w[SCAN_TPB / 64];
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_IPT;
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_IPT; ++k) { int idx = base + k; s += idx < n ? in[idx] : 0; }
    int tot;
    block_excl_scan_256(s, s_w, tot);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(SCAN_TPB)
k_scan_partials(int *__restrict__ partial, int nb) {
    __shared__ int s_w[SCAN_TPB / 64];
    int carry = 0;
    for (int b0 = 0; b0 < nb; b0 += SCAN_TPB) {
        int idx = b0 + threadIdx.x;
        int v = idx < nb ? partial[idx] : 0;
        int tot;
        int ex = block_excl_scan_256(v, s_w, tot);
        if (idx < nb) partial[idx] = carry + ex;
        carry += tot;
    }
}

__global__ void __launch_bounds__(SCAN_TPB)
k_scan_final(const int *__restrict__ in, int n, const int *__restrict__ partial, int *__restrict__ out,
             int total_particles) {
    __shared__ int s_w[SCAN_TPB / 64];
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_IPT;
    int v[SCAN_IPT];
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_IPT; ++k) { int idx = base + k; v[k] = idx < n ? in[idx] : 0; s += v[k]; }
    int tot;
    int ex = block_excl_scan_256(s, s_w, tot) + partial[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_IPT; ++k) {
        int idx = base + k;
        if (idx < n) out[idx] = ex;
        ex += v[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = total_particles;
}

// deterministic mode: list the source indices of every cell, so that the scatter can compute a
// stable rank (= serial execution of base_container.py:510-515).
__global__ void __launch_bounds__(256)
k_scatter_index(int n, const int *__restrict__ cellid, const int *__restrict__ rank,
                const int *__restrict__ cell_start, int *__restrict__ tmp_idx) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    tmp_idx[cell_start[cellid[i]] + rank[i]] = i;
}

struct SortArrays {
    const float4 *posv_in, *velm_in, *orig_in;
    const int *meta_in, *pid_in;
    const unsigned *color_in;
    const float *rho_in;
    float4 *posv_out, *velm_out, *orig_out;
    int *meta_out, *pid_out;
    unsigned *color_out;
    float *rho_out;
};

// base_container.py:506 reorder_particles as one gather/scatter (no copy-back pass: the arrays
// are double-buffered).
template <bool STABLE>
__global__ void __launch_bounds__(256)
k_scatter(int n, const int *__restrict__ cellid, const int *__restrict__ rank,
          const int *__restrict__ cell_start, const int *__restrict__ tmp_idx, SortArrays a) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int cell = cellid[i];
    int s = cell_start[cell];
    int r;
    if (STABLE) {
        int e = cell_start[cell + 1];
        r = 0;
        for (int k = s; k < e; ++k) r += tmp_idx[k] < i ? 1 : 0;
    } else {
        r = rank[i];
    }
    int d = s + r;
    a.posv_out[d] = a.posv_in[i];
    a.velm_out[d] = a.velm_in[i];
    a.meta_out[d] = a.meta_in[i];
    a.pid_out[d] = a.pid_in[i];
    a.color_out[d] = a.color_in[i];
    a.rho_out[d] = a.rho_in[i];
    if (a.orig_in) a.orig_out[d] = a.orig_in[i];
}

// ------------------------------------------------------------------ generic neighbour pass
template <bool LDS, class P>
__device__ __forceinline__ void nbr_loop(const Consts &c, const int *__restrict__ cell_start, const P &p,
                                         typename P::Own &own, int i, float xi, float yi, float zi,
                                         int cx, int cy, int cz, const float4 *sA,
                                         const typename P::BT *sB, const int *s_rs, const int *s_base,
                                         unsigned &npairs) {
    const int z0 = cz > 0 ? cz - 1 : 0;
    const int z1 = cz < c.nz - 1 ? cz + 1 : c.nz - 1;
    for (int k = 0; k < 9; ++k) {
        const int xx = cx + k / 3 - 1, yy = cy + k % 3 - 1;
        if (xx < 0 || xx >= c.nx || yy < 0 || yy >= c.ny) continue;
        const int lin0 = (xx * c.ny + yy) * c.nz + z0;
        const int js = cell_start[lin0];
        const int je = cell_start[lin0 + (z1 - z0) + 1];
        const int loff = LDS ? (s_base[k] - s_rs[k]) : 0;
        for (int j0 = js; j0 < je; j0 += 32) {
            const int m = (je - j0) < 32 ? (je - j0) : 32;
            unsigned mask = 0;
#pragma unroll 4
            for (int t = 0; t < m; ++t) {
                const int j = j0 + t;
                const float4 a = LDS ? sA[j + loff] : p.loadA(j);
                const float dx = xi - a.x, dy = yi - a.y, dz = zi - a.z;
                const float r2 = dx * dx + dy * dy + dz * dz;
                const unsigned ok = (r2 < c.h2 && j != i) ? 1u : 0u;
                mask |= ok << t;
            }
            npairs += __popc(mask);
            while (mask) {
                const int t = __ffs(mask) - 1;
                mask &= mask - 1;
                const int j = j0 + t;
                const float4 a = LDS ? sA[j + loff] : p.loadA(j);
                const float dx = xi - a.x, dy = yi - a.y, dz = zi - a.z;
                const float r2 = dx * dx + dy * dy + dz * dz;
                typename P::BT bj = typename P::BT();
                if (P::HAS_B) bj = LDS ? sB[j + loff] : p.loadB(j);
                p.pair(c, own, dx, dy, dz, r2, a, bj, j);
            }
        }
    }
}

template <class P>
__global__ void __launch_bounds__(P::BLOCK)
k_nbr_pass(const Consts c, const int *__restrict__ cell_start, const P p, DevScalars *__restrict__ scal,
           int nblocks) {
    constexpr int BLOCK = P::BLOCK;
    constexpr int CAP = P::CAP;
    __shared__ float4 sA[CAP];
    __shared__ typename P::BT sB[P::HAS_B ? CAP : 1];
    __shared__ int s_rs[9], s_len[9], s_base[10], s_c[2], s_any;

    const int tid = threadIdx.x;
    const int b = xcd_remap(blockIdx.x, nblocks);
    const int i0 = b * BLOCK;
    const int i = i0 + tid;
    const bool valid = i < c.n;
    const int nvalid = (c.n - i0) < BLOCK ? (c.n - i0) : BLOCK;

    float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
    typename P::Own own;
    bool active = false;
    int cx = 0, cy = 0, cz = 0;
    if (tid == 0) s_any = 0;
    __syncthreads();
    if (valid) {
        pi = p.posv[i];
        cx = cell_coord(pi.x, c.grid_size, c.nx);
        cy = cell_coord(pi.y, c.grid_size, c.ny);
        cz = cell_coord(pi.z, c.grid_size, c.nz);
        const int lin = (cx * c.ny + cy) * c.nz + cz;
        if (tid == 0) s_c[0] = lin;
        if (tid == nvalid - 1) s_c[1] = lin;
        active = p.begin(c, i, pi, own);
        if (active) s_any = 1;
    }
    __syncthreads();
    if (s_any) {  // workgroup-uniform
        if (tid < 9) {
            const int shift = (tid / 3 - 1) * c.ny * c.nz + (tid % 3 - 1) * c.nz;
            int lo = s_c[0] + shift - 1, hi = s_c[1] + shift + 1;
            int rs = 0, re = 0;
            if (hi >= 0 && lo <= c.G - 1) {
                lo = lo < 0 ? 0 : lo;
                hi = hi > c.G - 1 ? c.G - 1 : hi;
                rs = cell_start[lo];
                re = cell_start[hi + 1];
            }
            s_rs[tid] = rs;
            s_len[tid] = re - rs;
        }
        __syncthreads();
        if (tid == 0) {
            int acc = 0;
            for (int k = 0; k < 9; ++k) { s_base[k] = acc; acc += s_len[k]; }
            s_base[9] = acc;
        }
        __syncthreads();
        const int total = s_base[9];
        const bool use_lds = (total <= CAP) && !c.force_global;
        unsigned npairs = 0;
        if (use_lds) {
            int base[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) base[k] = s_base[k];
            for (int t = tid; t < total; t += BLOCK) {
                int k = 0;
#pragma unroll
                for (int q = 1; q < 9; ++q) k += t >= base[q] ? 1 : 0;
                const int j = s_rs[k] + (t - s_base[k]);
                typename P::BT bj = typename P::BT();
                sA[t] = p.stage(c, j, bj);
                if (P::HAS_B) sB[t] = bj;
            }
            __syncthreads();
            if (active)
                nbr_loop<true>(c, cell_start, p, own, i, pi.x, pi.y, pi.z, cx, cy, cz, sA, sB, s_rs, s_base, npairs);
        } else {
            if (tid == 0) atomicAdd(&scal->fallback, 1ull);
            if (active)
                nbr_loop<false>(c, cell_start, p, own, i, pi.x, pi.y, pi.z, cx, cy, cz, sA, sB, s_rs, s_base, npairs);
        }
        if (P::COUNT_PAIRS) {
            float fp = wave_sum((float)npairs);  // <= 64 * few hundred: exact in f32
            if ((tid & 63) == 0 && fp > 0.0f) atomicAdd(&scal->pairs, (unsigned long long)fp * P::PAIR_WEIGHT);
        }
    }
    float red = 0.0f;
    if (valid) {
        if (active) red = p.finish(c, i, pi, own);
        else p.passive(c, i, pi);
    }
    if constexpr (P::HAS_REDUCE) {
        // deterministic per-workgroup partial sum (fixed tree), finished by k_reduce_partials
        __shared__ float s_red[BLOCK / 64];
        const float w = wave_sum(red);
        if ((tid & 63) == 0) s_red[tid >> 6] = w;
        __syncthreads();
        if (tid == 0) {
            float t = 0.0f;
#pragma unroll
            for (int k = 0; k < BLOCK / 64; ++k) t += s_red[k];
            p.red_out[b] = t;
        }
    }
}

// sums partial[0..n) in a fixed order into scal->red[slot]
__global__ void __launch_bounds__(256)
k_reduce_partials(const float *__restrict__ partial, int n, DevScalars *__restrict__ scal, int slot) {
    __shared__ float s_w[4];
    float t = 0.0f;
    for (int k = threadIdx.x; k < n; k += 256) t += partial[k];
    t = wave_sum(t);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) scal->red[slot] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}
