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
#include <limits.h>

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
// (a / b) / c: the strict build keeps the reference's two divisions; the fast build spends one reciprocal
// (v_rcp_f32 issues at a quarter of the plain VALU rate) on b * c
__device__ __forceinline__ float fdiv2(float a, float b, float c) {
#if SPH_FAST
    return a * __builtin_amdgcn_rcpf(b * c);
#else
    return (a / b) / c;
#endif
}
__device__ __forceinline__ float fsqrt(float a) {
#if SPH_FAST
    return __builtin_amdgcn_sqrtf(a);
#else
    return __builtin_sqrtf(a);
#endif
}

// the scene's y of a library-frame position (emitter threshold g_upper)
__device__ __forceinline__ float up_coord(const Consts &c, const float4 &p) { return c.up_axis == 1 ? p.y : (c.up_axis == 0 ? p.x : p.z); }

// base_container.py:468 pos_to_index, one axis (IEEE division in both builds: cell assignment
// must agree with the hash kernel), clamped into the grid.
__device__ __forceinline__ int cell_coord(float x, float gs, int n) {
    int c = (int)(x / gs);
    c = c < 0 ? 0 : c;
    c = c > n - 1 ? n - 1 : c;
    return c;
}

// The slab axis (x or z, Consts::slab_axis): a slab-sharded rank builds its cell lists on its own layers only (local layer = global
// layer - offset); a particle outside them (it is about to be dropped) is clamped onto the nearest local layer.  Unsharded, and for the
// axis that is not cut: offset 0, n = n_glob.
__device__ __forceinline__ int cell_coord_x(const Consts &c, float x) {
    int cx = cell_coord(x, c.grid_size, c.nx_glob) - c.cx_off;
    cx = cx < 0 ? 0 : cx;
    return cx > c.nx - 1 ? c.nx - 1 : cx;
}
__device__ __forceinline__ int cell_coord_z(const Consts &c, float z) {
    int cz = cell_coord(z, c.grid_size, c.nz_glob) - c.cz_off;
    cz = cz < 0 ? 0 : cz;
    return cz > c.nz - 1 ? c.nz - 1 : cz;
}
// global cell layer of a position along the slab axis (compared with the slab bounds z_lo / z_hi)
__device__ __forceinline__ int slab_layer(const Consts &c, const float4 &p) {
    return c.slab_axis == 0 ? cell_coord(p.x, c.grid_size, c.nx_glob) : cell_coord(p.z, c.grid_size, c.nz_glob);
}

// Per-pair geometry shared by kernel_W / kernel_gradient.  Strict build: rn = sqrt(r2), q = rn / h (IEEE).
// Fast build: one v_rsq_f32 gives 1/rn; rn = r2 * (1/rn), q = rn * (1/h), 1/(rn h) = (1/rn)(1/h).
struct Geom { float rn, q, inv_rnh, rinv; };
__device__ __forceinline__ Geom geom(const Consts &c, float r2) {
    Geom g;
#if SPH_FAST
    const float rinv = __builtin_amdgcn_rsqf(fmaxf(r2, 1e-30f));
    g.rn = r2 * rinv;
    g.q = g.rn * c.inv_h;
    g.inv_rnh = rinv * c.inv_h;
    g.rinv = rinv;
#else
    g.rn = __builtin_sqrtf(r2);
    g.q = g.rn / c.h;
    g.inv_rnh = 0.0f;  // unused
    g.rinv = 0.0f;
#endif
    return g;
}

// base_solver.py:57 kernel_W.  pow(1-q, 3.0) is evaluated as t*t*t (<= 2 ulp from powf).
template <bool ACCEPTED = true>
__device__ __forceinline__ float kernW(const Consts &c, const Geom &g) {
    float res = 0.0f;
    const float q = g.q;
#if SPH_FAST
    // branch-free (accepted pairs have q < 1): both pieces evaluated, one v_cndmask
    const float t = 1.0f - q;
    const float lo = 1.0f - 6.0f * (q * q) * t;   // = 6q^3 - 6q^2 + 1
    const float hi = 2.0f * (t * t * t);
    res = c.kW * (q <= 0.5f ? lo : hi);
    // ACCEPTED: g belongs to a pair that passed r2 < h2, so q exceeds 1 by rounding at most, where hi = 2 (1 - q)^3 is ~1e-21:
    // no test.  (PCISPH's rho* evaluates W at PREDICTED distances, which can exceed h: it keeps the test.)
    return (ACCEPTED || q <= 1.0f) ? res : 0.0f;
#endif
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

// fast build: grad W_ij = kernGradScale * (x_i - x_j) -- for the functors that fold the scalar into their pair coefficients instead of
// carrying the gradient as a vector (three multiplies and three registers less per pair)
__device__ __forceinline__ float kernGradScale(const Consts &c, const Geom &g) {
    const float q = g.q, f = 1.0f - q;
    const float s = (q <= 0.5f ? q * (3.0f * q - 2.0f) : -f * f) * g.rinv;
    return g.rn > 1e-5f ? s * c.kGh : 0.0f;   // kG / (rn h); (no q <= 1 test, see kernW)
}
// kernW / kW of an accepted pair (fast build): for the functors that fold kW into a per-particle coefficient
__device__ __forceinline__ float kernWpoly(const Geom &g) {
    const float q = g.q, t = 1.0f - q;
    return q <= 0.5f ? 1.0f - 6.0f * (q * q) * t : 2.0f * (t * t * t);
}

// base_solver.py:81 kernel_gradient; R = x_i - x_j
__device__ __forceinline__ void kernGrad(const Consts &c, float dx, float dy, float dz, const Geom &g,
                                         float &gx, float &gy, float &gz) {
    gx = gy = gz = 0.0f;
    const float q = g.q;
#if SPH_FAST
    {
        const float f = 1.0f - q;
        float s = c.kG * (q <= 0.5f ? q * (3.0f * q - 2.0f) : -f * f);
        s = g.rn > 1e-5f ? s * g.inv_rnh : 0.0f;   // (no q <= 1 test, see kernW)
        gx = s * dx; gy = s * dy; gz = s * dz;
        return;
    }
#endif
    if (g.rn > 1e-5f && q <= 1.0f) {
        float s;
        if (q <= 0.5f) s = c.kG * q * (3.0f * q - 2.0f);
        else { float f = 1.0f - q; s = c.kG * (-f * f); }
#if SPH_FAST
        const float si = s * g.inv_rnh;
        gx = si * dx; gy = si * dy; gz = si * dz;
#else
        float den = g.rn * c.h;
        gx = s * (dx / den); gy = s * (dy / den); gz = s * (dz / den);
#endif
    }
}

// element j of a uniform base pointer through a 32-bit byte offset: lets the backend use the SGPR-base + VGPR-offset
// form of global_load (one address VGPR instead of two per load)
template <class T> __device__ __forceinline__ T ldg_idx(const T *base, int j) {
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + (unsigned)((unsigned)j * (unsigned)sizeof(T)));
}

// XCD-aware, bijective workgroup remap: consecutive tiles share neighbour runs, so keep them on
// one XCD's L2 (dispatch places workgroup b on XCD b % 8).
// chunk = 0: every XCD walks ONE contiguous eighth of the tiles (most reuse of candidate runs in its L2; no balance between XCDs: the
// hardware deals workgroups round-robin, so an XCD whose eighth is cheap idles).  chunk = C > 0: the tiles are dealt in chunks of C
// consecutive tiles, XCD k takes chunks k, k + 8, ... -- neighbouring tiles still share an L2, and every XCD sees every part of the scene
// (Consts::xcd_chunk).  Bijective on [0, nb): the ragged end (fewer than 8 C tiles) keeps the dispatch order.
__device__ __forceinline__ int xcd_remap(int b, int nb, int chunk = 0) {
#ifdef SPH_NO_XCD_REMAP
    return b;
#endif
    if (chunk > 0) {
        const int full = nb / (8 * chunk) * (8 * chunk);
        if (b >= full) return b;
        const int j = b >> 3, m = j / chunk;
        return (m * 8 + (b & 7)) * chunk + (j - m * chunk);
    }
    int xcd = b & 7, q = nb >> 3, r = nb & 7;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (b >> 3);
}

// Slot of this workgroup's partial sum (P::HAS_REDUCE passes and the per-particle CG kernels): its tile index, or -- in a launch over the
// list of fluid-holding tiles -- its position in the list, so that the consumers add up the first *blk_count slots without reading the list.
__device__ __forceinline__ int red_slot(const int *blk_list, int b) { return blk_list ? (int)blockIdx.x : b; }

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

// Runs of equal key among the consecutive lanes of a wave (the input of the sort is last step's sorted order, so a wave is a few
// runs of ~8 particles that stay in their cell, plus strays): head = first lane of a run, hl = the head's lane, len = run length
// (meaningful in the head lane).  k_hash_count, k_scatter_index and k_scatter<true> must see the same runs: same thread -> particle map.
// `slot`: the slot the particle got inside its cell (k_scatter_index / k_scatter): a run must also be CONTIGUOUS there.  Particles that
// took their slots one by one (arrivals appended by k_halo_unpack2 under slab sharding: one atomic each) would otherwise be merged into a
// "run" whose slots lie anywhere in the cell.
__device__ __forceinline__ void wave_runs(int key, int lane, bool &head, int &hl, int &len, int slot = 0, bool use_slot = false) {
    const int prev = __shfl_up(key, 1, 64);
    const int prev_slot = __shfl_up(slot, 1, 64);
    head = lane == 0 || key != prev || (use_slot && slot != prev_slot + 1);
    const unsigned long long hm = __ballot(head);
    const unsigned long long upto = hm & ((2ull << lane) - 1ull);          // heads at or below this lane
    hl = 63 - __clzll(upto);
    const unsigned long long above = lane == 63 ? 0ull : (hm >> (lane + 1));
    len = above ? __ffsll(above) : 64 - lane;
}

// Tile sums of the scan (State::scan_partial): every valid lane's particle counts once into the tile of SCAN_TILE cells its cell lies
// in.  One atomic per wave and DISTINCT tile among its lanes (one at rest: the lanes are consecutive particles of last step's sorted
// order; a handful where the fluid moves: a step in x is 5 tiles away), and the counters sit SCAN_PARTIAL_STRIDE ints apart: atomics on
// one 128-byte line are served one at a time (lesson 1 of round 4) -- the first version of this fold (one atomic per run, counters packed
// 32 to a line) cost the force pass 88 us in motion (profiles/r06_scanfold_ab.txt).
#define SCAN_TILE_SHIFT 11
#define SCAN_PARTIAL_STRIDE 8
// The sum of the LAST tile is never read (a tile needs the sums of the tiles before it): particles of its cells are not added.  That is
// where the graveyard cells of a slab-sharded rank sit (G .. G + SPH_NGRAVE - 1): every wave of the classifying kernels holds a few dead
// lanes -- last step's ghosts sit at both ends of every z column -- and their atomics would all land on that ONE counter (measured:
// +17 % on a two-rank step, profiles/r06_two_ranks_scanfold_ab.txt; lesson 1 a third time).
__device__ __forceinline__ void tile_sum_add(int *tile_sum, int lin, bool valid, int cells_total) {
    const int lane = threadIdx.x & 63;
    const int t = lin >> SCAN_TILE_SHIFT;
    valid = valid && t != ((cells_total - 1) >> SCAN_TILE_SHIFT);
    unsigned long long rem = __ballot(valid);
    while (rem) {   // wave-uniform
        const int src = __ffsll((long long)rem) - 1;
        const int t0 = __shfl(t, src, 64);
        const unsigned long long same = __ballot(valid && t == t0);
        if (lane == src) atomicAdd(&tile_sum[t0 * SCAN_PARTIAL_STRIDE], __popcll(same));
        rem &= ~same;
    }
}

// Links a run (first particle i_first, `len` particles of cell `lin`) into its cell's list: RunList, sph_common.hpp.  Called by the run's
// head lane right next to the run's histogram atomic.
__device__ __forceinline__ void run_list_file(const RunList &rl, int lin, int i_first, int len, int base) {
    int prev = i_first, at = rl.first_off + lin;   // first arrival of its cell: (first particle, length) into first[cell], no atomic (RunList)
    if (base != 0) {
        const unsigned long long mine = ((unsigned long long)rl.epoch << 32) | (unsigned long long)(unsigned)i_first;
        const unsigned long long old = atomicExch(&rl.head[lin], mine);
        prev = (unsigned)(old >> 32) == rl.epoch ? (int)(unsigned)(old & 0xffffffffull) : -1;   // a head of another sort: empty list
        at = i_first;
    }
    rl.rec[at] = make_int2(prev, len);
}

// ------------------------------------------------------------------ grid build
// base_container.py:496 init_grid: cell id + histogram.  The atomic's return value is the
// particle's arrival rank inside its cell, which replaces the second atomic pass of :515.
__global__ void __launch_bounds__(256)
k_hash_count(const Consts c, const float4 *__restrict__ posv, int *__restrict__ cellid,
             int *__restrict__ rank, int *__restrict__ cell_count, const int *__restrict__ meta_dead, int *__restrict__ tile_sum, const RunList rl) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = i < live_n(c);
    int lin = -1 - lane;  // distinct dummy key for lanes past the end
    if (valid) {
        const float4 p = posv[i];
        const int cx = cell_coord_x(c, p.x);
        const int cy = cell_coord(p.y, c.grid_size, c.ny);
        const int cz = cell_coord_z(c, p.z);
        lin = (cx * c.ny + cy) * c.nz + cz;
        if (meta_dead && META_DEAD(meta_dead[i])) lin = c.G + ((i >> 6) & (SPH_NGRAVE - 1));   // slab sharding: a graveyard cell behind the grid
        cellid[i] = lin;
    }
    // The input is the previous step's sorted order, so lanes of one wave fall into a few runs of equal
    // cell id.  One atomic per run (by its first lane) instead of one per particle: ~8x fewer L2 atomics.
    bool head; int hl, len;
    wave_runs(lin, lane, head, hl, len);
    int base = 0;
    if (head && valid) base = atomicAdd(&cell_count[lin], len);
    const int base_run = __shfl(base, hl, 64);   // (the wait for the atomic's answer sits here, in front of the record's store -- vmcnt counts stores too)
    if (rl.head && head && valid) run_list_file(rl, lin, i, len, base);
    if (valid) rank[i] = base_run + (lane - hl);
    if (tile_sum) tile_sum_add(tile_sum, lin, valid, c.G + (meta_dead ? SPH_NGRAVE : 0));
}

// base_container.py:546 PrefixSumExecutor.run -- here an exclusive scan into cell_start[0..G],
// two launches: tile sums, then tile rescans (each adds up the tile sums before it).
#define SCAN_TPB 256
#define SCAN_IPT 8
#define SCAN_TILE (SCAN_TPB * SCAN_IPT)
static_assert(SCAN_TILE == 1 << SCAN_TILE_SHIFT, "tile_sum_add and the scan kernels must agree on the tile");

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

// Tile layout of both scan kernels: a tile is SCAN_TILE = 2048 consecutive cells = two chunks of 1024; thread t owns
// cells [4t, 4t + 4) of either chunk, so that every load / store instruction of a wave is one contiguous 1 KiB
// (16 B per lane) -- the per-thread-contiguous layout (8 consecutive ints per thread, 32 B lane stride) this replaces
// moved the same bytes with 8 strided dword instructions per thread and ran at a third of the copy rate.
__device__ __forceinline__ int4 scan_load4(const int *__restrict__ in, int idx, int n) {
    int4 v = make_int4(0, 0, 0, 0);
    if (idx + 3 < n) v = *reinterpret_cast<const int4 *>(in + idx);
    else { if (idx < n) v.x = in[idx]; if (idx + 1 < n) v.y = in[idx + 1]; if (idx + 2 < n) v.z = in[idx + 2]; }
    return v;
}

__global__ void __launch_bounds__(SCAN_TPB)
k_scan_reduce(const int *__restrict__ in, int n, int *__restrict__ partial) {
    __shared__ int s_w[SCAN_TPB / 64];
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * 4;
    const int4 a = scan_load4(in, base, n), b = scan_load4(in, base + SCAN_TILE / 2, n);
    int s = (a.x + a.y) + (a.z + a.w) + (b.x + b.y) + (b.z + b.w);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x * SCAN_PARTIAL_STRIDE] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

__global__ void __launch_bounds__(SCAN_TPB)
k_scan_final(int *__restrict__ in, int n, const int *__restrict__ partial, int *__restrict__ out,
             int total_particles, DevScalars *__restrict__ scal, int clear_bank, const int *__restrict__ total_dev, int *__restrict__ partial_next,
             int *__restrict__ tile_state) {
    // side jobs of the kernel that runs every step: clears cell_count behind itself (the next histogram starts from
    // zero without a memset) and clears the statistics bank of the next step
    __shared__ int s_w[SCAN_TPB / 64];
    if (partial_next && threadIdx.x == 0) partial_next[blockIdx.x * SCAN_PARTIAL_STRIDE] = 0;   // the OTHER bank of tile sums: nobody reads it in this launch; the next histogram adds to it
    if (blockIdx.x * SCAN_TPB + threadIdx.x < SPH_STAT_SLOTS) {   // first SPH_STAT_SLOTS / SCAN_TPB workgroups (the grid is never smaller)
        scal->pairs[clear_bank][blockIdx.x * SCAN_TPB + threadIdx.x] = 0ull;
        scal->evals[clear_bank][blockIdx.x * SCAN_TPB + threadIdx.x] = 0ull;
        scal->fallback[clear_bank][blockIdx.x * SCAN_TPB + threadIdx.x] = 0ull;
    }
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * 4;
    const int4 a = scan_load4(in, base, n), b = scan_load4(in, base + SCAN_TILE / 2, n);
    // offset of this tile = sum of the tile sums before it (every workgroup adds them up itself: a few thousand ints
    // out of L2 instead of a third launch)
    int before = 0;
    for (int k = threadIdx.x; k < (int)blockIdx.x; k += SCAN_TPB) before += partial[k * SCAN_PARTIAL_STRIDE];
    int tot0, tot1, btot;
    // EMPTY tiles (most of them: the fluid fills a corner of the domain -- 930 of C2's 1041 tiles hold no particle) need no histogram and no
    // zeroing, only cell_start = (particles before the tile) in all their cells; and not even that when the tile was empty with the same
    // count before it at the last scan (tile_state[t] = that count + 1, 0 = cells hold a real scan).  The tile's own sum comes from whoever
    // took the histogram (or k_scan_reduce); the last tile's is never counted (tile_sum_add), block 0 also files cell_start[n].  The loads
    // above are issued regardless: skipping them would put a dependent round trip in front of every tile that does hold particles.
    const bool maybe_empty = tile_state && blockIdx.x > 0 && (int)blockIdx.x < ((n - 1) >> SCAN_TILE_SHIFT);
    const int own_sum = maybe_empty ? partial[blockIdx.x * SCAN_PARTIAL_STRIDE] : 1;
    const int was = maybe_empty ? tile_state[blockIdx.x] : 0;
    block_excl_scan_256(before, s_w, btot);
    if (own_sum == 0) {   // (workgroup-uniform)
        if (was == btot + 1) return;
        const int4 o = make_int4(btot, btot, btot, btot);
        *reinterpret_cast<int4 *>(out + base) = o;                   // (not the last tile: every index is inside the grid)
        *reinterpret_cast<int4 *>(out + base + SCAN_TILE / 2) = o;
        if (threadIdx.x == 0) tile_state[blockIdx.x] = btot + 1;
        return;
    }
    if (tile_state && threadIdx.x == 0 && (int)blockIdx.x <= ((n - 1) >> SCAN_TILE_SHIFT)) tile_state[blockIdx.x] = 0;
    const int sa = (a.x + a.y) + (a.z + a.w), sb = (b.x + b.y) + (b.z + b.w);
    int ea = block_excl_scan_256(sa, s_w, tot0) + btot;
    int eb = block_excl_scan_256(sb, s_w, tot1) + btot + tot0;
    const int4 oa = make_int4(ea, ea + a.x, ea + a.x + a.y, ea + a.x + a.y + a.z);
    const int4 ob = make_int4(eb, eb + b.x, eb + b.x + b.y, eb + b.x + b.y + b.z);
    const int4 zero = make_int4(0, 0, 0, 0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int idx = base + h * (SCAN_TILE / 2);
        const int4 o = h ? ob : oa;
        if (idx + 3 < n) { *reinterpret_cast<int4 *>(out + idx) = o; *reinterpret_cast<int4 *>(in + idx) = zero; }
        else {
            if (idx < n) { out[idx] = o.x; in[idx] = 0; }
            if (idx + 1 < n) { out[idx + 1] = o.y; in[idx + 1] = 0; }
            if (idx + 2 < n) { out[idx + 2] = o.z; in[idx + 2] = 0; }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = total_dev ? *total_dev : total_particles;
}

// deterministic mode, first half: one record per RUN (first source index, length), filed at the slot the run's first particle got
// from the histogram atomics.  The slots [cell_start[c], cell_start[c + 1]) of a cell are partitioned by its runs (one atomicAdd of
// `len` per run, k_hash_count), so a walk from cell_start[c] that jumps by the recorded lengths visits exactly the run records;
// the slots inside a run are never read.  (Until round 5 this kernel listed every particle's source index and the scatter counted,
// per particle, the entries of its cell below its own index: one dependent L2 round trip per cell-mate -- 8 at rest, 16-24 where the
// fluid has piled up: the scatter went from 23.5 to 49 us in motion, profiles/r04_c2_in_motion_per_kernel.txt.)
__global__ void __launch_bounds__(256)
k_scatter_index(int n, const int *__restrict__ cellid, const int *__restrict__ rank,
                const int *__restrict__ cell_start, int2 *__restrict__ runs, const int *__restrict__ n_dev) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    if (n_dev) n = *n_dev;
    const bool valid = i < n;
    const int cell = valid ? cellid[i] : -1 - lane;
    const int slot = valid ? rank[i] : 0;
    bool head; int hl, len;
    wave_runs(cell, lane, head, hl, len, slot, true);
    if (valid && head) runs[cell_start[cell] + slot] = make_int2(i, len);
}

struct SortArrays {
    int G;   // number of grid cells (cell id G = graveyard of the slab sharding)
    const float4 *posv_in, *velm_in, *orig_in;
    const int *meta_in, *pid_in;
    const unsigned *color_in;
    const float *rho_in;
    const int *xidx_in; int *xidx_out;   // slab sharding only (else null)
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
          const int *__restrict__ cell_start, const int2 *__restrict__ runs, SortArrays a, const int *__restrict__ n_dev) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    if (n_dev) n = *n_dev;
    const bool valid = i < n;
    const int cell = valid ? cellid[i] : -1 - lane;
    int r = 0, s = 0;
    if (STABLE) {
        // stable rank = serial execution of base_container.py:510-515 = number of particles of the same cell with a lower source index
        //             = (lengths of the cell's runs that start below this particle's run) + (position inside its own run)
        bool head; int hl, len;
        const int slot = valid ? rank[i] : 0;
        wave_runs(cell, lane, head, hl, len, slot, true);
        int r0 = 0;
        if (valid) s = cell_start[cell];
        if (valid && head && cell < a.G) {   // (a graveyard cell of the slab sharding may hold 1e5 particles nobody looks at again)
            const int e = cell_start[cell + 1];
            if (e - s != len) {              // other runs share the cell (at rest: only where a cell straddles two waves)
                for (int p = s; p < e;) {
                    const int2 rec = runs[p];
                    r0 += rec.x < i ? rec.y : 0;
                    p += rec.y > 0 ? rec.y : 1;
                }
            }
        }
        r0 = __shfl(r0, hl, 64);
        r = r0 + (lane - hl);
        if (valid && cell >= a.G) r = slot;
    } else if (valid) {
        s = cell_start[cell];
        r = rank[i];
    }
    if (!valid) return;
    const int d = s + r;
    a.posv_out[d] = a.posv_in[i];
    a.velm_out[d] = a.velm_in[i];
    a.meta_out[d] = a.meta_in[i];
    a.pid_out[d] = a.pid_in[i];
    a.color_out[d] = a.color_in[i];
    a.rho_out[d] = a.rho_in[i];
    if (a.orig_in) a.orig_out[d] = a.orig_in[i];
    if (a.xidx_in) a.xidx_out[d] = a.xidx_in[i];
}

// Deterministic sort by run lists, first half (replaces k_scatter_index AND the rank walk of k_scatter<true>): the destination of every
// particle = cell_start[cell] + stable rank, where the stable rank (serial execution of base_container.py:510-515) = (lengths of the
// cell's runs whose first particle has a lower source index) + (position inside its own run).  The cell's runs hang on the list the hashers
// filed (run_list_file); a run that IS its cell (the usual case) has nothing to walk.  Leaves the INVERSE map inv[dest] = source: the second
// half (k_gather_prep) is organised by destination tile, so that it can prepare the tile for the neighbour passes while it moves it.
__global__ void __launch_bounds__(256)
k_sort_rank(int n, const int *__restrict__ cellid, const int *__restrict__ cell_start, const RunList rl, int *__restrict__ inv) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = i < n;
    const int cell = valid ? cellid[i] : -1 - lane;
    bool head; int hl, len;
    wave_runs(cell, lane, head, hl, len);   // the same runs the hasher saw: same thread -> particle map
    int d0 = 0;
    if (valid && head) {
        const int s = cell_start[cell], e = cell_start[cell + 1];
        int r0 = 0;
        if (e - s != len) {   // other runs share the cell
            const int2 f = rl.rec[rl.first_off + cell];                            // the cell's first arrival (of this sort: the cell holds several runs)
            r0 = f.x < i ? f.y : 0;
            const unsigned long long hd = rl.head[cell];              // ... and the later ones, newest first
            int p = (unsigned)(hd >> 32) == rl.epoch ? (int)(unsigned)(hd & 0xffffffffull) : -1;
            for (int hops = 0; p >= 0 && hops < e - s; ++hops) {      // a cell has at most as many runs as particles
                const int2 rec = rl.rec[p];
                r0 += p < i ? rec.y : 0;
                p = rec.x;
            }
        }
        d0 = s + r0;
    }
    d0 = __shfl(d0, hl, 64);
    if (valid) inv[d0 + (lane - hl)] = i;
}

// ------------------------------------------------------------------ generic neighbour pass
// One (ox, oy) run of one lane: candidates [js, je) of the sorted arrays; LDS index = j + loff.
// LDS tile layout: two float2 arrays (x,y) and (z,w).  ds_read_b64 is serviced in two 32-lane groups
// with a 64-bank modulus, so lanes of neighbouring cells (8 particles = 64 B apart) hit disjoint
// banks and lanes of one cell broadcast: phase 1 costs 2 x 2 LDS cycles per candidate, conflict-free
// for typical cell populations (a float4 tile read as b128/b96 is 2-way conflicted at 8 per cell).
#define NBR_PAD 40   // phase 1 may read up to 32+7 slots past a lane's run (masked afterwards)
typedef float v2f __attribute__((ext_vector_type(2)));
// 8 candidates = 8 ds_read_b64 (x,y) from `a` + 8u and 8 ds_read_b32 (z alone: w is not needed for the distance test,
// 12 instead of 16 LDS bytes per candidate and lane) from `a` + zw_off + 8u, one s_waitcnt at the end.
template <int ZW_OFF>
__device__ __forceinline__ void lds_load_chunk_imm(unsigned a, v2f (&xy)[8], float (&z)[8]) {
    asm volatile(
        "ds_read_b64 %0, %16\n\tds_read_b32 %8, %16 offset:%17\n\t"
        "ds_read_b64 %1, %16 offset:8\n\tds_read_b32 %9, %16 offset:%17+8\n\t"
        "ds_read_b64 %2, %16 offset:16\n\tds_read_b32 %10, %16 offset:%17+16\n\t"
        "ds_read_b64 %3, %16 offset:24\n\tds_read_b32 %11, %16 offset:%17+24\n\t"
        "ds_read_b64 %4, %16 offset:32\n\tds_read_b32 %12, %16 offset:%17+32\n\t"
        "ds_read_b64 %5, %16 offset:40\n\tds_read_b32 %13, %16 offset:%17+40\n\t"
        "ds_read_b64 %6, %16 offset:48\n\tds_read_b32 %14, %16 offset:%17+48\n\t"
        "ds_read_b64 %7, %16 offset:56\n\tds_read_b32 %15, %16 offset:%17+56\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(xy[0]), "=&v"(xy[1]), "=&v"(xy[2]), "=&v"(xy[3]), "=&v"(xy[4]), "=&v"(xy[5]), "=&v"(xy[6]), "=&v"(xy[7]),
          "=&v"(z[0]), "=&v"(z[1]), "=&v"(z[2]), "=&v"(z[3]), "=&v"(z[4]), "=&v"(z[5]), "=&v"(z[6]), "=&v"(z[7])
        : "v"(a), "n"(ZW_OFF)
        : "memory");
}
// 4 candidates per LDS round trip: half the registers of the 8-candidate chunk (what lets the density pass run 5
// workgroups per CU without spilling, see nbr_waves_per_simd)
template <int ZW_OFF>
__device__ __forceinline__ void lds_load_half_imm(unsigned a, v2f (&xy)[4], float (&z)[4]) {
    asm volatile(
        "ds_read_b64 %0, %8\n\tds_read_b32 %4, %8 offset:%9\n\t"
        "ds_read_b64 %1, %8 offset:8\n\tds_read_b32 %5, %8 offset:%9+8\n\t"
        "ds_read_b64 %2, %8 offset:16\n\tds_read_b32 %6, %8 offset:%9+16\n\t"
        "ds_read_b64 %3, %8 offset:24\n\tds_read_b32 %7, %8 offset:%9+24\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(xy[0]), "=&v"(xy[1]), "=&v"(xy[2]), "=&v"(xy[3]), "=&v"(z[0]), "=&v"(z[1]), "=&v"(z[2]), "=&v"(z[3])
        : "v"(a), "n"(ZW_OFF)
        : "memory");
}
#ifndef SPH_P1_CHUNK
#define SPH_P1_CHUNK 4   // candidates per LDS round trip of the unrolled phase 1 (8: r01 layout, 4: fits 96 VGPRs)
#endif
#ifndef SPH_P1_TRIP
#define SPH_P1_TRIP 4    // candidates per wave-uniform trip of phase 1.  8 (until round 6): two LDS round trips of 4 per loop test; 4: one -- 3.6 % / 6.9 % fewer
                         // tests from rest / in motion (tools/analysis/zbin_estimate.py) for one more loop test per 8: C2 +-0 from rest, -1.1 % in motion (profiles/r06_p1_trip4_ab.txt)
#endif
typedef __attribute__((address_space(3))) const unsigned long long lds_cu64;
typedef __attribute__((address_space(3))) const int lds_ci32;
__device__ __forceinline__ int lds_ld_i32(const int *p) { return *(lds_ci32 *)p; }
typedef __attribute__((address_space(3))) const unsigned short lds_cu16;
__device__ __forceinline__ int lds_ld_u16(const unsigned short *p) { return (int)*(lds_cu16 *)p; }
__device__ __forceinline__ unsigned lds_addr(const float2 *p) { return (unsigned)(size_t)(lds_cu64 *)p; }
__device__ __forceinline__ float2 lds_ld2a(unsigned byte_addr) {  // plain (schedulable) ds_read_b64 from an LDS byte address
    const unsigned long long v = *(lds_cu64 *)(size_t)byte_addr;
    return make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
}
// P::FLUID_BLOCKS_ONLY: the functor is active for fluid particles only and its passive() is empty, so workgroups without
// a single fluid particle (the bulk of a scene with a sampled domain box) need not be launched at all.
template <class P, class = void> struct PassFluidOnly { static constexpr bool value = false; };
template <class P> struct PassFluidOnly<P, decltype((void)P::FLUID_BLOCKS_ONLY)> { static constexpr bool value = P::FLUID_BLOCKS_ONLY; };

// P::MODES: which MASKMODE instantiations of k_nbr_pass a functor is ever launched with (bit m = mode m).  Default: 0 and 2
// (a pass that reuses the masks stored by the first pass after a sort, with mode 0 as its fallback when none are valid).
// The passes that run first after a sort (density, DFSPH density + alpha) declare 0 | 1, the rigid volume pass 0 only.
// Instantiating only what can run keeps dead, register-hungry variants (the unrolled phase 1 inside a force pass) out
// of the code object; a mask-reusing pass also takes the lean one-candidate-at-a-time phase 1 in its mode-0 fallback.
// P::SPLIT3: the functor's sum is linear in its pairs and it has partial(): launched with gridDim.y == 3, workgroup (b, g)
// walks only x-offset group g of tile b and stores its part of the sum; a small kernel of the functor's owner adds the three
// parts and does what finish() does.  For passes over FEW particles (a 100 k-particle sheet is 416 tiles: less than half of
// what the chip holds at once, one wave per SIMD, every pass pure latency) this triples the waves in flight.
template <class P, class = void> struct PassSplit { static constexpr bool value = false; };
template <class P> struct PassSplit<P, decltype((void)P::SPLIT3)> { static constexpr bool value = P::SPLIT3; };

template <class P, class = void> struct PassModes { static constexpr int value = 0b101; };
template <class P> struct PassModes<P, decltype((void)P::MODES)> { static constexpr int value = P::MODES; };
template <class P> constexpr bool pass_builds_masks() { return (PassModes<P>::value & 0b010) != 0; }
template <class P> constexpr bool pass_reuses_masks() { return (PassModes<P>::value & 0b100) != 0; }

// P::HAS_PROLOGUE: the functor has `bool prologue(DevScalars *) const`, run by every thread of a workgroup before anything else
// (it may use barriers and set `mutable` members, e.g. a coefficient every workgroup reduces from per-workgroup partials);
// false = this workgroup has nothing to do.
// P::MASK_PIPELINE = false: the functor has no three registers to spare for next round's mask words across its pair loop; they are
// then loaded at the top of the round (still without a wait of their own).
template <class P, class = void> struct PassMaskPipe { static constexpr bool value = true; };
template <class P> struct PassMaskPipe<P, decltype((void)P::MASK_PIPELINE)> { static constexpr bool value = P::MASK_PIPELINE; };
template <class P, class = void> struct PassUsesJ0 { static constexpr bool value = true; };
template <class P> struct PassUsesJ0<P, decltype((void)P::USES_J)> { static constexpr bool value = P::USES_J; };
// (default: the all-fluid instantiations pipeline; the ones that tell rigid neighbours apart -- USES_J -- run closer to their limit)
template <class P> constexpr bool pass_mask_pipe() { return PassMaskPipe<P>::value && !(P::HAS_B && PassUsesJ0<P>::value); }
// P::HAS_WRENCH: pair() may call add_wrench (fluid -- dynamic rigid body pairs): the kernel opens the per-wave LDS rows at its start and
// flushes them behind its pair loops (sph_passes.hpp)
template <class P, class = void> struct PassWrench { static constexpr bool value = false; };
template <class P> struct PassWrench<P, decltype((void)P::HAS_WRENCH)> { static constexpr bool value = P::HAS_WRENCH; };
__device__ __forceinline__ void wrench_init_all(const RigidPose *pose);
__device__ __forceinline__ void wrench_flush_all(DevScalars *scal);
template <class P, class = void> struct PassStatW { static constexpr bool value = false; };
template <class P> struct PassStatW<P, decltype((void)P::STAT_W)> { static constexpr bool value = P::STAT_W; };
// P::NEXT_HASH: the functor's finish() leaves the particle's NEW cell in Own::lin_new and carries a NextHash `nh`: when nh.on, the kernel's
// epilogue does for the next step's sort what k_hash_count would do (WcsphForcePass)
template <class P, class = void> struct PassNextHash { static constexpr bool value = false; };
template <class P> struct PassNextHash<P, decltype((void)P::NEXT_HASH)> { static constexpr bool value = P::NEXT_HASH; };
template <class P, class = void> struct PassPrologue { static constexpr bool value = false; };
template <class P> struct PassPrologue<P, decltype((void)P::HAS_PROLOGUE)> { static constexpr bool value = P::HAS_PROLOGUE; };

// Optional second per-candidate payload (P::HAS_C, P::CT): staged into its own LDS array; stage() and pair() of
// such a functor take it as one more argument.
template <class P, class = void> struct PassC { static constexpr bool value = false; typedef int type; };
template <class P> struct PassC<P, decltype((void)P::HAS_C)> { static constexpr bool value = P::HAS_C; typedef typename P::CT type; };

template <class P>
__device__ __forceinline__ float4 pass_stage(const P &p, const Consts &c, int j, typename P::BT &bj, typename PassC<P>::type &cj) {
    if constexpr (PassC<P>::value) return p.stage(c, j, bj, cj);
    else return p.stage(c, j, bj);
}
template <class P>
__device__ __forceinline__ void pass_pair(const P &p, const Consts &c, typename P::Own &own, float dx, float dy, float dz,
                                          float r2, const float4 &a, const typename P::BT &bj,
                                          const typename PassC<P>::type &cj, int j) {
    if constexpr (PassC<P>::value) p.pair(c, own, dx, dy, dz, r2, a, bj, cj, j);
    else p.pair(c, own, dx, dy, dz, r2, a, bj, j);
}

// MASKMODE: 0 = compute the acceptance masks; 1 = compute and store the first 32-candidate chunk of every run
// (first neighbour pass after a sort); 2 = reuse the stored chunk (later passes over the same sorted positions:
// phase 1 disappears).  Stored form: bit t = candidate js + t accepted (self already removed).

// Phase 1 for the <= 32 candidates at tile slots [base, base + m).  Must be called by all lanes of the wave
// (wave-uniform trip count; lanes without candidates pass m = 0).  Returns bit t = slot base + t accepted.
// LEAN: one candidate per iteration, few registers -- for the passes that reuse stored masks and only get here for
// the rare candidates beyond a run's first 32 (keeps those kernels at 80 VGPRs = 6 waves per SIMD).
template <int ZW_OFF, bool LEAN = false>
__device__ __forceinline__ unsigned phase1_mask(const float2 *sXY, int base, int m, float xi, float yi, float zi,
                                                float h2) {
    if (LEAN) {
        unsigned nm = 0;
        const unsigned tile = lds_addr(&sXY[base]);
#pragma unroll 1
        for (int t = 0; __any(t < m); ++t) {
            const float2 xy = lds_ld2a(tile + 8u * t), zw = lds_ld2a(tile + 8u * t + ZW_OFF);
            const float dx = xi - xy.x, dy = yi - xy.y, dz = zi - zw.x;
            const float r2 = dx * dx + dy * dy + dz * dz;
            nm |= (t < m && r2 < h2) ? 1u << t : 0u;
        }
        return nm;
    }
    // The acceptance bit of every slot is shifted into `mask` from the right by v_cmp (-> VCC) + v_addc_co
    // (mask = 2 mask + VCC): two VOPC/VOP2 instructions per slot, no SGPR-pair results, no shift constants.
    // After S pushes slot t sits at bit S-1-t.
    unsigned mask = 0;
    int S = 0;   // (== t0 behind the loop: one counter)
    int t0 = 0;
    do {   // bottom-tested (a top-tested loop keeps a copy of t0 and of the mask for its exit: two moves per chunk, 72 -> 67 VALU); a wave
           // whose lanes ALL have an empty run tests one chunk at its base for nothing (reads inside the tile, bits dropped below)
        // All 16 ds_read_b64 of the chunk are issued back to back from one base register with immediate
        // offsets and waited for once (hand-placed: left to itself the scheduler keeps at most one candidate
        // in flight, or fuses neighbours into ds_read2_b64 = 8 LDS cycles per 16 bytes).
        if constexpr (SPH_P1_CHUNK == 8) {
            v2f xy[8];
            float zz[8];
            lds_load_chunk_imm<ZW_OFF>(lds_addr(&sXY[base + t0]), xy, zz);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float dx = xi - xy[u].x, dy = yi - xy[u].y, dz = zi - zz[u];
                const float r2 = dx * dx + dy * dy + dz * dz;
                asm("v_cmp_gt_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                    : "+v"(mask) : "v"(r2), "s"(h2) : "vcc");  // not volatile: ordered by the dependence on mask
            }
        } else {
#pragma unroll
            for (int hh = 0; hh < SPH_P1_TRIP / 4; ++hh) {
                v2f xy[4];
                float zz[4];
                lds_load_half_imm<ZW_OFF>(lds_addr(&sXY[base + t0 + 4 * hh]), xy, zz);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float dx = xi - xy[u].x, dy = yi - xy[u].y, dz = zi - zz[u];
                    const float r2 = dx * dx + dy * dy + dz * dz;
                    asm("v_cmp_gt_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                        : "+v"(mask) : "v"(r2), "s"(h2) : "vcc");
                }
            }
        }
        t0 += SPH_P1_CHUNK == 8 ? 8 : SPH_P1_TRIP;
    } while (__any(t0 < m));
    S = t0;
    unsigned nm = S > 0 ? __brev(mask) >> (32 - S) : 0u;   // bit t = slot t
    nm &= m >= 32 ? 0xffffffffu : ((1u << m) - 1u);        // drop slots past this lane's run
    return nm;
}

// Tile overflow (one run longer than the tile; every run in debug mode 1): the run goes through the tile in chunks of CAP slots and every
// lane walks the part of its candidates [js, je) that lies in the chunk [cs, ce) (sorted indices), in 32-candidate blocks aligned to js --
// so the two stored mask words keep their meaning -- and ascending throughout: the reference's order.  Masks are always recomputed here (a
// stored word may come from a pass with a larger tile, which did not overflow); the pass that stores them collects its two words over the
// chunks in w01.  (Until round 3 such a run was walked out of L2, candidate by candidate; that variant cost every instantiation of
// k_nbr_pass 10-20 VGPRs it almost never used: the density pass needs 96 with it, 80 without -- a sixth workgroup per CU.)
template <int ZW_OFF, int MASKMODE, class P>
__device__ __forceinline__ void process_chunk(const Consts &c, const P &p, typename P::Own &own, int i, float xi, float yi, float zi,
                                              int js, int je, int cs, int ce, const float2 *sXY, const float2 *sZW,
                                              const typename P::BT *sB, const typename PassC<P>::type *sC, unsigned &npairs,
                                              unsigned long long &w01) {
    const int lo = js > cs ? js : cs, hi = je < ce ? je : ce;   // this lane's candidates inside the chunk: [lo, hi)
    int k = lo > js ? (lo - js) >> 5 : 0;
    for (; __any(lo < hi && js + 32 * k < hi); ++k) {           // wave-uniform trip count; a lane that is done passes m = 0
        const int b0r = js + 32 * k;                            // block k of this lane's run
        const int b0 = b0r > lo ? b0r : lo;
        const int b1 = b0r + 32 < hi ? b0r + 32 : hi;
        int m = lo < hi ? b1 - b0 : 0;
        m = m < 0 ? 0 : m;
        const int base = m > 0 ? b0 - cs : 0;                   // tile slot of candidate b0
        unsigned nm = phase1_mask<ZW_OFF, true>(sXY, base, m, xi, yi, zi, c.h2);   // bit t = candidate b0 + t (the lean form: few registers, no reads past the chunk)
        const unsigned self = (unsigned)(i - b0);
        if (self < 32u) nm &= ~(1u << self);                    // p_i != p_j (base_container.py:559)
        if (MASKMODE == 1 && m > 0 && k < 2) w01 |= (unsigned long long)nm << (unsigned)(b0 - js);   // bit t of the two words = candidate js + t
        npairs += __popc(nm);
        while (nm) {
            const int t = __ffs(nm) - 1;
            nm &= nm - 1;
            const float2 xy = sXY[base + t];
            const float2 zw = sZW[base + t];
            const float dx = xi - xy.x, dy = yi - xy.y, dz = zi - zw.x;
            const float r2 = dx * dx + dy * dy + dz * dz;
            typename P::BT bj = typename P::BT();
            if (P::HAS_B) bj = sB[base + t];
            typename PassC<P>::type cj = typename PassC<P>::type();
            if (PassC<P>::value) cj = sC[base + t];
            pass_pair(p, c, own, dx, dy, dz, r2, make_float4(xy.x, xy.y, zw.x, zw.y), bj, cj, b0 + t);
        }
    }
}

#ifndef SPH_P2_PAIR2_MEDIUM
#define SPH_P2_PAIR2_MEDIUM 1
#endif
template <class P> constexpr bool pass_is_medium();
template <class P> constexpr bool pass_pair2() {
#ifdef SPH_P2_PAIR2
    return true;
#else
#ifdef SPH_P2_PAIR2_LIGHT
    if (!P::HAS_B) return true;
#endif
    return SPH_P2_PAIR2_MEDIUM && pass_is_medium<P>();
#endif
}

// pair() receives the neighbour's sorted index j only where the functor needs it (rigid-body wrench):
// P::USES_J = false lets the merged loop drop the bookkeeping.
template <class P, class = void> struct PassUsesJ { static constexpr bool value = true; };
template <class P> struct PassUsesJ<P, decltype((void)P::USES_J)> { static constexpr bool value = P::USES_J; };

// Phase 2 of one staging group: the accepted candidates of the group's three runs (masks m0..m2, bit t = tile slot
// base_q + t, byte offsets a_q = 8 base_q) are consumed in ONE loop, run after run in ascending order -- the
// reference's accumulation order -- so a wave iterates max_lanes(sum of the three runs) times instead of
// sum_runs max_lanes(run).  Together with the lane permutation (k_lane_perm) this removes most of the divergence
// loss: C2 lattice 45 -> 26 iterations per wave, disordered dam break 96 -> 53 (tools/analysis/imbalance.py).
template <class P, int ZW_OFF, class M>   // M = unsigned (runs of <= 32 candidates) or unsigned long long (<= 64)
__device__ __forceinline__ void merged_phase2(const Consts &c, const P &p, typename P::Own &own, float xi, float yi,
                                              float zi, M m0, M m1, M m2, unsigned a0,
                                              unsigned a1, unsigned a2, const float2 *sXY,
                                              const typename P::BT *sB, const typename PassC<P>::type *sC,
                                              const int *s_loff3) {
    constexpr bool UJ = PassUsesJ<P>::value;
    int q0 = 0, q1 = 1, q2 = 2;
    // non-empty runs first (order kept)
    if (m0 == 0) { m0 = m1; a0 = a1; q0 = q1; m1 = m2; a1 = a2; q1 = q2; m2 = 0; }
    if (m0 == 0) { m0 = m1; a0 = a1; q0 = q1; m1 = 0; }
    if (m1 == 0) { m1 = m2; a1 = a2; q1 = q2; m2 = 0; }
    M cur = m0;
    unsigned ca = a0;
    int cq = q0;
    const unsigned tile = lds_addr(sXY);
    if constexpr (pass_pair2<P>()) {
    // Two accepted neighbours of a run per trip -- all their LDS reads issued before the first pair is evaluated, half the loop control;
    // order of accumulation unchanged (first, then second).  Pays where runs hold several accepted neighbours and the pair is light: the
    // 20-28-byte solver walks (pass_is_medium) in motion, C3 -2.7 % at step 1000, +-0 from rest; the WCSPH force pass loses 1.2 % from rest
    // with it (profiles/r06_pair2_ab.txt).  -DSPH_P2_PAIR2: every pass (A/B); -DSPH_P2_PAIR2_MEDIUM=0: none.
    while (cur) {
        const int t = (sizeof(M) == 8 ? __ffsll((long long)cur) : __ffs((int)cur)) - 1;
        cur &= cur - 1;
        const bool two = cur != 0;
        const int t2 = two ? (sizeof(M) == 8 ? __ffsll((long long)cur) : __ffs((int)cur)) - 1 : t;
        if (two) cur &= cur - 1;
        const unsigned ad = ca + ((unsigned)t << 3), ad2 = ca + ((unsigned)t2 << 3);
        const float2 xy = lds_ld2a(tile + ad), zw = lds_ld2a(tile + ad + ZW_OFF);
        const float2 xy2 = lds_ld2a(tile + ad2), zw2 = lds_ld2a(tile + ad2 + ZW_OFF);
        typename P::BT bj = typename P::BT(), bj2 = typename P::BT();
        if (P::HAS_B) { bj = sB[ad >> 3]; bj2 = sB[ad2 >> 3]; }
        typename PassC<P>::type cj = typename PassC<P>::type(), cj2 = typename PassC<P>::type();
        if (PassC<P>::value) { cj = sC[ad >> 3]; cj2 = sC[ad2 >> 3]; }
        {
            const float dx = xi - xy.x, dy = yi - xy.y, dz = zi - zw.x;
            int j = 0;
            if (UJ) j = (int)(ad >> 3) - s_loff3[cq];
            pass_pair(p, c, own, dx, dy, dz, dx * dx + dy * dy + dz * dz, make_float4(xy.x, xy.y, zw.x, zw.y), bj, cj, j);
        }
        if (two) {
            const float dx = xi - xy2.x, dy = yi - xy2.y, dz = zi - zw2.x;
            int j = 0;
            if (UJ) j = (int)(ad2 >> 3) - s_loff3[cq];
            pass_pair(p, c, own, dx, dy, dz, dx * dx + dy * dy + dz * dz, make_float4(xy2.x, xy2.y, zw2.x, zw2.y), bj2, cj2, j);
        }
        if (cur == 0) { cur = m1; ca = a1; m1 = m2; a1 = a2; m2 = 0; if (UJ) { cq = q1; q1 = q2; } }
    }
    } else {
    while (cur) {
        const int t = (sizeof(M) == 8 ? __ffsll((long long)cur) : __ffs((int)cur)) - 1;
        cur &= cur - 1;
        const unsigned ad = ca + ((unsigned)t << 3);
        const float2 xy = lds_ld2a(tile + ad);
        const float2 zw = lds_ld2a(tile + ad + ZW_OFF);
        typename P::BT bj = typename P::BT();
        if (P::HAS_B) bj = sB[ad >> 3];
        typename PassC<P>::type cj = typename PassC<P>::type();
        if (PassC<P>::value) cj = sC[ad >> 3];
        const float dx = xi - xy.x, dy = yi - xy.y, dz = zi - zw.x;
        const float r2 = dx * dx + dy * dy + dz * dz;
        int j = 0;
        if (UJ) j = (int)(ad >> 3) - s_loff3[cq];
        pass_pair(p, c, own, dx, dy, dz, r2, make_float4(xy.x, xy.y, zw.x, zw.y), bj, cj, j);
        if (cur == 0) { cur = m1; ca = a1; m1 = m2; a1 = a2; m2 = 0; if (UJ) { cq = q1; q1 = q2; } }
    }
    }
}

// Per-workgroup data prepared once per sort (k_block_prep), read by every neighbour pass of the sort epoch:
// header = cells of the first / last particle + the 9 candidate-run windows (start, length); lane permutation.
#define BLK_HDR_INTS 20   // [0] first cell, [1] last cell, [2..10] run start, [11..19] run length
#define NBR_CS_SPAN 124   // cell_start window cached in LDS per run: cells [first-1, last+1] (+ end) of the workgroup
#define NBR_CS_PITCH (NBR_CS_SPAN + 4)
#define NBR_BLOCK 256

struct BlockPrepTables { int *tab[8]; };
// slab sharding: the boundary / interior tile lists of the compute / halo overlap (State::tile_list).  lo_layers / hi_layers = local
// layers at the low / high end of the slab that belong to the boundary set (ghost layer + two own layers; 0 where the slab has no
// neighbour on that side).
struct TilePlanOut { int *list_b, *list_i, *cnt; unsigned char *cls; int lo_layers, hi_layers; int bound_b; int *status; volatile int *mirror_nb; };
// Lane permutation of a workgroup (256 consecutive sorted particles): particles stably sorted by their x position
// inside the cell.  The x-offset groups (-1, 0, +1) of the neighbour pass hold very different numbers of accepted
// neighbours for particles in the low-x and the high-x part of a cell; putting like with like makes the 64 lanes
// of a wave agree on their trip counts per group.  Any permutation is correct (every lane still walks its own
// particle's neighbours in reference order); this one is only faster.  perm[b * 256 + lane] = particle of the lane.
// The per-tile part (header, cell words, lane permutation, fluid flag) of tile blockIdx.x, given every thread's particle: `p` / `meta_i` are
// the position and meta word of particle blockIdx.x * 256 + threadIdx.x (anything where that is >= n).  Shared by k_block_prep (which reads
// them from the sorted arrays) and k_gather_prep (which has just moved them there).  All 256 threads of the workgroup must call it.
__device__ __forceinline__ void block_prep_tile(const Consts &c, int n, const float4 p, int meta_i, const int *__restrict__ cell_start,
                                                int *__restrict__ blk_hdr, unsigned char *__restrict__ perm, int *__restrict__ blk_flag,
                                                unsigned *__restrict__ cellw, int (*s_cnt)[64], int *s_c) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i0 = blockIdx.x * 256;
    const int i = i0 + tid;
    const int nvalid = (n - i0) < 256 ? (n - i0) : 256;
    int key = 63;  // slots past the end go last
    int my_lin = 0, my_dz0 = 0, my_dz = 0;   // this particle's cell, z0 - cz and z1 - z0 (for its cell word, below)
    unsigned my_dom = 0u;
    if (i < n) {
        const int cxg = cell_coord(p.x, c.grid_size, c.nx_glob);   // (the key is the position inside the cell: global layer)
        const int k = (int)((p.x / c.grid_size - (float)cxg) * 62.0f);
        const int cx = cell_coord_x(c, p.x);
        key = k < 0 ? 0 : (k > 62 ? 62 : k);
        {
            const int cy = cell_coord(p.y, c.grid_size, c.ny);
            const int cz = cell_coord_z(c, p.z);
            const int lin = (cx * c.ny + cy) * c.nz + cz;
            if (tid == 0) s_c[0] = lin;
            if (tid == nvalid - 1) s_c[1] = lin;
            const int z0 = cz > 0 ? cz - 1 : 0;
            const int z1 = cz < c.nz - 1 ? cz + 1 : c.nz - 1;
            my_lin = lin; my_dz0 = z0 - cz; my_dz = z1 - z0;
            // bit k = 3 (ox + 1) + (oy + 1): column (cx + ox, cy + oy) lies inside the grid (k_nbr_pass)
            const unsigned by = (cy > 0 ? 1u : 0u) | 2u | (cy < c.ny - 1 ? 4u : 0u);
            my_dom = (cx > 0 ? by : 0u) | (by << 3) | (cx < c.nx - 1 ? by << 6 : 0u);
        }
    }
    (&s_cnt[0][0])[tid] = 0;
    if (blk_flag) {   // does this workgroup hold any fluid particle? (k_compact_blocks lists those that do)
        const int anyf = __syncthreads_or((i < n && META_ACTIVE_FLUID(meta_i)) ? 1 : 0);
        if (tid == 0) blk_flag[blockIdx.x] = anyf ? 1 : 0;
    }
    unsigned long long peers = ~0ull;   // lanes of this wave holding the same key
#pragma unroll
    for (int bit = 0; bit < 6; ++bit) {
        const bool on = (key >> bit) & 1;
        const unsigned long long bal = __ballot(on);
        peers &= on ? bal : ~bal;
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    const int rank_in_wave = __popcll(peers & below);
    __syncthreads();
    if (cellw && i < n) {
        // The CELL WORD of every particle, for all neighbour passes of this sort epoch (round 5): what a pass needs of its lane's cell --
        // e0 = window-cache entry of (.., .., z0) in every run (bits 0-7; meaningful while the tile spans <= NBR_CS_SPAN cells), z1 - z0 + 1
        // (bits 8-10), the nine "column exists" bits (11-19).  Each pass used to derive them from the position: three IEEE divisions
        // (the cell must agree with the hash kernel's), clamps and range tests, ~100 VALU per wave and pass (profiles/r05_isa_census.txt).
        const int e0 = (my_lin - s_c[0]) + my_dz0 + 1;
        cellw[i] = (unsigned)(e0 & 0xff) | ((unsigned)(my_dz + 1) << 8) | (my_dom << 11);
    }
    if ((peers & below) == 0ull) s_cnt[w][key] = __popcll(peers);
    if (tid < 9) {
        const int cfirst = s_c[0], clast = s_c[1];
        const int shift = (tid / 3 - 1) * c.ny * c.nz + (tid % 3 - 1) * c.nz;
        int lo = cfirst + shift - 1, hi = clast + shift + 1;
        int rs = 0, re = 0;
        if (clast >= cfirst && hi >= 0 && lo <= c.G - 1) {
            lo = lo < 0 ? 0 : lo;
            hi = hi > c.G - 1 ? c.G - 1 : hi;
            rs = cell_start[lo];
            re = cell_start[hi + 1];
        }
        int *h = blk_hdr + (size_t)blockIdx.x * BLK_HDR_INTS;
        if (tid == 0) { h[0] = cfirst; h[1] = clast; }
        h[2 + tid] = rs;
        h[11 + tid] = re - rs;
    }
    __syncthreads();
    if (perm) {
        const int tot = s_cnt[0][lane] + s_cnt[1][lane] + s_cnt[2][lane] + s_cnt[3][lane];
        const int incl = wave_incl_scan(tot);
        int dest = __shfl(incl - tot, key, 64) + rank_in_wave;
        for (int k = 0; k < w; ++k) dest += s_cnt[k][key];
        perm[i0 + dest] = (unsigned char)tid;
    }
}

__global__ void __launch_bounds__(256)
k_block_prep(const Consts c, const float4 *__restrict__ posv, const int *__restrict__ meta,
             const int *__restrict__ cell_start, int *__restrict__ blk_hdr, unsigned char *__restrict__ perm,
             int *__restrict__ blk_flag, const int *__restrict__ xidx, BlockPrepTables tabs, TilePlanOut plan, unsigned *__restrict__ cellw) {
    __shared__ int s_cnt[4][64];
    __shared__ int s_c[2];
    const int tid = threadIdx.x;
    const int i0 = blockIdx.x * 256;
    const int i = i0 + tid;
    const int n = live_n(c);
    if (plan.list_b && tid == 64 && i0 < n) {
        // which set does this tile belong to?  Interior = entirely inside the particle range [r1, r2) of the layers that are neither
        // ghost layers nor within two layers of a face; everything else (and every tile of a slab too thin to have such layers) = boundary.
        const int layer = c.ny * c.nz;
        const int r1 = plan.lo_layers > 0 ? cell_start[(plan.lo_layers < c.nx ? plan.lo_layers : c.nx) * layer] : 0;
        const int r2 = plan.hi_layers > 0 ? cell_start[(c.nx - plan.hi_layers > 0 ? c.nx - plan.hi_layers : 0) * layer] : n;
        const int nt = (n + 255) >> 8;
        int t1 = (r1 + 255) >> 8, t2 = r2 >> 8;          // interior tiles: [t1, t2)
        if (t2 > nt) t2 = nt;
        if (t2 < t1) t2 = t1 = nt;                         // none
        const int t = (int)blockIdx.x, ni = t2 - t1;
        if (t < t1) plan.list_b[t] = t;
        else if (t >= t2) plan.list_b[t1 + (t - t2)] = t;
        else plan.list_i[t - t1] = t1 + xcd_remap(t - t1, ni);   // slot k holds the k-th tile of an XCD-aware order (bijective)
        plan.cls[t] = (t < t1 || t >= t2) ? 1 : 0;
        if (t == 0) {
            plan.cnt[0] = nt - ni; plan.cnt[1] = ni;
            if (nt - ni > plan.bound_b && plan.status) atomicOr(plan.status, SLAB_ST_BOUND);   // boundary launches of this sort epoch would miss tiles
            if (plan.mirror_nb) *plan.mirror_nb = nt - ni;
        }
    }
    if (i0 >= n) {   // launch bound of an asynchronous slab step: no such tile
        if (blk_flag && tid == 0) blk_flag[blockIdx.x] = 0;
        return;
    }
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    int meta_i = 0;
    if (i < n) {
        if (xidx) {   // slab sharding, push transport: the halo slot tables of this sort (sph_halo.hpp k_halo_tables) ride along
            const int x = xidx[i];
            const int kind = (int)(((unsigned)x) >> 28);
            if (kind >= 1 && kind <= 8) tabs.tab[kind - 1][x & 0x0fffffff] = i;
        }
        p = posv[i];
        if (blk_flag) meta_i = meta[i];
    }
    block_prep_tile(c, n, p, meta_i, cell_start, blk_hdr, perm, blk_flag, cellw, s_cnt, s_c);
}

// Deterministic sort by run lists, second half: base_container.py:506 reorder_particles as a GATHER by destination tile (inv from
// k_sort_rank) with k_block_prep's work fused in -- the workgroup that fills the 256 slots of a tile holds their positions in registers, so
// it writes the tile's header, cell words and lane permutation as well: the sort is scan + rank + this (3 launches; 4 with k_scatter_index
// / k_scatter / k_block_prep), and nobody reads the 16 B positions a second time.  Unsharded scenes only (n is exact, no slot tables).
__global__ void __launch_bounds__(256)
k_gather_prep(const Consts c, int n, const int *__restrict__ inv, SortArrays a, const int *__restrict__ cell_start,
              int *__restrict__ blk_hdr, unsigned char *__restrict__ perm, int *__restrict__ blk_flag, unsigned *__restrict__ cellw) {
    __shared__ int s_cnt[4][64];
    __shared__ int s_c[2];
    const int d = blockIdx.x * 256 + threadIdx.x;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    int meta_i = 0;
    if (d < n) {
        const int i = inv[d];
        p = a.posv_in[i];
        meta_i = a.meta_in[i];
        a.posv_out[d] = p;
        a.velm_out[d] = a.velm_in[i];
        a.meta_out[d] = meta_i;
        a.pid_out[d] = a.pid_in[i];
        if (a.color_in) a.color_out[d] = a.color_in[i];   // (null: the colours stay at home, State::color_home)
        if (a.rho_in) a.rho_out[d] = a.rho_in[i];         // (null: the next kernel recomputes every density, State::sort_skip_rho)
        if (a.orig_in) a.orig_out[d] = a.orig_in[i];
    }
    block_prep_tile(c, n, p, meta_i, cell_start, blk_hdr, perm, blk_flag, cellw, s_cnt, s_c);
}

// State::color in sorted order from the colours at home (color_home[particle id]; ids = append order)
__global__ void __launch_bounds__(256)
k_color_from_home(int n, const int *__restrict__ pid, const unsigned *__restrict__ home, unsigned *__restrict__ color) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) color[i] = home[pid[i]];
}

// ascending list of the workgroups whose flag is set (one workgroup, fixed order: the list is deterministic).  Every thread owns a
// contiguous stretch of flags, counts it, ONE block scan gives its offset, then it files its stretch (until round 5: a block scan
// with two barriers per 256 flags -- 34 of them, 20 us, for the 8,496 tiles of the buckling scene).
__global__ void __launch_bounds__(256)
k_compact_blocks(const int *__restrict__ flag, int nb, int *__restrict__ list, int *__restrict__ count, volatile int *count_host) {
    __shared__ int s_w[4];
    const int per = (nb + 255) / 256;
    const int lo = threadIdx.x * per, hi = lo + per < nb ? lo + per : nb;
    int mine = 0;
    for (int b = lo; b < hi; ++b) mine += flag[b] ? 1 : 0;
    int tot;
    int at = block_excl_scan_256(mine, s_w, tot);
    for (int b = lo; b < hi; ++b) if (flag[b]) list[at++] = b;
    if (threadIdx.x == 0) { *count = tot; if (count_host) *count_host = tot; }   // (pinned host memory: State::list_count_pinned)
}

// LDS particle slots per staging group.  All instantiations run 4 workgroups per CU (see nbr_waves_per_simd), so each
// may use a quarter of the 160 KB: the tile gets what is left after the cell windows, up to 1280 slots (5 per thread).
// A rest-density workgroup needs ~3 x 34 cells x 8 = 816; moving fluid piles up to 900-1000, and every overflow sends
// a whole group down the ordered path.
// "Medium" functors: payload arrays, but a record of <= SPH_NBR_MEDIUM_BYTES bytes and a small pair() (the DFSPH / PCISPH solver walks and
// the unfused pressure walk: 20-28 B per slot, a dozen flops per pair) that reuse stored masks.  At 4 workgroups per CU they sat at
// 103-113 VGPRs only because the launch bound allowed 128; a fifth of the CU's LDS still holds a rest-density group (816 slots of 28 B =
// 22 KB), so they run 5 workgroups per CU (<= 96 VGPRs, 32 KB) -- one more wave per SIMD for walks that are pure latency.
// What had to give for 96 registers: the staging batch (SPH_MEDIUM_SB slots per thread in flight instead of all 4-5: with 4 the 28-byte
// walks spill 10 VGPRs, with 2 nothing spills in either the all-fluid or the rigid-aware instantiations).  Measured at C3 / PCISPH
// (profiles/r03n_ab_medium_occupancy.txt): 24-byte subset -2 %, 28 bytes with spills -3.5 %, 28 bytes / batches of 2 (this) -3 % from
// rest and -6 % in motion, no spills.  0 bytes = off.
#ifndef SPH_NBR_LIGHT_CAP
#define SPH_NBR_LIGHT_CAP 1280
#endif
#ifndef SPH_NBR_HEAVY_BUDGET
#define SPH_NBR_HEAVY_BUDGET 40960   // LDS bytes per workgroup of the functors that are neither light nor medium (A/B: 32768 with SPH_NBR_WAVES_HEAVY=5)
#endif
#ifndef SPH_NBR_MEDIUM_BYTES
#define SPH_NBR_MEDIUM_BYTES 28
#endif
#ifndef SPH_HEAVY_SB
#define SPH_HEAVY_SB 3   // staging slots per batch of the wide records (32-36 B per slot; all 5 at once spill at 128 VGPRs)
#endif
#ifndef SPH_MEDIUM_SB
#define SPH_MEDIUM_SB 3   // staging slots per batch of a medium functor (round 6: 3 -- no spills any more; C3 -0.5 % from rest, -1.2 % in motion: profiles/r06_maskpipe_ab.txt)
#endif
template <class P> constexpr int pass_slot_bytes() {
    return 16 + (P::HAS_B ? (int)sizeof(typename P::BT) : 0) + (PassC<P>::value ? (int)sizeof(typename PassC<P>::type) : 0);
}
template <class P, class = void> struct PassMediumOk { static constexpr bool value = false; };
template <class P> struct PassMediumOk<P, decltype((void)P::MEDIUM_OK)> { static constexpr bool value = P::MEDIUM_OK; };
template <class P> constexpr bool pass_is_medium() {   // by record size, or opted in by the functor (P::MEDIUM_OK: records up to 32 B)
    return SPH_FAST && P::HAS_B && !pass_builds_masks<P>() &&
           (pass_slot_bytes<P>() <= SPH_NBR_MEDIUM_BYTES || (PassMediumOk<P>::value && pass_slot_bytes<P>() <= 32));
}
template <class P> constexpr int nbr_tile_cap() {
    const int per_slot = pass_slot_bytes<P>();
    const int budget = pass_is_medium<P>() ? 32768 : SPH_NBR_HEAVY_BUDGET;   // a fifth / a quarter of the CU's 160 KB
    const int slots = (budget - 9 * NBR_CS_PITCH * 2 - 1536) / per_slot - NBR_PAD;   // 1.5 KB for the small arrays and the allocation granule
    // the payload-free passes (16 B per slot) could hold far more than a group ever needs: SPH_NBR_LIGHT_CAP slots, so that their tile
    // leaves room for SPH_NBR_WAVES_LIGHT workgroups per CU (1280 slots = 23.7 KB: six; 1232 = 22.9 KB: seven)
    return slots > SPH_NBR_LIGHT_CAP ? SPH_NBR_LIGHT_CAP : slots / 8 * 8;
}
// Which of the tile's nine candidate runs (k = 3 (ox + 1) + (oy + 1)) form staging group g?  (Consts::run_grouping)
//   0: the three runs of x offset g - 1, oy ascending -- the reference's order of accumulation (strict build, always), and the one whose
//      runs overlap on thin grids (nbr_plan "chain": slab-sharded ranks);
//   1 (round 6, fast build, unsharded grids with nz >= 40): the middle group as before, the outer two MIXED:
//        g = 0: (-1,-1) (-1,+1) (+1, 0)      g = 2: (-1, 0) (+1,-1) (+1,+1)
//      A lattice particle in the low-x half of its cell has 9 / 17 / 0 accepted neighbours in the x-offset groups, one in the high-x half
//      0 / 17 / 9: waves sorted by x (lane permutation) are uniform inside, but the four waves of a workgroup meet at a barrier behind every
//      group (the tile is restaged), so the workgroup's clock sees 9 + 17 + 9 = 35 merged-loop trips for 26 per wave.  With the outer runs
//      mixed every particle has 3 or 6 in either outer group: 6 + 17 + 6 = 29.  On real states (tools/analysis/grouping.py): per-workgroup
//      trips 34.0 -> 28.3 at rest, 73.4 -> 62.9 in motion, per-wave trips unchanged; best of all partitions into three groups of three.
//      The sums change by association only (the fast build does not keep the reference's order anyway).
__device__ __forceinline__ int run_of(int grouping, int g, int q) {
    const int i = g * 3 + q;
    return grouping ? (int)((0x861543720ull >> (4 * i)) & 15ull) : i;
}

// Staging plan of one round of one x-offset group (workgroup-uniform; see the group loop of k_nbr_pass): which of the group's three
// runs go into the tile (bit q of rm), at which tile offset (lo_[q] = offset - run start; INT_MIN: not staged), how many slots in all.
template <int CAP>
__device__ __forceinline__ void nbr_plan(const int *__restrict__ hdr, int g, int qa, int fg, int (&rs_)[3], int (&ln_)[3], int (&lo_)[3],
                                         int &total, int &qb, unsigned &rm, bool &overflow, int grouping) {
    total = 0; qb = 3; rm = 7u; overflow = false;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int k = run_of(grouping, g, q);
        rs_[q] = hdr[2 + k];
        ln_[q] = hdr[11 + k] > 0 ? hdr[11 + k] : 0;
    }
    // Thin grids (nz <= the workgroup's cell span + 2: small scenes, and the slabs of a sharded scene -- C4 on 8 ranks has 12
    // layers per rank): the three runs of a group are windows of one and the same stretch of the sorted arrays, nz cells apart,
    // and overlap.  Stage that stretch ONCE (all three runs share one tile offset) instead of three overlapping copies:
    // 34 + 2 nz cells instead of 102 -- at nz = 12 40 % less staging traffic and LDS, and piled-up groups fit more often.
    const int ulen = rs_[2] + ln_[2] - rs_[0];
    const bool chain = !grouping && ln_[0] > 0 && ln_[1] > 0 && ln_[2] > 0 && rs_[1] >= rs_[0] && rs_[1] <= rs_[0] + ln_[0] &&
                       rs_[2] >= rs_[1] && rs_[2] <= rs_[1] + ln_[1] && ulen >= ln_[2];
    if (qa == 0 && chain && ulen <= CAP && fg != 1 && fg != 5) {
        lo_[0] = lo_[1] = lo_[2] = -rs_[0];
        total = ulen;
    } else if (qa == 0 && ln_[0] + ln_[1] + ln_[2] <= CAP && fg != 1) {   // the usual case: the whole group fits
        lo_[0] = -rs_[0]; lo_[1] = ln_[0] - rs_[1]; lo_[2] = ln_[0] + ln_[1] - rs_[2];
        total = ln_[0] + ln_[1] + ln_[2];
    } else {   // the round is the longest prefix [qa, qb) of the runs not yet done that fits
        rm = 0u; qb = qa;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            lo_[q] = INT_MIN;
            if (q >= qa && q == qb && !overflow) {
                if (fg != 1 && total + ln_[q] <= CAP) { lo_[q] = total - rs_[q]; total += ln_[q]; qb = q + 1; rm |= 1u << q; }
                else if (q == qa) { overflow = true; qb = q + 1; rm |= 1u << q; }
            }
        }
    }
}
// LDS bytes of k_nbr_pass<P, MASKMODE> (tile + cell_start windows + small change)
// debug (build with -DSPH_TIMELINE, run with SPH_DEBUG_MODE=20): shader-clock stamps of workgroup phases, thread 0, slot k of 16 per workgroup
#ifdef SPH_TIMELINE
#define NBR_STAMP(k) do { if (timeline && threadIdx.x == 0) timeline[(size_t)blockIdx.x * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define NBR_STAMP(k) do { } while (0)
#endif
template <class P, int MASKMODE> constexpr int nbr_lds_bytes() {
    return (nbr_tile_cap<P>() + ((MASKMODE == 2 || pass_reuses_masks<P>()) ? 0 : NBR_PAD)) * (16 + (P::HAS_B ? (int)sizeof(typename P::BT) : 0) + (PassC<P>::value ? (int)sizeof(typename PassC<P>::type) : 0)) + 9 * NBR_CS_PITCH * 2 + 64;
}
// Second launch bound = minimum waves per SIMD = workgroups per CU.  Default 4 (<= 128 VGPRs).  The payload-free functors
// with a one- or two-word accumulator (density, rigid volume: 23.7 KB of LDS) run 5 (<= 96 VGPRs) in the fast build: with
// the 4-candidate phase-1 chunks and without SLP packing they fit without spilling, and the density pass of C2 went
// 141 -> 127 us (profiles/r02_ab_*.txt).  The functors with payload arrays are capped at 4 by their 33-38 KB tiles; the
// strict build and the 5-float DFSPH density+alpha accumulator spill at 96 and stay at 4 (tools/check_spills.py gates).
#ifndef SPH_NBR_WAVES_LIGHT
#define SPH_NBR_WAVES_LIGHT (SPH_FAST ? 6 : 4)
#endif
#ifndef SPH_NBR_WAVES_HEAVY
#define SPH_NBR_WAVES_HEAVY (SPH_FAST ? 4 : 3)   // strict build (IEEE division / sqrt sequences): 3, i.e. <= 168 VGPRs, rather than spills
#endif
// P::MAX_WAVES: a functor that runs rarely and would spill at its class's register budget asks for fewer waves (RigidVolumePass)
template <class P, class = void> struct PassMaxWaves { static constexpr int value = 8; };
template <class P> struct PassMaxWaves<P, decltype((void)P::MAX_WAVES)> { static constexpr int value = P::MAX_WAVES; };
template <class P, int MASKMODE> constexpr int nbr_waves_per_simd() {
    const int w = P::HAS_B ? (pass_is_medium<P>() ? 5 : SPH_NBR_WAVES_HEAVY) : (sizeof(typename P::Own) <= 8 ? SPH_NBR_WAVES_LIGHT : 4);
    return w < PassMaxWaves<P>::value ? w : PassMaxWaves<P>::value;
}
// P::SPLIT3 launches: gridDim.y = 3 -> workgroup (b, y) walks group y; gridDim.y = 2 -> (b, 0) walks groups 0 and 1, (b, 1) group 2
// (for grids that would not be resident at once three ways: 3 x 416 tiles of the buckling sheet = 1248 > 1024 = a second, mostly empty round)
__device__ __forceinline__ int split_lo(int ny, int y) { return ny == 3 ? y : (y == 0 ? 0 : 2); }
__device__ __forceinline__ int split_hi(int ny, int y) { return ny == 3 ? y + 1 : (y == 0 ? 2 : 3); }
template <class P, int MASKMODE>
__global__ void __launch_bounds__(P::BLOCK, (nbr_waves_per_simd<P, MASKMODE>()))
k_nbr_pass(const Consts c, const int *__restrict__ cell_start, const P p, DevScalars *__restrict__ scal,
           int nblocks, unsigned *__restrict__ nbr_mask, unsigned *__restrict__ nbr_mask_hi, int mask_stride,
           const int *__restrict__ blk_hdr, const unsigned char *__restrict__ lane_perm,
           unsigned long long *__restrict__ timeline, const int *__restrict__ stop_flag,
           const int *__restrict__ blk_list, const int *__restrict__ blk_count, const unsigned char *__restrict__ tile_skip) {
    // First round trip: everything that depends on nothing, requested together (the stop flag, the length of the list and this workgroup's
    // entry used to be three dependent scalar loads; the list has an entry per tile of the scene, so any blockIdx.x reads inside it).
    const int stop_now = stop_flag ? *stop_flag : 0;
    const int list_len = blk_list ? *blk_count : INT_MAX;
    const int b_listed = blk_list ? blk_list[blockIdx.x] : 0;
    if (stop_now) return;   // iteration launched past the convergence of a device-controlled loop
    if ((int)blockIdx.x >= list_len) {   // only the workgroups that hold fluid were listed
        // (a functor whose prologue keeps a solver loop's books gets it run by workgroup (0, 0) even when the list is EMPTY --
        //  an emitter scene before its first release has no active fluid particle at all)
        if constexpr (PassPrologue<P>::value) { if (blockIdx.x == 0 && blockIdx.y == 0) p.prologue(scal); }
        return;
    }
    constexpr int BLOCK = P::BLOCK;
    constexpr int CAP = nbr_tile_cap<P>();   // LDS particle slots per staging group
    constexpr int NS = (CAP + BLOCK - 1) / BLOCK;   // tile slots staged per thread
    constexpr int GROUPS = 3, RPG = 3;   // one x-offset (3 runs) staged at a time
    static_assert(P::GROUPS == 3 && BLOCK == 256, "lane permutation is stored as one byte per particle");
    typedef typename P::Own Own;
    typedef typename P::BT BT;
    constexpr int PAD = (MASKMODE == 2 || pass_reuses_masks<P>()) ? 0 : NBR_PAD;   // only the unrolled phase 1 reads past a run's end
    __shared__ float2 sT[2 * (CAP + PAD)];   // (x,y) slots followed by (z,w) slots: fixed byte distance
    float2 *const sXY = sT;
    float2 *const sZW = sT + (CAP + PAD);
    constexpr int ZW_OFF = (CAP + PAD) * 8;
    __shared__ BT sB[P::HAS_B ? CAP + PAD : 1];
    __shared__ typename PassC<P>::type sC[PassC<P>::value ? CAP + PAD : 1];
    __shared__ unsigned short s_cs[9][NBR_CS_PITCH];   // cell_start - run start
    __shared__ int s_loff[9];   // tile offset - run start of every run (INT_MIN: not staged)

    const int tid = threadIdx.x;
    NBR_STAMP(0);
    const int b = blk_list ? b_listed : xcd_remap(blockIdx.x, nblocks, c.xcd_chunk);
    const int i0 = b * BLOCK;
    // Second round trip, requested BEFORE a functor's prologue (whose barriers keep later loads behind its own: the CG walk's prologue adds
    // up two arrays of partial sums first): the lane permutation, the header, the skip flag of a slab's interior launch.
#ifndef SPH_NO_EARLY_LOADS
    const int who_early = lane_perm ? (int)lane_perm[i0 + tid] : tid;
    const int *hdr = blk_hdr + (size_t)b * BLK_HDR_INTS;
    const int cfirst = hdr[0], clast = hdr[1];
    const int skip_tile = tile_skip ? (int)tile_skip[b] : 0;
    const int n_live = live_n(c);
#endif
    if constexpr (PassWrench<P>::value) wrench_init_all(p.pose);   // (published by the prologue's barrier)
    if constexpr (PassPrologue<P>::value) { if (!p.prologue(scal)) return; }   // workgroup-uniform
#ifdef SPH_NO_EARLY_LOADS
    const int who_early = lane_perm ? (int)lane_perm[i0 + tid] : tid;
    const int *hdr = blk_hdr + (size_t)b * BLK_HDR_INTS;
    const int cfirst = hdr[0], clast = hdr[1];
    const int skip_tile = tile_skip ? (int)tile_skip[b] : 0;
    const int n_live = live_n(c);
#endif
    if (i0 >= n_live) {   // launch bound of an asynchronous slab step: no such tile (its header was never written)
        if constexpr (P::HAS_REDUCE) {
            if (tid == 0) {
                const int bs = red_slot(blk_list, b);
                if constexpr (PassSplit<P>::value) {
                    if (gridDim.y > 1) {
                        if (float *o = p.split_out(split_lo((int)gridDim.y, (int)blockIdx.y))) o[bs] = 0.0f;
                        // two-way split: nobody walks "part 1", but the consumers add up three parts (the normal path zeroes it, this one did not)
                        if (gridDim.y == 2 && blockIdx.y == 0) { if (float *o = p.split_out(1)) o[bs] = 0.0f; }
                    } else p.red_out[bs] = 0.0f;
                }
                else p.red_out[bs] = 0.0f;
            }
        }
        return;
    }
    // which particle of the workgroup this lane owns for the whole pass
    const int who = who_early;
    const int i = i0 + who;
    const bool valid = i < n_live;

    // workgroup header (uniform; requested above)
    const int span = clast - cfirst;
    bool cs_lds = span >= 0 && span <= NBR_CS_SPAN;
#pragma unroll
    for (int k = 0; k < 9; ++k) cs_lds = cs_lds && hdr[11 + k] < 65536;   // windows are cached as 16-bit offsets
    // In flight together, behind the one dependent load above (the lane permutation): own particle and what begin() reads of it,
    // the first staging round, the cell_start windows.  Straight-line code on a clamped index: a load inside a divergent `if (valid)`
    // gets an `s_waitcnt vmcnt(0)` at the end of its block (the merge of its result), which used to serialise the prologue into
    // five memory round trips (permutation -> position -> windows -> begin() -> first staging round).  begin() only reads.
    const int ic = valid ? i : i0;
#ifndef SPH_NO_CELL_WORD
    // the lane's cell word (k_block_prep), filed behind the headers of all tiles (mask_stride = particle capacity)
    const unsigned cw = reinterpret_cast<const unsigned *>(blk_hdr + (size_t)((mask_stride + 255) >> 8) * BLK_HDR_INTS)[ic];
#endif
    const float4 pi = p.posv[ic];
    Own own;
    bool active = p.begin(c, ic, pi, own) && valid;
    unsigned mk0[3] = {0u, 0u, 0u};   // first mask words of the first group's runs
    if (MASKMODE == 2 && pass_mask_pipe<P>()) {
        const int g0 = (PassSplit<P>::value && gridDim.y > 1) ? split_lo((int)gridDim.y, (int)blockIdx.y) : 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) mk0[q] = nbr_mask[(size_t)run_of(c.run_grouping, g0, q) * mask_stride + i];   // (the array has a tile of slack)
    }
    // -DSPH_PRESTAGE (off): the first staging round depends on the header alone, so its global loads can go out HERE and be written to
    // the tile after the prologue's barrier; the records sit in registers only across the prologue, where little else is live.  One
    // round trip less per workgroup -- and 1-2 % SLOWER at C2 / C3 (profiles/r03w_ab_round_trips.txt): kept as a switch, like PAIR2.
    typedef typename PassC<P>::type CT;
    const int g_first = (PassSplit<P>::value && gridDim.y > 1) ? split_lo((int)gridDim.y, (int)blockIdx.y) : 0;
#ifdef SPH_PRESTAGE
    constexpr bool PRESTAGE = SPH_FAST || sizeof(BT) < 16;   // (the strict build's widest functors sit at the 128-VGPR limit already)
#else
    constexpr bool PRESTAGE = false;
#endif
    float4 pa_[PRESTAGE ? NS : 1];
    BT pb_[PRESTAGE ? NS : 1];
    CT pc_[PRESTAGE ? NS : 1];
    int ptotal = 0;
    if constexpr (PRESTAGE) {
        int rs_[RPG], ln_[RPG], lo_[RPG], qb; unsigned rm; bool overflow;
        nbr_plan<CAP>(hdr, g_first, 0, c.force_global, rs_, ln_, lo_, ptotal, qb, rm, overflow, c.run_grouping);
        if (c.force_global == 10 || c.force_global == 11) ptotal = 0;
        const int n0 = lo_[0] != INT_MIN ? ln_[0] : 0;
        const int n01 = n0 + (lo_[1] != INT_MIN ? ln_[1] : 0);
        if (ptotal > 0) {   // uniform.  Slots past the end re-read the last record (same cache line) instead of branching:
#pragma unroll              // the whole prologue stays one basic block, and the scheduler puts every load ahead of the first wait
            for (int u = 0; u < NS; ++u) {
                int t = tid + u * BLOCK;
                t = t < ptotal ? t : ptotal - 1;
                const int j = t < n0 ? t - lo_[0] : (t < n01 ? t - lo_[1] : t - lo_[2]);
                pa_[u] = pass_stage(p, c, j, pb_[u], pc_[u]);
            }
        }
    }
    if (cs_lds && tid < NBR_CS_PITCH) {   // uniform per wave (a window has <= 128 entries: the upper two waves have nothing to fetch); threads past the window re-read its last entry
        const int tc = tid < span + 4 ? tid : span + 3;
        int cs_[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            int cell = cfirst + (k / 3 - 1) * c.ny * c.nz + (k % 3 - 1) * c.nz - 1 + tc;
            cell = cell < 0 ? 0 : (cell > c.G ? c.G : cell);
            cs_[k] = cell_start[cell];
        }
        if (tid < span + 4) {
#pragma unroll
            for (int k = 0; k < 9; ++k) s_cs[k][tid] = (unsigned short)(cs_[k] - hdr[2 + k]);
        }
    }
    // What the group loop keeps of this lane's cell: two window-cache entries and nine "column exists" bits (three registers instead of
    // cx, cy, z0, z1 - z0; the rare path that reads the windows from global memory recomputes the cell from the position).
    int e0, e1;
    unsigned dom = 0u;
#ifndef SPH_NO_CELL_WORD
    e0 = (int)(cw & 0xffu);
    e1 = e0 + (int)((cw >> 8) & 7u);
    dom = cw >> 11;
#else
    {
        const int cx = cell_coord_x(c, pi.x);
        const int cy = cell_coord(pi.y, c.grid_size, c.ny);
        const int cz = cell_coord_z(c, pi.z);
        const int lin = (cx * c.ny + cy) * c.nz + cz;
        const int z0 = cz > 0 ? cz - 1 : 0;
        const int z1 = cz < c.nz - 1 ? cz + 1 : c.nz - 1;
        e0 = (lin - cfirst) + (z0 - cz) + 1;   // s_cs entry of (.., .., z0) in every run
        e1 = e0 + (z1 - z0) + 1;
        // bit k = 3 (ox + 1) + (oy + 1): column (cx + ox, cy + oy) lies inside the grid.  The three y bits, copied to where the x offsets
        // that exist put them (nine range tests -> four)
        const unsigned by = (cy > 0 ? 1u : 0u) | 2u | (cy < c.ny - 1 ? 4u : 0u);
        dom = (cx > 0 ? by : 0u) | (by << 3) | (cx < c.nx - 1 ? by << 6 : 0u);
    }
#endif
    if (skip_tile) return;   // (uniform)
    if (__syncthreads_or(active ? 1 : 0)) {  // workgroup-uniform; also publishes s_cs
        NBR_STAMP(1);
        unsigned npairs = 0;
        // split launch (uniform): this workgroup walks the x-offset groups [g_lo, g_hi) only -- one of three, or {0, 1} / {2} of a two-way split
        const bool is_split = PassSplit<P>::value && gridDim.y > 1;
        const int g_lo = is_split ? split_lo((int)gridDim.y, (int)blockIdx.y) : 0;
        const int g_hi = is_split ? split_hi((int)gridDim.y, (int)blockIdx.y) : GROUPS;
        bool prestaged = false;   // the first round's records are already on their way (above)
        unsigned mkn[RPG] = {mk0[0], mk0[1], mk0[2]};   // first mask words of the coming round's group
        if constexpr (PRESTAGE) {
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                const int t = tid + u * BLOCK;
                if (t < ptotal) {
                    sXY[t] = make_float2(pa_[u].x, pa_[u].y);
                    sZW[t] = make_float2(pa_[u].z, pa_[u].w);
                    if (P::HAS_B) sB[t] = pb_[u];
                    if (PassC<P>::value) sC[t] = pc_[u];
                }
            }
            prestaged = true;
        }
#pragma unroll 1
        for (int g = g_lo, qa = 0; g < (c.force_global == 11 ? 0 : g_hi); ) {
            // One group = the three runs of an x offset.  Its runs are staged in ROUNDS (uniform plan): a round takes the longest
            // prefix of the runs not yet done that fits the tile, laid out back to back -- normally all three in one round; where
            // the fluid has piled up, two rounds (e.g. {0, 1} then {2}) instead of sending the group down the slow ordered walk.
            // A run that does not fit the tile on its own is a round of its own and goes through the tile in chunks.  Runs are consumed in
            // order, so the accumulation order stays the reference's.
            int rs_[RPG], ln_[RPG], lo_[RPG];
            int total, qb;
            unsigned rm;                 // runs of this round (bit q)
            bool overflow;
            nbr_plan<CAP>(hdr, g, qa, c.force_global, rs_, ln_, lo_, total, qb, rm, overflow, c.run_grouping);
#define NBR_IN_ROUND(q) ((rm >> (q)) & 1u)
            if (tid < RPG) {
                s_loff[g * RPG + tid] = tid == 0 ? lo_[0] : (tid == 1 ? lo_[1] : lo_[2]);
                if (overflow && tid == qa && (tid == 0 ? ln_[0] : (tid == 1 ? ln_[1] : ln_[2])) > 0) atomicAdd(&scal->fallback[c.stat_bank][b & (SPH_STAT_SLOTS - 1)], 1ull);
            }
            // Everything this round reads from global memory goes out in ONE round trip (round 3; it used to be up to five: a load
            // under a divergent `if` is waited for at the end of its block, and the batches of a wide record waited for each other):
            // the first mask words were requested one iteration ago (mkn), the staging loads follow here on clamped indices (slots
            // past the end re-read the last record), then the next round's mask words and the rare second mask words; the first
            // wait is the one in front of the tile writes.
            unsigned mk[RPG], mh[RPG] = {0u, 0u, 0u};
            if constexpr (pass_mask_pipe<P>() || MASKMODE != 2) {
#pragma unroll
                for (int q = 0; q < RPG; ++q) mk[q] = mkn[q];
            } else {
#pragma unroll
                for (int q = 0; q < RPG; ++q) mk[q] = nbr_mask[(size_t)run_of(c.run_grouping, g, q) * mask_stride + i];
            }
            // stage the runs that fit; consecutive t -> consecutive j: coalesced.  SB slots per batch: all of them where the
            // registers allow it (the medium functors' 96-VGPR budget and the strict build's wide records do not).
            constexpr int SB = pass_is_medium<P>() ? (NS > SPH_MEDIUM_SB ? SPH_MEDIUM_SB : NS)
                                                   : ((P::HAS_B && sizeof(BT) >= 16) ? ((PassUsesJ0<P>::value || !SPH_FAST) ? 2 : (SPH_HEAVY_SB < NS ? SPH_HEAVY_SB : NS)) : NS);
            const int n0 = lo_[0] != INT_MIN ? ln_[0] : 0;
            const int n01 = n0 + (lo_[1] != INT_MIN ? ln_[1] : 0);
            const bool stage_now = total > 0 && c.force_global != 10 && !prestaged;   // uniform
            float4 a_[SB];
            BT b_[SB];
            CT c_[SB];
            if (stage_now) {
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    int t = tid + u * BLOCK;
                    t = t < total ? t : total - 1;
                    const int j = t < n0 ? t - lo_[0] : (t < n01 ? t - lo_[1] : t - lo_[2]);
                    a_[u] = pass_stage(p, c, j, b_[u], c_[u]);
                }
            }
            if (MASKMODE == 2 && pass_mask_pipe<P>()) {   // first mask words of the NEXT round's group ([run][particle] layout)
                const int gn = qb >= RPG ? (g + 1 < GROUPS ? g + 1 : g) : g;
#pragma unroll
                for (int q = 0; q < RPG; ++q) mkn[q] = nbr_mask[(size_t)run_of(c.run_grouping, gn, q) * mask_stride + i];
            }
            // candidate sub-ranges of this lane's particle in the three runs (from the cached cell_start windows)
            bool inr[RPG];
            int js_[RPG], m_[RPG];
            bool wide = false, longrun = false;
#pragma unroll
            for (int q = 0; q < RPG; ++q) {
                const int k = run_of(c.run_grouping, g, q);
                inr[q] = NBR_IN_ROUND(q) && active && ((dom >> k) & 1u);
                js_[q] = 0; m_[q] = 0;
                if (inr[q]) {
                    if (cs_lds) {
                        // explicit LDS loads: a plain s_cs[k][e] gets merged with the global branch into ONE flat load
                        // through a generic pointer -- measured 200 us of a 280 us pass.
                        const int o0 = lds_ld_u16(&s_cs[k][e0]);
                        js_[q] = rs_[q] + o0;
                        m_[q] = lds_ld_u16(&s_cs[k][e1]) - o0;
                    } else {   // (a workgroup spanning more cells than the window cache holds: sparse spray)
                        float px = pi.x, py = pi.y, pz = pi.z;
                        asm volatile("" : "+v"(px), "+v"(py), "+v"(pz));   // keeps the recomputation IN this branch (it is loop-invariant: hoisted, it would hold four registers for every workgroup)
                        const int xx = cell_coord_x(c, px) + k / 3 - 1, yy = cell_coord(py, c.grid_size, c.ny) + k % 3 - 1;
                        const int cz = cell_coord_z(c, pz);
                        const int z0 = cz > 0 ? cz - 1 : 0;
                        const int z1 = cz < c.nz - 1 ? cz + 1 : c.nz - 1;
                        const int lin0 = (xx * c.ny + yy) * c.nz + z0;
                        js_[q] = cell_start[lin0];
                        m_[q] = cell_start[lin0 + (z1 - z0) + 1] - js_[q];
                    }
                    wide = wide || m_[q] > 32;
                    longrun = longrun || m_[q] > 64;
                }
                // second word only for the runs beyond 32 candidates; a run without candidates (or out of range, or of another
                // round, or of an inactive lane) may hold a stale first word
                if (MASKMODE == 2 && m_[q] > 32 && c.force_global != 12) mh[q] = nbr_mask_hi[(size_t)k * mask_stride + i];
                if (MASKMODE == 2 && (m_[q] <= 0 || c.force_global == 12)) mk[q] = 0u;
            }
            if (stage_now) {
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const int t = tid + u * BLOCK;
                    if (t < total) {
                        sXY[t] = make_float2(a_[u].x, a_[u].y);
                        sZW[t] = make_float2(a_[u].z, a_[u].w);
                        if (P::HAS_B) sB[t] = b_[u];
                        if (PassC<P>::value) sC[t] = c_[u];
                    }
                }
            }
            // the remaining batches of a record too wide to be staged at once
#pragma unroll 1
            for (int u0 = SB; u0 < NS && u0 * BLOCK < total && stage_now; u0 += SB) {
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const int t = tid + (u0 + u) * BLOCK;
                    if (t < total) {
                        const int j = t < n0 ? t - lo_[0] : (t < n01 ? t - lo_[1] : t - lo_[2]);
                        a_[u] = pass_stage(p, c, j, b_[u], c_[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const int t = tid + (u0 + u) * BLOCK;
                    if (t < total) {
                        sXY[t] = make_float2(a_[u].x, a_[u].y);
                        sZW[t] = make_float2(a_[u].z, a_[u].w);
                        if (P::HAS_B) sB[t] = b_[u];
                        if (PassC<P>::value) sC[t] = c_[u];
                    }
                }
            }
            prestaged = false;
            __syncthreads();
            NBR_STAMP(2 + g * 4);
            // the merged loop handles runs of <= 64 candidates out of the tile (one or two mask words per run); anything
            // else (tile overflow, a pile-up of > 64 particles in three cells, forced debug modes) walks its runs one by
            // one, wave-uniformly
            if (overflow) {
                // (workgroup-uniform) this round is ONE run, and it does not fit the tile: through the tile in chunks (process_chunk)
                const int k = run_of(c.run_grouping, g, qa);
                const int rsq = qa == 0 ? rs_[0] : (qa == 1 ? rs_[1] : rs_[2]);
                const int lnq = qa == 0 ? ln_[0] : (qa == 1 ? ln_[1] : ln_[2]);
                const bool in = qa == 0 ? inr[0] : (qa == 1 ? inr[1] : inr[2]);
                const int js = in ? (qa == 0 ? js_[0] : (qa == 1 ? js_[1] : js_[2])) : 0;
                const int m = in ? (qa == 0 ? m_[0] : (qa == 1 ? m_[1] : m_[2])) : 0;
                unsigned long long w01 = 0ull;
#pragma unroll 1
                for (int cs0 = 0; cs0 < lnq; cs0 += CAP) {
                    const int cl = lnq - cs0 < CAP ? lnq - cs0 : CAP;
                    for (int t = tid; t < cl; t += BLOCK) {
                        BT bj; CT cj;
                        const float4 a = pass_stage(p, c, rsq + cs0 + t, bj, cj);
                        sXY[t] = make_float2(a.x, a.y);
                        sZW[t] = make_float2(a.z, a.w);
                        if (P::HAS_B) sB[t] = bj;
                        if (PassC<P>::value) sC[t] = cj;
                    }
                    __syncthreads();
                    process_chunk<ZW_OFF, MASKMODE>(c, p, own, i, pi.x, pi.y, pi.z, js, js + m, rsq + cs0, rsq + cs0 + cl, sXY, sZW, sB, sC, npairs, w01);
                    __syncthreads();
                }
                if (MASKMODE == 1 && in) {
                    nbr_mask[(size_t)k * mask_stride + i] = (unsigned)w01;
                    if (m > 32) nbr_mask_hi[(size_t)k * mask_stride + i] = (unsigned)(w01 >> 32);
                }
            } else if (c.force_global == 4 || __any(longrun)) {
                // a pile-up of more than 64 candidates in three cells (or debug mode 4): the runs one by one, wave-uniformly, through the same
                // ordered walk that takes a run longer than the tile (process_chunk: the staged run is ONE chunk at tile offset lo_, masks
                // recomputed).  Until round 4 a second walk (process_run) did this with the stored masks; it is rare enough not to deserve
                // its registers.
#pragma unroll 1
                for (int q = 0; q < RPG; ++q) {
                    const int k = run_of(c.run_grouping, g, q);
                    const bool in = q == 0 ? inr[0] : (q == 1 ? inr[1] : inr[2]);
                    const int js = in ? (q == 0 ? js_[0] : (q == 1 ? js_[1] : js_[2])) : 0;
                    const int m = in ? (q == 0 ? m_[0] : (q == 1 ? m_[1] : m_[2])) : 0;
                    const int loff = q == 0 ? lo_[0] : (q == 1 ? lo_[1] : lo_[2]);
                    const int cs = loff == INT_MIN ? 0 : -loff;          // sorted index of tile slot 0 as this run sees it (thin grids: the three
                    const int ce = loff == INT_MIN ? 0 : cs + total;     // runs share one staged stretch); the lane's own [js, js + m) lies inside
                    unsigned long long w01 = 0ull;
                    process_chunk<ZW_OFF, MASKMODE>(c, p, own, i, pi.x, pi.y, pi.z, js, js + m, cs, ce, sXY, sZW, sB, sC, npairs, w01);
                    if (MASKMODE == 1 && in) {
                        nbr_mask[(size_t)k * mask_stride + i] = (unsigned)w01;
                        if (m > 32) nbr_mask_hi[(size_t)k * mask_stride + i] = (unsigned)(w01 >> 32);
                    }
                }
            } else {
                // acceptance masks of the three runs, then one merged loop
                const bool anywide = __any(wide);   // wave-uniform: some run in this wave has more than 32 candidates
                unsigned ab[RPG];
#pragma unroll
                for (int q = 0; q < RPG; ++q) {
                    const int k = run_of(c.run_grouping, g, q);
                    const int base = inr[q] ? js_[q] + lo_[q] : 0;
                    unsigned nm, nh = 0u;
                    if (MASKMODE == 2) {
                        nm = mk[q]; nh = mh[q];   // zero where the run is empty / has no second chunk (not loaded)
                    } else {
                        const int mlo = m_[q] < 32 ? m_[q] : 32;
                        constexpr bool LEAN1 = pass_reuses_masks<P>();
                        nm = c.force_global == 13 ? 0u : phase1_mask<ZW_OFF, LEAN1>(sXY, base, mlo, pi.x, pi.y, pi.z, c.h2);
                        if (anywide) nh = phase1_mask<ZW_OFF, LEAN1>(sXY, m_[q] > 32 ? base + 32 : 0, m_[q] > 32 ? m_[q] - 32 : 0, pi.x, pi.y, pi.z, c.h2);
                        const unsigned self = (unsigned)(i - js_[q]);
                        if (self < 32u) nm &= ~(1u << self);      // p_i != p_j (base_container.py:559)
                        else if (self < 64u) nh &= ~(1u << (self - 32u));
                        if (MASKMODE == 1 && inr[q]) {
                            nbr_mask[(size_t)k * mask_stride + i] = nm;
                            if (m_[q] > 32) nbr_mask_hi[(size_t)k * mask_stride + i] = nh;
                        }
                    }
                    mk[q] = nm; mh[q] = nh;
                    ab[q] = (unsigned)base << 3;
                    npairs += __popc(nm) + __popc(nh);
                }
                NBR_STAMP(3 + g * 4);
                if (c.force_global < 9 || c.force_global >= 20) {
                    if (anywide) {   // 64-bit masks: same loop, same order, a few more VALU per iteration
                        typedef unsigned long long u64;
                        merged_phase2<P, ZW_OFF, u64>(c, p, own, pi.x, pi.y, pi.z, (u64)mk[0] | ((u64)mh[0] << 32), (u64)mk[1] | ((u64)mh[1] << 32),
                                                      (u64)mk[2] | ((u64)mh[2] << 32), ab[0], ab[1], ab[2], sXY, sB, sC, &s_loff[g * RPG]);
                    } else {
                        merged_phase2<P, ZW_OFF, unsigned>(c, p, own, pi.x, pi.y, pi.z, mk[0], mk[1], mk[2], ab[0], ab[1], ab[2], sXY, sB, sC, &s_loff[g * RPG]);
                    }
                }
            }
            NBR_STAMP(4 + g * 4);
            __syncthreads();  // LDS is restaged by the next round / group
            NBR_STAMP(5 + g * 4);
            qa = qb;
            if (qa >= RPG) { qa = 0; ++g; }
        }
        if constexpr (PassWrench<P>::value) { __syncthreads(); wrench_flush_all(scal); }   // six atomics per body and workgroup that touched it
        if (P::COUNT_PAIRS) {
            // (P::STAT_W: the functor carries its weights -- the WCSPH density pass also books the fused force pass, which walks the same
            //  accepted pairs out of the stored masks and does not count them again)
            unsigned long long wp = P::PAIR_WEIGHT, we = 1ull;
            if constexpr (PassStatW<P>::value) { wp = (unsigned long long)p.stat_pairs; we = (unsigned long long)p.stat_evals; }
            float fp = wave_sum((float)npairs);  // <= 64 * few hundred: exact in f32
            if ((tid & 63) == 0 && fp > 0.0f) {
                // two 64-bit tallies per slot: pairs weighted by the reference passes this walk stands for (SURVEY 8d
                // metric) and pairs as evaluated.  (They used to share one word, 32 bits each: hundreds of solver
                // iterations over ~2^28 particles could carry from one into the other.)
                const int slot = (b * (BLOCK / 64) + (tid >> 6)) & (SPH_STAT_SLOTS - 1);
                atomicAdd(&scal->pairs[c.stat_bank][slot], (unsigned long long)fp * wp);
                atomicAdd(&scal->evals[c.stat_bank][slot], (unsigned long long)fp * we);
            }
        }
    }
    NBR_STAMP(14);
    float red = 0.0f;
    bool split_launch = false;
    float *red_to = nullptr;   // where this workgroup's partial sum goes (split launch: the functor may keep one per x-offset group)
    if constexpr (P::HAS_REDUCE) red_to = p.red_out;
    if constexpr (PassSplit<P>::value) {
        split_launch = gridDim.y > 1;
        const int part = split_lo((int)gridDim.y, (int)blockIdx.y);   // the part of the sum this workgroup leaves: its first group's
        if (split_launch && valid && active) red = p.partial(c, i, part, own);
        if (split_launch) red_to = p.split_out(part);
        // two-way split: nobody walks "part 1" (its group rides in part 0), but the consumers add up three parts: zeros
        if (split_launch && gridDim.y == 2 && blockIdx.y == 0) {
            if (valid && active) p.partial_zero(i, 1);
            if (tid == 0) { if (float *o = p.split_out(1)) o[red_slot(blk_list, b)] = 0.0f; }
        }
    }
    if (valid && !split_launch) {
        if (active) red = p.finish(c, i, pi, own);
        else p.passive(c, i, pi);
    }
    NBR_STAMP(15);
    if constexpr (PassNextHash<P>::value) {
        if (p.nh.on) {   // (uniform) this pass is the next step's k_hash_count: cell ids, histogram, arrival ranks from the positions just stored
            // The lanes own the tile's particles in permuted order; the histogram wants them in SORTED order, where the particles of a cell
            // are neighbours and one atomic serves a whole run: through LDS (the tile is dead: barrier at the end of the last group).
            int *const s_lin = reinterpret_cast<int *>(sT);
            __syncthreads();   // (workgroups that skipped the group loop have no barrier behind their last tile access)
            s_lin[who] = (valid && active) ? own.lin_new : -1 - who;   // all-fluid, unsharded: every valid particle is active
            __syncthreads();
            const int lane = tid & 63;
            const bool v2 = i0 + tid < n_live;
            const int lin = v2 ? s_lin[tid] : -1 - lane;
            bool head; int hl, len;
            wave_runs(lin, lane, head, hl, len);
            int base = 0;
            if (head && v2) base = atomicAdd(&p.nh.cell_count[lin], len);
            const int base_run = __shfl(base, hl, 64);   // (the wait for the atomic's answer sits here, in front of the record's store -- vmcnt counts stores too)
            if (p.nh.rl.head && head && v2) run_list_file(p.nh.rl, lin, i0 + tid, len, base);
            if (v2) { p.nh.cellid[i0 + tid] = lin; p.nh.rank[i0 + tid] = base_run + (lane - hl); }
            if (p.nh.tile_sum) tile_sum_add(p.nh.tile_sum, lin, v2, c.G);   // (unsharded only: no graveyard cells)
        }
    }
    if (split_launch && !red_to) return;   // uniform; the combining kernel reduces
    if constexpr (P::HAS_REDUCE) {
        // deterministic per-workgroup partial sum: particle order and a fixed tree (whatever the lane permutation),
        // finished by k_reduce_partials
        float *const s_val = reinterpret_cast<float *>(sT);   // the tile is dead by now (barrier at the end of the last group)
        float *const s_red = s_val + BLOCK;
        __syncthreads();   // workgroups that skipped the group loop have no barrier behind their last tile access either
        s_val[who] = red;
        __syncthreads();
        const float w = wave_sum(s_val[tid]);
        if ((tid & 63) == 0) s_red[tid >> 6] = w;
        __syncthreads();
        if (tid == 0) {
            float t = 0.0f;
#pragma unroll
            for (int k = 0; k < BLOCK / 64; ++k) t += s_red[k];
            red_to[red_slot(blk_list, b)] = t;
        }
    }
}

// sums partial[0..n) in a fixed order into scal->red[slot].  Inside a device-controlled loop (kind != 0) it also counts
// the iteration and raises the stop flag when the solver's criterion is met (DFSPH.py:150/:239, PCISPH.py:122).
__global__ void __launch_bounds__(256)
k_reduce_partials(const float *__restrict__ partial, int n, DevScalars *__restrict__ scal, int slot, int kind,
                  float denom, double thr, const int *__restrict__ blk_list, const int *__restrict__ blk_count) {
    if (kind && scal->flags[0]) return;
    __shared__ float s_w[4];
    float t = 0.0f;
    if (blk_list) { const int m = *blk_count; for (int k = threadIdx.x; k < m; k += 256) t += partial[k]; }   // the pass ran the listed tiles only and filed their sums in list order (red_slot)
    else for (int k = threadIdx.x; k < n; k += 256) t += partial[k];
    t = wave_sum(t);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float sum = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
        scal->red[slot] = sum;
        if (kind) {
            const float avg = denom > 0.0f ? sum / denom : 0.0f;
            scal->flags[1] += 1;
            if (kind == 1 ? ((double)avg <= thr) : (avg < (float)thr)) scal->flags[0] = 1;
        }
    }
}
