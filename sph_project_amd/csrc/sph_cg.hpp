// sph_cg.hpp -- implicit viscosity (base_solver.py:282-517): matrix-free, block-Jacobi-scaled CG
// over the neighbour graph.  Included inside the per-build namespace.
#pragma once
#include "sph_passes.hpp"

__device__ __forceinline__ void mat3_inverse(const float *a, float *o) {
    // ti.math.inverse for 3x3 (cofactor form, 1/det first) -- same formula as oracle/sph_ref.c m3_inverse
    const float det = a[0] * (a[4] * a[8] - a[7] * a[5]) - a[3] * (a[1] * a[8] - a[7] * a[2]) + a[6] * (a[1] * a[5] - a[4] * a[2]);
    const float inv = 1.0f / det;
#define E(x, y) a[((x) % 3) * 3 + ((y) % 3)]
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            o[j * 3 + i] = inv * (E(i + 1, j + 1) * E(i + 2, j + 2) - E(i + 2, j + 1) * E(i + 1, j + 2));
#undef E
}

// base_solver.py:282 prepare_conjugate_gradient_solver1 (+tasks :326, :334, :349)
// Bytes / particle: R posv 16 + velm 16 + rho 4 + cg_x 16 -> W dinv 36 + b 16 + p 16 + x 16 + v0 16 (+ zero fills).
template <bool AF>
struct CgPreparePass {
    // Active for fluid only and passive() empty (round 5): launched over the fluid-holding tiles alone.  It used to zero the five solver
    // vectors of every OTHER particle -- 2.07 M boundary particles x 80 B per step in the buckling scene, for entries nothing reads: the
    // A p walk skips rigid neighbours before it looks at their search direction (CgApPass::pair), the vector kernels test is_fluid, a
    // ghost's search direction comes from its owner before every walk.  Those entries now keep what they held (zero from the
    // allocation, or a former fluid occupant's last value -- the solver state is slot-indexed and not reordered by the sort anyway).
    static constexpr bool FLUID_BLOCKS_ONLY = true;
    static constexpr int BLOCK = 256, GROUPS = 3;
    static constexpr bool USES_J = !AF;   // pair() looks at j only for rigid neighbours
    static constexpr bool HAS_B = true, COUNT_PAIRS = true, HAS_REDUCE = false;
    static constexpr int PAIR_WEIGHT = 2;  // A_ii pass + b_i pass of the reference
    typedef float4 BT;
    struct Own { float m, rho; float a[9]; float bx, by, bz; };
    const float4 *posv, *velm; const int *meta; const float *rho;
    float4 *cg_x, *cg_p, *cg_b, *cg_r, *cg_Ap, *cg_v0; float *dinv; float rho0; float *red_out;

    __device__ float4 stage_impl(int j, BT &bj) const {
        const float4 p = posv[j], v = velm[j];
        if (AF) { bj = make_float4(v.x, v.y, v.z, rho[j]); return make_float4(p.x, p.y, p.z, v.w); }
        const bool fl = META_MAT(meta[j]) == 1;
        bj = make_float4(v.x, v.y, v.z, fl ? rho[j] : -1.0f);
        return make_float4(p.x, p.y, p.z, fl ? v.w : p.w);  // fluid: mass, rigid: rest volume
    }
    __device__ float4 loadA(int j) const { BT b; return stage_impl(j, b); }
    __device__ BT loadB(int j) const { BT b; stage_impl(j, b); return b; }
    __device__ float4 stage(const Consts &, int j, BT &bj) const { return stage_impl(j, bj); }
    __device__ bool begin(const Consts &c, int i, const float4 &, Own &o) const {
        bool ok = true;   // no early return: begin()'s loads go out with the rest of the prologue (k_nbr_pass)
        if (!AF || c.ghosts) ok = META_ACTIVE_FLUID(meta[i]);
        o.m = velm[i].w; o.rho = rho[i];
#pragma unroll
        for (int k = 0; k < 9; ++k) o.a[k] = 0.0f;
        o.bx = o.by = o.bz = 0.0f;
        return ok;
    }
    __device__ void pair(const Consts &c, Own &o, float dx, float dy, float dz, float r2, const float4 &a,
                         const BT &bj, int) const {
        float g[3];
        kernGrad(c, dx, dy, dz, geom(c, r2), g[0], g[1], g[2]);
        const float R[3] = {dx, dy, dz};
        float s;
        if (AF || bj.w >= 0.0f) {
            const float m_ij = (o.m + a.w) * 0.5f;
            s = fdiv2(-c.cv * m_ij, bj.w, r2 + c.visc_eps);
        } else {
            const float m_ij = c.rho0 * a.w;
            s = fdiv2(-c.cvb * m_ij, o.rho, r2 + c.visc_eps);
            const float cb = fdiv(fdiv(c.cvb * c.rho0 * a.w, o.rho) * (bj.x * dx + bj.y * dy + bj.z * dz), r2 + c.visc_eps);
            o.bx += cb * g[0]; o.by += cb * g[1]; o.bz += cb * g[2];
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = 0; q < 3; ++q) o.a[p * 3 + q] -= s * (g[p] * R[q]);
    }
    __device__ float finish(const Consts &c, int i, const float4 &, Own &o) const {
        float d[9], inv[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) d[k] = ((k % 4 == 0) ? 1.0f : 0.0f) - fdiv(o.a[k] * c.dt, c.rho0);
        mat3_inverse(d, inv);
#pragma unroll
        for (int k = 0; k < 9; ++k) dinv[(size_t)i * 9 + k] = inv[k];
        const float4 v = velm[i];
        float4 x = cg_x[i];
        x.x += v.x; x.y += v.y; x.z += v.z;                     // :293 initial guess
        cg_x[i] = x;
        cg_v0[i] = make_float4(v.x, v.y, v.z, 0.f);             // :298
        cg_b[i] = make_float4(v.x - fdiv(c.dt * o.bx, c.rho0), v.y - fdiv(c.dt * o.by, c.rho0), v.z - fdiv(c.dt * o.bz, c.rho0), 0.f);
        cg_p[i] = x;                                            // :315
        cg_r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        cg_Ap[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        return 0.0f;
    }
    __device__ void passive(const Consts &, int, const float4 &) const {}
};

// ---- one CG iteration in three launches (A p pass, x / r update, p update) instead of six: the kernels that consume a dot
// product finish the reduction themselves -- every workgroup adds up the per-workgroup partials in the same fixed order, so
// all of them hold the same alpha / beta -- and workgroup 0 of the p update does the loop's book-keeping (error, iteration
// count, stop flag).  Partials: rr[2] (|r|^2, ping-pong: the x / r update reads one and writes the other), den (p . Ap, from
// the A p pass), rold (|r|^2 before the update, the denominator of beta).
// (stride3 != 0: `part` is three arrays stride3 apart -- the per-group partials of a split A p walk -- added up entry by entry)
// A launch over the list of fluid-holding tiles files the partial of its k-th LISTED workgroup at slot k (red_slot, sph_device.hpp):
// the sum runs over the first *blk_count entries, in list order -- the order it always had -- without the list itself (until round 5
// every reader went count -> list -> partial: three dependent round trips at the top of every workgroup of the next kernel).
__device__ __forceinline__ float cg_total(const float *part, int nb, const int *blk_list, const int *blk_count, float *s4, int stride3 = 0) {
    float a = 0.f;
    const int m = blk_list ? *blk_count : nb;
    if (stride3) {
        const float *p1 = part + stride3, *p2 = part + 2 * (size_t)stride3;
        for (int k = threadIdx.x; k < m; k += 256) a += (part[k] + p1[k]) + p2[k];
    }
    else for (int k = threadIdx.x; k < m; k += 256) a += part[k];
    a = wave_sum(a);
    __syncthreads();   // s4 may still be read from a previous call
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = a;
    __syncthreads();
    return (s4[0] + s4[1]) + (s4[2] + s4[3]);
}
// two sums in one sweep: the loads of both arrays are in flight together and the workgroup meets at two barriers instead of four
// (same per-array order of additions as two cg_total calls: bit-identical results)
__device__ __forceinline__ void cg_total2(const float *pa, const float *pb, int nb, const int *blk_list, const int *blk_count, float *s8,
                                          float &ra, float &rb, int stride3_b = 0) {
    float a = 0.f, b = 0.f;
    const int m = blk_list ? *blk_count : nb;
    if (stride3_b) {
        const float *p1 = pb + stride3_b, *p2 = pb + 2 * (size_t)stride3_b;
        for (int k = threadIdx.x; k < m; k += 256) { a += pa[k]; b += (pb[k] + p1[k]) + p2[k]; }
    }
    else for (int k = threadIdx.x; k < m; k += 256) { a += pa[k]; b += pb[k]; }
    a = wave_sum(a); b = wave_sum(b);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { s8[threadIdx.x >> 6] = a; s8[4 + (threadIdx.x >> 6)] = b; }
    __syncthreads();
    ra = (s8[0] + s8[1]) + (s8[2] + s8[3]);
    rb = (s8[4] + s8[5]) + (s8[6] + s8[7]);
}

// base_solver.py:374 compute_Ap (+task :386)
// Bytes / particle: R posv 16 + m 4 + rho 4 + dinv 36 + p 16 -> W Ap 16.
template <bool AF>
struct CgApPass {
    static constexpr bool FLUID_BLOCKS_ONLY = true;   // active for fluid only, passive() empty
    static constexpr int BLOCK = 256, GROUPS = 3;
    static constexpr bool USES_J = false;  // pair() never looks at j (rigid neighbours are skipped: they enter D_ii and b only)
    // (tried in round 3: 5 workgroups per CU for this 32-byte-record walk, so that all 3 x 416 split workgroups of the buckling sheet are
    //  resident at once instead of 1024 + a mostly empty second round: cg_ap 24.3 -> 25.6 us, C5 slower -- the smaller tile and the
    //  96-register budget cost more than the second round; P::MEDIUM_OK stays available in sph_device.hpp)
    static constexpr bool HAS_B = true, COUNT_PAIRS = true;
    static constexpr bool HAS_REDUCE = true;   // per-workgroup partial of p . Ap (the denominator of alpha, :394): no separate dot kernel
    static constexpr int PAIR_WEIGHT = 1;
    static constexpr bool SPLIT3 = true;       // see PassSplit / k_cg_ap_combine
    static constexpr bool HAS_PROLOGUE = true;
    typedef float4 BT;
#if SPH_FAST
    struct Own { float m; float x, y, z; };
#else
    struct Own { float m; float d[9]; float x, y, z; };
#endif
    const float4 *posv, *velm; const int *meta; const float *rho; const float4 *cg_p; const float *dinv;
    float4 *cg_Ap; float *red_out;
    float4 *part; int part_stride;             // [3][part_stride] partial sums of a split launch
    // fuse != 0: update_p (base_solver.py:434) of the PREVIOUS iteration happens here, on the fly: p = r + beta p_old for every
    // staged neighbour and for the own particle (written to p_out: other workgroups still read p_old), beta = |r_new|^2 / |r_old|^2
    // reduced by every workgroup from the x / r update's per-workgroup partials (fixed order: all hold the same value) -- one
    // launch per CG iteration less.  Workgroup (0, 0) also keeps the loop's books for that previous iteration (:445 `while tol > 1e-6`).
    const float4 *cg_r; float4 *p_out; const float *part_num, *part_den; int nb_part; const int *pl_list, *pl_count;
    int fuse, looped; float tol;
    mutable float beta;
    __device__ bool prologue(DevScalars *scal) const {
        beta = 0.0f;
        if (!fuse) return true;
        __shared__ float s8[8];
        float num, den;
        cg_total2(part_num, part_den, nb_part, pl_list, pl_count, s8, num, den);
        beta = den > 1e-18f ? num / den : 0.0f;
        const float err = __builtin_sqrtf(num);
        const bool done = looped && !(err > tol);
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
            scal->red[5] = beta; scal->red[3] = err;   // cg_beta, cg_error
            if (done) scal->flags[0] = 1;              // the kernels behind this one (and later iterations) see it at their top
        }
        return !done;
    }
    __device__ float4 own_p(int i) const {
        float4 q = cg_p[i];
        if (fuse) { const float4 r = cg_r[i]; q = make_float4(r.x + beta * q.x, r.y + beta * q.y, r.z + beta * q.z, 0.f); p_out[i] = q; }
        return q;
    }
    // fast build: the pair loop accumulates s = sum t (grad W) and D^-1 is applied to the sum afterwards (linear: D^-1 sum t g = sum t D^-1 g);
    // nine registers less in the loop than carrying D^-1 through it, which is what lets the pass run 5 workgroups per CU
    __device__ void apply_dinv(int i, float &x, float &y, float &z) const {
#if SPH_FAST
        const float *d = dinv + (size_t)i * 9;
        const float sx = x, sy = y, sz = z;
        x = d[0] * sx + d[1] * sy + d[2] * sz;
        y = d[3] * sx + d[4] * sy + d[5] * sz;
        z = d[6] * sx + d[7] * sy + d[8] * sz;
#endif
    }
    // Split launch (one workgroup per tile and x-offset group): this group's part of the sum, and -- part_dot set: the unsharded loop with
    // the fused p update -- its share of p . A p, so that no combining kernel is needed between the walk and the x / r update:
    //   A p = dt / rho0 (part_0 + part_1 + part_2) + p   =>   p . A p = sum_g p . (dt / rho0 part_g)  +  |p|^2   (the latter counted by group 0);
    // the x / r update adds the three parts up on the fly (k_cg_update_xr2).  Two launches per CG iteration instead of three.
    float *part_dot;   // [3][dot_stride] per-workgroup partials of the split walks, or null
    int dot_stride;
    __device__ float *split_out(int g) const { return part_dot ? part_dot + (size_t)g * dot_stride : nullptr; }
    __device__ void partial_zero(int i, int g) const { part[(size_t)g * part_stride + i] = make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ float partial(const Consts &c, int i, int g, const Own &o) const {
        float x = o.x, y = o.y, z = o.z;
        apply_dinv(i, x, y, z);
        part[(size_t)g * part_stride + i] = make_float4(x, y, z, 0.f);
        if (!part_dot) return 0.0f;
        const float4 p = own_p(i);
        const float ax = fdiv(x * c.dt, c.rho0), ay = fdiv(y * c.dt, c.rho0), az = fdiv(z * c.dt, c.rho0);
        float d = p.x * ax + p.y * ay + p.z * az;
        if (g == 0) d += p.x * p.x + p.y * p.y + p.z * p.z;
        return d;
    }

    __device__ float4 stage_impl(int j, BT &bj) const {
        const float4 p = posv[j];
        float4 q = cg_p[j];
        if (fuse) { const float4 r = cg_r[j]; q.x = r.x + beta * q.x; q.y = r.y + beta * q.y; q.z = r.z + beta * q.z; }
        const float m = velm[j].w;
        const bool fl = AF || META_MAT(meta[j]) == 1;
        // a rigid particle or a ghost of one carries no solver vectors (CgPreparePass runs over the fluid tiles only and its passive() is
        // empty, round 5): whatever cg_p / cg_r hold at its slot -- stale values of an earlier occupant -- must never reach the arithmetic.
        // pair() drops such a neighbour by bj.w < 0; its search direction is staged as zero so that nothing depends on that alone.
        bj = fl ? make_float4(q.x, q.y, q.z, rho[j]) : make_float4(0.f, 0.f, 0.f, -1.0f);
        return make_float4(p.x, p.y, p.z, m);
    }
    __device__ float4 loadA(int j) const { BT b; return stage_impl(j, b); }
    __device__ BT loadB(int j) const { BT b; stage_impl(j, b); return b; }
    __device__ float4 stage(const Consts &, int j, BT &bj) const { return stage_impl(j, bj); }
    __device__ bool begin(const Consts &c, int i, const float4 &, Own &o) const {
        bool ok = true;   // no early return: begin()'s loads go out with the rest of the prologue (k_nbr_pass)
        if (!AF || c.ghosts) ok = META_ACTIVE_FLUID(meta[i]);
        o.m = velm[i].w;
#if !SPH_FAST
#pragma unroll
        for (int k = 0; k < 9; ++k) o.d[k] = dinv[(size_t)i * 9 + k];
#endif
        o.x = o.y = o.z = 0.0f;
        return ok;
    }
    __device__ void pair(const Consts &c, Own &o, float dx, float dy, float dz, float r2, const float4 &a,
                         const BT &bj, int) const {
        if (!AF && bj.w < 0.0f) return;
        float g[3];
        kernGrad(c, dx, dy, dz, geom(c, r2), g[0], g[1], g[2]);
        const float m_ij = (o.m + a.w) * 0.5f;
        const float s = fdiv2(-c.cv * m_ij, bj.w, r2 + c.visc_eps);
        const float R[3] = {dx, dy, dz};
#if SPH_FAST
        // (-A) = -s g R^T  =>  Dinv (-A) p = -s (R.p) (Dinv g)
        const float t = -s * (R[0] * bj.x + R[1] * bj.y + R[2] * bj.z);
        o.x += t * g[0]; o.y += t * g[1]; o.z += t * g[2];   // D^-1 applied to the sum (apply_dinv)
#else
        // M = Dinv (-A), x += M p, with the roundings of the literal form (nA = -(s (g R^T)); M = Dinv nA; M p summed left
        // to right) but column by column, so that only one column of nA and of M is alive at a time (the 18 temporaries
        // of the literal form made this instantiation spill)
        const float bq[3] = {bj.x, bj.y, bj.z};
        float t[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float n0 = -(s * (g[0] * R[q])), n1 = -(s * (g[1] * R[q])), n2 = -(s * (g[2] * R[q]));
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const float M = o.d[p * 3] * n0 + o.d[p * 3 + 1] * n1 + o.d[p * 3 + 2] * n2;
                t[p] = q == 0 ? M * bq[0] : t[p] + M * bq[q];
            }
        }
        o.x += t[0]; o.y += t[1]; o.z += t[2];
#endif
    }
    __device__ float finish(const Consts &c, int i, const float4 &, Own &o) const {
        const float4 p = own_p(i);
        apply_dinv(i, o.x, o.y, o.z);
        float x = o.x * c.dt, y = o.y * c.dt, z = o.z * c.dt;
        x = fdiv(x, c.rho0); y = fdiv(y, c.rho0); z = fdiv(z, c.rho0);
        const float4 a = make_float4(x + p.x, y + p.y, z + p.z, 0.f);
        cg_Ap[i] = a;
        return p.x * a.x + p.y * a.y + p.z * a.z;
    }
    __device__ void passive(const Consts &, int, const float4 &) const {}
};

// ---- per-particle vector kernels with fixed-order block reductions -------------------------------
__device__ __forceinline__ void block_sum2(float a, float b, float *out_a, float *out_b, int blk) {   // blk: partial-sum slot, < 0: none
    __shared__ float s_a[4], s_b[4];
    a = wave_sum(a); b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = a; s_b[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0 && blk >= 0) {
        out_a[blk] = (s_a[0] + s_a[1]) + (s_a[2] + s_a[3]);
        out_b[blk] = (s_b[0] + s_b[1]) + (s_b[2] + s_b[3]);
    }
}

// slab sharding: ghosts are somebody else's rows of the system (launchers pass all_fluid = 0 when there are ghosts)
__device__ __forceinline__ bool is_fluid(const int *meta, int i, int all_fluid) { return all_fluid || META_ACTIVE_FLUID(meta[i]); }
// workgroup index of a vector kernel: the k-th listed workgroup when only those that hold fluid are launched (-1: none)
__device__ __forceinline__ int cg_block(const int *blk_list, const int *blk_count) {
    if (!blk_list) return blockIdx.x;
    return (int)blockIdx.x < *blk_count ? blk_list[blockIdx.x] : -1;
}

// :318 prepare_conjugate_gradient_solver2
__global__ void __launch_bounds__(256)
k_cg_prepare2(int n, const int *meta, int all_fluid, const float *dinv, const float4 *b, const float4 *Ap, float4 *r, float4 *p) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !is_fluid(meta, i, all_fluid)) return;
    const float *d = dinv + (size_t)i * 9;
    const float4 bb = b[i], a = Ap[i];
    const float4 rr = make_float4((d[0] * bb.x + d[1] * bb.y + d[2] * bb.z) - a.x, (d[3] * bb.x + d[4] * bb.y + d[5] * bb.z) - a.y,
                                  (d[6] * bb.x + d[7] * bb.y + d[8] * bb.z) - a.z, 0.f);
    r[i] = rr; p[i] = rr;
}

// partial sums of |r|^2 (and p.Ap): run once per solve, for the numerator of the first alpha (:394)
__global__ void __launch_bounds__(256)
k_cg_dots(int n, const int *meta, int all_fluid, const float4 *r, const float4 *p, const float4 *Ap, float *part_a, float *part_b,
          const int *stop_flag, const int *blk_list, const int *blk_count) {
    if (stop_flag && *stop_flag) return;
    const int blk = cg_block(blk_list, blk_count);
    if (blk < 0) return;
    int i = blk * 256 + threadIdx.x;
    float num = 0.f, den = 0.f;
    if (i < n && is_fluid(meta, i, all_fluid)) {
        const float4 rr = r[i], pp = p[i], a = Ap[i];
        num = rr.x * rr.x + rr.y * rr.y + rr.z * rr.z;
        den = pp.x * a.x + pp.y * a.y + pp.z * a.z;
    }
    block_sum2(num, den, part_a, part_b, red_slot(blk_list, blk));
}

// second half of a split A p pass (CgApPass::SPLIT3): adds the three per-group parts, then CgApPass::finish + its partial of p . Ap
__global__ void __launch_bounds__(256)
k_cg_ap_combine(const Consts c, const int *meta, int all_fluid, const float4 *part, int stride, const float4 *p, float4 *Ap,
                float *part_den, const int *stop_flag, const int *blk_list, const int *blk_count, const float4 *r_fuse, float4 *p_out,
                const DevScalars *scal) {
    if (stop_flag && *stop_flag) return;
    const int blk = cg_block(blk_list, blk_count);
    if (blk < 0) return;
    const int i = blk * 256 + threadIdx.x;
    float dot = 0.f;
    if (i < live_n(c) && is_fluid(meta, i, all_fluid)) {
        const float4 a0 = part[i], a1 = part[(size_t)stride + i], a2 = part[2 * (size_t)stride + i];
        float4 pp = p[i];
        if (r_fuse) {   // fused p update (CgApPass::fuse): beta was reduced and published by the A p pass in front of this kernel
            const float beta = scal->red[5];
            const float4 r = r_fuse[i];
            pp = make_float4(r.x + beta * pp.x, r.y + beta * pp.y, r.z + beta * pp.z, 0.f);
            p_out[i] = pp;
        }
        float x = ((a0.x + a1.x) + a2.x) * c.dt, y = ((a0.y + a1.y) + a2.y) * c.dt, z = ((a0.z + a1.z) + a2.z) * c.dt;
        x = fdiv(x, c.rho0); y = fdiv(y, c.rho0); z = fdiv(z, c.rho0);
        const float4 a = make_float4(x + pp.x, y + pp.y, z + pp.z, 0.f);
        Ap[i] = a;
        dot = pp.x * a.x + pp.y * a.y + pp.z * a.z;
    }
    __shared__ float s_d[4];
    dot = wave_sum(dot);
    if ((threadIdx.x & 63) == 0) s_d[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) part_den[red_slot(blk_list, blk)] = (s_d[0] + s_d[1]) + (s_d[2] + s_d[3]);
}

// Slab sharding: the dot products are sums over all ranks.  One workgroup adds up this rank's partials into out[0..1]; the
// step orchestration all-reduces the two floats in place (slab_allreduce_dev) and the update kernels take them from there
// (`glob`) instead of adding up the partials themselves.
__global__ void __launch_bounds__(256)
k_cg_fold(int nb, const float *part_a, const float *part_b, float *out_a, float *out_b, const int *stop_flag, const int *blk_list,
          const int *blk_count) {
    if (stop_flag && *stop_flag) return;
    __shared__ float s4[4];
    if (part_a) { const float a = cg_total(part_a, nb, blk_list, blk_count, s4); if (threadIdx.x == 0) *out_a = a; }
    if (part_b) { const float b = cg_total(part_b, nb, blk_list, blk_count, s4); if (threadIdx.x == 0) *out_b = b; }
}

// fused p update: the stop test of an iteration is otherwise made by the NEXT A p pass; at the end of a batch of launches this one
// workgroup makes it, so that the host's flag read-back sees a convergence reached in the batch's last iteration
__global__ void __launch_bounds__(256)
k_cg_check(int nb, const float *part_rr, DevScalars *scal, float tol, const int *blk_list, const int *blk_count) {
    if (scal->flags[0]) return;
    __shared__ float s4[4];
    const float num = cg_total(part_rr, nb, blk_list, blk_count, s4);
    if (threadIdx.x == 0) {
        const float err = __builtin_sqrtf(num);
        scal->red[3] = err;
        if (!(err > tol)) scal->flags[0] = 1;
    }
}

// :394 compute_cg_alpha + :409 update_cg_x + :415 update_cg_r_and_beta (partials)
__global__ void __launch_bounds__(256)
k_cg_update_xr2(const Consts c, int n, int nb, const int *meta, int all_fluid, float4 *x, float4 *r, const float4 *p, const float4 *Ap,
                const float *part_rr, const float *part_den, float *part_rr_next, float *part_rold, DevScalars *scal,
                const int *stop_flag, const int *blk_list, const int *blk_count, const float *glob, int count_iteration,
                const float4 *part3, int part_stride, int den_stride) {
    if (stop_flag && *stop_flag) return;
    // (a physical workgroup beyond the list of fluid-holding ones has no particles, but workgroup 0 keeps the loop's books
    //  whatever the list holds -- with NO active fluid particle at all, e.g. an emitter scene before its first release, the
    //  list is empty and the books would never be kept: 1000 empty iterations per step)
    const int blk = cg_block(blk_list, blk_count);
    if (blk < 0 && blockIdx.x != 0) return;
    // this particle's operands are requested BEFORE the sums that give alpha (none of them depends on it): one round trip for both
    const int i = blk * 256 + threadIdx.x;
    const bool mine = blk >= 0 && i < n && is_fluid(meta, i, all_fluid);
    const int ic = mine ? i : 0;
    float4 xx = x[ic];
    const float4 pp = p[ic], rr = r[ic];
    float4 a;
    if (part3) {   // A p of the split walks, added up here (what k_cg_ap_combine did in a launch of its own)
        const float4 a0 = part3[ic], a1 = part3[(size_t)part_stride + ic], a2 = part3[2 * (size_t)part_stride + ic];
        float ax = ((a0.x + a1.x) + a2.x) * c.dt, ay = ((a0.y + a1.y) + a2.y) * c.dt, az = ((a0.z + a1.z) + a2.z) * c.dt;
        ax = fdiv(ax, c.rho0); ay = fdiv(ay, c.rho0); az = fdiv(az, c.rho0);
        a = make_float4(ax + pp.x, ay + pp.y, az + pp.z, 0.f);
    } else a = Ap[ic];
    __shared__ float s8[8];
    float num_a, den_a;
#ifdef SPH_NO_EARLY_LOADS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // A/B: the operands first, then the sums (two round trips, as before round 5)
#endif
    // (split walks without a combining kernel: part_den = [3][den_stride], CgApPass::partial)
    if (glob) { num_a = glob[0]; den_a = glob[1]; }
    else cg_total2(part_rr, part_den, nb, blk_list, blk_count, s8, num_a, den_a, part3 ? den_stride : 0);
    const float alpha = den_a > 1e-18f ? num_a / den_a : 0.0f;     // :403
    if (blockIdx.x == 0 && threadIdx.x == 0) { scal->red[4] = alpha; if (count_iteration) scal->flags[1] += 1; }   // (fused p update: this kernel ends the iteration)
    float num = 0.f, den = 0.f;
    if (mine) {
        xx.x += alpha * pp.x; xx.y += alpha * pp.y; xx.z += alpha * pp.z;
        x[i] = xx;
        const float4 nr = make_float4(rr.x - alpha * a.x, rr.y - alpha * a.y, rr.z - alpha * a.z, 0.f);
        num = nr.x * nr.x + nr.y * nr.y + nr.z * nr.z;
        den = rr.x * rr.x + rr.y * rr.y + rr.z * rr.z;
        r[i] = nr;
    }
    block_sum2(num, den, part_rr_next, part_rold, blk >= 0 ? red_slot(blk_list, blk) : -1);
}

// :427-431 beta, error + :434 update_p; workgroup 0: loop book-keeping (:445 `while tol > 1e-6`)
__global__ void __launch_bounds__(256)
k_cg_update_p2(int n, int nb, const int *meta, int all_fluid, const float4 *r, float4 *p, const float *part_rr_next,
               const float *part_rold, DevScalars *scal, int looped, float tol, const int *stop_flag, const int *blk_list,
               const int *blk_count, const float *glob) {
    if (stop_flag && *stop_flag) return;
    const int blk = cg_block(blk_list, blk_count);
    if (blk < 0 && blockIdx.x != 0) return;   // (workgroup 0 keeps the books even when the list is empty, see k_cg_update_xr2)
    __shared__ float s8[8];
    float num, den;
    if (glob) { num = glob[0]; den = glob[1]; }
    else cg_total2(part_rr_next, part_rold, nb, blk_list, blk_count, s8, num, den);
    const float beta = den > 1e-18f ? num / den : 0.0f;
    int i = blk * 256 + threadIdx.x;
    if (blk >= 0 && i < n && is_fluid(meta, i, all_fluid)) {
        const float4 rr = r[i];
        float4 pp = p[i];
        pp.x = rr.x + beta * pp.x; pp.y = rr.y + beta * pp.y; pp.z = rr.z + beta * pp.z;
        p[i] = pp;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const float err = __builtin_sqrtf(num);
        scal->red[5] = beta; scal->red[3] = err;   // cg_beta, cg_error
        // raised after this workgroup's own p update; workgroups that see it early skip theirs, which only touches cg_p --
        // re-initialised by the next solve (:318)
        if (looped) { scal->flags[1] += 1; if (!(err > tol)) scal->flags[0] = 1; }
    }
}

// :440 prepare_guess
__global__ void __launch_bounds__(256)
k_cg_prepare_guess(int n, const int *meta, int all_fluid, float4 *x, const float4 *v0) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (!is_fluid(meta, i, all_fluid)) {
        // slab sharding: a ghost slot holds the neighbour rank's solved VELOCITY (it fed the viscosity formula), not a guess;
        // the slot is somebody else's next step (the guess is slot-indexed, base_container.py:506 does not reorder it), and a
        // velocity there would start that particle's solve an O(|v|) off.  Zero = "start from the current velocity".
        if (META_GHOST(meta[i])) x[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    float4 xx = x[i];
    const float4 v = v0[i];
    xx.x -= v.x; xx.y -= v.y; xx.z -= v.z;
    x[i] = xx;
}
