// sph_solvers.hpp -- DFSPH / PCISPH pair functors for k_nbr_pass (included inside the per-build namespace).
#pragma once
#include "sph_passes.hpp"

// ---------------------------------------------------------------------------------------
// base_solver.py:522 compute_density + DFSPH.py:23 compute_alpha (+task :48), fused (both run
// right after the sort and need the same neighbours).  Staged w = +V fluid / -V rigid.
// Bytes / particle: R posv 16 -> W rho 4 + alpha 4.
template <bool AF>
struct DfsphDensityAlphaPass {
    static constexpr int MODES = 0b011;               // first pass after a sort: computes and stores the acceptance masks
    static constexpr bool FLUID_BLOCKS_ONLY = true;   // active for fluid only, passive() empty
    static constexpr int BLOCK = 256, GROUPS = 3;
    static constexpr bool USES_J = !AF;   // pair() looks at j only for rigid neighbours
    static constexpr bool HAS_B = false, COUNT_PAIRS = true, HAS_REDUCE = false;
    static constexpr int PAIR_WEIGHT = 2;  // density pass + alpha pass of the reference
    typedef int BT;
    struct Own { float sum, s3, gx, gy, gz; };
    const float4 *posv; const int *meta; float *rho, *alpha; float *red_out;

    __device__ float4 stage_impl(int j) const {
        float4 p = posv[j];
        if (!AF && META_MAT(meta[j]) != 1) p.w = -p.w;
        return p;
    }
    __device__ float4 loadA(int j) const { return stage_impl(j); }
    __device__ BT loadB(int) const { return 0; }
    __device__ float4 stage(const Consts &, int j, BT &) const { return stage_impl(j); }
    __device__ bool begin(const Consts &c, int i, const float4 &, Own &o) const {
        o.sum = o.s3 = o.gx = o.gy = o.gz = 0.0f;
        if (AF && !c.ghosts) return true;
        return META_ACTIVE_FLUID(meta[i]);   // slab sharding: ghost copies are neighbours only
    }
    __device__ void pair(const Consts &c, Own &o, float dx, float dy, float dz, float r2, const float4 &a, const BT &,
                         int) const {
        const Geom g = geom(c, r2);
        const float V = fabsf(a.w);
#if SPH_FAST
        o.sum += V * kernWpoly(g);   // kW is applied to the sum (finish)
#else
        o.sum += V * kernW(c, g);
#endif
        float gx, gy, gz;
        kernGrad(c, dx, dy, dz, g, gx, gy, gz);
        const float px = -V * gx, py = -V * gy, pz = -V * gz;
        if (AF || a.w > 0.0f) o.s3 += px * px + py * py + pz * pz;
        o.gx += px; o.gy += py; o.gz += pz;
    }
    __device__ float finish(const Consts &c, int i, const float4 &pi, Own &o) const {
        float den = pi.w * c.W0;
#if SPH_FAST
        den += c.kW * o.sum;
#else
        den += o.sum;
#endif
        den *= c.rho0;
        rho[i] = den;
        float s = o.s3;
        s += o.gx * o.gx + o.gy * o.gy + o.gz * o.gz;
        alpha[i] = s > 1e-5f ? 1.0f / s : 0.0f;
        return 0.0f;
    }
    __device__ void passive(const Consts &, int, const float4 &) const {}
};

// ---------------------------------------------------------------------------------------
// The three neighbour sums that follow DFSPH's end-of-step sort in ONE walk: compute_density (base_solver.py:522), compute_alpha
// (DFSPH.py:23) and the compute_density_derivative that opens correct_divergence_error (DFSPH.py:66, :140) -- the first depends on
// the positions only, the third on positions and velocities, none on the results of the others (kappa_v = D rho / Dt * alpha is
// per particle: finish()).  One pass less per step than DfsphDensityAlphaPass + DfsphRhoAdvPass<0> (same arithmetic, same order).
// Bytes / particle: R posv 16 + velm 16 -> W rho 4 + alpha 4 + D rho / Dt 4 + kappa_v 4.
template <bool AF>
struct DfsphDensityAlphaDivPass {
    static constexpr int MODES = 0b011;               // first pass after a sort: computes and stores the acceptance masks
    static constexpr bool FLUID_BLOCKS_ONLY = true;
    static constexpr int BLOCK = 256, GROUPS = 3;
    static constexpr bool USES_J = !AF;
    static constexpr bool HAS_B = true, COUNT_PAIRS = true, HAS_REDUCE = true;
    static constexpr int PAIR_WEIGHT = 3;             // density + alpha + density-derivative passes of the reference
    struct BT { float x, y, z; };                     // v_j
    struct Own { float sum, s3, gx, gy, gz, vx, vy, vz, dsum; int cnt; };
    const float4 *posv, *velm; const int *meta; float *rho, *alpha, *out_adv, *out_kappa; float *red_out;

    __device__ float4 stage_impl(int j, BT &bj) const {
        float4 p = ldg_idx(posv, j);
        const float4 v = ldg_idx(velm, j);
        bj.x = v.x; bj.y = v.y; bj.z = v.z;
        if (!AF && META_MAT(meta[j]) != 1) p.w = -p.w;
        return p;
    }
    __device__ float4 loadA(int j) const { BT b; return stage_impl(j, b); }
    __device__ float4 stage(const Consts &, int j, BT &bj) const { return stage_impl(j, bj); }
    __device__ bool begin(const Consts &c, int i, const float4 &, Own &o) const {
        o.sum = o.s3 = o.gx = o.gy = o.gz = o.dsum = 0.0f; o.cnt = 0;
        bool ok = true;   // no early return: begin()'s loads go out with the rest of the prologue (k_nbr_pass)
        if (!AF || c.ghosts) ok = META_ACTIVE_FLUID(meta[i]);
        const float4 v = velm[i];
        o.vx = v.x; o.vy = v.y; o.vz = v.z;
        return ok;
    }
    __device__ void pair(const Consts &c, Own &o, float dx, float dy, float dz, float r2, const float4 &a, const BT &bj, int) const {
        const Geom g = geom(c, r2);
        const float V = fabsf(a.w);
#if SPH_FAST
        o.sum += V * kernWpoly(g);   // kW is applied to the sum (finish)
#else
        o.sum += V * kernW(c, g);
#endif
        float gx, gy, gz;
        kernGrad(c, dx, dy, dz, g, gx, gy, gz);
        const float px = -V * gx, py = -V * gy, pz = -V * gz;
        if (AF || a.w > 0.0f) o.s3 += px * px + py * py + pz * pz;
        o.gx += px; o.gy += py; o.gz += pz;
        // DfsphRhoAdvPass<.., 0>::pair: V_j (v_i - v_j) . grad W, fluid and rigid neighbours alike (DFSPH.py:86-101)
        o.dsum += V * ((o.vx - bj.x) * gx + (o.vy - bj.y) * gy + (o.vz - bj.z) * gz);
        o.cnt += 1;
    }
    __device__ float finish(const Consts &c, int i, const float4 &pi, Own &o) const {
        float den = pi.w * c.W0;
#if SPH_FAST
        den += c.kW * o.sum;
#else
        den += o.sum;
#endif
        den *= c.rho0;
        rho[i] = den;
        float s = o.s3;
        s += o.gx * o.gx + o.gy * o.gy + o.gz * o.gz;
        const float al = s > 1e-5f ? 1.0f / s : 0.0f;
        alpha[i] = al;
        float adv = fmaxf(o.dsum, 0.0f);
        if (o.cnt < 20) adv = 0.0f;
        out_adv[i] = adv;
        out_kappa[i] = adv * al;
        return c.rho0 * adv;
    }
    __device__ void passive(const Consts &, int, const float4 &) const {}
};

// ---------------------------------------------------------------------------------------
// DFSPH.py:66 compute_density_derivative (+:133 compute_kappa_v, +:206 error) when MODE == 0,
// DFSPH.py:105 compute_density_star (+:218 compute_kappa, +:286 error) when MODE == 1.
// Bytes / particle: R posv 16 + velm 16 (+rho, alpha 8) -> W 8.
template <bool AF, int MODE>
struct DfsphRhoAdvPass {
    static constexpr bool FLUID_BLOCKS_ONLY = true;   // active for fluid only, passive() empty
    static constexpr int BLOCK = 256, GROUPS = 3;
    static constexpr bool USES_J = !AF;   // pair() looks at j only for rigid neighbours
    static constexpr bool HAS_B = true, COUNT_PAIRS = true, HAS_REDUCE = true;
    static constexpr int PAIR_WEIGHT = 1;
    struct BT { float x, y, z; };   // v_j: 12 bytes keep the staging of a group in one batch of loads
    struct Own { float vx, vy, vz, sum; int cnt; };
    const float4 *posv, *velm; const int *meta; const float *rho, *alpha;
    float *out_adv, *out_kappa; float *red_out;

    __device__ float4 loadA(int j) const { return posv[j]; }
    __device__ float4 stage(const Consts &, int j, BT &bj) const {
        const float4 v = ldg_idx(velm, j);
        bj.x = v.x; bj.y = v.y; bj.z = v.z;
        return ldg_idx(posv, j);
    }
    __device__ bool begin(const Consts &c, int i, const float4 &, Own &o) const {
        bool ok = true;   // no early return: begin()'s loads go out with the rest of the prologue (k_nbr_pass)
        if (!AF || c.ghosts) ok = META_ACTIVE_FLUID(meta[i]);
        const float4 v = velm[i];
        o.vx = v.x; o.vy = v.y; o.vz = v.z; o.sum = 0.0f; o.cnt = 0;
        return ok;
    }
    __device__ void pair(const Consts &c, Own &o, float dx, float dy, float dz, float r2, const float4 &a,
                         const BT &bj, int) const {
#if SPH_FAST
        // grad W = scale x (x_i - x_j): the scalar goes into the coefficient (three multiplies less per pair)
        o.sum += (a.w * kernGradScale(c, geom(c, r2))) * ((o.vx - bj.x) * dx + (o.vy - bj.y) * dy + (o.vz - bj.z) * dz);
#else
        float gx, gy, gz;
        kernGrad(c, dx, dy, dz, geom(c, r2), gx, gy, gz);
        o.sum += a.w * ((o.vx - bj.x) * gx + (o.vy - bj.y) * gy + (o.vz - bj.z) * gz);
#endif
        o.cnt += 1;
    }
    __device__ float finish(const Consts &c, int i, const float4 &, Own &o) const {
        if (MODE == 0) {
            float adv = fmaxf(o.sum, 0.0f);
            if (o.cnt < 20) adv = 0.0f;
            out_adv[i] = adv;
            out_kappa[i] = adv * alpha[i];
            return c.rho0 * adv;
        } else {
            const float adv = rho[i] / c.rho0 + c.dt * o.sum;
            const float star = fmaxf(adv, 1.0f);
            out_adv[i] = star;
            out_kappa[i] = (star - 1.0f) * alpha[i] * c.inv_dt;
            return star - 1.0f;
        }
    }
    __device__ void passive(const Consts &, int, const float4 &) const {}
};

// ---------------------------------------------------------------------------------------
// DFSPH.py:162 correct_divergence_step (+task :173) when MODE == 0 (dv accumulated, then added),
// DFSPH.py:246 correct_density_error_step (+task :255) when MODE == 1 (velocity updated pair by pair).
// Bytes / particle: R posv 16 + kappa 4 + rho 4 + velm 16 -> W velm 16.
template <bool AF, int MODE>
struct DfsphCorrectPass {
    static constexpr bool HAS_WRENCH = !AF;           // pair() may call add_wrench (sph_passes.hpp)
    static constexpr bool FLUID_BLOCKS_ONLY = true;   // active for fluid only, passive() empty
    static constexpr int BLOCK = 256, GROUPS = 3;
    static constexpr bool USES_J = !AF;   // pair() looks at j only for rigid neighbours
    static constexpr bool HAS_B = true, COUNT_PAIRS = true, HAS_REDUCE = false;
    static constexpr int PAIR_WEIGHT = 1;
    typedef float2 BT;
    struct Own { float k, rho, vx, vy, vz, m0; };
    const float4 *posv; const int *meta; const float *kappa, *rho;
    float4 *velm; DevScalars *scal; const RigidPose *pose; float rho0; float *red_out;

    __device__ float4 stage_impl(int j, BT &bj) const {
        float4 p = posv[j];
        if (!AF) {
            const int m = meta[j];
            if (META_MAT(m) != 1) { p.w = -p.w; bj = make_float2(META_DYN(m) ? 1.0f + (float)META_OBJ(m) : 0.0f, 1.0f); return p; }   // dynamic rigid: 1 + its body
        }
        bj = make_float2(kappa[j], rho[j]);
        return p;
    }
    __device__ float4 loadA(int j) const { BT b; return stage_impl(j, b); }
    __device__ BT loadB(int j) const { BT b; stage_impl(j, b); return b; }
    __device__ float4 stage(const Consts &, int j, BT &bj) const { return stage_impl(j, bj); }
    __device__ bool begin(const Consts &c, int i, const float4 &pi, Own &o) const {
        bool ok = true;   // no early return: begin()'s loads go out with the rest of the prologue (k_nbr_pass)
        if (!AF || c.ghosts) ok = META_ACTIVE_FLUID(meta[i]);
        o.k = kappa[i]; o.rho = rho[i];
        if (MODE == 1) { const float4 v = velm[i]; o.vx = v.x; o.vy = v.y; o.vz = v.z; }
        else { o.vx = o.vy = o.vz = 0.0f; }
        o.m0 = pi.w * rho0;
        return ok;
    }
    __device__ void pair(const Consts &c, Own &o, float dx, float dy, float dz, float r2, const float4 &a,
                         const BT &bj, int j) const {
#if SPH_FAST
        // scalar-coefficient form: V_j grad W (k_i / rho_i + k_j / rho_j) rho0 = (one scalar) x (x_i - x_j): 6 VALU per pair instead of 15
        if (AF || a.w > 0.0f) {
            const float ks = o.k + bj.x;
            if (fabsf(ks) > c.thr_kappa) {
                const float cc = fdiv(o.k, o.rho) + fdiv(bj.x, bj.y);
                const float k = ((a.w * kernGradScale(c, geom(c, r2))) * cc) * c.rho0;
                o.vx -= k * dx; o.vy -= k * dy; o.vz -= k * dz;
            }
        } else {
            if (fabsf(o.k) > c.thr_kappa) {
                const float k = ((-a.w * kernGradScale(c, geom(c, r2))) * fdiv(o.k, o.rho)) * c.rho0;
                const float tx = k * dx, ty = k * dy, tz = k * dz;
                o.vx -= tx; o.vy -= ty; o.vz -= tz;
                if (bj.x >= 1.0f) {  // dynamic rigid: DFSPH.py:195-204 / :277-285 (body and position staged with the neighbour, centre of mass in LDS)
                    const int obj = (int)bj.x - 1;
                    const float fx = fdiv(tx, c.dt) * o.m0, fy = fdiv(ty, c.dt) * o.m0, fz = fdiv(tz, c.dt) * o.m0;
                    const float rx = a.x - wrench_com()[obj][0], ry = a.y - wrench_com()[obj][1], rz = a.z - wrench_com()[obj][2];
                    add_wrench(scal, obj, fx, fy, fz, ry * fz - rz * fy, rz * fx - rx * fz, rx * fy - ry * fx);
                }
            }
        }
#else
        float gx, gy, gz;
        if (AF || a.w > 0.0f) {
            const float ks = o.k + bj.x;
            if (fabsf(ks) > c.thr_kappa) {
                kernGrad(c, dx, dy, dz, geom(c, r2), gx, gy, gz);
                const float cc = fdiv(o.k, o.rho) + fdiv(bj.x, bj.y);
                o.vx -= ((a.w * gx) * cc) * c.rho0; o.vy -= ((a.w * gy) * cc) * c.rho0; o.vz -= ((a.w * gz) * cc) * c.rho0;
            }
        } else {
            if (fabsf(o.k) > c.thr_kappa) {
                kernGrad(c, dx, dy, dz, geom(c, r2), gx, gy, gz);
                const float V = -a.w;
                const float cc = fdiv(o.k, o.rho);
                const float tx = ((V * gx) * cc) * c.rho0, ty = ((V * gy) * cc) * c.rho0, tz = ((V * gz) * cc) * c.rho0;
                o.vx -= tx; o.vy -= ty; o.vz -= tz;
                if (bj.x >= 1.0f) {  // dynamic rigid: DFSPH.py:195-204 / :277-285 (body and position staged with the neighbour, centre of mass in LDS)
                    const int obj = (int)bj.x - 1;
                    const float fx = fdiv(tx, c.dt) * o.m0, fy = fdiv(ty, c.dt) * o.m0, fz = fdiv(tz, c.dt) * o.m0;
                    const float rx = a.x - wrench_com()[obj][0], ry = a.y - wrench_com()[obj][1], rz = a.z - wrench_com()[obj][2];
                    add_wrench(scal, obj, fx, fy, fz, ry * fz - rz * fy, rz * fx - rx * fz, rx * fy - ry * fx);
                }
            }
        }
#endif
    }
    __device__ float finish(const Consts &, int i, const float4 &, Own &o) const {
        float4 v = velm[i];
        if (MODE == 1) { v.x = o.vx; v.y = o.vy; v.z = o.vz; }
        else { v.x += o.vx; v.y += o.vy; v.z += o.vz; }
        velm[i] = v;
        return 0.0f;
    }
    __device__ void passive(const Consts &, int, const float4 &) const {}
};

// ---------------------------------------------------------------------------------------
// PCISPH.py:33 compute_density_star (+task :49) + :66 update_pressure, fused.  The neighbour
// test runs on the current positions (for_all_neighbors), the kernel on predicted positions of
// fluid neighbours / current positions of rigid neighbours.
// Bytes / particle: R posv 16 + ppos 16 + prs 4 + rho 4 -> W rho_star 4 + prs 4 + ptm 4.
template <bool AF>
struct PcisphRhoStarPass {
    static constexpr bool FLUID_BLOCKS_ONLY = true;   // active for fluid only, passive() empty
    static constexpr int BLOCK = 256, GROUPS = 3;
    static constexpr bool USES_J = !AF;   // pair() looks at j only for rigid neighbours
    static constexpr bool HAS_B = true, COUNT_PAIRS = true, HAS_REDUCE = true;
    static constexpr int PAIR_WEIGHT = 1;
    struct BT { float x, y, z; };   // predicted position of a fluid neighbour / current position of a rigid one
    struct Own { float px, py, pz, sum; };
    const float4 *posv, *ppos; const int *meta; const float *rho;
    float *rho_star, *prs, *ptm; float *red_out;
#ifdef SPH_TEST_HOOKS
    float *cap_prev;   // test-hook library: the pressure this update started from (SPH_F_DEBUG_CAPTURE); a launch past the stop of a device loop writes nothing
#endif

    __device__ float4 loadA(int j) const { return posv[j]; }
    __device__ float4 stage(const Consts &, int j, BT &bj) const {
        const float4 p = ldg_idx(posv, j);
        float4 q;
        if (!AF && META_MAT(meta[j]) != 1) q = p;
        else q = ldg_idx(ppos, j);
        bj.x = q.x; bj.y = q.y; bj.z = q.z;
        return p;
    }
    __device__ bool begin(const Consts &c, int i, const float4 &, Own &o) const {
        bool ok = true;   // no early return: begin()'s loads go out with the rest of the prologue (k_nbr_pass)
        if (!AF || c.ghosts) ok = META_ACTIVE_FLUID(meta[i]);
        const float4 q = ppos[i];
        o.px = q.x; o.py = q.y; o.pz = q.z; o.sum = 0.0f;
        return ok;
    }
    __device__ void pair(const Consts &c, Own &o, float, float, float, float, const float4 &a, const BT &bj, int) const {
        const float dx = o.px - bj.x, dy = o.py - bj.y, dz = o.pz - bj.z;
        o.sum += a.w * kernW<false>(c, geom(c, dx * dx + dy * dy + dz * dz));   // predicted distance: may exceed h
    }
    __device__ float finish(const Consts &c, int i, const float4 &, Own &o) const {
        const float star = o.sum * c.rho0;
        rho_star[i] = star;
#ifdef SPH_TEST_HOOKS
        if (cap_prev) cap_prev[i] = prs[i];
#endif
        float p = prs[i] + c.pcisph_k * (c.rho0 - star);
        if (p < 0.0f) p = 0.0f;
        prs[i] = p;
        const float r = rho[i];
        ptm[i] = p / (r * r);
        return fmaxf(0.0f, o.sum - 1.0f);
    }
    __device__ void passive(const Consts &, int, const float4 &) const {}
};

// PCISPH.py:75 compute_temp_pressure_acceleration (+task :85) + :19 compute_predicted_velocity
// + :26 compute_predicted_position, fused.
// Bytes / particle: R posv 16 + velm 16 + ptm 4 + acc 16 -> W pacc 16 + ppos 16 (+pvel 16).
template <bool AF>
struct PcisphPressureAccelPass {
    static constexpr int BLOCK = 256, GROUPS = 3;
    static constexpr bool USES_J = !AF;   // pair() looks at j only for rigid neighbours
    static constexpr bool HAS_B = true, COUNT_PAIRS = true, HAS_REDUCE = false;
    static constexpr int PAIR_WEIGHT = 1;
    typedef float BT;
    struct Own { float pt, ax, ay, az; };
    const float4 *posv, *velm; const int *meta; const float *ptm;
    const float4 *acc_np, *vel0;   // non-pressure acceleration, velocity at the start of the step
    float4 *pacc, *pvel, *ppos; float rho0; float *red_out;

    __device__ float4 stage_impl(int j, BT &bj) const {
        const float4 p = posv[j];
        const float m = velm[j].w;
        if (AF) { bj = ptm[j]; return make_float4(p.x, p.y, p.z, m); }
        const bool fl = META_MAT(meta[j]) == 1;
        bj = fl ? ptm[j] : -1.0f;
        return make_float4(p.x, p.y, p.z, fl ? m : rho0 * p.w);
    }
    __device__ float4 loadA(int j) const { BT b; return stage_impl(j, b); }
    __device__ BT loadB(int j) const { BT b; stage_impl(j, b); return b; }
    __device__ float4 stage(const Consts &, int j, BT &bj) const { return stage_impl(j, bj); }
    __device__ bool begin(const Consts &c, int i, const float4 &, Own &o) const {
        bool ok = true;   // no early return: begin()'s loads go out with the rest of the prologue (k_nbr_pass)
        if (!AF || c.ghosts) ok = META_ACTIVE_FLUID(meta[i]);
        o.pt = ptm[i];
        o.ax = o.ay = o.az = 0.0f;
        return ok;
    }
    __device__ void pair(const Consts &c, Own &o, float dx, float dy, float dz, float r2, const float4 &a,
                         const BT &bj, int) const {
#if SPH_FAST
        const float k = ((AF || bj >= 0.0f) ? -a.w * (o.pt + bj) : -a.w * o.pt) * kernGradScale(c, geom(c, r2));   // scalar x (x_i - x_j)
        o.ax += k * dx; o.ay += k * dy; o.az += k * dz;
#else
        float gx, gy, gz;
        kernGrad(c, dx, dy, dz, geom(c, r2), gx, gy, gz);
        const float cc = (AF || bj >= 0.0f) ? -a.w * (o.pt + bj) : -a.w * o.pt;
        o.ax += cc * gx; o.ay += cc * gy; o.az += cc * gz;
#endif
    }
    __device__ float finish(const Consts &c, int i, const float4 &pi, Own &o) const {
        pacc[i] = make_float4(o.ax, o.ay, o.az, 0.0f);
        const float4 a = acc_np[i], v = vel0[i];
        const float vx = v.x + c.dt * (a.x + o.ax), vy = v.y + c.dt * (a.y + o.ay), vz = v.z + c.dt * (a.z + o.az);
        pvel[i] = make_float4(vx, vy, vz, 0.0f);
        ppos[i] = make_float4(pi.x + c.dt * vx, pi.y + c.dt * vy, pi.z + c.dt * vz, 0.0f);
        return 0.0f;
    }
    __device__ void passive(const Consts &, int i, const float4 &) const { pacc[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
};
