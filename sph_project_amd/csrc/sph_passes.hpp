// sph_passes.hpp -- pair-physics functors plugged into k_nbr_pass (see sph_device.hpp).
// Each functor cites the reference kernel(s) it fuses.  AF = "all fluid" specialisation
// (no rigid / emitter particles in the container): material tests compile away.
#pragma once
#include "sph_device.hpp"

// Force / torque of one fluid -- dynamic-rigid pair onto the body's accumulators (base_solver.py:174-187, :272-278; DFSPH.py:173-203).
// Called from inside the pair loop by whichever lanes of the wave met such a neighbour in this trip.  Until round 3: six f32 atomicAdd per
// pair and lane onto the SAME six words of global memory.  Same-address atomics are served one at a time (~0.1 us each): a plate under a
// 50 k-particle block, 10^5 fluid-rigid pairs per pass, took a 0.09 ms step to 0.8 ms and more -- and the f32 sums depended on the order
// the workgroups finished in.  Now every WAVE owns a row of six floats per body in LDS (4 waves x 20 bodies x 24 B = 1.9 KB):
//   * a pair is six ds_add_f32 onto the wave's own row -- nothing else in the pair loop (summing over the lanes in registers first, or
//     flushing from inside the loop, cost the rigid-aware passes 10-50 spilled VGPRs);
//   * behind the pair loops (wrench_flush_all) the four rows of a body are added up in a fixed order and go out as six 64-bit
//     FIXED-POINT atomics per body and workgroup (DevScalars::wrench).  Integer addition commutes, and the lanes of ONE ds_add_f32 are
//     served in an order that depends on the instruction's lanes and addresses, not on timing: the wrench is bit-reproducible from run
//     to run (tests/test_hip_rigid.py::test_wrench_is_bit_reproducible_and_cheap).
typedef float WrenchRows[4][SPH_NOBJ][6];
__device__ __forceinline__ WrenchRows &wrench_rows() {
    __shared__ WrenchRows s_rows;   // (one instance per kernel: every caller gets the same array)
    return s_rows;
}
// the bodies' centres of mass (rigid_body_centers_of_mass, the reference point of the torques), copied into LDS once per workgroup: the
// pair loop read them from global memory behind the neighbour's meta word -- two dependent round trips per fluid-rigid pair
typedef float WrenchCom[SPH_NOBJ][3];
__device__ __forceinline__ WrenchCom &wrench_com() {
    __shared__ WrenchCom s_com;
    return s_com;
}
// start of a workgroup (every thread, before the first barrier) / behind the pair loops (every thread, behind a barrier)
__device__ __forceinline__ void wrench_init_all(const RigidPose *pose) {
    float *r = &wrench_rows()[0][0][0];
    for (int k = threadIdx.x; k < 4 * SPH_NOBJ * 6; k += 256) r[k] = 0.0f;
    if (threadIdx.x < SPH_NOBJ * 3) (&wrench_com()[0][0])[threadIdx.x] = (&pose->com[0][0])[threadIdx.x];
}
__device__ __forceinline__ void wrench_flush_all(DevScalars *scal) {
    const int k = threadIdx.x;   // (body, component)
    if (k >= SPH_NOBJ * 6) return;
    const WrenchRows &w = wrench_rows();
    const int obj = k / 6, q = k % 6;
    const float v = ((w[0][obj][q] + w[1][obj][q]) + w[2][obj][q]) + w[3][obj][q];
    if (v != 0.0f)
        atomicAdd((unsigned long long *)(scal->wrench + (q < 3 ? 0 : SPH_NOBJ * 3) + obj * 3 + q % 3),
                  (unsigned long long)__double2ll_rn((double)v * SPH_WRENCH_SCALE));
}
__device__ __forceinline__ void add_wrench(DevScalars *, int obj, float fx, float fy, float fz, float tx, float ty, float tz) {
    float *r = wrench_rows()[threadIdx.x >> 6][obj];
    atomicAdd(r + 0, fx); atomicAdd(r + 1, fy); atomicAdd(r + 2, fz);
    atomicAdd(r + 3, tx); atomicAdd(r + 4, ty); atomicAdd(r + 5, tz);
}

// ---------------------------------------------------------------------------------------
// base_solver.py:522 compute_density (+task :535); with EOS also WCSPH.py:17 compute_pressure.
// Algorithmic HBM bytes / particle: R posv 16 -> W rho 4 (+ rho_raw 4, prs 4, ptm 4 with EOS).
template <bool AF, bool EOS>
struct DensityPass {
    static constexpr int MODES = 0b011;               // first pass after a sort: computes and stores the acceptance masks
    static constexpr bool FLUID_BLOCKS_ONLY = true;   // active for fluid only, passive() empty
    static constexpr int BLOCK = 256, GROUPS = 3;
    static constexpr bool USES_J = !AF;   // pair() looks at j only for rigid neighbours
    static constexpr bool HAS_B = false, COUNT_PAIRS = true;
    static constexpr int PAIR_WEIGHT = 1;
    static constexpr bool HAS_REDUCE = false;
    typedef int BT;
    struct Own { float sum; };
    const float4 *posv; const int *meta;
    float *rho_raw, *rho, *prs, *ptm;
    HaloFieldSend fs;   // slab sharding (EOS form only): boundary values go straight into the neighbours' field message; fs.on = 0 otherwise
    static constexpr bool STAT_W = true;
    int stat_pairs, stat_evals;   // weights of this walk's accepted pairs in the pair statistics (1, 1; 4, 2 when it books the fused force pass too)

    __device__ float4 loadA(int j) const { return posv[j]; }
    __device__ BT loadB(int) const { return 0; }
    __device__ float4 stage(const Consts &, int j, BT &) const { return posv[j]; }
    __device__ bool begin(const Consts &c, int i, const float4 &, Own &o) const {
        o.sum = 0.0f;
        if (AF && !c.ghosts) return true;
        return META_ACTIVE_FLUID(meta[i]);   // all-fluid slab: every particle is fluid, ghosts are not targets
    }
    __device__ void pair(const Consts &c, Own &o, float, float, float, float r2, const float4 &a, const BT &,
                         int) const {
#if SPH_FAST
        // W / kW from q^2 = r^2 / h^2 and one v_sqrt (the gradient passes need 1 / r and go through v_rsq; here three VALU less per pair);
        // kW is applied to the sum (finish)
        const float q2 = r2 * c.inv_h2, q = __builtin_amdgcn_sqrtf(q2), t = 1.0f - q;
        const float lo = 1.0f - (6.0f * q2) * t, hi = 2.0f * (t * t * t);
        o.sum += a.w * (q <= 0.5f ? lo : hi);
#else
        o.sum += a.w * kernW(c, geom(c, r2));
#endif
    }
    __device__ float finish(const Consts &c, int i, const float4 &pi, Own &o) const {
        float den = pi.w * c.W0;
#if SPH_FAST
        den += c.kW * o.sum;
#else
        den += o.sum;
#endif
        den *= c.rho0;
        if (EOS) {
            rho_raw[i] = den;
            float rc = fmaxf(den, c.rho0);
            rho[i] = rc;
#if SPH_FAST
            const float x = rc * c.inv_rho0, x2 = x * x, x4 = x2 * x2;
            float pr = 50000.0f * (x4 * x2 * x - 1.0f);          // (rho / rho0)^7 by squaring: 4 multiplies instead of powf
#else
            float pr = 50000.0f * (powf(rc / c.rho0, 7.0f) - 1.0f);
#endif
            prs[i] = pr;
            const float pt = pr / (rc * rc);
            ptm[i] = pt;
            if (fs.on) {   // (uniform) layout of a field message: [ my n_send records | the records I received from that side ], k_halo_pack2
                // (bounded by this step's CLAMPED counts: after SLAB_ST_SEND_OVERFLOW k_halo_classify still tags particles with k >= rec_cap,
                //  and the neighbour's field region -- IPC-mapped remote memory -- holds 2 rec_cap entries)
                const int x = fs.xidx[i], kind = HALO_KIND(x), k = HALO_IDX(x);
                if (kind == HALO_SEND || kind == HALO_SEND + 1) {
                    const int side = kind - HALO_SEND;
                    if (fs.out[side] && k < fs.dyn->n_send[side]) fs.out[side][k] = make_float4(den, rc, pr, pt);
                } else if (kind == HALO_ECHO_SEND || kind == HALO_ECHO_SEND + 1) {
                    const int side = kind - HALO_ECHO_SEND;
                    if (fs.out[side] && k < fs.dyn->n_recv[side]) fs.out[side][fs.dyn->n_send[side] + k] = make_float4(den, rc, pr, pt);
                }
            }
        } else {
            rho[i] = den;
        }
        return 0.0f;
    }
    __device__ void passive(const Consts &, int, const float4 &) const {}
};

// ---------------------------------------------------------------------------------------
// base_solver.py:203 gravity + :210 surface tension (+task :218) + :232 explicit viscosity
// (+task :240) + :643 update_fluid_velocity, fused: v* = v + dt (g + a_st + a_visc / rho0).
// Bytes / particle: R posv 16 + velm 16 + rho_raw 4 -> W velm 16.
template <bool AF>
struct NonPressurePass {
    static constexpr bool HAS_WRENCH = !AF;   // pair() may call add_wrench: k_nbr_pass opens / flushes the per-wave rows
    static constexpr int BLOCK = 256, GROUPS = 3;
    static constexpr bool USES_J = !AF;   // pair() looks at j only for rigid neighbours
    static constexpr bool HAS_B = true, COUNT_PAIRS = true;
    static constexpr int PAIR_WEIGHT = 2;  // surface tension (:210) + viscosity (:232) = two reference passes
    static constexpr bool HAS_REDUCE = false;
    typedef float4 BT;
    struct Own { float vx, vy, vz, m, rho, st_m, sx, sy, sz, ax, ay, az; };
    const float4 *posv, *velm; const int *meta; const float *rho_raw;
    float4 *vel_out; DevScalars *scal; const RigidPose *pose; float rho0;
    int skip_viscosity;
    float4 *acc_out;     // PCISPH keeps the non-pressure acceleration (PCISPH.py:22); null otherwise
    const float4 *visc_vel;  // implicit viscosity: velocities the viscous term is evaluated with (cg_x), else null  // implicit viscosity handles the viscous term elsewhere

    __device__ float4 loadA(int j) const { BT b; return stage_impl(j, b); }
    __device__ BT loadB(int j) const { BT b; stage_impl(j, b); return b; }
    __device__ float4 stage_impl(int j, BT &bj) const {
        const float4 p = ldg_idx(posv, j);
        float4 v = ldg_idx(velm, j);
        if (visc_vel && (AF || META_MAT(meta[j]) == 1)) { const float4 u = visc_vel[j]; v.x = u.x; v.y = u.y; v.z = u.z; }
        float aw, bw;
        if (AF) { aw = v.w; bw = ldg_idx(rho_raw, j); }
        else {
            const int m = meta[j];
            const bool fl = META_MAT(m) == 1;
            aw = fl ? v.w : rho0 * p.w;
            bw = fl ? rho_raw[j] : (META_DYN(m) ? -2.0f - (float)META_OBJ(m) : -1.0f);   // dynamic rigid: the body rides along (no meta load in the pair loop)
        }
        bj = make_float4(v.x, v.y, v.z, bw);
        return make_float4(p.x, p.y, p.z, aw);
    }
    __device__ float4 stage(const Consts &, int j, BT &bj) const { return stage_impl(j, bj); }
    __device__ bool begin(const Consts &c, int i, const float4 &, Own &o) const {
        bool ok = true;   // no early return: begin()'s loads go out with the rest of the prologue (k_nbr_pass)
        if (!AF || c.ghosts) ok = META_ACTIVE_FLUID(meta[i]);
        float4 v = velm[i];
        if (visc_vel) { const float4 u = visc_vel[i]; v.x = u.x; v.y = u.y; v.z = u.z; }  // base_solver.py:464
        o.vx = v.x; o.vy = v.y; o.vz = v.z; o.m = v.w;
        o.rho = rho_raw[i];
#if SPH_FAST
        o.st_m = (fdiv(c.st, v.w) * c.rho0) * c.kW;   // (the fast pair() keeps rho0 x (surface tension + viscosity) in o.ax and takes W without its kW)
#else
        o.st_m = fdiv(c.st, v.w);
#endif
        o.sx = o.sy = o.sz = 0.0f;
        o.ax = o.ay = o.az = 0.0f;
        return ok;
    }
    __device__ void pair(const Consts &c, Own &o, float dx, float dy, float dz, float r2, const float4 &a,
                         const BT &bj, int j) const {
        const Geom g = geom(c, r2);
#if SPH_FAST
        // scalar-coefficient form (see WcsphForcePass::pair): o.ax accumulates rho0 x (surface tension + viscosity), one fma per component
        if (AF || bj.w >= 0.0f) {
            const float w = r2 > c.diameter2 ? kernWpoly(g) : c.Wd_poly;
            float k = -((o.st_m * a.w) * w);                       // (st_m carries rho0 and kW)
            if (!skip_viscosity) {
                const float v_xy = (o.vx - bj.x) * dx + (o.vy - bj.y) * dy + (o.vz - bj.z) * dz;
                const float m_ij = (o.m + a.w) * 0.5f;
                k += (fdiv2(c.cv * m_ij, bj.w, r2 + c.visc_eps) * v_xy) * kernGradScale(c, g);
            }
            o.ax += k * dx; o.ay += k * dy; o.az += k * dz;
        } else {
            if (skip_viscosity) return;
            const float v_xy = (o.vx - bj.x) * dx + (o.vy - bj.y) * dy + (o.vz - bj.z) * dz;
            const float k = (fdiv2(c.cvb * a.w, o.rho, r2 + c.visc_eps) * v_xy) * kernGradScale(c, g);
            const float acx = k * dx, acy = k * dy, acz = k * dz;
            o.ax += acx; o.ay += acy; o.az += acz;
            if (bj.w <= -2.0f) {  // dynamic rigid neighbour: base_solver.py:272-278
                const int obj = (int)(-bj.w) - 2;
                const float fx = fdiv(-acx * o.m, c.rho0), fy = fdiv(-acy * o.m, c.rho0), fz = fdiv(-acz * o.m, c.rho0);
                const float rx = a.x - wrench_com()[obj][0], ry = a.y - wrench_com()[obj][1], rz = a.z - wrench_com()[obj][2];
                add_wrench(scal, obj, fx, fy, fz, ry * fz - rz * fy, rz * fx - rx * fz, rx * fy - ry * fx);
            }
        }
#else
        const float rn2 = g.rn * g.rn;      // base_solver.py:254 R.norm()**2
        float gx, gy, gz;
        if (AF || bj.w >= 0.0f) {
            // surface tension
            const float cst = o.st_m * a.w;
            const float w = r2 > c.diameter2 ? kernW(c, g) : c.Wd;
            o.sx -= (cst * dx) * w; o.sy -= (cst * dy) * w; o.sz -= (cst * dz) * w;
            if (skip_viscosity) return;
            kernGrad(c, dx, dy, dz, g, gx, gy, gz);
            const float v_xy = (o.vx - bj.x) * dx + (o.vy - bj.y) * dy + (o.vz - bj.z) * dz;
            const float m_ij = (o.m + a.w) * 0.5f;
            const float cc = fdiv2(c.cv * m_ij, bj.w, rn2 + c.visc_eps) * v_xy;
            o.ax += cc * gx; o.ay += cc * gy; o.az += cc * gz;
        } else {
            if (skip_viscosity) return;
            kernGrad(c, dx, dy, dz, g, gx, gy, gz);
            const float v_xy = (o.vx - bj.x) * dx + (o.vy - bj.y) * dy + (o.vz - bj.z) * dz;
            const float cc = fdiv2(c.cvb * a.w, o.rho, rn2 + c.visc_eps) * v_xy;
            const float acx = cc * gx, acy = cc * gy, acz = cc * gz;
            o.ax += acx; o.ay += acy; o.az += acz;
            if (bj.w <= -2.0f) {  // dynamic rigid neighbour: base_solver.py:272-278
                const int obj = (int)(-bj.w) - 2;
                const float fx = fdiv(-acx * o.m, c.rho0), fy = fdiv(-acy * o.m, c.rho0), fz = fdiv(-acz * o.m, c.rho0);
                const float rx = a.x - wrench_com()[obj][0], ry = a.y - wrench_com()[obj][1], rz = a.z - wrench_com()[obj][2];
                add_wrench(scal, obj, fx, fy, fz, ry * fz - rz * fy, rz * fx - rx * fz, rx * fy - ry * fx);
            }
        }
#endif
    }
    __device__ float finish(const Consts &c, int i, const float4 &, Own &o) const {
        float ax = c.gx, ay = c.gy, az = c.gz;
        ax += o.sx; ay += o.sy; az += o.sz;
        ax += fdiv(o.ax, c.rho0); ay += fdiv(o.ay, c.rho0); az += fdiv(o.az, c.rho0);
        if (acc_out) acc_out[i] = make_float4(ax, ay, az, 0.0f);
        float vx = o.vx, vy = o.vy, vz = o.vz;
        if (visc_vel) { const float4 v = velm[i]; vx = v.x; vy = v.y; vz = v.z; }  // :470 copy_back_original_velocity
        vel_out[i] = make_float4(vx + c.dt * ax, vy + c.dt * ay, vz + c.dt * az, o.m);
        return 0.0f;
    }
    __device__ void passive(const Consts &, int i, const float4 &) const { vel_out[i] = velm[i]; }
};

// base_solver.py:575 enforce_domain_boundary_3D + :545 simulate_collisions
__device__ __forceinline__ void enforce_boundary(const Consts &c, float &x, float &y, float &z, float &vx,
                                                 float &vy, float &vz) {
    float nx = 0.f, ny = 0.f, nz = 0.f;
    const float ox = x, oy = y, oz = z;
    if (ox > c.hix) { nx += 1.0f; x = c.hix; }
    if (ox <= c.pad) { nx += -1.0f; x = c.pad; }
    if (oy > c.hiy) { ny += 1.0f; y = c.hiy; }
    if (oy <= c.pad) { ny += -1.0f; y = c.pad; }
    if (oz > c.hiz) { nz += 1.0f; z = c.hiz; }
    if (oz <= c.pad) { nz += -1.0f; z = c.pad; }
    const float len = __builtin_sqrtf(nx * nx + ny * ny + nz * nz);
    if (len > 1e-6f) {
        const float ux = nx / len, uy = ny / len, uz = nz / len;
        const float s = (1.0f + 0.5f) * (vx * ux + vy * uy + vz * uz);
        vx -= s * ux; vy -= s * uy; vz -= s * uz;
    }
}

// ---------------------------------------------------------------------------------------
// base_solver.py:136 compute_pressure_acceleration (+task :147) + :643 update_fluid_velocity
// + :652 update_fluid_position + :575 enforce_domain_boundary_3D, fused.
// Bytes / particle: R posv 16 + velm 16 + ptm 4 -> W acc 16 + posv 16 + velm 16.
template <bool AF>
struct PressurePass {
    static constexpr bool HAS_WRENCH = !AF;
    static constexpr int BLOCK = 256, GROUPS = 3;
    static constexpr bool USES_J = !AF;   // pair() looks at j only for rigid neighbours
    static constexpr bool HAS_B = true, COUNT_PAIRS = true;
    static constexpr int PAIR_WEIGHT = 1;
    static constexpr bool HAS_REDUCE = false;
    typedef float BT;
    struct Own { float pt, p, rho2, ax, ay, az, x, y, z, m0; };
    const float4 *posv; const int *meta; const float *ptm, *prs, *rho;
    float4 *velm;      // in: v* (+m), out: v
    float4 *acc, *posv_out; DevScalars *scal; const RigidPose *pose; float rho0;
    int integrate;     // 1: WCSPH/PCISPH tail (v, x update + boundary); 0: acceleration only

    __device__ float4 stage_impl(int j, BT &bj) const {
        const float4 p = posv[j];
        const float m = velm[j].w;
        if (AF) { bj = ptm[j]; return make_float4(p.x, p.y, p.z, m); }
        const int mt = meta[j];
        const bool fl = META_MAT(mt) == 1;
        bj = fl ? ptm[j] : (META_DYN(mt) ? -2.0f - (float)META_OBJ(mt) : -1.0f);
        return make_float4(p.x, p.y, p.z, fl ? m : rho0 * p.w);
    }
    __device__ float4 loadA(int j) const { BT b; return stage_impl(j, b); }
    __device__ BT loadB(int j) const { BT b; stage_impl(j, b); return b; }
    __device__ float4 stage(const Consts &, int j, BT &bj) const { return stage_impl(j, bj); }
    __device__ bool begin(const Consts &c, int i, const float4 &pi, Own &o) const {
        bool ok = true;   // no early return: begin()'s loads go out with the rest of the prologue (k_nbr_pass)
        if (!AF) { const int m = meta[i]; ok = META_ACTIVE_FLUID(m) && META_DYN(m); }
        else if (c.ghosts) ok = !META_GHOST(meta[i]);
        o.x = pi.x; o.y = pi.y; o.z = pi.z; o.m0 = rho0 * pi.w;
        o.pt = ptm[i]; o.p = prs[i];
        const float r = rho[i];
        o.rho2 = r * r;
        o.ax = o.ay = o.az = 0.0f;
        return ok;
    }
    __device__ void pair(const Consts &c, Own &o, float dx, float dy, float dz, float r2, const float4 &a,
                         const BT &bj, int j) const {
        const Geom g = geom(c, r2);
#if SPH_FAST
        if (AF || bj >= 0.0f) {   // scalar x (x_i - x_j), see WcsphForcePass::pair
            const float k = (-a.w * (o.pt + bj)) * kernGradScale(c, g);
            o.ax += k * dx; o.ay += k * dy; o.az += k * dz;
            return;
        }
#endif
        float gx, gy, gz;
        kernGrad(c, dx, dy, dz, g, gx, gy, gz);
        if (AF || bj >= 0.0f) {
            const float cc = -a.w * (o.pt + bj);
            o.ax += cc * gx; o.ay += cc * gy; o.az += cc * gz;
        } else {
            const float cc = fdiv(-a.w * o.p, o.rho2);
            o.ax += cc * gx; o.ay += cc * gy; o.az += cc * gz;
            if (bj <= -2.0f) {  // base_solver.py:174-187 (torque about pos_i, sic)
                const int obj = (int)(-bj) - 2;
                const float cf = fdiv(a.w * o.p, o.rho2);
                const float fx = (cf * gx) * o.m0, fy = (cf * gy) * o.m0, fz = (cf * gz) * o.m0;
                const float rx = o.x - wrench_com()[obj][0], ry = o.y - wrench_com()[obj][1], rz = o.z - wrench_com()[obj][2];
                add_wrench(scal, obj, fx, fy, fz, ry * fz - rz * fy, rz * fx - rx * fz, rx * fy - ry * fx);
            }
        }
    }
    __device__ float finish(const Consts &c, int i, const float4 &pi, Own &o) const {
        acc[i] = make_float4(o.ax, o.ay, o.az, 0.0f);
        if (!integrate) return 0.0f;
        const float4 v = velm[i];
        float vx = v.x + c.dt * o.ax, vy = v.y + c.dt * o.ay, vz = v.z + c.dt * o.az;
        float x = pi.x + c.dt * vx, y = pi.y + c.dt * vy, z = pi.z + c.dt * vz;
        enforce_boundary(c, x, y, z, vx, vy, vz);
        posv_out[i] = make_float4(x, y, z, pi.w);
        velm[i] = make_float4(vx, vy, vz, v.w);
        return 0.0f;
    }
    __device__ void passive(const Consts &, int i, const float4 &pi) const {
        acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (integrate) posv_out[i] = pi;
    }
};

// ---------------------------------------------------------------------------------------
// WCSPH.py:30-36, 45: NonPressurePass + PressurePass in ONE neighbour walk.  The pressure acceleration
// (base_solver.py:136) reads positions, densities and pressures only -- not the velocities the non-pressure update
// (:643) has just written -- so both sums can be accumulated side by side; finish() then replays the reference's
// update sequence (v* = v + dt a_np; v = v* + dt a_p; x += dt v; boundary) with the same roundings.
// Per candidate: A = (x, y, z, m_j | rho0 V_j), B = (v_j, rho_raw_j | -1 static / -2 dynamic rigid), C = p_j / rho_j^2.
// Bytes / particle: R posv 16 + velm 16 + rho_raw 4 + ptm 4 (+ own prs, rho 8) -> W acc 16 + posv 16 + velm 16.
// CNT: the pass counts its own accepted pairs.  Inside wcsph_step they are booked by the density pass whose masks it walks
// (DensityPass::stat_pairs, State::density_books_forces); a caller that launches this pass any other way gets the counting
// instantiation (l_wcsph_forces), so that pair_interactions never under-reports.
template <bool AF, bool UM = false, bool CNT = false>   // UM (fast build, all fluid): every particle carries the same mass -- m_ij = m, the products with it are hoisted
struct WcsphForcePass {
    static constexpr bool HAS_WRENCH = !AF;
    static constexpr int BLOCK = 256, GROUPS = 3;
    static constexpr bool USES_J = !AF;   // pair() looks at j only for rigid neighbours
    static constexpr bool HAS_B = true, HAS_C = true;
    static constexpr bool COUNT_PAIRS = CNT;     // false: booked by the density pass that stored the masks this pass walks (DensityPass::stat_pairs)
    static constexpr int MAX_WAVES = CNT ? (SPH_FAST ? 4 : 3) : 8;   // (the rarely used counting instantiation takes its class's register budget, never less)
#ifndef SPH_FORCE_MASK_PIPE
#define SPH_FORCE_MASK_PIPE 1   // round 6: 1 (C2 -1.2 %, in motion -1.0 %: profiles/r06_maskpipe_ab.txt; it spilled when it was last tried, it does not any more)
#endif
    static constexpr bool MASK_PIPELINE = SPH_FORCE_MASK_PIPE && AF;   // the pair loop holds 111-124 of 128 VGPRs: batches of 3 OR the pipeline
    static constexpr int PAIR_WEIGHT = 3;  // surface tension (:210) + viscosity (:232) + pressure (:136)
    static constexpr bool HAS_REDUCE = false;
    typedef float4 BT;
    typedef float CT;
    struct Own {
        float vx, vy, vz, m, rho, st_m, sx, sy, sz, ax, ay, az;   // non-pressure part (NonPressurePass::Own)
        float pt, p, rho2, px, py, pz, x, y, z, m0;                // pressure part (PressurePass::Own)
        float cvm, cstm;                                           // UM: c.cv m and (st / m rho0) m, the same roundings as per pair
        int dyn;
        int lin_new;                                               // cell of the position finish() / passive() stored (NEXT_HASH)
    };
    static constexpr bool NEXT_HASH = AF;   // the all-fluid instantiations can hash for the next step (k_nbr_pass epilogue); `nh.on` decides per launch
    const float4 *posv, *velm; const int *meta; const float *rho_raw, *ptm, *prs, *rho;
    float4 *vel_out, *acc, *posv_out; DevScalars *scal; const RigidPose *pose; float rho0;
    HaloSend hs;   // slab sharding: classify + send the next step message from here (sph_halo_defs.hpp); hs.on = 0 otherwise
    NextHash nh;   // unsharded all-fluid steps with another step queued behind them: this pass is the next step's k_hash_count (nh.on)
    __device__ int cell_of(const Consts &c, float x, float y, float z) const {   // k_hash_count's cell (IEEE division in both builds)
        return (cell_coord_x(c, x) * c.ny + cell_coord(y, c.grid_size, c.ny)) * c.nz + cell_coord_z(c, z);
    }

    __device__ float4 loadA(int j) const { return posv[j]; }
    __device__ float4 stage(const Consts &, int j, BT &bj, CT &cj) const {
        const float4 p = ldg_idx(posv, j);
        const float4 v = ldg_idx(velm, j);
        if (AF) {
            bj = make_float4(v.x, v.y, v.z, ldg_idx(rho_raw, j));
            cj = ldg_idx(ptm, j);
            return make_float4(p.x, p.y, p.z, v.w);
        }
        const int m = meta[j];
        const bool fl = META_MAT(m) == 1;
        bj = make_float4(v.x, v.y, v.z, fl ? rho_raw[j] : (META_DYN(m) ? -2.0f - (float)META_OBJ(m) : -1.0f));
        cj = fl ? ptm[j] : 0.0f;
        return make_float4(p.x, p.y, p.z, fl ? v.w : rho0 * p.w);
    }
    __device__ bool begin(const Consts &c, int i, const float4 &pi, Own &o) const {
        // no early return: the loads below are unconditional, so that they go out together with the rest of the prologue
        // (k_nbr_pass calls begin() on a clamped index; the result is only used where it returns true)
        o.dyn = 1;
        bool ok = true;
        if (AF && c.ghosts) ok = !META_GHOST(meta[i]);   // all-fluid slab: ghosts are neighbours only (uniform branch)
        if (!AF) {
            const int m = meta[i];
            ok = META_ACTIVE_FLUID(m);
            if (!ok) return false;   // (this instantiation sits at the 128-VGPR limit: it keeps its old shape)
            o.dyn = META_DYN(m);
        }
        const float4 v = velm[i];
        o.vx = v.x; o.vy = v.y; o.vz = v.z; o.m = v.w;
        o.rho = rho_raw[i];
#if SPH_FAST
        o.st_m = (fdiv(c.st, v.w) * c.rho0) * c.kW;   // (the fast pair() keeps rho0 x (surface tension + viscosity) in o.ax and takes W without its kW)
#else
        o.st_m = fdiv(c.st, v.w);
#endif
        o.cvm = c.cv * ((v.w + v.w) * 0.5f); o.cstm = o.st_m * v.w;
        o.sx = o.sy = o.sz = 0.0f;
        o.ax = o.ay = o.az = 0.0f;
        o.x = pi.x; o.y = pi.y; o.z = pi.z; o.m0 = rho0 * pi.w;
        o.pt = ptm[i]; o.p = prs[i];
        const float r = rho[i];
        o.rho2 = r * r;
        o.px = o.py = o.pz = 0.0f;
        return ok;
    }
    __device__ void pair(const Consts &c, Own &o, float dx, float dy, float dz, float r2, const float4 &a,
                         const BT &bj, const CT &cj, int j) const {
        const Geom g = geom(c, r2);
#if SPH_FAST
        // Fast build: every pair term is (scalar) x (x_i - x_j), so the scalars are combined first and each accumulator takes ONE fma per
        // component: surface tension and viscosity share an accumulator (o.ax = rho0 x their sum: st_m carries the rho0, finish() divides once),
        // the gradient never exists as a vector.  5 VALU and 6 registers less per pair than the term-by-term form the strict build keeps;
        // the sums differ from it by reassociation only (~1 ulp per pair).
        const float gs = kernGradScale(c, g);
        const float v_xy = (o.vx - bj.x) * dx + (o.vy - bj.y) * dy + (o.vz - bj.z) * dz;
        if (AF || bj.w >= 0.0f) {
            const float w = r2 > c.diameter2 ? kernWpoly(g) : c.Wd_poly;            // (kW sits in st_m)
            const float cw = (UM ? o.cstm : o.st_m * a.w) * w;                      // surface tension (:210), times rho0
            const float cvm = UM ? o.cvm : c.cv * ((o.m + a.w) * 0.5f);             // viscosity (:232); UM: (m + m) / 2 = m exactly
            const float cc = fdiv2(cvm, bj.w, r2 + c.visc_eps) * v_xy;
            const float k = cc * gs - cw;
            o.ax += k * dx; o.ay += k * dy; o.az += k * dz;
            const float cp = (-(UM ? o.m : a.w) * (o.pt + cj)) * gs;                // pressure (:136)
            o.px += cp * dx; o.py += cp * dy; o.pz += cp * dz;
        } else {
            const float k = (fdiv2(c.cvb * a.w, o.rho, r2 + c.visc_eps) * v_xy) * gs;
            const float acx = k * dx, acy = k * dy, acz = k * dz;
            o.ax += acx; o.ay += acy; o.az += acz;
            const float cp = fdiv(-a.w * o.p, o.rho2) * gs;
            o.px += cp * dx; o.py += cp * dy; o.pz += cp * dz;
            if (bj.w <= -2.0f) {  // dynamic rigid neighbour (see the strict form below)
                const int obj = (int)(-bj.w) - 2;
                float fx = fdiv(-acx * o.m, c.rho0), fy = fdiv(-acy * o.m, c.rho0), fz = fdiv(-acz * o.m, c.rho0);
                float rx = a.x - wrench_com()[obj][0], ry = a.y - wrench_com()[obj][1], rz = a.z - wrench_com()[obj][2];
                float tx = ry * fz - rz * fy, ty = rz * fx - rx * fz, tz = rx * fy - ry * fx;
                if (AF || o.dyn) {
                    const float cf = -cp * o.m0;
                    const float gxf = cf * dx, gyf = cf * dy, gzf = cf * dz;
                    rx = o.x - wrench_com()[obj][0]; ry = o.y - wrench_com()[obj][1]; rz = o.z - wrench_com()[obj][2];
                    tx += ry * gzf - rz * gyf; ty += rz * gxf - rx * gzf; tz += rx * gyf - ry * gxf;
                    fx += gxf; fy += gyf; fz += gzf;
                }
                add_wrench(scal, obj, fx, fy, fz, tx, ty, tz);
            }
        }
#else
        const float rn2 = g.rn * g.rn;      // base_solver.py:254 R.norm()**2
        float gx, gy, gz;
        kernGrad(c, dx, dy, dz, g, gx, gy, gz);
        const float v_xy = (o.vx - bj.x) * dx + (o.vy - bj.y) * dy + (o.vz - bj.z) * dz;
        if (AF || bj.w >= 0.0f) {
            const float cst = o.st_m * a.w;                             // surface tension (:210)
            const float w = r2 > c.diameter2 ? kernW(c, g) : c.Wd;
            o.sx -= (cst * dx) * w; o.sy -= (cst * dy) * w; o.sz -= (cst * dz) * w;
            const float m_ij = (o.m + a.w) * 0.5f;                      // viscosity (:232)
            const float cc = fdiv2(c.cv * m_ij, bj.w, rn2 + c.visc_eps) * v_xy;
            o.ax += cc * gx; o.ay += cc * gy; o.az += cc * gz;
            const float cp = -a.w * (o.pt + cj);                        // pressure (:136)
            o.px += cp * gx; o.py += cp * gy; o.pz += cp * gz;
        } else {
            const float cc = fdiv2(c.cvb * a.w, o.rho, rn2 + c.visc_eps) * v_xy;
            const float acx = cc * gx, acy = cc * gy, acz = cc * gz;
            o.ax += acx; o.ay += acy; o.az += acz;
            const float cp = fdiv(-a.w * o.p, o.rho2);
            o.px += cp * gx; o.py += cp * gy; o.pz += cp * gz;
            if (bj.w <= -2.0f) {  // dynamic rigid neighbour
                const int obj = (int)(-bj.w) - 2;
                // the viscous part (base_solver.py:272-278, torque about pos_j) and the pressure part (:174-187, torque about pos_i, sic) of
                // this pair go to the body's accumulators together: one add_wrench (six LDS adds) instead of two
                float fx = fdiv(-acx * o.m, c.rho0), fy = fdiv(-acy * o.m, c.rho0), fz = fdiv(-acz * o.m, c.rho0);
                float rx = a.x - wrench_com()[obj][0], ry = a.y - wrench_com()[obj][1], rz = a.z - wrench_com()[obj][2];
                float tx = ry * fz - rz * fy, ty = rz * fx - rx * fz, tz = rx * fy - ry * fx;
                if (AF || o.dyn) {
                    const float cf = fdiv(a.w * o.p, o.rho2);
                    const float gxf = (cf * gx) * o.m0, gyf = (cf * gy) * o.m0, gzf = (cf * gz) * o.m0;
                    rx = o.x - wrench_com()[obj][0]; ry = o.y - wrench_com()[obj][1]; rz = o.z - wrench_com()[obj][2];
                    tx += ry * gzf - rz * gyf; ty += rz * gxf - rx * gzf; tz += rx * gyf - ry * gxf;
                    fx += gxf; fy += gyf; fz += gzf;
                }
                add_wrench(scal, obj, fx, fy, fz, tx, ty, tz);
            }
        }
#endif
    }
    __device__ float finish(const Consts &c, int i, const float4 &pi, Own &o) const {
        // non-pressure update (:643 after :203-240)
        float ax = c.gx, ay = c.gy, az = c.gz;
        ax += o.sx; ay += o.sy; az += o.sz;
        ax += fdiv(o.ax, c.rho0); ay += fdiv(o.ay, c.rho0); az += fdiv(o.az, c.rho0);
        float vx = o.vx + c.dt * ax, vy = o.vy + c.dt * ay, vz = o.vz + c.dt * az;
        if (!AF && !o.dyn) {   // PressurePass::begin() leaves such a particle alone
            vel_out[i] = make_float4(vx, vy, vz, o.m);
            acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            posv_out[i] = pi;
            if (hs.on) halo_presend(c, hs, i, pi, make_float4(vx, vy, vz, o.m), rho[i]);
            if (NEXT_HASH && nh.on) o.lin_new = cell_of(c, pi.x, pi.y, pi.z);
            return 0.0f;
        }
        // pressure update, advection, boundary (:136, :643, :652, :575)
        acc[i] = make_float4(o.px, o.py, o.pz, 0.0f);
        vx = vx + c.dt * o.px; vy = vy + c.dt * o.py; vz = vz + c.dt * o.pz;
        float x = pi.x + c.dt * vx, y = pi.y + c.dt * vy, z = pi.z + c.dt * vz;
        enforce_boundary(c, x, y, z, vx, vy, vz);
        posv_out[i] = make_float4(x, y, z, pi.w);
        vel_out[i] = make_float4(vx, vy, vz, o.m);
        if (hs.on) halo_presend(c, hs, i, make_float4(x, y, z, pi.w), make_float4(vx, vy, vz, o.m), rho[i]);
        if (NEXT_HASH && nh.on) o.lin_new = cell_of(c, x, y, z);
        return 0.0f;
    }
    __device__ void passive(const Consts &c, int i, const float4 &pi) const {
        const float4 v = velm[i];
        vel_out[i] = v;
        acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        posv_out[i] = pi;
        if (hs.on) halo_presend(c, hs, i, pi, v, rho[i]);   // (last step's ghosts die here; a static particle in a boundary layer is copied)
    }
};

// ---------------------------------------------------------------------------------------
// base_solver.py:106 compute_rigid_particle_volume (+task :117).  i rigid, j same object.
struct RigidVolumePass {
    static constexpr int MODES = 0b001;               // runs before the density pass (no masks yet), rarely
    static constexpr int MAX_WAVES = 5;               // (one register over the six-wave budget since phase 1 became bottom-tested; it runs once per body)
    static constexpr int BLOCK = 256, GROUPS = 3;
    static constexpr bool USES_J = false;
    static constexpr bool HAS_B = false, COUNT_PAIRS = false;
    static constexpr int PAIR_WEIGHT = 0;
    static constexpr bool HAS_REDUCE = false;
    typedef int BT;
    struct Own { float sum; int obj; };
    float4 *posv; float4 *velm; const int *meta;

    __device__ float4 stage_impl(int j) const {
        const float4 p = posv[j];
        const int m = meta[j];
        const int key = META_MAT(m) == 2 ? META_OBJ(m) : -100;
        return make_float4(p.x, p.y, p.z, __int_as_float(key));
    }
    __device__ float4 loadA(int j) const { return stage_impl(j); }
    __device__ BT loadB(int) const { return 0; }
    __device__ float4 stage(const Consts &, int j, BT &) const { return stage_impl(j); }
    __device__ bool begin(const Consts &c, int i, const float4 &pi, Own &o) const {
        const int m = meta[i];
        if (META_MAT(m) != 2 || META_GHOST(m) || META_FRESH(m) || !(up_coord(c, pi) <= c.g_upper)) return false;
        o.obj = META_OBJ(m);
        o.sum = c.W0;
        return true;
    }
    __device__ void pair(const Consts &c, Own &o, float, float, float, float r2, const float4 &a, const BT &,
                         int) const {
        if (__float_as_int(a.w) == o.obj) o.sum += kernW(c, geom(c, r2));
    }
    __device__ float finish(const Consts &c, int i, const float4 &pi, Own &o) const {
        const float V = 1.0f / o.sum;
        posv[i] = make_float4(pi.x, pi.y, pi.z, V);  // only .w changes; readers use xyz + meta
        float4 v = velm[i];
        v.w = c.rho0 * V;
        velm[i] = v;
        return 0.0f;
    }
    __device__ void passive(const Consts &, int, const float4 &) const {}
};
