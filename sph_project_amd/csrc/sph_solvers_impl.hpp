// sph_solvers_impl.hpp -- launchers of the iterative pressure solvers; included by sph_kernels.hip
// inside the per-build namespace.
#pragma once

// DFSPH position update: base_solver.py:652 update_fluid_position + :575 enforce_domain_boundary_3D
// nh.on (round 6, all-fluid unsharded scenes inside a whole-step call): the kernel is also the k_hash_count of the sort that follows in
// the same step (DFSPH.py:316) -- it holds every particle's final position, its threads own the particles in sorted order like
// k_hash_count's: cell id, histogram atomic per run of equal cells, arrival rank, tile sums of the scan.  One launch less per step.
__global__ void __launch_bounds__(256)
k_advect_boundary(const Consts c, float4 *posv, float4 *velm, int *meta, const RigidPose *pose, int all_fluid, const NextHash nh) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < live_n(c);
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) {
        p = posv[i];
        float4 v = velm[i];
        const float4 v_in = v;
        const int m = all_fluid ? META_PACK(0, 1, 1) : meta[i];
        if (META_MAT(m) == 1) {
            p.x += c.dt * v.x; p.y += c.dt * v.y; p.z += c.dt * v.z;
            if (META_DYN(m)) enforce_boundary(c, p.x, p.y, p.z, v.x, v.y, v.z);
            posv[i] = p;
            if (__float_as_int(v.x) != __float_as_int(v_in.x) || __float_as_int(v.y) != __float_as_int(v_in.y) || __float_as_int(v.z) != __float_as_int(v_in.z)) velm[i] = v;   // (only a particle that hit a wall: 16 B per particle not written)
        } else if (up_coord(c, p) > c.g_upper) {  // emitter branch :660-666
            const int obj = META_OBJ(m);
            if (obj >= 0 && pose->material[obj] == 1) {
                p.x += c.dt * v.x; p.y += c.dt * v.y; p.z += c.dt * v.z;
                if (up_coord(c, p) <= c.g_upper) {
                    meta[i] = META_SET_MAT(m, 1);
                    if (META_DYN(m)) enforce_boundary(c, p.x, p.y, p.z, v.x, v.y, v.z);
                    velm[i] = v;
                }
                posv[i] = p;
            }
        }
    }
    if (nh.on) {   // (uniform)
        const int lane = threadIdx.x & 63;
        int lin = -1 - lane;
        if (valid) {
            lin = (cell_coord_x(c, p.x) * c.ny + cell_coord(p.y, c.grid_size, c.ny)) * c.nz + cell_coord_z(c, p.z);
            nh.cellid[i] = lin;
        }
        bool head; int hl, len;
        wave_runs(lin, lane, head, hl, len);
        int base = 0;
        if (head && valid) base = atomicAdd(&nh.cell_count[lin], len);
        const int base_run = __shfl(base, hl, 64);
        if (nh.rl.head && head && valid) run_list_file(nh.rl, lin, i, len, base);
        if (valid) nh.rank[i] = base_run + (lane - hl);
        if (nh.tile_sum) tile_sum_add(nh.tile_sum, lin, valid, c.G);
    }
}

static void l_advect_boundary(State &s) {
    NextHash nh{0, s.cellid, s.rank, s.cell_count, tile_sum_bank(s), RunList{nullptr, nullptr, 0, 0u}};
    if (s.nexthash.on && s.c.all_fluid && !s.slab_active && s.cell_count_clean && s.c.n > 0) nh.on = 1;
    s.nexthash.on = 0;
    if (s.c.n == 0) return;
    if (nh.on) nh.rl = run_list_of(s, true);
    s.masks_valid = 0;  // positions move
    hipLaunchKernelGGL(k_advect_boundary, dim3(cdiv(s.c.n, 256)), dim3(256), 0, s.stream, s.c, s.posv.cur(),
                       s.velm.cur(), s.meta.cur(), s.pose, s.c.all_fluid, nh);
    if (nh.on) { s.prehashed = 1; s.cell_count_clean = 0; s.tile_sums_ready = nh.tile_sum != nullptr; s.hist_taken = 1; s.run_lists_filed = nh.rl.head != nullptr; }
}

static void l_reduce_sum(State &s, int slot, int nblocks) {
    // under slab sharding the residual is a sum over ranks: this kernel leaves the local sum, the step orchestration
    // all-reduces it and runs the stop test (slab_finish_reduction, k_loop_criterion)
    const int kind = (s.loop_flag && s.loop_slot == slot && !s.slab_active) ? s.loop_kind : 0;
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, s.stream, s.red_partial, nblocks, s.scal, slot, kind,
                       s.loop_denom, s.loop_thr, s.last_pass_listed ? s.blk_list : nullptr, s.last_pass_listed ? s.blk_count : nullptr);
}

static void l_dfsph_density_alpha(State &s) {
    if (s.c.all_fluid) { DfsphDensityAlphaPass<true> p{s.posv.cur(), s.meta.cur(), s.rho.cur(), s.alpha, s.red_partial}; launch_pass(s, p, 1); }
    else { DfsphDensityAlphaPass<false> p{s.posv.cur(), s.meta.cur(), s.rho.cur(), s.alpha, s.red_partial}; launch_pass(s, p, 1); }
}

// density + alpha + the density derivative that opens the divergence solve, one walk (DfsphDensityAlphaDivPass); the partial sums
// of the residual are left unreduced: the reference does not look at the pre-loop value either (DFSPH.py:140-150)
static void l_dfsph_density_alpha_div(State &s) {
    if (s.c.all_fluid) { DfsphDensityAlphaDivPass<true> p{s.posv.cur(), s.velm.cur(), s.meta.cur(), s.rho.cur(), s.alpha, s.rho_deriv, s.kappa_v_next, s.red_partial}; launch_pass(s, p, 1); }
    else { DfsphDensityAlphaDivPass<false> p{s.posv.cur(), s.velm.cur(), s.meta.cur(), s.rho.cur(), s.alpha, s.rho_deriv, s.kappa_v_next, s.red_partial}; launch_pass(s, p, 1); }
}

template <int MODE> static void dfsph_rho_adv_t(State &s, int slot) {
    float *out_adv = MODE == 0 ? s.rho_deriv : s.rho_star;
    float *out_k = MODE == 0 ? s.kappa_v_next : s.kappa_next;
    if (s.c.all_fluid) {
        DfsphRhoAdvPass<true, MODE> p{s.posv.cur(), s.velm.cur(), s.meta.cur(), s.rho.cur(), s.alpha, out_adv, out_k, s.red_partial};
        launch_pass(s, p, 2);
    } else {
        DfsphRhoAdvPass<false, MODE> p{s.posv.cur(), s.velm.cur(), s.meta.cur(), s.rho.cur(), s.alpha, out_adv, out_k, s.red_partial};
        launch_pass(s, p, 2);
    }
    if (s.c.n > 0 && !s.skip_residual) l_reduce_sum(s, slot, cdiv(s.c.n, NBR_BLOCK));
}
static void l_dfsph_rho_adv(State &s, int mode) { if (mode == 0) dfsph_rho_adv_t<0>(s, 0); else dfsph_rho_adv_t<1>(s, 1); }

template <int MODE> static void dfsph_correct_t(State &s) {
    const float *kap = MODE == 0 ? s.kappa_v : s.kappa;
    if (s.c.all_fluid) {
        DfsphCorrectPass<true, MODE> p{s.posv.cur(), s.meta.cur(), kap, s.rho.cur(), s.velm.cur(), s.scal, s.pose, s.c.rho0, s.red_partial};
        launch_pass(s, p, 2);
    } else {
        DfsphCorrectPass<false, MODE> p{s.posv.cur(), s.meta.cur(), kap, s.rho.cur(), s.velm.cur(), s.scal, s.pose, s.c.rho0, s.red_partial};
        launch_pass(s, p, 2);
    }
}
static void l_dfsph_correct(State &s, int mode) { if (mode == 0) dfsph_correct_t<0>(s); else dfsph_correct_t<1>(s); }

// PCISPH.py:154 init_step
__global__ void __launch_bounds__(256)
k_pcisph_init(const Consts c, const float4 *posv, const float4 *vel0, const float4 *acc_np, const int *meta,
              float *prs, float *ptm, float4 *pacc, float4 *pvel, float4 *ppos, int all_fluid) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= live_n(c)) return;
    prs[i] = 0.0f; ptm[i] = 0.0f;
    pacc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!all_fluid && META_MAT(meta[i]) != 1) return;
    const float4 x = posv[i], v = vel0[i], a = acc_np[i];
    const float vx = v.x + c.dt * a.x, vy = v.y + c.dt * a.y, vz = v.z + c.dt * a.z;
    pvel[i] = make_float4(vx, vy, vz, 0.f);
    ppos[i] = make_float4(x.x + c.dt * vx, x.y + c.dt * vy, x.z + c.dt * vz, 0.f);
}

// called after the non-pressure pass: velm.cur() = v*, velm.alt() = v (start of step)
static void l_pcisph_init(State &s) {
    if (s.c.n == 0) return;
    hipLaunchKernelGGL(k_pcisph_init, dim3(cdiv(s.c.n, 256)), dim3(256), 0, s.stream, s.c, s.posv.cur(), s.velm.alt(),
                       s.acc_np, s.meta.cur(), s.prs, s.ptm, s.pacc, s.pvel, s.ppos, s.c.all_fluid);
}

static void l_pcisph_rho_star(State &s) {
    if (s.c.all_fluid) {
        PcisphRhoStarPass<true> p{s.posv.cur(), s.ppos, s.meta.cur(), s.rho.cur(), s.rho_star, s.prs, s.ptm, s.red_partial
#ifdef SPH_TEST_HOOKS
                                  , s.rho_raw   // (PCISPH reads rho_raw nowhere after the density pass)
#endif
        };
        launch_pass(s, p, 2);
    } else {
        PcisphRhoStarPass<false> p{s.posv.cur(), s.ppos, s.meta.cur(), s.rho.cur(), s.rho_star, s.prs, s.ptm, s.red_partial
#ifdef SPH_TEST_HOOKS
                                   , s.rho_raw
#endif
        };
        launch_pass(s, p, 2);
    }
    // the partial sums are finished by l_pcisph_pressure_accel: the criterion (PCISPH.py:122) is tested after the whole
    // iteration, so inside a device-controlled loop the stop flag must not rise before the second pass has run
}

static void l_pcisph_pressure_accel(State &s) {
    if (s.c.all_fluid) {
        PcisphPressureAccelPass<true> p{s.posv.cur(), s.velm.cur(), s.meta.cur(), s.ptm, s.acc_np, s.velm.alt(), s.pacc, s.pvel, s.ppos, s.c.rho0, s.red_partial};
        launch_pass(s, p, 2);
    } else {
        PcisphPressureAccelPass<false> p{s.posv.cur(), s.velm.cur(), s.meta.cur(), s.ptm, s.acc_np, s.velm.alt(), s.pacc, s.pvel, s.ppos, s.c.rho0, s.red_partial};
        launch_pass(s, p, 2);
    }
    if (s.c.n > 0 && !s.skip_residual) l_reduce_sum(s, 2, cdiv(s.c.n, NBR_BLOCK));   // density error of this iteration's rho* pass
}

// ---- implicit viscosity
// WCSPH clamps particle_densities only after the non-pressure accelerations (WCSPH.py:29-33): its solve reads the unclamped
// densities, like the explicit viscosity does (see l_non_pressure)
#define CG_RHO (s.visc_rho_raw ? s.rho_raw : s.rho.cur())
static void l_cg_prepare(State &s) {
    if (s.c.all_fluid) { CgPreparePass<true> p{s.posv.cur(), s.velm.cur(), s.meta.cur(), CG_RHO, s.cg_x, s.cg_p, s.cg_b, s.cg_r, s.cg_Ap, s.cg_v0, s.cg_dinv, s.c.rho0, s.red_partial}; launch_pass(s, p, 2); }
    else { CgPreparePass<false> p{s.posv.cur(), s.velm.cur(), s.meta.cur(), CG_RHO, s.cg_x, s.cg_p, s.cg_b, s.cg_r, s.cg_Ap, s.cg_v0, s.cg_dinv, s.c.rho0, s.red_partial}; launch_pass(s, p, 2); }
}
// partial-sum arrays of the CG kernels inside red_partial (each red_blocks floats): rr ping-pong, p . Ap, |r_old|^2
#define CG_PART(k) (s.red_partial + (size_t)(k) * s.red_blocks)
#define CG_AF (s.c.all_fluid && !s.c.ghosts)   /* "every particle is a row of the system" */
// slab sharding: all-reduced dot products live in scal->red[6..7] (see k_cg_fold)
#define CG_GLOB (s.slab_active ? &s.scal->red[6] : (const float *)nullptr)
// the per-particle CG kernels run the workgroups that hold fluid only (same list as the neighbour passes)
// (the same condition as launch_pass's use_list: the A p pass and the vector kernels must agree on where the partial sums are filed, red_slot)
#define CG_LISTED (!s.c.all_fluid && s.list_n == s.c.n && s.c.force_global == 0)
#define CG_LIST CG_LISTED ? s.blk_list : nullptr, CG_LISTED ? s.blk_count : nullptr
// grid of a per-particle CG kernel: the listed workgroups only once the host knows how many there are (list_grid)
#define CG_GRID(nb) (CG_LISTED ? list_grid(s, (nb)) : (nb))
static void l_cg_ap(State &s) {
    const bool split = s.cg_split && s.cg_part && s.c.n > 0;
    s.split_next_pass = split ? s.cg_split : 0;   // 2 or 3 ways (sph_cg_steps.hpp)
    // s.cg_fuse: this A p pass applies the previous iteration's p update on the fly (CgApPass::fuse), p_old = cg_p, p_new = cg_p2
    const int fuse = s.cg_fuse ? 1 : 0;
    const bool lst = CG_LISTED;
    const int nb = s.c.n > 0 ? cdiv(s.c.n, 256) : 0;
    // inside the unsharded loop with the fused p update the split walks leave their shares of p . A p themselves (CG_PART(4..6)) and the
    // x / r update adds the three parts up: no combining kernel (SPH_CG_COMBINE=1: the round-3 sequence, for A/B)
    static const bool keep_combine = getenv("SPH_CG_COMBINE") != nullptr;
    const bool nocombine = split && s.cg_fused_loop && !s.slab_active && !keep_combine && s.red_blocks >= nb;
    s.cg_nocombine = nocombine ? 1 : 0;
    float *pdot = nocombine ? CG_PART(4) : nullptr;
    if (s.c.all_fluid) {
        CgApPass<true> p{s.posv.cur(), s.velm.cur(), s.meta.cur(), CG_RHO, s.cg_p, s.cg_dinv, s.cg_Ap, CG_PART(2), s.cg_part, s.cap,
                         s.cg_r, s.cg_p2, CG_PART(s.cg_parity), CG_PART(3), nb, lst ? s.blk_list : nullptr, lst ? s.blk_count : nullptr,
                         fuse, s.loop_flag ? 1 : 0, (float)s.loop_thr, 0.0f, pdot, s.red_blocks};
        launch_pass(s, p, 2);
    } else {
        CgApPass<false> p{s.posv.cur(), s.velm.cur(), s.meta.cur(), CG_RHO, s.cg_p, s.cg_dinv, s.cg_Ap, CG_PART(2), s.cg_part, s.cap,
                          s.cg_r, s.cg_p2, CG_PART(s.cg_parity), CG_PART(3), nb, lst ? s.blk_list : nullptr, lst ? s.blk_count : nullptr,
                          fuse, s.loop_flag ? 1 : 0, (float)s.loop_thr, 0.0f, pdot, s.red_blocks};
        launch_pass(s, p, 2);
    }
    if (split && !nocombine)   // the three parts -> A p and the partials of p . A p (what finish() and the pass's reduction do otherwise)
        hipLaunchKernelGGL(k_cg_ap_combine, dim3(CG_GRID(cdiv(s.c.n, 256))), dim3(256), 0, s.stream, s.c, s.meta.cur(), CG_AF, s.cg_part, s.cap, s.cg_p,
                           s.cg_Ap, CG_PART(2), s.loop_flag, CG_LIST, fuse ? s.cg_r : (const float4 *)nullptr, s.cg_p2, s.scal);
    if (fuse) std::swap(s.cg_p, s.cg_p2);   // cg_p is the search direction of the running iteration again
}
static void l_cg_prepare2(State &s) {
    if (s.c.n == 0) return;
    hipLaunchKernelGGL(k_cg_prepare2, dim3(cdiv(s.c.n, 256)), dim3(256), 0, s.stream, s.c.n, s.meta.cur(), CG_AF, s.cg_dinv, s.cg_b, s.cg_Ap, s.cg_r, s.cg_p);
}
// |r0|^2 partials of a fresh solve (the numerator of the first alpha); later iterations reuse the |new r|^2 partials of
// the previous x / r update, which are the same numbers
static void l_cg_alpha(State &s) {
    if (s.c.n == 0) return;
    const int nb = cdiv(s.c.n, 256);
    s.cg_parity = 0;
    hipLaunchKernelGGL(k_cg_dots, dim3(CG_GRID(nb)), dim3(256), 0, s.stream, s.c.n, s.meta.cur(), CG_AF, s.cg_r, s.cg_p, s.cg_Ap, CG_PART(0), CG_PART(3), (const int *)nullptr, CG_LIST);
}
// slab sharding: this rank's sums of two partial arrays -> scal->red[6], red[7] (which = 0: |r0|^2 after l_cg_alpha,
// 1: p . Ap after the A p pass -> red[7] only, 2: |new r|^2 and |old r|^2 after the x / r update)
static void l_cg_fold(State &s, int which) {
    const int nb = s.c.n > 0 ? cdiv(s.c.n, 256) : 0;
    const float *a = which == 0 ? CG_PART(0) : which == 1 ? (const float *)nullptr : CG_PART(1 - s.cg_parity);
    const float *b = which == 0 ? (const float *)nullptr : which == 1 ? CG_PART(2) : CG_PART(3);
    hipLaunchKernelGGL(k_cg_fold, dim3(1), dim3(256), 0, s.stream, nb, a, b, &s.scal->red[6], &s.scal->red[7],
                       which == 0 ? (const int *)nullptr : s.loop_flag, CG_LIST);
}
static void l_cg_update_xr(State &s) {
    if (s.c.n == 0) return;
    const int nb = cdiv(s.c.n, 256);
    const bool nc = s.cg_nocombine != 0;   // the A p pass in front of this kernel left three parts and three sets of p . A p partials
    hipLaunchKernelGGL(k_cg_update_xr2, dim3(CG_GRID(nb)), dim3(256), 0, s.stream, s.c, s.c.n, nb, s.meta.cur(), CG_AF, s.cg_x, s.cg_r, s.cg_p, s.cg_Ap,
                       CG_PART(s.cg_parity), nc ? CG_PART(4) : CG_PART(2), CG_PART(1 - s.cg_parity), CG_PART(3), s.scal, s.loop_flag, CG_LIST, CG_GLOB,
                       (s.cg_fused_loop && s.loop_flag) ? 1 : 0, nc ? s.cg_part : (const float4 *)nullptr, s.cap, s.red_blocks);
    s.cg_nocombine = 0;
    // fused p update: nothing else closes the iteration -- the partials just written are what the next A p pass reads
    if (s.cg_fused_loop) s.cg_parity = 1 - s.cg_parity;
}
static void l_cg_update_p(State &s) {
    if (s.c.n == 0) return;
    const int nb = cdiv(s.c.n, 256);
    hipLaunchKernelGGL(k_cg_update_p2, dim3(CG_GRID(nb)), dim3(256), 0, s.stream, s.c.n, nb, s.meta.cur(), CG_AF, s.cg_r, s.cg_p,
                       CG_PART(1 - s.cg_parity), CG_PART(3), s.scal, s.loop_flag ? 1 : 0, (float)s.loop_thr, s.loop_flag, CG_LIST, CG_GLOB);
    s.cg_parity = 1 - s.cg_parity;
}
static void l_cg_check(State &s) {
    if (s.c.n == 0 || !s.loop_flag) return;
    hipLaunchKernelGGL(k_cg_check, dim3(1), dim3(256), 0, s.stream, cdiv(s.c.n, 256), CG_PART(s.cg_parity), s.scal, (float)s.loop_thr, CG_LIST);
}
static void l_cg_prepare_guess(State &s) {
    if (s.c.n == 0) return;
    hipLaunchKernelGGL(k_cg_prepare_guess, dim3(cdiv(s.c.n, 256)), dim3(256), 0, s.stream, s.c.n, s.meta.cur(), CG_AF, s.cg_x, s.cg_v0);
}

static void register_solver_launchers(Launch &L) {
    L.cg_prepare = l_cg_prepare; L.cg_ap = l_cg_ap; L.cg_prepare2 = l_cg_prepare2; L.cg_alpha = l_cg_alpha;
    L.cg_check = l_cg_check;
    L.cg_update_xr = l_cg_update_xr; L.cg_update_p = l_cg_update_p; L.cg_prepare_guess = l_cg_prepare_guess; L.cg_fold = l_cg_fold;
    L.dfsph_density_alpha = l_dfsph_density_alpha;
    L.dfsph_density_alpha_div = l_dfsph_density_alpha_div;
    L.dfsph_rho_adv = l_dfsph_rho_adv;
    L.dfsph_correct = l_dfsph_correct;
    L.advect_boundary = l_advect_boundary;
    L.pcisph_init = l_pcisph_init;
    L.pcisph_rho_star = l_pcisph_rho_star;
    L.pcisph_pressure_accel = l_pcisph_pressure_accel;
}
