// sph_cg_steps.hpp -- implicit viscosity (base_solver.py:509 implicit_viscosity_solve) orchestration
#pragma once

// gravity + surface tension + implicit viscosity + v += dt a  (base_solver.py:190-198, :643)
static int implicit_viscosity_non_pressure(SphHandle *h) {
    State &s = h->st;
    const int fixed = h->prm.fixed_iterations;
    const bool slab = s.slab_active != 0;
    int comm_rc = SPH_OK;
    // few fluid particles (the buckling sheet of C5: 106 k of 2.2 M particles = 416 tiles, one wave per SIMD): every A p pass is
    // pure latency; splitting it by x-offset group triples the waves in flight (PassSplit).  A scene that fills the chip
    // anyway (> ~1500 tiles) gains nothing from it and keeps the single launch.
    static const char *no_split = getenv("SPH_NO_CG_SPLIT");
    s.cg_split = (!no_split && h->n_fluid > 0 && h->n_fluid <= 400000) ? 3 : 0;
    // (SPH_CG_SPLIT_WAYS=2: groups {0, 1} and {2} -- 832 instead of 1248 workgroups at C5, all resident in one round at 4 per CU; built in
    //  round 5 on the theory that the second, mostly empty round of the three-way split costs a workgroup lifetime.  Measured: A p walk
    //  0.94 -> 1.08 ms per step, C5 1.639 -> 1.757 ms/step: the longer life of the two-group workgroups costs more.  Kept as a switch.)
    if (s.cg_split) { static const char *ways = getenv("SPH_CG_SPLIT_WAYS"); if (ways && atoi(ways) == 2) s.cg_split = 2; }
    // Slab sharding: the ghosts are the neighbour ranks' rows of the system.  Their search direction goes out before every
    // A p pass (12 B per ghost through the slot tables), the three dot products of an iteration are summed over the ranks
    // (k_cg_fold + one 1-float and one 2-float all-reduce), and the solved velocities of the ghosts after the loop.
    auto refresh = [&](float4 *arr) { if (slab && !comm_rc) comm_rc = slab_exchange_vel(h, arr); };
    auto dots = [&](int which) {
        if (!slab || comm_rc) return;
        { ProfScope p(h, SPH_K_CG_VECTOR); h->L->cg_fold(s, which); }
        comm_rc = which == 1 ? slab_allreduce_dev(h, &s.scal->red[7], 1) : slab_allreduce_dev(h, &s.scal->red[6], which == 0 ? 1 : 2);
    };
    // One CG iteration = A p pass (+ combine when split), x / r update, p update.  Unsharded, the p update is folded into the
    // NEXT iteration's A p pass (CgApPass::fuse: p = r + beta p_old on the fly while staging; beta, the error and the stop flag
    // come from the x / r update's partials): two or three launches per iteration instead of three or four.  Sharded, the
    // ghosts' p has to exist in memory to be exchanged, so the p update stays a kernel of its own.
    static const char *no_fuse = getenv("SPH_NO_CG_FUSED_P");
    const bool fused = !slab && !no_fuse && s.cg_p2;
    s.cg_fused_loop = 0;   // (set for the loop only: the A p pass in front of it feeds prepare2 through cg_Ap)
    bool first = true;
    auto iteration = [&]() {
        refresh(s.cg_p);
        s.cg_fuse = (fused && !first) ? 1 : 0;
        { ProfScope p(h, SPH_K_CG_AP); h->L->cg_ap(s); }
        s.cg_fuse = 0;
        dots(1);
        { ProfScope p(h, SPH_K_CG_VECTOR); h->L->cg_update_xr(s); }
        dots(2);
        if (!fused) { ProfScope p(h, SPH_K_CG_VECTOR); h->L->cg_update_p(s); }
        first = false;
    };
    { ProfScope p(h, SPH_K_CG_PREPARE); h->L->cg_prepare(s); }                     // :510
    refresh(s.cg_p);                                                               // the ghosts' initial guess
    { ProfScope p(h, SPH_K_CG_AP); h->L->cg_ap(s); }                               // :511
    { ProfScope p(h, SPH_K_CG_VECTOR); h->L->cg_prepare2(s); h->L->cg_alpha(s); }  // :512 (+ |r0|^2 for the first alpha)
    dots(0);
    s.cg_fused_loop = fused ? 1 : 0;
    float tol = 1000.0f;
    int itr = 0;
    const int max_itr = fixed > 0 ? fixed : 1000;
    if (fixed <= 0) {   // :445 conjugate_gradient_loop, stop test on the device (see device_loop)
        int launched = 0;
        int rc = device_loop(h, max_itr, 3, 3, 1.0f, 1e-6, iteration, &itr, &launched, &tol, fused ? h->L->cg_check : nullptr);
        if (rc) return rc;
    }
    while (fixed > 0 && itr < max_itr) {
        iteration();
        itr++;
    }
    s.cg_fused_loop = 0;
    if (comm_rc) return comm_rc;
    refresh(s.cg_x);                                                               // solved velocities of the ghosts (:514)
    if (comm_rc) return comm_rc;
    h->last.iter_cg = itr; h->last.err_cg = tol;
    // :514-516: the explicit viscosity formula evaluated with the solved velocities gives the acceleration; the
    // fused pass adds gravity + surface tension and advances the ORIGINAL velocities (:470, :643)
    s.np_visc_vel = s.cg_x;
    s.skip_viscosity = 0;
    { ProfScope p(h, SPH_K_NON_PRESSURE); h->L->non_pressure(s); }
    s.np_visc_vel = nullptr;
    { ProfScope p(h, SPH_K_CG_VECTOR); h->L->cg_prepare_guess(s); }                // :517
    return SPH_OK;
}
