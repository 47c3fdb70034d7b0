// sph_cg_steps.hpp -- implicit viscosity (base_solver.py:509 implicit_viscosity_solve) orchestration
#pragma once

// gravity + surface tension + implicit viscosity + v += dt a  (base_solver.py:190-198, :643)
static int implicit_viscosity_non_pressure(SphHandle *h) {
    State &s = h->st;
    const int fixed = h->prm.fixed_iterations;
    { ProfScope p(h, SPH_K_CG_PREPARE); h->L->cg_prepare(s); }                     // :510
    { ProfScope p(h, SPH_K_CG_AP); h->L->cg_ap(s); }                               // :511
    { ProfScope p(h, SPH_K_CG_VECTOR); h->L->cg_prepare2(s); h->L->cg_alpha(s); }  // :512 (+ |r0|^2 for the first alpha)
    float tol = 1000.0f;
    int itr = 0;
    const int max_itr = fixed > 0 ? fixed : 1000;
    if (fixed <= 0) {   // :445 conjugate_gradient_loop, stop test on the device (see device_loop)
        int launched = 0;
        int rc = device_loop(h, max_itr, 3, 3, 1.0f, 1e-6, [&]() {
            { ProfScope p(h, SPH_K_CG_AP); h->L->cg_ap(s); }
            { ProfScope p(h, SPH_K_CG_VECTOR); h->L->cg_update_xr(s); h->L->cg_update_p(s); }
        }, &itr, &launched, &tol);
        if (rc) return rc;
    }
    while (fixed > 0 && itr < max_itr) {
        { ProfScope p(h, SPH_K_CG_AP); h->L->cg_ap(s); }
        { ProfScope p(h, SPH_K_CG_VECTOR); h->L->cg_update_xr(s); h->L->cg_update_p(s); }
        itr++;
        if (fixed > 0) continue;
        int rc = read_red(h, 3, &tol); if (rc) return rc;                          // :457 tol = cg_error[None]
    }
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
