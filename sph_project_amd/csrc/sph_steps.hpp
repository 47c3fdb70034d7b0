// sph_steps.hpp -- per-method step orchestration (included by sph_api.hip).
// Order of operations follows WCSPH.py:27, DFSPH.py:298, PCISPH.py:165 and base_solver.py:692.
#pragma once

// base_solver.py:190 compute_non_pressure_acceleration + :643 update_fluid_velocity
static int run_non_pressure(SphHandle *h) {
    if (h->prm.viscosity_implicit) return fail(h, SPH_ERR_UNSUPPORTED, "implicit viscosity is not built in this round");
    ProfScope p(h, SPH_K_NON_PRESSURE);
    h->L->non_pressure(h->st);
    return SPH_OK;
}

static int wcsph_step(SphHandle *h) {
    State &s = h->st;
    ph_neighbor_search(h);                                                    // WCSPH.py:28
    ph_rigid_volume(h);                                                       // base_solver.py:696 (see ph_rigid_volume)
    { ProfScope p(h, SPH_K_DENSITY); h->L->density(s, 1); }                   // :29 + :33 (EOS fused)
    int rc = run_non_pressure(h); if (rc) return rc;                          // :30-31
    { ProfScope p(h, SPH_K_PRESSURE_INTEGRATE); h->L->pressure_integrate(s); } // :34-36, :45
    return SPH_OK;
}

static int dfsph_step(SphHandle *h, bool allow_readback) {
    (void)allow_readback;
    return fail(h, SPH_ERR_UNSUPPORTED, "dfsph step not built yet");
}

static int pcisph_step(SphHandle *h, bool allow_readback) {
    (void)allow_readback;
    return fail(h, SPH_ERR_UNSUPPORTED, "pcisph step not built yet");
}

static int method_prepare(SphHandle *h) {
    if (h->prm.method == SPH_METHOD_WCSPH) return SPH_OK;
    return fail(h, SPH_ERR_UNSUPPORTED, "method %d not built yet", h->prm.method);
}

static int method_run_phase(SphHandle *h, int phase) {
    return fail(h, SPH_ERR_INVALID, "unknown phase %d", phase);
}
