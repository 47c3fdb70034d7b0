// sph_cg_steps.hpp -- implicit viscosity (base_solver.py:509) orchestration
#pragma once
static int implicit_viscosity_non_pressure(SphHandle *h) {
    return fail(h, SPH_ERR_UNSUPPORTED, "implicit viscosity is not built yet");
}
