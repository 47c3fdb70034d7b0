// sph_solvers_impl.hpp -- launchers of the iterative pressure solvers; included by sph_kernels.hip
#pragma once
static void register_solver_launchers(Launch &L) { (void)L; }
