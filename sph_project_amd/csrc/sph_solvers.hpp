// sph_solvers.hpp -- DFSPH / PCISPH / implicit-viscosity pass functors (filled in sph_solvers_impl.hpp)
#pragma once
#include "sph_passes.hpp"
