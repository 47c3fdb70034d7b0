// sph_comm_api.hpp -- multi-GPU entry points (included by sph_api.hip)
#pragma once
extern "C" int sph_comm_unique_id(void *out128) { (void)out128; return SPH_ERR_UNSUPPORTED; }
extern "C" int sph_comm_init(SphHandle *h, int rank, int nranks, const void *id128) {
    (void)rank; (void)nranks; (void)id128;
    return fail(h, SPH_ERR_UNSUPPORTED, "multi-GPU slab sharding is not built yet");
}
extern "C" int sph_comm_set_slab(SphHandle *h, int z_lo, int z_hi) { (void)z_lo; (void)z_hi; return fail(h, SPH_ERR_UNSUPPORTED, "not built"); }
extern "C" int sph_comm_get_slab(SphHandle *h, int *z_lo, int *z_hi, int *n_owned, int *n_ghost) {
    (void)z_lo; (void)z_hi; (void)n_owned; (void)n_ghost; return fail(h, SPH_ERR_UNSUPPORTED, "not built");
}
