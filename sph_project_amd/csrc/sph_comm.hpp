// sph_comm.hpp -- z-slab communicator state (RCCL point-to-point over xGMI).
#pragma once
struct SlabComm {
    void *nccl = nullptr;   // ncclComm_t
    int rank = 0, nranks = 1;
    int z_lo = 0, z_hi = 0; // owned global cell layers [z_lo, z_hi)
    bool active = false;
};
static inline void slab_comm_destroy(SlabComm &c) { (void)c; }
