// sph_comm.hpp -- z-slab communicator: neighbour-to-neighbour message transport.
//   kind 1: RCCL ncclSend/ncclRecv pairs over xGMI (production; one process per GPU)
//   kind 2: POSIX shared-memory mailboxes with host staging (same-node processes; lets several ranks share one
//           GPU, which is how the multi-rank device path is tested on a single-GPU box)
#pragma once
#include <atomic>
#include <string>

struct ShmMailbox {           // one per (writer rank, direction); written by `rank`, read by rank -/+ 1
    std::atomic<uint64_t> seq_written;
    std::atomic<uint64_t> seq_read;
    uint64_t nbytes;
    uint64_t pad[5];
};
#define SHM_MAX_RANKS 64
#define SHM_RED_MAX 128   /* doubles per sph_comm_allreduce (the rigid wrench of all objects is 120) */
struct ShmHeader {
    std::atomic<int> attached; std::atomic<int> detached; int nranks; int pad; uint64_t mbox_cap;
    // sph_comm_barrier / sph_comm_allreduce of the shm transport: arrival counter + generation, one row of doubles per rank
    std::atomic<uint64_t> bar_count, bar_gen;
    double red[SHM_MAX_RANKS][SHM_RED_MAX];
};

struct SlabComm {
    int kind = 0;             // 0 none, 1 rccl, 2 shm
    int rank = 0, nranks = 1;
    void *nccl = nullptr;     // ncclComm_t
    // shm transport
    void *shm_base = nullptr; size_t shm_size = 0; std::string shm_name; uint64_t seq = 0; uint64_t mbox_cap = 0;
    // per-step message sizes (particle records)
    int n_send[2] = {0, 0}, n_recv[2] = {0, 0};
    int *cnt_dev = nullptr;   // 4 ints: send counts [0..1], recv counts [2..3] (rccl size exchange)
    int *cnt_host = nullptr;  // pinned mirror
    double *red_dev = nullptr;   // SHM_RED_MAX doubles: sph_comm_allreduce (rccl)
    double *red_host = nullptr;  // pinned mirror
    float *self_dev = nullptr;   // sph_comm_selftest buffers (send | recv)
    int slab_ready = 0;          // halo buffers allocated (sph_comm_set_slab)
    int rebalance_every = 0;     // re-plan the slab cuts from the z histogram every this many steps (0: never)
    int rebalance_moves = 0;     // cuts moved so far (this rank's lower + upper face)
    int *hist_dev = nullptr;     // nz_glob + nranks ints: layer histogram | every rank's z_lo
    int *hist_host = nullptr;    // pinned mirror
    unsigned char id[128];
};
