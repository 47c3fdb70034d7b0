// sph_comm.hpp -- z-slab communicator.
// Control plane (barriers, all-reduces, setup messages):
//   kind 1: RCCL over xGMI (production; one process per GPU)
//   kind 2: POSIX shared-memory segment (same-node processes; lets several ranks share one GPU, which is how the multi-rank
//           device path is tested on a single-GPU box)
// Data plane (halo payload, neighbour to neighbour):
//   push   : device stores straight into the neighbour's inbox, mapped through hipIpc (xGMI peer access between GPUs, plain
//            device memory between two ranks of one GPU), message headers and numbers in the inbox, waits on the device
//            (sph_halo.hpp).  No copy engine, no host, no per-message launch of a communication kernel.  Default when its
//            set-up and self-test succeed on every rank.
//   rccl   : ncclSend / ncclRecv pairs inside one group (fallback of kind 1; SPH_COMM_TRANSPORT=rccl forces it)
//   shm    : host-staged mailboxes (kind 2; SPH_COMM_TRANSPORT=shm; "shm+ipc" = shm control plane + push data plane)
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
    int *cnt_dev = nullptr;   // 8 ints: [2..3] received counts, [4..5] my info words, [6..7] received info words (rccl size exchange)
    int *cnt_host = nullptr;  // pinned: [0..3] halo_counts, [4..11] cnt_dev
    int sticky_status = 0;    // SLAB_ST_* bits this rank has run into; travels in the next message header
    double *red_dev = nullptr;   // SHM_RED_MAX doubles: sph_comm_allreduce (rccl)
    double *red_host = nullptr;  // pinned mirror
    float *self_dev = nullptr;   // sph_comm_selftest buffers (send | recv)
    int slab_ready = 0;          // halo buffers allocated (sph_comm_set_slab)
    int rebalance_every = 0;     // re-plan the slab cuts from the z histogram every this many steps (0: never)
    int rebalance_moves = 0;     // cuts moved so far (this rank's lower + upper face)
    int *hist_dev = nullptr;     // nx_glob + nranks ints: layer histogram | every rank's z_lo
    int *hist_host = nullptr;    // pinned mirror
    unsigned char id[128];
    // data plane
    int push_wanted = 0;         // 0: never, 1: try, fall back to the control plane's transport if it cannot be set up, 2: required
    void *ipc_mapped[2] = {nullptr, nullptr};   // hipIpcOpenMemHandle results (closed in slab_comm_destroy)
    void *inbox_alloc = nullptr; // my inbox (hipFree)
    int *bad_dev = nullptr;      // self-test mismatch counter
    int *n_stage = nullptr;      // pinned: particle count on its way into SlabDyn::n_live
    // asynchronous WCSPH steps over the push transport: the host launches from bounds, the device keeps the counts
    int async_enabled = 0;       // this handle may run them (push transport + SPH_SLAB_ASYNC != 0)
    int bound_live = 0;          // launch bound of the running / last step for the post-sort particle count
    int est_recv = 0;            // records received per step, last known (sizes the first bound after an exact count)
    double timeout_s = 60.0;     // bounded host waits (SPH_COMM_TIMEOUT_S); the device waits are half as long
    char transport[48] = "none";
};
