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
struct ShmHeader { std::atomic<int> attached; std::atomic<int> detached; int nranks; int pad; uint64_t mbox_cap; };

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
    unsigned char id[128];
};
