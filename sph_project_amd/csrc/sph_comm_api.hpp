// sph_comm_api.hpp -- multi-GPU entry points and the slab exchange orchestration (included by sph_api.hip)
#pragma once
#include <rccl/rccl.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <time.h>

#define NCCLCHK(h, call)                                                                                   \
    do {                                                                                                   \
        ncclResult_t r_ = (call);                                                                          \
        if (r_ != ncclSuccess) return fail((h), SPH_ERR_COMM, "%s failed: %s", #call, ncclGetErrorString(r_)); \
    } while (0)

static int push_setup(SphHandle *h);
static int slab_settle(SphHandle *h);

static inline ShmMailbox *shm_mbox(SlabComm &c, int writer, int dir) {
    char *base = (char *)c.shm_base + sizeof(ShmHeader);
    const size_t stride = sizeof(ShmMailbox) + c.mbox_cap;
    return (ShmMailbox *)(base + ((size_t)writer * 2 + dir) * stride);
}

static void slab_comm_destroy(SlabComm &c) {
    for (int side = 0; side < 2; ++side) if (c.ipc_mapped[side]) hipIpcCloseMemHandle(c.ipc_mapped[side]);
    if (c.inbox_alloc) hipFree(c.inbox_alloc);
    if (c.bad_dev) hipFree(c.bad_dev);
    if (c.n_stage) hipHostFree(c.n_stage);
    if (c.kind == 1 && c.nccl) ncclCommDestroy((ncclComm_t)c.nccl);
    if (c.kind == 2 && c.shm_base) {
        ShmHeader *hd = (ShmHeader *)c.shm_base;
        hd->detached.fetch_add(1);
        munmap(c.shm_base, c.shm_size);
        shm_unlink(c.shm_name.c_str());  // idempotent: first rank out removes the name
    }
    if (c.cnt_dev) hipFree(c.cnt_dev);
    if (c.cnt_host) hipHostFree(c.cnt_host);
    if (c.red_dev) hipFree(c.red_dev);
    if (c.red_host) hipHostFree(c.red_host);
    if (c.self_dev) hipFree(c.self_dev);
    if (c.hist_dev) hipFree(c.hist_dev);
    if (c.hist_host) hipHostFree(c.hist_host);
    c = SlabComm();
}

extern "C" int sph_comm_unique_id(void *out128) {
    if (!out128) return SPH_ERR_INVALID;
    memset(out128, 0, 128);
    const char *t = getenv("SPH_COMM_TRANSPORT");
    if (!(t && !strncmp(t, "shm", 3))) {
        ncclUniqueId id;
        if (ncclGetUniqueId(&id) == ncclSuccess) { memcpy(out128, &id, sizeof(id) < 128 ? sizeof(id) : 128); return SPH_OK; }
    }
    int fd = open("/dev/urandom", O_RDONLY);
    if (fd < 0) return SPH_ERR_COMM;
    ssize_t got = read(fd, out128, 128);
    close(fd);
    return got == 128 ? SPH_OK : SPH_ERR_COMM;
}

static int shm_attach(SphHandle *h, SlabComm &c, size_t mbox_cap) {
    char name[64];
    snprintf(name, sizeof(name), "/sphhalo_%02x%02x%02x%02x%02x%02x%02x%02x", c.id[8], c.id[9], c.id[10], c.id[11], c.id[12],
             c.id[13], c.id[14], c.id[15]);
    c.shm_name = name;
    c.mbox_cap = mbox_cap;
    c.shm_size = sizeof(ShmHeader) + (size_t)c.nranks * 2 * (sizeof(ShmMailbox) + mbox_cap);
    int fd = -1;
    if (c.rank == 0) {
        shm_unlink(name);
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) return fail(h, SPH_ERR_COMM, "shm_open(%s) failed", name);
        if (ftruncate(fd, (off_t)c.shm_size) != 0) { close(fd); return fail(h, SPH_ERR_COMM, "ftruncate(%s) failed", name); }
    } else {
        for (int tries = 0; tries < 20000 && fd < 0; ++tries) {  // up to ~20 s for rank 0 to create it
            fd = shm_open(name, O_RDWR, 0600);
            if (fd >= 0) { struct stat st; if (fstat(fd, &st) != 0 || (size_t)st.st_size < c.shm_size) { close(fd); fd = -1; } }
            if (fd < 0) usleep(1000);
        }
        if (fd < 0) return fail(h, SPH_ERR_COMM, "shm_open(%s): rank 0 never created the segment", name);
    }
    c.shm_base = mmap(nullptr, c.shm_size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c.shm_base == MAP_FAILED) { c.shm_base = nullptr; return fail(h, SPH_ERR_COMM, "mmap(%s) failed", name); }
    ShmHeader *hd = (ShmHeader *)c.shm_base;
    if (c.rank == 0) { hd->nranks = c.nranks; hd->mbox_cap = mbox_cap; }
    hd->attached.fetch_add(1);
    for (int tries = 0; hd->attached.load() < c.nranks; ++tries) {
        if (tries > 60000) return fail(h, SPH_ERR_COMM, "shm transport: only %d of %d ranks attached", hd->attached.load(), c.nranks);
        usleep(1000);
    }
    return SPH_OK;
}

extern "C" int sph_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

// Communicator only: rank / size / transport.  Usable without slab sharding (bench replicas: barrier + all-reduce);
// sph_comm_set_slab() turns the handle into one z-slab of the global grid.
extern "C" int sph_comm_init(SphHandle *h, int rank, int nranks, const void *id128) {
    if (!h || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(h, SPH_ERR_INVALID, "comm_init: bad arguments");
    if (h->comm.kind) return fail(h, SPH_ERR_INVALID, "comm_init: already initialised");
    HIPCHK(h, hipSetDevice(h->device));
    State &s = h->st;
    SlabComm &c = h->comm;
    c.rank = rank; c.nranks = nranks;
    memcpy(c.id, id128, 128);
    // Message capacity = particle capacity: a face message can never hold more records than this rank has particles, so
    // no pile-up in the boundary layers can overflow it (256 + 32 bytes per particle of capacity for the four message
    // buffers -- sized for the 64-byte records of scenes with a dynamic rigid body -- and the eight slot tables: ~3 GB at
    // 10 M particles, nothing next to 288 GB of HBM).
    size_t cap = (size_t)s.cap < 1024 ? 1024 : (size_t)s.cap;
    s.halo_cap = (int)cap;
    // the shared-memory test transport keeps its mailboxes at a z-face's worth (24 particles per cell, two layers)
    const size_t face = (size_t)s.c.nx * s.c.ny;
    size_t mbox_records = face * 24 * 2;
    if (mbox_records > cap) mbox_records = cap;
    if (mbox_records < 1024) mbox_records = 1024;
    HIPCHK(h, hipMalloc((void **)&c.cnt_dev, 8 * sizeof(int)));
    HIPCHK(h, hipHostMalloc((void **)&c.cnt_host, 12 * sizeof(int), hipHostMallocDefault));
    HIPCHK(h, hipMalloc((void **)&c.red_dev, SHM_RED_MAX * sizeof(double)));
    HIPCHK(h, hipHostMalloc((void **)&c.red_host, SHM_RED_MAX * sizeof(double), hipHostMallocDefault));
    // SPH_COMM_TRANSPORT: unset / "auto" = RCCL control plane, push data plane if it can be set up on every rank, else RCCL
    // send/recv; "ipc" = the same, but a push transport that cannot be set up is an error; "rccl" = RCCL only; "shm" = shared-
    // memory control plane + host-staged mailboxes; "shm+ipc" = shared-memory control plane + push data plane (several ranks
    // on one GPU: the test rig of the push transport)
    const char *t = getenv("SPH_COMM_TRANSPORT");
    // ("shm+auto": shared-memory control plane, push data plane if it can be set up on every rank, else the mailboxes -- the test rig of
    //  the all-ranks fall-back decision)
    if (t && strcmp(t, "auto") && strcmp(t, "ipc") && strcmp(t, "rccl") && strcmp(t, "shm") && strcmp(t, "shm+ipc") && strcmp(t, "shm+auto"))
        return fail(h, SPH_ERR_INVALID, "SPH_COMM_TRANSPORT=%s: expected auto, ipc, rccl, shm, shm+ipc or shm+auto", t);
    c.push_wanted = (!t || !strcmp(t, "auto") || !strcmp(t, "shm+auto")) ? 1 : ((!strcmp(t, "ipc") || !strcmp(t, "shm+ipc")) ? 2 : 0);
    if (const char *to = getenv("SPH_COMM_TIMEOUT_S")) { const double v = atof(to); if (v > 0.0) c.timeout_s = v; }
    if (t && !strncmp(t, "shm", 3)) {
        if (nranks > SHM_MAX_RANKS) return fail(h, SPH_ERR_INVALID, "shm transport: at most %d ranks", SHM_MAX_RANKS);
        int rc = shm_attach(h, c, mbox_records * 64);
        if (rc) return rc;
        c.kind = 2;
        snprintf(c.transport, sizeof(c.transport), "shm");
    } else {
        ncclUniqueId id;
        memset(&id, 0, sizeof(id));
        memcpy(&id, id128, sizeof(id) < 128 ? sizeof(id) : 128);
        ncclComm_t comm;
        NCCLCHK(h, ncclCommInitRank(&comm, nranks, id, rank));
        c.nccl = comm;
        c.kind = 1;
        snprintf(c.transport, sizeof(c.transport), "rccl");
    }
    return SPH_OK;
}

extern "C" const char *sph_comm_transport(SphHandle *h) { return (h && h->comm.kind) ? h->comm.transport : "none"; }

// Host waits with an end: a neighbour that died (or never sent) must not hang this process for ever.  Polls the stream;
// on time-out the RCCL communicator is aborted (its kernels may be what blocks the stream) and the call fails.
static int stream_sync_bounded(SphHandle *h, const char *what) {
    State &s = h->st;
    struct timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned spins = 0;; ++spins) {
        const hipError_t e = hipStreamQuery(s.stream);
        if (e == hipSuccess) return SPH_OK;
        if (e != hipErrorNotReady) return fail(h, SPH_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
        if (spins > 2000) usleep(spins > 20000 ? 200 : 20);   // the first ~2000 polls spin: a step is a fraction of a millisecond
        if ((spins & 255) == 255) {
            struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > h->comm.timeout_s) {
                if (h->comm.kind == 1 && h->comm.nccl) ncclCommAbort((ncclComm_t)h->comm.nccl), h->comm.nccl = nullptr, h->comm.kind = 0;
                return fail(h, SPH_ERR_COMM, "%s: no answer within %.0f s (a neighbour rank is gone or stuck)", what, h->comm.timeout_s);
            }
        }
    }
}

static const char *slab_status_text(int st) {
    static thread_local char buf[256];
    snprintf(buf, sizeof(buf), "%s%s%s%s%s%s",
             (st & SLAB_ST_SEND_OVERFLOW) ? "a halo message of this rank exceeds the message capacity; " : "",
             (st & SLAB_ST_PEER) ? "a neighbour rank reported a failure; " : "",
             (st & SLAB_ST_STRIDE) ? "halo record sizes differ between neighbours (a dynamic rigid body must be registered with sph_set_object on EVERY rank); " : "",
             (st & SLAB_ST_CAPACITY) ? "own + received particles exceed particle_max_num; " : "",
             (st & SLAB_ST_TIMEOUT) ? "a neighbour's halo message did not arrive in time; " : "",
             (st & SLAB_ST_BOUND) ? "the particle count outran the launch bound of an asynchronous step (SPH_SLAB_ASYNC=0 runs exact launches); " : "");
    return buf;
}

extern "C" int sph_comm_set_slab(SphHandle *h, int z_lo, int z_hi) {
    if (!h || !h->comm.kind) return fail(h, SPH_ERR_INVALID, "comm_set_slab: communicator not initialised");
    Consts &c = h->st.c;
    State &s = h->st;
    if (h->n > 0) return fail(h, SPH_ERR_INVALID, "comm_set_slab: set the slab before particles are appended");
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->comm.slab_ready) {
        // Which library axis the slabs are cut along (Consts::slab_axis, sph_common.hpp).  Default: z, the scene's own z and the fastest
        // axis of the cell order -- no permutation.  SPH_SLAB_LAYOUT=slow: the scene's z becomes the library's x, the slowest axis
        // (library axes = scene z, x, y; SPH_AXIS_ORDER may name another order that starts with z), which is what the boundary-first /
        // interior-later launches of the compute / halo overlap need.  Nothing has been appended yet; the constants are derived again.
        const char *lay = getenv("SPH_SLAB_LAYOUT");
        const bool slow = lay && (lay[0] == 's' || lay[0] == 'S');
        const char *ord = getenv("SPH_AXIS_ORDER");
        if (slow) {
            // (the permutation is derived HERE, after sph_create: a pose pushed earlier sits in pose_h in the old library frame and would
            //  be silently wrong -- ADVICE r04; poses go in after the slab is set)
            if (h->pose_given) return fail(h, SPH_ERR_INVALID, "comm_set_slab: SPH_SLAB_LAYOUT=slow changes the library frame; call sph_set_rigid_pose after sph_comm_set_slab, not before");
            if (!set_axis_order(h, ord ? ord : "zxy") || h->perm[0] != 2)
                return fail(h, SPH_ERR_INVALID, "comm_set_slab: with SPH_SLAB_LAYOUT=slow, SPH_AXIS_ORDER must be a permutation of xyz that starts with z");
            h->slab_axis = 0;
        } else {
            if (h->perm[2] != 2) return fail(h, SPH_ERR_INVALID, "comm_set_slab: SPH_AXIS_ORDER must end with z unless SPH_SLAB_LAYOUT=slow");
            h->slab_axis = 2;
        }
        fill_consts(h);
        refresh_counts(h);
    }
    if (z_lo < 0 || z_hi > slab_layers_glob(c) || z_hi - z_lo < 2) return fail(h, SPH_ERR_INVALID, "comm_set_slab: a slab needs >= 2 cell layers inside the grid");
    s.has_down = h->comm.rank > 0; s.has_up = h->comm.rank < h->comm.nranks - 1;
    bool first = false;
    if (!h->comm.slab_ready) {
        first = true;
        const size_t cap = (size_t)s.halo_cap;
        for (int k = 0; k < 2; ++k) {
            int rc = dalloc(h, &s.xidx[k], (size_t)s.cap); if (rc) return rc;
            rc = dalloc(h, &s.sendbuf[k], 4 * cap); if (rc) return rc;
            rc = dalloc(h, &s.recvbuf[k], 4 * cap); if (rc) return rc;
        }
        for (int k = 0; k < 8; ++k) { int rc = dalloc(h, &s.halo_tab[k], cap); if (rc) return rc; }
        { int rc = dalloc(h, &s.halo_counts, 2 * HC_BANK); if (rc) return rc; }
        if (h->slab_axis == 0 && !getenv("SPH_NO_SLAB_OVERLAP")) {   // boundary / interior tile lists (State::tile_list): slowest-axis slabs only
            for (int k = 0; k < 2; ++k) { int rc = dalloc(h, &s.tile_list[k], ((size_t)s.cap + 255) / 256 + 1); if (rc) return rc; }
            int rc = dalloc(h, &s.tile_cnt, 2); if (rc) return rc;
            rc = dalloc(h, &s.tile_class, ((size_t)s.cap + 255) / 256 + 1); if (rc) return rc;
        }
        { int rc = dalloc(h, &s.dyn, 2); if (rc) return rc; }
        s.dyn_cur = s.dyn;
        HIPCHK(h, hipMalloc((void **)&h->comm.bad_dev, sizeof(int)));
        HIPCHK(h, hipHostMalloc((void **)&h->comm.n_stage, 16 * sizeof(int), hipHostMallocDefault));
        s.xcur = 0;
        HIPCHK(h, hipMalloc((void **)&h->comm.hist_dev, sizeof(int) * (size_t)(slab_layers_glob(c) + h->comm.nranks)));
        HIPCHK(h, hipHostMalloc((void **)&h->comm.hist_host, sizeof(int) * (size_t)(slab_layers_glob(c) + h->comm.nranks), hipHostMallocDefault));
        const char *rb = getenv("SPH_SLAB_REBALANCE");
        h->comm.rebalance_every = rb ? atoi(rb) : 64;
        h->comm.slab_ready = 1;
    }
    h->L->ensure_color(s); s.color_home_ok = 0;   // sharded: global ids, the colours travel in the halo records
    s.slab_active = 1;
    s.z_lo = z_lo; s.z_hi = z_hi;
    // the cell lists cover the own layers plus one ghost layer per interior side only: G, and with it the histogram, the
    // scan and the cell windows, shrink from the global grid to the slab (weak scaling would otherwise scan N times as
    // many cells on every rank)
    slab_set_window(c, z_lo, z_hi);
    if (first && h->comm.push_wanted) { int rc = push_setup(h); if (rc) return rc; }
    return SPH_OK;
}

extern "C" int sph_comm_get_slab(SphHandle *h, int *z_lo, int *z_hi, int *n_owned, int *n_ghost) {
    if (!h || !h->comm.kind) return fail(h, SPH_ERR_INVALID, "comm_get_slab: communicator not initialised");
    HIPCHK(h, hipSetDevice(h->device));
    if (z_lo) *z_lo = h->st.z_lo;
    if (z_hi) *z_hi = h->st.z_hi;
    if (n_owned || n_ghost) {
        { int rc = slab_settle(h); if (rc) return rc; }
        refresh_counts(h);
        h->L->count_ghosts(h->st, h->comm.cnt_dev + 3);   // 4 bytes come back, not the whole meta array
        int g = 0;
        HIPCHK(h, hipMemcpyAsync(&g, h->comm.cnt_dev + 3, sizeof(int), hipMemcpyDeviceToHost, h->st.stream));
        HIPCHK(h, hipStreamSynchronize(h->st.stream));
        if (n_ghost) *n_ghost = g;
        if (n_owned) *n_owned = h->n - g;
    }
    return SPH_OK;
}

// ---- transport: one message to / from each neighbour.  bytes_recv[side] is filled in when sizes_known == false.
static int shm_wait(std::atomic<uint64_t> &a, uint64_t want, SphHandle *h, const char *what) {
    struct timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned spins = 0; a.load(std::memory_order_acquire) < want; ++spins) {
        if ((spins & 1023) == 1023) {
            struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((double)(t1.tv_sec - t0.tv_sec) > h->comm.timeout_s) return fail(h, SPH_ERR_COMM, "shm transport: timed out waiting for %s", what);
            usleep(50);
        }
    }
    return SPH_OK;
}

// ---- host-visible collectives of the communicator (bench launcher, solver residuals under sharding)
static int shm_barrier(SphHandle *h, SlabComm &c) {
    ShmHeader *hd = (ShmHeader *)c.shm_base;
    const uint64_t gen = hd->bar_gen.load(std::memory_order_acquire);
    if (hd->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint64_t)c.nranks) {
        hd->bar_count.store(0, std::memory_order_relaxed);
        hd->bar_gen.fetch_add(1, std::memory_order_release);
        return SPH_OK;
    }
    struct timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned spins = 0; hd->bar_gen.load(std::memory_order_acquire) == gen; ++spins) {
        if ((spins & 255) == 255) {
            struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((double)(t1.tv_sec - t0.tv_sec) > 2.0 * c.timeout_s) return fail(h, SPH_ERR_COMM, "shm transport: barrier timed out");
            usleep(20);
        }
    }
    return SPH_OK;
}

// op: 0 sum, 1 max, 2 min over all ranks, in place; count <= SHM_RED_MAX doubles.  Synchronous.
extern "C" int sph_comm_allreduce(SphHandle *h, double *inout, int count, int op) {
    if (!h || !inout || count < 0 || count > SHM_RED_MAX || op < 0 || op > 2) return fail(h, SPH_ERR_INVALID, "comm_allreduce: bad arguments");
    SlabComm &c = h->comm;
    if (!c.kind) return fail(h, SPH_ERR_INVALID, "comm_allreduce: communicator not initialised");
    if (count == 0) return SPH_OK;
    HIPCHK(h, hipSetDevice(h->device));
    if (c.kind == 2) {
        ShmHeader *hd = (ShmHeader *)c.shm_base;
        for (int k = 0; k < count; ++k) hd->red[c.rank][k] = inout[k];
        int rc = shm_barrier(h, c); if (rc) return rc;
        for (int k = 0; k < count; ++k) {
            double v = hd->red[0][k];
            for (int r = 1; r < c.nranks; ++r) { const double w = hd->red[r][k]; v = op == 0 ? v + w : (op == 1 ? (w > v ? w : v) : (w < v ? w : v)); }
            inout[k] = v;
        }
        return shm_barrier(h, c);   // nobody overwrites its row before everyone has read it
    }
    State &s = h->st;
    memcpy(c.red_host, inout, sizeof(double) * count);
    HIPCHK(h, hipMemcpyAsync(c.red_dev, c.red_host, sizeof(double) * count, hipMemcpyHostToDevice, s.stream));
    NCCLCHK(h, ncclAllReduce(c.red_dev, c.red_dev, (size_t)count, ncclDouble, op == 0 ? ncclSum : (op == 1 ? ncclMax : ncclMin), (ncclComm_t)c.nccl, s.stream));
    HIPCHK(h, hipMemcpyAsync(c.red_host, c.red_dev, sizeof(double) * count, hipMemcpyDeviceToHost, s.stream));
    { int rcs = stream_sync_bounded(h, "all-reduce over RCCL"); if (rcs) return rcs; }
    memcpy(inout, c.red_host, sizeof(double) * count);
    return SPH_OK;
}

// all ranks' streams drained, then everybody leaves together
extern "C" int sph_comm_barrier(SphHandle *h) {
    if (!h || !h->comm.kind) return fail(h, SPH_ERR_INVALID, "comm_barrier: communicator not initialised");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->st.stream));
    if (h->comm.kind == 2) return shm_barrier(h, h->comm);
    double one = 1.0;
    return sph_comm_allreduce(h, &one, 1, 0);
}

// Transport self-test: every rank sends n floats of a rank-tagged pattern to rank + 1 (mod size) and receives from
// rank - 1 (a self send/recv pair when size == 1), checks every word, then checks a sum all-reduce.  RCCL: the same
// ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd sequence the halo exchange uses.
extern "C" int sph_comm_selftest(SphHandle *h, int n) {
    if (!h || !h->comm.kind || n < 1 || n > (1 << 24)) return fail(h, SPH_ERR_INVALID, "comm_selftest: bad arguments");
    SlabComm &c = h->comm;
    State &s = h->st;
    HIPCHK(h, hipSetDevice(h->device));
    const int to = (c.rank + 1) % c.nranks, from = (c.rank + c.nranks - 1) % c.nranks;
    std::vector<float> src((size_t)n), dst((size_t)n, -1.0f);
    for (int k = 0; k < n; ++k) src[k] = (float)(c.rank * 1000 + (k % 997));
    if (c.kind == 1) {
        if (c.self_dev) { hipFree(c.self_dev); c.self_dev = nullptr; }
        HIPCHK(h, hipMalloc((void **)&c.self_dev, sizeof(float) * 2 * (size_t)n));
        HIPCHK(h, hipMemcpyAsync(c.self_dev, src.data(), sizeof(float) * n, hipMemcpyHostToDevice, s.stream));
        HIPCHK(h, hipMemsetAsync(c.self_dev + n, 0xff, sizeof(float) * n, s.stream));
        NCCLCHK(h, ncclGroupStart());
        NCCLCHK(h, ncclSend(c.self_dev, (size_t)n, ncclFloat, to, (ncclComm_t)c.nccl, s.stream));
        NCCLCHK(h, ncclRecv(c.self_dev + n, (size_t)n, ncclFloat, from, (ncclComm_t)c.nccl, s.stream));
        NCCLCHK(h, ncclGroupEnd());
        HIPCHK(h, hipMemcpyAsync(dst.data(), c.self_dev + n, sizeof(float) * n, hipMemcpyDeviceToHost, s.stream));
        { int rcs = stream_sync_bounded(h, "RCCL self-test"); if (rcs) return rcs; }
    } else {
        // shm: the mailbox pair of the "up" direction, wrapped around (rank size-1 -> rank 0 uses its unused up box)
        if ((size_t)n * 4 > c.mbox_cap) return fail(h, SPH_ERR_CAPACITY, "comm_selftest: %d floats exceed the mailbox", n);
        ShmMailbox *out = shm_mbox(c, c.rank, 1);
        memcpy((char *)(out + 1), src.data(), sizeof(float) * n);
        out->nbytes = (uint64_t)n * 4;
        int rc = shm_barrier(h, c); if (rc) return rc;
        ShmMailbox *in = shm_mbox(c, from, 1);
        if (in->nbytes != (uint64_t)n * 4) return fail(h, SPH_ERR_COMM, "comm_selftest: size mismatch");
        memcpy(dst.data(), (char *)(in + 1), sizeof(float) * n);
        rc = shm_barrier(h, c); if (rc) return rc;
    }
    for (int k = 0; k < n; ++k)
        if (dst[k] != (float)(from * 1000 + (k % 997))) return fail(h, SPH_ERR_COMM, "comm_selftest: word %d from rank %d is %g", k, from, (double)dst[k]);
    if (s.push.on && c.nranks > 1) {   // the push transport through its own kernels: pattern into both neighbours' inboxes, wait, check
        const int m = std::min(n, (int)(s.push.fld_bytes / sizeof(float)));
        HIPCHK(h, hipMemsetAsync(c.bad_dev, 0, sizeof(int), s.stream));
        h->L->halo_selftest(s, m, c.rank, c.rank - 1, c.rank + 1, c.bad_dev);
        int bad = -1, st = 0;
        HIPCHK(h, hipMemcpyAsync(&bad, c.bad_dev, sizeof(int), hipMemcpyDeviceToHost, s.stream));
        HIPCHK(h, hipMemcpyAsync(&st, &s.dyn_cur->status, sizeof(int), hipMemcpyDeviceToHost, s.stream));
        int rc = stream_sync_bounded(h, "comm_selftest (push transport)"); if (rc) return rc;
        if (bad || st) return fail(h, SPH_ERR_COMM, "comm_selftest: push transport delivered %d wrong words (status %d)", bad, st);
    }
    double v[2] = {(double)(c.rank + 1), 1.0};
    int rc = sph_comm_allreduce(h, v, 2, 0); if (rc) return rc;
    if (v[0] != 0.5 * c.nranks * (c.nranks + 1) || v[1] != (double)c.nranks) return fail(h, SPH_ERR_COMM, "comm_selftest: all-reduce gave %g, %g", v[0], v[1]);
    return SPH_OK;
}

static int comm_exchange(SphHandle *h, const void *send[2], const size_t bytes_send[2], void *recv[2], size_t bytes_recv[2],
                         bool sizes_known) {
    SlabComm &c = h->comm;
    State &s = h->st;
    const int peer[2] = {c.rank - 1, c.rank + 1};
    const bool has[2] = {s.has_down != 0, s.has_up != 0};
    if (c.kind == 2) {
        HIPCHK(h, hipStreamSynchronize(s.stream));
        const uint64_t t = ++c.seq;
        for (int side = 0; side < 2; ++side) {
            if (!has[side]) continue;
            ShmMailbox *mb = shm_mbox(c, c.rank, side);
            int rc = shm_wait(mb->seq_read, t - 1, h, "the previous message to be consumed"); if (rc) return rc;
            if (bytes_send[side] > c.mbox_cap) return fail(h, SPH_ERR_CAPACITY, "halo message of %zu bytes exceeds the mailbox", bytes_send[side]);
            if (bytes_send[side]) HIPCHK(h, hipMemcpy((char *)(mb + 1), send[side], bytes_send[side], hipMemcpyDeviceToHost));
            mb->nbytes = bytes_send[side];
            mb->seq_written.store(t, std::memory_order_release);
        }
        for (int side = 0; side < 2; ++side) {
            if (!has[side]) { bytes_recv[side] = 0; continue; }
            ShmMailbox *mb = shm_mbox(c, peer[side], 1 - side);  // the neighbour's message in my direction
            int rc = shm_wait(mb->seq_written, t, h, "a neighbour's message"); if (rc) return rc;
            const size_t nb = (size_t)mb->nbytes;
            if (sizes_known && nb != bytes_recv[side]) return fail(h, SPH_ERR_COMM, "halo message size mismatch (%zu vs %zu)", nb, bytes_recv[side]);
            bytes_recv[side] = nb;
            if (nb) HIPCHK(h, hipMemcpy(recv[side], (char *)(mb + 1), nb, hipMemcpyHostToDevice));
            mb->seq_read.store(t, std::memory_order_release);
        }
        return SPH_OK;
    }
    ncclComm_t comm = (ncclComm_t)c.nccl;
    if (!sizes_known) {  // exchange the two message sizes first (device ints; ncclSend/Recv move device memory only)
        c.cnt_host[0] = (int)bytes_send[0]; c.cnt_host[1] = (int)bytes_send[1]; c.cnt_host[2] = c.cnt_host[3] = 0;
        HIPCHK(h, hipMemcpyAsync(c.cnt_dev, c.cnt_host, 4 * sizeof(int), hipMemcpyHostToDevice, s.stream));
        NCCLCHK(h, ncclGroupStart());
        for (int side = 0; side < 2; ++side) {
            if (!has[side]) continue;
            NCCLCHK(h, ncclSend(c.cnt_dev + side, 1, ncclInt32, peer[side], comm, s.stream));
            NCCLCHK(h, ncclRecv(c.cnt_dev + 2 + side, 1, ncclInt32, peer[side], comm, s.stream));
        }
        NCCLCHK(h, ncclGroupEnd());
        HIPCHK(h, hipMemcpyAsync(c.cnt_host + 4, c.cnt_dev, 4 * sizeof(int), hipMemcpyDeviceToHost, s.stream));
        { int rcs = stream_sync_bounded(h, "halo count exchange over RCCL"); if (rcs) return rcs; }
        bytes_recv[0] = has[0] ? (size_t)c.cnt_host[6] : 0;
        bytes_recv[1] = has[1] ? (size_t)c.cnt_host[7] : 0;
    }
    NCCLCHK(h, ncclGroupStart());
    for (int side = 0; side < 2; ++side) {
        if (!has[side]) continue;
        if (bytes_send[side]) NCCLCHK(h, ncclSend(send[side], bytes_send[side], ncclChar, peer[side], comm, s.stream));
        if (bytes_recv[side]) NCCLCHK(h, ncclRecv(recv[side], bytes_recv[side], ncclChar, peer[side], comm, s.stream));
    }
    NCCLCHK(h, ncclGroupEnd());
    return SPH_OK;
}

// ---------------------------------------------------------------------------------------------- push transport set-up
// Every rank allocates its inbox, hands an IPC handle of it to both neighbours (one small message each through the control
// plane's neighbour transport), maps theirs, and the ranks then run a self-test through the very kernels the halo exchange
// uses (payload stores into the neighbour's inbox, system-scope fence, message number, bounded wait, check).  Whether the
// push transport is used is agreed by all ranks (all-reduce of the per-rank verdicts): all or none.
static int push_teardown(SphHandle *h) {
    State &s = h->st; SlabComm &c = h->comm;
    for (int side = 0; side < 2; ++side) { if (c.ipc_mapped[side]) hipIpcCloseMemHandle(c.ipc_mapped[side]); c.ipc_mapped[side] = nullptr; s.push.peer[side] = nullptr; }
    if (c.inbox_alloc) { hipFree(c.inbox_alloc); c.inbox_alloc = nullptr; }
    if (s.push.mirror) { hipHostFree(s.push.mirror); s.push.mirror = nullptr; }
    s.push.on = 0; s.push.inbox = nullptr;
    (void)hipGetLastError();
    return SPH_OK;
}

static int push_setup(SphHandle *h) {
    State &s = h->st; SlabComm &c = h->comm;
    memset(&s.push, 0, sizeof(s.push));
    s.push.timeout_ticks = (long long)(0.5 * c.timeout_s * 1.0e8);   // 100 MHz wall clock; the device gives up before the host does
    int ok = 1;
    char why[160] = "";
    void *inbox = nullptr;
    size_t bytes = 0;
    // Memory the other device writes and this one polls: uncached / fine-grained where the runtime offers it (what RCCL uses for its
    // own flags), plain device memory otherwise (enough between two ranks of ONE device).  Message capacity = particle capacity (no
    // pile-up in the boundary layers can overflow it); if the runtime will not give that much coherent memory, a quarter of it (a
    // longer message is then refused by both sides through the status word, it does not corrupt anything).
    int inbox_kind = 2;   // 0 uncached, 1 fine-grained, 2 plain
    hipError_t e = hipErrorOutOfMemory;
    const int caps[2] = {s.halo_cap, std::max(s.halo_cap / 4, std::min(s.halo_cap, 262144))};
    auto alloc_inbox = [&](int size_class) {   // coherent kinds first; returns hipSuccess with inbox / inbox_kind / bytes set
        s.push.rec_cap = caps[size_class];
        s.push.rec_bytes = (size_t)s.push.rec_cap * 64;        // records of 64 B (with the rest position of a dynamic rigid body)
        s.push.fld_bytes = (size_t)s.push.rec_cap * 2 * 16;    // n_send + n_recv <= 2 rec_cap records of <= 16 B
        bytes = 2 * sizeof(HaloCtl) + 4 * s.push.rec_bytes + 4 * s.push.fld_bytes;
        hipError_t r = hipErrorOutOfMemory;
        for (int kind = 0; kind < 3 && r != hipSuccess; ++kind) {
            // between devices only coherent memory will do (checked against the neighbours' bus ids below); do not grab the plain
            // kind at full size when a coherent quarter might still be had
            if (kind == 2 && size_class == 0 && caps[1] != caps[0] && c.nranks > 1) break;
            inbox_kind = kind;
            r = kind == 2 ? hipMalloc(&inbox, bytes) : hipExtMallocWithFlags(&inbox, bytes, kind == 0 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
            if (r != hipSuccess) { (void)hipGetLastError(); inbox = nullptr; }
        }
        return r;
    };
    int size_class = 0;
    if (const char *fr = getenv("SPH_COMM_TEST_FAIL_PUSH_RANK")) {   // test hook: this rank cannot set the push transport up
        if (atoi(fr) == c.rank) { ok = 0; snprintf(why, sizeof(why), "SPH_COMM_TEST_FAIL_PUSH_RANK (test hook)"); }
    }
    e = alloc_inbox(0);
    if (e != hipSuccess && caps[1] != caps[0]) { size_class = 1; e = alloc_inbox(1); }
    // the layout of an inbox depends on the message capacity: every rank uses the same one (the smallest anybody got)
    { double cls[1] = {(double)size_class}; int rc = sph_comm_allreduce(h, cls, 1, 1); if (rc) return rc;
      if ((int)cls[0] != size_class) { if (inbox) hipFree(inbox); inbox = nullptr; size_class = (int)cls[0]; e = alloc_inbox(size_class); } }
    char busid[32] = "";
    if (hipDeviceGetPCIBusId(busid, sizeof(busid), h->device) != hipSuccess) { (void)hipGetLastError(); snprintf(busid, sizeof(busid), "dev%d-pid%d", h->device, (int)getpid()); }
    if (e != hipSuccess) { ok = 0; snprintf(why, sizeof(why), "inbox allocation of %zu bytes: %s", bytes, hipGetErrorString(e)); (void)hipGetLastError(); }
    c.inbox_alloc = inbox;   // owned from here on, whatever `ok` says (push_teardown / comm_free release it)
    hipIpcMemHandle_t mine;
    memset(&mine, 0, sizeof(mine));
    if (ok) {
        s.push.inbox = (char *)inbox;
        e = hipMemset(inbox, 0, 2 * sizeof(HaloCtl));
        if (e == hipSuccess && c.nranks > 1) e = hipIpcGetMemHandle(&mine, inbox);
        if (e != hipSuccess) { ok = 0; snprintf(why, sizeof(why), "hipIpcGetMemHandle: %s", hipGetErrorString(e)); (void)hipGetLastError(); }
    }
    if (hipHostMalloc((void **)&s.push.mirror, sizeof(SlabDyn), hipHostMallocDefault) != hipSuccess) { ok = 0; snprintf(why, sizeof(why), "pinned mirror"); (void)hipGetLastError(); }
    else { memset(s.push.mirror, 0, sizeof(SlabDyn)); s.push.mirror->n_btiles = -1; }
    // handles to the neighbours (even a rank that failed so far takes part: the exchange is collective)
    struct Hello { unsigned magic; int ok; int rank; int inbox_kind; char busid[32]; hipIpcMemHandle_t handle; } hello_out, hello_in[2];
    static_assert(sizeof(Hello) <= 256, "hello message");
    memset(&hello_out, 0, sizeof(hello_out));
    hello_out.magic = 0x53504831u; hello_out.ok = ok; hello_out.rank = c.rank; hello_out.inbox_kind = inbox_kind; hello_out.handle = mine;
    memcpy(hello_out.busid, busid, sizeof(hello_out.busid));
    memset(hello_in, 0, sizeof(hello_in));
    if (c.nranks > 1) {
        for (int side = 0; side < 2; ++side) HIPCHK(h, hipMemcpy(s.sendbuf[side], &hello_out, sizeof(Hello), hipMemcpyHostToDevice));
        const void *send[2] = {s.sendbuf[0], s.sendbuf[1]};
        void *recv[2] = {s.recvbuf[0], s.recvbuf[1]};
        const size_t bs[2] = {sizeof(Hello), sizeof(Hello)};
        size_t br[2] = {sizeof(Hello), sizeof(Hello)};
        int rc = comm_exchange(h, send, bs, recv, br, true); if (rc) return rc;
        rc = stream_sync_bounded(h, "push transport hello exchange"); if (rc) return rc;   // a dead neighbour must not hang the setup
        const int peer[2] = {c.rank - 1, c.rank + 1};
        for (int side = 0; side < 2; ++side) {
            if (!(side == 0 ? s.has_down : s.has_up)) continue;
            HIPCHK(h, hipMemcpy(&hello_in[side], s.recvbuf[side], sizeof(Hello), hipMemcpyDeviceToHost));
            if (hello_in[side].magic != 0x53504831u || hello_in[side].rank != peer[side]) { ok = 0; snprintf(why, sizeof(why), "bad hello from rank %d", peer[side]); continue; }
            if (!hello_in[side].ok || !ok) { ok = 0; continue; }
            // plain (cached) device memory is only coherent between two ranks that share ONE device; across devices the
            // inbox must be uncached / fine-grained on both ends, or remotely written lines may be read stale out of the L2
            const bool same_device = !memcmp(hello_in[side].busid, busid, sizeof(busid));
            if (!same_device && (inbox_kind == 2 || hello_in[side].inbox_kind == 2)) {
                ok = 0; snprintf(why, sizeof(why), "no uncached / fine-grained device memory for an inbox shared with rank %d on another GPU", peer[side]); continue;
            }
            void *mapped = nullptr;
            e = hipIpcOpenMemHandle(&mapped, hello_in[side].handle, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) { ok = 0; snprintf(why, sizeof(why), "hipIpcOpenMemHandle(rank %d): %s", peer[side], hipGetErrorString(e)); (void)hipGetLastError(); continue; }
            c.ipc_mapped[side] = mapped; s.push.peer[side] = (char *)mapped;
        }
    }
    // self-test through the real kernels; a rank that is not ok so far still answers its neighbours' waits as far as it can
    double verdict[1] = {(double)ok};
    { int rc = sph_comm_allreduce(h, verdict, 1, 2); if (rc) return rc; }   // min over ranks
    if (verdict[0] >= 1.0 && c.nranks > 1) {
        s.push.on = 1;
        const int n = 1 << 16;
        HIPCHK(h, hipMemsetAsync(c.bad_dev, 0, sizeof(int), s.stream));
        h->L->halo_selftest(s, n, c.rank, c.rank - 1, c.rank + 1, c.bad_dev);
        int bad = -1, st = 0;
        hipError_t e2 = hipMemcpyAsync(&bad, c.bad_dev, sizeof(int), hipMemcpyDeviceToHost, s.stream);
        if (e2 == hipSuccess) e2 = hipMemcpyAsync(&st, &s.dyn_cur->status, sizeof(int), hipMemcpyDeviceToHost, s.stream);
        int rc = e2 == hipSuccess ? stream_sync_bounded(h, "push transport self-test") : SPH_ERR_HIP;
        if (rc || bad != 0 || st != 0) { ok = 0; snprintf(why, sizeof(why), "self-test: %d wrong words, status %d", bad, st); }
        HIPCHK(h, hipMemsetAsync(s.dyn, 0, 2 * sizeof(SlabDyn), s.stream));   // (the status word is sticky: a failed self-test must not outlive the fall-back)
        verdict[0] = (double)ok;
        rc = sph_comm_allreduce(h, verdict, 1, 2); if (rc) return rc;
    }
    if (verdict[0] < 1.0) {
        push_teardown(h);
        if (c.push_wanted == 2) return fail(h, SPH_ERR_COMM, "push transport (SPH_COMM_TRANSPORT=ipc) cannot be set up%s%s", why[0] ? ": " : " (another rank failed)", why);
        fprintf(stderr, "[libsph_hip] rank %d: push transport not available%s%s -- halo exchange through %s\n", c.rank, why[0] ? ": " : " on another rank", why, c.transport);
        return SPH_OK;
    }
    s.push.on = 1;
    const char *as = getenv("SPH_SLAB_ASYNC");
    c.async_enabled = !(as && atoi(as) == 0);
    char base[24];
    snprintf(base, sizeof(base), "%s", c.transport);
    snprintf(c.transport, sizeof(c.transport), "ipc-push+%s", base);
    if (getenv("SPH_COMM_VERBOSE"))
        fprintf(stderr, "[libsph_hip] rank %d on %s: push transport up (inbox %.1f MB, %s device memory)\n", c.rank, busid, bytes / 1e6,
                inbox_kind == 0 ? "uncached" : (inbox_kind == 1 ? "fine-grained" : "plain"));
    return SPH_OK;
}

// Asynchronous steps leave the host with launch bounds instead of counts.  Before anything on the host needs the exact
// particle count (download, append, statistics, a solver that launches exact grids) the stream is drained -- bounded -- and
// the counts of the last step come back from the pinned mirror, together with the sticky status word.
static int slab_settle(SphHandle *h) {
    State &s = h->st; SlabComm &c = h->comm;
    if (h->n_exact) return SPH_OK;
    int rc = stream_sync_bounded(h, "waiting for the asynchronous slab steps"); if (rc) return rc;
    const SlabDyn m = *(const SlabDyn *)s.push.mirror;
    h->n = m.n_live; h->n_exact = true;
    for (int side = 0; side < 2; ++side) { c.n_send[side] = m.n_send[side]; c.n_recv[side] = m.n_recv[side]; }
    c.est_recv = m.n_recv[0] + m.n_recv[1];
    s.halo_longest = m.longest;
    s.async_counts = 0; s.c.n_dev = nullptr;
    s.perm_n = s.list_n = -1;   // headers / lists were built for the launch bound
    refresh_counts(h);
    if (m.status) return fail(h, SPH_ERR_COMM, "sharded step failed (status %d): %s", m.status, slab_status_text(m.status));
    return SPH_OK;
}

// Slab cuts that follow the fluid (SURVEY 8e "slab cuts from a per-z-layer histogram"): every rebalance_every steps the
// ranks add up their owned-particle histograms over the global z layers, re-plan balanced cuts exactly like
// sph_project_amd/slab.py:plan_slabs does for the initial lattice, and move every interior cut by AT MOST ONE layer
// towards its target.  One layer is what the per-step protocol already handles: the particles of the layer that changed
// hands are migrants of the next classify (kept behind as echo ghosts), nothing else is needed.
static void plan_cuts(const std::vector<long long> &hist, int nranks, std::vector<int> &cuts) {
    const int nz = (int)hist.size(), min_layers = 2;
    std::vector<double> cum(nz + 1, 0.0);
    for (int k = 0; k < nz; ++k) cum[k + 1] = cum[k] + (double)hist[k];
    const double total = cum[nz];
    cuts.assign(1, 0);
    for (int r = 1; r < nranks; ++r) {
        const double target = total * r / nranks;
        int k = (int)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());
        if (k > 0 && fabs(cum[k - 1] - target) <= fabs(cum[std::min(k, nz)] - target)) k -= 1;
        k = std::max(k, cuts.back() + min_layers);
        k = std::min(k, nz - (nranks - r) * min_layers);
        cuts.push_back(k);
    }
    cuts.push_back(nz);
}

static int slab_rebalance(SphHandle *h) {
    State &s = h->st;
    SlabComm &c = h->comm;
    Consts &k = s.c;
    if (c.nranks < 2) return SPH_OK;
    const int nz = slab_layers_glob(k), len = nz + c.nranks;   // layers along the slab axis
    ProfScope p(h, SPH_K_HALO);
    h->L->layer_hist(s, c.hist_dev);
    HIPCHK(h, hipMemsetAsync(c.hist_dev + nz, 0, sizeof(int) * c.nranks, s.stream));
    HIPCHK(h, hipMemcpyAsync(c.hist_dev + nz + c.rank, &s.z_lo, sizeof(int), hipMemcpyHostToDevice, s.stream));
    if (c.kind == 1) {
        NCCLCHK(h, ncclAllReduce(c.hist_dev, c.hist_dev, (size_t)len, ncclInt32, ncclSum, (ncclComm_t)c.nccl, s.stream));
        HIPCHK(h, hipMemcpyAsync(c.hist_host, c.hist_dev, sizeof(int) * len, hipMemcpyDeviceToHost, s.stream));
        { int rcs = stream_sync_bounded(h, "histogram all-reduce over RCCL"); if (rcs) return rcs; }
    } else {
        HIPCHK(h, hipMemcpyAsync(c.hist_host, c.hist_dev, sizeof(int) * len, hipMemcpyDeviceToHost, s.stream));
        HIPCHK(h, hipStreamSynchronize(s.stream));
        for (int o = 0; o < len; o += SHM_RED_MAX) {
            double v[SHM_RED_MAX];
            const int m = std::min(SHM_RED_MAX, len - o);
            for (int q = 0; q < m; ++q) v[q] = (double)c.hist_host[o + q];
            int rc = sph_comm_allreduce(h, v, m, 0); if (rc) return rc;
            for (int q = 0; q < m; ++q) c.hist_host[o + q] = (int)v[q];
        }
    }
    std::vector<long long> hist(nz);
    for (int q = 0; q < nz; ++q) hist[q] = c.hist_host[q];
    std::vector<int> target, cur(c.nranks + 1);
    plan_cuts(hist, c.nranks, target);
    for (int r = 0; r < c.nranks; ++r) cur[r] = c.hist_host[nz + r];
    cur[c.nranks] = nz;
    std::vector<int> next(cur);
    for (int r = 1; r < c.nranks; ++r) {
        const int d = target[r] - cur[r];
        next[r] = cur[r] + (d > 0 ? 1 : (d < 0 ? -1 : 0));
        next[r] = std::max(next[r], next[r - 1] + 2);                       // >= 2 layers below ...
        next[r] = std::min(next[r], nz - (c.nranks - r) * 2);               // ... and room above
        if (next[r] > cur[r] + 1 || next[r] < cur[r] - 1) next[r] = cur[r]; // never more than one layer per event
    }
    const int z_lo = next[c.rank], z_hi = next[c.rank + 1];
    c.rebalance_moves += (z_lo != s.z_lo) + (z_hi != s.z_hi);
    s.z_lo = z_lo; s.z_hi = z_hi;
    slab_set_window(k, z_lo, z_hi);
    return SPH_OK;
}

// An object entered late (entryTime > 0): `n` particles in the whole scene, of which this rank appended its slab's share.
// Keeps particle_num of the scene (the denominator of DFSPH's residual means, DFSPH.py:212 / :293) right on every rank.
extern "C" int sph_comm_add_global_count(SphHandle *h, int n, int n_fluid) {
    if (!h || !h->comm.kind || n < 0 || n_fluid < 0 || n_fluid > n) return fail(h, SPH_ERR_INVALID, "comm_add_global_count: bad arguments");
    if (h->prepared) { h->comm_n_global += n; h->comm_nfluid_global += n_fluid; }   // before prepare() the counts are all-reduced there
    return SPH_OK;
}

extern "C" int sph_comm_set_rebalance(SphHandle *h, int every_steps) {
    if (!h || !h->comm.kind || every_steps < 0) return fail(h, SPH_ERR_INVALID, "comm_set_rebalance: bad arguments");
    h->comm.rebalance_every = every_steps;
    return SPH_OK;
}

// ---- step message over the push transport (sph_halo.hpp).  Two flavours:
//   exact : the counts come back to the host once per step (one bounded wait on the stream), every launch is sized exactly --
//           what the iterative solvers use (their loops read flags back anyway);
//   async : WCSPH.  Nothing comes back.  The halo kernels keep the counts in device memory (SlabDyn), every kernel of the step
//           reads the particle count from there, and the host sizes its launches from BOUNDS: the last counts it has seen in
//           the pinned mirror (written by the wait kernel, a step or a few behind) plus a margin.  The wait kernel checks the
//           bounds against the real counts (SLAB_ST_BOUND, sticky, reported at the next settle); the host never runs more than
//           SLAB_MAX_LAG steps ahead of the mirror (back-pressure, not a drain).
#define SLAB_MAX_LAG 3
static int slab_neighbor_search_push(SphHandle *h, bool async, bool cut_moved = false) {
    State &s = h->st;
    SlabComm &c = h->comm;
    if (!async) {
        const int est_before = c.est_recv;
        int rc = slab_settle(h); if (rc) return rc;
        s.async_counts = 0; s.c.n_dev = nullptr;
        { ProfScope p(h, SPH_K_HALO);
          if (s.preclassified) { s.preclassified = 0; h->L->hash_count(s); } else h->L->halo_classify_pack(s, h->n);
          h->L->halo_unpack2(s, h->n, 0, 0, c.est_recv + c.est_recv / 4 + 4096); }
        int dev_status = 0;   // every workgroup ORs its verdict into the device word; the mirror carries workgroup 0's only
        HIPCHK(h, hipMemcpyAsync(&dev_status, &s.dyn_cur->status, sizeof(int), hipMemcpyDeviceToHost, s.stream));
        rc = stream_sync_bounded(h, "halo exchange (step message)"); if (rc) return rc;
        SlabDyn m = *(const SlabDyn *)s.push.mirror;
        m.status |= dev_status;
        if (m.status) return fail(h, SPH_ERR_COMM, "halo exchange failed (status %d): %s", m.status, slab_status_text(m.status));
        for (int side = 0; side < 2; ++side) { c.n_send[side] = m.n_send[side]; c.n_recv[side] = m.n_recv[side]; }
        c.est_recv = m.n_recv[0] + m.n_recv[1];
        // the step in which a layer changes hands is not typical for the rank that GIVES it: the new owner has no boundary copies of that
        // layer to send yet (they arrive as migrants in this very message), so this rank receives next to nothing now -- and a full layer
        // of copies one step later.  The estimate of the steps before stands.
        if (cut_moved) c.est_recv = std::max(c.est_recv, est_before);
        h->n = m.n_app;
        refresh_counts(h);
        ph_sort_hashed(h);
        h->n = m.n_live;
        refresh_counts(h);
        s.halo_longest = m.longest;
        { ProfScope p(h, SPH_K_HALO); h->L->halo_build_tables(s); }
        return SPH_OK;
    }
    long long live_known, app_known;
    int lag = 0, grid_n;
    if (h->n_exact) {   // first asynchronous step after an exact count: hand it to the device
        c.n_stage[0] = h->n;
        HIPCHK(h, hipMemcpyAsync(&s.dyn_cur->n_live, c.n_stage, sizeof(int), hipMemcpyHostToDevice, s.stream));   // (the bank the next message reads as "last step's")
        live_known = h->n; app_known = (long long)h->n + c.est_recv; grid_n = h->n;
        s.push.mirror->seq = s.push.rec_seq; s.push.mirror->n_live = h->n; s.push.mirror->n_app = (int)app_known; s.push.mirror->status = 0;
        s.push.mirror->wseq = 0;   // (the stream is idle here: slab_settle drained it)
    } else {
        volatile SlabDyn *mv = s.push.mirror;
        struct timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
        for (unsigned spins = 0;; ++spins) {
            const unsigned q0 = mv->wseq;   // bumped before AND after the fields are written: odd = being written
            __sync_synchronize();
            live_known = mv->n_live; app_known = mv->n_app;
            const int st = mv->status;
            const unsigned sq = mv->seq;
            __sync_synchronize();
            if ((q0 & 1u) || mv->wseq != q0) continue;   // torn read: the settle kernel was writing
            if (st) return slab_settle(h);  // drains the stream and reports
            lag = (int)(s.push.rec_seq - sq);
            if (lag <= SLAB_MAX_LAG) break;
            if ((spins & 1023) == 1023) {
                struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
                if ((double)(t1.tv_sec - t0.tv_sec) > c.timeout_s) return fail(h, SPH_ERR_COMM, "asynchronous slab steps: the device fell %d steps behind for %.0f s", lag, c.timeout_s);
            }
        }
        grid_n = c.bound_live;
    }
    // margin: a sixteenth of the slab (>= 16 k particles), one more cell layer's worth for thin slabs (a layer of boundary copies that was
    // not there the step before: a cut moved, fluid reached a face), and a little per step the host runs ahead of the mirror
    const int layers = std::max(1, s.z_hi - s.z_lo);
    const long long margin = std::max<long long>(16384, live_known / 16) + (layers < 16 ? live_known / layers : 0) +
                             (long long)lag * std::max<long long>(4096, live_known / 64);
    const int bound_app = (int)std::min<long long>(s.cap, app_known + margin);
    const int bound_live = (int)std::min<long long>(s.cap, live_known + margin);
    h->n_exact = false;
    s.async_counts = 1;
    s.c.n = grid_n; s.c.n_dev = &s.dyn_cur->n_live;
    { ProfScope p(h, SPH_K_HALO);
      // (preclassified: the last step's force pass classified the particles and sent this message's records itself; the hash is left)
      if (s.preclassified) { s.preclassified = 0; h->L->hash_count(s); } else h->L->halo_classify_pack(s, grid_n);
      h->L->halo_unpack2(s, -1, bound_app, bound_live, c.est_recv + c.est_recv / 4 + 4096); }   // (moves dyn_cur to this message's bank)
    h->n = bound_app; refresh_counts(h); s.c.n_dev = &s.dyn_cur->n_app;
    ph_sort_hashed(h);
    h->n = bound_live; refresh_counts(h); s.c.n_dev = &s.dyn_cur->n_live;
    c.bound_live = bound_live;
    { ProfScope p(h, SPH_K_HALO); h->L->halo_build_tables(s); }
    return SPH_OK;
}

// replaces ph_neighbor_search in slab mode: migrate + ghost exchange, then the usual sort, then the slot tables
static int slab_neighbor_search(SphHandle *h, bool allow_async = false) {
    State &s = h->st;
    SlabComm &c = h->comm;
    // records: 48 B, or 64 B with the rest position (rigid_particle_original_positions) once the scene has a dynamic rigid body --
    // its particles take it along when they change owner.  EVERY rank must have been told about the body (sph_set_object):
    // the record size travels in the message header and a mismatch fails on both sides instead of mis-parsing the payload.
    const size_t rec = s.orig.cur() ? 64 : 48;
    bool cut_moved = false;
    if (c.rebalance_every > 0 && h->prepared && h->steps > 0 && h->steps % c.rebalance_every == 0 && !h->any_rigid_object) {
        int rc = slab_settle(h); if (rc) return rc;
        const int moves = c.rebalance_moves;
        rc = slab_rebalance(h); if (rc) return rc;
        cut_moved = c.rebalance_moves != moves;
    }
    // A cut that moved hands a whole cell layer over in this step's message: ~n / layers particles on top of the usual trickle, more than
    // the margin of an asynchronous launch bound as soon as a slab has fewer than 16 layers (C4 on 8 ranks: 12).  That one step runs with
    // exact launches (one read-back); est_recv then holds the layer, and the following asynchronous steps are sized from real counts.
    if (s.push.on) return slab_neighbor_search_push(h, allow_async && c.async_enabled != 0 && !cut_moved, cut_moved);
    { ProfScope p(h, SPH_K_HALO); h->L->halo_classify_pack(s, h->n); }
    const void *send[2] = {s.sendbuf[0], s.sendbuf[1]};
    void *recv[2] = {s.recvbuf[0], s.recvbuf[1]};
    int dropped = 0;
    if (c.kind == 1) {
        // RCCL: the record counts go to the neighbours straight from device memory, each followed by an info word (record
        // size | this rank's sticky status << 8); ONE read-back brings my own counts and theirs to the host (the payload calls
        // need both).  A failure either side knows about at this point (message over capacity, record sizes that differ, a
        // failure carried over) is seen by BOTH before any payload call is posted: both return, nobody is left in a receive.
        ProfScope p(h, SPH_K_HALO);
        ncclComm_t comm = (ncclComm_t)c.nccl;
        const int peer[2] = {c.rank - 1, c.rank + 1};
        const bool has[2] = {s.has_down != 0, s.has_up != 0};
        c.n_stage[4] = c.n_stage[5] = (int)rec | (c.sticky_status << 8);
        HIPCHK(h, hipMemcpyAsync(c.cnt_dev + 4, c.n_stage + 4, 2 * sizeof(int), hipMemcpyHostToDevice, s.stream));
        NCCLCHK(h, ncclGroupStart());
        for (int side = 0; side < 2; ++side) {
            if (!has[side]) continue;
            NCCLCHK(h, ncclSend(s.halo_counts + HC(side), 1, ncclInt32, peer[side], comm, s.stream));
            NCCLCHK(h, ncclSend(c.cnt_dev + 4 + side, 1, ncclInt32, peer[side], comm, s.stream));
            NCCLCHK(h, ncclRecv(c.cnt_dev + 2 + side, 1, ncclInt32, peer[side], comm, s.stream));
            NCCLCHK(h, ncclRecv(c.cnt_dev + 6 + side, 1, ncclInt32, peer[side], comm, s.stream));
        }
        NCCLCHK(h, ncclGroupEnd());
        for (int q = 0; q < 3; ++q) HIPCHK(h, hipMemcpyAsync(c.cnt_host + q, s.halo_counts + HC(q), sizeof(int), hipMemcpyDeviceToHost, s.stream));
        HIPCHK(h, hipMemcpyAsync(c.cnt_host + 4, c.cnt_dev, 8 * sizeof(int), hipMemcpyDeviceToHost, s.stream));
        { int rc = stream_sync_bounded(h, "halo exchange (message sizes)"); if (rc) return rc; }
        dropped = c.cnt_host[2];
        int st = c.sticky_status;
        for (int side = 0; side < 2; ++side) {
            c.n_send[side] = c.cnt_host[side];
            c.n_recv[side] = has[side] ? c.cnt_host[4 + 2 + side] : 0;
            if (c.n_send[side] > s.halo_cap) st |= SLAB_ST_SEND_OVERFLOW;
            if (!has[side]) continue;
            const int info = c.cnt_host[4 + 6 + side];
            if ((info & 0xff) != (int)rec) st |= SLAB_ST_STRIDE;
            if (info >> 8) st |= SLAB_ST_PEER;
            if (c.n_recv[side] < 0 || c.n_recv[side] > s.halo_cap) st |= SLAB_ST_PEER;   // the neighbour saw its own overflow too
        }
        if (st) { c.sticky_status |= st; return fail(h, SPH_ERR_CAPACITY, "halo exchange refused by both sides (status %d): %s", st, slab_status_text(st)); }
        const size_t bs[2] = {(size_t)c.n_send[0] * rec, (size_t)c.n_send[1] * rec};
        size_t br[2] = {(size_t)c.n_recv[0] * rec, (size_t)c.n_recv[1] * rec};
        int rc = comm_exchange(h, send, bs, recv, br, true); if (rc) return rc;
    } else {
        for (int q = 0; q < 3; ++q) HIPCHK(h, hipMemcpyAsync(c.cnt_host + q, s.halo_counts + HC(q), sizeof(int), hipMemcpyDeviceToHost, s.stream));
        HIPCHK(h, hipStreamSynchronize(s.stream));
        dropped = c.cnt_host[2];
        for (int side = 0; side < 2; ++side) {
            c.n_send[side] = c.cnt_host[side];
            if (c.n_send[side] > s.halo_cap) return fail(h, SPH_ERR_CAPACITY, "halo message of %d particles exceeds capacity %d", c.n_send[side], s.halo_cap);
        }
        const size_t bs[2] = {(size_t)c.n_send[0] * rec, (size_t)c.n_send[1] * rec};
        size_t br[2] = {0, 0};
        { ProfScope p(h, SPH_K_HALO); int rc = comm_exchange(h, send, bs, recv, br, false); if (rc) return rc; }
        if (br[0] % rec || br[1] % rec) return fail(h, SPH_ERR_COMM, "halo record sizes differ between neighbours (a dynamic rigid body must be registered with sph_set_object on EVERY rank)");
        c.n_recv[0] = (int)(br[0] / rec); c.n_recv[1] = (int)(br[1] / rec);
        if (c.n_recv[0] > s.halo_cap || c.n_recv[1] > s.halo_cap) return fail(h, SPH_ERR_CAPACITY, "received halo exceeds capacity");
    }
    // nothing was compacted: the arrivals are appended behind the old particles (dead ones included), the sort files the
    // dead ones into the graveyard cell behind all live particles, and only then does the particle count shrink
    const int n_old = h->n;
    if ((long long)n_old + c.n_recv[0] + c.n_recv[1] > s.cap) {
        // the payload has been exchanged, so no neighbour is stuck in this step; the status travels in the next header
        c.sticky_status |= SLAB_ST_CAPACITY;
        return fail(h, SPH_ERR_CAPACITY, "slab holds %d + %d + %d particles, particle_max_num is %d", n_old, c.n_recv[0], c.n_recv[1], s.cap);
    }
    { ProfScope p(h, SPH_K_HALO);
      h->L->halo_unpack_append(s, 0, c.n_recv[0], n_old);
      h->L->halo_unpack_append(s, 1, c.n_recv[1], n_old + c.n_recv[0]); }
    h->n = n_old + c.n_recv[0] + c.n_recv[1];
    refresh_counts(h);
    ph_neighbor_search(h);
    h->n -= dropped;
    refresh_counts(h);   // ghosts are told apart through the meta word (Consts::ghosts)
    s.halo_longest = std::max(std::max(c.n_send[0], c.n_send[1]), std::max(c.n_recv[0], c.n_recv[1]));
    { ProfScope p(h, SPH_K_HALO); h->L->halo_build_tables(s); }
    return SPH_OK;
}

// grid hint of the field-message kernels (grid-stride: a hint, not a bound)
static int slab_field_hint(SphHandle *h) {
    const SlabComm &c = h->comm;
    return h->n_exact ? c.n_send[0] + c.n_recv[0] + c.n_send[1] + c.n_recv[1] : 2 * c.est_recv + c.est_recv / 2 + 4096;
}

// field messages over the push transport: pack straight into the neighbours' inboxes (+ message number), wait, scatter
static int slab_exchange_push(SphHandle *h, int kind, float *f0, float4 *v) {
    State &s = h->st;
    ProfScope p(h, SPH_K_HALO);
    const int hint = slab_field_hint(h);
    h->L->halo_push_fields(s, kind, f0, v, hint);
    h->L->halo_pull_fields(s, kind, f0, v, hint);
    return SPH_OK;
}

// density / pressure of the ghosts after the density pass (SURVEY 8e message (3))
static int slab_exchange_fields(SphHandle *h) {
    State &s = h->st;
    SlabComm &c = h->comm;
    if (s.push.on) return slab_exchange_push(h, 2, nullptr, nullptr);
    ProfScope p(h, SPH_K_HALO);
    for (int side = 0; side < 2; ++side) h->L->halo_pack_fields(s, side, c.n_send[side], c.n_recv[side]);
    const void *send[2] = {s.sendbuf[0], s.sendbuf[1]};
    void *recv[2] = {s.recvbuf[0], s.recvbuf[1]};
    const size_t bs[2] = {(size_t)(c.n_send[0] + c.n_recv[0]) * 16, (size_t)(c.n_send[1] + c.n_recv[1]) * 16};
    size_t br[2] = {bs[0], bs[1]};
    int rc = comm_exchange(h, send, bs, recv, br, true); if (rc) return rc;
    for (int side = 0; side < 2; ++side) h->L->halo_unpack_fields(s, side, c.n_recv[side], c.n_send[side]);
    return SPH_OK;
}

// ---- ghost refreshes and residual reductions of the sharded iterative solvers (DFSPH)
// boundary values of the per-particle scalar `arr` -> the neighbours' ghost copies (4 B per boundary particle)
static int slab_exchange_scalar(SphHandle *h, float *arr) {
    State &s = h->st;
    SlabComm &c = h->comm;
    if (s.push.on) return slab_exchange_push(h, 0, arr, nullptr);
    ProfScope p(h, SPH_K_HALO);
    for (int side = 0; side < 2; ++side) h->L->halo_pack_scalar(s, side, c.n_send[side], c.n_recv[side], arr);
    const void *send[2] = {s.sendbuf[0], s.sendbuf[1]};
    void *recv[2] = {s.recvbuf[0], s.recvbuf[1]};
    const size_t bs[2] = {(size_t)(c.n_send[0] + c.n_recv[0]) * 4, (size_t)(c.n_send[1] + c.n_recv[1]) * 4};
    size_t br[2] = {bs[0], bs[1]};
    int rc = comm_exchange(h, send, bs, recv, br, true); if (rc) return rc;
    for (int side = 0; side < 2; ++side) h->L->halo_unpack_scalar(s, side, c.n_recv[side], c.n_send[side], arr);
    return SPH_OK;
}

// xyz of a float4 per-particle array of the boundary particles -> ghost copies (16 B records, w untouched); default: the
// velocities
static int slab_exchange_vel(SphHandle *h, float4 *arr = nullptr) {
    State &s = h->st;
    SlabComm &c = h->comm;
    if (!arr) arr = s.velm.cur();
    if (s.push.on) return slab_exchange_push(h, 1, nullptr, arr);
    ProfScope p(h, SPH_K_HALO);
    for (int side = 0; side < 2; ++side) h->L->halo_pack_vel(s, side, c.n_send[side], c.n_recv[side], arr);
    const void *send[2] = {s.sendbuf[0], s.sendbuf[1]};
    void *recv[2] = {s.recvbuf[0], s.recvbuf[1]};
    const size_t bs[2] = {(size_t)(c.n_send[0] + c.n_recv[0]) * 16, (size_t)(c.n_send[1] + c.n_recv[1]) * 16};
    size_t br[2] = {bs[0], bs[1]};
    int rc = comm_exchange(h, send, bs, recv, br, true); if (rc) return rc;
    for (int side = 0; side < 2; ++side) h->L->halo_unpack_vel(s, side, c.n_recv[side], c.n_send[side], arr);
    return SPH_OK;
}

// `count` floats in device memory become their sums over all ranks, in place (CG dot products: scal->red[6..7]).
// RCCL: one ncclAllReduce on the compute stream, no host sync.
static int slab_allreduce_dev(SphHandle *h, float *dev, int count) {
    State &s = h->st;
    SlabComm &c = h->comm;
    if (c.nranks <= 1) return SPH_OK;
    ProfScope p(h, SPH_K_HALO);
    if (c.kind == 1) {
        NCCLCHK(h, ncclAllReduce(dev, dev, (size_t)count, ncclFloat, ncclSum, (ncclComm_t)c.nccl, s.stream));
        return SPH_OK;
    }
    float v[8]; double d[8];
    if (count > 8) return fail(h, SPH_ERR_INVALID, "slab_allreduce_dev: count");
    HIPCHK(h, hipMemcpyAsync(v, dev, sizeof(float) * count, hipMemcpyDeviceToHost, s.stream));
    HIPCHK(h, hipStreamSynchronize(s.stream));
    for (int k = 0; k < count; ++k) d[k] = (double)v[k];
    int rc = sph_comm_allreduce(h, d, count, 0); if (rc) return rc;
    for (int k = 0; k < count; ++k) v[k] = (float)d[k];
    HIPCHK(h, hipMemcpy(dev, v, sizeof(float) * count, hipMemcpyHostToDevice));
    return SPH_OK;
}

// scal->red[slot] holds this rank's partial sum (k_reduce_partials): make it the sum over all ranks, in device memory, and
// run the stop test of a device-controlled loop on it.  RCCL: one 4-byte ncclAllReduce on the compute stream, no host sync.
static int slab_finish_reduction(SphHandle *h, int slot) {
    State &s = h->st;
    SlabComm &c = h->comm;
    ProfScope p(h, SPH_K_HALO);
    if (c.nranks > 1) {
        float *red = &s.scal->red[slot];
        if (c.kind == 1) {
            NCCLCHK(h, ncclAllReduce(red, red, 1, ncclFloat, ncclSum, (ncclComm_t)c.nccl, s.stream));
        } else {
            float v = 0.0f;
            HIPCHK(h, hipMemcpyAsync(&v, red, sizeof(float), hipMemcpyDeviceToHost, s.stream));
            HIPCHK(h, hipStreamSynchronize(s.stream));
            double d = (double)v;
            int rc = sph_comm_allreduce(h, &d, 1, 0); if (rc) return rc;
            v = (float)d;
            HIPCHK(h, hipMemcpy(red, &v, sizeof(float), hipMemcpyHostToDevice));
        }
    }
    h->L->loop_criterion(s, slot);
    return SPH_OK;
}
