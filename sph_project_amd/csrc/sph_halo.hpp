// sph_halo.hpp -- slab sharding, device side (included inside the per-build namespace).
//
// The slab axis is one of the library's axes (Consts::slab_axis: z, the fastest axis of the cell order, by default; x, the slowest,
// with SPH_SLAB_LAYOUT=slow -- sph_common.hpp).  `z_lo` / `z_hi` keep their names from the API (sph_comm_set_slab: cell layers of the
// SCENE's z): here they are layer numbers along the slab axis.
// Rank r owns the global cell layers [z_lo, z_hi) (>= 2 layers) and keeps one ghost layer per interior
// side.  Once per step, right before the sort, ONE message per neighbour carries both kinds of
// records (48 B each: posv, velm, meta|pid|color|rho; 64 B with the rest position when the scene has a dynamic rigid body):
//   * migrants: particles that left the slab -> ownership moves (ghost flag 0);
//   * boundary copies: owned particles in layer z_lo / z_hi-1 -> ghosts of the neighbour (ghost flag 1).
// A migrant that lands in the neighbour's boundary layer (it moved < 1 layer) is needed back as a ghost; both
// sides can tell that from the record itself, so the sender keeps it as a local ghost ("echo ghost", indexed by
// its position k in the sender's message) and the receiver tags it "echo send k": no second round.
// xidx = (kind << 28 | k) rides through the sort; k_halo_tables turns it into slot tables so that later field
// exchanges (density / pressure of the ghosts after the density pass) are gather -> message -> scatter with
// message sizes both sides already know.
#pragma once

// (record kinds, HaloArrays, halo_write_record, halo_wave_slot and the classification rule itself: sph_halo_defs.hpp)

// ---- push transport (sph_comm.hpp): a rank's INBOX is one device allocation that its two neighbours map through hipIpc and
// write into directly -- no send buffer, no copy engine, no host.  Per side (= which neighbour writes): a control block, two
// step-message regions (parity of the message number) and two field-message regions.
//   writer:  kernel K stores the payload (k_halo_classify: the records; k_halo_pack2: the field values) with plain stores;
//            the NEXT kernel on the stream (k_halo_unpack2 / k_halo_unpack2f, which also consume the neighbours' messages)
//            begins -- workgroup 0, one lane per neighbour -- by storing the header, a system-scope fence, and the message
//            number (release).  The kernel boundary in between has completed all payload stores; stores from one device to
//            one destination are delivered in order, so whoever sees the number sees the payload.
//   reader:  every workgroup of the consuming kernel polls the number in ITS OWN inbox (relaxed load + s_sleep, bounded by
//            the wall clock: a neighbour that never answers raises SLAB_ST_TIMEOUT instead of hanging the GPU), acquires,
//            reads the header, and consumes the payload in the same launch.  No separate wait kernel, no per-workgroup
//            fences, no tickets: the step message costs one extra launch (classify) on top of the unpack, a field message two.
// Both sides run the same sequence of exchanges, so message numbers stay in lockstep; message m + 2 can only be written after
// the writer has received the reader's message m + 1, which was sent after message m had been consumed: two regions per
// direction suffice.  The per-step counts live in device memory, double-buffered by message parity (SlabDyn[2]: a kernel
// reads last step's bank and writes this step's; halo_counts[2][4] likewise), and are mirrored into pinned host memory.
// (HaloCtl, the control block of one inbox side, and SlabDyn: sph_common.hpp)
__device__ __forceinline__ void halo_store_sys(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ int halo_load_sys(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ bool halo_poll(const unsigned *word, unsigned seq, long long timeout_ticks) {
    const long long t0 = (long long)wall_clock64();
    for (;;) {
        const unsigned v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int)(v - seq) >= 0) return true;
        if ((long long)wall_clock64() - t0 > timeout_ticks) return false;
        __builtin_amdgcn_s_sleep(16);
    }
}

// Test hooks (-DSPH_TEST_HOOKS: libsph_hip_testhooks.so only, never the production library; csrc/Makefile).  halo_test_delay stalls a
// consumer between its own announce and its poll (workgroup 0) or before its poll (the others, staggered by workgroup): what time
// slicing of several ranks on one GPU does at random, on demand.  With the pre-round-5 single header (SPH_TEST_SINGLE_HEADER) the stall
// makes the race of tests/test_halo_protocol_model.py deterministic; with the header per message parity nothing may change.
__device__ __forceinline__ void halo_test_delay(long long ticks) {
#ifdef SPH_TEST_HOOKS
    if (ticks <= 0) return;
    const long long d = blockIdx.x == 0 ? ticks : ticks * (long long)(1 + (blockIdx.x % 3)) / 4;
    const long long t0 = (long long)wall_clock64();
    while ((long long)wall_clock64() - t0 < d) __builtin_amdgcn_s_sleep(64);
#endif
}

// In place: nothing is compacted or copied.  Particles that are no longer this rank's business (last step's ghosts,
// migrants beyond the neighbour's boundary layer) get the DEAD bit; the sort that follows files them into the
// graveyard cell G behind every live particle, and the live count shrinks by counts[2].
// counts[0] = records for the lower rank, counts[1] = upper rank, counts[2] = particles that died.
// send_down / send_up: the local send buffers, or (push transport) the step-message regions of the neighbours' inboxes.
// hash != null (push transport): the kernel is also this step's k_hash_count for the particles that are here already (cell id,
// histogram, arrival rank; the dead ones into the graveyard cell G) -- both read the same positions, and the arrivals are hashed
// by k_halo_unpack2 when they land.
struct HaloHash { int *cellid, *rank, *cell_count, *tile_sum; };
__global__ void __launch_bounds__(256)
k_halo_classify(const Consts c, int n_host, const int *__restrict__ n_dev, int z_lo, int z_hi, int has_down, int has_up, HaloArrays a,
                float4 *send_down, float4 *send_up, int cap, int *counts, HaloHash hash) {
    const int n = n_dev ? *n_dev : n_host;
    const int i = blockIdx.x * 256 + threadIdx.x;
    int side = -1, dead = 0, xi = 0, mnew = 0, mrec = 0;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
        const int m = a.meta[i];
        if (!(META_GHOST(m) || META_DEAD(m))) p = a.posv[i];
        const HaloVerdict vd = halo_classify_one(m, slab_layer(c, p), z_lo, z_hi, has_down, has_up);   // (sph_halo_defs.hpp)
        side = vd.side; dead = vd.dead; mnew = vd.mnew; mrec = vd.mrec;
    }
    const int k0 = halo_wave_slot(side == 0, &counts[HC(0)]);
    const int k1 = halo_wave_slot(side == 1, &counts[HC(1)]);
    halo_wave_slot(dead != 0, &counts[HC(2)]);
    if (hash.cellid) {   // k_hash_count (sph_device.hpp), same wave-aggregated atomics: one per run of equal cell ids
        const int lane = threadIdx.x & 63;
        int lin = -1 - lane;
        if (i < n) {
            if (dead) lin = c.G + ((i >> 6) & (SPH_NGRAVE - 1));   // (one of the graveyard cells: by wave, so that a run of dead lanes stays one run)
            else {
                const float4 q = a.posv[i];
                lin = (cell_coord_x(c, q.x) * c.ny + cell_coord(q.y, c.grid_size, c.ny)) * c.nz + cell_coord_z(c, q.z);
            }
            hash.cellid[i] = lin;
        }
        const int prev = __shfl_up(lin, 1, 64);
        const bool head = lane == 0 || lin != prev;
        const unsigned long long hm = __ballot(head);
        const unsigned long long upto = hm & ((2ull << lane) - 1ull);
        const int hl = 63 - __clzll(upto);
        const unsigned long long above = lane == 63 ? 0ull : (hm >> (lane + 1));
        const int len = above ? __ffsll(above) : 64 - lane;
        int base = 0;
        if (head && i < n) base = atomicAdd(&hash.cell_count[lin], len);
        base = __shfl(base, hl, 64);
        if (i < n) hash.rank[i] = base + (lane - hl);
        if (hash.tile_sum) tile_sum_add(hash.tile_sum, lin, i < n, c.G + SPH_NGRAVE);
    }
    if (i >= n) return;
    if (side >= 0) {
        const int k = side == 0 ? k0 : k1;
        if (k < cap) {
            float4 *buf = side == 0 ? send_down : send_up;
            const int rs = a.orig ? 4 : 3;
            halo_write_record(buf, rs, k, p, a.velm[i], mrec, a.pid[i], a.color[i], a.rho[i]);
            if (a.orig) buf[rs * k + 3] = a.orig[i];
        }
        const bool migrant = !META_GHOST(mrec);
        xi = HALO_PACK((migrant ? HALO_ECHO_GHOST : HALO_SEND) + side, k);
    }
    if (dead) { mnew |= 1 << 12; xi = 0; }
    a.meta[i] = mnew;
    a.xidx[i] = xi;
}

// Step message, receiving side and settlement of the step's counts (see the protocol above).
struct HaloStep {
    HaloCtl *out_ctl[2];          // the neighbours' control blocks for messages from me (null: no neighbour on that side)
    const HaloCtl *in_ctl[2];     // my inbox control blocks
    const float4 *recv[2];        // my step-message regions for this message
    unsigned seq;
    unsigned hdr_bank_mask;       // 1: header of message m in rec[m & 1] (normal); 0: one header for all messages (SPH_TEST_SINGLE_HEADER: the pre-round-5 protocol, for the A/B that pins the race)
    int stride, cap, halo_cap;
    long long test_delay_ticks;   // test-hook build only (halo_test_delay), else 0
    int n_old;                    // particle count before the exchange when the host knows it exactly, else -1: last step's n_live
    int bound_app, bound_live;    // launch bounds of an asynchronous step (0: the host launches exact grids afterwards)
    long long timeout_ticks;      // of the 100 MHz wall clock
    const int *counts;            // what k_halo_classify just counted (bank seq & 1)
    int *counts_next;             // the other bank: zeroed here for the next step's classify
    const SlabDyn *dyn_old;       // last step's counts (bank (seq - 1) & 1)
    SlabDyn *dyn_new;             // this step's (bank seq & 1); the status word only ever accumulates (sticky)
    volatile SlabDyn *mirror;     // pinned host copy
};
struct HaloTables { int *tab[8]; };
__global__ void __launch_bounds__(256)
k_halo_unpack2(const Consts c, HaloStep w, int z_lo, int z_hi, float4 *posv, float4 *velm, int *meta, int *pid, unsigned *color,
               float *rho, int *xidx, float4 *orig, HaloTables t, HaloHash hash) {
    __shared__ int s_v[6];   // r0, r1, n_old, longest, status
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        // my own header first (workgroup 0): a neighbour waiting for it is released before I start waiting for its
        if (blockIdx.x == 0 && lane < 2 && w.out_ctl[lane]) {
            HaloRecHdr *hd = &w.out_ctl[lane]->rec[w.seq & w.hdr_bank_mask];
            const int cnt = w.counts[HC(lane)];
            halo_store_sys(&hd->count, cnt);
            halo_store_sys(&hd->status, w.dyn_old->status | (cnt > w.halo_cap ? SLAB_ST_SEND_OVERFLOW : 0));
            halo_store_sys(&hd->stride, w.stride);
            __threadfence_system();
            __hip_atomic_store(&hd->seq, w.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        halo_test_delay(w.test_delay_ticks);
        int cnt = 0, st = 0;
        if (lane < 2 && w.in_ctl[lane]) {
            const HaloRecHdr *hd = &w.in_ctl[lane]->rec[w.seq & w.hdr_bank_mask];   // (this message's own header: the next one goes to the other)
            // (after a time-out the exchange is dead: later waits give up at once instead of stacking 30 s each on the stream)
            const long long patience = (w.dyn_old->status & SLAB_ST_TIMEOUT) ? 0 : w.timeout_ticks;
            if (!halo_poll(&hd->seq, w.seq, patience)) st |= SLAB_ST_TIMEOUT;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // system scope: the header and the records behind the number
            if (!st) {
                cnt = halo_load_sys(&hd->count);
                if (halo_load_sys(&hd->status)) st |= SLAB_ST_PEER;
                if (halo_load_sys(&hd->stride) != w.stride) st |= SLAB_ST_STRIDE;
                if (cnt < 0 || cnt > w.halo_cap) { st |= SLAB_ST_PEER; cnt = 0; }
            }
        }
        int r0 = __shfl(cnt, 0, 64), r1 = __shfl(cnt, 1, 64);
        st = __shfl(st, 0, 64) | __shfl(st, 1, 64);
        if (lane == 0) {
            const int s0 = w.counts[HC(0)], s1 = w.counts[HC(1)], dropped = w.counts[HC(2)];
            st |= w.dyn_old->status;                        // sticky
            if (s0 > w.halo_cap || s1 > w.halo_cap) st |= SLAB_ST_SEND_OVERFLOW;
            const int n_old = w.n_old >= 0 ? w.n_old : w.dyn_old->n_live;
            if (st) r0 = r1 = 0;                            // after a failure nothing is appended: every later kernel stays inside its arrays
            if ((long long)n_old + r0 + r1 > (long long)w.cap) { st |= SLAB_ST_CAPACITY; r0 = r1 = 0; }
            int n_app = n_old + r0 + r1, n_live = n_app - dropped;
            if (w.bound_app > 0 && (n_app > w.bound_app || n_live > w.bound_live)) {
                // the grids of this step were launched for fewer particles: appending the arrivals would leave some of them unhashed and
                // unscattered and the cell lists inconsistent for up to SLAB_MAX_LAG more steps.  Nothing is appended (as for every other
                // failure); the arrivals of this step are LOST, the state is not recoverable: restart with SPH_SLAB_ASYNC=0.
                st |= SLAB_ST_BOUND; r0 = r1 = 0;
                n_app = n_old; n_live = n_app - dropped;
            }
            int longest = s0 > s1 ? s0 : s1;
            longest = longest > r0 ? longest : r0; longest = longest > r1 ? longest : r1;
            longest = longest > w.halo_cap ? w.halo_cap : longest;
            s_v[0] = r0; s_v[1] = r1; s_v[2] = n_old; s_v[3] = longest; s_v[4] = st;
            // a workgroup that timed out while another did not would append a different set: every workgroup leaves its
            // verdict in the status word, which the host (and the next header) sees
            if (st) atomicOr(&w.dyn_new->status, st);
            if (blockIdx.x == 0) {
                SlabDyn *d = w.dyn_new;
                d->n_app = n_app; d->n_live = n_live; d->n_send[0] = s0 > w.halo_cap ? w.halo_cap : s0; d->n_send[1] = s1 > w.halo_cap ? w.halo_cap : s1;
                d->n_recv[0] = r0; d->n_recv[1] = r1; d->dropped = dropped; d->longest = longest; d->seq = w.seq;
                w.counts_next[HC(0)] = w.counts_next[HC(1)] = w.counts_next[HC(2)] = 0;   // the next classify starts from zero without a memset
                if (w.mirror) {
                    volatile SlabDyn *m = w.mirror;
                    m->wseq = 2u * w.seq + 1u;   // odd: fields in flux (the host's reader retries)
                    __threadfence_system();
                    m->n_app = n_app; m->n_live = n_live; m->n_send[0] = d->n_send[0]; m->n_send[1] = d->n_send[1];
                    m->n_recv[0] = r0; m->n_recv[1] = r1; m->dropped = dropped; m->longest = longest; m->status = st;
                    __threadfence_system();
                    m->seq = w.seq;
                    __threadfence_system();
                    m->wseq = 2u * w.seq + 2u;   // even again: a host that reads wseq, the fields, then the same even wseq has one consistent set
                }
            }
        }
    }
    __syncthreads();
    const int r0 = s_v[0], r1 = s_v[1], n_old = s_v[2];
    const int rs = orig ? 4 : 3;
    const int stride = (int)(gridDim.x * 256);
    for (int q = blockIdx.x * 256 + threadIdx.x; q < r0 + r1; q += stride) {
        const int side = q < r0 ? 0 : 1;
        const int k = side ? q - r0 : q;
        const float4 *recv = w.recv[side];
        const float4 p = recv[rs * k];
        const float4 v4 = recv[rs * k + 2];
        const int m = __float_as_int(v4.x);
        const int d = n_old + q;
        posv[d] = p; velm[d] = recv[rs * k + 1];
        if (orig) orig[d] = recv[rs * k + 3];
        meta[d] = m; pid[d] = __float_as_int(v4.y); color[d] = __float_as_uint(v4.z); rho[d] = v4.w;
        int xi;
        if (META_GHOST(m)) xi = HALO_PACK(HALO_GHOST + side, k);
        else {  // a migrant, now owned here; echo it back if it sits in my boundary layer facing the sender
            const int cz = slab_layer(c, p);
            const int edge = side == 0 ? z_lo : z_hi - 1;
            xi = cz == edge ? HALO_PACK(HALO_ECHO_SEND + side, k) : 0;
        }
        xidx[d] = xi;
        if (hash.cellid) {   // the arrival's share of this step's k_hash_count
            const int lin = (cell_coord_x(c, p.x) * c.ny + cell_coord(p.y, c.grid_size, c.ny)) * c.nz + cell_coord_z(c, p.z);
            hash.cellid[d] = lin;
            hash.rank[d] = atomicAdd(&hash.cell_count[lin], 1);
            if (hash.tile_sum && (lin >> SCAN_TILE_SHIFT) != ((c.G + SPH_NGRAVE - 1) >> SCAN_TILE_SHIFT))   // (the last tile's sum is never read: tile_sum_add)
                atomicAdd(&hash.tile_sum[(lin >> SCAN_TILE_SHIFT) * SCAN_PARTIAL_STRIDE], 1);   // arrivals: a few thousand per step, spread over the ghost layers' tiles
        }
    }
    const int longest = s_v[3];
    for (int k = blockIdx.x * 256 + threadIdx.x; k < longest; k += stride) {
#pragma unroll
        for (int q = 0; q < 8; ++q) t.tab[q][k] = -1;
    }
}

// appends `count` records received from `side` (0 = lower rank, 1 = upper rank) at [offset, offset + count)
__global__ void __launch_bounds__(256)
k_halo_unpack(const Consts c, int count, int offset, int side, int z_lo, int z_hi, const float4 *recv, float4 *posv,
              float4 *velm, int *meta, int *pid, unsigned *color, float *rho, int *xidx, float4 *orig) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    const int rs = orig ? 4 : 3;
    const float4 p = recv[rs * k];
    const float4 q = recv[rs * k + 2];
    const int m = __float_as_int(q.x);
    const int d = offset + k;
    posv[d] = p; velm[d] = recv[rs * k + 1];
    if (orig) orig[d] = recv[rs * k + 3];
    meta[d] = m; pid[d] = __float_as_int(q.y); color[d] = __float_as_uint(q.z); rho[d] = q.w;
    int xi;
    if (META_GHOST(m)) xi = HALO_PACK(HALO_GHOST + side, k);
    else {  // a migrant, now owned here; echo it back if it sits in my boundary layer facing the sender
        const int cz = slab_layer(c, p);
        const int edge = side == 0 ? z_lo : z_hi - 1;
        xi = cz == edge ? HALO_PACK(HALO_ECHO_SEND + side, k) : 0;
    }
    xidx[d] = xi;
}

// slot tables start out empty (-1) up to the longest message of this step
__global__ void __launch_bounds__(256) k_halo_tab_reset(int count, int *t0, int *t1, int *t2, int *t3, int *t4, int *t5, int *t6, int *t7) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    t0[k] = -1; t1[k] = -1; t2[k] = -1; t3[k] = -1; t4[k] = -1; t5[k] = -1; t6[k] = -1; t7[k] = -1;
}

// after the sort: slot tables from the xidx that rode along.  tab[kind - 1] for kinds 1..8.
__global__ void __launch_bounds__(256) k_halo_tables(int n, const int *__restrict__ n_dev, const int *xidx, HaloTables t) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (n_dev) n = *n_dev;
    if (i >= n) return;
    const int x = xidx[i];
    const int kind = HALO_KIND(x);
    if (kind >= 1 && kind <= 8) t.tab[kind - 1][HALO_IDX(x)] = i;
}

// density / pressure of boundary particles after the density pass: (rho_raw, rho, prs, ptm).
// Message to `side`: [ my n_send records' slots (send table) | the n_recv records I got from that side (echo-send table) ].
__global__ void __launch_bounds__(256)
k_halo_pack_fields(int n_send, int n_recv, const int *send_slots, const int *echo_slots, const float *rho_raw,
                   const float *rho, const float *prs, const float *ptm, float4 *buf) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_send + n_recv) return;
    const int s = k < n_send ? send_slots[k] : echo_slots[k - n_send];
    buf[k] = s >= 0 ? make_float4(rho_raw[s], rho[s], prs[s], ptm[s]) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// Message from `side`: [ its n_recv records (ghost table) | my n_send records (echo-ghost table) ].
__global__ void __launch_bounds__(256)
k_halo_unpack_fields(int n_recv, int n_send, const int *ghost_slots, const int *echo_ghost_slots, const float4 *buf,
                     float *rho_raw, float *rho, float *prs, float *ptm) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_recv + n_send) return;
    const int s = k < n_recv ? ghost_slots[k] : echo_ghost_slots[k - n_recv];
    if (s < 0) return;
    const float4 q = buf[k];
    rho_raw[s] = q.x; rho[s] = q.y; prs[s] = q.z; ptm[s] = q.w;
}

// ---- per-pass ghost refreshes of the iterative solvers (DFSPH / PCISPH exchange kappa (4 B) and v (12 B) of
// the boundary particles once per solver iteration).  Same slot tables and message layout as the density / pressure
// exchange above: message to `side` = [ my n_send records | the n_recv records I got from that side ].
__global__ void __launch_bounds__(256)
k_halo_pack_scalar(int n_send, int n_recv, const int *send_slots, const int *echo_slots, const float *src, float *buf) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_send + n_recv) return;
    const int s = k < n_send ? send_slots[k] : echo_slots[k - n_send];
    buf[k] = s >= 0 ? src[s] : 0.0f;
}
__global__ void __launch_bounds__(256)
k_halo_unpack_scalar(int n_recv, int n_send, const int *ghost_slots, const int *echo_ghost_slots, const float *buf, float *dst) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_recv + n_send) return;
    const int s = k < n_recv ? ghost_slots[k] : echo_ghost_slots[k - n_recv];
    if (s >= 0) dst[s] = buf[k];
}
__global__ void __launch_bounds__(256)
k_halo_pack_vel(int n_send, int n_recv, const int *send_slots, const int *echo_slots, const float4 *velm, float4 *buf) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_send + n_recv) return;
    const int s = k < n_send ? send_slots[k] : echo_slots[k - n_send];
    buf[k] = s >= 0 ? velm[s] : make_float4(0.f, 0.f, 0.f, 0.f);
}
__global__ void __launch_bounds__(256)
k_halo_unpack_vel(int n_recv, int n_send, const int *ghost_slots, const int *echo_ghost_slots, const float4 *buf, float4 *velm) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_recv + n_send) return;
    const int s = k < n_recv ? ghost_slots[k] : echo_ghost_slots[k - n_recv];
    if (s < 0) return;
    const float4 q = buf[k];
    float4 v = velm[s];
    v.x = q.x; v.y = q.y; v.z = q.z;    // the mass of the ghost copy arrived with its record
    velm[s] = v;
}

// ---- field messages of the push transport: both sides in ONE launch each way, payload straight into the neighbours' inboxes,
// counts from SlabDyn (grid-stride: the grid is a hint).  k_halo_pack2 stores; k_halo_unpack2f announces (workgroup 0), waits
// (every workgroup, bounded) and scatters -- see the protocol at the top.
// KIND 0: one float (kappa, rho, p / rho^2 ...); 1: xyz of a float4 array (v, predicted x, cg_p; w stays); 2: the four
// scalars that follow the density pass (rho_raw, rho, p, p / rho^2).  Layout of a message as for the other transports:
// to `side`: [ my n_send records (send table) | the n_recv records I got from that side (echo-send table) ].
struct HaloFld {
    SlabDyn *dyn;                                 // this step's counts
    const int *send_slots[2], *echo_slots[2];     // pack: tables HALO_SEND + side, HALO_ECHO_SEND + side
    const int *ghost_slots[2], *eghost_slots[2];  // unpack: tables HALO_GHOST + side, HALO_ECHO_GHOST + side
    void *out[2];          // the neighbours' field regions for this message (pack) / null
    const void *in[2];     // my own field regions (unpack)
    HaloCtl *out_ctl[2];   // the neighbours' control blocks
    const HaloCtl *in_ctl[2];
    unsigned seq;
    long long timeout_ticks;
    long long test_delay_ticks;   // test-hook build only (halo_test_delay), else 0
    volatile SlabDyn *mirror;
    float *f0, *f1, *f2, *f3;   // KIND 0: f0; KIND 2: rho_raw, rho, prs, ptm
    float4 *v;                  // KIND 1
};
template <int KIND>
__global__ void __launch_bounds__(256) k_halo_pack2(HaloFld a) {
    const int ns0 = a.dyn->n_send[0], nr0 = a.dyn->n_recv[0], ns1 = a.dyn->n_send[1], nr1 = a.dyn->n_recv[1];
    const int t0 = a.out[0] ? ns0 + nr0 : 0, t1 = a.out[1] ? ns1 + nr1 : 0;
    const int stride = (int)(gridDim.x * 256);
    for (int q = blockIdx.x * 256 + threadIdx.x; q < t0 + t1; q += stride) {
        const int side = q < t0 ? 0 : 1;
        const int k = side ? q - t0 : q;
        const int ns = side ? ns1 : ns0;
        const int sl = k < ns ? a.send_slots[side][k] : a.echo_slots[side][k - ns];
        if (KIND == 0) ((float *)a.out[side])[k] = sl >= 0 ? a.f0[sl] : 0.0f;
        else if (KIND == 1) ((float4 *)a.out[side])[k] = sl >= 0 ? a.v[sl] : make_float4(0.f, 0.f, 0.f, 0.f);
        else ((float4 *)a.out[side])[k] = sl >= 0 ? make_float4(a.f0[sl], a.f1[sl], a.f2[sl], a.f3[sl]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
// announce my field message (workgroup 0), then wait for the neighbours' (every workgroup); false after a time-out
__device__ __forceinline__ bool halo_fld_handshake(HaloCtl *const out_ctl[2], const HaloCtl *const in_ctl[2], unsigned seq,
                                                   long long timeout_ticks, SlabDyn *dyn, volatile SlabDyn *mirror, long long test_delay_ticks = 0) {
    __shared__ int s_ok;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (blockIdx.x == 0 && lane < 2 && out_ctl[lane]) {
            __threadfence_system();
            __hip_atomic_store(&out_ctl[lane]->fld_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        halo_test_delay(test_delay_ticks);
        int st = 0;
        const long long patience = (dyn->status & SLAB_ST_TIMEOUT) ? 0 : timeout_ticks;   // fail fast once the exchange is dead
        if (lane < 2 && in_ctl[lane] && !halo_poll(&in_ctl[lane]->fld_seq, seq, patience)) st = SLAB_ST_TIMEOUT;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        st = __shfl(st, 0, 64) | __shfl(st, 1, 64);
        if (lane == 0) {
            s_ok = st == 0;
            if (st) { atomicOr(&dyn->status, st); if (mirror) mirror->status = st; }
        }
    }
    __syncthreads();
    return s_ok != 0;
}
// message from `side`: [ its records = my n_recv (ghost table) | my n_send records (echo-ghost table) ]
template <int KIND>
__global__ void __launch_bounds__(256) k_halo_unpack2f(HaloFld a) {
    if (!halo_fld_handshake(a.out_ctl, a.in_ctl, a.seq, a.timeout_ticks, a.dyn, a.mirror, a.test_delay_ticks)) return;   // stale payload is not scattered
    const int ns0 = a.dyn->n_send[0], nr0 = a.dyn->n_recv[0], ns1 = a.dyn->n_send[1], nr1 = a.dyn->n_recv[1];
    const int t0 = a.in[0] ? ns0 + nr0 : 0, t1 = a.in[1] ? ns1 + nr1 : 0;
    const int stride = (int)(gridDim.x * 256);
    for (int q = blockIdx.x * 256 + threadIdx.x; q < t0 + t1; q += stride) {
        const int side = q < t0 ? 0 : 1;
        const int k = side ? q - t0 : q;
        const int nr = side ? nr1 : nr0;
        const int sl = k < nr ? a.ghost_slots[side][k] : a.eghost_slots[side][k - nr];
        if (sl < 0) continue;
        if (KIND == 0) a.f0[sl] = ((const float *)a.in[side])[k];
        else if (KIND == 1) {
            const float4 w = ((const float4 *)a.in[side])[k];
            float4 v = a.v[sl];
            v.x = w.x; v.y = w.y; v.z = w.z;    // w (mass / volume of the ghost copy) arrived with its record
            a.v[sl] = v;
        } else {
            const float4 w = ((const float4 *)a.in[side])[k];
            a.f0[sl] = w.x; a.f1[sl] = w.y; a.f2[sl] = w.z; a.f3[sl] = w.w;
        }
    }
}

// transport self-test of the push path: a pattern into the neighbours' field regions, then the same handshake and a check
__global__ void __launch_bounds__(256) k_halo_selftest_push(float *out0, float *out1, int n, int tag) {
    const int stride = (int)(gridDim.x * 256);
    for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += stride) {
        if (out0) out0[k] = (float)(tag * 1000 + (k % 997));
        if (out1) out1[k] = (float)(tag * 1000 + 500 + (k % 997));
    }
}
__global__ void __launch_bounds__(256) k_halo_selftest_check(HaloFld a, int n, int tag0, int tag1, int *bad) {
    if (!halo_fld_handshake(a.out_ctl, a.in_ctl, a.seq, a.timeout_ticks, a.dyn, a.mirror)) { if (threadIdx.x == 0) atomicAdd(bad, 1 << 20); return; }
    const float *in0 = (const float *)a.in[0], *in1 = (const float *)a.in[1];
    const int stride = (int)(gridDim.x * 256);
    int b = 0;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += stride) {
        if (in0 && in0[k] != (float)(tag0 * 1000 + 500 + (k % 997))) b++;   // my lower neighbour wrote its "up" pattern
        if (in1 && in1[k] != (float)(tag1 * 1000 + (k % 997))) b++;         // my upper neighbour wrote its "down" pattern
    }
    if (b) atomicAdd(bad, b);
}

// Stop test of a device-controlled solver loop on a residual that was all-reduced over the ranks behind
// k_reduce_partials (which, under sharding, only leaves this rank's sum in scal->red[slot]).
__global__ void k_loop_criterion(DevScalars *scal, int slot, int kind, float denom, double thr) {
    if (scal->flags[0]) return;
    const float avg = denom > 0.0f ? scal->red[slot] / denom : 0.0f;
    scal->flags[1] += 1;
    if (kind == 1 ? ((double)avg <= thr) : (avg < (float)thr)) scal->flags[0] = 1;
}

// owned particles per global cell layer (slab rebalancing: the cuts follow the fluid), from the cell lists of the last sort: one workgroup
// per local layer adds up the populations of its cells -- no atomics.  (Until round 4: one atomicAdd per particle onto the ~50 layer
// counters; atomics on one address are served one at a time, and the kernel took 6 ms at 1.23 M particles -- 0.09 ms per step at a
// re-plan every 64 steps.  A particle that changed layer since the last sort is counted where it was: this is a planning figure.)
__global__ void __launch_bounds__(256)
k_layer_hist(const Consts c, const int *__restrict__ cell_start, int z_lo, int z_hi, int *__restrict__ hist) {
    __shared__ int s_w[4];
    const int L = blockIdx.x;                                      // local layer
    const int glob = L + (c.slab_axis == 0 ? c.cx_off : c.cz_off);
    if (glob < z_lo || glob >= z_hi) return;                       // ghost layer (uniform)
    int sum = 0;
    if (c.slab_axis == 0) {                                        // the layer is one contiguous stretch of cells
        if (threadIdx.x == 0) sum = cell_start[(L + 1) * c.ny * c.nz] - cell_start[L * c.ny * c.nz];
    } else {
        for (int col = threadIdx.x; col < c.nx * c.ny; col += 256) { const int lin = col * c.nz + L; sum += cell_start[lin + 1] - cell_start[lin]; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) hist[glob] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

// number of ghost copies among the first n particles (sph_comm_get_slab): one atomic per wave
__global__ void __launch_bounds__(256) k_count_ghosts(int n, const int *meta, int *out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const unsigned long long m = __ballot(i < n && META_GHOST(meta[i]));
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(out, __popcll(m));
}
