// sph_halo_impl.hpp -- launchers of the slab-sharding kernels (included inside the per-build namespace)
#pragma once

// inbox layout (sph_halo.hpp): [ctl side 0 | ctl side 1 | rec (side, parity) x 4 | fld (side, parity) x 4]
static inline HaloCtl *inbox_ctl(char *inbox, int side) { return (HaloCtl *)(inbox + (size_t)side * sizeof(HaloCtl)); }
static inline char *inbox_rec(const State &s, char *inbox, int side, unsigned seq) { return inbox + 2 * sizeof(HaloCtl) + (size_t)(side * 2 + (seq & 1u)) * s.push.rec_bytes; }
static inline char *inbox_fld(const State &s, char *inbox, int side, unsigned seq) { return inbox + 2 * sizeof(HaloCtl) + 4 * s.push.rec_bytes + (size_t)(side * 2 + (seq & 1u)) * s.push.fld_bytes; }

// in place: marks the particles this rank drops, tags the ones a neighbour needs and writes the two messages -- into the
// local send buffers, or (push transport) straight into the step-message regions of the neighbours' inboxes
static void l_halo_classify_pack(State &s, int n) {
    HaloArrays a{s.posv.cur(), s.velm.cur(), s.meta.cur(), s.pid.cur(), s.color.cur(), s.rho.cur(), s.xidx[s.xcur], s.orig.cur()};
    float4 *dst[2] = {s.sendbuf[0], s.sendbuf[1]};
    int *counts = s.halo_counts;
    const int *n_dev = nullptr;
    HaloHash hash{nullptr, nullptr, nullptr, nullptr};
    if (s.push.on) {
        // this kernel and k_halo_unpack2 are the step's k_hash_count as well (ph_sort_hashed follows instead of ph_neighbor_search)
        if (!s.cell_count_clean) clear_histogram(s);
        s.cell_count_clean = 0;
        hash = HaloHash{s.cellid, s.rank, s.cell_count, tile_sum_bank(s)};
        s.tile_sums_ready = hash.tile_sum != nullptr;   // (k_halo_unpack2 adds the arrivals' share)
        s.hist_taken = 1;
        const unsigned seq = ++s.push.rec_seq;
        for (int side = 0; side < 2; ++side)
            if (s.push.peer[side]) dst[side] = (float4 *)inbox_rec(s, s.push.peer[side], 1 - side, seq);   // I am the neighbour's OTHER side
        counts = s.halo_counts + HC_BANK * (seq & 1u);   // zeroed by the previous step's k_halo_unpack2
        if (s.async_counts) n_dev = &s.dyn[(seq - 1) & 1u].n_live;
        if (n <= 0) return;
    } else {
        hipMemsetAsync(s.halo_counts, 0, HC_BANK * sizeof(int), s.stream);
        if (n <= 0) return;
    }
    hipLaunchKernelGGL(k_halo_classify, dim3(cdiv(n, 256)), dim3(256), 0, s.stream, s.c, n, n_dev, s.z_lo, s.z_hi, s.has_down, s.has_up, a,
                       dst[0], dst[1], s.push.on ? s.push.rec_cap : s.halo_cap, counts, hash);
}

// Push transport, fused form of the above (HaloSend, sph_halo_defs.hpp): the NEXT step's message is begun here -- its number, its regions
// in the neighbours' inboxes, its count bank -- and the force pass about to be launched fills it; the next step's exchange then starts
// with the hash (l_hash_count) instead of k_halo_classify.
static void l_halo_presend_begin(State &s) {
    const unsigned seq = ++s.push.rec_seq;
    HaloSend &hs = s.presend;
    hs.on = 1; hs.z_lo = s.z_lo; hs.z_hi = s.z_hi; hs.has_down = s.has_down; hs.has_up = s.has_up;
    hs.cap = s.push.rec_cap; hs.rs = s.orig.cur() ? 4 : 3;
    for (int side = 0; side < 2; ++side)
        hs.dst[side] = s.push.peer[side] ? (float4 *)inbox_rec(s, s.push.peer[side], 1 - side, seq) : s.sendbuf[side];
    hs.counts = s.halo_counts + HC_BANK * (seq & 1u);   // zeroed by this step's k_halo_unpack2
    hs.meta_w = s.meta.cur(); hs.xidx = s.xidx[s.xcur];
    hs.pid = s.pid.cur(); hs.color = s.color.cur(); hs.orig = s.orig.cur();
}

// Test hooks: compiled into libsph_hip_testhooks.so only (-DSPH_TEST_HOOKS, csrc/Makefile).  The production library has neither the
// pre-round-5 single step-message header (a protocol with a known race: tests/test_halo_protocol_model.py) nor the consumer delay.
static unsigned halo_hdr_bank_mask() {
#ifdef SPH_TEST_HOOKS
    static const unsigned m = getenv("SPH_TEST_SINGLE_HEADER") ? 0u : 1u;
    return m;
#else
    return 1u;
#endif
}
static long long halo_test_delay_ticks() {   // SPH_TEST_HALO_DELAY_US -> ticks of the 100 MHz wall clock
#ifdef SPH_TEST_HOOKS
    static const long long t = getenv("SPH_TEST_HALO_DELAY_US") ? 100ll * atoll(getenv("SPH_TEST_HALO_DELAY_US")) : 0ll;
    return t;
#else
    return 0;
#endif
}

static int halo_grid(int count_hint) {   // grid-stride kernels: enough workgroups for the hint, at least one, never a huge launch
    int g = cdiv(count_hint > 0 ? count_hint : 1, 256);
    return g > 2048 ? 2048 : g;
}
// The kernels that WAIT for a neighbour's message (every workgroup polls) stay small: a spinning workgroup holds its CU slot, and when
// several ranks share one GPU (the test rig; 8 ranks x hundreds of pollers filled every slot of the chip and the kernels that had to
// PRODUCE the awaited messages could not be scheduled until the waits timed out) the pollers must leave room for everybody's producers.
// 64 workgroups stride through 100 k records in ~6 trips each: a few microseconds.
static int halo_grid_wait(int count_hint) {
    const int g = halo_grid(count_hint);
    return g > 64 ? 64 : g;
}

static void l_halo_unpack2(State &s, int n_old, int bound_app, int bound_live, int count_hint) {
    HaloTables t;
    for (int k = 0; k < 8; ++k) t.tab[k] = s.halo_tab[k];
    const unsigned seq = s.push.rec_seq;
    HaloStep w;
    for (int side = 0; side < 2; ++side) {
        w.out_ctl[side] = s.push.peer[side] ? inbox_ctl(s.push.peer[side], 1 - side) : nullptr;
        w.in_ctl[side] = s.push.peer[side] ? inbox_ctl(s.push.inbox, side) : nullptr;
        w.recv[side] = (const float4 *)inbox_rec(s, s.push.inbox, side, seq);
    }
    w.hdr_bank_mask = halo_hdr_bank_mask();
    w.test_delay_ticks = halo_test_delay_ticks();
    w.seq = seq; w.stride = s.orig.cur() ? 4 : 3; w.cap = s.cap; w.halo_cap = s.push.rec_cap;
    w.n_old = n_old; w.bound_app = bound_app; w.bound_live = bound_live; w.timeout_ticks = s.push.timeout_ticks;
    w.counts = s.halo_counts + HC_BANK * (seq & 1u); w.counts_next = s.halo_counts + HC_BANK * ((seq + 1) & 1u);
    w.dyn_old = s.dyn + ((seq - 1) & 1u); w.dyn_new = s.dyn + (seq & 1u);
    w.mirror = (volatile SlabDyn *)s.push.mirror;
    s.dyn_cur = s.dyn + (seq & 1u);
    hipLaunchKernelGGL(k_halo_unpack2, dim3(halo_grid_wait(count_hint)), dim3(256), 0, s.stream, s.c, w, s.z_lo, s.z_hi, s.posv.cur(), s.velm.cur(),
                       s.meta.cur(), s.pid.cur(), s.color.cur(), s.rho.cur(), s.xidx[s.xcur], s.orig.cur(), t, HaloHash{s.cellid, s.rank, s.cell_count, tile_sum_bank(s)});
}

static HaloFld halo_fld_args(State &s, unsigned seq, float *f0, float4 *v) {
    HaloFld a;
    a.dyn = s.dyn_cur;
    for (int side = 0; side < 2; ++side) {
        a.send_slots[side] = s.halo_tab[HALO_SEND - 1 + side]; a.echo_slots[side] = s.halo_tab[HALO_ECHO_SEND - 1 + side];
        a.ghost_slots[side] = s.halo_tab[HALO_GHOST - 1 + side]; a.eghost_slots[side] = s.halo_tab[HALO_ECHO_GHOST - 1 + side];
        a.out[side] = s.push.peer[side] ? (void *)inbox_fld(s, s.push.peer[side], 1 - side, seq) : nullptr;
        a.in[side] = s.push.peer[side] ? (const void *)inbox_fld(s, s.push.inbox, side, seq) : nullptr;
        a.out_ctl[side] = s.push.peer[side] ? inbox_ctl(s.push.peer[side], 1 - side) : nullptr;
        a.in_ctl[side] = s.push.peer[side] ? inbox_ctl(s.push.inbox, side) : nullptr;
    }
    a.seq = seq; a.timeout_ticks = s.push.timeout_ticks; a.mirror = (volatile SlabDyn *)s.push.mirror;
    a.test_delay_ticks = halo_test_delay_ticks();
    a.f0 = f0; a.f1 = s.rho.cur(); a.f2 = s.prs; a.f3 = s.ptm; a.v = v;
    return a;
}
// kind 0: scalar f0; 1: xyz of v; 2: (rho_raw, rho, prs, ptm)
static void l_halo_push_fields(State &s, int kind, float *f0, float4 *v, int count_hint) {
    HaloFld a = halo_fld_args(s, ++s.push.fld_seq, kind == 2 ? s.rho_raw : f0, v);
    const dim3 g(halo_grid(count_hint)), b(256);
    if (kind == 0) hipLaunchKernelGGL(k_halo_pack2<0>, g, b, 0, s.stream, a);
    else if (kind == 1) hipLaunchKernelGGL(k_halo_pack2<1>, g, b, 0, s.stream, a);
    else hipLaunchKernelGGL(k_halo_pack2<2>, g, b, 0, s.stream, a);
}
static void l_halo_pull_fields(State &s, int kind, float *f0, float4 *v, int count_hint) {
    HaloFld a = halo_fld_args(s, s.push.fld_seq, kind == 2 ? s.rho_raw : f0, v);
    const dim3 g(halo_grid_wait(count_hint)), b(256);
    if (kind == 0) hipLaunchKernelGGL(k_halo_unpack2f<0>, g, b, 0, s.stream, a);
    else if (kind == 1) hipLaunchKernelGGL(k_halo_unpack2f<1>, g, b, 0, s.stream, a);
    else hipLaunchKernelGGL(k_halo_unpack2f<2>, g, b, 0, s.stream, a);
}
// fused form of l_halo_push_fields(kind 2) (HaloFieldSend, sph_common.hpp): the message is begun here, the WCSPH density pass launched
// next stores the values where it computes them; l_halo_pull_fields follows as usual
static void l_halo_fieldsend_begin(State &s) {
    const unsigned seq = ++s.push.fld_seq;
    HaloFieldSend &fs = s.fieldsend;
    fs.on = 1; fs.xidx = s.xidx[s.xcur]; fs.dyn = s.dyn_cur;
    for (int side = 0; side < 2; ++side) fs.out[side] = s.push.peer[side] ? (float4 *)inbox_fld(s, s.push.peer[side], 1 - side, seq) : nullptr;
}
// self-test: my pattern into both neighbours' field regions (as one field message), the handshake, then check theirs
static void l_halo_selftest(State &s, int n, int tag, int tag_down, int tag_up, int *bad_dev) {
    if (!s.dyn_cur) s.dyn_cur = s.dyn;
    HaloFld a = halo_fld_args(s, ++s.push.fld_seq, nullptr, nullptr);
    hipLaunchKernelGGL(k_halo_selftest_push, dim3(halo_grid(n)), dim3(256), 0, s.stream, (float *)a.out[0], (float *)a.out[1], n, tag);
    hipLaunchKernelGGL(k_halo_selftest_check, dim3(halo_grid_wait(n)), dim3(256), 0, s.stream, a, n, tag_down, tag_up, bad_dev);
}

static void l_halo_unpack_append(State &s, int side, int count, int offset) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_halo_unpack, dim3(cdiv(count, 256)), dim3(256), 0, s.stream, s.c, count, offset, side, s.z_lo, s.z_hi,
                       s.recvbuf[side], s.posv.cur(), s.velm.cur(), s.meta.cur(), s.pid.cur(), s.color.cur(), s.rho.cur(),
                       s.xidx[s.xcur], s.orig.cur());
}

static void l_halo_build_tables(State &s) {
    if (s.c.n == 0) return;
    if (s.push.on) {   // k_halo_unpack2 emptied the tables; the per-workgroup prep kernel of this sort fills them (one launch less)
        s.tables_pending = 1; s.perm_n = s.list_n = -1;
        l_block_prep(s);
        return;
    }
    const int longest = s.push.on ? 0 : s.halo_longest;   // longest message of this step (push transport: k_halo_unpack2 reset the tables)
    if (longest > 0)
        hipLaunchKernelGGL(k_halo_tab_reset, dim3(cdiv(longest, 256)), dim3(256), 0, s.stream, longest, s.halo_tab[0], s.halo_tab[1],
                           s.halo_tab[2], s.halo_tab[3], s.halo_tab[4], s.halo_tab[5], s.halo_tab[6], s.halo_tab[7]);
    HaloTables t;
    for (int k = 0; k < 8; ++k) t.tab[k] = s.halo_tab[k];
    hipLaunchKernelGGL(k_halo_tables, dim3(cdiv(s.c.n, 256)), dim3(256), 0, s.stream, s.c.n, s.c.n_dev, s.xidx[s.xcur], t);
}

static void l_halo_pack_fields(State &s, int side, int n_send, int n_recv) {
    if (n_send + n_recv <= 0) return;
    hipLaunchKernelGGL(k_halo_pack_fields, dim3(cdiv(n_send + n_recv, 256)), dim3(256), 0, s.stream, n_send, n_recv,
                       s.halo_tab[HALO_SEND - 1 + side], s.halo_tab[HALO_ECHO_SEND - 1 + side], s.rho_raw, s.rho.cur(), s.prs,
                       s.ptm, s.sendbuf[side]);
}

static void l_halo_unpack_fields(State &s, int side, int n_recv, int n_send) {
    if (n_send + n_recv <= 0) return;
    hipLaunchKernelGGL(k_halo_unpack_fields, dim3(cdiv(n_send + n_recv, 256)), dim3(256), 0, s.stream, n_recv, n_send,
                       s.halo_tab[HALO_GHOST - 1 + side], s.halo_tab[HALO_ECHO_GHOST - 1 + side], s.recvbuf[side], s.rho_raw,
                       s.rho.cur(), s.prs, s.ptm);
}

static void l_halo_pack_scalar(State &s, int side, int n_send, int n_recv, const float *src) {
    if (n_send + n_recv <= 0) return;
    hipLaunchKernelGGL(k_halo_pack_scalar, dim3(cdiv(n_send + n_recv, 256)), dim3(256), 0, s.stream, n_send, n_recv,
                       s.halo_tab[HALO_SEND - 1 + side], s.halo_tab[HALO_ECHO_SEND - 1 + side], src, (float *)s.sendbuf[side]);
}
static void l_halo_unpack_scalar(State &s, int side, int n_recv, int n_send, float *dst) {
    if (n_send + n_recv <= 0) return;
    hipLaunchKernelGGL(k_halo_unpack_scalar, dim3(cdiv(n_send + n_recv, 256)), dim3(256), 0, s.stream, n_recv, n_send,
                       s.halo_tab[HALO_GHOST - 1 + side], s.halo_tab[HALO_ECHO_GHOST - 1 + side], (const float *)s.recvbuf[side], dst);
}
// xyz of a float4 array (velocities: velm, the mass in w stays; PCISPH predicted positions: ppos)
static void l_halo_pack_vel(State &s, int side, int n_send, int n_recv, const float4 *arr) {
    if (n_send + n_recv <= 0) return;
    hipLaunchKernelGGL(k_halo_pack_vel, dim3(cdiv(n_send + n_recv, 256)), dim3(256), 0, s.stream, n_send, n_recv,
                       s.halo_tab[HALO_SEND - 1 + side], s.halo_tab[HALO_ECHO_SEND - 1 + side], arr, s.sendbuf[side]);
}
static void l_halo_unpack_vel(State &s, int side, int n_recv, int n_send, float4 *arr) {
    if (n_send + n_recv <= 0) return;
    hipLaunchKernelGGL(k_halo_unpack_vel, dim3(cdiv(n_send + n_recv, 256)), dim3(256), 0, s.stream, n_recv, n_send,
                       s.halo_tab[HALO_GHOST - 1 + side], s.halo_tab[HALO_ECHO_GHOST - 1 + side], s.recvbuf[side], arr);
}
static void l_loop_criterion(State &s, int slot) {
    if (!s.loop_flag || s.loop_slot != slot) return;
    hipLaunchKernelGGL(k_loop_criterion, dim3(1), dim3(1), 0, s.stream, s.scal, slot, s.loop_kind, s.loop_denom, s.loop_thr);
}

static void l_layer_hist(State &s, int *hist) {
    hipMemsetAsync(hist, 0, sizeof(int) * (size_t)(s.c.slab_axis == 0 ? s.c.nx_glob : s.c.nz_glob), s.stream);
    const int nloc = s.c.slab_axis == 0 ? s.c.nx : s.c.nz;
    if (s.c.n > 0) hipLaunchKernelGGL(k_layer_hist, dim3(nloc), dim3(256), 0, s.stream, s.c, s.cell_start, s.z_lo, s.z_hi, hist);
}

static void l_count_ghosts(State &s, int *out) {
    hipMemsetAsync(out, 0, sizeof(int), s.stream);
    if (s.c.n > 0) hipLaunchKernelGGL(k_count_ghosts, dim3(cdiv(s.c.n, 256)), dim3(256), 0, s.stream, s.c.n, s.meta.cur(), out);
}
