// sph_halo_impl.hpp -- launchers of the slab-sharding kernels (included inside the per-build namespace)
#pragma once

// in place: marks the particles this rank drops, tags the ones a neighbour needs and writes the two messages
static void l_halo_classify_pack(State &s, int n) {
    hipMemsetAsync(s.halo_counts, 0, 4 * sizeof(int), s.stream);
    if (n > 0) {
        HaloArrays a{s.posv.cur(), s.velm.cur(), s.meta.cur(), s.pid.cur(), s.color.cur(), s.rho.cur(), s.xidx[s.xcur], s.orig.cur()};
        hipLaunchKernelGGL(k_halo_classify, dim3(cdiv(n, 256)), dim3(256), 0, s.stream, s.c, n, s.z_lo, s.z_hi, s.has_down,
                           s.has_up, a, s.sendbuf[0], s.sendbuf[1], s.halo_cap, s.halo_counts);
    }
}

static void l_halo_unpack_append(State &s, int side, int count, int offset) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_halo_unpack, dim3(cdiv(count, 256)), dim3(256), 0, s.stream, s.c, count, offset, side, s.z_lo, s.z_hi,
                       s.recvbuf[side], s.posv.cur(), s.velm.cur(), s.meta.cur(), s.pid.cur(), s.color.cur(), s.rho.cur(),
                       s.xidx[s.xcur], s.orig.cur());
}

static void l_halo_build_tables(State &s) {
    if (s.c.n == 0) return;
    const int longest = s.halo_longest;   // longest message of this step (sent or received)
    if (longest > 0)
        hipLaunchKernelGGL(k_halo_tab_reset, dim3(cdiv(longest, 256)), dim3(256), 0, s.stream, longest, s.halo_tab[0], s.halo_tab[1],
                           s.halo_tab[2], s.halo_tab[3], s.halo_tab[4], s.halo_tab[5], s.halo_tab[6], s.halo_tab[7]);
    HaloTables t;
    for (int k = 0; k < 8; ++k) t.tab[k] = s.halo_tab[k];
    hipLaunchKernelGGL(k_halo_tables, dim3(cdiv(s.c.n, 256)), dim3(256), 0, s.stream, s.c.n, s.xidx[s.xcur], t);
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
    hipMemsetAsync(hist, 0, sizeof(int) * (size_t)s.c.nz_glob, s.stream);
    if (s.c.n > 0) hipLaunchKernelGGL(k_layer_hist, dim3(cdiv(s.c.n, 256)), dim3(256), 0, s.stream, s.c, s.c.n, s.posv.cur(), s.meta.cur(), hist);
}

static void l_count_ghosts(State &s, int *out) {
    hipMemsetAsync(out, 0, sizeof(int), s.stream);
    if (s.c.n > 0) hipLaunchKernelGGL(k_count_ghosts, dim3(cdiv(s.c.n, 256)), dim3(256), 0, s.stream, s.c.n, s.meta.cur(), out);
}
