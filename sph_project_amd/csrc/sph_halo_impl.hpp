// sph_halo_impl.hpp -- launchers of the slab-sharding kernels (included inside the per-build namespace)
#pragma once

// reads the current arrays, writes the kept particles to the alt buffers (then flips) and the messages
static void l_halo_classify_pack(State &s, int n) {
    hipMemsetAsync(s.halo_counts, 0, 4 * sizeof(int), s.stream);
    for (int k = 0; k < 8; ++k) hipMemsetAsync(s.halo_tab[k], 0xff, sizeof(int) * (size_t)s.halo_cap, s.stream);
    if (n > 0) {
        HaloArrays a{s.posv.cur(), s.velm.cur(), s.meta.cur(), s.pid.cur(), s.color.cur(), s.rho.cur(),
                     s.posv.alt(), s.velm.alt(), s.meta.alt(), s.pid.alt(), s.color.alt(), s.rho.alt(), s.xidx[1 - s.xcur]};
        hipLaunchKernelGGL(k_halo_classify, dim3(cdiv(n, 256)), dim3(256), 0, s.stream, s.c, n, s.z_lo, s.z_hi, s.has_down,
                           s.has_up, a, s.sendbuf[0], s.sendbuf[1], s.halo_cap, s.halo_counts);
    }
    s.posv.flip(); s.velm.flip(); s.meta.flip(); s.pid.flip(); s.color.flip(); s.rho.flip();
    s.xcur = 1 - s.xcur;
}

static void l_halo_unpack_append(State &s, int side, int count, int offset) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_halo_unpack, dim3(cdiv(count, 256)), dim3(256), 0, s.stream, s.c, count, offset, side, s.z_lo, s.z_hi,
                       s.recvbuf[side], s.posv.cur(), s.velm.cur(), s.meta.cur(), s.pid.cur(), s.color.cur(), s.rho.cur(),
                       s.xidx[s.xcur]);
}

static void l_halo_build_tables(State &s) {
    if (s.c.n == 0) return;
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
