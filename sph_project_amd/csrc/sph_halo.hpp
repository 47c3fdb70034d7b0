// sph_halo.hpp -- z-slab sharding, device side (included inside the per-build namespace).
//
// Rank r owns the global cell layers cz in [z_lo, z_hi) (>= 2 layers) and keeps one ghost layer per interior
// side.  Once per step, right before the sort, ONE message per neighbour carries both kinds of
// records (48 B each: posv, velm, meta|pid|color|rho):
//   * migrants: particles that left the slab -> ownership moves (ghost flag 0);
//   * boundary copies: owned particles in cz == z_lo / z_hi-1 -> ghosts of the neighbour (ghost flag 1).
// A migrant that lands in the neighbour's boundary layer (it moved < 1 layer) is needed back as a ghost; both
// sides can tell that from the record itself, so the sender keeps it as a local ghost ("echo ghost", indexed by
// its position k in the sender's message) and the receiver tags it "echo send k": no second round.
// xidx = (kind << 28 | k) rides through the sort; k_halo_tables turns it into slot tables so that later field
// exchanges (density / pressure of the ghosts after the density pass) are gather -> message -> scatter with
// message sizes both sides already know.
#pragma once

#define HALO_SEND 1        // + side: owned boundary particle exported as ghost, k = index in my message
#define HALO_GHOST 3       // + side: ghost received, k = index in the neighbour's message
#define HALO_ECHO_SEND 5   // + side: migrant received into my boundary layer, k = index in the neighbour's message
#define HALO_ECHO_GHOST 7  // + side: my migrant kept as ghost, k = index in my message
#define HALO_PACK(kind, idx) (((kind) << 28) | (idx))
#define HALO_KIND(x) ((int)(((unsigned)(x)) >> 28))
#define HALO_IDX(x) ((x) & 0x0fffffff)
#define META_SET_GHOST(m, g) (((m) & ~(1 << 11)) | ((g) << 11))

struct HaloArrays {
    const float4 *posv_in, *velm_in; const int *meta_in, *pid_in; const unsigned *color_in; const float *rho_in;
    float4 *posv_out, *velm_out; int *meta_out, *pid_out; unsigned *color_out; float *rho_out; int *xidx_out;
};

__device__ __forceinline__ void halo_write_record(float4 *buf, int k, const float4 &p, const float4 &v, int meta,
                                                  int pid, unsigned color, float rho) {
    buf[3 * k] = p;
    buf[3 * k + 1] = v;
    buf[3 * k + 2] = make_float4(__int_as_float(meta), __int_as_float(pid), __uint_as_float(color), rho);
}

// counts[0] = records for the lower rank, counts[1] = upper rank, counts[2] = particles kept locally
__global__ void __launch_bounds__(256)
k_halo_classify(const Consts c, int n, int z_lo, int z_hi, int has_down, int has_up, HaloArrays a,
                float4 *send_down, float4 *send_up, int cap, int *counts) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int m = a.meta_in[i];
    if (META_GHOST(m)) return;  // last step's ghosts are re-sent by their owners
    const float4 p = a.posv_in[i];
    const float4 v = a.velm_in[i];
    const int pid = a.pid_in[i];
    const unsigned col = a.color_in[i];
    const float rho = a.rho_in[i];
    const int cz = cell_coord(p.z, c.grid_size, c.nz);
    int keep = 1, ghost_local = 0, xi = 0;
    if (cz < z_lo && has_down) {           // left through the lower face: ownership moves down
        const int k = atomicAdd(&counts[0], 1);
        if (k < cap) halo_write_record(send_down, k, p, v, META_SET_GHOST(m, 0), pid, col, rho);
        keep = cz == z_lo - 1; ghost_local = 1;
        xi = HALO_PACK(HALO_ECHO_GHOST + 0, k);
    } else if (cz >= z_hi && has_up) {
        const int k = atomicAdd(&counts[1], 1);
        if (k < cap) halo_write_record(send_up, k, p, v, META_SET_GHOST(m, 0), pid, col, rho);
        keep = cz == z_hi; ghost_local = 1;
        xi = HALO_PACK(HALO_ECHO_GHOST + 1, k);
    } else if (cz == z_lo && has_down) {
        const int k = atomicAdd(&counts[0], 1);
        if (k < cap) halo_write_record(send_down, k, p, v, META_SET_GHOST(m, 1), pid, col, rho);
        xi = HALO_PACK(HALO_SEND + 0, k);
    } else if (cz == z_hi - 1 && has_up) {
        const int k = atomicAdd(&counts[1], 1);
        if (k < cap) halo_write_record(send_up, k, p, v, META_SET_GHOST(m, 1), pid, col, rho);
        xi = HALO_PACK(HALO_SEND + 1, k);
    }
    if (keep) {
        const int d = atomicAdd(&counts[2], 1);
        a.posv_out[d] = p; a.velm_out[d] = v;
        a.meta_out[d] = META_SET_GHOST(m, ghost_local);
        a.pid_out[d] = pid; a.color_out[d] = col; a.rho_out[d] = rho;
        a.xidx_out[d] = xi;
    }
}

// appends `count` records received from `side` (0 = lower rank, 1 = upper rank) at [offset, offset + count)
__global__ void __launch_bounds__(256)
k_halo_unpack(const Consts c, int count, int offset, int side, int z_lo, int z_hi, const float4 *recv, float4 *posv,
              float4 *velm, int *meta, int *pid, unsigned *color, float *rho, int *xidx) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    const float4 p = recv[3 * k];
    const float4 q = recv[3 * k + 2];
    const int m = __float_as_int(q.x);
    const int d = offset + k;
    posv[d] = p; velm[d] = recv[3 * k + 1];
    meta[d] = m; pid[d] = __float_as_int(q.y); color[d] = __float_as_uint(q.z); rho[d] = q.w;
    int xi;
    if (META_GHOST(m)) xi = HALO_PACK(HALO_GHOST + side, k);
    else {  // a migrant, now owned here; echo it back if it sits in my boundary layer facing the sender
        const int cz = cell_coord(p.z, c.grid_size, c.nz);
        const int edge = side == 0 ? z_lo : z_hi - 1;
        xi = cz == edge ? HALO_PACK(HALO_ECHO_SEND + side, k) : 0;
    }
    xidx[d] = xi;
}

// after the sort: slot tables from the xidx that rode along.  tab[kind - 1] for kinds 1..8.
struct HaloTables { int *tab[8]; };
__global__ void __launch_bounds__(256) k_halo_tables(int n, const int *xidx, HaloTables t) {
    const int i = blockIdx.x * 256 + threadIdx.x;
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
