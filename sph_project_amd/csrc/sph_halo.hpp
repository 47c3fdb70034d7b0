// sph_halo.hpp -- z-slab sharding, device side (included inside the per-build namespace).
//
// Rank r owns the global cell layers cz in [z_lo, z_hi) (>= 2 layers) and keeps one ghost layer per interior
// side.  Once per step, right before the sort, ONE message per neighbour carries both kinds of
// records (48 B each: posv, velm, meta|pid|color|rho; 64 B with the rest position when the scene has a dynamic rigid body):
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
    const float4 *posv, *velm; int *meta; const int *pid; const unsigned *color; const float *rho; int *xidx;
    const float4 *orig;   // rigid_particle_original_positions, or null: then records are 3 float4 (rs = 3), else 4
};

__device__ __forceinline__ void halo_write_record(float4 *buf, int rs, int k, const float4 &p, const float4 &v, int meta,
                                                  int pid, unsigned color, float rho) {
    buf[rs * k] = p;
    buf[rs * k + 1] = v;
    buf[rs * k + 2] = make_float4(__int_as_float(meta), __int_as_float(pid), __uint_as_float(color), rho);
}

// one atomic per wave and counter: base index of this lane among the lanes with `want`
__device__ __forceinline__ int halo_wave_slot(bool want, int *counter) {
    const unsigned long long m = __ballot(want);
    if (!m) return 0;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(m));
    base = __shfl(base, leader, 64);
    return base + __popcll(m & ((1ull << lane) - 1ull));
}

// In place: nothing is compacted or copied.  Particles that are no longer this rank's business (last step's ghosts,
// migrants beyond the neighbour's boundary layer) get the DEAD bit; the sort that follows files them into the
// graveyard cell G behind every live particle, and the live count shrinks by counts[2].
// counts[0] = records for the lower rank, counts[1] = upper rank, counts[2] = particles that died.
__global__ void __launch_bounds__(256)
k_halo_classify(const Consts c, int n, int z_lo, int z_hi, int has_down, int has_up, HaloArrays a,
                float4 *send_down, float4 *send_up, int cap, int *counts) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    int side = -1, dead = 0, xi = 0, mnew = 0, mrec = 0;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
        const int m = a.meta[i];
        mnew = m;
        if (META_GHOST(m) || META_DEAD(m)) {   // last step's ghosts are re-sent by their owners
            dead = 1;
        } else {
            p = a.posv[i];
            const int cz = cell_coord(p.z, c.grid_size, c.nz_glob);   // global layer: compared with the slab bounds
            if (cz < z_lo && has_down) {           // left through the lower face: ownership moves down
                side = 0; mrec = META_SET_GHOST(m, 0);
                if (cz == z_lo - 1) mnew = META_SET_GHOST(m, 1); else dead = 1;   // kept as "echo ghost" / gone
            } else if (cz >= z_hi && has_up) {
                side = 1; mrec = META_SET_GHOST(m, 0);
                if (cz == z_hi) mnew = META_SET_GHOST(m, 1); else dead = 1;
            } else if (cz == z_lo && has_down) { side = 0; mrec = META_SET_GHOST(m, 1); }
            else if (cz == z_hi - 1 && has_up) { side = 1; mrec = META_SET_GHOST(m, 1); }
        }
    }
    const int k0 = halo_wave_slot(side == 0, &counts[0]);
    const int k1 = halo_wave_slot(side == 1, &counts[1]);
    halo_wave_slot(dead != 0, &counts[2]);
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
        const int cz = cell_coord(p.z, c.grid_size, c.nz_glob);
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

// Stop test of a device-controlled solver loop on a residual that was all-reduced over the ranks behind
// k_reduce_partials (which, under sharding, only leaves this rank's sum in scal->red[slot]).
__global__ void k_loop_criterion(DevScalars *scal, int slot, int kind, float denom, double thr) {
    if (scal->flags[0]) return;
    const float avg = denom > 0.0f ? scal->red[slot] / denom : 0.0f;
    scal->flags[1] += 1;
    if (kind == 1 ? ((double)avg <= thr) : (avg < (float)thr)) scal->flags[0] = 1;
}

// owned particles per global cell layer (slab rebalancing: the cuts follow the fluid)
__global__ void __launch_bounds__(256)
k_layer_hist(const Consts c, int n, const float4 *posv, const int *meta, int *hist) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int m = meta[i];
    if (META_GHOST(m) || META_DEAD(m)) return;
    atomicAdd(&hist[cell_coord(posv[i].z, c.grid_size, c.nz_glob)], 1);
}

// number of ghost copies among the first n particles (sph_comm_get_slab): one atomic per wave
__global__ void __launch_bounds__(256) k_count_ghosts(int n, const int *meta, int *out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const unsigned long long m = __ballot(i < n && META_GHOST(meta[i]));
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(out, __popcll(m));
}
