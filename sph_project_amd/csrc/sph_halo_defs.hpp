// sph_halo_defs.hpp -- slab sharding: record kinds, the classification rule of the step message, and its fused form (included inside
// the per-build namespace, BEFORE the passes: the WCSPH force pass can classify its own particles, see HaloSend).
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


// What happens to particle i at the start of the next step (SURVEY 8e message (1) + (2)), decided from its NEW position -- shared by
// k_halo_classify (sph_halo.hpp) and the fused form below.  side: -1 no record, 0 / 1 record for the lower / upper rank; mrec: the meta
// word the record carries; mnew: the meta word the particle keeps; dead: it is dropped here (last step's ghost, or a migrant that went
// further than the neighbour's boundary layer).
struct HaloVerdict { int side, dead, mnew, mrec; };
__device__ __forceinline__ HaloVerdict halo_classify_one(int m, int layer, int z_lo, int z_hi, int has_down, int has_up) {
    HaloVerdict v{-1, 0, m, 0};
    if (META_GHOST(m) || META_DEAD(m)) { v.dead = 1; return v; }   // last step's ghosts are re-sent by their owners
    if (layer < z_lo && has_down) {           // left through the lower face: ownership moves down
        v.side = 0; v.mrec = META_SET_GHOST(m, 0);
        if (layer == z_lo - 1) v.mnew = META_SET_GHOST(m, 1); else v.dead = 1;   // kept as "echo ghost" / gone
    } else if (layer >= z_hi && has_up) {
        v.side = 1; v.mrec = META_SET_GHOST(m, 0);
        if (layer == z_hi) v.mnew = META_SET_GHOST(m, 1); else v.dead = 1;
    } else if (layer == z_lo && has_down) { v.side = 0; v.mrec = META_SET_GHOST(m, 1); }
    else if (layer == z_hi - 1 && has_up) { v.side = 1; v.mrec = META_SET_GHOST(m, 1); }
    return v;
}

// Fused step message (round 4): the pass that produces the new positions -- the fused WCSPH force pass, the last kernel of a step --
// classifies each particle right where its new position is known and stores its record straight into the neighbour's inbox.  The
// payload of step k + 1's message then travels over the link WHILE the force pass of step k is still running (the records leave as the
// workgroups finish), instead of in a kernel of its own at the start of step k + 1 with the link busy and the CUs idle and the consumer
// waiting right behind it.  What is left for the next step: the hash (k_hash_count) and the header / wait / append kernel as before.
// Only inside sph_step_async(n) when the next step follows at once (nothing on the host looks at the half-classified state), without
// emitter or rigid bodies (they move particles after the force pass) and not before a step that re-plans the cuts.
// (struct HaloSend: sph_common.hpp -- State carries the one the next force pass will use)
__device__ __forceinline__ void halo_presend(const Consts &c, const HaloSend &hs, int i, const float4 &p, const float4 &v, float rho) {
    const int m = hs.meta_w[i];
    const HaloVerdict vd = halo_classify_one(m, slab_layer(c, p), hs.z_lo, hs.z_hi, hs.has_down, hs.has_up);
    const int k0 = halo_wave_slot(vd.side == 0, &hs.counts[HC(0)]);
    const int k1 = halo_wave_slot(vd.side == 1, &hs.counts[HC(1)]);
    halo_wave_slot(vd.dead != 0, &hs.counts[HC(2)]);
    int xi = 0, mnew = vd.mnew;
    if (vd.side >= 0) {
        const int k = vd.side == 0 ? k0 : k1;
        if (k < hs.cap) {
            float4 *buf = hs.dst[vd.side];
            halo_write_record(buf, hs.rs, k, p, v, vd.mrec, hs.pid[i], hs.color[i], rho);
            if (hs.orig) buf[hs.rs * k + 3] = hs.orig[i];
        }
        xi = HALO_PACK((!META_GHOST(vd.mrec) ? HALO_ECHO_GHOST : HALO_SEND) + vd.side, k);
    }
    if (vd.dead) { mnew |= 1 << 12; xi = 0; }
    hs.meta_w[i] = mnew;
    hs.xidx[i] = xi;
}
