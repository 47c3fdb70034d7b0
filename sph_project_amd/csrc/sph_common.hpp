// sph_common.hpp -- host/device shared declarations of libsph_hip (gfx950 only).
//
// Data layout in HBM (all arrays have capacity particle_max_num, index = slot in the current
// cell-sorted order; cells linearised z-fastest as in the reference, lin = (cx*ny + cy)*nz + cz, so the
// 27 neighbour cells of a particle are 9 contiguous particle runs):
//   posv  float4  (x, y, z, rest volume V)       double-buffered (sort scatter / integration)
//   velm  float4  (vx, vy, vz, mass m)           double-buffered (sort scatter / v* update)
//   meta  int32   object_id+1 | material<<8 | is_dynamic<<10      } moved by the sort together with
//   pid   int32   insertion index (persistent particle id)        } color (r|g<<8|b<<16), rho and
//   rho   float   particle_densities                              } the rigid original positions
// Per-step scratch that the reference does not reorder either (base_container.py:506 list):
//   rho_raw, prs, ptm (= p / rho^2), acc, DFSPH alpha/kappa/..., PCISPH predicted state, CG vectors.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SPH_NOBJ 20

// meta packing
#define META_OBJ(m) (((m) & 0xff) - 1)
#define META_MAT(m) (((m) >> 8) & 0x3)
#define META_DYN(m) (((m) >> 10) & 0x1)
#define META_PACK(obj, mat, dyn) ((((obj) + 1) & 0xff) | (((mat) & 0x3) << 8) | (((dyn) & 1) << 10))
#define META_GHOST(m) (((m) >> 11) & 0x1)   /* slab sharding: copy of a neighbour rank's boundary particle */
#define META_DEAD(m) (((m) >> 12) & 0x1)    /* slab sharding: left this rank (or last step's ghost); sorted into the graveyard cell G */
#define META_FRESH(m) (((m) >> 13) & 0x1)   /* rigid particle appended after prepare(): the reference's end-of-step volume pass saw it on a stale grid */
#define META_FRESH_BIT (1 << 13)
#define META_ACTIVE_FLUID(m) ((((m) >> 8) & 0xB) == 1) /* material fluid and not a ghost */
#define META_SET_MAT(m, mat) (((m) & ~(0x3 << 8)) | (((mat) & 0x3) << 8))

struct Consts {
    int nx, ny, nz, G;      // grid the cell lists are built on (slab sharding: nx = own layers + ghost layers, see cx_off)
    // Slab sharding cuts the grid along ONE library axis (slab_axis) and builds the cell lists on the own layers + ghost layers only:
    //   slab_axis 2 (default): z, the FASTEST axis of the cell order lin = (cx ny + cy) nz + cz.  The three runs of an x-offset group of a
    //     thin slab overlap and are staged once (nbr_plan "chain"): at 12-24 layers per rank a step is 20-50 % cheaper than with
    //   slab_axis 0: x, the SLOWEST axis (the scene's z is mapped onto it at the C-ABI boundary, sph_api.hip set_axis_order): a rank's
    //     ghost / boundary / interior layers are contiguous index ranges, so boundary tiles can run first and interior ones while the halo
    //     flies -- built and measured in round 4 (profiles/r04_slab_layouts.txt): the split costs a workgroup lifetime per pass and the
    //     thin-slab staging advantage is lost; kept behind SPH_SLAB_LAYOUT=slow.
    // n?_glob = layers of the whole scene along the axis, c?_off = global layer of local layer 0 (0 for the axis that is not cut).
    int nx_glob, cx_off, nz_glob, cz_off, slab_axis;
    float grid_size;        // f32(dh): cell size
    float h, h2, inv_h;     // support radius, squared, reciprocal (fast build)
    float kW, kG;           // cubic spline constants (base_solver.py:57, :81)
    float kGh, Wd_poly;     // fast build: kG / h; kernel_W(particle diameter) / kW (the polynomial alone)
    float inv_h2;           // fast build: 1 / h^2
    float W0, Wd;           // kernel_W(0), kernel_W(particle diameter)
    float diameter2;
    float dt, inv_dt, rho0, inv_rho0, g_upper;
    int up_axis;            // library axis that holds the scene's y (gravitationUpper is a height: base_solver.py:18-23); 1 unless the axes are permuted
    float gx, gy, gz;
    float st;               // surface tension coefficient (0.01)
    float cv, cvb, visc_eps;// 2*(dim+2)*viscosity, ..*viscosity_b, 0.01*dh^2
    float pad, hix, hiy, hiz; // domain clamp
    float thr_kappa;        // DFSPH m_eps * dt
    float V0;
    float pcisph_k;
    int   n;                // particle_num
    int   all_fluid;        // no rigid / emitter particles in the container
    int   force_global;
    int   ghosts;           // slab sharding: ghost particles present (meta bit 11) even when all_fluid
    int   stat_bank;        // DevScalars bank the running step counts into (step parity)
    int   xcd_chunk;        // tile order of the neighbour passes (sph_device.hpp xcd_remap): 0 one contiguous eighth per XCD, C > 0 chunks of C tiles dealt round-robin
    int   run_grouping;     // which candidate runs form a staging group of k_nbr_pass (sph_device.hpp run_of): 0 x offsets, 1 outer runs mixed
    // slab sharding, device-resident counts (sph_halo.hpp SlabDyn): when set, the particle count lives in device memory
    // (the halo exchange changes it without the host looking) and `n` above is only the launch bound the grid was sized for
    const int *n_dev;
};

// particle_num as the kernels see it
__device__ __forceinline__ int live_n(const Consts &c) { return c.n_dev ? *c.n_dev : c.n; }

// z-slab sharding: per-step counts, kept on the device by the halo kernels (sph_halo.hpp) and mirrored into pinned host memory.
// status bits: see SLAB_ST_* below.
// Slab sharding files the particles a rank drops (last step's ghosts, migrants) behind its live particles: SPH_NGRAVE graveyard cells
// G .. G + SPH_NGRAVE - 1 behind the grid.  More than one because every run of dead lanes costs an atomic on its graveyard cell's
// counter, the dead sit in short runs all over the sorted order (the ghost cells at both ends of every z column), and atomics on ONE
// address are served one at a time: a single graveyard cell took the hash kernel from 8 to 50 us with two ranks (profiles/r04_two_ranks_*).
#define SPH_NGRAVE 64
// the three counters of a step message (records for the lower / upper rank, particles that die) sit in separate 128-byte lines of L2 and
// a bank (message parity) is HC_BANK ints: thousands of waves add to them per step, and atomics on one line are served one at a time
#define HC(k) ((k) * 32)
#define HC_BANK 128
struct SlabDyn {
    int n_app;       // particles while the step's sort runs: last step's + arrivals (dead ones still in)
    int n_live;      // after the sort: the dead ones (last step's ghosts, migrants that left) dropped
    int n_send[2];   // records sent to the lower / upper rank this step (migrants + boundary copies)
    int n_recv[2];   // records received from them
    int dropped;
    int longest;     // longest of the four messages (slot tables are reset up to here)
    int status;      // sticky: SLAB_ST_*
    unsigned seq;    // number of the step message these counts belong to
    int n_btiles;    // pinned mirror only: boundary tiles of the last sort (k_block_prep), sizes the boundary launches of the next steps
    unsigned wseq;   // pinned mirror only: 2 seq + 1 while the settle kernel writes the fields, 2 seq + 2 when they are complete (seqlock)
};
// Header of one step message.  TWO per inbox side, by message parity (round 5): the writer announces message m + 1 as soon as its own
// step is through -- it only needs the reader's message m for that, which the reader announced BEFORE consuming the writer's m.  With one
// header per side, a consuming kernel that is held up (a late workgroup; a process time-sliced off the GPU with 8 ranks on one device)
// met the number m + 1 and took m + 1's record count for the payload of m: records lost, or garbage appended
// (tests/test_hip_slab.py::test_c4_sharded_over_8_ranks..., 2 failures in ~35 runs, only inside the full suite).  The header of m is now
// overwritten by m + 2 only, which the lockstep argument of sph_halo.hpp already rules out while m is being consumed.
struct HaloRecHdr {
    unsigned seq;           // number of the message this header belongs to (stored last, system-scope release)
    int count, status, stride;   // records, the sender's status bits, float4 per record
};
struct HaloCtl {            // push transport: control block of one side of a rank's inbox, 256 bytes, written by that neighbour
    HaloRecHdr rec[2];      // step messages: header of message m in rec[m & 1]
    unsigned fld_seq;       // number of the last complete field message (its sizes are known to both sides: no header)
    int pad[55];
};
static_assert(sizeof(HaloCtl) == 256, "HaloCtl layout");
#define SLAB_ST_SEND_OVERFLOW 1   /* a face message of mine exceeds the message capacity */
#define SLAB_ST_PEER 2            /* a neighbour reported a failure in its message header */
#define SLAB_ST_STRIDE 4          /* the neighbour's record size differs from mine (dynamic rigid body not registered on every rank) */
#define SLAB_ST_CAPACITY 8        /* own + received particles exceed particle_max_num */
#define SLAB_ST_TIMEOUT 16        /* a neighbour's message did not arrive in time */
#define SLAB_ST_BOUND 32          /* particle count outran the launch bound of an asynchronous step */

// Device-side scalar block (zeroed / read back by the host)
#define SPH_STAT_SLOTS 2048  // statistics are striped over many words: ~20k same-address atomics per launch cost >200 us
struct DevScalars {
    // two banks: step k counts into bank k & 1 while its scan kernel clears the other one for step k + 1
    unsigned long long pairs[2][SPH_STAT_SLOTS];     // accepted pairs of a step weighted by the reference passes a walk stands for (sum over slots)
    unsigned long long evals[2][SPH_STAT_SLOTS];     // accepted pairs as evaluated (one per neighbour walk); separate 64-bit words: no carry between the two tallies
    unsigned long long fallback[2][SPH_STAT_SLOTS];  // neighbour runs that did not fit the LDS tile
    // rigid_body_forces, rigid_body_torques (base_container.py:161-162) as 64-bit FIXED POINT, 2^-32 N (N m) per unit: integer atomics
    // commute, so the accumulated wrench is bit-reproducible from run to run whatever order the workgroups finish in (the reference's f32
    // atomics -- and this code until round 3 -- are not); range +-2.1e9, resolution 2.3e-10.  Written by add_wrench (sph_passes.hpp).
    long long wrench[2 * SPH_NOBJ * 3];
#define SPH_WRENCH_SCALE 4294967296.0
    float red[8];                    // reduction results (errors, CG dots)
    int   flags[4];
};

struct RigidPose {  // rigid_body_* fields of the container (base_container.py:156-164)
    float com[SPH_NOBJ][3], com0[SPH_NOBJ][3], rot[SPH_NOBJ][9], vel[SPH_NOBJ][3], angvel[SPH_NOBJ][3];
    int   is_dynamic[SPH_NOBJ], material[SPH_NOBJ];
};

// slab sharding: what a pass needs to classify its own particles and store their records into the neighbours' inboxes (sph_halo_defs.hpp)
struct HaloSend {
    int on;                       // 0: the pass leaves the classification to k_halo_classify
    int z_lo, z_hi, has_down, has_up, cap, rs;
    float4 *dst[2];               // step-message regions of the neighbours' inboxes for the NEXT message
    int *counts;                  // its count bank: [0..1] records per side, [2] particles that die
    int *meta_w, *xidx;           // the particle's meta word (rewritten) and exchange tag
    const int *pid; const unsigned *color; const float4 *orig;
};

// What the WCSPH force pass needs to be the NEXT step's k_hash_count as well (round 5): it knows every particle's new position when it
// stores it, so it files the new cell id, takes the histogram atomic (one per run of equal cells among the tile's particles in sorted
// order, like k_hash_count) and leaves the arrival rank.  on = 0: the next step hashes as usual.
// Run lists of the deterministic sort (round 6, fourth session).  The lanes of a hashing wave are a few RUNS of particles that share a cell
// (wave_runs); whoever takes a run's histogram atomic also links the run into a per-cell list: one 64-bit exchange on head[cell] (sort epoch in
// the high word, so the heads are never reset: a head of another epoch is an empty list) and one record rec[first particle] = (previous head's
// first particle or -1, run length).  The sort then ranks every particle from its cell's list (k_sort_rank) -- the launch that used to file the
// records once the scan had said where (k_scatter_index) is gone.  The run that ARRIVES FIRST in its cell (its histogram atomic returned 0:
// at rest every run, in motion most) takes no second atomic: it leaves (first particle, length) in first[cell] by a plain store -- exactly one
// run per cell and sort does -- and only later arrivals chain themselves behind head[cell].  first[] needs no epoch: it is read for cells
// that hold several runs of this sort only, and such a cell has had its first arrival.  head == nullptr: off (slab sharding).
// ONE array holds both kinds of record -- rec[i] for i < first_off (the particle capacity), first[cell] = rec[first_off + cell] -- so that a
// hasher's store goes through one base pointer and an index chosen in registers (with two pointers in the kernel arguments the compiler
// fetched the chosen one by a dependent VECTOR load from the argument segment: one more round trip at the end of every force-pass workgroup).
struct RunList { unsigned long long *head; int2 *rec; int first_off; unsigned epoch; };
struct NextHash { int on; int *cellid, *rank, *cell_count, *tile_sum; RunList rl; };

// ... and what the WCSPH density pass needs to store (rho_raw, rho, p, p / rho^2) of its boundary particles straight into the field
// message of the neighbours' inboxes (the message k_halo_pack2<2> would gather afterwards): out[side] = that message's region, xidx = the
// exchange tags of this sort, dyn = this step's counts (the echo part of a message starts behind its n_send records)
struct HaloFieldSend { int on; float4 *out[2]; const int *xidx; const SlabDyn *dyn; };

template <class T> struct DBuf {
    T *b[2] = {nullptr, nullptr};
    int c = 0;
    T *cur() const { return b[c]; }
    T *alt() const { return b[1 - c]; }
    void flip() { c = 1 - c; }
};

// Everything a launcher needs.  Owned by SphHandle.
struct State {
    Consts c;
    hipStream_t stream;
    int cap;             // particle_max_num
    // sorted, carried state
    DBuf<float4> posv, velm, orig;
    DBuf<int> meta, pid;
    DBuf<unsigned> color;
    DBuf<float> rho;
    // grid
    int *cell_count;     // G+1
    int *cell_start;     // G+1 (exclusive scan, cell_start[G] = n)
    int *cellid, *rank;  // per particle (pre-sort)
    // deterministic sort by run lists (RunList above; unsharded scenes): list heads per cell (G + 1, never reset), the epoch of the histogram
    // being taken / last taken, "the hashers of the coming sort filed their runs", and the sort's gather index (dest slot -> source particle)
    unsigned long long *run_head;
    int2 *run_rec;       // cap + G + 1 records (RunList::rec)
    unsigned sort_epoch;
    int run_lists_filed;
    int *sort_inv;
    long long n_list_sorts;   // sorts that went through k_sort_rank + k_gather_prep, since create (SphStats)
    // Bytes the sort's gather does not have to move (k_gather_prep, unsharded list sorts only):
    //  * sort_skip_rho: set by a step whose next kernel recomputes every particle's density (WCSPH step of an all-fluid scene: the density
    //    pass writes rho for all particles straight after the sort) -- consumed by the next sort;
    //  * colours stay AT HOME: color_home[particle id] is written when a particle is appended and never moves, while the ids are the append
    //    order (color_home_ok; ids set from outside or slab sharding end that).  color_stale: the sorted copy (State::color) is out of date --
    //    whoever needs it (download, a sort by run records, slab activation) calls Launch::ensure_color first.
    int sort_skip_rho;
    unsigned *color_home;
    int color_home_ok, color_stale;
    int *tmp_idx;        // stable-sort scratch: int2 (first source index, length) per run, filed at the run's first slot (2 x cap ints)
    // Sums of the histogram over the scan's tiles of SCAN_TILE cells, two banks of scan_blocks + 1 ints.  Round 6: whoever takes the
    // histogram atomics (k_hash_count, the NextHash epilogue of the force pass, the slab kernels) adds its particles to the tile sums of
    // bank `scan_bank` as well -- one more atomic per WAVE, its lanes' cells lie in one tile nearly always -- so the scan is ONE launch:
    // k_scan_final reads that bank and clears the other one for the next histogram (like cell_count, which it clears behind itself).
    int *scan_partial;
    int scan_blocks;
    int *scan_tile_state;   // per scan tile: (particles before it) + 1 if its cells hold that constant because it was EMPTY at the last scan, else 0 (k_scan_final)
    int scan_bank;       // bank the hashers of the coming sort add to / the coming scan reads
    int skip_residual;   // fixed-iteration solves (no stop test, nobody reads the residual): the walks leave their partial sums, k_reduce_partials is not launched
    int tile_sums_ready; // ... and they did (else l_scan launches k_scan_reduce first: SPH_NO_SCAN_FOLD)
    int cell_count_clean;              // cell_count is all zero (the scan clears it behind itself)
    int hist_taken;                    // the histogram of the coming sort has been taken (by k_hash_count, a NextHash epilogue, the slab kernels) and not yet scanned
    int state_error;                   // sticky: a launcher met a state it must never see (l_scan without a histogram: ADVICE r05); reported by check_async
    // per-step scratch
    float *rho_raw, *prs, *ptm;
    float4 *acc;
    // DFSPH
    float *alpha, *kappa, *kappa_v, *rho_star, *rho_deriv;
    float *kappa_next, *kappa_v_next;  // written by the rho_adv pass, consumed (after a swap) by the next correction
    float2 *kr;          // (kappa, rho) staging pair for the correction pass
    // PCISPH
    float4 *pacc, *pvel, *ppos, *acc_np;
    // CG (implicit viscosity)
    float4 *cg_p, *cg_Ap, *cg_x, *cg_b, *cg_r, *cg_v0;
    float4 *cg_p2;       // second search-direction buffer (fused p update: the A p pass reads the old one and writes the new one)
    int cg_fuse;         // the next A p pass applies the previous iteration's p update on the fly (CgApPass::fuse)
    int cg_fused_loop;   // this solve runs the two-launch iteration (A p [+ combine], x / r update)
    float *cg_dinv;      // 9 floats per particle
    float4 *cg_part;     // 3 x cap: per-group parts of A p when the pass is split (CgApPass::SPLIT3)
    int cg_nocombine;    // the last A p pass was split and left its dot-product shares itself: the x / r update adds the parts up (no k_cg_ap_combine)
    int cg_split;        // this solve splits its A p passes (few fluid particles: see implicit_viscosity_non_pressure)
    int split_next_pass; // launch_pass: launch the next SPLIT3 functor with gridDim.y = 3
    int cg_parity;       // which of the two |r|^2 partial arrays the next x / r update reads
    // reductions
    float *red_partial;  // per-block partial sums
    int red_blocks;
    DevScalars *scal;    // device
    RigidPose *pose;     // device copy
    int has_dynamic_rigid;
    int has_rigid;
    // z-slab sharding (sph_halo.hpp): exchange tables and message buffers
    int *xidx[2];        // carried through the sort: (kind << 29) | index, see HALO_KIND_*
    int xcur;
    int *halo_tab[8];    // slot tables, index = kind - 1 (sph_halo.hpp HALO_*): send, ghost, echo-send, echo-ghost x {down, up}
    float4 *sendbuf[2], *recvbuf[2];     // 3 float4 per particle record
    int *halo_counts;    // device, two banks of 4 (push transport: bank = message parity): [0..1] send counts, [2] dropped
    SlabDyn *dyn;        // device-resident counts of a step, two banks (message parity): a kernel reads the last step's, writes this step's
    SlabDyn *dyn_cur;    // the bank of the last step message (what c.n_dev and the field kernels look at)
    int tables_pending;  // push transport: the next k_block_prep also fills the halo slot tables from xidx
    int async_counts;    // the running step takes its counts from `dyn` (c.n_dev set, c.n = launch bound)
    // push transport (sph_halo.hpp): my inbox, the neighbours' inboxes mapped through hipIpc, message numbers
    struct PushState {
        int on;                        // halo payload travels by device stores into the neighbour's inbox
        char *inbox;                   // mine (device memory the neighbours write)
        char *peer[2];                 // lower / upper neighbour's inbox in my address space (null: no neighbour)
        int rec_cap;                   // records a step message can hold (normally the particle capacity)
        size_t rec_bytes, fld_bytes;   // size of one step-message / field-message region
        unsigned rec_seq, fld_seq;     // messages sent (= received) so far
        long long timeout_ticks;       // bounded waits of the device (100 MHz wall clock)
        SlabDyn *mirror;               // pinned host copy of `dyn`, written by the wait kernel
    } push;
    // slab sharding, compute / halo overlap: the layers of a slab are contiguous index ranges of the sorted arrays (x is the slowest axis of
    // the cell order), so the 256-particle tiles fall into a BOUNDARY set (anything within two own layers of a face, plus the ghost layers)
    // and an INTERIOR set; k_block_prep lists both once per sort.  A pass launched with tile_sel = 1 / 2 walks one set only: boundary
    // tiles first, their values go out, interior tiles run while the message is in flight.
    int *tile_list[2];   // [0] boundary tiles, [1] interior tiles (XCD-remapped order), device
    int *tile_cnt;       // their lengths (2 ints, device)
    unsigned char *tile_class;   // per tile: 1 = boundary set.  The INTERIOR launch is an ordinary launch of every tile (no list: a list costs each
                                 // workgroup one more dependent load in front of its header) whose boundary tiles leave at once.
    int tile_sel;        // 0: every tile (default); 1: boundary set; 2: interior set -- read by launch_pass
    int tile_plan_n;     // particle count (launch bound) the lists were built for, -1: none
    int tile_bound_b;    // workgroups a boundary launch gets: last known number of boundary tiles + margin (checked on the device: SLAB_ST_BOUND)
    HaloSend presend;    // on != 0: the next wcsph_forces launch classifies its particles and sends the next step message itself
    HaloFieldSend fieldsend;   // on != 0: the next WCSPH density launch stores its boundary values into the field message itself
    int preclassified;   // ... and has done so: the next step's exchange starts with the hash alone
    NextHash nexthash;   // on != 0: the next wcsph_forces launch hashes the particles for the next step's sort itself
    int prehashed;       // ... and has done so: the next step's sort starts with the scan (l_hash_count returns at once)
    long long n_hash_launches, n_prehashed_sorts;   // k_hash_count launches / sorts that started from a NextHash hash, since create (SphStats)
    int halo_cap;        // particles per message buffer
    int halo_longest;    // longest halo message of the running step, in particles (sent or received)
    int slab_active, z_lo, z_hi, has_down, has_up;
    unsigned *nbr_mask;  // [9][cap]: acceptance mask of the first 32 candidates of every (particle, run), see process_run
    unsigned *nbr_mask_hi;  // [9][cap]: candidates 32..63 of the runs that have them
    int masks_valid;
    int density_books_forces;   // WCSPH step with the fused force pass: the density pass counts that pass's pairs too (it walks the same masks)
    int uniform_mass;    // all-fluid scene whose particles all carry the same mass (sph_api.hip refresh_counts): WcsphForcePass<true, true>
    // device-controlled solver loops (sph_steps.hpp device_loop): every kernel of an iteration starts with a look at
    // scal->flags[0] (stop) when loop_flag is set; the iteration's reduction kernel evaluates the stop criterion itself
    const int *loop_flag;    // null outside such a loop
    int loop_slot;           // reduction slot the criterion applies to
    int loop_kind;           // 1: (double)(sum / denom) <= thr   2: (sum / denom) < (float)thr   3: CG, !(err > thr)
    float loop_denom; double loop_thr;
    unsigned char *lane_perm;  // [ceil(cap / 256) * 256]: lane -> particle map of every 256-particle workgroup (k_lane_perm)
    int *blk_flag, *blk_list, *blk_count;   // per-workgroup 'holds fluid' flag, ascending list of those workgroups, its length (device)
    int list_n;                // particle count the list was built for (-1: none)
    // Length of the list as the HOST knows it: k_compact_blocks stores it into pinned memory, an event behind it tells when it has landed.
    // A launch over the list then gets exactly that many workgroups (list_grid, sph_kernels.hip) instead of one per tile of the scene
    // with all but the listed ones leaving at once: the buckling scene has 8,496 tiles of which ~420 hold fluid, and the 24,000 empty
    // workgroups of a three-way split A p launch cost 16 % of the walk (round 5: C5 1.63 -> 1.47 ms/step).
    volatile int *list_count_pinned;
    hipEvent_t list_count_event;
    int list_count_known;      // -1: not seen yet (the launches use the full grid until the event has passed)
    int last_pass_listed;      // the last pass with a reduction ran the listed workgroups only: so must the sum of its partials
    int *blk_hdr;              // [ceil(cap / 256)][BLK_HDR_INTS]: cell span and candidate-run windows of every workgroup
    int perm_n;                // particle count blk_hdr / lane_perm were built for (-1: none)
    int has_emitter;     // gravitationUpper set (base_solver.py:18-23)
    int visc_rho_raw;    // viscosity reads rho_raw (WCSPH: density before the EOS clamp)
    int skip_viscosity;  // implicit viscosity: explicit term not part of the fused pass
    float4 *np_acc_out;  // non-pressure acceleration sink (PCISPH), else null
    const float4 *np_visc_vel; // velocities for the viscous term (implicit viscosity: cg_x), else null
};

// Function table implemented twice (strict / fast math), see sph_kernels.hip.
struct Launch {
    void (*ensure_color)(State &);   // materialise State::color in sorted order from color_home (no-op unless color_stale)
    void (*hash_count)(State &);
    void (*scan)(State &);
    void (*scatter)(State &);
    void (*scatter_stable)(State &);
    void (*density)(State &, int eos);
    void (*non_pressure)(State &);
    void (*pressure_integrate)(State &);
    void (*wcsph_forces)(State &);              // non_pressure + pressure_integrate in one neighbour walk (WCSPH)
    void (*rigid_volume)(State &);
    void (*renew_rigid)(State &);
    void (*prepare_emitter)(State &);
    void (*post_insert)(State &, int first, int stale_volume);   // appended particles [first, n): boundary (+ V = 1 / W(0) for rigid ones)
    void (*clear_fresh)(State &);
    // DFSPH
    void (*dfsph_density_alpha)(State &);
    void (*dfsph_density_alpha_div)(State &);   // + the first density derivative of the divergence solve (one neighbour walk less per step)
    void (*dfsph_rho_adv)(State &, int mode);   // 0: density derivative (+kappa_v), 1: density star (+kappa)
    void (*dfsph_correct)(State &, int mode);   // 0: divergence step, 1: density step
    void (*advect_boundary)(State &);           // x += dt v, emitter, boundary (DFSPH position update)
    // PCISPH
    void (*pcisph_init)(State &);
    void (*pcisph_rho_star)(State &);
    void (*pcisph_pressure_accel)(State &);
    // z-slab sharding
    void (*halo_classify_pack)(State &, int n);
    void (*halo_unpack_append)(State &, int side, int count, int offset);
    void (*halo_build_tables)(State &);
    void (*halo_pack_fields)(State &, int side, int n_send, int n_recv);
    void (*halo_unpack_fields)(State &, int side, int n_recv, int n_send);
    void (*halo_pack_scalar)(State &, int side, int n_send, int n_recv, const float *src);
    void (*halo_unpack_scalar)(State &, int side, int n_recv, int n_send, float *dst);
    void (*halo_pack_vel)(State &, int side, int n_send, int n_recv, const float4 *arr);   // xyz of a float4 array
    void (*halo_unpack_vel)(State &, int side, int n_recv, int n_send, float4 *arr);
    void (*count_ghosts)(State &, int *out);
    void (*layer_hist)(State &, int *hist);      // owned particles per global cell layer
    void (*loop_criterion)(State &, int slot);   // stop test on an all-reduced residual (sharded solver loops)
    // push transport: the step message is written by halo_classify_pack itself; then
    // announce my message, wait for both neighbours', settle SlabDyn, append their records, reset the slot tables
    void (*halo_unpack2)(State &, int n_old, int bound_app, int bound_live, int count_hint);
    void (*halo_push_fields)(State &, int kind, float *f0, float4 *v, int count_hint);   // pack a field message per neighbour
    void (*halo_pull_fields)(State &, int kind, float *f0, float4 *v, int count_hint);   // announce, wait, scatter the neighbours' field messages
    void (*halo_selftest)(State &, int n, int tag, int tag_down, int tag_up, int *bad_dev);
    void (*halo_fieldsend_begin)(State &);   // begin a field message of kind 2: the density pass launched next fills it (State::fieldsend)
    void (*halo_presend_begin)(State &);   // begin the next step message: the force pass launched next fills it (State::presend)
    // implicit viscosity (CG)
    void (*cg_prepare)(State &);
    void (*cg_ap)(State &);
    void (*cg_prepare2)(State &);
    void (*cg_alpha)(State &);
    void (*cg_check)(State &);   // fused p update: stop test at the end of a batch of iterations
    void (*cg_update_xr)(State &);
    void (*cg_update_p)(State &);
    void (*cg_prepare_guess)(State &);
    void (*cg_fold)(State &, int which);   // slab sharding: local sums of the dot-product partials -> scal->red[6..7]
};

const Launch *sph_launch_strict();
const Launch *sph_launch_fast();
