// sph_api.hip -- C-ABI of libsph_hip.so (include/sph_hip.h): device state, step orchestration,
// host<->device transfers, HIP-event profiler.  Host code only; kernels live in sph_kernels.hip.
#include "../../include/sph_hip.h"
#include "sph_common.hpp"
#include "sph_comm.hpp"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include <stdlib.h>
#include <utility>
#include <array>
#include "sph_voxel.hpp"
#include "sph_export.hpp"

static thread_local std::string g_create_error;

struct ProfSlot {
    bool on = false;
    int64_t launches = 0;
    double ms = 0.0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct SphHandle {
    SphParams prm;
    State st;
    const Launch *L = nullptr;
    std::string err;
    int device = 0;
    int n = 0;            // particle_num (a launch BOUND while n_exact is false: asynchronous slab steps, see slab_settle)
    bool n_exact = true;
    int n_fluid = 0;      // fluid_particle_num
    int n_nonfluid = 0;
    // every fluid particle appended so far carries the same mass (one fluid density: the usual scene), and nobody has uploaded masses:
    // the fused force pass then runs its uniform-mass instantiation (bit-identical sums, four VALU less per pair)
    bool fluid_mass_seen = false, fluid_mass_uniform = true; float fluid_mass0 = 0.0f;
    bool any_rigid_object = false;   // a non-fluid object was registered (slab sharding: its particles may live on another rank)
    int64_t steps = 0;
    int steps_to_follow = 0;       // sph_step_async(n): steps of this call still to come after the running one
    bool whole_step = false;       // both halves of the running step are ONE call (step_once): nothing on the host happens between them
    double total_time = 0.0;
    bool prepared = false;
    bool pose_dirty = false;
    int loop_hint[4] = {0, 0, 0, 0};   // iterations the last solve of each device-controlled loop took (sph_steps.hpp device_loop), by reduction slot
    struct LoopPub *loop_pub = nullptr;   // pinned: residual + flags of a solver loop's batch, published by a kernel (sph_steps.hpp k_publish_loop)
    unsigned loop_seq = 0, stats_seq = 0;
    bool loop_flags_clean = false;   // scal->flags[0..1] are zero (the publishing kernel of a stopped loop reset them): device_loop needs no memset
    int dev_cus = 256;           // compute units of the device (sizing of grids that should be resident at once)
    bool pose_given = false;     // sph_set_rigid_pose was called: pose_h holds library-frame vectors of the CURRENT axis order
    bool rigid_volume_done = false;
    bool sort_dirty = false;      // particles appended since the last sort
    bool in_step = false;         // between sph_step_begin and sph_step_end
    int n_mark = 0;               // particle count at the end of sph_step_begin: [n_mark, n) was appended mid-step
    int fresh_state = 0;          // rigid particles appended after prepare(): 1 = the next post-sort volume pass leaves them alone, 2 = the one after includes them (see ph_rigid_volume)
    RigidPose pose_h;
    SphStats last;
    ProfSlot prof[SPH_K_COUNT_];
    std::vector<hipEvent_t> ev_pool;
    std::vector<void *> allocs;
    DevScalars *scal_h = nullptr;  // pinned
    SlabComm comm;
    long long comm_n_global = 0;   // particle_num of the whole scene (sum of the ranks' owned particles), see sph_prepare
    long long comm_nfluid_global = 0;   // fluid_particle_num of the whole scene (PCISPH's error mean divides by it)
    // Slab sharding: the scene's slab axis (z by contract; SPH_SLAB_AXIS=x|y|z) is SWAPPED with x at this ABI boundary,
    // because the library cuts its grid along x, the slowest axis of its cell order (layers = contiguous index ranges of the sorted
    // arrays: boundary workgroups first, interior ones while the halo flies).  0: no swap.  Everything a caller hands over or reads
    // back stays in the scene's frame: positions / velocities / every 3-vector field, gravity, domain, grid, rigid poses and
    // wrenches are permuted here (P = P^-1, det P = -1: axial vectors -- torque, angular velocity -- also change sign).
    int swap_axis = 0;             // != 0: the frames differ (kept as a flag; the permutation itself follows)
    int perm[3] = {0, 1, 2};       // library axis k  = scene axis perm[k]
    int inv[3] = {0, 1, 2};        // scene axis k    = library axis inv[k]
    float axial = 1.0f;            // parity of the permutation: sign of axial vectors (torque, angular velocity) across the boundary
    int slab_axis = 2;             // library axis the slabs are cut along (Consts::slab_axis): 2 = z (default), 0 = x (SPH_SLAB_LAYOUT=slow)
};

// layers of the whole scene along the slab axis / setting the local window [lo_ghost, top) of a rank
static inline int slab_layers_glob(const Consts &c) { return c.slab_axis == 0 ? c.nx_glob : c.nz_glob; }
static inline void slab_set_window(Consts &c, int z_lo, int z_hi) {
    const int glob = slab_layers_glob(c);
    const int off = z_lo > 0 ? z_lo - 1 : 0;              // one ghost layer per interior side
    const int top = z_hi < glob ? z_hi + 1 : glob;
    if (c.slab_axis == 0) { c.cx_off = off; c.nx = top - off; } else { c.cz_off = off; c.nz = top - off; }
    c.G = c.nx * c.ny * c.nz;
}

// "zxy" = library (x, y, z) <- scene (z, x, y)
static bool set_axis_order(SphHandle *h, const char *order) {
    int p[3], seen = 0;
    for (int k = 0; k < 3; ++k) {
        const char ch = order ? order[k] : 0;
        p[k] = (ch == 'x' || ch == 'X') ? 0 : (ch == 'y' || ch == 'Y') ? 1 : (ch == 'z' || ch == 'Z') ? 2 : -1;
        if (p[k] < 0) return false;
        seen |= 1 << p[k];
    }
    if (seen != 7 || order[3]) return false;
    for (int k = 0; k < 3; ++k) { h->perm[k] = p[k]; h->inv[p[k]] = k; }
    const bool even = (p[0] == 0 && p[1] == 1) || (p[0] == 1 && p[1] == 2) || (p[0] == 2 && p[1] == 0);
    h->axial = even ? 1.0f : -1.0f;
    h->swap_axis = !(p[0] == 0 && p[1] == 1 && p[2] == 2);
    return true;
}

static int fail(SphHandle *h, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(h, call)                                                                       \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail((h), SPH_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

template <class T> static int dalloc(SphHandle *h, T **p, size_t count) {
    void *q = nullptr;
    if (count == 0) count = 1;
    HIPCHK(h, hipMalloc(&q, count * sizeof(T)));
    HIPCHK(h, hipMemset(q, 0, count * sizeof(T)));
    h->allocs.push_back(q);
    *p = (T *)q;
    return SPH_OK;
}

// host replica of base_solver.py:57 kernel_W, same rounding as oracle/sph_ref.c (host TU is built
// with -ffp-contract=off)
static float host_kernel_W(double hd, float r) {
    float res = 0.0f;
    float k = (float)(8.0 / M_PI);
    k /= (float)(hd * hd * hd);
    float q = r / (float)hd;
    if (q <= 1.0f) {
        if (q <= 0.5f) { float q2 = q * q; float q3 = q2 * q; res = k * (6.0f * q3 - 6.0f * q2 + 1.0f); }
        else res = k * 2.0f * powf(1.0f - q, 3.0f);
    }
    return res;
}

static void fill_consts(SphHandle *h) {
    const SphParams &p = h->prm;
    Consts &c = h->st.c;
    memset(&c, 0, sizeof(c));
    const int *ix = h->perm;
    c.nx = c.nx_glob = p.grid_num[ix[0]]; c.ny = p.grid_num[ix[1]]; c.nz = c.nz_glob = p.grid_num[ix[2]];
    c.cx_off = c.cz_off = 0;
    c.slab_axis = h->slab_axis;
    c.G = c.nx * c.ny * c.nz;
    const double hd = p.support_radius;
    c.grid_size = (float)hd;
    c.h = (float)hd;
    // Acceptance test of a neighbour: the reference asks `(x_i - x_j).norm() < dh` (base_container.py:559), the kernels ask
    // `r2 < h2` without the square root.  sqrtf is monotone and correctly rounded, so there is an exact threshold: the
    // smallest f32 t with sqrtf(t) >= h.  r2 < t  <=>  sqrtf(r2) < h, bit for bit (lattices put many pairs exactly one
    // support radius apart; with h2 = h * h those could fall on the other side by one ulp).
    {
        float t = c.h * c.h;
        while (sqrtf(nextafterf(t, 0.0f)) >= c.h) t = nextafterf(t, 0.0f);
        while (sqrtf(t) < c.h) t = nextafterf(t, INFINITY);
        c.h2 = t;
    }
    c.inv_h = 1.0f / c.h;
    float k = (float)(8.0 / M_PI);
    c.kW = k / (float)(hd * hd * hd);
    c.kG = 6.0f * k / (float)(hd * hd * hd);
    c.W0 = host_kernel_W(hd, 0.0f);
    const float d = (float)(2.0 * p.particle_radius);
    c.Wd = host_kernel_W(hd, sqrtf(d * d + 0.0f + 0.0f));
    c.kGh = c.kG * c.inv_h;
    c.inv_h2 = c.inv_h * c.inv_h;
    c.Wd_poly = c.Wd / c.kW;
    const double dd = 2.0 * p.particle_radius;
    c.diameter2 = (float)(dd * dd);
    c.dt = (float)p.dt;
    c.inv_dt = 1.0f / c.dt;
    c.rho0 = (float)p.density_0;
    c.inv_rho0 = 1.0f / c.rho0;
    c.g_upper = (float)p.g_upper;
    c.up_axis = h->inv[1];
    c.gx = (float)p.gravity[ix[0]]; c.gy = (float)p.gravity[ix[1]]; c.gz = (float)p.gravity[ix[2]];
    c.st = (float)p.surface_tension;
    c.cv = (float)(2 * (3 + 2) * p.viscosity);
    c.cvb = (float)(2 * (3 + 2) * p.viscosity_b);
    c.visc_eps = (float)(0.01 * hd * hd);
    c.pad = (float)p.padding;
    c.hix = (float)(p.domain_size[ix[0]] - p.padding);
    c.hiy = (float)(p.domain_size[ix[1]] - p.padding);
    c.hiz = (float)(p.domain_size[ix[2]] - p.padding);
    c.thr_kappa = (float)1e-5 * c.dt;
    c.V0 = (float)p.V0;
    c.force_global = p.force_global;
    c.stat_bank = 0;
    c.run_grouping = 0;
    { static const int xc = getenv("SPH_XCD_CHUNK") ? atoi(getenv("SPH_XCD_CHUNK")) : 0; c.xcd_chunk = xc > 0 ? xc : 0; }
    c.ghosts = 0;
}

static void refresh_counts(SphHandle *h) {
    h->st.c.n = h->n;
    h->st.has_emitter = h->prm.g_upper < 9999.0;
    h->st.c.all_fluid = (h->n_nonfluid == 0 && !h->st.has_emitter && !(h->st.slab_active && h->any_rigid_object)) ? 1 : 0;
    h->st.c.ghosts = h->st.slab_active ? 1 : 0;
    // staging groups of the neighbour passes (sph_device.hpp run_of): outer runs mixed in the fast build on unsharded grids that are not thin
    // (a slab's runs overlap and are staged once, nbr_plan "chain": x-offset groups there); SPH_RUN_GROUPING=0|1 overrides (A/B)
    {
        static const int env = getenv("SPH_RUN_GROUPING") ? atoi(getenv("SPH_RUN_GROUPING")) : -1;
        const int want = (h->prm.fast_math && !h->st.slab_active && h->st.c.nz >= 40) ? 1 : 0;
        h->st.c.run_grouping = (env >= 0 ? (env != 0) : want) && h->prm.fast_math ? 1 : 0;
    }
    h->st.has_rigid = h->n_nonfluid > 0;
    h->st.uniform_mass = (h->st.c.all_fluid && h->fluid_mass_uniform && !h->st.slab_active && !getenv("SPH_NO_UNIFORM_MASS")) ? 1 : 0;
}

extern "C" const char *sph_last_error(SphHandle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

extern "C" const char *sph_kernel_name(int k) {
    static const char *names[SPH_K_COUNT_] = {
        "hash_count", "scan", "scatter", "density", "non_pressure", "pressure_integrate", "rigid_volume",
        "dfsph_density_alpha", "dfsph_rho_adv", "dfsph_correct", "reduce", "pcisph_rho_star",
        "pcisph_pressure_accel", "cg_prepare", "cg_ap", "cg_vector", "misc", "halo", "wcsph_forces"};
    return (k >= 0 && k < SPH_K_COUNT_) ? names[k] : "?";
}

static void slab_comm_destroy(SlabComm &c);
static int slab_settle(SphHandle *h);
static inline int slab_settle_if_needed(SphHandle *h) { return h->n_exact ? SPH_OK : slab_settle(h); }

extern "C" void sph_destroy(SphHandle *h) {
    if (!h) return;
    hipSetDevice(h->device);
    if (h->st.stream) hipStreamSynchronize(h->st.stream);
    slab_comm_destroy(h->comm);
    if (h->st.push.mirror) { hipHostFree(h->st.push.mirror); h->st.push.mirror = nullptr; }
    for (auto &s : h->prof) for (auto &pr : s.pending) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    for (auto e : h->ev_pool) hipEventDestroy(e);
    for (void *p : h->allocs) hipFree(p);
    if (h->scal_h) hipHostFree(h->scal_h);
    if (h->loop_pub) hipHostFree((void *)h->loop_pub);
    if (h->st.list_count_pinned) { hipHostFree((void *)h->st.list_count_pinned); h->st.list_count_pinned = nullptr; }
    if (h->st.list_count_event) { hipEventDestroy(h->st.list_count_event); h->st.list_count_event = nullptr; }
    if (h->st.stream) hipStreamDestroy(h->st.stream);
    delete h;
}

extern "C" int sph_create(const SphParams *params, SphHandle **out) {
    if (!params || !out) return fail(nullptr, SPH_ERR_INVALID, "sph_create: null argument");
    *out = nullptr;
    const SphParams &p = *params;
    if (p.particle_max_num < 0 || p.grid_num[0] <= 0 || p.grid_num[1] <= 0 || p.grid_num[2] <= 0 ||
        !(p.support_radius > 0) || !(p.dt > 0))
        return fail(nullptr, SPH_ERR_INVALID, "sph_create: invalid parameters");
    if ((double)p.grid_num[0] * p.grid_num[1] * p.grid_num[2] > 2.0e9)
        return fail(nullptr, SPH_ERR_INVALID, "sph_create: grid too large");
    // particle indices travel as 32-bit BYTE offsets into float4 arrays (ldg_idx) and as 28-bit slot numbers of the slab
    // sharding: 2^28 - 1 particles per handle (~70 GB of state; a bigger scene is sharded over GPUs)
    if (p.particle_max_num > 0x0fffffff)
        return fail(nullptr, SPH_ERR_CAPACITY, "sph_create: particle_max_num %d exceeds 268435455 per GPU; shard the scene (sph_comm_set_slab)", p.particle_max_num);
    if (p.method < 0 || p.method > 2) return fail(nullptr, SPH_ERR_INVALID, "sph_create: unknown method %d", p.method);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, SPH_ERR_NO_DEVICE, "sph_create: no HIP device visible (libsph_hip has no CPU path)");
    SphHandle *h = new SphHandle();
    h->prm = p;
    int dev = p.device;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= ndev) { delete h; return fail(nullptr, SPH_ERR_NO_DEVICE, "sph_create: device %d not present", dev); }
    h->device = dev;
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) h->dev_cus = cus; }
#define CHK_CREATE(call)                                                                   \
    do {                                                                                   \
        int rc_ = (call);                                                                  \
        if (rc_ != SPH_OK) { g_create_error = h->err; sph_destroy(h); return rc_; }        \
    } while (0)
#define HIP_CREATE(call)                                                                   \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            fail(nullptr, SPH_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));     \
            sph_destroy(h);                                                                \
            return SPH_ERR_HIP;                                                            \
        }                                                                                  \
    } while (0)
    HIP_CREATE(hipSetDevice(dev));
    hipDeviceProp_t prop;
    HIP_CREATE(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fail(nullptr, SPH_ERR_NO_DEVICE, "sph_create: device %d is %s, this library is built for gfx950 only", dev, prop.gcnArchName);
        sph_destroy(h);
        return SPH_ERR_NO_DEVICE;
    }
    h->L = p.fast_math ? sph_launch_fast() : sph_launch_strict();
    State &s = h->st;
    memset(&s.c, 0, sizeof(s.c));
    s.stream = nullptr;
    HIP_CREATE(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    if (const char *ord = getenv("SPH_AXIS_ORDER")) {   // which scene axes the library's (x, y, z) are, e.g. "zxy" (A/B of the cell order; sph_comm_set_slab picks one for sharded runs)
        if (!set_axis_order(h, ord)) { fail(nullptr, SPH_ERR_INVALID, "sph_create: SPH_AXIS_ORDER must be a permutation of xyz"); sph_destroy(h); return SPH_ERR_INVALID; }
    }
    fill_consts(h);
    const size_t cap = (size_t)p.particle_max_num;
    s.cap = p.particle_max_num;
    for (int k = 0; k < 2; ++k) {
        CHK_CREATE(dalloc(h, &s.posv.b[k], cap)); CHK_CREATE(dalloc(h, &s.velm.b[k], cap));
        CHK_CREATE(dalloc(h, &s.meta.b[k], cap)); CHK_CREATE(dalloc(h, &s.pid.b[k], cap));
        CHK_CREATE(dalloc(h, &s.color.b[k], cap)); CHK_CREATE(dalloc(h, &s.rho.b[k], cap));
    }
    s.orig.b[0] = s.orig.b[1] = nullptr;
    const size_t G = (size_t)s.c.G;
    CHK_CREATE(dalloc(h, &s.cell_count, G + SPH_NGRAVE + 1)); CHK_CREATE(dalloc(h, &s.cell_start, G + SPH_NGRAVE + 1));   // + graveyard cells (slab sharding)
    CHK_CREATE(dalloc(h, &s.cellid, cap)); CHK_CREATE(dalloc(h, &s.rank, cap)); CHK_CREATE(dalloc(h, &s.tmp_idx, 2 * cap));   // (int2 run records of the stable sort)
    s.run_head = nullptr; s.run_rec = nullptr; s.sort_inv = nullptr; s.sort_epoch = 0u; s.run_lists_filed = 0; s.n_list_sorts = 0;
    s.sort_skip_rho = 0; s.color_home = nullptr; s.color_home_ok = 0; s.color_stale = 0;
    if (h->prm.deterministic && !getenv("SPH_NO_RUN_LISTS")) { CHK_CREATE(dalloc(h, &s.run_head, G + 1)); CHK_CREATE(dalloc(h, &s.run_rec, cap + G + 1)); CHK_CREATE(dalloc(h, &s.sort_inv, cap)); CHK_CREATE(dalloc(h, &s.color_home, cap)); s.color_home_ok = 1; }   // deterministic sort by run lists (RunList, sph_common.hpp)
    s.scan_blocks = (int)((G + SPH_NGRAVE + 2047) / 2048);
    if (s.scan_blocks < SPH_STAT_SLOTS / 256) s.scan_blocks = SPH_STAT_SLOTS / 256;   // k_scan_final also clears the statistics slots
    s.scan_tile_state = nullptr; if (!getenv("SPH_SCAN_ALL_TILES")) CHK_CREATE(dalloc(h, &s.scan_tile_state, (size_t)s.scan_blocks + 1));
    CHK_CREATE(dalloc(h, &s.scan_partial, 2 * ((size_t)s.scan_blocks + 1) * 8));   // two banks of tile sums, SCAN_PARTIAL_STRIDE ints apart (State::scan_bank)
    s.scan_bank = 0; s.tile_sums_ready = 0; s.skip_residual = 0; s.hist_taken = 0; s.state_error = 0;
    s.cell_count_clean = 1;
    CHK_CREATE(dalloc(h, &s.rho_raw, cap)); CHK_CREATE(dalloc(h, &s.prs, cap)); CHK_CREATE(dalloc(h, &s.ptm, cap));
    CHK_CREATE(dalloc(h, &s.acc, cap));
    s.nbr_mask = nullptr; s.masks_valid = 0; s.density_books_forces = 0; s.uniform_mass = 0;
    s.nbr_mask_hi = nullptr;
    if (!getenv("SPH_NO_MASK_REUSE")) { CHK_CREATE(dalloc(h, &s.nbr_mask, cap * 9 + 256)); CHK_CREATE(dalloc(h, &s.nbr_mask_hi, cap * 9 + 256)); }   // + 256: the lanes past the last particle of the last tile read (and drop) a word too
    s.lane_perm = nullptr; s.perm_n = -1;
    s.loop_flag = nullptr; s.loop_slot = 0; s.loop_kind = 0; s.loop_denom = 1.0f; s.loop_thr = 0.0;
    CHK_CREATE(dalloc(h, &s.blk_hdr, (cap + 255) / 256 * (20 + 256)));   // headers of all tiles, then one cell word per particle slot (k_block_prep)
    s.blk_flag = s.blk_list = s.blk_count = nullptr; s.list_n = -1; s.last_pass_listed = 0; s.list_count_pinned = nullptr; s.list_count_event = nullptr; s.list_count_known = -1; s.nexthash = NextHash{0, nullptr, nullptr, nullptr, nullptr, RunList{nullptr, nullptr, 0, 0u}}; s.prehashed = 0; s.n_hash_launches = s.n_prehashed_sorts = 0;
    if (!getenv("SPH_NO_BLOCK_LIST")) {
        CHK_CREATE(dalloc(h, &s.blk_flag, (cap + 255) / 256)); CHK_CREATE(dalloc(h, &s.blk_list, (cap + 255) / 256)); CHK_CREATE(dalloc(h, &s.blk_count, 1));
    }
    if (s.nbr_mask && !getenv("SPH_NO_LANE_PERM")) CHK_CREATE(dalloc(h, &s.lane_perm, (cap + 255) / 256 * 256));
    s.alpha = s.kappa = s.kappa_v = s.rho_star = s.rho_deriv = s.kappa_next = s.kappa_v_next = nullptr; s.kr = nullptr;
    s.pacc = s.pvel = s.ppos = s.acc_np = nullptr; s.np_acc_out = nullptr; s.np_visc_vel = nullptr;
    s.cg_p2 = nullptr; s.cg_fuse = s.cg_fused_loop = 0;
    s.cg_p = s.cg_Ap = s.cg_x = s.cg_b = s.cg_r = s.cg_v0 = nullptr; s.cg_dinv = nullptr; s.cg_part = nullptr; s.cg_split = 0; s.cg_nocombine = 0; s.split_next_pass = 0;
    if (p.method == SPH_METHOD_DFSPH) {
        CHK_CREATE(dalloc(h, &s.alpha, cap)); CHK_CREATE(dalloc(h, &s.kappa, cap)); CHK_CREATE(dalloc(h, &s.kappa_v, cap));
        CHK_CREATE(dalloc(h, &s.rho_star, cap)); CHK_CREATE(dalloc(h, &s.rho_deriv, cap)); CHK_CREATE(dalloc(h, &s.kr, cap));
        CHK_CREATE(dalloc(h, &s.kappa_next, cap)); CHK_CREATE(dalloc(h, &s.kappa_v_next, cap));
    }
    if (p.method == SPH_METHOD_PCISPH) {
        CHK_CREATE(dalloc(h, &s.pacc, cap)); CHK_CREATE(dalloc(h, &s.pvel, cap)); CHK_CREATE(dalloc(h, &s.ppos, cap));
        CHK_CREATE(dalloc(h, &s.rho_star, cap)); CHK_CREATE(dalloc(h, &s.acc_np, cap));
        s.np_acc_out = s.acc_np;
    }
    if (p.viscosity_implicit) {
        CHK_CREATE(dalloc(h, &s.cg_p, cap)); CHK_CREATE(dalloc(h, &s.cg_Ap, cap)); CHK_CREATE(dalloc(h, &s.cg_x, cap));
        CHK_CREATE(dalloc(h, &s.cg_b, cap)); CHK_CREATE(dalloc(h, &s.cg_r, cap)); CHK_CREATE(dalloc(h, &s.cg_v0, cap));
        CHK_CREATE(dalloc(h, &s.cg_dinv, cap * 9));
        CHK_CREATE(dalloc(h, &s.cg_part, cap * 3));
        CHK_CREATE(dalloc(h, &s.cg_p2, cap));
    }
    s.red_blocks = (int)((cap + 255) / 256) + 1;
    CHK_CREATE(dalloc(h, &s.red_partial, (size_t)s.red_blocks * 8));   // (also the four partial-sum arrays of the CG kernels)
    s.cg_parity = 0;
    CHK_CREATE(dalloc(h, &s.scal, 1)); CHK_CREATE(dalloc(h, &s.pose, 1));
    HIP_CREATE(hipHostMalloc((void **)&h->scal_h, sizeof(DevScalars), hipHostMallocDefault));
    { void *lp = nullptr; HIP_CREATE(hipHostMalloc(&lp, 128, hipHostMallocDefault)); memset(lp, 0, 128); h->loop_pub = (struct LoopPub *)lp; }   // + StatsPub behind it
    { int *lc = nullptr; HIP_CREATE(hipHostMalloc((void **)&lc, 64, hipHostMallocDefault)); *lc = 0; s.list_count_pinned = lc; }
    HIP_CREATE(hipEventCreateWithFlags(&s.list_count_event, hipEventDisableTiming));
    s.list_count_known = -1;
    memset(h->scal_h, 0, sizeof(DevScalars));
    memset(&h->pose_h, 0, sizeof(h->pose_h));
    for (int o = 0; o < SPH_NOBJ; ++o) { h->pose_h.rot[o][0] = h->pose_h.rot[o][4] = h->pose_h.rot[o][8] = 1.0f; }
    memset(&h->last, 0, sizeof(h->last));
    s.has_dynamic_rigid = 0; s.has_rigid = 0;
    s.dyn = s.dyn_cur = nullptr; s.async_counts = 0; s.tables_pending = 0; memset(&s.push, 0, sizeof(s.push));
    s.tile_list[0] = s.tile_list[1] = nullptr; s.tile_cnt = nullptr; s.tile_class = nullptr; s.tile_sel = 0; s.tile_plan_n = -1;
    s.tile_bound_b = 0; memset(&s.presend, 0, sizeof(s.presend)); memset(&s.fieldsend, 0, sizeof(s.fieldsend)); s.preclassified = 0;
    s.slab_active = 0; s.xcur = 0; s.halo_cap = 0; s.z_lo = 0; s.z_hi = slab_layers_glob(s.c); s.has_down = s.has_up = 0;
    s.xidx[0] = s.xidx[1] = nullptr; s.halo_counts = nullptr;
    s.visc_rho_raw = (p.method == SPH_METHOD_WCSPH);
    s.skip_viscosity = 0;
    refresh_counts(h);
    *out = h;
    return SPH_OK;
}

// ---------------------------------------------------------------------------------- scene upload
// rigid_particle_original_positions (base_container.py:155): allocated with the first dynamic rigid body
static int ensure_orig(SphHandle *h) {
    State &s = h->st;
    if (!s.orig.b[0]) {
        int rc = dalloc(h, &s.orig.b[0], (size_t)s.cap); if (rc) return rc;
        rc = dalloc(h, &s.orig.b[1], (size_t)s.cap); if (rc) return rc;
        // existing particles: original position = current position
        HIPCHK(h, hipMemcpyAsync(s.orig.cur(), s.posv.cur(), sizeof(float4) * (size_t)h->n, hipMemcpyDeviceToDevice, s.stream));
    }
    s.has_dynamic_rigid = 1;
    return SPH_OK;
}

extern "C" int sph_append_particles(SphHandle *h, int object_id, int n, const float *pos, const float *vel,
                                    const float *density, const float *pressure, const int32_t *material,
                                    const int32_t *is_dynamic, const int32_t *color) {
    if (!h) return SPH_ERR_INVALID;
    if (n < 0 || object_id < -1 || object_id >= SPH_MAX_OBJECTS) return fail(h, SPH_ERR_INVALID, "append: bad object id / count");
    if (n == 0) return SPH_OK;
    if (!pos || !vel || !density || !material || !is_dynamic) return fail(h, SPH_ERR_INVALID, "append: null array");
    { int rc = slab_settle_if_needed(h); if (rc) return rc; }
    if (h->n + n > h->st.cap) return fail(h, SPH_ERR_CAPACITY, "append: %d + %d exceeds particle_max_num %d", h->n, n, h->st.cap);
    HIPCHK(h, hipSetDevice(h->device));
    State &s = h->st;
    const float V0 = (float)h->prm.V0;
    std::vector<float4> hp(n), hv(n), ho;
    std::vector<int> hm(n), hid(n);
    std::vector<unsigned> hc(n);
    std::vector<float> hr(n), hpr(n);
    bool any_dyn_rigid = false;
    int nfl = 0;
    const int ix0 = h->perm[0], ix1 = h->perm[1], ix2 = h->perm[2];   // scene frame -> library frame
    for (int k = 0; k < n; ++k) {
        // base_container.py:404 add_particle
        hp[k] = make_float4(pos[3 * k + ix0], pos[3 * k + ix1], pos[3 * k + ix2], V0);
        hv[k] = make_float4(vel[3 * k + ix0], vel[3 * k + ix1], vel[3 * k + ix2], V0 * density[k]);
        hm[k] = META_PACK(object_id, material[k], is_dynamic[k] ? 1 : 0);
        if (h->prepared && material[k] == SPH_MAT_RIGID) { hm[k] |= META_FRESH_BIT; h->fresh_state = 1; }
        hid[k] = h->n + k;
        unsigned r = color ? (unsigned)(color[3 * k] & 0xff) : 0, g = color ? (unsigned)(color[3 * k + 1] & 0xff) : 0,
                 b = color ? (unsigned)(color[3 * k + 2] & 0xff) : 0;
        hc[k] = r | (g << 8) | (b << 16);
        hr[k] = density[k];
        hpr[k] = pressure ? pressure[k] : 0.0f;
        if (material[k] == SPH_MAT_FLUID) {
            nfl++;
            if (!h->fluid_mass_seen) { h->fluid_mass_seen = true; h->fluid_mass0 = hv[k].w; }
            else if (hv[k].w != h->fluid_mass0) h->fluid_mass_uniform = false;
        }
        if (material[k] == SPH_MAT_RIGID && is_dynamic[k]) any_dyn_rigid = true;
    }
    if (any_dyn_rigid) { int rc = ensure_orig(h); if (rc) return rc; }
    HIPCHK(h, hipStreamSynchronize(s.stream));
    const size_t off = (size_t)h->n;
    HIPCHK(h, hipMemcpy(s.posv.cur() + off, hp.data(), sizeof(float4) * n, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(s.velm.cur() + off, hv.data(), sizeof(float4) * n, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(s.meta.cur() + off, hm.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(s.pid.cur() + off, hid.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(s.color.cur() + off, hc.data(), sizeof(unsigned) * n, hipMemcpyHostToDevice));
    if (s.color_home) HIPCHK(h, hipMemcpy(s.color_home + off, hc.data(), sizeof(unsigned) * n, hipMemcpyHostToDevice));   // (particle id = append index: hid above)
    HIPCHK(h, hipMemcpy(s.rho.cur() + off, hr.data(), sizeof(float) * n, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(s.prs + off, hpr.data(), sizeof(float) * n, hipMemcpyHostToDevice));
    if (s.orig.cur()) HIPCHK(h, hipMemcpy(s.orig.cur() + off, hp.data(), sizeof(float4) * n, hipMemcpyHostToDevice));
    h->n += n;
    h->n_fluid += nfl;
    h->n_nonfluid += n - nfl;
    if (!h->prepared) h->rigid_volume_done = false;   // later arrivals: see fresh_state
    if (h->prepared) h->sort_dirty = true;
    s.masks_valid = 0;
    s.perm_n = -1; s.list_n = -1;
    refresh_counts(h);
    return SPH_OK;
}

// persistent ids of the n particles appended last (a rank of a sharded scene numbers its particles with their global
// insertion indices, so that ids mean the same thing on every rank and in a single-GPU run)
extern "C" int sph_set_appended_ids(SphHandle *h, int n, const int32_t *ids) {
    if (h) { int rc = slab_settle_if_needed(h); if (rc) return rc; }
    if (!h || !ids || n < 0 || n > h->n) return fail(h, SPH_ERR_INVALID, "set_appended_ids: bad arguments");
    if (n == 0) return SPH_OK;
    HIPCHK(h, hipSetDevice(h->device));
    h->L->ensure_color(h->st); h->st.color_home_ok = 0;   // ids from outside: the colours travel with the particles from now on
    HIPCHK(h, hipStreamSynchronize(h->st.stream));
    HIPCHK(h, hipMemcpy(h->st.pid.cur() + (size_t)(h->n - n), ids, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice));
    return SPH_OK;
}

static int upload_pose(SphHandle *h) {
    HIPCHK(h, hipMemcpyAsync(h->st.pose, &h->pose_h, sizeof(RigidPose), hipMemcpyHostToDevice, h->st.stream));
    HIPCHK(h, hipStreamSynchronize(h->st.stream));  // pose_h may change right after
    return SPH_OK;
}

extern "C" int sph_set_object(SphHandle *h, int object_id, int material, int is_dynamic) {
    if (!h || object_id < 0 || object_id >= SPH_MAX_OBJECTS) return fail(h, SPH_ERR_INVALID, "set_object: bad object id");
    HIPCHK(h, hipSetDevice(h->device));
    h->pose_h.material[object_id] = material;
    if (material != 1) h->any_rigid_object = true;
    h->pose_h.is_dynamic[object_id] = is_dynamic ? 1 : 0;
    // a rank of a sharded scene may hold none of the body's particles now and receive them later as migrants (which carry
    // their rest positions): every rank that is told about the body keeps the array
    if (material == SPH_MAT_RIGID && is_dynamic) { int rc = ensure_orig(h); if (rc) return rc; }
    return upload_pose(h);
}

extern "C" int sph_set_rigid_pose(SphHandle *h, int o, const float *com, const float *rot9, const float *vel,
                                  const float *angvel, const float *com0) {
    if (!h || o < 0 || o >= SPH_MAX_OBJECTS || !com || !rot9 || !vel || !angvel) return fail(h, SPH_ERR_INVALID, "set_rigid_pose: bad argument");
    HIPCHK(h, hipSetDevice(h->device));
    // scene frame -> library frame: polar vectors P v, the axial angular velocity det(P) P w, the rotation P R P^T
    for (int a = 0; a < 3; ++a) {
        const int b = h->perm[a];
        h->pose_h.com[o][a] = com[b]; h->pose_h.vel[o][a] = vel[b]; h->pose_h.angvel[o][a] = h->axial * angvel[b]; if (com0) h->pose_h.com0[o][a] = com0[b];
        for (int q = 0; q < 3; ++q) h->pose_h.rot[o][3 * a + q] = rot9[3 * b + h->perm[q]];
    }
    h->pose_dirty = true;
    h->pose_given = true;
    return upload_pose(h);
}

extern "C" int sph_get_rigid_wrench(SphHandle *h, float *force, float *torque, int reset) {
    if (!h || !force || !torque) return fail(h, SPH_ERR_INVALID, "get_rigid_wrench: null");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(h->scal_h, h->st.scal, sizeof(DevScalars), hipMemcpyDeviceToHost, h->st.stream));
    HIPCHK(h, hipStreamSynchronize(h->st.stream));
    // library frame -> scene frame: the force is polar (P^T f), the torque axial (det(P) P^T t)
    for (int o = 0; o < SPH_NOBJ; ++o)
        for (int a = 0; a < 3; ++a) {
            const int b = h->inv[a];
            force[3 * o + a] = (float)((double)h->scal_h->wrench[3 * o + b] / SPH_WRENCH_SCALE);
            torque[3 * o + a] = h->axial * (float)((double)h->scal_h->wrench[SPH_NOBJ * 3 + 3 * o + b] / SPH_WRENCH_SCALE);
        }
    if (h->st.slab_active && h->comm.nranks > 1) {
        // sharded scene: every rank holds the contributions of ITS fluid particles (SURVEY 8e "rigid coupling under sharding");
        // the body's wrench is their sum.  Collective: every rank calls this at the same point of the step (the host
        // rigid solver does, between sph_step_begin and sph_step_end).
        double w[2 * SPH_NOBJ * 3];
        for (int k = 0; k < SPH_NOBJ * 3; ++k) { w[k] = force[k]; w[SPH_NOBJ * 3 + k] = torque[k]; }
        { int rc = sph_comm_allreduce(h, w, 2 * SPH_NOBJ * 3, 0); if (rc) return rc; }   // one collective
        for (int k = 0; k < SPH_NOBJ * 3; ++k) { force[k] = (float)w[k]; torque[k] = (float)w[SPH_NOBJ * 3 + k]; }
    }
    if (reset)
        HIPCHK(h, hipMemsetAsync((char *)h->st.scal + offsetof(DevScalars, wrench), 0, sizeof(long long) * 2 * SPH_NOBJ * 3, h->st.stream));
    return SPH_OK;
}

// ---------------------------------------------------------------------------------- profiling
static hipEvent_t get_event(SphHandle *h) {
    if (!h->ev_pool.empty()) { hipEvent_t e = h->ev_pool.back(); h->ev_pool.pop_back(); return e; }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}

static void prof_resolve(SphHandle *h) {
    hipStreamSynchronize(h->st.stream);
    for (auto &s : h->prof) {
        for (auto &pr : s.pending) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { s.ms += ms; s.launches++; }
            h->ev_pool.push_back(pr.first); h->ev_pool.push_back(pr.second);
        }
        s.pending.clear();
    }
}

struct ProfScope {
    SphHandle *h; int k; hipEvent_t e1 = nullptr;
    ProfScope(SphHandle *h_, int k_) : h(h_), k(k_) {
        if (h->prof[k].on) {
            if (h->prof[k].pending.size() > 8192) prof_resolve(h);
            hipEvent_t e0 = get_event(h); e1 = get_event(h);
            hipEventRecord(e0, h->st.stream);
            h->prof[k].pending.push_back({e0, e1});
        }
    }
    ~ProfScope() { if (e1) hipEventRecord(e1, h->st.stream); }
};

extern "C" int sph_profile_enable(SphHandle *h, int kernel_id, int on) {
    if (!h || kernel_id >= SPH_K_COUNT_) return SPH_ERR_INVALID;
    for (int k = 0; k < SPH_K_COUNT_; ++k) if (kernel_id < 0 || kernel_id == k) h->prof[k].on = on != 0;
    return SPH_OK;
}
extern "C" int sph_profile_reset(SphHandle *h) {
    if (!h) return SPH_ERR_INVALID;
    hipSetDevice(h->device);
    prof_resolve(h);
    for (auto &s : h->prof) { s.launches = 0; s.ms = 0.0; }
    return SPH_OK;
}
extern "C" int sph_profile_read(SphHandle *h, int kernel_id, int64_t *launches, double *total_ms) {
    if (!h || kernel_id < 0 || kernel_id >= SPH_K_COUNT_) return SPH_ERR_INVALID;
    hipSetDevice(h->device);
    prof_resolve(h);
    if (launches) *launches = h->prof[kernel_id].launches;
    if (total_ms) *total_ms = h->prof[kernel_id].ms;
    return SPH_OK;
}

extern "C" int sph_device_info(SphHandle *h, char *name256, int *cu_count, int64_t *hbm_bytes) {
    if (!h) return SPH_ERR_INVALID;
    hipDeviceProp_t prop;
    HIPCHK(h, hipGetDeviceProperties(&prop, h->device));
    // some driver stacks leave prop.name empty: report the architecture string alone then
    if (name256) { if (prop.name[0]) snprintf(name256, 256, "%s (%s)", prop.name, prop.gcnArchName); else snprintf(name256, 256, "%s, %d CUs", prop.gcnArchName, prop.multiProcessorCount); }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return SPH_OK;
}

// device copy rate (read + write bytes per second) of `reps` D2D copies of `bytes` bytes, HIP events on the handle's stream
extern "C" int sph_measure_copy_rate(SphHandle *h, size_t bytes, int reps, double *gb_per_s) {
    if (!h || !gb_per_s || bytes < 4096 || reps < 1) return fail(h, SPH_ERR_INVALID, "measure_copy_rate: bad arguments");
    HIPCHK(h, hipSetDevice(h->device));
    void *a = nullptr, *b = nullptr;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) {
        if (a) hipFree(a);
        return fail(h, SPH_ERR_HIP, "measure_copy_rate: cannot allocate 2 x %zu bytes", bytes);
    }
    hipStream_t st = h->st.stream;
    hipEvent_t e0 = get_event(h), e1 = get_event(h);
    hipError_t e = hipMemsetAsync(a, 1, bytes, st);
    if (e == hipSuccess) e = hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, st);   // warm-up (page tables, clocks)
    if (e == hipSuccess) e = hipEventRecord(e0, st);
    for (int k = 0; k < reps && e == hipSuccess; ++k) e = hipMemcpyAsync((k & 1) ? a : b, (k & 1) ? b : a, bytes, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipEventRecord(e1, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    float ms = 0.0f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    h->ev_pool.push_back(e0); h->ev_pool.push_back(e1);
    hipFree(a); hipFree(b);
    if (e != hipSuccess || !(ms > 0.0f)) return fail(h, SPH_ERR_HIP, "measure_copy_rate: %s", hipGetErrorString(e));
    *gb_per_s = 2.0 * (double)bytes * reps / ((double)ms * 1e-3) / 1e9;
    return SPH_OK;
}

// ---------------------------------------------------------------------------------- phases
static int check_async(SphHandle *h) {
    if (h->st.state_error) { const int se = h->st.state_error; h->st.state_error = 0; return fail(h, SPH_ERR_INVALID, "internal state error %d (a sort was scanned without a histogram of its own)", se); }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, SPH_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    return SPH_OK;
}

// base_container.py:544 prepare_neighborhood_search
static void ph_neighbor_search(SphHandle *h) {
    State &s = h->st;
    { ProfScope p(h, SPH_K_HASH_COUNT); h->L->hash_count(s); }
    { ProfScope p(h, SPH_K_SCAN); h->L->scan(s); }
    { ProfScope p(h, SPH_K_SCATTER); if (h->prm.deterministic) h->L->scatter_stable(s); else h->L->scatter(s); }
    h->sort_dirty = false;
}

// the same without the hash kernel: the push transport's classify / unpack kernels have hashed every particle already
static void ph_sort_hashed(SphHandle *h) {
    State &s = h->st;
    { ProfScope p(h, SPH_K_SCAN); h->L->scan(s); }
    { ProfScope p(h, SPH_K_SCATTER); if (h->prm.deterministic) h->L->scatter_stable(s); else h->L->scatter(s); }
    h->sort_dirty = false;
}

static void ph_rigid_volume(SphHandle *h) {
    // base_solver.py:106.  The reference runs it at the end of every step() (:696) on the grid of that step's sort.
    // Static boundaries: the sum only involves same-object (static) particles, so the value computed once after the
    // first sort is bit-identical to recomputing every step; with dynamic bodies it is recomputed after every sort.
    // Rigid particles appended after prepare() (late entryTime) carry META_FRESH: the reference's pass at the end of
    // their insertion step ran on a grid that did not contain them (WCSPH / PCISPH: V = 1 / W(0), set by post_insert;
    // DFSPH: they were sorted in before :317, with V = V0 from add_particle), and only the pass one step later saw them.
    if (!h->st.has_rigid) return;
    bool force = false;
    if (h->fresh_state == 2) { h->L->clear_fresh(h->st); h->fresh_state = 0; force = true; }
    else if (h->fresh_state == 1 && h->prm.method != SPH_METHOD_DFSPH) h->fresh_state = 2;   // this pass skips them (RigidVolumePass::begin)
    if (!force && h->rigid_volume_done && !h->st.has_dynamic_rigid) return;
    ProfScope p(h, SPH_K_RIGID_VOLUME);
    h->L->rigid_volume(h->st);
    h->rigid_volume_done = true;
}

static void step_begin(SphHandle *h) {
    State &s = h->st;
    s.c.stat_bank = (int)(h->steps & 1);   // cleared by the previous step's scan kernel (k_scan_lookback)
    // a pose pushed between two steps is NOT applied here: the reference reads the rigid_body_* fields in the middle of
    // _step() (renew_rigid_particle_state, WCSPH.py:43 / DFSPH.py:309 / PCISPH.py:183), after the fluid passes of the step --
    // step_second_half does that (pinned by the rigid_* fixtures, which write a pose between steps 1 and 2)
}

#include "sph_comm_api.hpp"
#include "sph_steps.hpp"

static int read_scalars(SphHandle *h) {
    { int rc = slab_settle_if_needed(h); if (rc) return rc; }
    static const bool no_publish = getenv("SPH_NO_LOOP_PUBLISH") != nullptr;
    if (!no_publish && h->loop_pub) {   // the sums of the last step's bank, added up on the device and published into pinned memory (sph_steps.hpp)
        StatsPub *pub = (StatsPub *)((char *)h->loop_pub + 64);
        const unsigned want = ++h->stats_seq;
        const int bank_d = h->steps > 0 ? (int)((h->steps - 1) & 1) : 0;
        hipLaunchKernelGGL(k_publish_stats, dim3(1), dim3(256), 0, h->st.stream, h->st.scal, bank_d, pub, want);
        if (spin_for(h->st, &pub->seq, want)) {
            h->last.pair_interactions = (int64_t)pub->pairs;
            h->last.pair_evaluations = (int64_t)pub->evals;
            h->last.lds_fallback_blocks = (int64_t)pub->fallback;
            return SPH_OK;
        }
    }
    HIPCHK(h, hipMemcpyAsync(h->scal_h, h->st.scal, sizeof(DevScalars), hipMemcpyDeviceToHost, h->st.stream));
    HIPCHK(h, hipStreamSynchronize(h->st.stream));
    unsigned long long pairs = 0, evals = 0, fb = 0;
    const int bank = h->steps > 0 ? (int)((h->steps - 1) & 1) : 0;   // bank of the last completed step
    for (int k = 0; k < SPH_STAT_SLOTS; ++k) {
        pairs += h->scal_h->pairs[bank][k]; evals += h->scal_h->evals[bank][k];
        fb += h->scal_h->fallback[bank][k];
    }
    h->last.pair_interactions = (int64_t)pairs;
    h->last.pair_evaluations = (int64_t)evals;
    h->last.lds_fallback_blocks = (int64_t)fb;
    return SPH_OK;
}

// Waiting for the stream and reading small results back (round 6; profiles/r06_sync_latency.txt).  What is slow on this runtime is not
// hipStreamSynchronize but a D2H copy issued on an IDLE stream: the 97 KB statistics copy of sph_get_stats cost ~0.3 ms per call once the same
// copy at the end of sph_step (issued while the step's kernels were still running, which hid its start-up) was taken away -- a synchronous C5
// step went 1.39 -> 1.75 ms.  The statistics are therefore added up on the device and published into pinned host memory like the solver
// loops' residuals (read_scalars -> k_publish_stats; sph_steps.hpp), no copy at all; sph_step / sph_step_end just wait.
// sph_synchronize -- the fence a caller times asynchronous steps with -- waits by the same publish + spin (ends ~30 us sooner than the
// interrupt-driven hipStreamSynchronize: C2 in the driver's 3 x 20-step configuration -0.5 %); SPH_SLOW_SYNC=1: plain hipStreamSynchronize.
static int fast_stream_sync(SphHandle *h) {
    static const bool slow = getenv("SPH_SLOW_SYNC") != nullptr;
    if (!slow) return loop_readback(h);          // (returns only when the stream has reached the kernel it appended)
    HIPCHK(h, hipStreamSynchronize(h->st.stream));
    return SPH_OK;
}
// sph_step / sph_step_end are synchronous on return; the statistics are brought over when somebody asks (sph_get_stats)
static int read_scalars(SphHandle *h);
static int finish_sync(SphHandle *h) {
    static const bool copy_stats = getenv("SPH_STEP_COPIES_STATS") != nullptr;   // A/B: as until round 6
    if (copy_stats) return read_scalars(h);
    { int rc = slab_settle_if_needed(h); if (rc) return rc; }
    HIPCHK(h, hipStreamSynchronize(h->st.stream));
    return SPH_OK;
}

extern "C" int sph_prepare(SphHandle *h) {
    if (!h) return SPH_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    refresh_counts(h);
    State &s = h->st;
    int rc = upload_pose(h); if (rc) return rc;
    // base_solver.py:683 prepare: prepare_emitter, renew_rigid_particle_state, neighbour search,
    // compute_rigid_particle_volume (+ DFSPH.py:321 / PCISPH.py:188)
    { ProfScope p(h, SPH_K_MISC); h->L->prepare_emitter(s); h->L->renew_rigid(s); }
    h->pose_dirty = false;
    if (s.slab_active) {
        double n_own[2] = {(double)h->n, (double)h->n_fluid};   // before any ghost arrives: every particle is owned by exactly one rank
        rc = sph_comm_allreduce(h, n_own, 2, 0); if (rc) return rc;
        h->comm_n_global = (long long)n_own[0];
        h->comm_nfluid_global = (long long)n_own[1];
        // halo records carry the rest position (64 instead of 48 B) once the scene has a dynamic rigid body: all ranks agree
        // on that here, whoever was told about the body (later mismatches are caught by the record size in the message header)
        double dyn_rigid[1] = {s.orig.cur() ? 1.0 : 0.0};
        rc = sph_comm_allreduce(h, dyn_rigid, 1, 1); if (rc) return rc;
        if (dyn_rigid[0] > 0.0 && !s.orig.cur()) { rc = ensure_orig(h); if (rc) return rc; }
        rc = slab_neighbor_search(h); if (rc) return rc;
    }
    else ph_neighbor_search(h);
    h->rigid_volume_done = false;
    ph_rigid_volume(h);
    if (s.slab_active && s.has_rigid) {  // ghost copies of boundary particles must carry the volumes just computed
        rc = slab_neighbor_search(h); if (rc) return rc;
    }
    rc = method_prepare(h); if (rc) return rc;
    // the passes above counted into statistics bank 0, which the first step uses as well
    HIPCHK(h, hipMemsetAsync(s.scal, 0, sizeof(unsigned long long) * SPH_STAT_SLOTS, s.stream));
    HIPCHK(h, hipMemsetAsync((char *)s.scal + offsetof(DevScalars, evals), 0, sizeof(unsigned long long) * SPH_STAT_SLOTS, s.stream));
    HIPCHK(h, hipMemsetAsync((char *)s.scal + offsetof(DevScalars, fallback), 0, sizeof(unsigned long long) * SPH_STAT_SLOTS, s.stream));
    rc = check_async(h); if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(s.stream));
    h->prepared = true;
    return SPH_OK;
}

// First half of a step: everything the reference's _step() does before `self.rigid_solver.step()`.
static int step_first_half(SphHandle *h, bool allow_readback) {
    if (h->in_step) return fail(h, SPH_ERR_INVALID, "sph_step_begin: the previous step was not ended");
    step_begin(h);
    int rc;
    switch (h->prm.method) {
        case SPH_METHOD_WCSPH: rc = wcsph_step(h); break;                     // WCSPH.py:28-36 (:45 boundary fused into the position update)
        case SPH_METHOD_DFSPH: rc = dfsph_step_begin(h, allow_readback); break;
        default: rc = pcisph_step(h, allow_readback); break;                  // PCISPH.py:166-177
    }
    if (rc) return rc;
    h->n_mark = h->n;
    h->in_step = true;
    return SPH_OK;
}

// Second half: renew_rigid_particle_state (:616) for a pose the host pushed in between, the boundary (and the stale-grid
// rigid volume) for particles the host appended in between, DFSPH's post-insertion passes, then step()'s tail.
static int step_second_half(SphHandle *h, bool allow_readback) {
    if (!h->in_step) return fail(h, SPH_ERR_INVALID, "sph_step_end without sph_step_begin");
    State &s = h->st;
    h->in_step = false;
    if (h->pose_dirty || h->fresh_state || h->prm.method != SPH_METHOD_WCSPH) { int rc = slab_settle_if_needed(h); if (rc) return rc; }
    if (h->pose_dirty) { ProfScope p(h, SPH_K_MISC); h->L->renew_rigid(s); h->pose_dirty = false; }
    if (h->n > h->n_mark) {
        ProfScope p(h, SPH_K_MISC);
        h->L->post_insert(s, h->n_mark, h->prm.method != SPH_METHOD_DFSPH);
    }
    // (late entry under sharding: every rank appended the part of the object that lies in its slab, possibly nothing; the
    //  host tells the library the object's whole size through sph_comm_add_global_count -- no collective per step)
    if (h->prm.method == SPH_METHOD_DFSPH) {
        int rc = dfsph_step_end(h, allow_readback); if (rc) return rc;
        if (h->fresh_state == 1) { h->fresh_state = 2; ph_rigid_volume(h); }  // base_solver.py:696 on the fresh grid: now it sees them
    } else if (h->fresh_state == 2) {
        // WCSPH / PCISPH, one step after the insertion: this step's sort took the new body in, so the reference's
        // end-of-step pass (:696) now finds its particles.  The cell lists of this step's sort are still right for the
        // rigid particles (static ones have not moved), which is all this pass looks at.
        ph_rigid_volume(h);
    }
    h->total_time += (double)s.c.dt;  // base_solver.py:694
    h->steps++;
    return SPH_OK;
}

static int step_once(SphHandle *h, bool allow_readback) {
    h->whole_step = true;
    int rc = step_first_half(h, allow_readback);
    h->whole_step = false;
#ifdef SPH_TEST_HOOKS
    // test-hook library only: SPH_TEST_FAIL_STEP=k makes the step with h->steps == k fail between its halves, ONCE -- what a device or
    // exchange error in the middle of a sph_step_async(n) looks like to the state machine (a hash made for a sort that will not come)
    {
        static int fail_at = getenv("SPH_TEST_FAIL_STEP") ? atoi(getenv("SPH_TEST_FAIL_STEP")) : -1;
        if (!rc && fail_at >= 0 && h->steps == fail_at) { fail_at = -1; h->in_step = false; rc = fail(h, SPH_ERR_INVALID, "SPH_TEST_FAIL_STEP: injected failure"); }
    }
#endif
    if (!rc) rc = step_second_half(h, allow_readback);
    // a failed step may leave a hash made for a sort that will not come (NextHash: the WCSPH force pass for the next step's sort, the
    // DFSPH position update for this step's): the next sort, whoever asks for it, must hash for itself on a clean histogram
    if (rc && h->st.prehashed) { h->st.prehashed = 0; h->st.cell_count_clean = 0; h->st.hist_taken = 0; h->st.run_lists_filed = 0; }
    return rc;
}

extern "C" int sph_step_begin(SphHandle *h) {
    if (!h) return SPH_ERR_INVALID;
    if (!h->prepared) return fail(h, SPH_ERR_INVALID, "sph_step_begin before sph_prepare");
    HIPCHK(h, hipSetDevice(h->device));
    refresh_counts(h);
    int rc = step_first_half(h, true); if (rc) return rc;
    rc = slab_settle_if_needed(h); if (rc) return rc;   // the host may append next: it needs the count
    h->n_mark = h->n;
    return check_async(h);
}

extern "C" int sph_step_end(SphHandle *h) {
    if (!h) return SPH_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    refresh_counts(h);
    int rc = step_second_half(h, true); if (rc) return rc;
    rc = check_async(h); if (rc) return rc;
    return finish_sync(h);
}

extern "C" int sph_step_async(SphHandle *h, int nsteps) {
    if (!h || nsteps < 0) return SPH_ERR_INVALID;
    if (!h->prepared) return fail(h, SPH_ERR_INVALID, "sph_step before sph_prepare");
    if (h->prm.method != SPH_METHOD_WCSPH && h->prm.fixed_iterations <= 0)
        return fail(h, SPH_ERR_UNSUPPORTED, "sph_step_async needs wcsph or fixed_iterations > 0");
    HIPCHK(h, hipSetDevice(h->device));
    refresh_counts(h);
    for (int k = 0; k < nsteps; ++k) {
        h->steps_to_follow = nsteps - 1 - k;   // (a sharded WCSPH step may start the next step's halo message behind its own force pass)
        int rc = step_once(h, false);
        h->steps_to_follow = 0;
        if (rc) return rc;   // (step_once drops a hash made for a sort that will not come)
    }
    return check_async(h);
}

extern "C" int sph_synchronize(SphHandle *h) {
    if (!h) return SPH_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->n_exact) return slab_settle(h);   // drains the stream (bounded) and brings the counts of the asynchronous steps back
    return fast_stream_sync(h);
}

extern "C" int sph_step(SphHandle *h, int nsteps) {
    if (!h || nsteps < 0) return SPH_ERR_INVALID;
    if (!h->prepared) return fail(h, SPH_ERR_INVALID, "sph_step before sph_prepare");
    HIPCHK(h, hipSetDevice(h->device));
    refresh_counts(h);
    for (int k = 0; k < nsteps; ++k) { int rc = step_once(h, true); if (rc) return rc; }
    int rc = check_async(h); if (rc) return rc;
    return finish_sync(h);
}

extern "C" int sph_run_phase(SphHandle *h, int phase) {
    if (!h) return SPH_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    { int rc = slab_settle_if_needed(h); if (rc) return rc; }
    refresh_counts(h);
    State &s = h->st;
    switch (phase) {
        case SPH_PH_NEIGHBOR_SEARCH: ph_neighbor_search(h); break;
        case SPH_PH_RIGID_VOLUME: h->rigid_volume_done = false; ph_rigid_volume(h); break;
        case SPH_PH_DENSITY: { ProfScope p(h, SPH_K_DENSITY); h->L->density(s, h->prm.method == SPH_METHOD_WCSPH); } break;
        case SPH_PH_NON_PRESSURE: { int rc = run_non_pressure(h); if (rc) return rc; } break;
        case SPH_PH_PRESSURE_INTEGRATE: { ProfScope p(h, SPH_K_PRESSURE_INTEGRATE); h->L->pressure_integrate(s); } break;
        default: { int rc = method_run_phase(h, phase); if (rc) return rc; }
    }
    int rc = check_async(h); if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(s.stream));
    return SPH_OK;
}

// ---------------------------------------------------------------------------------- state access
extern "C" int sph_particle_num(SphHandle *h) {
    if (!h) return SPH_ERR_INVALID;
    if (!h->n_exact) { hipSetDevice(h->device); const int rc = slab_settle(h); if (rc) return rc; }
    return h->n;
}
extern "C" int sph_fluid_particle_num(SphHandle *h) { return h ? h->n_fluid : SPH_ERR_INVALID; }

extern "C" int sph_get_stats(SphHandle *h, SphStats *out) {
    if (!h || !out) return SPH_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    int rc = read_scalars(h); if (rc) return rc;
    h->last.steps = h->steps;
    h->last.particle_num = h->n;
    h->last.fluid_particle_num = h->n_fluid;
    h->last.total_time = h->total_time;
    h->last.hash_launches = h->st.n_hash_launches;
    h->last.prehashed_sorts = h->st.n_prehashed_sorts;
    h->last.list_sorts = h->st.n_list_sorts;
    *out = h->last;
    return SPH_OK;
}

static const float4 *vec_field(SphHandle *h, int field) {
    State &s = h->st;
    switch (field) {
        case SPH_F_POSITION: case SPH_F_REST_VOLUME: return s.posv.cur();
        case SPH_F_VELOCITY: case SPH_F_MASS: return s.velm.cur();
        case SPH_F_ACCELERATION: return s.acc;
        case SPH_F_PRESSURE_ACCEL: return s.pacc;
        case SPH_F_PREDICTED_VEL: return s.pvel;
        case SPH_F_PREDICTED_POS: return s.ppos;
        case SPH_F_CG_X: return s.cg_x;
        case SPH_F_ORIG_POSITION: return s.orig.cur() ? s.orig.cur() : s.posv.cur();
        default: return nullptr;
    }
}
static const float *scalar_field(SphHandle *h, int field) {
    State &s = h->st;
    switch (field) {
        case SPH_F_DENSITY: return s.rho.cur();
        case SPH_F_PRESSURE: return s.prs;
        case SPH_F_DFSPH_ALPHA: return s.alpha;
        case SPH_F_DFSPH_KAPPA: return s.kappa;
        case SPH_F_DFSPH_KAPPA_V: return s.kappa_v;
        case SPH_F_DENSITY_STAR: return s.rho_star;
        case SPH_F_DENSITY_DERIV: return s.rho_deriv;
        case SPH_F_DFSPH_KAPPA_NEXT: return s.kappa_next;
        case SPH_F_DFSPH_KAPPA_V_NEXT: return s.kappa_v_next;
        case SPH_F_DEBUG_CAPTURE: return s.rho_raw;   // (test-hook build: PcisphRhoStarPass::cap_prev points here)
        default: return nullptr;
    }
}

extern "C" int sph_download(SphHandle *h, int field, void *dst, size_t bytes) {
    if (!h || !dst) return SPH_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    { int rc = slab_settle_if_needed(h); if (rc) return rc; }
    State &s = h->st;
    const size_t n = (size_t)h->n;
    HIPCHK(h, hipStreamSynchronize(s.stream));
    const bool is_vec3 = field == SPH_F_POSITION || field == SPH_F_VELOCITY || field == SPH_F_ACCELERATION ||
                         field == SPH_F_PRESSURE_ACCEL || field == SPH_F_PREDICTED_VEL || field == SPH_F_PREDICTED_POS ||
                         field == SPH_F_CG_X || field == SPH_F_ORIG_POSITION;
    if (is_vec3) {
        const float4 *src = vec_field(h, field);
        if (!src) return fail(h, SPH_ERR_UNSUPPORTED, "download: field %d not allocated for this method", field);
        if (bytes != n * 12) return fail(h, SPH_ERR_INVALID, "download: field %d needs %zu bytes, got %zu", field, n * 12, bytes);
        std::vector<float4> tmp(n);
        HIPCHK(h, hipMemcpy(tmp.data(), src, n * sizeof(float4), hipMemcpyDeviceToHost));
        float *d = (float *)dst;
        const int ix0 = h->inv[0], ix1 = h->inv[1], ix2 = h->inv[2];   // library frame -> scene frame
        for (size_t i = 0; i < n; ++i) { const float v[3] = {tmp[i].x, tmp[i].y, tmp[i].z}; d[3 * i] = v[ix0]; d[3 * i + 1] = v[ix1]; d[3 * i + 2] = v[ix2]; }
        return SPH_OK;
    }
    if (field == SPH_F_REST_VOLUME || field == SPH_F_MASS) {
        if (bytes != n * 4) return fail(h, SPH_ERR_INVALID, "download: size mismatch");
        std::vector<float4> tmp(n);
        HIPCHK(h, hipMemcpy(tmp.data(), vec_field(h, field), n * sizeof(float4), hipMemcpyDeviceToHost));
        float *d = (float *)dst;
        for (size_t i = 0; i < n; ++i) d[i] = tmp[i].w;
        return SPH_OK;
    }
    if (const float *src = scalar_field(h, field)) {
        if (bytes != n * 4) return fail(h, SPH_ERR_INVALID, "download: size mismatch");
        HIPCHK(h, hipMemcpy(dst, src, n * 4, hipMemcpyDeviceToHost));
        return SPH_OK;
    }
    if (field == SPH_F_MATERIAL || field == SPH_F_OBJECT_ID || field == SPH_F_IS_DYNAMIC || field == SPH_F_GHOST) {
        if (bytes != n * 4) return fail(h, SPH_ERR_INVALID, "download: size mismatch");
        std::vector<int> tmp(n);
        HIPCHK(h, hipMemcpy(tmp.data(), s.meta.cur(), n * 4, hipMemcpyDeviceToHost));
        int32_t *d = (int32_t *)dst;
        for (size_t i = 0; i < n; ++i)
            d[i] = field == SPH_F_MATERIAL ? META_MAT(tmp[i]) : field == SPH_F_OBJECT_ID ? META_OBJ(tmp[i]) :
                   field == SPH_F_GHOST ? META_GHOST(tmp[i]) : META_DYN(tmp[i]);
        return SPH_OK;
    }
    if (field == SPH_F_PARTICLE_ID) {
        if (bytes != n * 4) return fail(h, SPH_ERR_INVALID, "download: size mismatch");
        HIPCHK(h, hipMemcpy(dst, s.pid.cur(), n * 4, hipMemcpyDeviceToHost));
        return SPH_OK;
    }
    if (field == SPH_F_COLOR) {
        if (bytes != n * 12) return fail(h, SPH_ERR_INVALID, "download: size mismatch");
        h->L->ensure_color(s); HIPCHK(h, hipStreamSynchronize(s.stream));
        std::vector<unsigned> tmp(n);
        HIPCHK(h, hipMemcpy(tmp.data(), s.color.cur(), n * 4, hipMemcpyDeviceToHost));
        int32_t *d = (int32_t *)dst;
        for (size_t i = 0; i < n; ++i) { d[3 * i] = tmp[i] & 0xff; d[3 * i + 1] = (tmp[i] >> 8) & 0xff; d[3 * i + 2] = (tmp[i] >> 16) & 0xff; }
        return SPH_OK;
    }
    if (field == SPH_F_GRID_ID) {
        if (bytes != n * 4) return fail(h, SPH_ERR_INVALID, "download: size mismatch");
        std::vector<float4> tmp(n);
        HIPCHK(h, hipMemcpy(tmp.data(), s.posv.cur(), n * sizeof(float4), hipMemcpyDeviceToHost));
        int32_t *d = (int32_t *)dst;
        // the REFERENCE's flat cell id: scene frame, whole grid (whatever this rank's slab or the library's axis order)
        const Consts &c = s.c;
        const int *gn = h->prm.grid_num;
        const int ix0 = h->inv[0], ix1 = h->inv[1], ix2 = h->inv[2];
        auto cc = [](float x, float gs, int nn) { int v = (int)(x / gs); v = v < 0 ? 0 : v; return v > nn - 1 ? nn - 1 : v; };
        for (size_t i = 0; i < n; ++i) {
            const float v[3] = {tmp[i].x, tmp[i].y, tmp[i].z};
            d[i] = (cc(v[ix0], c.grid_size, gn[0]) * gn[1] + cc(v[ix1], c.grid_size, gn[1])) * gn[2] + cc(v[ix2], c.grid_size, gn[2]);
        }
        return SPH_OK;
    }
    return fail(h, SPH_ERR_INVALID, "download: unknown field %d", field);
}

extern "C" int sph_upload(SphHandle *h, int field, const void *src, size_t bytes) {
    if (!h || !src) return SPH_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    { int rc = slab_settle_if_needed(h); if (rc) return rc; }
    State &s = h->st;
    const size_t n = (size_t)h->n;
    HIPCHK(h, hipStreamSynchronize(s.stream));
    s.masks_valid = 0;  // state edited from outside
    float4 *vdst = nullptr;
    bool w_only = false;
    switch (field) {
        case SPH_F_POSITION: vdst = s.posv.cur(); break;
        case SPH_F_VELOCITY: vdst = s.velm.cur(); break;
        case SPH_F_REST_VOLUME: vdst = s.posv.cur(); w_only = true; break;
        case SPH_F_MASS: vdst = s.velm.cur(); w_only = true; h->fluid_mass_uniform = false; refresh_counts(h); break;
        case SPH_F_CG_X: vdst = s.cg_x; break;
        default: break;
    }
    if (vdst) {
        const size_t need = w_only ? n * 4 : n * 12;
        if (bytes != need) return fail(h, SPH_ERR_INVALID, "upload: field %d needs %zu bytes", field, need);
        std::vector<float4> tmp(n);
        HIPCHK(h, hipMemcpy(tmp.data(), vdst, n * sizeof(float4), hipMemcpyDeviceToHost));
        const float *f = (const float *)src;
        const int ix0 = h->perm[0], ix1 = h->perm[1], ix2 = h->perm[2];   // scene frame -> library frame
        for (size_t i = 0; i < n; ++i) {
            if (w_only) tmp[i].w = f[i];
            else { tmp[i].x = f[3 * i + ix0]; tmp[i].y = f[3 * i + ix1]; tmp[i].z = f[3 * i + ix2]; }
        }
        HIPCHK(h, hipMemcpy(vdst, tmp.data(), n * sizeof(float4), hipMemcpyHostToDevice));
        return SPH_OK;
    }
    if (field == SPH_F_PARTICLE_ID) {  // global ids when a scene is split over ranks
        if (bytes != n * 4) return fail(h, SPH_ERR_INVALID, "upload: size mismatch");
        h->L->ensure_color(s); s.color_home_ok = 0; HIPCHK(h, hipStreamSynchronize(s.stream));
        HIPCHK(h, hipMemcpy(s.pid.cur(), src, n * 4, hipMemcpyHostToDevice));
        return SPH_OK;
    }
    float *fdst = nullptr;
    switch (field) {
        case SPH_F_DENSITY: fdst = s.rho.cur(); break;
        case SPH_F_PRESSURE: fdst = s.prs; break;
        default: break;
    }
    if (fdst) {
        if (bytes != n * 4) return fail(h, SPH_ERR_INVALID, "upload: size mismatch");
        HIPCHK(h, hipMemcpy(fdst, src, n * 4, hipMemcpyHostToDevice));
        return SPH_OK;
    }
    return fail(h, SPH_ERR_UNSUPPORTED, "upload: field %d is read-only", field);
}

// ---------------------------------------------------------------------------------- frame export (host code)
// replaces ti.tools.PLYWriter(...).add_vertex_pos(...).export_ascii(path) of run_simulation.py:139-144: n vertices, xyz f32[n][3]
extern "C" int sph_write_ply_ascii(const char *path, const float *xyz, int64_t n) {
    const int rc = sphexp::write_ply_ascii(path, xyz, n);
    return rc == 0 ? SPH_OK : (rc == -1 ? SPH_ERR_INVALID : SPH_ERR_UNSUPPORTED);
}
// str(np.float32(v)) into out (>= 48 bytes), returns its length: the number format of the PLY body, exported for the tests
extern "C" int sph_format_f32(float v, char *out) { return out ? sphexp::format_f32(v, out) : SPH_ERR_INVALID; }

// ---------------------------------------------------------------------------------- mesh -> particles (host code)
// replaces trimesh's `mesh.voxelized(pitch).fill().points` (base_container.py:641-642).  Call with out == NULL to get the
// count, then with a buffer of 3 * count floats.
extern "C" int sph_voxelize_mesh(const double *vertices, int n_vertices, const int32_t *faces, int n_faces, double pitch,
                                 float *out_xyz, int64_t capacity_points, int64_t *n_points) {
    if (!vertices || !faces || !n_points) return SPH_ERR_INVALID;
    for (int k = 0; k < 3 * n_faces; ++k) if (faces[k] < 0 || faces[k] >= n_vertices) return SPH_ERR_INVALID;
    std::vector<float> pts;
    if (sphvox::voxelize_fill(vertices, n_vertices, faces, n_faces, pitch, pts) != 0) return SPH_ERR_INVALID;
    *n_points = (int64_t)(pts.size() / 3);
    if (!out_xyz) return SPH_OK;
    if (capacity_points < *n_points) return SPH_ERR_CAPACITY;
    memcpy(out_xyz, pts.data(), pts.size() * sizeof(float));
    return SPH_OK;
}

// replaces `mesh.contains(points)` on the np.arange lattice of base_container.py:686-694: inside[(i*ny + j)*nz + k] = 1 if
// (xs[i], ys[j], zs[k]) lies inside the closed mesh
extern "C" int sph_points_in_mesh(const double *vertices, int n_vertices, const int32_t *faces, int n_faces, const double *xs,
                                  int nx, const double *ys, int ny, const double *zs, int nz, uint8_t *inside) {
    if (!vertices || !faces || !xs || !ys || !zs || !inside) return SPH_ERR_INVALID;
    for (int k = 0; k < 3 * n_faces; ++k) if (faces[k] < 0 || faces[k] >= n_vertices) return SPH_ERR_INVALID;
    return sphvox::contains_lattice(vertices, n_vertices, faces, n_faces, xs, nx, ys, ny, zs, nz, inside) == 0 ? SPH_OK : SPH_ERR_INVALID;
}
