// sph_kernels.hip -- kernel instantiations + launchers.  Compiled twice:
//   -DSPH_FAST=0 -ffp-contract=off   -> sph_launch_strict()  (IEEE div/sqrt, no FMA contraction)
//   -DSPH_FAST=1 -ffp-contract=fast  -> sph_launch_fast()    (v_rcp_f32 / v_sqrt_f32, FMA)
#include "sph_common.hpp"

#if SPH_FAST
#define SPH_LAUNCH_FN sph_launch_fast
#define SPH_NS sph_fast_ns
#else
#define SPH_LAUNCH_FN sph_launch_strict
#define SPH_NS sph_strict_ns
#endif

// every kernel lives in a per-build namespace so the strict and fast objects can be linked together
#include <vector>
#include <cstdio>
namespace SPH_NS {
#include "sph_device.hpp"
#include "sph_halo_defs.hpp"
#include "sph_passes.hpp"
#include "sph_solvers.hpp"
#include "sph_cg.hpp"
#include "sph_halo.hpp"

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// tile sums of the coming scan (State::scan_partial): the bank the hashers add to, or null (k_scan_reduce computes them: SPH_NO_SCAN_FOLD, and
// every slab-sharded rank -- there the fold LOSES: the arrivals take one more atomic each in a waiting kernel of 64 workgroups and a thin slab's
// waves straddle tiles; two ranks on one GPU +1.4 ... +2.3 %, eight ranks +19 %, profiles/r06_two_ranks_scanfold_ab.txt)
static int *tile_sum_bank(State &s) {
    static const bool off = getenv("SPH_NO_SCAN_FOLD") != nullptr;
    static const bool force = getenv("SPH_SCAN_FOLD_SLAB") != nullptr;
    if (off || (s.slab_active && !force)) return nullptr;
    return s.scan_partial + (size_t)s.scan_bank * (s.scan_blocks + 1) * SCAN_PARTIAL_STRIDE;
}
// the histogram is about to be taken on a cell_count that is not known to be clean: clear it, and the tile sums with it
static void clear_histogram(State &s) {
    hipMemsetAsync(s.cell_count, 0, sizeof(int) * (size_t)(s.c.G + SPH_NGRAVE + 1), s.stream);
    hipMemsetAsync(s.scan_partial + (size_t)s.scan_bank * (s.scan_blocks + 1) * SCAN_PARTIAL_STRIDE, 0, sizeof(int) * (size_t)(s.scan_blocks + 1) * SCAN_PARTIAL_STRIDE, s.stream);
}

// Run lists of the deterministic sort (RunList, sph_common.hpp): what a hasher of the COMING sort files its runs with, or an off list.
// Unsharded scenes only (a slab's arrivals take their slots one by one, in kernels of their own); SPH_NO_RUN_LISTS=1: never allocated.
// open_epoch: a new histogram begins (the lists of every earlier one die with their epoch).
static RunList run_list_of(State &s, bool open_epoch) {
    if (!s.run_head || s.slab_active) { s.run_lists_filed = 0; return RunList{nullptr, nullptr, 0, 0u}; }
    if (open_epoch) { if (++s.sort_epoch == 0u) s.sort_epoch = 1u; }
    return RunList{s.run_head, s.run_rec, s.cap, s.sort_epoch};
}

void l_hash_count(State &s) {
    const int n = s.c.n;
    if (s.prehashed) {   // the last step's force pass has hashed for this sort (NextHash): cell ids, histogram and ranks are in place
        s.prehashed = 0;
        s.cell_count_clean = 0;
        s.n_prehashed_sorts++;
        return;   // (tile_sums_ready and hist_taken were set by the pass that hashed)
    }
    if (!s.cell_count_clean) clear_histogram(s);
    s.cell_count_clean = 0;
    int *ts = tile_sum_bank(s);
    s.tile_sums_ready = ts != nullptr;
    s.hist_taken = 1;
    s.run_lists_filed = 0;
    if (n == 0) return;
    s.n_hash_launches++;
    const RunList rl = run_list_of(s, true);
    hipLaunchKernelGGL(k_hash_count, dim3(cdiv(n, 256)), dim3(256), 0, s.stream, s.c, s.posv.cur(), s.cellid,
                       s.rank, s.cell_count, s.slab_active ? s.meta.cur() : nullptr, ts, rl);
    s.run_lists_filed = rl.head != nullptr;
}

void l_scan(State &s) {
    //  the scan must follow a histogram of THIS sort: taken by l_hash_count, by the pass that hashed ahead (NextHash), or by
    // the slab kernels.  Anything else is a bug in the step orchestration; the sort would file every particle into cell 0.
    if (!s.hist_taken) s.state_error |= 1;
    s.hist_taken = 0;
    const int G = s.c.G + (s.slab_active ? SPH_NGRAVE : 0);   // + graveyard cells
    int nb = cdiv(G, SCAN_TILE);                      // <= s.scan_blocks (sized for the global grid)
    if (nb < SPH_STAT_SLOTS / SCAN_TPB) nb = SPH_STAT_SLOTS / SCAN_TPB;   // k_scan_final also clears the statistics slots
    int *part = s.scan_partial + (size_t)s.scan_bank * (s.scan_blocks + 1) * SCAN_PARTIAL_STRIDE;
    int *part_next = s.scan_partial + (size_t)(1 - s.scan_bank) * (s.scan_blocks + 1) * SCAN_PARTIAL_STRIDE;
    // the tile sums: left by whoever took the histogram (one launch less per sort, round 6), else by k_scan_reduce
    if (!s.tile_sums_ready) hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(SCAN_TPB), 0, s.stream, s.cell_count, G, part);
    hipLaunchKernelGGL(k_scan_final, dim3(nb), dim3(SCAN_TPB), 0, s.stream, s.cell_count, G, part,
                       s.cell_start, s.c.n, s.scal, 1 - s.c.stat_bank, s.c.n_dev, part_next,
                       (s.scan_tile_state && !s.slab_active) ? s.scan_tile_state : nullptr);   // (a slab's grid moves with its cuts: every tile scanned)
    s.cell_count_clean = 1;   // ... and so is the bank of tile sums the next histogram adds to
    s.scan_bank = 1 - s.scan_bank;
    s.tile_sums_ready = 0;
}

// per-workgroup header + lane permutation of the neighbour passes (k_block_prep); valid until the order changes
// launch = false: k_gather_prep has done the kernel's work for this sort (l_scatter_impl); the bookkeeping behind it is the same
void l_block_prep(State &s, bool launch = true) {
    const int n = s.c.n;
    if (n == 0) return;
    const bool lst = !s.c.all_fluid && s.blk_list;
    BlockPrepTables tabs;
    for (int k = 0; k < 8; ++k) tabs.tab[k] = s.halo_tab[k];
    const int *xi = (s.slab_active && s.tables_pending) ? s.xidx[s.xcur] : nullptr;   // push transport: slot tables built here (l_halo_build_tables)
    s.tables_pending = 0;
    TilePlanOut plan{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, nullptr};
    s.tile_plan_n = -1;
    if (s.slab_active && s.tile_list[0]) {   // boundary / interior tile lists of this sort (compute / halo overlap)
        // boundary launches get last sort's number of boundary tiles + a quarter + 64 workgroups (an empty workgroup still costs its LDS /
        // register allocation: a launch of every tile that mostly exits was measured at +20 us per step)
        const int nt = cdiv(n, 256);
        int known = s.push.mirror ? ((volatile SlabDyn *)s.push.mirror)->n_btiles : -1;
        if (known < 0) known = nt;   // no sort with a plan has finished on the device yet (the mirror starts at -1): every tile
        s.tile_bound_b = std::min(nt, known + known / 4 + 64);
        plan = TilePlanOut{s.tile_list[0], s.tile_list[1], s.tile_cnt, s.tile_class, s.has_down ? 3 : 0, s.has_up ? 3 : 0, s.tile_bound_b,
                           s.dyn_cur ? &s.dyn_cur->status : nullptr, s.push.mirror ? &((volatile SlabDyn *)s.push.mirror)->n_btiles : nullptr};
        s.tile_plan_n = n;
    }
    if (launch)
    hipLaunchKernelGGL(k_block_prep, dim3(cdiv(n, 256)), dim3(256), 0, s.stream, s.c, s.posv.cur(), s.meta.cur(), s.cell_start,
                       s.blk_hdr, s.lane_perm, lst ? s.blk_flag : nullptr, xi, tabs, plan,
                       reinterpret_cast<unsigned *>(s.blk_hdr + (size_t)((s.cap + 255) / 256) * BLK_HDR_INTS));   // cell words behind the headers
    if (lst) {
        hipLaunchKernelGGL(k_compact_blocks, dim3(1), dim3(256), 0, s.stream, s.blk_flag, cdiv(n, 256), s.blk_list, s.blk_count, s.list_count_pinned);   // (tiles past the live count carry flag 0)
        s.list_count_known = -1;
        if (s.list_count_event) hipEventRecord(s.list_count_event, s.stream);
    }
    s.list_n = lst ? n : -1;
    s.perm_n = n;
}

void l_ensure_color(State &s) {
    if (!s.color_stale) return;
    s.color_stale = 0;
    if (s.c.n > 0) hipLaunchKernelGGL(k_color_from_home, dim3(cdiv(s.c.n, 256)), dim3(256), 0, s.stream, s.c.n, s.pid.cur(), s.color_home, s.color.cur());
}

void l_scatter_impl(State &s, bool stable) {
    const int n = s.c.n;
    if (n == 0) { s.sort_skip_rho = 0; return; }
    SortArrays a;
    a.G = s.c.G;
    a.posv_in = s.posv.cur(); a.posv_out = s.posv.alt();
    a.velm_in = s.velm.cur(); a.velm_out = s.velm.alt();
    a.meta_in = s.meta.cur(); a.meta_out = s.meta.alt();
    a.pid_in = s.pid.cur(); a.pid_out = s.pid.alt();
    a.color_in = s.color.cur(); a.color_out = s.color.alt();
    a.rho_in = s.rho.cur(); a.rho_out = s.rho.alt();
    a.orig_in = s.orig.cur(); a.orig_out = s.orig.alt();
    a.xidx_in = s.slab_active ? s.xidx[s.xcur] : nullptr; a.xidx_out = s.slab_active ? s.xidx[1 - s.xcur] : nullptr;
    // deterministic sort of an unsharded scene whose hashers filed their runs into per-cell lists: rank from the lists, then a gather by
    // destination tile that prepares the tile for the neighbour passes as it goes -- 2 launches instead of 3 (k_scatter_index, k_scatter, k_block_prep)
    const bool by_lists = stable && s.run_lists_filed && s.run_head && s.sort_inv && !s.slab_active;
    if (!by_lists) l_ensure_color(s);   // (k_scatter moves State::color)
    s.run_lists_filed = 0;
    const bool skip_rho = by_lists && s.sort_skip_rho && s.c.all_fluid;
    s.sort_skip_rho = 0;
    if (by_lists) {
        // bytes that need not move: the colours (at home, keyed by the particle id, while the ids are the append order) and a density that
        // the next kernel recomputes for every particle
        static const bool move_all = getenv("SPH_SORT_MOVE_ALL") != nullptr;   // A/B
        if (s.color_home && s.color_home_ok && !move_all) { a.color_in = nullptr; s.color_stale = 1; }
        if (skip_rho && !move_all) a.rho_in = nullptr;
        hipLaunchKernelGGL(k_sort_rank, dim3(cdiv(n, 256)), dim3(256), 0, s.stream, n, s.cellid, s.cell_start,
                           RunList{s.run_head, s.run_rec, s.cap, s.sort_epoch}, s.sort_inv);
        const bool lst = !s.c.all_fluid && s.blk_list;
        hipLaunchKernelGGL(k_gather_prep, dim3(cdiv(n, 256)), dim3(256), 0, s.stream, s.c, n, s.sort_inv, a, s.cell_start,
                           s.blk_hdr, s.lane_perm, lst ? s.blk_flag : nullptr,
                           reinterpret_cast<unsigned *>(s.blk_hdr + (size_t)((s.cap + 255) / 256) * BLK_HDR_INTS));
        s.n_list_sorts++;
    } else if (stable) {
        hipLaunchKernelGGL(k_scatter_index, dim3(cdiv(n, 256)), dim3(256), 0, s.stream, n, s.cellid, s.rank,
                           s.cell_start, (int2 *)s.tmp_idx, s.c.n_dev);
        hipLaunchKernelGGL(k_scatter<true>, dim3(cdiv(n, 256)), dim3(256), 0, s.stream, n, s.cellid, s.rank,
                           s.cell_start, (const int2 *)s.tmp_idx, a, s.c.n_dev);
    } else {
        hipLaunchKernelGGL(k_scatter<false>, dim3(cdiv(n, 256)), dim3(256), 0, s.stream, n, s.cellid, s.rank,
                           s.cell_start, (const int2 *)s.tmp_idx, a, s.c.n_dev);
    }
    s.posv.flip(); s.velm.flip(); s.meta.flip(); s.pid.flip(); s.color.flip(); s.rho.flip();
    s.masks_valid = 0;  // new order, new candidate runs
    if (!s.slab_active) l_block_prep(s, !by_lists);
    else s.perm_n = s.list_n = -1;   // slab sharding: rebuilt once the dead particles behind the live ones are dropped (launch_pass)
    if (s.orig.cur()) s.orig.flip();
    if (s.slab_active) s.xcur = 1 - s.xcur;
}
void l_scatter(State &s) { l_scatter_impl(s, false); }
void l_scatter_stable(State &s) { l_scatter_impl(s, true); }

// Workgroups a launch over the list of fluid-holding tiles needs: the list's length once the host has seen it (State::list_count_pinned),
// one per tile of the scene until then (the kernels send the surplus home at their top).  Never zero: a functor whose prologue keeps a
// solver loop's books needs workgroup (0, 0) even when the list is empty.
int list_grid(State &s, int nb) {
    static const bool off = getenv("SPH_NO_LIST_GRID") != nullptr;
    if (off || !s.list_count_pinned) return nb;
    if (s.list_count_known < 0 && s.list_count_event) {
        const hipError_t q = hipEventQuery(s.list_count_event);
        if (q == hipSuccess) s.list_count_known = *s.list_count_pinned;
        else if (q == hipErrorNotReady && hipPeekAtLastError() == hipErrorNotReady) (void)hipGetLastError();   // "not yet" is no launch failure (check_async, sph_api.hip)
    }
    const int g = s.list_count_known >= 0 ? s.list_count_known : nb;
    return g < 1 ? 1 : (g > nb ? nb : g);
}

// mask_mode: 0 compute, 1 compute + store (first pass after a sort), 2 reuse (see process_run)
template <class P> void launch_pass(State &s, const P &p, int mask_mode = 0) {
    const int n = s.c.n;
    if (n == 0) return;
    const int nb = cdiv(n, P::BLOCK);
    if (!s.nbr_mask || s.c.force_global == 1) mask_mode = 0;
    if (mask_mode == 2 && !s.masks_valid) mask_mode = 0;
    // lane permutation (k_lane_perm): only for the passes that reuse stored masks -- a pass that runs phase 1 keeps
    // neighbouring lanes on neighbouring cells, which is what makes its LDS reads conflict-free
    if (s.perm_n != n) l_block_prep(s);   // particles were appended since the last sort
    if (!(PassModes<P>::value & (1 << mask_mode))) mask_mode = 0;
    const unsigned char *perm = (mask_mode == 2 && s.lane_perm) ? s.lane_perm : nullptr;
    // workgroups without fluid are not launched for functors that have nothing to do there
    const bool use_list = PassFluidOnly<P>::value && !s.c.all_fluid && s.list_n == n && s.c.force_global == 0;
    const int *bl = use_list ? s.blk_list : nullptr, *bc = use_list ? s.blk_count : nullptr;
    int nb_launch = nb;
    const unsigned char *skip = nullptr;
    if (s.tile_sel && s.tile_plan_n == n && !use_list) {   // one set of the slab's tiles only (compute / halo overlap)
        if (s.tile_sel == 1) { bl = s.tile_list[0]; bc = s.tile_cnt; nb_launch = s.tile_bound_b > 0 ? s.tile_bound_b : 1; }   // the listed boundary tiles; the grid is a bound
        else skip = s.tile_class;   // every tile, the boundary ones leave
    }
    if (use_list) nb_launch = list_grid(s, nb);
    if (P::HAS_REDUCE) s.last_pass_listed = use_list ? 1 : 0;   // whose partial sums l_reduce_sum will finish
    unsigned long long *tl = (s.c.force_global == 20 && (size_t)nb * 16 * 8 <= (size_t)s.cap * 4) ? (unsigned long long *)s.tmp_idx : nullptr;
    if (tl) hipMemsetAsync(tl, 0, (size_t)nb * 16 * 8, s.stream);
    constexpr int MODES = PassModes<P>::value;
    if (!(MODES & (1 << mask_mode))) mask_mode = 0;   // every functor has mode 0
    const int gy = (PassSplit<P>::value && s.split_next_pass) ? s.split_next_pass : 1;   // 3: one workgroup per (tile, x-offset group); 2: groups {0, 1} / {2}
    s.split_next_pass = 0;
    // debug ("time against resident workgroups"): SPH_DEBUG_EXTRA_LDS=<bytes> of unused dynamic LDS per workgroup lower the
    // number of workgroups a CU can hold without touching the code
    static const int extra_lds = getenv("SPH_DEBUG_EXTRA_LDS") ? atoi(getenv("SPH_DEBUG_EXTRA_LDS")) : 0;
#define SPH_LAUNCH_NBR(M) hipLaunchKernelGGL((k_nbr_pass<P, M>), dim3(nb_launch, gy), dim3(P::BLOCK), extra_lds, s.stream, s.c, s.cell_start, p, s.scal, nb, s.nbr_mask, s.nbr_mask_hi, s.cap, s.blk_hdr, perm, tl, s.loop_flag, bl, bc, skip)
    if (mask_mode == 1) {
        if constexpr ((MODES & 0b010) != 0) { SPH_LAUNCH_NBR(1); s.masks_valid = 1; }
    } else if (mask_mode == 2) {
        if constexpr ((MODES & 0b100) != 0) { SPH_LAUNCH_NBR(2); }
    } else {
        SPH_LAUNCH_NBR(0);
    }
#undef SPH_LAUNCH_NBR
    if (tl) {   // debug: mean shader-clock deltas between the phase stamps of k_nbr_pass (tmp_idx is free between sorts)
        static int shown = 0;
        hipStreamSynchronize(s.stream);
        std::vector<unsigned long long> hbuf((size_t)nb * 16);
        hipMemcpy(hbuf.data(), tl, hbuf.size() * 8, hipMemcpyDeviceToHost);
        if (shown++ % 8 < 2) {
            // stamps: 0 start, 1 after prologue barrier, per group g: 2+4g tile staged, 3+4g masks ready (merged path only),
            // 4+4g group computed, 5+4g end-of-group barrier passed; 14 after the loop, 15 after finish()
            double d[16] = {0}; double cnt[16] = {0};
            unsigned long long tmin = ~0ull, tmax = 0;
            for (int b = 0; b < nb; ++b) {
                const unsigned long long *t = &hbuf[(size_t)b * 16];
                int prev = 0;
                for (int k = 1; k < 16; ++k) if (t[k]) { d[k] += (double)(t[k] - t[prev]); cnt[k] += 1; prev = k; }
                if (t[0] && t[0] < tmin) tmin = t[0];
                if (t[15] > tmax) tmax = t[15];
            }
            double life = 0; for (int b = 0; b < nb; ++b) life += (double)(hbuf[(size_t)b * 16 + 15] - hbuf[(size_t)b * 16]);
            fprintf(stderr, "timeline mode %d BT %d: kernel %.1f kclk, workgroup lifetime %.1f kclk; mean clk per stamp:", mask_mode, (int)sizeof(typename P::BT), (tmax - tmin) * 1e-3, life / nb * 1e-3);
            for (int k = 1; k < 16; ++k) fprintf(stderr, " [%d]%.0f", k, cnt[k] ? d[k] / cnt[k] : 0.0);
            fprintf(stderr, "\n");
        }
    }
}

void l_density(State &s, int eos) {
    HaloFieldSend fs = s.fieldsend;
    if (!eos) fs.on = 0;
    s.fieldsend.on = 0;
    const int sp = s.density_books_forces ? 1 + 3 : 1, se = s.density_books_forces ? 2 : 1;   // (WcsphForcePass: PAIR_WEIGHT 3, one evaluation per pair)
    if (s.c.all_fluid) {
        if (eos) { DensityPass<true, true> p{s.posv.cur(), s.meta.cur(), s.rho_raw, s.rho.cur(), s.prs, s.ptm, fs, sp, se}; launch_pass(s, p, 1); }
        else { DensityPass<true, false> p{s.posv.cur(), s.meta.cur(), s.rho_raw, s.rho.cur(), s.prs, s.ptm, fs, sp, se}; launch_pass(s, p, 1); }
    } else {
        if (eos) { DensityPass<false, true> p{s.posv.cur(), s.meta.cur(), s.rho_raw, s.rho.cur(), s.prs, s.ptm, fs, sp, se}; launch_pass(s, p, 1); }
        else { DensityPass<false, false> p{s.posv.cur(), s.meta.cur(), s.rho_raw, s.rho.cur(), s.prs, s.ptm, fs, sp, se}; launch_pass(s, p, 1); }
    }
}

// rho_src: WCSPH viscosity reads the unclamped density (rho_raw); DFSPH/PCISPH read particle_densities.
void l_non_pressure(State &s) {
    const float *rho_src = s.visc_rho_raw ? s.rho_raw : s.rho.cur();
    if (s.c.all_fluid) {
        NonPressurePass<true> p{s.posv.cur(), s.velm.cur(), s.meta.cur(), rho_src, s.velm.alt(), s.scal, s.pose, s.c.rho0, s.skip_viscosity, s.np_acc_out, s.np_visc_vel};
        launch_pass(s, p, 2);
    } else {
        NonPressurePass<false> p{s.posv.cur(), s.velm.cur(), s.meta.cur(), rho_src, s.velm.alt(), s.scal, s.pose, s.c.rho0, s.skip_viscosity, s.np_acc_out, s.np_visc_vel};
        launch_pass(s, p, 2);
    }
    s.velm.flip();
}

// base_solver.py:660-666 emitter branch of update_fluid_position, run after the position pass so
// that no neighbour sees a material flip mid-kernel.
__global__ void __launch_bounds__(256)
k_emitter_advance(const Consts c, float4 *posv, const float4 *velm, int *meta, const RigidPose *pose) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= live_n(c)) return;
    const int m = meta[i];
    if (META_MAT(m) == 1) return;
    float4 p = posv[i];
    if (up_coord(c, p) > c.g_upper) {
        const int obj = META_OBJ(m);
        if (obj >= 0 && pose->material[obj] == 1) {
            const float4 v = velm[i];
            p.x += c.dt * v.x; p.y += c.dt * v.y; p.z += c.dt * v.z;
            posv[i] = p;
            if (up_coord(c, p) <= c.g_upper) meta[i] = META_SET_MAT(m, 1);
        }
    }
}

void l_pressure_integrate(State &s) {
    if (s.c.all_fluid) {
        PressurePass<true> p{s.posv.cur(), s.meta.cur(), s.ptm, s.prs, s.rho.cur(), s.velm.cur(), s.acc, s.posv.alt(), s.scal, s.pose, s.c.rho0, 1};
        launch_pass(s, p, 2);
    } else {
        PressurePass<false> p{s.posv.cur(), s.meta.cur(), s.ptm, s.prs, s.rho.cur(), s.velm.cur(), s.acc, s.posv.alt(), s.scal, s.pose, s.c.rho0, 1};
        launch_pass(s, p, 2);
    }
    s.posv.flip();
    s.masks_valid = 0;  // positions moved
    if (s.has_emitter && s.c.n > 0)
        hipLaunchKernelGGL(k_emitter_advance, dim3(cdiv(s.c.n, 256)), dim3(256), 0, s.stream, s.c, s.posv.cur(),
                           s.velm.cur(), s.meta.cur(), s.pose);
}

// WCSPH.py:30-36, 45 as one pass (see WcsphForcePass); same buffer choreography as the two passes it replaces
void l_wcsph_forces(State &s) {
    // this pass as the next step's k_hash_count (NextHash): only where the histogram is clean (the scan cleared it behind itself) and
    // every particle is an active fluid particle of an unsharded scene (wcsph_step decides whether another step follows untouched)
    NextHash nh{0, s.cellid, s.rank, s.cell_count, tile_sum_bank(s), RunList{nullptr, nullptr, 0, 0u}};
    if (s.nexthash.on && s.c.all_fluid && !s.slab_active && s.cell_count_clean && s.density_books_forces && s.c.n > 0) nh.on = 1;
    s.nexthash.on = 0;
    if (nh.on) nh.rl = run_list_of(s, true);
    if (!s.density_books_forces) {   // launched outside wcsph_step's density + forces pair: nobody has booked this walk's pairs
        if (s.c.all_fluid) {
            WcsphForcePass<true, false, true> p{s.posv.cur(), s.velm.cur(), s.meta.cur(), s.rho_raw, s.ptm, s.prs, s.rho.cur(), s.velm.alt(), s.acc, s.posv.alt(), s.scal, s.pose, s.c.rho0, s.presend, nh};
            launch_pass(s, p, 2);
        } else {
            WcsphForcePass<false, false, true> p{s.posv.cur(), s.velm.cur(), s.meta.cur(), s.rho_raw, s.ptm, s.prs, s.rho.cur(), s.velm.alt(), s.acc, s.posv.alt(), s.scal, s.pose, s.c.rho0, s.presend, nh};
            launch_pass(s, p, 2);
        }
    } else
#if SPH_FAST
    if (s.c.all_fluid && s.uniform_mass && !s.slab_active) {   // one fluid mass in the whole scene: the instantiation with the mass products hoisted (same sums)
        WcsphForcePass<true, true> p{s.posv.cur(), s.velm.cur(), s.meta.cur(), s.rho_raw, s.ptm, s.prs, s.rho.cur(), s.velm.alt(), s.acc, s.posv.alt(), s.scal, s.pose, s.c.rho0, s.presend, nh};
        launch_pass(s, p, 2);
    } else
#endif
    if (s.c.all_fluid) {
        WcsphForcePass<true> p{s.posv.cur(), s.velm.cur(), s.meta.cur(), s.rho_raw, s.ptm, s.prs, s.rho.cur(), s.velm.alt(), s.acc, s.posv.alt(), s.scal, s.pose, s.c.rho0, s.presend, nh};
        launch_pass(s, p, 2);
    } else {
        WcsphForcePass<false> p{s.posv.cur(), s.velm.cur(), s.meta.cur(), s.rho_raw, s.ptm, s.prs, s.rho.cur(), s.velm.alt(), s.acc, s.posv.alt(), s.scal, s.pose, s.c.rho0, s.presend, nh};
        launch_pass(s, p, 2);
    }
    if (s.presend.on) { s.presend.on = 0; s.preclassified = 1; }
    if (nh.on) { s.prehashed = 1; s.cell_count_clean = 0; s.tile_sums_ready = nh.tile_sum != nullptr; s.hist_taken = 1; s.run_lists_filed = nh.rl.head != nullptr; }
    s.velm.flip();
    s.posv.flip();
    s.masks_valid = 0;  // positions moved
    if (s.has_emitter && s.c.n > 0)
        hipLaunchKernelGGL(k_emitter_advance, dim3(cdiv(s.c.n, 256)), dim3(256), 0, s.stream, s.c, s.posv.cur(),
                           s.velm.cur(), s.meta.cur(), s.pose);
}

void l_rigid_volume(State &s) {
    RigidVolumePass p{s.posv.cur(), s.velm.cur(), s.meta.cur()};
    launch_pass(s, p);
}

// base_solver.py:616 _renew_rigid_particle_state
__global__ void __launch_bounds__(256)
k_renew_rigid(const Consts c, float4 *posv, float4 *velm, const float4 *orig, const int *meta, const RigidPose *pose) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= live_n(c)) return;
    const int m = meta[i];
    if (META_MAT(m) != 2 || !META_DYN(m)) return;
    const int o = META_OBJ(m);
    if (o < 0 || !pose->is_dynamic[o]) return;
    const float4 q0 = orig[i];
    const float qx = q0.x - pose->com0[o][0], qy = q0.y - pose->com0[o][1], qz = q0.z - pose->com0[o][2];
    const float *R = pose->rot[o];
    const float px = R[0] * qx + R[1] * qy + R[2] * qz;
    const float py = R[3] * qx + R[4] * qy + R[5] * qz;
    const float pz = R[6] * qx + R[7] * qy + R[8] * qz;
    float4 p = posv[i], v = velm[i];
    p.x = pose->com[o][0] + px; p.y = pose->com[o][1] + py; p.z = pose->com[o][2] + pz;
    const float wx = pose->angvel[o][0], wy = pose->angvel[o][1], wz = pose->angvel[o][2];
    v.x = pose->vel[o][0] + (wy * pz - wz * py);
    v.y = pose->vel[o][1] + (wz * px - wx * pz);
    v.z = pose->vel[o][2] + (wx * py - wy * px);
    posv[i] = p; velm[i] = v;
}

void l_renew_rigid(State &s) {
    if (!s.has_dynamic_rigid || s.c.n == 0 || !s.orig.cur()) return;
    s.masks_valid = 0;  // rigid particles move
    hipLaunchKernelGGL(k_renew_rigid, dim3(cdiv(s.c.n, 256)), dim3(256), 0, s.stream, s.c, s.posv.cur(),
                       s.velm.cur(), s.orig.cur(), s.meta.cur(), s.pose);
}

// base_solver.py:670 prepare_emitter
__global__ void __launch_bounds__(256) k_prepare_emitter(const Consts c, const float4 *posv, int *meta) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= live_n(c)) return;
    const int m = meta[i];
    if (META_MAT(m) == 1 && up_coord(c, posv[i]) > c.g_upper) meta[i] = META_SET_MAT(m, 2);
}

void l_prepare_emitter(State &s) {
    if (!s.has_emitter || s.c.n == 0) return;
    hipLaunchKernelGGL(k_prepare_emitter, dim3(cdiv(s.c.n, 256)), dim3(256), 0, s.stream, s.c, s.posv.cur(),
                       s.meta.cur());
}

// Particles appended mid-step by the host (base_container.py:212 insert_object, called from _step()): what the rest of
// the reference's step does to them.  Fluid: enforce_domain_boundary_3D (:575).  Rigid, WCSPH / PCISPH only: step()'s
// compute_rigid_particle_volume (:696) runs on the grid of the last sort, which does not contain them, so the sum
// over same-object neighbours is empty: V = 1 / W(0), m = rho0 V (:113-:115).
__global__ void __launch_bounds__(256)
k_post_insert(const Consts c, int first, float4 *posv, float4 *velm, const int *meta, int stale_volume) {
    const int i = first + blockIdx.x * 256 + threadIdx.x;
    if (i >= live_n(c)) return;
    const int m = meta[i];
    float4 p = posv[i], v = velm[i];
    if (META_MAT(m) == 1) {
        if (META_DYN(m)) { enforce_boundary(c, p.x, p.y, p.z, v.x, v.y, v.z); posv[i] = p; velm[i] = v; }
    } else if (stale_volume && META_MAT(m) == 2 && up_coord(c, p) <= c.g_upper) {
        const float V = 1.0f / c.W0;
        p.w = V; v.w = c.rho0 * V;
        posv[i] = p; velm[i] = v;
    }
}
void l_post_insert(State &s, int first, int stale_volume) {
    if (s.c.n <= first) return;
    hipLaunchKernelGGL(k_post_insert, dim3(cdiv(s.c.n - first, 256)), dim3(256), 0, s.stream, s.c, first, s.posv.cur(),
                       s.velm.cur(), s.meta.cur(), stale_volume);
}
__global__ void __launch_bounds__(256) k_clear_fresh(int n, int *meta) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const int m = meta[i]; if (m & META_FRESH_BIT) meta[i] = m & ~META_FRESH_BIT; }
}
void l_clear_fresh(State &s) {
    if (s.c.n > 0) hipLaunchKernelGGL(k_clear_fresh, dim3(cdiv(s.c.n, 256)), dim3(256), 0, s.stream, s.c.n, s.meta.cur());
}

#include "sph_solvers_impl.hpp"
#include "sph_halo_impl.hpp"
}  // namespace SPH_NS

using namespace SPH_NS;

const Launch *SPH_LAUNCH_FN() {
    static Launch L;
    static bool init = false;
    if (!init) {
        L.ensure_color = l_ensure_color;
        L.hash_count = l_hash_count;
        L.scan = l_scan;
        L.scatter = l_scatter;
        L.scatter_stable = l_scatter_stable;
        L.density = l_density;
        L.non_pressure = l_non_pressure;
        L.pressure_integrate = l_pressure_integrate;
        L.wcsph_forces = l_wcsph_forces;
        L.rigid_volume = l_rigid_volume;
        L.renew_rigid = l_renew_rigid;
        L.prepare_emitter = l_prepare_emitter;
        L.post_insert = l_post_insert;
        L.clear_fresh = l_clear_fresh;
        register_solver_launchers(L);
        L.halo_classify_pack = l_halo_classify_pack; L.halo_unpack_append = l_halo_unpack_append;
        L.halo_build_tables = l_halo_build_tables; L.halo_pack_fields = l_halo_pack_fields;
        L.halo_unpack_fields = l_halo_unpack_fields;
        L.halo_pack_scalar = l_halo_pack_scalar; L.halo_unpack_scalar = l_halo_unpack_scalar;
        L.halo_pack_vel = l_halo_pack_vel; L.halo_unpack_vel = l_halo_unpack_vel;
        L.loop_criterion = l_loop_criterion;
        L.halo_unpack2 = l_halo_unpack2; L.halo_push_fields = l_halo_push_fields;
        L.halo_pull_fields = l_halo_pull_fields; L.halo_selftest = l_halo_selftest; L.halo_presend_begin = l_halo_presend_begin; L.halo_fieldsend_begin = l_halo_fieldsend_begin;
        L.layer_hist = l_layer_hist;
        L.count_ghosts = l_count_ghosts;
        init = true;
    }
    return &L;
}
