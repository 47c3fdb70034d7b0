// sph_steps.hpp -- per-method step orchestration (included by sph_api.hip).
// Order of operations follows WCSPH.py:27, DFSPH.py:298, PCISPH.py:165 and base_solver.py:692.
#pragma once
#include <sched.h>

// reads scal->red[slot] (one small D2H copy + stream sync: the reference's python loops read one
// scalar per iteration too, DFSPH.py:150, :236; PCISPH.py:122)
static int read_red(SphHandle *h, int slot, float *out) {
    HIPCHK(h, hipMemcpyAsync(&h->scal_h->red[slot], &h->st.scal->red[slot], sizeof(float), hipMemcpyDeviceToHost, h->st.stream));
    HIPCHK(h, hipStreamSynchronize(h->st.stream));
    *out = h->scal_h->red[slot];
    return SPH_OK;
}

static int implicit_viscosity_non_pressure(SphHandle *h);

// Read-back of a solver loop's batch (round 6).  hipMemcpyAsync D2H + hipStreamSynchronize cost 35-75 us of idle GPU per batch (the copy is a
// kernel of the runtime, the synchronise sleeps on an interrupt: profiles/r05_gaps_c3_motion.txt) -- two to three per solve, five to six per
// step of the buckling scene: a fifth of its step.  Instead a one-wave kernel behind the batch stores the residuals and the flags into pinned
// host memory, the batch number last (system-scope release), and the host spins on that number: the wait ends a few microseconds after the
// batch's last kernel.  (The same kernel -> pinned memory -> polling host pattern as the slab counts' mirror, sph_halo.hpp.)  Bounded: the host
// looks at hipStreamQuery every few microseconds and falls back to the copy when the stream is idle or broken.  SPH_NO_LOOP_PUBLISH=1: the copy.
struct LoopPub { float red[8]; int flags[4]; unsigned seq; unsigned pad[3]; };
static_assert(sizeof(LoopPub) == 64, "LoopPub layout");
__global__ void __launch_bounds__(64) k_publish_loop(DevScalars *scal, LoopPub *pub, unsigned seq) {
    const int t = threadIdx.x;
    const int stop = scal->flags[0];
    if (t < 8) __hip_atomic_store(&pub->red[t], scal->red[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (t < 12) __hip_atomic_store(&pub->flags[t - 8], scal->flags[t - 8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (t == 0) {
        // a loop that has stopped is over (the host launches nothing more for it): its flags are reset here, so that the next loop needs
        // no memset in front of it (a fill kernel + a launch boundary per solve)
        if (stop) { scal->flags[0] = 0; scal->flags[1] = 0; }
        __hip_atomic_store(&pub->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// host side of a publish: spin until the kernel's number shows up in pinned memory; false: the stream went idle or broke without it
static bool spin_for(State &s, const volatile unsigned *seq, unsigned want) {
    for (unsigned spins = 0;; ++spins) {
        if (__atomic_load_n(seq, __ATOMIC_ACQUIRE) == want) return true;
        if ((spins & 255u) == 255u) {
            const hipError_t q = hipStreamQuery(s.stream);
            if (q != hipErrorNotReady) return q == hipSuccess && __atomic_load_n(seq, __ATOMIC_ACQUIRE) == want;   // idle: one more look; or broken
            if (hipPeekAtLastError() == hipErrorNotReady) (void)hipGetLastError();   // "not yet" is no failure (check_async)
            if (spins > 4096u) sched_yield();   // a long wait (many steps queued): let whoever else wants this core have it; returns at once otherwise
        }
        __builtin_ia32_pause();
    }
}
// The pair statistics of one bank (3 x SPH_STAT_SLOTS striped 64-bit counters) added up on the device and published the same way: a D2H copy of
// the 97 KB DevScalars issued on an IDLE stream costs ~0.3 ms on this runtime (profiles/r06_sync_latency.txt)
struct StatsPub { unsigned long long pairs, evals, fallback; unsigned seq; unsigned pad[9]; };
static_assert(sizeof(StatsPub) == 64, "StatsPub layout");
__global__ void __launch_bounds__(256) k_publish_stats(const DevScalars *scal, int bank, StatsPub *pub, unsigned seq) {
    __shared__ unsigned long long s_w[3][4];
    unsigned long long a = 0, b = 0, c = 0;
    for (int k = threadIdx.x; k < SPH_STAT_SLOTS; k += 256) { a += scal->pairs[bank][k]; b += scal->evals[bank][k]; c += scal->fallback[bank][k]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); b += __shfl_down(b, o, 64); c += __shfl_down(c, o, 64); }
    if ((threadIdx.x & 63) == 0) { s_w[0][threadIdx.x >> 6] = a; s_w[1][threadIdx.x >> 6] = b; s_w[2][threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(&pub->pairs, s_w[0][0] + s_w[0][1] + s_w[0][2] + s_w[0][3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&pub->evals, s_w[1][0] + s_w[1][1] + s_w[1][2] + s_w[1][3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&pub->fallback, s_w[2][0] + s_w[2][1] + s_w[2][2] + s_w[2][3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        __hip_atomic_store(&pub->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// brings scal->red[0..8) and scal->flags[0..4) of the stream's current end into h->scal_h
static int loop_readback(SphHandle *h) {
    State &s = h->st;
    static const bool no_publish = getenv("SPH_NO_LOOP_PUBLISH") != nullptr;
    if (!no_publish && h->loop_pub) {
        const unsigned want = ++h->loop_seq;
        volatile LoopPub *pub = h->loop_pub;
        hipLaunchKernelGGL(k_publish_loop, dim3(1), dim3(64), 0, s.stream, s.scal, h->loop_pub, want);
        const bool got = spin_for(s, &pub->seq, want);
        if (got) {
            for (int k = 0; k < 8; ++k) h->scal_h->red[k] = pub->red[k];
            for (int k = 0; k < 4; ++k) h->scal_h->flags[k] = pub->flags[k];
            if (h->scal_h->flags[0] != 0) h->loop_flags_clean = true;   // (the kernel reset them behind the copy; else: as they were -- this is
                                                                        //  also the plain wait of sph_synchronize / sph_step, outside any loop)
            return SPH_OK;
        }
        h->loop_flags_clean = false;   // (whether the kernel ran is not known here)
    }
    hipError_t e = hipMemcpyAsync(h->scal_h->red, s.scal->red, 8 * sizeof(float) + 4 * sizeof(int), hipMemcpyDeviceToHost, s.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s.stream);
    if (e != hipSuccess) return fail(h, SPH_ERR_HIP, "solver loop read-back failed: %s", hipGetErrorString(e));
    return SPH_OK;
}

// particle_num the reference's residual means divide by (DFSPH.py:212, :293): the whole scene's, not this slab's
static long long dfsph_particle_num(SphHandle *h) { return h->st.slab_active ? h->comm_n_global : (long long)h->n; }

// Solver loop with the stop test on the device.  `body` launches one iteration; its last reduction kernel evaluates
// the reference's criterion (kind / denom / thr, see State::loop_kind), counts the iteration and raises
// scal->flags[0]; every kernel of a later iteration starts with a look at that flag and returns.  Iterations go out in
// batches with ONE read-back per batch instead of one error read-back per iteration (the reference, and this code before,
// synchronise with the host every iteration).  The state after the loop is the state after exactly the iteration the
// reference would have stopped at, whatever the batch sizes.
// Batch sizes (round 5): a solve takes about as many iterations as the same solve of the last step (C5: 41.5 +- 1 CG
// iterations; C3 in motion: 26.5 +- 1 density iterations), so the FIRST batch is last step's count less one and the
// loop then goes on in small batches -- two read-backs per solve instead of five to seven.  Every read-back is a
// stream drain plus the host's launch latency before the GPU has work again (~25-40 us of idle chip: the rocprofv3
// traces of round 5 show 188 us of kernels per DFSPH iteration in motion against 285 us of wall time, profiles/
// r05_rocprofv3_c3_motion_summary.txt).  Without a hint (first step, SPH_NO_LOOP_HINT): 2, 4, 8, 8, ... as before.
template <class F>
static int device_loop(SphHandle *h, int max_itr, int slot, int kind, float denom, double thr, F body, int *executed,
                       int *launched, float *last_val, void (*batch_end)(State &) = nullptr) {
    State &s = h->st;
    if (!h->loop_flags_clean) HIPCHK(h, hipMemsetAsync(&s.scal->flags[0], 0, 2 * sizeof(int), s.stream));
    h->loop_flags_clean = false;
    s.loop_flag = &s.scal->flags[0];
    s.loop_slot = slot; s.loop_kind = kind; s.loop_denom = denom; s.loop_thr = thr;
    static const bool no_hint = getenv("SPH_NO_LOOP_HINT") != nullptr;
    const int hint = (no_hint || slot < 0 || slot >= 4) ? 0 : h->loop_hint[slot];
    int n_launched = 0, batch = hint > 3 ? hint - 1 : 2, rc = SPH_OK;
    bool predicted = hint > 3;
    // red[8] and flags[4] are neighbours in DevScalars: one 48-byte copy brings the residual and the flags
    static_assert(offsetof(DevScalars, flags) == offsetof(DevScalars, red) + 8 * sizeof(float), "red / flags layout");
    while (n_launched < max_itr) {
        const int nb = batch < max_itr - n_launched ? batch : max_itr - n_launched;
        for (int k = 0; k < nb; ++k) body();
        if (batch_end) batch_end(s);
        n_launched += nb;
        rc = loop_readback(h);
        if (rc) break;
        if (h->scal_h->flags[0]) break;
        if (predicted) { batch = 2; predicted = false; }
        else if (batch < 8) batch *= 2;
    }
    s.loop_flag = nullptr;
    *executed = h->scal_h->flags[1];
    *launched = n_launched;
    *last_val = h->scal_h->red[slot];
    if (slot >= 0 && slot < 4 && rc == SPH_OK) h->loop_hint[slot] = h->scal_h->flags[1];
    return rc;
}

// base_solver.py:190 compute_non_pressure_acceleration + :643 update_fluid_velocity
static int run_non_pressure(SphHandle *h) {
    if (h->prm.viscosity_implicit) {
        int rc = implicit_viscosity_non_pressure(h);
        return rc;
    }
    ProfScope p(h, SPH_K_NON_PRESSURE);
    h->L->non_pressure(h->st);
    return SPH_OK;
}

static int wcsph_step(SphHandle *h) {
    State &s = h->st;
    // sharded: + migration / ghost exchange; over the push transport a plain WCSPH step needs nothing back from the device
    // (slab_neighbor_search_push, "async"): implicit viscosity and the unfused force passes launch exact grids instead
    const bool fused = !h->prm.viscosity_implicit && !getenv("SPH_NO_FUSED_FORCES");
    if (s.slab_active) { int rc = slab_neighbor_search(h, fused); if (rc) return rc; }
    else { s.sort_skip_rho = s.c.all_fluid; ph_neighbor_search(h); }          // WCSPH.py:28 (the density pass below rewrites every rho: the sort need not move it)
    ph_rigid_volume(h);                                                       // base_solver.py:696 (see ph_rigid_volume)
    s.density_books_forces = (fused && !getenv("SPH_FORCES_COUNT_OWN")) ? 1 : 0;   // (switch: the force pass counts its own pairs -- the counting instantiation any other caller of l_wcsph_forces gets)
    struct Unbook { State &s; ~Unbook() { s.density_books_forces = 0; } } unbook{s};
    // Sharded over the push transport, fluid only: the density pass runs the slab's BOUNDARY tiles first, their rho / p go out to the
    // neighbours, and the INTERIOR tiles (everything more than two layers from a face) run while that message is in flight; only
    // then does the stream wait for the neighbours' (SURVEY 8e "compute interior cells while halos are in flight").
    const bool overlap = s.slab_active && s.push.on && s.tile_list[0] && s.c.all_fluid && s.tile_plan_n == s.c.n && (s.has_down || s.has_up);
    if (overlap) {
        const int hint = slab_field_hint(h);
        s.tile_sel = 1;
        { ProfScope p(h, SPH_K_DENSITY); h->L->density(s, 1); }
        { ProfScope p(h, SPH_K_HALO); h->L->halo_push_fields(s, 2, nullptr, nullptr, hint); }
        s.tile_sel = 2;
        { ProfScope p(h, SPH_K_DENSITY); h->L->density(s, 1); }
        s.tile_sel = 0;
        { ProfScope p(h, SPH_K_HALO); h->L->halo_pull_fields(s, 2, nullptr, nullptr, hint); }
    } else if (s.slab_active && s.push.on && !getenv("SPH_NO_SLAB_FUSED_FIELDS")) {
        // ghost rho, p: the density pass stores the values of its boundary particles straight into the neighbours' field message
        // (HaloFieldSend) -- no gather kernel, and the message travels while the pass is still running; then the usual wait + scatter
        const int hint = slab_field_hint(h);
        h->L->halo_fieldsend_begin(s);
        { ProfScope p(h, SPH_K_DENSITY); h->L->density(s, 1); }               // :29 + :33 (EOS fused)
        { ProfScope p(h, SPH_K_HALO); h->L->halo_pull_fields(s, 2, nullptr, nullptr, hint); }
    } else {
        { ProfScope p(h, SPH_K_DENSITY); h->L->density(s, 1); }               // :29 + :33 (EOS fused)
        if (s.slab_active) { int rc = slab_exchange_fields(h); if (rc) return rc; }   // ghost rho, p
    }
    if (!h->prm.viscosity_implicit && !getenv("SPH_NO_FUSED_FORCES")) {
        // Another step of this call follows (sph_step_async(n)) and nothing on the host happens in between: the force pass classifies its
        // own particles and stores the NEXT step's message into the neighbours' inboxes as its workgroups finish (HaloSend,
        // sph_halo_defs.hpp) -- the records travel while the pass is still running, and the next step starts with the hash alone.  Not
        // with an emitter or rigid bodies (they move particles after this pass) and not before a step that re-plans the cuts.
        const SlabComm &cm = h->comm;
        const bool rebalance_next = cm.rebalance_every > 0 && (h->steps + 1) % cm.rebalance_every == 0 && !h->any_rigid_object;
        if (s.slab_active && s.push.on && h->steps_to_follow > 0 && !rebalance_next && !s.has_emitter && !h->any_rigid_object &&
            !h->sort_dirty && !getenv("SPH_NO_SLAB_PRESEND"))
            h->L->halo_presend_begin(s);
        // Unsharded, all fluid, another step of this call queued right behind (nothing can touch the particles in between): the force pass
        // hashes the positions it stores for the next step's sort (NextHash) -- one launch less per step.
        static const bool no_nexthash = getenv("SPH_NO_NEXT_HASH") != nullptr;
        s.nexthash.on = (!no_nexthash && !s.slab_active && s.c.all_fluid && h->steps_to_follow > 0 && !s.has_emitter && !h->any_rigid_object &&
                         !h->sort_dirty) ? 1 : 0;
        ProfScope p(h, SPH_K_WCSPH_FORCES); h->L->wcsph_forces(s);            // :30-31 + :34-36, :45 in one neighbour walk
        return SPH_OK;
    }
    int rc = run_non_pressure(h); if (rc) return rc;                          // :30-31
    { ProfScope p(h, SPH_K_PRESSURE_INTEGRATE); h->L->pressure_integrate(s); } // :34-36, :45
    return SPH_OK;
}

// A solve with a FIXED iteration count (SphParams::fixed_iterations: bench mode, sph_step_async) has no stop test, and nothing reads
// the residual of an iteration (SphStats::err_* report 0 in this mode): the walks still leave their per-workgroup partial sums, but the
// one-workgroup kernel that adds them up -- 5 us and a launch boundary per iteration, 2.4 % of a C3 step -- is not launched.  Unsharded
// only (a sharded solve all-reduces the residual either way).  SPH_FIXED_KEEP_RESIDUAL=1: launch it as before (A/B).
struct SkipResidual {
    State &s;
    SkipResidual(SphHandle *h) : s(h->st) {
        static const bool keep = getenv("SPH_FIXED_KEEP_RESIDUAL") != nullptr;
        s.skip_residual = (h->prm.fixed_iterations > 0 && !s.slab_active && !keep) ? 1 : 0;
    }
    ~SkipResidual() { s.skip_residual = 0; }
};

// DFSPH.py:139 correct_divergence_error
static int dfsph_divergence(SphHandle *h, bool allow_readback, bool first_derivative_done = false) {
    State &s = h->st;
    SkipResidual skip(h);
    const int fixed = h->prm.fixed_iterations;
    const int max_itr = fixed > 0 ? fixed : 1000;
    // DFSPH.py:140 compute_density_derivative before the loop (dfsph_step_end has it fused into the density + alpha walk)
    if (!first_derivative_done) { ProfScope p(h, SPH_K_DFSPH_RHO_ADV); h->L->dfsph_rho_adv(s, 0); }
    int itr = 0;
    float avg = 0.0f;
    const float n_all = (float)dfsph_particle_num(h);   // DFSPH.py:212 divides by particle_num (every rank's, under sharding)
    int comm_rc = SPH_OK;
    // one solver iteration; under slab sharding the ghosts' kappa_v goes out before the correction and their velocities
    // after it, and the residual is summed over the ranks
    auto iteration = [&]() {
        std::swap(s.kappa_v, s.kappa_v_next);  // DFSPH.py:133 compute_kappa_v: value of the last density-derivative pass
        if (s.slab_active && !comm_rc) comm_rc = slab_exchange_scalar(h, s.kappa_v);
        { ProfScope p(h, SPH_K_DFSPH_CORRECT); h->L->dfsph_correct(s, 0); }
        if (s.slab_active && !comm_rc) comm_rc = slab_exchange_vel(h);
        { ProfScope p(h, SPH_K_DFSPH_RHO_ADV); h->L->dfsph_rho_adv(s, 0); }
        if (s.slab_active && !comm_rc) comm_rc = slab_finish_reduction(h, 0);
    };
    if (fixed <= 0 && allow_readback) {
        const double eta = 0.001 * h->prm.density_0 / (double)s.c.dt;  // :150
        int launched = 0; float sum = 0.0f;
        int rc = device_loop(h, max_itr, 0, 1, n_all, eta, iteration, &itr, &launched, &sum);
        if (rc) return rc;
        if (comm_rc) return comm_rc;
        if ((launched - itr) & 1) std::swap(s.kappa_v, s.kappa_v_next);   // iterations past the stop did not run
        h->last.iter_divergence = itr; h->last.err_divergence = sum / n_all;
        return SPH_OK;
    }
    while (itr < 1 || itr < max_itr) {
        iteration();
        if (comm_rc) return comm_rc;
        itr++;
        if (fixed > 0) continue;
        if (!allow_readback) return fail(h, SPH_ERR_UNSUPPORTED, "dfsph needs host read-back unless fixed_iterations > 0");
        float sum; int rc = read_red(h, 0, &sum); if (rc) return rc;
        avg = sum / n_all;                                          // DFSPH.py:212 (divides by particle_num)
        const double eta = 0.001 * h->prm.density_0 / (double)s.c.dt;  // :150
        if ((double)avg <= eta) break;
    }
    h->last.iter_divergence = itr; h->last.err_divergence = avg;
    return SPH_OK;
}

// DFSPH.py:225 correct_density_error
static int dfsph_density(SphHandle *h, bool allow_readback) {
    State &s = h->st;
    SkipResidual skip(h);
    const int fixed = h->prm.fixed_iterations;
    const int max_itr = fixed > 0 ? fixed : 1000;
    { ProfScope p(h, SPH_K_DFSPH_RHO_ADV); h->L->dfsph_rho_adv(s, 1); }
    int itr = 0;
    float avg = 0.0f;
    const float n_all = (float)dfsph_particle_num(h);
    int comm_rc = SPH_OK;
    auto iteration = [&]() {
        std::swap(s.kappa, s.kappa_next);      // DFSPH.py:218 compute_kappa
        if (s.slab_active && !comm_rc) comm_rc = slab_exchange_scalar(h, s.kappa);
        { ProfScope p(h, SPH_K_DFSPH_CORRECT); h->L->dfsph_correct(s, 1); }
        if (s.slab_active && !comm_rc) comm_rc = slab_exchange_vel(h);
        { ProfScope p(h, SPH_K_DFSPH_RHO_ADV); h->L->dfsph_rho_adv(s, 1); }
        if (s.slab_active && !comm_rc) comm_rc = slab_finish_reduction(h, 1);
    };
    if (fixed <= 0 && allow_readback) {
        int launched = 0; float sum = 0.0f;
        int rc = device_loop(h, max_itr, 1, 1, n_all, 0.0001, iteration, &itr, &launched, &sum);   // :239
        if (rc) return rc;
        if (comm_rc) return comm_rc;
        if ((launched - itr) & 1) std::swap(s.kappa, s.kappa_next);
        h->last.iter_density = itr; h->last.err_density = sum / n_all;
        return SPH_OK;
    }
    while (itr < 1 || itr < max_itr) {
        iteration();
        if (comm_rc) return comm_rc;
        itr++;
        if (fixed > 0) continue;
        if (!allow_readback) return fail(h, SPH_ERR_UNSUPPORTED, "dfsph needs host read-back unless fixed_iterations > 0");
        float sum; int rc = read_red(h, 1, &sum); if (rc) return rc;
        avg = sum / n_all;                                          // DFSPH.py:293
        if ((double)avg <= 0.0001) break;                            // :239
    }
    h->last.iter_density = itr; h->last.err_density = avg;
    return SPH_OK;
}

// DFSPH.py:298 _step, first half: up to (not including) rigid_solver.step() / insert_object() at :305-:308
static int dfsph_step_begin(SphHandle *h, bool allow_readback) {
    State &s = h->st;
    if (h->sort_dirty) {
        // particles were appended outside a step (plain C-ABI use): the passes below walk the cell lists, so bring them
        // (and density / alpha, which the reference refreshes after every sort, DFSPH.py:316-318) up to date first.
        // Sharded: the re-sort invalidates the halo slot tables, so it has to be the full slab search (classify, exchange,
        // sort, tables) -- a COLLECTIVE: every rank of a sharded DFSPH scene that appends between steps gets here together
        // (the Python containers append mid-step and re-sort in step_end; this is the plain C-ABI path).
        if (s.slab_active) { int rc = slab_neighbor_search(h); if (rc) return rc; }
        else ph_neighbor_search(h);
        ph_rigid_volume(h);
        { ProfScope p(h, SPH_K_DFSPH_DENSITY_ALPHA); h->L->dfsph_density_alpha(s); }
        if (s.slab_active) { int rc = slab_exchange_scalar(h, s.rho.cur()); if (rc) return rc; }   // ghost densities, as in dfsph_step_end
    }
    int rc = run_non_pressure(h); if (rc) return rc;                          // DFSPH.py:299-300
    if (s.slab_active) { rc = slab_exchange_vel(h); if (rc) return rc; }      // the density solver reads v_j of the ghosts
    rc = dfsph_density(h, allow_readback); if (rc) return rc;                 // :301
    // the sort of this step's second half follows at once when the whole step is one call (step_once): the position update hashes for it
    static const bool no_nexthash = getenv("SPH_NO_NEXT_HASH") != nullptr;
    s.nexthash.on = (!no_nexthash && h->whole_step && !s.slab_active && s.c.all_fluid && !s.has_emitter && !h->any_rigid_object &&
                     !h->sort_dirty && !h->pose_dirty) ? 1 : 0;
    { ProfScope p(h, SPH_K_MISC); h->L->advect_boundary(s); }                 // :303, :311-314 (boundary fused: it only looks at the particle itself)
    return SPH_OK;
}

// second half: :309 renew_rigid_particle_state and :311 boundary for what the host just inserted (step_insert_tail),
// then :316-:319
static int dfsph_step_end(SphHandle *h, bool allow_readback) {
    State &s = h->st;
    if (s.slab_active) { int rc = slab_neighbor_search(h); if (rc) return rc; }   // + migration / ghost exchange
    else { s.sort_skip_rho = s.c.all_fluid; ph_neighbor_search(h); }          // :316 (the density pass below rewrites every rho: the sort need not move it)
    ph_rigid_volume(h);
    static const bool unfused = getenv("SPH_NO_DFSPH_FUSED_DIV") != nullptr;   // A/B switch
    if (unfused) { ProfScope p(h, SPH_K_DFSPH_DENSITY_ALPHA); h->L->dfsph_density_alpha(s); } // :317-318
    else { ProfScope p(h, SPH_K_DFSPH_DENSITY_ALPHA); h->L->dfsph_density_alpha_div(s); }     // :317-318 + the D rho / Dt of :140
    if (s.slab_active) { int rc = slab_exchange_scalar(h, s.rho.cur()); if (rc) return rc; }   // ghost densities (kappa_j / rho_j, viscosity)
    return dfsph_divergence(h, allow_readback, !unfused);                     // :319
}

// PCISPH.py:110 refine.  Under slab sharding the ghosts' p / rho^2 goes out between the two passes of an
// iteration and their predicted positions after it; the density error is summed over the ranks.
static int pcisph_refine(SphHandle *h, bool allow_readback) {
    State &s = h->st;
    SkipResidual skip(h);
    const int fixed = h->prm.fixed_iterations;
    const int max_itr = fixed > 0 ? fixed : 1000;
    int itr = 0;
    float err = 100.0f;
    const float n_fl = (float)(s.slab_active ? h->comm_nfluid_global : (long long)h->n_fluid);   // PCISPH.py:43-46 divides by fluid_particle_num
    int comm_rc = SPH_OK;
    auto iteration = [&]() {
        { ProfScope p(h, SPH_K_PCISPH_RHO_STAR); h->L->pcisph_rho_star(s); }
        if (s.slab_active && !comm_rc) comm_rc = slab_exchange_scalar(h, s.ptm);
        { ProfScope p(h, SPH_K_PCISPH_PRESSURE_ACCEL); h->L->pcisph_pressure_accel(s); }
        if (s.slab_active && !comm_rc) comm_rc = slab_exchange_vel(h, s.ppos);
        if (s.slab_active && !comm_rc) comm_rc = slab_finish_reduction(h, 2);
    };
    if (fixed <= 0 && allow_readback) {
        int launched = 0; float sum = 0.0f;
        int rc = device_loop(h, max_itr, 2, 2, n_fl, 0.001, iteration, &itr, &launched, &sum);   // PCISPH.py:43-46, :122
        if (rc) return rc;
        if (comm_rc) return comm_rc;
        h->last.iter_pcisph = itr; h->last.err_pcisph = n_fl > 0 ? sum / n_fl : 0.0f;
        return SPH_OK;
    }
    while (itr < max_itr) {
        iteration();
        if (comm_rc) return comm_rc;
        itr++;
        if (fixed > 0) continue;
        if (!allow_readback) return fail(h, SPH_ERR_UNSUPPORTED, "pcisph needs host read-back unless fixed_iterations > 0");
        float sum; int rc = read_red(h, 2, &sum); if (rc) return rc;
        err = n_fl > 0 ? sum / n_fl : 0.0f;                          // PCISPH.py:43-46
        if (err < 0.001f) break;                                    // :122
    }
    h->last.iter_pcisph = itr; h->last.err_pcisph = err;
    return SPH_OK;
}

static int pcisph_step(SphHandle *h, bool allow_readback) {
    State &s = h->st;
    if (s.slab_active) { int rc = slab_neighbor_search(h); if (rc) return rc; }   // + migration / ghost exchange
    else { s.sort_skip_rho = s.c.all_fluid; ph_neighbor_search(h); }          // PCISPH.py:166 (:167 below rewrites every rho)
    ph_rigid_volume(h);
    { ProfScope p(h, SPH_K_DENSITY); h->L->density(s, 0); }                   // :167
    if (s.slab_active) { int rc = slab_exchange_scalar(h, s.rho.cur()); if (rc) return rc; }   // ghost densities (viscosity)
    int rc = run_non_pressure(h); if (rc) return rc;                          // :168 (+ :174, v* kept aside)
    { ProfScope p(h, SPH_K_MISC); h->L->pcisph_init(s); }                     // :169
    if (s.slab_active) { rc = slab_exchange_vel(h, s.ppos); if (rc) return rc; }   // predicted positions of the ghosts
    rc = pcisph_refine(h, allow_readback); if (rc) return rc;                 // :170
    { ProfScope p(h, SPH_K_PRESSURE_INTEGRATE); h->L->pressure_integrate(s); } // :175-177, :185
    return SPH_OK;
}

// host replica of PCISPH.py:129 compute_pcisph_k (same arithmetic as oracle/sph_ref.c)
static float host_pcisph_k(const SphParams &p) {
    const double hd = p.support_radius;
    const float hf = (float)hd;
    float kg = (float)(8.0 / M_PI);
    kg = 6.0f * kg / (float)(hd * hd * hd);
    const float diam = (float)(2.0 * p.particle_radius * 0.97);
    float sx = 0.f, sy = 0.f, sz = 0.f, s2 = 0.f;
    const int max_i = (int)(hf / diam) + 1;
    for (int i = -max_i; i <= max_i; i++)
        for (int j = -max_i; j <= max_i; j++)
            for (int k = -max_i; k <= max_i; k++) {
                const float rx = 0.0f - (float)i * diam, ry = 0.0f - (float)j * diam, rz = 0.0f - (float)k * diam;
                const float rn = sqrtf(rx * rx + ry * ry + rz * rz);
                if (rn < hf) {
                    float gx = 0.f, gy = 0.f, gz = 0.f;
                    const float q = rn / hf;
                    if (rn > 1e-5f && q <= 1.0f) {
                        const float den = rn * hf;
                        float sc;
                        if (q <= 0.5f) sc = kg * q * (3.0f * q - 2.0f);
                        else { const float f = 1.0f - q; sc = kg * (-f * f); }
                        gx = sc * (rx / den); gy = sc * (ry / den); gz = sc * (rz / den);
                    }
                    sx += gx; sy += gy; sz += gz;
                    s2 += gx * gx + gy * gy + gz * gz;
                }
            }
    const float dtV0 = (float)p.dt * (float)p.V0;
    return -0.5f / dtV0 / dtV0 / ((sx * sx + sy * sy + sz * sz) + s2);
}

static int method_prepare(SphHandle *h) {
    State &s = h->st;
    if (h->prm.method == SPH_METHOD_DFSPH) {  // DFSPH.py:321-324
        { ProfScope p(h, SPH_K_DFSPH_DENSITY_ALPHA); h->L->dfsph_density_alpha(s); }
        if (s.slab_active) { int rc = slab_exchange_scalar(h, s.rho.cur()); if (rc) return rc; }
    } else if (h->prm.method == SPH_METHOD_PCISPH) {  // PCISPH.py:188-190
        s.c.pcisph_k = host_pcisph_k(h->prm);
    }
    return SPH_OK;
}

static int method_run_phase(SphHandle *h, int phase) {
    State &s = h->st;
    if (h->prm.method == SPH_METHOD_DFSPH) {
        switch (phase) {
            case SPH_PH_DFSPH_ALPHA: { ProfScope p(h, SPH_K_DFSPH_DENSITY_ALPHA); h->L->dfsph_density_alpha(s); } return SPH_OK;
            case SPH_PH_DFSPH_DIVERGENCE: return dfsph_divergence(h, true);
            case SPH_PH_DFSPH_DENSITY: return dfsph_density(h, true);
            default: break;
        }
    }
    return fail(h, SPH_ERR_INVALID, "unknown phase %d for method %d", phase, h->prm.method);
}

#include "sph_cg_steps.hpp"
