/*
 * include/sph_hip.h -- C-ABI of libsph_hip.so, the MI355X (gfx950) SPH hot path.
 *
 * The reference (jason-huang03/SPH_Project) is 100 % Python + Taichi and has no FFI of its
 * own; this header is the boundary a reference-side binding would target (ctypes stub in
 * INTEGRATION.md).  Each entry point names the reference interface it replaces (paths
 * relative to the reference root).  Conventions:
 *   - plain pointers and sizes only; the caller owns every host buffer (C-contiguous
 *     f32 / i32), the library owns all device memory behind the opaque handle;
 *   - every call returns 0 or a negative SphStatus; text via sph_last_error();
 *     HIP / RCCL failures are captured, never fatal, nothing throws across the ABI;
 *   - one host thread per handle; one HIP compute stream (+ one comm stream) per handle;
 *   - python-side quantities of the reference are doubles here and are rounded to f32
 *     exactly where a reference Taichi kernel would consume them.
 */
#ifndef SPH_HIP_H
#define SPH_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SPH_MAX_OBJECTS 20 /* SPH/containers/base_container.py:52 max_num_object */
#define SPH_MAT_FLUID 1    /* base_container.py:30 */
#define SPH_MAT_RIGID 2    /* base_container.py:29 */

typedef enum {
    SPH_OK = 0,
    SPH_ERR_INVALID = -1,   /* bad argument / bad state */
    SPH_ERR_CAPACITY = -2,  /* particle_max_num exceeded */
    SPH_ERR_HIP = -3,       /* HIP runtime error (text in sph_last_error) */
    SPH_ERR_NO_DEVICE = -4, /* no gfx950 device visible */
    SPH_ERR_COMM = -5,      /* RCCL error */
    SPH_ERR_UNSUPPORTED = -6
} SphStatus;

typedef enum { SPH_METHOD_WCSPH = 0, SPH_METHOD_DFSPH = 1, SPH_METHOD_PCISPH = 2 } SphMethod;

/* Scene / solver constants.  Mirrors what BaseContainer.__init__ (base_container.py:10-60)
   and BaseSolver.__init__ (SPH/fluid_solvers/base_solver.py:9-54) derive from the JSON. */
typedef struct {
    double domain_size[3];   /* base_container.py:23 (domainStart must be 0, run_simulation.py:11) */
    double particle_radius;  /* :33 dx */
    double support_radius;   /* :37 dh = 4 dx (or "supportRadius") = grid cell size :55 */
    double V0;               /* :49 0.8 * diameter^3 */
    double padding;          /* :58 */
    int32_t grid_num[3];     /* :56 ceil(domain_size / dh) */
    double gravity[3];       /* base_solver.py:16 */
    double g_upper;          /* :21-23 gravitationUpper (10000 if absent) */
    double viscosity;        /* :26 */
    double viscosity_b;      /* :27-29 */
    double density_0;        /* :31 */
    double surface_tension;  /* :32 (0.01) */
    double dt;               /* :36 timeStepSize */
    int32_t particle_max_num;/* base_container.py:116 */
    int32_t viscosity_implicit; /* base_solver.py:40 viscosityMethod == "implicit" */
    int32_t method;          /* SphMethod; run_simulation.py:46-63 */
    int32_t fixed_iterations;/* 0: reference stopping rules; >0: exactly this many DFSPH/PCISPH/CG iterations */
    int32_t fast_math;       /* 0: IEEE div/sqrt, no FMA contraction; 1: v_rcp/v_rsq + FMA */
    int32_t device;          /* HIP device ordinal, -1: current */
    int32_t force_global;    /* debug mode of the neighbour passes: 0 normal; 1 every candidate run through the tile in chunks; 4 every group down the ordered walk; ... */
    int32_t deterministic;   /* 1: stable within-cell order (bit-reproducible sums) */
} SphParams;

typedef struct SphHandle SphHandle;

/* Per-particle fields addressable by sph_download / sph_upload.  Layouts are the reference's
   (base_container.py:138-185; dfsph_container.py:13-17; pcisph_container.py:15-19), in the
   current sorted order of the container (as BaseContainer.dump, base_container.py:599). */
typedef enum {
    SPH_F_POSITION = 0,      /* f32[n][3] particle_positions */
    SPH_F_VELOCITY = 1,      /* f32[n][3] particle_velocities */
    SPH_F_ACCELERATION = 2,  /* f32[n][3] particle_accelerations */
    SPH_F_DENSITY = 3,       /* f32[n]    particle_densities */
    SPH_F_PRESSURE = 4,      /* f32[n]    particle_pressures */
    SPH_F_REST_VOLUME = 5,   /* f32[n]    particle_rest_volumes */
    SPH_F_MASS = 6,          /* f32[n]    particle_masses */
    SPH_F_MATERIAL = 7,      /* i32[n]    particle_materials */
    SPH_F_OBJECT_ID = 8,     /* i32[n]    particle_object_ids */
    SPH_F_IS_DYNAMIC = 9,    /* i32[n]    particle_is_dynamic */
    SPH_F_COLOR = 10,        /* i32[n][3] particle_colors */
    SPH_F_PARTICLE_ID = 11,  /* i32[n]    insertion index (not in the reference; lets tests match particles) */
    SPH_F_GRID_ID = 12,      /* i32[n]    reference flat cell id (ix*ny+iy)*nz+iz, base_container.py:473 */
    SPH_F_DFSPH_ALPHA = 13,  /* f32[n] */
    SPH_F_DFSPH_KAPPA = 14,  /* f32[n] */
    SPH_F_DFSPH_KAPPA_V = 15,/* f32[n] */
    SPH_F_DENSITY_STAR = 16, /* f32[n] */
    SPH_F_DENSITY_DERIV = 17,/* f32[n] */
    SPH_F_PRESSURE_ACCEL = 18,   /* f32[n][3] pcisph particle_pressure_accelerations */
    SPH_F_PREDICTED_VEL = 19,    /* f32[n][3] */
    SPH_F_PREDICTED_POS = 20,    /* f32[n][3] */
    SPH_F_CG_X = 21,             /* f32[n][3] base_solver.py:46 */
    SPH_F_ORIG_POSITION = 22,    /* f32[n][3] rigid_particle_original_positions */
    SPH_F_GHOST = 23,            /* i32[n]    1 for ghost copies of a neighbour slab's particles (multi-GPU), else 0 */
    /* solver scratch the per-term checks of tests/test_hip_solvers.py need (not fields of the reference: DFSPH.py:133 / :218 recompute
       kappa from density_star / density_derivative + alpha right before every correction; this library computes it in the pass that
       produces them and hands it to the next correction): */
    SPH_F_DFSPH_KAPPA_NEXT = 24,   /* f32[n] (rho*_i - 1) alpha_i / dt of the LAST density_star pass (pairs with SPH_F_DENSITY_STAR) */
    SPH_F_DFSPH_KAPPA_V_NEXT = 25, /* f32[n] D rho_i / Dt * alpha_i of the LAST density_derivative pass (pairs with SPH_F_DENSITY_DERIV) */
    SPH_F_DEBUG_CAPTURE = 26,      /* f32[n] libsph_hip_testhooks.so only: PCISPH pressure BEFORE the last executed update_pressure
                                      (PCISPH.py:66-73); the production library leaves the density pass's scratch there */
    SPH_F_COUNT_
} SphField;

/* Individually callable phases (tests, profiling).  sph_step() runs them in the order of the
   reference's _step() (WCSPH.py:27, DFSPH.py:298, PCISPH.py:165) followed by step() :692. */
typedef enum {
    SPH_PH_NEIGHBOR_SEARCH = 0,  /* base_container.py:544 prepare_neighborhood_search */
    SPH_PH_RIGID_VOLUME = 1,     /* base_solver.py:106 */
    SPH_PH_DENSITY = 2,          /* :522 (+ WCSPH.py:17 EOS when method == wcsph) */
    SPH_PH_NON_PRESSURE = 3,     /* :190 + :643 (gravity, surface tension, viscosity, v += dt a) */
    SPH_PH_PRESSURE_INTEGRATE = 4, /* :136 + :643 + :652 + :575 */
    SPH_PH_DFSPH_ALPHA = 5,      /* DFSPH.py:23 */
    SPH_PH_DFSPH_DIVERGENCE = 6, /* DFSPH.py:139 */
    SPH_PH_DFSPH_DENSITY = 7,    /* DFSPH.py:225 */
    SPH_PH_COUNT_
} SphPhase;

typedef struct {
    int64_t steps;               /* steps executed since create */
    int64_t pair_interactions;   /* accepted (i fluid, j != i, |x_ij| < dh) pairs summed over the
                                    neighbour passes of the LAST step (SURVEY 8d metric) */
    int32_t particle_num;
    int32_t fluid_particle_num;
    int32_t iter_divergence, iter_density, iter_pcisph, iter_cg; /* last step */
    float   err_divergence, err_density, err_pcisph, err_cg;
    int64_t lds_fallback_blocks; /* neighbour-pass workgroups that overflowed the LDS tile (last step) */
    double  total_time;          /* container.total_time, base_solver.py:694 */
    int64_t pair_evaluations;    /* the same accepted pairs counted once per neighbour walk the library actually makes
                                    (a fused kernel that does the work of k reference passes adds k to
                                    pair_interactions and 1 here), last step */
    int64_t hash_launches;       /* launches of the cell-id / histogram kernel (init_grid, base_container.py:496) since create */
    int64_t prehashed_sorts;     /* sorts since create whose init_grid was done by the force pass of the step before (inside one
                                    sph_step_async(n) call of an unsharded all-fluid WCSPH scene): hash_launches + prehashed_sorts
                                    = sorts.  Lets a test assert which path a timed region really took. */
    int64_t list_sorts;          /* deterministic sorts since create that ranked the particles from per-cell run lists filed by whoever
                                    hashed them (reorder_particles, base_container.py:506-515, in two launches: rank + gather with the
                                    per-tile preparation of the neighbour passes fused in) instead of run records filed after the scan */
} SphStats;

/* Kernel ids for the HIP-event profiler (sph_profile_*). */
typedef enum {
    SPH_K_HASH_COUNT = 0, SPH_K_SCAN = 1, SPH_K_SCATTER = 2, SPH_K_DENSITY = 3,
    SPH_K_NON_PRESSURE = 4, SPH_K_PRESSURE_INTEGRATE = 5, SPH_K_RIGID_VOLUME = 6,
    SPH_K_DFSPH_DENSITY_ALPHA = 7, SPH_K_DFSPH_RHO_ADV = 8, SPH_K_DFSPH_CORRECT = 9,
    SPH_K_REDUCE = 10, SPH_K_PCISPH_RHO_STAR = 11, SPH_K_PCISPH_PRESSURE_ACCEL = 12,
    SPH_K_CG_PREPARE = 13, SPH_K_CG_AP = 14, SPH_K_CG_VECTOR = 15, SPH_K_MISC = 16,
    SPH_K_HALO = 17, SPH_K_WCSPH_FORCES = 18, SPH_K_COUNT_
} SphKernelId;

/* --- lifetime -------------------------------------------------------------------------- */
/* replaces XContainer.__init__ allocation (base_container.py:129-185) + XSolver.__init__.
   particle_max_num <= 268,435,455 per handle (SPH_ERR_CAPACITY beyond: shard the scene over GPUs) */
int sph_create(const SphParams *params, SphHandle **out);
void sph_destroy(SphHandle *h);
/* message of the last failure on this handle (h == NULL: last failure of sph_create) */
const char *sph_last_error(SphHandle *h);

/* --- scene upload ---------------------------------------------------------------------- */
/* replaces BaseContainer._add_particles (base_container.py:441): append n particles of one object.
   pos/vel: f32[n][3]; density/pressure: f32[n]; material/is_dynamic: i32[n]; color: i32[n][3]. */
int sph_append_particles(SphHandle *h, int object_id, int n, const float *pos, const float *vel,
                         const float *density, const float *pressure, const int32_t *material,
                         const int32_t *is_dynamic, const int32_t *color);
/* persistent ids (SPH_F_PARTICLE_ID) of the n particles appended last; default = insertion index on this handle.  A rank
   of a sharded scene passes global insertion indices. */
int sph_set_appended_ids(SphHandle *h, int n, const int32_t *ids);
/* object_materials / rigid_body_is_dynamic (base_container.py:150,156; insert_object :237,:317,:332).  On a sharded scene
   (sph_comm_set_slab) EVERY rank registers every object, whether or not it holds any of its particles: the halo records of a
   scene with a dynamic rigid body carry the rest positions (64 instead of 48 bytes); sph_prepare agrees on that over all ranks,
   and a later mismatch fails the exchange on both sides (record size in the message header). */
int sph_set_object(SphHandle *h, int object_id, int material, int is_dynamic);
/* pose written by the host rigid solver (SPH/rigid_solver/bullet_solver.py:158-167); rot9 row-major.  Like the reference's
   rigid_body_* fields it is only READ at the renew_rigid_particle_state point of a step (inside sph_step_end / the second half of
   sph_step), wherever before that it was written: a pose pushed between two steps moves the particles after the fluid passes of
   the next step, not before them (base_solver.py:616, WCSPH.py:43). */
int sph_set_rigid_pose(SphHandle *h, int object_id, const float *com, const float *rot9,
                       const float *vel, const float *angvel, const float *com0);
/* rigid_body_forces / rigid_body_torques read by bullet_solver.py:150-156; reset != 0 zeroes them.  On a sharded scene
   (sph_comm_set_slab) this is a collective: every rank calls it at the same point of the step and gets the sum over the ranks. */
int sph_get_rigid_wrench(SphHandle *h, float *force, float *torque, int reset);

/* --- time stepping --------------------------------------------------------------------- */
/* replaces XSolver.prepare() (base_solver.py:683, DFSPH.py:321, PCISPH.py:188); particles of
   entryTime <= 0 must have been appended */
int sph_prepare(SphHandle *h);
/* replaces XSolver.step() (base_solver.py:692) called nsteps times; synchronous on return */
int sph_step(SphHandle *h, int nsteps);
/* enqueue nsteps without the trailing host synchronisation (bench / overlap); only valid when no
   per-iteration host read-back is needed (wcsph, or fixed_iterations > 0) */
int sph_step_async(SphHandle *h, int nsteps);
int sph_synchronize(SphHandle *h);
/* One step in two halves, for hosts that act in the middle of _step() exactly where the reference does:
   sph_step_begin runs _step() up to (not including) `self.rigid_solver.step()` (WCSPH.py:39, DFSPH.py:305,
   PCISPH.py:179); the host then integrates rigid bodies (sph_get_rigid_wrench / sph_set_rigid_pose =
   bullet_solver.py:144-167) and appends objects whose entryTime has come (sph_append_particles =
   base_container.py:212-341); sph_step_end runs the rest: renew_rigid_particle_state + enforce_domain_boundary
   (+ for DFSPH the neighbour search, density, alpha and divergence solve, DFSPH.py:311-319) and step()'s tail
   (base_solver.py:694-696, including compute_rigid_particle_volume on the grid of the last sort).
   sph_step(h, 1) == sph_step_begin(h); sph_step_end(h). */
int sph_step_begin(SphHandle *h);
int sph_step_end(SphHandle *h);
/* one reference kernel group at a time (tests) */
int sph_run_phase(SphHandle *h, int phase);

/* --- state access ---------------------------------------------------------------------- */
/* replaces field.to_numpy() / BaseContainer.dump (base_container.py:599).  bytes must equal
   particle_num * element size of the field. */
int sph_download(SphHandle *h, int field, void *dst, size_t bytes);
int sph_upload(SphHandle *h, int field, const void *src, size_t bytes);
int sph_particle_num(SphHandle *h);       /* container.particle_num[None] */
int sph_fluid_particle_num(SphHandle *h); /* container.fluid_particle_num[None] */
int sph_get_stats(SphHandle *h, SphStats *out);

/* --- profiling (HIP events on the compute stream) -------------------------------------- */
/* record a HIP event pair around every launch of `kernel_id` (-1: all kernels); accumulates. */
int sph_profile_enable(SphHandle *h, int kernel_id, int on);
int sph_profile_reset(SphHandle *h);
/* resolves pending events; returns launches and total milliseconds of that kernel id */
int sph_profile_read(SphHandle *h, int kernel_id, int64_t *launches, double *total_ms);
const char *sph_kernel_name(int kernel_id);
/* device properties the bench reports: name (<=255 chars), CU count, HBM bytes */
int sph_device_info(SphHandle *h, char *name256, int *cu_count, int64_t *hbm_bytes);

/* device-to-device copy rate of this GPU, measured now (the second denominator of the bench's HBM roofline next to the
   8 TB/s spec): `reps` hipMemcpyAsync D2D copies of `bytes` bytes between two scratch buffers on the handle's
   stream, bracketed by HIP events; *gb_per_s = (read + written bytes) / time = 2 * bytes * reps / t.  The scratch is freed
   before returning. */
int sph_measure_copy_rate(SphHandle *h, size_t bytes, int reps, double *gb_per_s);

/* --- multi-GPU: z-slab sharding, one process per GPU, RCCL over xGMI ------------------- */
/* number of HIP devices this process sees (launchers map local rank -> device without any other runtime) */
int sph_device_count(void);
/* 128-byte RCCL unique id, created on rank 0 and distributed by the host launcher */
int sph_comm_unique_id(void *out128);
/* attach this handle to a communicator of `nranks` processes (RCCL ncclCommInitRank; SPH_COMM_TRANSPORT=shm:
   POSIX shared memory, several ranks may then share one GPU -- test rig) */
int sph_comm_init(SphHandle *h, int rank, int nranks, const void *id128);
/* what carries the halo messages of this handle: "rccl", "shm", "ipc-push+rccl", "ipc-push+shm" ("none" before sph_comm_init).
   Default (SPH_COMM_TRANSPORT unset / "auto"): RCCL for barriers and all-reduces; halo payload by device stores straight into the
   neighbour's inbox, mapped through hipIpc ("push"), if that can be set up and passes its self-test on EVERY rank, else RCCL
   ncclSend / ncclRecv.  "ipc": push or fail; "rccl": RCCL only; "shm" / "shm+ipc": shared-memory control plane (several ranks on
   one GPU, test rig) with host-staged mailboxes / with the push transport. */
const char *sph_comm_transport(SphHandle *h);
/* turn the handle into one z-slab: it owns the cell layers [z_lo, z_hi) of the SCENE's z (global grid) plus one ghost layer on each
   interior side; before any particle is appended.  Everything a caller hands over or reads back stays in the scene's frame.  Inside,
   the slabs are cut along the fastest axis of the cell order (default) or -- SPH_SLAB_LAYOUT=slow -- the scene's z is mapped onto the
   library's slowest axis at this boundary (positions, velocities, every 3-vector field, gravity, domain, rigid poses and wrenches are
   permuted).  Inside a multi-step call (sph_step_async(n)) a sharded WCSPH step lets its force pass write the NEXT step's
   halo message: between such steps the device state is half-classified, which no host call can observe (every read settles first). */
int sph_comm_set_slab(SphHandle *h, int z_lo, int z_hi);
int sph_comm_get_slab(SphHandle *h, int *z_lo, int *z_hi, int *n_owned, int *n_ghost);
/* slab cuts follow the fluid: every `every_steps` steps (default 64, env SPH_SLAB_REBALANCE; 0 = never) the ranks
   all-reduce their per-layer particle histograms and move each interior cut by at most one cell layer towards the
   balanced plan; the layer that changes hands migrates through the ordinary per-step exchange */
int sph_comm_set_rebalance(SphHandle *h, int every_steps);
/* late entry (base_container.py:218-221) under sharding: the object holds n particles in the whole scene, n_fluid of them
   fluid; every rank calls this next to its sph_append_particles of the slab's share (which may be empty) */
int sph_comm_add_global_count(SphHandle *h, int n, int n_fluid);
/* host-visible collectives over the communicator (what a launcher otherwise needs MPI / torch.distributed for):
   in-place all-reduce of <= 128 doubles, op 0 sum / 1 max / 2 min (ncclAllReduce); barrier = drain the stream,
   then all-reduce; both synchronous.  The solver residuals of a sharded DFSPH / PCISPH step use the same path. */
int sph_comm_allreduce(SphHandle *h, double *inout, int count, int op);
int sph_comm_barrier(SphHandle *h);
/* transport self-test: ring shift of n floats (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd; a self pair when
   the communicator has one rank), every word checked, plus an all-reduce check */
int sph_comm_selftest(SphHandle *h, int n);

/* --- mesh -> particles (host code, no GPU involved) ------------------------------------ */
/* replaces trimesh's mesh.voxelized(pitch).fill().points in BaseContainer.load_rigid_body (base_container.py:641-642):
   voxel centres (integer multiples of pitch) of the surface voxels plus the region they enclose, f32 xyz, x slowest.
   out_xyz == NULL: only *n_points is set. */
int sph_voxelize_mesh(const double *vertices, int n_vertices, const int32_t *faces, int n_faces, double pitch,
                      float *out_xyz, int64_t capacity_points, int64_t *n_points);
/* replaces mesh.contains(lattice) in BaseContainer.load_fluid_body (base_container.py:686-694): crossing parity along z;
   inside[(i * ny + j) * nz + k] for the point (xs[i], ys[j], zs[k]) */
int sph_points_in_mesh(const double *vertices, int n_vertices, const int32_t *faces, int n_faces, const double *xs, int nx,
                       const double *ys, int ny, const double *zs, int nz, uint8_t *inside);

/* --- frame export (host code, no GPU involved) ------------------------------------------ */
/* replaces ti.tools.PLYWriter(num_vertices = n).add_vertex_pos(x, y, z).export_ascii(path) of run_simulation.py:139-144: the ASCII PLY of
   one fluid object and frame (header as Taichi's python/taichi/tools/ply.py prints it, every value as str(np.float32) followed by a
   blank).  xyz: f32[n][3] in the scene's frame, e.g. what sph_download(SPH_F_POSITION) returned.  SPH_ERR_UNSUPPORTED: the file could
   not be written. */
int sph_write_ply_ascii(const char *path, const float *xyz, int64_t n);
/* str(np.float32(v)) -- the number format of that file -- into out (>= 48 bytes, no terminator); returns the length */
int sph_format_f32(float v, char *out);

#ifdef __cplusplus
}
#endif
#endif /* SPH_HIP_H */
