/*
 * oracle/sph_ref.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (serial / optional OpenMP, strict f32, no FMA contraction) of
 * the per-timestep hot path of the reference project (Taichi kernels in
 * SPH/containers/base_container.py and SPH/fluid_solvers/{base_solver,WCSPH,
 * DFSPH,PCISPH}.py).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path never does.
 *
 * Parity status: the reference owns no tests / golden vectors and Taichi is not
 * installable here, so this oracle is pinned against the reference's own
 * *source* executed under a build-authored serial f32 interpreter shim
 * (oracle/taichi_shim, fixtures in tests/golden/), NOT against a real Taichi
 * run.  See DESIGN.md "Oracle".
 */
#ifndef SPH_REF_H
#define SPH_REF_H
#ifdef __cplusplus
extern "C" {
#endif

#define SPHREF_MAX_OBJECTS 20 /* base_container.py:52 */
#define SPHREF_MAT_FLUID 1    /* base_container.py:30 */
#define SPHREF_MAT_RIGID 2    /* base_container.py:29 */

typedef struct {
    /* python-side floats of the reference are doubles; they become f32 only
       where a Taichi kernel consumes them. */
    double domain_size[3];   /* base_container.py:23 */
    double particle_radius;  /* dx, :33 */
    double support_radius;   /* dh, :37/:42 */
    double V0;               /* :49 */
    double padding;          /* :58 */
    int    grid_num[3];      /* :56 */
    double gravity[3];       /* base_solver.py:16 */
    double g_upper;          /* :21-23 */
    double viscosity;        /* :26 */
    double viscosity_b;      /* :27-29 */
    double density_0;        /* :31 */
    double surface_tension;  /* :32 (0.01) */
    double dt;               /* :36 (stored into an f32 field) */
    int    particle_max_num; /* base_container.py:116 */
    int    viscosity_implicit; /* base_solver.py:40 */
    int    method;           /* 0 wcsph, 1 dfsph, 2 pcisph */
    int    fixed_iterations; /* >0: run exactly this many solver iterations (bench) */
} SphRefParams;

typedef struct SphRef SphRef;

SphRef *sphref_create(const SphRefParams *p);
void    sphref_destroy(SphRef *s);

/* base_container.py:441 _add_particles */
int sphref_add_particles(SphRef *s, int object_id, int n, const float *pos,
                         const float *vel, const float *density,
                         const float *pressure, const int *material,
                         const int *is_dynamic, const int *color);
void sphref_set_object(SphRef *s, int object_id, int material, int is_dynamic);
void sphref_set_rigid_pose(SphRef *s, int object_id, const float *com,
                           const float *rot9, const float *vel, const float *angvel,
                           const float *com0);

/* raw field access for tests: returns pointer to the named array */
void *sphref_field(SphRef *s, const char *name);
int   sphref_particle_num(SphRef *s);
int   sphref_fluid_particle_num(SphRef *s);
double sphref_scalar(SphRef *s, const char *name);
long long sphref_last_pairs(SphRef *s); /* accepted (i,j) pairs of the last step */

/* one entry point per reference kernel (names follow the reference) */
void sphref_init_grid(SphRef *s);                 /* base_container.py:496 */
void sphref_prefix_sum(SphRef *s);                /* :546 */
void sphref_reorder_particles(SphRef *s);         /* :506 */
void sphref_prepare_neighborhood_search(SphRef *s); /* :544 */
void sphref_compute_rigid_particle_volume(SphRef *s); /* base_solver.py:106 */
void sphref_compute_density(SphRef *s);           /* :522 */
void sphref_compute_gravity_acceleration(SphRef *s); /* :203 */
void sphref_compute_surface_tension_acceleration(SphRef *s); /* :210 */
void sphref_compute_viscosity_acceleration_standard(SphRef *s); /* :232 */
void sphref_implicit_viscosity_solve(SphRef *s);  /* :509 */
void sphref_compute_non_pressure_acceleration(SphRef *s); /* :190 */
void sphref_update_fluid_velocity(SphRef *s);     /* :643 */
void sphref_update_fluid_position(SphRef *s);     /* :652 */
void sphref_compute_pressure_acceleration(SphRef *s); /* :136 */
void sphref_enforce_domain_boundary_3D(SphRef *s); /* :575 */
void sphref_prepare_emitter(SphRef *s);           /* :670 */
void sphref_renew_rigid_particle_state(SphRef *s);/* :616 */
void sphref_wcsph_compute_pressure(SphRef *s);    /* WCSPH.py:17 */
void sphref_dfsph_compute_alpha(SphRef *s);       /* DFSPH.py:23 */
void sphref_dfsph_compute_density_derivative(SphRef *s); /* :66 */
void sphref_dfsph_compute_density_star(SphRef *s);/* :105 */
int  sphref_dfsph_correct_divergence_error(SphRef *s); /* :139 */
int  sphref_dfsph_correct_density_error(SphRef *s);    /* :225 */
void sphref_pcisph_compute_k(SphRef *s);          /* PCISPH.py:129 */
int  sphref_pcisph_refine(SphRef *s);             /* :110 */

void sphref_prepare(SphRef *s);                   /* base_solver.py:683 (+ DFSPH.py:321, PCISPH.py:188) */
void sphref_step(SphRef *s);                      /* base_solver.py:692 */
void sphref_step_begin(SphRef *s);                /* _step() up to rigid_solver.step() / insert_object() */
void sphref_step_end(SphRef *s);                  /* rest of _step() + step() tail */

#ifdef __cplusplus
}
#endif
#endif
