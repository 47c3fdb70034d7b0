/*
 * oracle/sph_ref.c -- TEST INFRASTRUCTURE ONLY (see sph_ref.h).
 *
 * Strict-f32 CPU restatement of the reference hot path.  Every function cites
 * the reference file:line it follows (paths relative to the reference root).
 * Rules used throughout, to mirror Taichi's default_fp=f32 semantics:
 *   - every kernel-local value is a `float`; python-side doubles are folded in
 *     double first and rounded to f32 at the point a kernel consumes them;
 *   - expressions keep the reference's operand order (no re-association);
 *   - build with -ffp-contract=off (no FMA) and without -ffast-math;
 *   - neighbour sums run in the reference's order: 27 cells with the x offset
 *     outermost (ti.ndrange), particles ascending inside a cell, and the
 *     counting sort is the serial (stable) execution of base_container.py:510.
 * Deviations (all documented in DESIGN.md): neighbour cells outside the grid
 * are skipped instead of aliased (base_container.py:553 has no bounds check),
 * cell coordinates are clamped into the grid.
 */
#include "sph_ref.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } v3;

static inline v3 v3_make(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static inline v3 v3_scale_l(float s, v3 a) { return v3_make(s * a.x, s * a.y, s * a.z); }
static inline v3 v3_div(v3 a, float s) { return v3_make(a.x / s, a.y / s, a.z / s); }
static inline float v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float v3_norm_sqr(v3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
static inline float v3_norm(v3 a) { return sqrtf(v3_norm_sqr(a)); }
static inline v3 v3_cross(v3 a, v3 b) {
    return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

typedef struct { float m[3][3]; } m3;

struct SphRef {
    SphRefParams prm;
    int G;               /* number of grid cells */
    int particle_num;    /* base_container.py:50 */
    int fluid_particle_num; /* :125 */
    double total_time;
    /* f32 views of python-side constants */
    float dh, grid_size, dt, density_0, g_upper, diameter;
    long long last_pairs;
    int last_iter_div, last_iter_den, last_iter_pci, last_iter_cg;
    float last_err_div, last_err_den, last_err_pci, last_err_cg;

    /* base_container.py:132-185 */
    int *grid_num_particles, *grid_num_particles_temp;
    int *particle_object_ids; v3 *particle_positions, *particle_velocities, *particle_accelerations;
    float *particle_rest_volumes, *particle_masses, *particle_densities, *particle_pressures;
    int *particle_materials, *particle_colors, *particle_is_dynamic;
    v3 *rigid_particle_original_positions;
    int object_materials[SPHREF_MAX_OBJECTS];
    int rigid_body_is_dynamic[SPHREF_MAX_OBJECTS];
    v3 rigid_body_original_centers_of_mass[SPHREF_MAX_OBJECTS];
    v3 rigid_body_centers_of_mass[SPHREF_MAX_OBJECTS];
    m3 rigid_body_rotations[SPHREF_MAX_OBJECTS];
    v3 rigid_body_torques[SPHREF_MAX_OBJECTS], rigid_body_forces[SPHREF_MAX_OBJECTS];
    v3 rigid_body_velocities[SPHREF_MAX_OBJECTS], rigid_body_angular_velocities[SPHREF_MAX_OBJECTS];
    /* sort buffers */
    int *particle_object_ids_buffer; v3 *particle_positions_buffer, *rigid_particle_original_positions_buffer,
        *particle_velocities_buffer;
    float *particle_rest_volumes_buffer, *particle_masses_buffer, *particle_densities_buffer;
    int *particle_materials_buffer, *particle_colors_buffer, *is_dynamic_buffer;
    int *grid_ids, *grid_ids_buffer, *grid_ids_new;
    /* dfsph_container.py:13-17 */
    float *particle_dfsph_alphas, *particle_dfsph_kappa, *particle_dfsph_kappa_v,
          *particle_densities_star, *particle_densities_derivatives;
    /* pcisph_container.py:15-19 */
    float density_error, pcisph_k;
    v3 *particle_pressure_accelerations, *particle_predicted_velocities, *particle_predicted_positions;
    /* base_solver.py:43-52 */
    v3 *cg_p, *original_velocity, *cg_Ap, *cg_x, *cg_b, *cg_r; m3 *cg_diagnol_ii_inv;
    float cg_alpha, cg_beta, cg_error;
};

/* Zeroed arrays whose pages are first touched round-robin by the OpenMP threads (page-granular schedule(static, 1)): with bound threads
   (OMP_PROC_BIND=spread) the pages end up interleaved over the NUMA nodes instead of all on the node of the thread that called calloc /
   wrote the particles first -- on a two-socket box that single node was the bottleneck of every pass beyond ~16 threads. */
static void *alloc_interleaved(size_t bytes) {
    if (bytes == 0) bytes = 1;
    char *p = malloc(bytes);
    if (!p) return NULL;
    const long long pages = (long long)((bytes + 4095) / 4096);
#pragma omp parallel for schedule(static, 1)
    for (long long k = 0; k < pages; k++) {
        const size_t o = (size_t)k * 4096;
        memset(p + o, 0, bytes - o < 4096 ? bytes - o : 4096);
    }
    return p;
}
#define ALLOC(ptr, n) do { (ptr) = alloc_interleaved(((size_t)(n) > 0 ? (size_t)(n) : 1) * sizeof(*(ptr))); } while (0)

SphRef *sphref_create(const SphRefParams *p) {
    SphRef *s = calloc(1, sizeof(SphRef));
    s->prm = *p;
    s->G = p->grid_num[0] * p->grid_num[1] * p->grid_num[2];
    s->dh = (float)p->support_radius;
    s->grid_size = (float)p->support_radius; /* base_container.py:55 */
    s->dt = (float)p->dt;
    s->density_0 = (float)p->density_0;
    s->g_upper = (float)p->g_upper;
    s->diameter = (float)(2.0 * p->particle_radius);
    int n = p->particle_max_num;
    ALLOC(s->grid_num_particles, s->G); ALLOC(s->grid_num_particles_temp, s->G);
    ALLOC(s->particle_object_ids, n); ALLOC(s->particle_positions, n); ALLOC(s->particle_velocities, n);
    ALLOC(s->particle_accelerations, n); ALLOC(s->particle_rest_volumes, n); ALLOC(s->particle_masses, n);
    ALLOC(s->particle_densities, n); ALLOC(s->particle_pressures, n); ALLOC(s->particle_materials, n);
    ALLOC(s->particle_colors, 3 * n); ALLOC(s->particle_is_dynamic, n);
    ALLOC(s->rigid_particle_original_positions, n);
    ALLOC(s->particle_object_ids_buffer, n); ALLOC(s->particle_positions_buffer, n);
    ALLOC(s->rigid_particle_original_positions_buffer, n); ALLOC(s->particle_velocities_buffer, n);
    ALLOC(s->particle_rest_volumes_buffer, n); ALLOC(s->particle_masses_buffer, n);
    ALLOC(s->particle_densities_buffer, n); ALLOC(s->particle_materials_buffer, n);
    ALLOC(s->particle_colors_buffer, 3 * n); ALLOC(s->is_dynamic_buffer, n);
    ALLOC(s->grid_ids, n); ALLOC(s->grid_ids_buffer, n); ALLOC(s->grid_ids_new, n);
    ALLOC(s->particle_dfsph_alphas, n); ALLOC(s->particle_dfsph_kappa, n); ALLOC(s->particle_dfsph_kappa_v, n);
    ALLOC(s->particle_densities_star, n); ALLOC(s->particle_densities_derivatives, n);
    ALLOC(s->particle_pressure_accelerations, n); ALLOC(s->particle_predicted_velocities, n);
    ALLOC(s->particle_predicted_positions, n);
    ALLOC(s->cg_p, n); ALLOC(s->original_velocity, n); ALLOC(s->cg_Ap, n); ALLOC(s->cg_x, n);
    ALLOC(s->cg_b, n); ALLOC(s->cg_r, n); ALLOC(s->cg_diagnol_ii_inv, n);
    for (int i = 0; i < n; i++) s->particle_object_ids[i] = -1; /* base_solver.py:680 init_object_id */
    for (int o = 0; o < SPHREF_MAX_OBJECTS; o++)
        for (int a = 0; a < 3; a++) s->rigid_body_rotations[o].m[a][a] = 1.0f;
    return s;
}

void sphref_destroy(SphRef *s) {
    if (!s) return;
    free(s->grid_num_particles); free(s->grid_num_particles_temp); free(s->particle_object_ids);
    free(s->particle_positions); free(s->particle_velocities); free(s->particle_accelerations);
    free(s->particle_rest_volumes); free(s->particle_masses); free(s->particle_densities);
    free(s->particle_pressures); free(s->particle_materials); free(s->particle_colors);
    free(s->particle_is_dynamic); free(s->rigid_particle_original_positions);
    free(s->particle_object_ids_buffer); free(s->particle_positions_buffer);
    free(s->rigid_particle_original_positions_buffer); free(s->particle_velocities_buffer);
    free(s->particle_rest_volumes_buffer); free(s->particle_masses_buffer); free(s->particle_densities_buffer);
    free(s->particle_materials_buffer); free(s->particle_colors_buffer); free(s->is_dynamic_buffer);
    free(s->grid_ids); free(s->grid_ids_buffer); free(s->grid_ids_new);
    free(s->particle_dfsph_alphas); free(s->particle_dfsph_kappa); free(s->particle_dfsph_kappa_v);
    free(s->particle_densities_star); free(s->particle_densities_derivatives);
    free(s->particle_pressure_accelerations); free(s->particle_predicted_velocities);
    free(s->particle_predicted_positions);
    free(s->cg_p); free(s->original_velocity); free(s->cg_Ap); free(s->cg_x); free(s->cg_b); free(s->cg_r);
    free(s->cg_diagnol_ii_inv);
    free(s);
}

/* base_container.py:404 add_particle / :441 _add_particles */
int sphref_add_particles(SphRef *s, int object_id, int n, const float *pos, const float *vel,
                         const float *density, const float *pressure, const int *material,
                         const int *is_dynamic, const int *color) {
    if (s->particle_num + n > s->prm.particle_max_num) return -1;
    const float V0 = (float)s->prm.V0;
    for (int k = 0; k < n; k++) {
        int p = s->particle_num + k;
        s->particle_object_ids[p] = object_id;
        s->particle_positions[p] = v3_make(pos[3 * k], pos[3 * k + 1], pos[3 * k + 2]);
        s->rigid_particle_original_positions[p] = s->particle_positions[p];
        s->particle_velocities[p] = v3_make(vel[3 * k], vel[3 * k + 1], vel[3 * k + 2]);
        s->particle_densities[p] = density[k];
        s->particle_rest_volumes[p] = V0;
        s->particle_masses[p] = V0 * density[k];
        s->particle_pressures[p] = pressure[k];
        s->particle_materials[p] = material[k];
        s->particle_is_dynamic[p] = is_dynamic[k];
        for (int c = 0; c < 3; c++) s->particle_colors[3 * p + c] = color[3 * k + c];
        if (material[k] == SPHREF_MAT_FLUID) s->fluid_particle_num++; /* base_container.py:797 */
    }
    s->particle_num += n;
    return 0;
}

void sphref_set_object(SphRef *s, int object_id, int material, int is_dynamic) {
    if (object_id < 0 || object_id >= SPHREF_MAX_OBJECTS) return;
    s->object_materials[object_id] = material;
    s->rigid_body_is_dynamic[object_id] = is_dynamic;
}

void sphref_set_rigid_pose(SphRef *s, int o, const float *com, const float *rot9, const float *vel,
                           const float *angvel, const float *com0) {
    s->rigid_body_centers_of_mass[o] = v3_make(com[0], com[1], com[2]);
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) s->rigid_body_rotations[o].m[a][b] = rot9[3 * a + b];
    s->rigid_body_velocities[o] = v3_make(vel[0], vel[1], vel[2]);
    s->rigid_body_angular_velocities[o] = v3_make(angvel[0], angvel[1], angvel[2]);
    if (com0) s->rigid_body_original_centers_of_mass[o] = v3_make(com0[0], com0[1], com0[2]);
}

int sphref_particle_num(SphRef *s) { return s->particle_num; }
int sphref_fluid_particle_num(SphRef *s) { return s->fluid_particle_num; }
long long sphref_last_pairs(SphRef *s) { return s->last_pairs; }

void *sphref_field(SphRef *s, const char *name) {
#define F(n) if (!strcmp(name, #n)) return (void *)s->n;
    F(grid_num_particles) F(grid_num_particles_temp) F(particle_object_ids) F(particle_positions)
    F(particle_velocities) F(particle_accelerations) F(particle_rest_volumes) F(particle_masses)
    F(particle_densities) F(particle_pressures) F(particle_materials) F(particle_colors)
    F(particle_is_dynamic) F(rigid_particle_original_positions) F(grid_ids) F(grid_ids_new)
    F(particle_dfsph_alphas) F(particle_dfsph_kappa) F(particle_dfsph_kappa_v) F(particle_densities_star)
    F(particle_densities_derivatives) F(particle_pressure_accelerations) F(particle_predicted_velocities)
    F(particle_predicted_positions) F(cg_p) F(original_velocity) F(cg_Ap) F(cg_x) F(cg_b) F(cg_r)
    F(cg_diagnol_ii_inv) F(rigid_body_forces) F(rigid_body_torques)
#undef F
    return NULL;
}

double sphref_scalar(SphRef *s, const char *name) {
    if (!strcmp(name, "density_error")) return s->density_error;
    if (!strcmp(name, "pcisph_k")) return s->pcisph_k;
    if (!strcmp(name, "cg_error")) return s->cg_error;
    if (!strcmp(name, "total_time")) return s->total_time;
    if (!strcmp(name, "last_iter_div")) return s->last_iter_div;
    if (!strcmp(name, "last_iter_den")) return s->last_iter_den;
    if (!strcmp(name, "last_iter_pci")) return s->last_iter_pci;
    if (!strcmp(name, "last_iter_cg")) return s->last_iter_cg;
    if (!strcmp(name, "last_err_div")) return s->last_err_div;
    if (!strcmp(name, "last_err_den")) return s->last_err_den;
    if (!strcmp(name, "last_err_pci")) return s->last_err_pci;
    if (!strcmp(name, "last_err_cg")) return s->last_err_cg;
    return NAN;
}

/* ------------------------------------------------------------------ grid */

/* base_container.py:468 pos_to_index: (pos / grid_size).cast(int), truncation toward zero.
   Deviation: clamped into the grid (the reference has no bounds check). */
static inline void pos_to_index(const SphRef *s, v3 pos, int idx[3]) {
    float c[3] = {pos.x / s->grid_size, pos.y / s->grid_size, pos.z / s->grid_size};
    for (int a = 0; a < 3; a++) {
        int i = (int)c[a];
        if (i < 0) i = 0;
        if (i > s->prm.grid_num[a] - 1) i = s->prm.grid_num[a] - 1;
        idx[a] = i;
    }
}

/* base_container.py:473 flatten_grid_index: (ix*ny + iy)*nz + iz */
static inline int flatten_grid_index(const SphRef *s, const int idx[3]) {
    return idx[0] * s->prm.grid_num[1] * s->prm.grid_num[2] + idx[1] * s->prm.grid_num[2] + idx[2];
}

/* base_container.py:496 */
void sphref_init_grid(SphRef *s) {
#pragma omp parallel for schedule(static)
    for (int c = 0; c < s->G; c++) s->grid_num_particles[c] = 0;
#pragma omp parallel for schedule(static)   /* per-particle, independent; the histogram is an integer atomic add like :501 (order-free) */
    for (int p = 0; p < s->particle_num; p++) {
        int idx[3];
        pos_to_index(s, s->particle_positions[p], idx);
        const int g = flatten_grid_index(s, idx);
        s->grid_ids[p] = g;
#pragma omp atomic
        s->grid_num_particles[g] += 1;
    }
#pragma omp parallel for schedule(static)
    for (int c = 0; c < s->G; c++) s->grid_num_particles_temp[c] = s->grid_num_particles[c];
}

/* base_container.py:546 ti.algorithms.PrefixSumExecutor.run -- inclusive, in place
   (semantics fixed by its use at :513-515 and :555-557).  Integer sums: blocked two-pass scan, any number of threads. */
void sphref_prefix_sum(SphRef *s) {
    enum { NB = 256 };
    int tot[NB + 1];
    const int G = s->G, bs = (G + NB - 1) / NB;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < NB; b++) {
        int acc = 0;
        const int lo = b * bs, hi = lo + bs < G ? lo + bs : G;
        for (int c = lo; c < hi; c++) { acc += s->grid_num_particles[c]; s->grid_num_particles[c] = acc; }
        tot[b + 1] = acc;
    }
    tot[0] = 0;
    for (int b = 0; b < NB; b++) tot[b + 1] += tot[b];
#pragma omp parallel for schedule(static)
    for (int b = 1; b < NB; b++) {
        const int lo = b * bs, hi = lo + bs < G ? lo + bs : G, add = tot[b];
        for (int c = lo; c < hi; c++) s->grid_num_particles[c] += add;
    }
}

/* base_container.py:506 -- serial execution order => stable counting sort */
void sphref_reorder_particles(SphRef *s) {
    const int N = s->particle_num;
    for (int i = 0; i < N; i++) {
        int p = N - 1 - i;
        int base_offset = 0;
        if (s->grid_ids[p] - 1 >= 0) base_offset = s->grid_num_particles[s->grid_ids[p] - 1];
        int old = s->grid_num_particles_temp[s->grid_ids[p]];
        s->grid_num_particles_temp[s->grid_ids[p]] = old - 1; /* ti.atomic_sub returns the old value */
        s->grid_ids_new[p] = old - 1 + base_offset;
    }
#pragma omp parallel for schedule(static)   /* grid_ids_new is a permutation: every target written once */
    for (int p = 0; p < N; p++) {
        int n = s->grid_ids_new[p];
        s->grid_ids_buffer[n] = s->grid_ids[p];
        s->particle_object_ids_buffer[n] = s->particle_object_ids[p];
        s->rigid_particle_original_positions_buffer[n] = s->rigid_particle_original_positions[p];
        s->particle_positions_buffer[n] = s->particle_positions[p];
        s->particle_velocities_buffer[n] = s->particle_velocities[p];
        s->particle_rest_volumes_buffer[n] = s->particle_rest_volumes[p];
        s->particle_masses_buffer[n] = s->particle_masses[p];
        s->particle_densities_buffer[n] = s->particle_densities[p];
        s->particle_materials_buffer[n] = s->particle_materials[p];
        for (int c = 0; c < 3; c++) s->particle_colors_buffer[3 * n + c] = s->particle_colors[3 * p + c];
        s->is_dynamic_buffer[n] = s->particle_is_dynamic[p];
    }
    /* the copy-back of base_container.py:532-542, one attribute per thread group */
#pragma omp parallel for schedule(static)
    for (int p = 0; p < N; p++) {
        s->grid_ids[p] = s->grid_ids_buffer[p];
        s->particle_object_ids[p] = s->particle_object_ids_buffer[p];
        s->rigid_particle_original_positions[p] = s->rigid_particle_original_positions_buffer[p];
        s->particle_positions[p] = s->particle_positions_buffer[p];
        s->particle_velocities[p] = s->particle_velocities_buffer[p];
        s->particle_rest_volumes[p] = s->particle_rest_volumes_buffer[p];
        s->particle_masses[p] = s->particle_masses_buffer[p];
        s->particle_densities[p] = s->particle_densities_buffer[p];
        s->particle_materials[p] = s->particle_materials_buffer[p];
        for (int c = 0; c < 3; c++) s->particle_colors[3 * p + c] = s->particle_colors_buffer[3 * p + c];
        s->particle_is_dynamic[p] = s->is_dynamic_buffer[p];
    }
}

/* base_container.py:544 */
void sphref_prepare_neighborhood_search(SphRef *s) {
    sphref_init_grid(s);
    sphref_prefix_sum(s);
    sphref_reorder_particles(s);
}

/* base_container.py:550 for_all_neighbors.  BODY sees p_i, p_j. */
#define FOR_ALL_NEIGHBORS(s, p_i, BODY)                                                          \
    do {                                                                                         \
        int cc_[3];                                                                              \
        pos_to_index((s), (s)->particle_positions[p_i], cc_);                                    \
        for (int ox_ = -1; ox_ <= 1; ox_++)                                                      \
            for (int oy_ = -1; oy_ <= 1; oy_++)                                                  \
                for (int oz_ = -1; oz_ <= 1; oz_++) {                                            \
                    int nc_[3] = {cc_[0] + ox_, cc_[1] + oy_, cc_[2] + oz_};                     \
                    if (nc_[0] < 0 || nc_[1] < 0 || nc_[2] < 0 || nc_[0] >= (s)->prm.grid_num[0] \
                        || nc_[1] >= (s)->prm.grid_num[1] || nc_[2] >= (s)->prm.grid_num[2])     \
                        continue;                                                                \
                    int gi_ = flatten_grid_index((s), nc_);                                      \
                    int start_ = 0, end_ = (s)->grid_num_particles[gi_];                         \
                    if (gi_ - 1 >= 0) start_ = (s)->grid_num_particles[gi_ - 1];                 \
                    for (int p_j = start_; p_j < end_; p_j++) {                                  \
                        if (p_i != p_j &&                                                        \
                            v3_norm(v3_sub((s)->particle_positions[p_i], (s)->particle_positions[p_j])) < (s)->dh) { \
                            npairs_++;                                                           \
                            BODY                                                                 \
                        }                                                                        \
                    }                                                                            \
                }                                                                                \
    } while (0)

/* ---------------------------------------------------------------- kernels */

/* base_solver.py:57 kernel_W (3-D branch) */
static inline float kernel_W(const SphRef *s, float R_mod) {
    float res = 0.0f;
    const double hd = s->prm.support_radius;
    float k = (float)(8.0 / M_PI);
    k /= (float)(hd * hd * hd);
    float q = R_mod / s->dh;
    if (q <= 1.0f) {
        if (q <= 0.5f) {
            float q2 = q * q;
            float q3 = q2 * q;
            res = k * (6.0f * q3 - 6.0f * q2 + 1.0f);
        } else {
            res = k * 2.0f * powf(1.0f - q, 3.0f);
        }
    }
    return res;
}

/* base_solver.py:81 kernel_gradient (3-D branch) */
static inline v3 kernel_gradient(const SphRef *s, v3 R) {
    const double hd = s->prm.support_radius;
    float k = (float)(8.0 / M_PI);
    k = 6.0f * k / (float)(hd * hd * hd);
    float R_mod = v3_norm(R);
    float q = R_mod / s->dh;
    v3 res = v3_make(0.0f, 0.0f, 0.0f);
    if (R_mod > 1e-5f && q <= 1.0f) {
        v3 grad_q = v3_div(R, R_mod * s->dh);
        if (q <= 0.5f) {
            res = v3_scale_l(k * q * (3.0f * q - 2.0f), grad_q);
        } else {
            float factor = 1.0f - q;
            res = v3_scale_l(k * (-factor * factor), grad_q);
        }
    }
    return res;
}

/* base_solver.py:106 compute_rigid_particle_volume (+task :117) */
void sphref_compute_rigid_particle_volume(SphRef *s) {
    long long npairs_ = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : npairs_)
    for (int p_i = 0; p_i < s->particle_num; p_i++) {
        if (s->particle_materials[p_i] == SPHREF_MAT_RIGID) {
            if (s->particle_positions[p_i].y <= s->g_upper) {
                float ret = kernel_W(s, 0.0f);
                FOR_ALL_NEIGHBORS(s, p_i, {
                    if (s->particle_object_ids[p_j] == s->particle_object_ids[p_i]) {
                        v3 R = v3_sub(s->particle_positions[p_i], s->particle_positions[p_j]);
                        ret += kernel_W(s, v3_norm(R));
                    }
                });
                s->particle_rest_volumes[p_i] = 1.0f / ret;
                s->particle_masses[p_i] = s->density_0 * s->particle_rest_volumes[p_i];
            }
        }
    }
    (void)npairs_; /* rigid-rigid pairs are not part of the pair metric (i must be fluid) */
}

/* base_solver.py:522 compute_density (+task :535) */
void sphref_compute_density(SphRef *s) {
    long long npairs_ = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : npairs_)
    for (int p_i = 0; p_i < s->particle_num; p_i++) {
        if (s->particle_materials[p_i] == SPHREF_MAT_FLUID) {
            float den = s->particle_rest_volumes[p_i] * kernel_W(s, 0.0f);
            float ret_i = 0.0f;
            FOR_ALL_NEIGHBORS(s, p_i, {
                v3 R = v3_sub(s->particle_positions[p_i], s->particle_positions[p_j]);
                ret_i += s->particle_rest_volumes[p_j] * kernel_W(s, v3_norm(R));
            });
            den += ret_i;
            den *= s->density_0;
            s->particle_densities[p_i] = den;
        }
    }
    s->last_pairs += npairs_;
}

/* WCSPH.py:17 compute_pressure (stiffness 50000, gamma 7: WCSPH.py:12-13) */
void sphref_wcsph_compute_pressure(SphRef *s) {
    const float stiffness = 50000.0f, gamma = 7.0f;
#pragma omp parallel for schedule(static)
    for (int p_i = 0; p_i < s->particle_num; p_i++) {
        if (s->particle_materials[p_i] == SPHREF_MAT_FLUID) {
            float rho_i = s->particle_densities[p_i];
            rho_i = fmaxf(rho_i, s->density_0);
            s->particle_densities[p_i] = rho_i;
            s->particle_pressures[p_i] = stiffness * (powf(rho_i / s->density_0, gamma) - 1.0f);
        }
    }
}

/* rigid_body_forces / _torques accumulate f32 sums over (fluid particle, rigid neighbour) pairs (base_solver.py:174-187, :272-278;
 * DFSPH.py:173-203).  The reference's serial semantics add them in the order p_i ascending, neighbours in walk order.  A parallel
 * loop that adds under a lock does it in whatever order the threads arrive (f32 addition does not commute in the last bits): the
 * checker was noisier than the product, whose wrench is bit-reproducible since round 4.  Now every thread LOGS its pairs (a particle
 * is walked by one thread, so its records are consecutive and in walk order), and flush_rigid_wrench() -- called behind each of the
 * three loops -- sorts the records by particle and adds them up serially: the same bits as a single-threaded run, whatever the
 * thread count or schedule. */
typedef struct { int p_i, obj; long long seq; v3 f, t; } WrenchRec;
#define SPHREF_MAX_THREADS 512
static WrenchRec *wlog_[SPHREF_MAX_THREADS];
static int wlen_[SPHREF_MAX_THREADS], wcap_[SPHREF_MAX_THREADS];

static int omp_tid_(void) {
#ifdef _OPENMP
    extern int omp_get_thread_num(void);
    return omp_get_thread_num() % SPHREF_MAX_THREADS;
#else
    return 0;
#endif
}

static void add_rigid_wrench(SphRef *s, int p_i, int obj, v3 force, v3 torque) {
    (void)s;
    const int t = omp_tid_();
    if (wlen_[t] == wcap_[t]) {
        wcap_[t] = wcap_[t] ? 2 * wcap_[t] : 4096;
        wlog_[t] = (WrenchRec *)realloc(wlog_[t], sizeof(WrenchRec) * (size_t)wcap_[t]);
    }
    WrenchRec *r = &wlog_[t][wlen_[t]];
    r->p_i = p_i; r->obj = obj; r->seq = wlen_[t]; r->f = force; r->t = torque;
    wlen_[t]++;
}

static int wrench_cmp_(const void *a, const void *b) {
    const WrenchRec *x = (const WrenchRec *)a, *y = (const WrenchRec *)b;
    if (x->p_i != y->p_i) return x->p_i < y->p_i ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0);   /* (same particle = same thread: its own order) */
}

static void flush_rigid_wrench(SphRef *s) {
    size_t total = 0;
    for (int t = 0; t < SPHREF_MAX_THREADS; t++) total += (size_t)wlen_[t];
    if (!total) return;
    WrenchRec *all = (WrenchRec *)malloc(sizeof(WrenchRec) * total);
    size_t k = 0;
    for (int t = 0; t < SPHREF_MAX_THREADS; t++) {
        if (wlen_[t]) memcpy(all + k, wlog_[t], sizeof(WrenchRec) * (size_t)wlen_[t]);
        k += (size_t)wlen_[t];
        wlen_[t] = 0;
    }
    qsort(all, total, sizeof(WrenchRec), wrench_cmp_);
    for (k = 0; k < total; k++) {
        s->rigid_body_forces[all[k].obj] = v3_add(s->rigid_body_forces[all[k].obj], all[k].f);
        s->rigid_body_torques[all[k].obj] = v3_add(s->rigid_body_torques[all[k].obj], all[k].t);
    }
    free(all);
}

/* base_solver.py:136 compute_pressure_acceleration (+task :147) */
void sphref_compute_pressure_acceleration(SphRef *s) {
    long long npairs_ = 0;
    memset(s->particle_accelerations, 0, sizeof(v3) * (size_t)s->prm.particle_max_num);
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : npairs_)
    for (int p_i = 0; p_i < s->particle_num; p_i++) {
        if (s->particle_is_dynamic[p_i]) {
            s->particle_accelerations[p_i] = v3_make(0, 0, 0);
            if (s->particle_materials[p_i] == SPHREF_MAT_FLUID) {
                v3 ret = v3_make(0, 0, 0);
                FOR_ALL_NEIGHBORS(s, p_i, {
                    v3 pos_i = s->particle_positions[p_i];
                    v3 pos_j = s->particle_positions[p_j];
                    float den_i = s->particle_densities[p_i];
                    v3 R = v3_sub(pos_i, pos_j);
                    v3 nabla_ij = kernel_gradient(s, R);
                    if (s->particle_materials[p_j] == SPHREF_MAT_FLUID) {
                        float den_j = s->particle_densities[p_j];
                        float c = -s->particle_masses[p_j] *
                                  (s->particle_pressures[p_i] / (den_i * den_i) +
                                   s->particle_pressures[p_j] / (den_j * den_j));
                        ret = v3_add(ret, v3_scale_l(c, nabla_ij));
                    } else if (s->particle_materials[p_j] == SPHREF_MAT_RIGID) {
                        float c = -s->density_0 * s->particle_rest_volumes[p_j] * s->particle_pressures[p_i] /
                                  (den_i * den_i);
                        ret = v3_add(ret, v3_scale_l(c, nabla_ij));
                        if (s->particle_is_dynamic[p_j]) {
                            int object_j = s->particle_object_ids[p_j];
                            v3 com_j = s->rigid_body_centers_of_mass[object_j];
                            float cf = s->density_0 * s->particle_rest_volumes[p_j] * s->particle_pressures[p_i] /
                                       (den_i * den_i);
                            v3 force_j = v3_scale(v3_scale_l(cf, nabla_ij),
                                                  s->density_0 * s->particle_rest_volumes[p_i]);
                            v3 torque_j = v3_cross(v3_sub(pos_i, com_j), force_j);
                            add_rigid_wrench(s, p_i, object_j, force_j, torque_j);
                        }
                    }
                });
                s->particle_accelerations[p_i] = ret;
            }
        }
    }
    flush_rigid_wrench(s);
    s->last_pairs += npairs_;
}

/* base_solver.py:203 */
void sphref_compute_gravity_acceleration(SphRef *s) {
    v3 g = v3_make((float)s->prm.gravity[0], (float)s->prm.gravity[1], (float)s->prm.gravity[2]);
#pragma omp parallel for schedule(static)
    for (int p_i = 0; p_i < s->particle_num; p_i++)
        if (s->particle_materials[p_i] == SPHREF_MAT_FLUID) s->particle_accelerations[p_i] = g;
}

/* base_solver.py:210 (+task :218) */
void sphref_compute_surface_tension_acceleration(SphRef *s) {
    long long npairs_ = 0;
    const float st = (float)s->prm.surface_tension;
    const double dd = 2.0 * s->prm.particle_radius;
    const float diameter2 = (float)(dd * dd);
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : npairs_)
    for (int p_i = 0; p_i < s->particle_num; p_i++) {
        if (s->particle_materials[p_i] == SPHREF_MAT_FLUID) {
            v3 a_i = v3_make(0, 0, 0);
            FOR_ALL_NEIGHBORS(s, p_i, {
                if (s->particle_materials[p_j] == SPHREF_MAT_FLUID) {
                    v3 R = v3_sub(s->particle_positions[p_i], s->particle_positions[p_j]);
                    float R2 = v3_dot(R, R);
                    float c = st / s->particle_masses[p_i] * s->particle_masses[p_j];
                    float w;
                    if (R2 > diameter2) w = kernel_W(s, v3_norm(R));
                    else w = kernel_W(s, v3_norm(v3_make(s->diameter, 0.0f, 0.0f)));
                    a_i = v3_sub(a_i, v3_scale(v3_scale_l(c, R), w));
                }
            });
            s->particle_accelerations[p_i] = v3_add(s->particle_accelerations[p_i], a_i);
        }
    }
    s->last_pairs += npairs_;
}

/* base_solver.py:232 (+task :240) */
void sphref_compute_viscosity_acceleration_standard(SphRef *s) {
    long long npairs_ = 0;
    const float cv = (float)(2 * (3 + 2) * s->prm.viscosity);
    const float cvb = (float)(2 * (3 + 2) * s->prm.viscosity_b);
    const float eps = (float)(0.01 * s->prm.support_radius * s->prm.support_radius);
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : npairs_)
    for (int p_i = 0; p_i < s->particle_num; p_i++) {
        if (s->particle_materials[p_i] == SPHREF_MAT_FLUID) {
            v3 a_i = v3_make(0, 0, 0);
            FOR_ALL_NEIGHBORS(s, p_i, {
                v3 pos_i = s->particle_positions[p_i];
                v3 pos_j = s->particle_positions[p_j];
                v3 R = v3_sub(pos_i, pos_j);
                v3 nabla_ij = kernel_gradient(s, R);
                float v_xy = v3_dot(v3_sub(s->particle_velocities[p_i], s->particle_velocities[p_j]), R);
                float Rn = v3_norm(R);
                if (s->particle_materials[p_j] == SPHREF_MAT_FLUID) {
                    float m_ij = (s->particle_masses[p_i] + s->particle_masses[p_j]) / 2.0f;
                    float c = cv * m_ij / s->particle_densities[p_j] / (Rn * Rn + eps) * v_xy;
                    a_i = v3_add(a_i, v3_scale_l(c, nabla_ij));
                } else if (s->particle_materials[p_j] == SPHREF_MAT_RIGID) {
                    float m_ij = s->density_0 * s->particle_rest_volumes[p_j];
                    float c = cvb * m_ij / s->particle_densities[p_i] / (Rn * Rn + eps) * v_xy;
                    v3 acc = v3_scale_l(c, nabla_ij);
                    a_i = v3_add(a_i, acc);
                    if (s->particle_is_dynamic[p_j]) {
                        int object_j = s->particle_object_ids[p_j];
                        v3 com_j = s->rigid_body_centers_of_mass[object_j];
                        v3 force_j = v3_div(v3_scale(v3_make(-acc.x, -acc.y, -acc.z), s->particle_masses[p_i]),
                                            s->density_0);
                        v3 torque_j = v3_cross(v3_sub(pos_j, com_j), force_j);
                        add_rigid_wrench(s, p_i, object_j, force_j, torque_j);
                    }
                }
            });
            s->particle_accelerations[p_i] = v3_add(s->particle_accelerations[p_i], v3_div(a_i, s->density_0));
        }
    }
    flush_rigid_wrench(s);
    s->last_pairs += npairs_;
}

/* ------------------------------------------------ implicit viscosity (CG) */

static inline v3 m3_mulv(const m3 *A, v3 x) {
    return v3_make(A->m[0][0] * x.x + A->m[0][1] * x.y + A->m[0][2] * x.z,
                   A->m[1][0] * x.x + A->m[1][1] * x.y + A->m[1][2] * x.z,
                   A->m[2][0] * x.x + A->m[2][1] * x.y + A->m[2][2] * x.z);
}
static inline m3 m3_mul(const m3 *A, const m3 *B) {
    m3 C;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            C.m[i][j] = A->m[i][0] * B->m[0][j] + A->m[i][1] * B->m[1][j] + A->m[i][2] * B->m[2][j];
    return C;
}
/* ti.math.inverse for 3x3 (Taichi lang/matrix_ops: cofactor form, 1/det first). Restated from
   memory of the Taichi >= 1.6 source; Taichi is not vendored by the reference. */
static inline m3 m3_inverse(const m3 *M) {
#define E(x, y) M->m[(x) % 3][(y) % 3]
    float det = M->m[0][0] * (M->m[1][1] * M->m[2][2] - M->m[2][1] * M->m[1][2]) -
                M->m[1][0] * (M->m[0][1] * M->m[2][2] - M->m[2][1] * M->m[0][2]) +
                M->m[2][0] * (M->m[0][1] * M->m[1][2] - M->m[1][1] * M->m[0][2]);
    float inv_det = 1.0f / det;
    m3 R;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            R.m[j][i] = inv_det * (E(i + 1, j + 1) * E(i + 2, j + 2) - E(i + 2, j + 1) * E(i + 1, j + 2));
#undef E
    return R;
}

/* base_solver.py:349 compute_A_ij */
static inline m3 compute_A_ij(const SphRef *s, int p_i, int p_j) {
    m3 A;
    memset(&A, 0, sizeof(A));
    const float eps = (float)(0.01 * s->prm.support_radius * s->prm.support_radius);
    v3 R = v3_sub(s->particle_positions[p_i], s->particle_positions[p_j]);
    v3 nabla_ij = kernel_gradient(s, R);
    float c = 0.0f;
    int any = 0;
    if (s->particle_materials[p_j] == SPHREF_MAT_FLUID) {
        float m_ij = (s->particle_masses[p_i] + s->particle_masses[p_j]) / 2.0f;
        c = (float)(-2 * (3 + 2) * s->prm.viscosity) * m_ij / s->particle_densities[p_j] / (v3_norm_sqr(R) + eps);
        any = 1;
    } else if (s->particle_materials[p_j] == SPHREF_MAT_RIGID) {
        float m_ij = s->density_0 * s->particle_rest_volumes[p_j];
        c = (float)(-2 * (3 + 2) * s->prm.viscosity_b) * m_ij / s->particle_densities[p_i] / (v3_norm_sqr(R) + eps);
        any = 1;
    }
    if (any) {
        float n[3] = {nabla_ij.x, nabla_ij.y, nabla_ij.z}, r[3] = {R.x, R.y, R.z};
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) A.m[a][b] = c * (n[a] * r[b]);
    }
    return A;
}

/* base_solver.py:374 compute_Ap (+task :386) */
static void compute_Ap(SphRef *s) {
    long long npairs_ = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : npairs_)
    for (int p_i = 0; p_i < s->particle_num; p_i++) {
        if (s->particle_materials[p_i] == SPHREF_MAT_FLUID) {
            v3 ret = v3_make(0, 0, 0);
            FOR_ALL_NEIGHBORS(s, p_i, {
                if (s->particle_materials[p_j] == SPHREF_MAT_FLUID) {
                    m3 A = compute_A_ij(s, p_i, p_j);
                    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) A.m[a][b] = -A.m[a][b];
                    m3 M = m3_mul(&s->cg_diagnol_ii_inv[p_i], &A);
                    ret = v3_add(ret, m3_mulv(&M, s->cg_p[p_j]));
                }
            });
            ret = v3_scale(ret, s->dt);
            ret = v3_div(ret, s->density_0);
            ret = v3_add(ret, s->cg_p[p_i]);
            s->cg_Ap[p_i] = ret;
        }
    }
    s->last_pairs += npairs_;
}

/* base_solver.py:282 prepare_conjugate_gradient_solver1 */
static void prepare_cg1(SphRef *s) {
    long long npairs_ = 0;
    size_t nmax = (size_t)s->prm.particle_max_num;
    memset(s->cg_r, 0, sizeof(v3) * nmax); memset(s->cg_p, 0, sizeof(v3) * nmax);
    memset(s->original_velocity, 0, sizeof(v3) * nmax); memset(s->cg_b, 0, sizeof(v3) * nmax);
    memset(s->cg_Ap, 0, sizeof(v3) * nmax);
    const float cvb = (float)(2 * (3 + 2) * s->prm.viscosity_b);
    const float eps = (float)(0.01 * s->prm.support_radius * s->prm.support_radius);
    for (int p_i = 0; p_i < s->particle_num; p_i++)
        if (s->particle_materials[p_i] == SPHREF_MAT_FLUID)
            s->cg_x[p_i] = v3_add(s->cg_x[p_i], s->particle_velocities[p_i]);
    for (int p_i = 0; p_i < s->particle_num; p_i++)
        if (s->particle_materials[p_i] == SPHREF_MAT_FLUID) s->original_velocity[p_i] = s->particle_velocities[p_i];
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : npairs_)
    for (int p_i = 0; p_i < s->particle_num; p_i++) {
        if (s->particle_materials[p_i] == SPHREF_MAT_FLUID) {
            m3 ret;
            memset(&ret, 0, sizeof(ret));
            FOR_ALL_NEIGHBORS(s, p_i, { /* :326 compute_A_ii_task */
                m3 A = compute_A_ij(s, p_i, p_j);
                for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) ret.m[a][b] -= A.m[a][b];
            });
            m3 diag;
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++)
                    diag.m[a][b] = (a == b ? 1.0f : 0.0f) - ret.m[a][b] * s->dt / s->density_0;
            s->cg_diagnol_ii_inv[p_i] = m3_inverse(&diag);
            v3 ret1 = v3_make(0, 0, 0);
            FOR_ALL_NEIGHBORS(s, p_i, { /* :334 compute_b_i_task */
                if (s->particle_materials[p_j] == SPHREF_MAT_RIGID) {
                    v3 R = v3_sub(s->particle_positions[p_i], s->particle_positions[p_j]);
                    v3 nabla_ij = kernel_gradient(s, R);
                    float c = cvb * s->density_0 * s->particle_rest_volumes[p_j] / s->particle_densities[p_i] *
                              v3_dot(s->particle_velocities[p_j], R) / (v3_norm_sqr(R) + eps);
                    ret1 = v3_add(ret1, v3_scale_l(c, nabla_ij));
                }
            });
            s->cg_b[p_i] = v3_sub(s->particle_velocities[p_i], v3_div(v3_scale_l(s->dt, ret1), s->density_0));
            s->cg_p[p_i] = s->cg_x[p_i];
        }
    }
    s->last_pairs += npairs_;
}

/* base_solver.py:509 implicit_viscosity_solve (and :318-:470 helpers) */
void sphref_implicit_viscosity_solve(SphRef *s) {
    const int N = s->particle_num;
    prepare_cg1(s);
    compute_Ap(s);
    for (int p = 0; p < N; p++) /* :318 prepare_conjugate_gradient_solver2 */
        if (s->particle_materials[p] == SPHREF_MAT_FLUID) {
            s->cg_r[p] = v3_sub(m3_mulv(&s->cg_diagnol_ii_inv[p], s->cg_b[p]), s->cg_Ap[p]);
            s->cg_p[p] = s->cg_r[p];
        }
    float tol = 1000.0f;
    int num_itr = 0;
    const int max_itr = s->prm.fixed_iterations > 0 ? s->prm.fixed_iterations : 1000;
    while ((s->prm.fixed_iterations > 0 || tol > 1e-6f) && num_itr < max_itr) { /* :445 conjugate_gradient_loop */
        compute_Ap(s);
        { /* :394 compute_cg_alpha */
            float numerator = 0.0f, denominator = 0.0f;
            for (int p = 0; p < N; p++)
                if (s->particle_materials[p] == SPHREF_MAT_FLUID) {
                    numerator += v3_norm_sqr(s->cg_r[p]);
                    denominator += v3_dot(s->cg_p[p], s->cg_Ap[p]);
                }
            s->cg_alpha = denominator > 1e-18f ? numerator / denominator : 0.0f;
        }
        for (int p = 0; p < N; p++) /* :409 update_cg_x */
            if (s->particle_materials[p] == SPHREF_MAT_FLUID)
                s->cg_x[p] = v3_add(s->cg_x[p], v3_scale_l(s->cg_alpha, s->cg_p[p]));
        { /* :415 update_cg_r_and_beta */
            float numerator = 0.0f, denominator = 0.0f, err = 0.0f;
            for (int p = 0; p < N; p++)
                if (s->particle_materials[p] == SPHREF_MAT_FLUID) {
                    v3 new_r = v3_sub(s->cg_r[p], v3_scale_l(s->cg_alpha, s->cg_Ap[p]));
                    numerator += v3_norm_sqr(new_r);
                    denominator += v3_norm_sqr(s->cg_r[p]);
                    err += v3_norm_sqr(new_r);
                    s->cg_r[p] = new_r;
                }
            s->cg_error = sqrtf(err);
            s->cg_beta = denominator > 1e-18f ? numerator / denominator : 0.0f;
        }
        for (int p = 0; p < N; p++) /* :434 update_p */
            if (s->particle_materials[p] == SPHREF_MAT_FLUID)
                s->cg_p[p] = v3_add(s->cg_r[p], v3_scale_l(s->cg_beta, s->cg_p[p]));
        tol = s->cg_error;
        num_itr++;
    }
    s->last_iter_cg = num_itr;
    s->last_err_cg = tol;
    for (int p = 0; p < N; p++) /* :464 viscosity_update_velocity */
        if (s->particle_materials[p] == SPHREF_MAT_FLUID) s->particle_velocities[p] = s->cg_x[p];
    sphref_compute_viscosity_acceleration_standard(s); /* :515 */
    for (int p = 0; p < N; p++) /* :470 copy_back_original_velocity */
        if (s->particle_materials[p] == SPHREF_MAT_FLUID) s->particle_velocities[p] = s->original_velocity[p];
    for (int p = 0; p < N; p++) /* :440 prepare_guess */
        if (s->particle_materials[p] == SPHREF_MAT_FLUID) s->cg_x[p] = v3_sub(s->cg_x[p], s->original_velocity[p]);
}

/* base_solver.py:190 */
void sphref_compute_non_pressure_acceleration(SphRef *s) {
    sphref_compute_gravity_acceleration(s);
    sphref_compute_surface_tension_acceleration(s);
    if (s->prm.viscosity_implicit) sphref_implicit_viscosity_solve(s);
    else sphref_compute_viscosity_acceleration_standard(s);
}

/* base_solver.py:643 */
void sphref_update_fluid_velocity(SphRef *s) {
#pragma omp parallel for schedule(static)
    for (int p = 0; p < s->particle_num; p++)
        if (s->particle_materials[p] == SPHREF_MAT_FLUID)
            s->particle_velocities[p] = v3_add(s->particle_velocities[p], v3_scale_l(s->dt, s->particle_accelerations[p]));
}

/* base_solver.py:652 (incl. the emitter branch :660-666) */
void sphref_update_fluid_position(SphRef *s) {
#pragma omp parallel for schedule(static)
    for (int p = 0; p < s->particle_num; p++) {
        if (s->particle_materials[p] == SPHREF_MAT_FLUID) {
            s->particle_positions[p] = v3_add(s->particle_positions[p], v3_scale_l(s->dt, s->particle_velocities[p]));
        } else if (s->particle_positions[p].y > s->g_upper) {
            int obj = s->particle_object_ids[p];
            if (obj >= 0 && s->object_materials[obj] == SPHREF_MAT_FLUID) {
                s->particle_positions[p] = v3_add(s->particle_positions[p], v3_scale_l(s->dt, s->particle_velocities[p]));
                if (s->particle_positions[p].y <= s->g_upper) s->particle_materials[p] = SPHREF_MAT_FLUID;
            }
        }
    }
}

/* base_solver.py:670 */
void sphref_prepare_emitter(SphRef *s) {
    for (int p = 0; p < s->particle_num; p++)
        if (s->particle_materials[p] == SPHREF_MAT_FLUID && s->particle_positions[p].y > s->g_upper)
            s->particle_materials[p] = SPHREF_MAT_RIGID;
}

/* base_solver.py:575 enforce_domain_boundary_3D(material_fluid) + :545 simulate_collisions */
void sphref_enforce_domain_boundary_3D(SphRef *s) {
    const float pad = (float)s->prm.padding;
    const float hi[3] = {(float)(s->prm.domain_size[0] - s->prm.padding), (float)(s->prm.domain_size[1] - s->prm.padding),
                         (float)(s->prm.domain_size[2] - s->prm.padding)};
#pragma omp parallel for schedule(static)
    for (int p = 0; p < s->particle_num; p++) {
        if (s->particle_materials[p] == SPHREF_MAT_FLUID && s->particle_is_dynamic[p]) {
            float pos[3] = {s->particle_positions[p].x, s->particle_positions[p].y, s->particle_positions[p].z};
            float out[3] = {pos[0], pos[1], pos[2]};
            float n[3] = {0.0f, 0.0f, 0.0f};
            for (int a = 0; a < 3; a++) {
                if (pos[a] > hi[a]) { n[a] += 1.0f; out[a] = hi[a]; }
                if (pos[a] <= pad) { n[a] += -1.0f; out[a] = pad; }
            }
            s->particle_positions[p] = v3_make(out[0], out[1], out[2]);
            float len = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            if (len > 1e-6f) {
                v3 vec = v3_make(n[0] / len, n[1] / len, n[2] / len);
                const float c_f = 0.5f;
                v3 v = s->particle_velocities[p];
                s->particle_velocities[p] = v3_sub(v, v3_scale_l((1.0f + c_f) * v3_dot(v, vec), vec));
            }
        }
    }
}

/* base_solver.py:616 */
void sphref_renew_rigid_particle_state(SphRef *s) {
    for (int p = 0; p < s->particle_num; p++) {
        if (s->particle_materials[p] == SPHREF_MAT_RIGID && s->particle_is_dynamic[p]) {
            int o = s->particle_object_ids[p];
            if (o >= 0 && s->rigid_body_is_dynamic[o]) {
                v3 q = v3_sub(s->rigid_particle_original_positions[p], s->rigid_body_original_centers_of_mass[o]);
                v3 pp = m3_mulv(&s->rigid_body_rotations[o], q);
                s->particle_positions[p] = v3_add(s->rigid_body_centers_of_mass[o], pp);
                s->particle_velocities[p] = v3_add(s->rigid_body_velocities[o], v3_cross(s->rigid_body_angular_velocities[o], pp));
            }
        }
    }
}

/* ----------------------------------------------------------------- DFSPH */

/* DFSPH.py:23 compute_alpha (+task :48) */
void sphref_dfsph_compute_alpha(SphRef *s) {
    long long npairs_ = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : npairs_)
    for (int p_i = 0; p_i < s->particle_num; p_i++) {
        if (s->particle_materials[p_i] != SPHREF_MAT_FLUID) continue;
        float ret[4] = {0, 0, 0, 0};
        FOR_ALL_NEIGHBORS(s, p_i, {
            v3 g = kernel_gradient(s, v3_sub(s->particle_positions[p_i], s->particle_positions[p_j]));
            v3 grad_p_j = v3_scale_l(-s->particle_rest_volumes[p_j], g);
            if (s->particle_materials[p_j] == SPHREF_MAT_FLUID) {
                ret[3] += v3_norm_sqr(grad_p_j);
                ret[0] += grad_p_j.x; ret[1] += grad_p_j.y; ret[2] += grad_p_j.z;
            } else if (s->particle_materials[p_j] == SPHREF_MAT_RIGID) {
                ret[0] += grad_p_j.x; ret[1] += grad_p_j.y; ret[2] += grad_p_j.z;
            }
        });
        float sum_grad_p_k = ret[3];
        sum_grad_p_k += v3_norm_sqr(v3_make(ret[0], ret[1], ret[2]));
        float factor = 0.0f;
        if (sum_grad_p_k > 1e-5f) factor = 1.0f / sum_grad_p_k;
        s->particle_dfsph_alphas[p_i] = factor;
    }
    s->last_pairs += npairs_;
}

/* DFSPH.py:66 compute_density_derivative (+task :92) */
void sphref_dfsph_compute_density_derivative(SphRef *s) {
    long long npairs_ = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : npairs_)
    for (int p_i = 0; p_i < s->particle_num; p_i++) {
        if (s->particle_materials[p_i] == SPHREF_MAT_FLUID) {
            float density_adv = 0.0f;
            int num_neighbors = 0;
            FOR_ALL_NEIGHBORS(s, p_i, {
                v3 dv = v3_sub(s->particle_velocities[p_i], s->particle_velocities[p_j]);
                v3 g = kernel_gradient(s, v3_sub(s->particle_positions[p_i], s->particle_positions[p_j]));
                density_adv += s->particle_rest_volumes[p_j] * v3_dot(dv, g);
                num_neighbors += 1;
            });
            density_adv = fmaxf(density_adv, 0.0f);
            if (num_neighbors < 20) density_adv = 0.0f;
            s->particle_densities_derivatives[p_i] = density_adv;
        }
    }
    s->last_pairs += npairs_;
}

/* DFSPH.py:105 compute_density_star (+task :118) */
void sphref_dfsph_compute_density_star(SphRef *s) {
    long long npairs_ = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : npairs_)
    for (int p_i = 0; p_i < s->particle_num; p_i++) {
        if (s->particle_materials[p_i] == SPHREF_MAT_FLUID) {
            float delta = 0.0f;
            FOR_ALL_NEIGHBORS(s, p_i, {
                v3 dv = v3_sub(s->particle_velocities[p_i], s->particle_velocities[p_j]);
                v3 g = kernel_gradient(s, v3_sub(s->particle_positions[p_i], s->particle_positions[p_j]));
                delta += s->particle_rest_volumes[p_j] * v3_dot(dv, g);
            });
            float density_adv = s->particle_densities[p_i] / s->density_0 + s->dt * delta;
            s->particle_densities_star[p_i] = fmaxf(density_adv, 1.0f);
        }
    }
    s->last_pairs += npairs_;
}

/* DFSPH.py:162 correct_divergence_step (+task :173) and :246 correct_density_error_step (+task :255).
   in_loop_update=0: dv accumulated then added (divergence); =1: velocity updated pair by pair (density). */
static void dfsph_correct_step(SphRef *s, const float *kappa, int in_loop_update) {
    long long npairs_ = 0;
    const float thr = (float)1e-5 * s->dt; /* m_eps * dt */
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : npairs_)
    for (int p_i = 0; p_i < s->particle_num; p_i++) {
        if (s->particle_materials[p_i] == SPHREF_MAT_FLUID) {
            float k_i = kappa[p_i];
            v3 dv = v3_make(0, 0, 0);
            v3 vel = s->particle_velocities[p_i];
            FOR_ALL_NEIGHBORS(s, p_i, {
                if (s->particle_materials[p_j] == SPHREF_MAT_FLUID) {
                    float k_j = kappa[p_j];
                    float k_sum = k_i + k_j;
                    if (fabsf(k_sum) > thr) {
                        v3 g = kernel_gradient(s, v3_sub(s->particle_positions[p_i], s->particle_positions[p_j]));
                        v3 grad_p_j = v3_scale_l(s->particle_rest_volumes[p_j], g);
                        float c = k_i / s->particle_densities[p_i] + k_j / s->particle_densities[p_j];
                        v3 t = v3_scale(v3_scale(grad_p_j, c), s->density_0);
                        if (in_loop_update) vel = v3_sub(vel, t); else dv = v3_sub(dv, t);
                    }
                } else if (s->particle_materials[p_j] == SPHREF_MAT_RIGID) {
                    float k_sum = k_i;
                    float den_i = s->particle_densities[p_i];
                    if (fabsf(k_sum) > thr) {
                        v3 g = kernel_gradient(s, v3_sub(s->particle_positions[p_i], s->particle_positions[p_j]));
                        v3 grad_p_j = v3_scale_l(s->particle_rest_volumes[p_j], g);
                        v3 t = v3_scale(v3_scale(grad_p_j, k_i / den_i), s->density_0);
                        if (in_loop_update) vel = v3_sub(vel, t); else dv = v3_sub(dv, t);
                        if (s->particle_is_dynamic[p_j]) {
                            int object_j = s->particle_object_ids[p_j];
                            v3 com_j = s->rigid_body_centers_of_mass[object_j];
                            v3 force_j = v3_scale(v3_div(t, s->dt), s->particle_rest_volumes[p_i] * s->density_0);
                            v3 torque_j = v3_cross(v3_sub(s->particle_positions[p_j], com_j), force_j);
                            add_rigid_wrench(s, p_i, object_j, force_j, torque_j);
                        }
                    }
                }
            });
            if (in_loop_update) s->particle_velocities[p_i] = vel;
            else s->particle_velocities[p_i] = v3_add(s->particle_velocities[p_i], dv);
        }
    }
    flush_rigid_wrench(s);
    s->last_pairs += npairs_;
}

/* DFSPH.py:139 correct_divergence_error */
int sphref_dfsph_correct_divergence_error(SphRef *s) {
    const int N = s->particle_num;
    int num_itr = 0;
    sphref_dfsph_compute_density_derivative(s);
    float avg = 0.0f;
    const int max_itr = s->prm.fixed_iterations > 0 ? s->prm.fixed_iterations : 1000;
    while (num_itr < 1 || num_itr < max_itr) {
        for (int p = 0; p < N; p++) /* :133 compute_kappa_v */
            if (s->particle_materials[p] == SPHREF_MAT_FLUID)
                s->particle_dfsph_kappa_v[p] = s->particle_densities_derivatives[p] * s->particle_dfsph_alphas[p];
        dfsph_correct_step(s, s->particle_dfsph_kappa_v, 0);
        sphref_dfsph_compute_density_derivative(s);
        { /* :206 compute_density_derivative_error */
            float e = 0.0f;
            for (int p = 0; p < N; p++)
                if (s->particle_materials[p] == SPHREF_MAT_FLUID) e += s->density_0 * s->particle_densities_derivatives[p];
            avg = e / (float)N;
        }
        double eta = 0.001 * s->prm.density_0 / (double)s->dt;
        num_itr++;
        if (s->prm.fixed_iterations <= 0 && (double)avg <= eta) break;
    }
    s->last_iter_div = num_itr;
    s->last_err_div = avg;
    return num_itr;
}

/* DFSPH.py:225 correct_density_error */
int sphref_dfsph_correct_density_error(SphRef *s) {
    const int N = s->particle_num;
    sphref_dfsph_compute_density_star(s);
    int num_itr = 0;
    float avg = 0.0f;
    const int max_itr = s->prm.fixed_iterations > 0 ? s->prm.fixed_iterations : 1000;
    while (num_itr < 1 || num_itr < max_itr) {
        { /* :218 compute_kappa */
            float delta_t_inv = 1.0f / s->dt;
            for (int p = 0; p < N; p++)
                if (s->particle_materials[p] == SPHREF_MAT_FLUID)
                    s->particle_dfsph_kappa[p] = (s->particle_densities_star[p] - 1.0f) * s->particle_dfsph_alphas[p] * delta_t_inv;
        }
        dfsph_correct_step(s, s->particle_dfsph_kappa, 1);
        sphref_dfsph_compute_density_star(s);
        { /* :286 compute_density_error */
            float e = 0.0f;
            for (int p = 0; p < N; p++)
                if (s->particle_materials[p] == SPHREF_MAT_FLUID) e += s->particle_densities_star[p] - 1.0f;
            avg = e / (float)N;
        }
        num_itr++;
        if (s->prm.fixed_iterations <= 0 && (double)avg <= 0.0001) break;
    }
    s->last_iter_den = num_itr;
    s->last_err_den = avg;
    return num_itr;
}

/* ---------------------------------------------------------------- PCISPH */

/* PCISPH.py:129 compute_pcisph_k */
void sphref_pcisph_compute_k(SphRef *s) {
    float support_radius = s->dh;
    float diam = (float)(2.0 * s->prm.particle_radius * 0.97);
    v3 sumGradW = v3_make(0, 0, 0);
    float sumGradW2 = 0.0f;
    int max_i = (int)(support_radius / diam) + 1;
    for (int i = -max_i; i <= max_i; i++)
        for (int j = -max_i; j <= max_i; j++)
            for (int k = -max_i; k <= max_i; k++) {
                v3 pos_j = v3_make((float)i * diam, (float)j * diam, (float)k * diam);
                v3 x_ij = v3_sub(v3_make(0, 0, 0), pos_j);
                if (v3_norm(x_ij) < support_radius) {
                    v3 nabla = kernel_gradient(s, x_ij);
                    sumGradW = v3_add(sumGradW, nabla);
                    sumGradW2 += v3_norm_sqr(nabla);
                }
            }
    float dtV0 = s->dt * (float)s->prm.V0;
    s->pcisph_k = -0.5f / dtV0 / dtV0 / (v3_norm_sqr(sumGradW) + sumGradW2);
}

/* PCISPH.py:33 compute_density_star (+task :49) */
static void pcisph_compute_density_star(SphRef *s) {
    long long npairs_ = 0;
    float error = 0.0f;
    for (int p_i = 0; p_i < s->particle_num; p_i++) { /* serial: f32 error sum in index order */
        if (s->particle_materials[p_i] == SPHREF_MAT_FLUID) {
            float ret = 0.0f;
            v3 pos_i = s->particle_predicted_positions[p_i];
            FOR_ALL_NEIGHBORS(s, p_i, {
                if (s->particle_materials[p_j] == SPHREF_MAT_FLUID) {
                    v3 R = v3_sub(pos_i, s->particle_predicted_positions[p_j]);
                    ret += s->particle_rest_volumes[p_j] * kernel_W(s, v3_norm(R));
                } else if (s->particle_materials[p_j] == SPHREF_MAT_RIGID) {
                    v3 R = v3_sub(pos_i, s->particle_positions[p_j]);
                    ret += s->particle_rest_volumes[p_j] * kernel_W(s, v3_norm(R));
                }
            });
            s->particle_densities_star[p_i] = ret * s->density_0;
            error += fmaxf(0.0f, ret - 1.0f);
        }
    }
    if (s->fluid_particle_num > 0) s->density_error = error / (float)s->fluid_particle_num;
    else s->density_error = 0.0f;
    s->last_pairs += npairs_;
}

/* PCISPH.py:75 compute_temp_pressure_acceleration (+task :85) */
static void pcisph_compute_temp_pressure_acceleration(SphRef *s) {
    long long npairs_ = 0;
    memset(s->particle_pressure_accelerations, 0, sizeof(v3) * (size_t)s->prm.particle_max_num);
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : npairs_)
    for (int p_i = 0; p_i < s->particle_num; p_i++) {
        if (s->particle_materials[p_i] == SPHREF_MAT_FLUID) {
            v3 ret = v3_make(0, 0, 0);
            FOR_ALL_NEIGHBORS(s, p_i, {
                v3 R = v3_sub(s->particle_positions[p_i], s->particle_positions[p_j]);
                v3 nabla_ij = kernel_gradient(s, R);
                float den_i = s->particle_densities[p_i];
                if (s->particle_materials[p_j] == SPHREF_MAT_FLUID) {
                    float den_j = s->particle_densities[p_j];
                    float c = -s->particle_masses[p_j] * (s->particle_pressures[p_i] / (den_i * den_i) +
                                                          s->particle_pressures[p_j] / (den_j * den_j));
                    ret = v3_add(ret, v3_scale_l(c, nabla_ij));
                } else if (s->particle_materials[p_j] == SPHREF_MAT_RIGID) {
                    float c = -s->density_0 * s->particle_rest_volumes[p_j] * (s->particle_pressures[p_i] / (den_i * den_i));
                    ret = v3_add(ret, v3_scale_l(c, nabla_ij));
                }
            });
            s->particle_pressure_accelerations[p_i] = ret;
        }
    }
    s->last_pairs += npairs_;
}

/* PCISPH.py:154 init_step */
static void pcisph_init_step(SphRef *s) {
    memset(s->particle_pressure_accelerations, 0, sizeof(v3) * (size_t)s->prm.particle_max_num);
    memset(s->particle_pressures, 0, sizeof(float) * (size_t)s->prm.particle_max_num);
    s->density_error = 100.0f;
    for (int p = 0; p < s->particle_num; p++)
        if (s->particle_materials[p] == SPHREF_MAT_FLUID) {
            s->particle_predicted_velocities[p] = v3_add(s->particle_velocities[p], v3_scale_l(s->dt, s->particle_accelerations[p]));
            s->particle_predicted_positions[p] = v3_add(s->particle_positions[p], v3_scale_l(s->dt, s->particle_predicted_velocities[p]));
        }
}

/* PCISPH.py:110 refine */
int sphref_pcisph_refine(SphRef *s) {
    int num_itr = 0;
    const int max_itr = s->prm.fixed_iterations > 0 ? s->prm.fixed_iterations : 1000;
    while (num_itr < max_itr) {
        pcisph_compute_density_star(s);
        for (int p = 0; p < s->particle_num; p++) /* :66 update_pressure */
            if (s->particle_materials[p] == SPHREF_MAT_FLUID) {
                s->particle_pressures[p] += s->pcisph_k * (s->density_0 - s->particle_densities_star[p]);
                if (s->particle_pressures[p] < 0.0f) s->particle_pressures[p] = 0.0f;
            }
        pcisph_compute_temp_pressure_acceleration(s);
        for (int p = 0; p < s->particle_num; p++) /* :19 compute_predicted_velocity */
            if (s->particle_materials[p] == SPHREF_MAT_FLUID)
                s->particle_predicted_velocities[p] =
                    v3_add(s->particle_velocities[p],
                           v3_scale_l(s->dt, v3_add(s->particle_accelerations[p], s->particle_pressure_accelerations[p])));
        for (int p = 0; p < s->particle_num; p++) /* :26 compute_predicted_position */
            if (s->particle_materials[p] == SPHREF_MAT_FLUID)
                s->particle_predicted_positions[p] =
                    v3_add(s->particle_positions[p], v3_scale_l(s->dt, s->particle_predicted_velocities[p]));
        num_itr++;
        if (s->prm.fixed_iterations <= 0 && s->density_error < 0.001f) break;
    }
    s->last_iter_pci = num_itr;
    s->last_err_pci = s->density_error;
    return num_itr;
}

/* -------------------------------------------------------- orchestration */

/* The reference calls rigid_solver.step(), container.insert_object() and rigid_solver.insert_rigid_object() in the
   middle of _step() (WCSPH.py:39-42, DFSPH.py:305-308, PCISPH.py:179-182): both are host-side.  sphref_step_begin()
   runs _step() up to that point, the host wrapper inserts what is due, sphref_step_end() runs the rest of _step()
   and step()'s tail.  sphref_step() = begin + end (nothing to insert). */
void sphref_step_begin(SphRef *s) {
    s->last_pairs = 0;
    if (s->prm.method == 0) {            /* WCSPH.py:27 _step */
        sphref_prepare_neighborhood_search(s);
        sphref_compute_density(s);
        sphref_compute_non_pressure_acceleration(s);
        sphref_update_fluid_velocity(s);
        sphref_wcsph_compute_pressure(s);
        sphref_compute_pressure_acceleration(s);
        sphref_update_fluid_velocity(s);
        sphref_update_fluid_position(s);
    } else if (s->prm.method == 1) {     /* DFSPH.py:298 _step */
        sphref_compute_non_pressure_acceleration(s);
        sphref_update_fluid_velocity(s);
        sphref_dfsph_correct_density_error(s);
        sphref_update_fluid_position(s);
    } else {                             /* PCISPH.py:165 _step */
        sphref_prepare_neighborhood_search(s);
        sphref_compute_density(s);
        sphref_compute_non_pressure_acceleration(s);
        pcisph_init_step(s);
        sphref_pcisph_refine(s);
        sphref_update_fluid_velocity(s);
        sphref_compute_pressure_acceleration(s);
        sphref_update_fluid_velocity(s);
        sphref_update_fluid_position(s);
    }
}

void sphref_step_end(SphRef *s) {
    sphref_renew_rigid_particle_state(s);     /* WCSPH.py:43, DFSPH.py:309, PCISPH.py:183 */
    sphref_enforce_domain_boundary_3D(s);     /* WCSPH.py:45, DFSPH.py:311, PCISPH.py:185 */
    if (s->prm.method == 1) {                 /* DFSPH.py:316-319 */
        sphref_prepare_neighborhood_search(s);
        sphref_compute_density(s);
        sphref_dfsph_compute_alpha(s);
        sphref_dfsph_correct_divergence_error(s);
    }
    s->total_time += (double)s->dt;           /* base_solver.py:694 */
    /* :696 -- on the grid of the last sort: particles inserted since then are in no cell range, exactly as in the
       reference (for_all_neighbors reads the prefix sums of the last prepare_neighborhood_search) */
    sphref_compute_rigid_particle_volume(s);
}

/* base_solver.py:683 prepare (particles were inserted by the host before this call) */
void sphref_prepare(SphRef *s) {
    sphref_prepare_emitter(s);
    sphref_renew_rigid_particle_state(s);
    sphref_prepare_neighborhood_search(s);
    sphref_compute_rigid_particle_volume(s);
    if (s->prm.method == 1) { /* DFSPH.py:321 */
        sphref_compute_density(s);
        sphref_dfsph_compute_alpha(s);
    } else if (s->prm.method == 2) { /* PCISPH.py:188 */
        sphref_pcisph_compute_k(s);
    }
}

/* base_solver.py:692 step */
void sphref_step(SphRef *s) {
    sphref_step_begin(s);
    sphref_step_end(s);
}
