/*
 * wafer_oracle.h -- CPU restatement of Wafer's grid::evolve hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under wafer_amd/ may include, link or
 * call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / reported CPU baseline.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   pinned by the reference's own unit-test vectors:
 *     wo_orthogonalise (grid.rs:721-746), wo_norm2 (grid.rs:780-786),
 *     wo_normalise (grid.rs:788-799), work-area dims (grid.rs:748-778),
 *     wo_calculate_r2 (potential.rs:434-443), wo_alphas (potential.rs:445-449),
 *     wo_mu (potential.rs:450-454), wo_trilerp_resize (input.rs:732-824).
 *   PARITY UNPINNED by any reference test or reference-binary output (the
 *   Rust reference cannot be built here: no cargo/rustc, 27 unvendored crates):
 *     wo_evolve, wo_observables, wo_ab, wo_potential (other than r2/alphas/mu),
 *     initial conditions, wo_solve, wo_symmetrise.  These follow the source text line by
 *     line and are cross-checked against discrete analytic eigenpairs and a
 *     dense-matrix construction in tests/test_oracle_physics.py.
 *
 * Array layout is the reference's: C-order [x][y][z], z contiguous, shape
 * (nx+2*ext, ny+2*ext, nz+2*ext) (config.rs:224-238, grid.rs:505-534).
 */
#ifndef WAFER_ORACLE_H
#define WAFER_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* PotentialType, in the order of config.rs:73-104. */
enum {
    WO_POT_NOPOTENTIAL = 0,
    WO_POT_CUBE,
    WO_POT_QUADWELL,
    WO_POT_PERIODIC,
    WO_POT_COULOMB,
    WO_POT_COMPLEXCOULOMB,
    WO_POT_ELIPTICALCOULOMB,
    WO_POT_SIMPLECORNELL,
    WO_POT_FULLCORNELL,
    WO_POT_HARMONIC,
    WO_POT_COMPLEXHARMONIC,
    WO_POT_DODECAHEDRON,
    WO_POT_FROMFILE,
    WO_POT_FROMSCRIPT
};

/* InitialCondition, in the order of config.rs:151-170. */
enum { WO_IC_FROMFILE = 0, WO_IC_GAUSSIAN, WO_IC_COULOMB, WO_IC_CONSTANT, WO_IC_BOOLEAN };

typedef struct {
    int64_t nx, ny, nz; /* config.grid.size (work area) */
    int32_t ext;        /* CentralDifference::ext(): 1, 2, 3 (config.rs:231-237) */
    int32_t potential;  /* WO_POT_* */
    double dn, dt;      /* config.grid.dn / dt */
    double mass;        /* config.mass */
    double sig;         /* config.sig */
} wo_config;

/* grid.rs:17-28, un-normalised. */
typedef struct {
    double energy, norm2, v_infinity, r2;
} wo_observables_t;

/* one row of the convergence table (grid.rs:126-221) */
typedef struct {
    uint64_t step;
    double tau, energy, norm2, v_infinity, r2, diff;
} wo_block_record;

size_t wo_padded_len(const wo_config *c);

/* potential.rs:366-371, 374-391, 394-398 */
double wo_calculate_r2(int64_t ix, int64_t iy, int64_t iz, int64_t nx, int64_t ny, int64_t nz);
double wo_alphas(double mu);
double wo_mu(double t);

/* potential.rs:188-319 (single index) and 46-62 (whole padded grid).
 * Returns non-zero for FromFile/FromScript (ErrorKind::PotentialNotAvailable). */
int wo_potential_at(const wo_config *c, int64_t ix, int64_t iy, int64_t iz, double *out);
int wo_potential_generate(const wo_config *c, double *v);
/* potential.rs:101-110 */
void wo_ab(const wo_config *c, const double *v, double *a, double *b);
/* potential.rs:112-153, 326-363.  *kind: 0 none, 1 scalar (*scalar), 2 array
 * (potsub, UNPADDED nx*ny*nz; may be NULL to query the kind only). */
int wo_potential_sub(const wo_config *c, int *kind, double *scalar, double *potsub);

/* config.rs:577-683.  Gaussian uses the oracle's own counter RNG (the
 * reference's thread_rng is unseeded, hence not reproducible). */
int wo_initial_condition(const wo_config *c, int ic, uint64_t seed, double *phi);

/* grid.rs:454-457 over the work area of a padded array */
double wo_norm2(const wo_config *c, const double *phi);
/* grid.rs:465-468 over the WHOLE padded array of n elements */
void wo_normalise(double *phi, size_t n, double norm2);
/* grid.rs:477-492; w_store = wnum padded arrays of n elements each */
void wo_orthogonalise(int wnum, double *phi, const double *const *w_store, size_t n);
/* config::symmetrise_wavefunction (config.rs:691-728), loop for loop: kind in the order of
 * SymmetryConstraint (config.rs:184-197): 0 NotConstrained (nothing), 1 AboutZ, 2 AntisymAboutZ,
 * 3 AboutY, 4 AntisymAboutY.  The reference hard-codes
 * the SevenPoint frame (offset 3, extent n+6) and would index out of bounds on a narrower one:
 * returns non-zero unless c->ext == 3. */
int wo_symmetrise(const wo_config *c, int kind, double *phi);
/* grid.rs:303-445; potsub_kind/scalar/array as wo_potential_sub */
void wo_observables(const wo_config *c, const double *v, int potsub_kind, double potsub_scalar,
                    const double *potsub, const double *phi, wo_observables_t *out);
/* grid.rs:544-687, same pass structure (stencil into work, copy back, and for
 * wnum>0 norm2 / normalise / Gram-Schmidt every step). */
void wo_evolve(const wo_config *c, int wnum, const double *a, const double *b, double *phi,
               const double *const *w_store, uint64_t steps);
/* one stencil application only, into an unpadded work array (grid.rs:568-664) */
void wo_stencil_step(const wo_config *c, const double *a, const double *b, const double *phi,
                     double *work);

/* grid.rs:50-246 for ONE state: phi in/out, returns number of records written
 * (<= max_records), *converged set as grid.rs:191. */
size_t wo_solve(const wo_config *c, int wnum, const double *v, const double *a, const double *b,
                int potsub_kind, double potsub_scalar, const double *potsub, double *phi,
                const double *const *w_store, double tolerance, uint64_t screen_update,
                int has_max_steps, uint64_t max_steps, wo_block_record *records,
                size_t max_records, int *converged);

/* input.rs:667-716: v (vx,vy,vz) -> out (sx,sy,sz) dense arrays.  Non-zero (nothing written) when an axis of v has
 * a single point: the reference panics there (usize underflow of nx - 1, then an out-of-bounds index). */
int wo_trilerp_resize(const double *v, int64_t vx, int64_t vy, int64_t vz, double *out,
                      int64_t sx, int64_t sy, int64_t sz);

/* same, with the linspace basis built for (bx,by,bz) points (the production call, see .c) */
int wo_trilerp_resize_basis(const double *v, int64_t vx, int64_t vy, int64_t vz, double *out,
                            int64_t sx, int64_t sy, int64_t sz, int64_t bx, int64_t by, int64_t bz);

/* ---- z-windows of grids whose arrays do not fit the host (1024^3, 2048^3; tests/test_gpu_fullsize.py) --------------------
 * The functions above ARE these with the window [0, pz): the same code on the same GLOBAL indices, restricted to the global
 * padded planes [zp0, zp0 + zcount); arrays are [px][py][zcount].  A window is then evolved as a grid of its own -- a config
 * with nz = zcount - 2 ext, wo_evolve / wo_stencil_step unchanged: the stencil has no notion of position -- and after s steps
 * its planes [s ext, zcount - s ext) are those of the global run (a window end that IS the global frame stays valid). */
int wo_potential_generate_zwindow(const wo_config *c, int64_t zp0, int64_t zcount, double *v);
int wo_initial_condition_zwindow(const wo_config *c, int ic, uint64_t seed, int64_t zp0, int64_t zcount, double *phi);
void wo_ab_n(double dt, const double *v, double *a, double *b, size_t n);
/* output planes [zbegin, zbegin + sz) of wo_trilerp_resize_basis's (sx, sy, .) target */
int wo_trilerp_resize_basis_zwindow(const double *v, int64_t vx, int64_t vy, int64_t vz, double *out,
                                    int64_t sx, int64_t sy, int64_t zbegin, int64_t sz, int64_t bx, int64_t by, int64_t bz);

void wo_set_threads(int n);
int wo_get_threads(void);
/* diagnostics of the timed baseline leg (bench.py cpu_baseline): the host's copy bandwidth on the same threads, and
 * the OpenMP runtime's binding policy (omp_proc_bind_t) and place count */
double wo_host_copy_gbps(size_t bytes, int iters);
int wo_thread_placement(int *proc_bind, int *num_places);

#ifdef __cplusplus
}
#endif
#endif
