/*
 * wafer_oracle.c -- CPU restatement of Wafer's grid::evolve hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see wafer_oracle.h for the rules and for which
 * functions are pinned by reference test vectors and which are "parity
 * unpinned").  Compile with -ffp-contract=off: rustc never fuses a*b+c, and
 * every expression below keeps the reference's left-to-right association so
 * that the per-point arithmetic is bit-for-bit what the Rust source spells.
 *
 * Global sums: the reference uses rayon's into_par_iter().sum(), whose
 * association is unspecified and varies run to run.  The oracle therefore
 * sums in extended precision in a fixed order (per x-plane, then over
 * planes), which is independent of the thread count and closer to the exact
 * sum than any association the reference can produce.
 */
#include "wafer_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define WO_PI 3.14159265358979323846264338327950288 /* std::f64::consts::PI */

typedef struct {
    int64_t px, py, pz; /* padded dims, config.rs:224-238 */
    int64_t nx, ny, nz;
    int64_t e;
} wo_dims;

static wo_dims dims_of(const wo_config *c)
{
    wo_dims d;
    d.e = c->ext;
    d.nx = c->nx;
    d.ny = c->ny;
    d.nz = c->nz;
    d.px = c->nx + 2 * d.e;
    d.py = c->ny + 2 * d.e;
    d.pz = c->nz + 2 * d.e;
    return d;
}

/* C-order [x][y][z] offset into a padded array */
#define PIDX(d, i, j, k) ((((size_t)(i)) * (size_t)(d).py + (size_t)(j)) * (size_t)(d).pz + (size_t)(k))
/* C-order offset into an unpadded (work-area sized) array */
#define WIDX(d, i, j, k) ((((size_t)(i)) * (size_t)(d).ny + (size_t)(j)) * (size_t)(d).nz + (size_t)(k))

size_t wo_padded_len(const wo_config *c)
{
    wo_dims d = dims_of(c);
    return (size_t)d.px * (size_t)d.py * (size_t)d.pz;
}

void wo_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int wo_get_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* What the host's memory delivers to the same threads: `iters` copies of `bytes` between two buffers that every thread
 * first touches in the static schedule it copies with (read + written bytes per second, GB/s).  Beside the timed
 * baseline, bench.py prints it: a stencil rate far under this figure says "threads", not "DRAM". */
double wo_host_copy_gbps(size_t bytes, int iters)
{
    const size_t n = bytes / sizeof(double);
    double *src = (double *)malloc(n * sizeof(double)), *dst = (double *)malloc(n * sizeof(double));
    if (!src || !dst || n == 0 || iters < 1) {
        free(src);
        free(dst);
        return 0.0;
    }
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < n; ++p) {
        src[p] = (double)p;
        dst[p] = 0.0;
    }
    double best = 0.0;
    for (int it = 0; it < iters; ++it) {
#ifdef _OPENMP
        const double t0 = omp_get_wtime();
#endif
#pragma omp parallel for schedule(static)
        for (size_t p = 0; p < n; ++p) dst[p] = src[p];
#ifdef _OPENMP
        const double dt = omp_get_wtime() - t0;
        if (dt > 0.0 && 2.0 * (double)bytes / dt / 1e9 > best) best = 2.0 * (double)bytes / dt / 1e9;
#endif
    }
    if (dst[n / 2] != src[n / 2]) best = 0.0; /* (keeps the copy observable) */
    free(src);
    free(dst);
    return best;
}

/* where the OpenMP runtime put its threads: "proc_bind=<policy> places=<count>" (OpenMP 4.5 API) */
int wo_thread_placement(int *proc_bind, int *num_places)
{
#ifdef _OPENMP
    *proc_bind = (int)omp_get_proc_bind();
    *num_places = omp_get_num_places();
    return 0;
#else
    *proc_bind = 0;
    *num_places = 0;
    return 1;
#endif
}

/* ---- fixed-order extended-precision sum of per-plane partials ------------ */
static double sum_planes(const long double *part, int64_t n)
{
    long double s = 0.0L;
    for (int64_t i = 0; i < n; ++i) s += part[i];
    return (double)s;
}

/* ======================================================================== *
 * potential.rs
 * ======================================================================== */

/* potential.rs:366-371 */
double wo_calculate_r2(int64_t ix, int64_t iy, int64_t iz, int64_t nx, int64_t ny, int64_t nz)
{
    double dx = (double)ix - ((double)nx + 1.) / 2.;
    double dy = (double)iy - ((double)ny + 1.) / 2.;
    double dz = (double)iz - ((double)nz + 1.) / 2.;
    return dx * dx + dy * dy + dz * dz;
}

/* potential.rs:374-391 */
double wo_alphas(double mu)
{
    const double nf = 2.0;
    const double b0 = 11. - 2. * nf / 3.;
    const double b1 = 51. - 19. * nf / 3.;
    const double b2 = 2857. - 5033. * nf / 9. + 325. * nf * nf / 27.;
    const double scale = 2.3;
    double l = 2. * log(mu / scale);
    double ll = log(l);
    double t1 = 2. * b1 * ll / (b0 * b0 * l);
    double t2 = 4. * b1 * b1 * ((ll - 0.5) * (ll - 0.5) + b2 * b0 / (8. * b1 * b1) - 5.0 / 4.0) /
                (b0 * b0 * b0 * b0 * l * l);
    return 4. * WO_PI * (1. - t1 + t2) / (b0 * l);
}

/* potential.rs:394-398 */
double wo_mu(double t)
{
    const double nf = 2.0;
    const double tc = 0.2;
    return 1.4 * sqrt((1. + nf / 6.) * 4. * WO_PI * wo_alphas(2. * WO_PI * t)) * t * tc;
}

/* the twelve half-spaces of potential.rs:283-308, each spelt with the
 * reference's own grouping so boundary points classify identically */
static int inside_dodecahedron(double x, double y, double z)
{
    const double A = 12.70820393249937, B = 11.210068307552588, C = 14.674169922690343;
    const double D = 5.605034153776295, D2 = 5.605034153776294;
    const double G = 3.23606797749979, H = 1.2360679774997896;
    const double P = 4.23606797749979, Q = 5.23606797749979;
    const double S = 18.1382715378281, T = 3.464101615137755;
    const double U = 9.06913576891405, W = 15.70820393249937, Y = 9.70820393249937;
    const double Z2 = 6.47213595499958, K = 25.41640786499874;
    const double R3 = 1.7320508075688772, E = 8.47213595499958;
    if (!(A + B * x >= C * z)) return 0;
    if (!(B * x <= A + C * z)) return 0;
    if (!(D * (G * x - H * z) <= 6. * (P + Q * y))) return 0;
    if (!(S * x + T * z <= A)) return 0;
    if (!(U * x + W * y <= A + T * z)) return 0;
    if (!(Y * y <= A + D2 * x + C * z)) return 0;
    if (!(A + D2 * x + Y * y + C * z >= 0.)) return 0;
    if (!(W * y + T * z <= A + U * x)) return 0;
    if (!(D * (-Z2 * x - H * z) <= K)) return 0;
    if (!(T * z <= U * x + 3. * (P + Q * y))) return 0;
    if (!(R3 * (G * x + E * z) <= 3. * (P + G * y))) return 0;
    if (!(D2 * x + Y * y + C * z <= A)) return 0;
    return 1;
}

/* Debye mass with the (fixed) t = 1, xi = 0 of potential.rs:252-260 */
static double fullcornell_md(const wo_config *c, double dz, double r)
{
    const double t = 1.0, xi = 0.0;
    double aniso = 1. - c->dn * c->dn * dz * dz / (r * r);
    return wo_mu(t) * (1. + (0.07 * pow(xi, 0.2)) * aniso) * pow(1. + xi, -0.29);
}

/* potential.rs:188-319; (ix,iy,iz) is a PADDED index, the centre is taken
 * from the unpadded size (potential.rs:52-53, 366-371). */
int wo_potential_at(const wo_config *c, int64_t ix, int64_t iy, int64_t iz, double *out)
{
    const int64_t nx = c->nx, ny = c->ny, nz = c->nz;
    switch (c->potential) {
    case WO_POT_NOPOTENTIAL:
        *out = 0.0;
        return 0;
    case WO_POT_CUBE: /* :192-201 */
        *out = ((ix > nx / 4 && ix <= 3 * nx / 4) && (iy > ny / 4 && iy <= 3 * ny / 4) &&
                (iz > nz / 4 && iz <= 3 * nz / 4))
                   ? -10.0
                   : 0.0;
        return 0;
    case WO_POT_QUADWELL: /* :202-211, short side along z */
        *out = ((ix > nx / 4 && ix <= 3 * nx / 4) && (iy > ny / 4 && iy <= 3 * ny / 4) &&
                (iz > 3 * nz / 8 && iz <= 5 * nz / 8))
                   ? -10.0
                   : 0.0;
        return 0;
    case WO_POT_PERIODIC: { /* :212-220 */
        double sx = sin(2. * WO_PI * ((double)ix - 1.) / ((double)nx - 1.));
        double sy = sin(2. * WO_PI * ((double)iy - 1.) / ((double)ny - 1.));
        double sz = sin(2. * WO_PI * ((double)iz - 1.) / ((double)nz - 1.));
        double temp = sx * sx;
        temp *= sy * sy;
        temp *= sz * sz;
        *out = -temp + 1.;
        return 0;
    }
    case WO_POT_COULOMB:
    case WO_POT_COMPLEXCOULOMB: { /* :221-229 */
        double r = c->dn * sqrt(wo_calculate_r2(ix, iy, iz, nx, ny, nz));
        *out = (r < c->dn) ? -1. / c->dn : -1. / r;
        return 0;
    }
    case WO_POT_ELIPTICALCOULOMB: { /* :230-240 */
        double dx = (double)ix - ((double)nx + 1.) / 2.;
        double dy = (double)iy - ((double)ny + 1.) / 2.;
        double dz = ((double)iz - ((double)nz + 1.) / 2.) * 2.;
        double r = c->dn * sqrt(dx * dx + dy * dy + dz * dz);
        *out = (r < c->dn) ? 0.0 : -1. / r + 1. / c->dn;
        return 0;
    }
    case WO_POT_SIMPLECORNELL: { /* :241-249 */
        double r = c->dn * sqrt(wo_calculate_r2(ix, iy, iz, nx, ny, nz));
        if (r < c->dn)
            *out = 4. * c->mass;
        else
            *out = (-0.5 * (4. / 3.)) / r + c->sig * r + 4. * c->mass;
        return 0;
    }
    case WO_POT_FULLCORNELL: { /* :250-269 */
        const double t = 1.0;
        double dz = (double)iz - ((double)nz + 1.) / 2.;
        double r = c->dn * sqrt(wo_calculate_r2(ix, iy, iz, nx, ny, nz));
        double md = fullcornell_md(c, dz, r);
        if (r < c->dn) {
            *out = 4. * c->mass;
        } else {
            double screen = exp(-md * r);
            *out = (-wo_alphas(2. * WO_PI * t) * (4. / 3.)) * screen / r +
                   c->sig * (1. - screen) / md - (0.8 * c->sig) / (4. * c->mass * c->mass * r) +
                   4. * c->mass;
        }
        return 0;
    }
    case WO_POT_HARMONIC:
    case WO_POT_COMPLEXHARMONIC: { /* :270-274 */
        double r = c->dn * sqrt(wo_calculate_r2(ix, iy, iz, nx, ny, nz));
        *out = r * r / 2.;
        return 0;
    }
    case WO_POT_DODECAHEDRON: { /* :275-314 */
        double dx = (double)ix - ((double)nx + 1.) / 2.;
        double dy = (double)iy - ((double)ny + 1.) / 2.;
        double dz = (double)iz - ((double)nz + 1.) / 2.;
        double x = dx / (((double)nx - 1.) / 2.);
        double y = dy / (((double)ny - 1.) / 2.);
        double z = dz / (((double)nz - 1.) / 2.);
        *out = inside_dodecahedron(x, y, z) ? -100. : 0.0;
        return 0;
    }
    default: /* FromFile / FromScript: ErrorKind::PotentialNotAvailable, :315-317 */
        return 1;
    }
}

/* potential.rs:46-62 on the global padded planes [zp0, zp0 + zcount) only: v is [px][py][zcount].  The same
 * wo_potential_at on the same GLOBAL indices -- for grids whose arrays do not fit the host (1024^3, 2048^3). */
int wo_potential_generate_zwindow(const wo_config *c, int64_t zp0, int64_t zcount, double *v)
{
    wo_dims d = dims_of(c);
    double probe;
    if (wo_potential_at(c, 0, 0, 0, &probe)) return 1;
    if (zp0 < 0 || zcount < 0 || zp0 + zcount > d.pz) return 2;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < d.px; ++i)
        for (int64_t j = 0; j < d.py; ++j)
            for (int64_t k = 0; k < zcount; ++k)
                wo_potential_at(c, i, j, zp0 + k, &v[(((size_t)i) * (size_t)d.py + (size_t)j) * (size_t)zcount + (size_t)k]);
    return 0;
}

/* potential.rs:46-62 */
int wo_potential_generate(const wo_config *c, double *v)
{
    return wo_potential_generate_zwindow(c, 0, c->nz + 2 * (int64_t)c->ext, v);
}

/* potential.rs:101-110 over n elements (elementwise: a z-window has its own length) */
void wo_ab_n(double dt, const double *v, double *a, double *b, size_t n)
{
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < n; ++p) {
        b[p] = 1. / (1. + dt * v[p] / 2.);
        a[p] = (1. - dt * v[p] / 2.) * b[p];
    }
}

/* potential.rs:101-110 */
void wo_ab(const wo_config *c, const double *v, double *a, double *b)
{
    wo_ab_n(c->dt, v, a, b, wo_padded_len(c));
}

/* potential.rs:112-153 (the no-file branch) with :326-363 */
int wo_potential_sub(const wo_config *c, int *kind, double *scalar, double *potsub)
{
    *scalar = 0.0;
    switch (c->potential) {
    case WO_POT_FULLCORNELL: { /* variable_pot_sub(), config.rs:108-127; potential.rs:134-144 */
        *kind = 2;
        if (!potsub) return 0;
        wo_dims d = dims_of(c);
        const double t = 1.0, xi = 0.0;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < d.nx; ++i)
            for (int64_t j = 0; j < d.ny; ++j)
                for (int64_t k = 0; k < d.nz; ++k) {
                    /* potential.rs:328-337: UNPADDED index, and a different
                     * grouping of the Debye-mass factors than :256-260 */
                    double dz = (double)k - ((double)d.nz + 1.) / 2.;
                    double r = c->dn * sqrt(wo_calculate_r2(i, j, k, d.nx, d.ny, d.nz));
                    double md = wo_mu(t) * 1. + (0.07 * pow(xi, 0.2)) *
                                                    (1. - c->dn * c->dn * dz * dz / (r * r)) *
                                                    pow(1. + xi, -0.29);
                    potsub[WIDX(d, i, j, k)] = c->sig / md + 4. * c->mass;
                }
        return 0;
    }
    case WO_POT_ELIPTICALCOULOMB: /* :359 */
        *scalar = 1. / c->dn;
        break;
    case WO_POT_SIMPLECORNELL: /* :360 */
        *scalar = 4.0 * c->mass;
        break;
    default: /* :348-358 */
        *scalar = 0.0;
        break;
    }
    /* potential.rs:148-152: only a strictly positive scalar is kept */
    *kind = (*scalar > 0.0) ? 1 : 0;
    return 0;
}

/* ======================================================================== *
 * config.rs: initial conditions
 * ======================================================================== */

/* counter RNG for the Gaussian IC: splitmix64 finaliser on (seed, counter) */
static uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static double gaussian_at(uint64_t seed, uint64_t counter, double sigma)
{
    uint64_t h1 = mix64(seed ^ mix64(2 * counter));
    uint64_t h2 = mix64(seed ^ mix64(2 * counter + 1));
    /* u1 in (0,1], u2 in [0,1) on a 2^-53 lattice */
    double u1 = ((double)(h1 >> 11) + 1.0) * (1.0 / 9007199254740992.0);
    double u2 = (double)(h2 >> 11) * (1.0 / 9007199254740992.0);
    return sigma * (sqrt(-2.0 * log(u1)) * cos(2.0 * WO_PI * u2));
}

/* config.rs:577-627 (FromFile is the caller's business) on the global padded planes [zp0, zp0 + zcount): phi is
 * [px][py][zcount]; every formula sees the GLOBAL index (the Gaussian counter is the global padded linear index) */
int wo_initial_condition_zwindow(const wo_config *c, int ic, uint64_t seed, int64_t zp0, int64_t zcount, double *phi)
{
    wo_dims d = dims_of(c);
    if (ic != WO_IC_GAUSSIAN && ic != WO_IC_COULOMB && ic != WO_IC_CONSTANT && ic != WO_IC_BOOLEAN)
        return 1;
    if (zp0 < 0 || zcount < 0 || zp0 + zcount > d.pz) return 2;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < d.px; ++i)
        for (int64_t j = 0; j < d.py; ++j)
            for (int64_t kw = 0; kw < zcount; ++kw) {
                const int64_t k = zp0 + kw;
                size_t p = (((size_t)i) * (size_t)d.py + (size_t)j) * (size_t)zcount + (size_t)kw;
                double val;
                /* Dirichlet frame, config.rs:597-622 */
                if (i < d.e || i >= d.px - d.e || j < d.e || j >= d.py - d.e || k < d.e ||
                    k >= d.pz - d.e) {
                    phi[p] = 0.;
                    continue;
                }
                switch (ic) {
                case WO_IC_GAUSSIAN: /* :636-642, own RNG keyed by the padded linear index */
                    val = gaussian_at(seed, (uint64_t)PIDX(d, i, j, k), c->sig);
                    break;
                case WO_IC_COULOMB: { /* :650-669 */
                    double dx = (double)i - ((double)d.px / 2.);
                    double dy = (double)j - ((double)d.py / 2.);
                    double dz = (double)k - ((double)d.pz / 2.);
                    double r = c->dn * sqrt(dx * dx + dy * dy + dz * dz);
                    double costheta = c->dn * dz / r;
                    double cosphi = c->dn * dx / r;
                    double mr2 = exp(-c->mass * r / 2.);
                    val = exp(-c->mass * r) + (2. - c->mass * r) * mr2 +
                          c->mass * r * mr2 * costheta +
                          c->mass * r * mr2 * sqrt(1. - costheta * costheta) * cosphi;
                    break;
                }
                case WO_IC_CONSTANT: /* :593 */
                    val = 0.1;
                    break;
                default: /* Boolean :676-683: ((((i%2)*j)%2)*k)%2 on padded indices */
                    val = fmod(fmod(fmod((double)i, 2.) * (double)j, 2.) * (double)k, 2.);
                    break;
                }
                phi[p] = val;
            }
    return 0;
}

int wo_initial_condition(const wo_config *c, int ic, uint64_t seed, double *phi)
{
    return wo_initial_condition_zwindow(c, ic, seed, 0, c->nz + 2 * (int64_t)c->ext, phi);
}

/* ======================================================================== *
 * grid.rs
 * ======================================================================== */

/* grid.rs:454-457 on get_work_area (grid.rs:505-513) */
double wo_norm2(const wo_config *c, const double *phi)
{
    wo_dims d = dims_of(c);
    long double *part = (long double *)malloc(sizeof(long double) * (size_t)d.nx);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < d.nx; ++i) {
        long double s = 0.0L;
        for (int64_t j = 0; j < d.ny; ++j)
            for (int64_t k = 0; k < d.nz; ++k) {
                double el = phi[PIDX(d, i + d.e, j + d.e, k + d.e)];
                s += (long double)(el * el);
            }
        part[i] = s;
    }
    double r = sum_planes(part, d.nx);
    free(part);
    return r;
}

/* grid.rs:465-468: every element of the padded array, true division */
void wo_normalise(double *phi, size_t n, double norm2)
{
    const double norm = sqrt(norm2);
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < n; ++p) phi[p] /= norm;
}

/* config.rs:691-728, the loops as written (sequential: later cells read cells this
 * pass has already overwritten) */
int wo_symmetrise(const wo_config *c, int kind, double *phi)
{
    if (kind == 0) return 0;
    if (kind < 0 || kind > 4) return 1;
    if (c->ext != 3) return 2; /* w[[sx, 3 + ny, 3 + nz]] is out of bounds for ext < 3 */
    const int64_t nx = c->nx, ny = c->ny, nz = c->nz;
    const int64_t py = ny + 6, pz = nz + 6;
    const double sign = (kind == 2 || kind == 4) ? -1.0 : 1.0;
    const int about_z = (kind == 1 || kind == 2);
#define W(i, j, k) phi[((i) * py + (j)) * pz + (k)]
    for (int64_t sx = 0; sx < nx + 6; ++sx)
        for (int64_t sy = 3; sy < 3 + ny + 1; ++sy) {
            int64_t y = sy;
            if (!about_z && y > (3 + ny) / 2) y = (3 + ny) + 1 - y;
            for (int64_t sz = 3; sz < 3 + nz + 1; ++sz) {
                int64_t z = sz;
                if (about_z && z > (3 + nz) / 2) z = (3 + nz) + 1 - z;
                W(sx, sy, sz) = sign * W(sx, y, z);
            }
        }
#undef W
    return 0;
}

/* grid.rs:477-492: modified Gram-Schmidt, lower states in storage order,
 * with the reference's temporary product array and separate sum pass */
void wo_orthogonalise(int wnum, double *phi, const double *const *w_store, size_t n)
{
    const size_t chunk = 1u << 16;
    const size_t nchunk = (n + chunk - 1) / chunk;
    for (int l = 0; l < wnum; ++l) {
        const double *lower = w_store[l];
        double *overlap = (double *)calloc(n, sizeof(double)); /* grid.rs:482 */
        long double *part = (long double *)malloc(sizeof(long double) * nchunk);
#pragma omp parallel for schedule(static)
        for (size_t p = 0; p < n; ++p) overlap[p] = lower[p] * phi[p];
#pragma omp parallel for schedule(static)
        for (size_t q = 0; q < nchunk; ++q) {
            size_t hi = (q + 1) * chunk < n ? (q + 1) * chunk : n;
            long double s = 0.0L;
            for (size_t p = q * chunk; p < hi; ++p) s += (long double)overlap[p];
            part[q] = s;
        }
        const double overlap_sum = sum_planes(part, (int64_t)nchunk);
#pragma omp parallel for schedule(static)
        for (size_t p = 0; p < n; ++p) phi[p] -= lower[p] * overlap_sum;
        free(part);
        free(overlap);
    }
}

/* The bracketed central-difference sum S of grid.rs:582-588 / 608-620 /
 * 642-659 (identical in compute_observables :326-331 / 350-362 / 382-399),
 * at padded position p, with strides sx (x), sy (y), 1 (z). */
static inline double stencil_sum(const double *phi, size_t p, size_t sx, size_t sy, int ext,
                                 double w)
{
    if (ext == 1) {
        return phi[p + sx] + phi[p - sx] + phi[p + sy] + phi[p - sy] + phi[p + 1] + phi[p - 1] -
               6. * w;
    } else if (ext == 2) {
        return -phi[p + 2 * sx] + 16. * phi[p + sx] + 16. * phi[p - sx] - phi[p - 2 * sx] -
               phi[p + 2 * sy] + 16. * phi[p + sy] + 16. * phi[p - sy] - phi[p - 2 * sy] -
               phi[p + 2] + 16. * phi[p + 1] + 16. * phi[p - 1] - phi[p - 2] - 90. * w;
    } else {
        return 2. * phi[p + 3 * sx] - 27. * phi[p + 2 * sx] + 270. * phi[p + sx] +
               270. * phi[p - sx] - 27. * phi[p - 2 * sx] + 2. * phi[p - 3 * sx] +
               2. * phi[p + 3 * sy] - 27. * phi[p + 2 * sy] + 270. * phi[p + sy] +
               270. * phi[p - sy] - 27. * phi[p - 2 * sy] + 2. * phi[p - 3 * sy] +
               2. * phi[p + 3] - 27. * phi[p + 2] + 270. * phi[p + 1] + 270. * phi[p - 1] -
               27. * phi[p - 2] + 2. * phi[p - 3] - 1470. * w;
    }
}

/* denominators of grid.rs:569 / 594 / 626 (and :314 / 337 / 367) */
static double stencil_denominator(const wo_config *c)
{
    const double lead = (c->ext == 1) ? 2. : (c->ext == 2) ? 24. : 360.;
    return lead * c->dn * c->dn * c->mass;
}

/* grid.rs:568-664: work = w*pa + pb*dt*S/denominator over the work area */
void wo_stencil_step(const wo_config *c, const double *a, const double *b, const double *phi,
                     double *work)
{
    wo_dims d = dims_of(c);
    const double den = stencil_denominator(c);
    const double dt = c->dt;
    const size_t sy = (size_t)d.pz, sx = (size_t)d.py * (size_t)d.pz;
    const int ext = c->ext;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < d.nx; ++i)
        for (int64_t j = 0; j < d.ny; ++j) {
            size_t p0 = PIDX(d, i + d.e, j + d.e, d.e);
            double *wrow = work + WIDX(d, i, j, 0);
            if (ext == 1) {
                for (int64_t k = 0; k < d.nz; ++k) {
                    size_t p = p0 + (size_t)k;
                    double w = phi[p];
                    wrow[k] = w * a[p] + b[p] * dt * stencil_sum(phi, p, sx, sy, 1, w) / den;
                }
            } else if (ext == 2) {
                for (int64_t k = 0; k < d.nz; ++k) {
                    size_t p = p0 + (size_t)k;
                    double w = phi[p];
                    wrow[k] = w * a[p] + b[p] * dt * stencil_sum(phi, p, sx, sy, 2, w) / den;
                }
            } else {
                for (int64_t k = 0; k < d.nz; ++k) {
                    size_t p = p0 + (size_t)k;
                    double w = phi[p];
                    wrow[k] = w * a[p] + b[p] * dt * stencil_sum(phi, p, sx, sy, 3, w) / den;
                }
            }
        }
}

/* grid.rs:544-687 */
void wo_evolve(const wo_config *c, int wnum, const double *a, const double *b, double *phi,
               const double *const *w_store, uint64_t steps_wanted)
{
    wo_dims d = dims_of(c);
    const size_t n = wo_padded_len(c);
    double *work = (double *)calloc((size_t)d.nx * (size_t)d.ny * (size_t)d.nz, sizeof(double));
    uint64_t steps = 0;
    for (;;) {
        wo_stencil_step(c, a, b, phi, work); /* :563-665 */
        /* copy-back pass, :666-673 */
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < d.nx; ++i)
            for (int64_t j = 0; j < d.ny; ++j)
                memcpy(phi + PIDX(d, i + d.e, j + d.e, d.e), work + WIDX(d, i, j, 0),
                       sizeof(double) * (size_t)d.nz);
        if (wnum > 0) { /* :674-681 */
            double norm2 = wo_norm2(c, phi);
            wo_normalise(phi, n, norm2);
            wo_orthogonalise(wnum, phi, w_store, n);
        }
        steps += 1; /* :682-685: at least one step is always taken */
        if (steps >= steps_wanted) break;
    }
    free(work);
}

/* grid.rs:303-445 */
void wo_observables(const wo_config *c, const double *v, int potsub_kind, double potsub_scalar,
                    const double *potsub, const double *phi, wo_observables_t *out)
{
    wo_dims d = dims_of(c);
    const double den = stencil_denominator(c);
    const size_t sy = (size_t)d.pz, sx = (size_t)d.py * (size_t)d.pz;
    const int ext = c->ext;
    long double *pe = (long double *)malloc(sizeof(long double) * 4 * (size_t)d.nx);
    long double *pn = pe + d.nx, *pv = pn + d.nx, *pr = pv + d.nx;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < d.nx; ++i) {
        long double se = 0.0L, sn = 0.0L, sv = 0.0L, sr = 0.0L;
        for (int64_t j = 0; j < d.ny; ++j)
            for (int64_t k = 0; k < d.nz; ++k) {
                size_t p = PIDX(d, i + d.e, j + d.e, k + d.e);
                double w = phi[p];
                double vv = v[p];
                /* :325-332: v*w*w - w*S/denominator */
                double e = vv * w * w - w * stencil_sum(phi, p, sx, sy, ext, w) / den;
                se += (long double)e;
                sn += (long double)(w * w); /* :407 */
                if (potsub_kind == 2)       /* :410-418, unpadded array */
                    sv += (long double)(w * w * potsub[WIDX(d, i, j, k)]);
                else if (potsub_kind == 1) /* :419-424 */
                    sv += (long double)(w * w * potsub_scalar);
                /* :428-437: r2 from the WORK-AREA index */
                sr += (long double)(w * w * wo_calculate_r2(i, j, k, d.nx, d.ny, d.nz));
            }
        pe[i] = se;
        pn[i] = sn;
        pv[i] = sv;
        pr[i] = sr;
    }
    out->energy = sum_planes(pe, d.nx);
    out->norm2 = sum_planes(pn, d.nx);
    out->v_infinity = (potsub_kind == 0) ? 0. : sum_planes(pv, d.nx);
    out->r2 = sum_planes(pr, d.nx);
    free(pe);
}

/* grid.rs:50-246 for one state, without the snapshot branch (:137-158) */
size_t wo_solve(const wo_config *c, int wnum, const double *v, const double *a, const double *b,
                int potsub_kind, double potsub_scalar, const double *potsub, double *phi,
                const double *const *w_store, double tolerance, uint64_t screen_update,
                int has_max_steps, uint64_t max_steps, wo_block_record *records,
                size_t max_records, int *converged)
{
    const size_t n = wo_padded_len(c);
    uint64_t step = 0;
    double last_energy = DBL_MAX; /* :124 */
    size_t nrec = 0;
    *converged = 0;
    for (;;) {
        wo_observables_t obs;
        wo_observables(c, v, potsub_kind, potsub_scalar, potsub, phi, &obs); /* :127 */
        double norm_energy = obs.energy / obs.norm2;                          /* :128 */
        double tau = (double)step * c->dt;                                    /* :129 */
        wo_normalise(phi, n, obs.norm2);                                      /* :130 */
        if (wnum > 0) wo_orthogonalise(wnum, phi, w_store, n);                /* :133-135 */
        double diff = fabs(norm_energy - last_energy);                        /* :161 */
        if (nrec < max_records) {
            wo_block_record *r = &records[nrec];
            r->step = step;
            r->tau = tau;
            r->energy = obs.energy;
            r->norm2 = obs.norm2;
            r->v_infinity = obs.v_infinity;
            r->r2 = obs.r2;
            r->diff = diff;
        }
        nrec++;
        if (diff < tolerance) { /* :162-192 */
            *converged = 1;
            break;
        }
        last_energy = norm_energy;                      /* :194 */
        if (has_max_steps && step > max_steps) break;   /* :211-213 */
        wo_evolve(c, wnum, a, b, phi, w_store, screen_update); /* :216 */
        step += screen_update;                                  /* :220 */
    }
    return nrec;
}

/* ======================================================================== *
 * input.rs:667-716
 * ======================================================================== */

/* ndarray's linspace(a, b, n): a + i*(b-a)/(n-1) */
static double linspace_at(double a, double b, int64_t n, int64_t i)
{
    double step = (n > 1) ? (b - a) / (double)(n - 1) : 0.;
    return a + step * (double)i;
}

/* (0..n).position(|xx| xx as f64 > look) -> (idx-1, idx), else (n-1, n) */
static void bracket(int64_t n, double look, int64_t *lo, int64_t *hi)
{
    for (int64_t q = 0; q < n; ++q)
        if ((double)q > look) {
            *lo = q - 1;
            *hi = q;
            return;
        }
    *lo = n - 1;
    *hi = n;
}

static inline double lerp1(double c0, double c1, double t) { return c0 * (1. - t) + c1 * t; }

/* `out` has dims (sx,sy,sz): output planes zbegin .. zbegin + sz of the z axis (zbegin = 0 and all of them in
 * wo_trilerp_resize_basis; a window of them for targets that do not fit the host); the sample positions are the first sx (sy, sz)
 * points of linspace(0, n, bx) (by, bz): input.rs:672-675 builds the basis from
 * the `size` argument while the loop runs over `output`'s own dims -- the
 * production call passes the PADDED target size with the unpadded work view
 * (input.rs:156-173, 640-656), the unit test passes the view's own dims. */
int wo_trilerp_resize_basis_zwindow(const double *v, int64_t vx, int64_t vy, int64_t vz, double *out,
                                    int64_t sx, int64_t sy, int64_t zbegin, int64_t sz, int64_t bx, int64_t by, int64_t bz)
{
    const int64_t nx = vx - 1, ny = vy - 1, nz = vz - 1;
    /* an axis of one point: (0..0).position(..) is None and the reference's (nx - 1, nx) underflows usize -- it panics
     * on the index that follows (input.rs:687-698).  Refused here (found by `make asan`: this used to read v[-1]). */
    if (nx < 1 || ny < 1 || nz < 1) return 1;
#define VAT(x, y, z) v[(((size_t)(x)) * (size_t)vy + (size_t)(y)) * (size_t)vz + (size_t)(z)]
#pragma omp parallel for schedule(static)
    for (int64_t x = 0; x < sx; ++x)
        for (int64_t y = 0; y < sy; ++y)
            for (int64_t z = 0; z < sz; ++z) {
                double xl = linspace_at(0., (double)nx, bx, x);
                double yl = linspace_at(0., (double)ny, by, y);
                double zl = linspace_at(0., (double)nz, bz, zbegin + z);
                int64_t x0, x1, y0, y1, z0, z1;
                bracket(nx, xl, &x0, &x1);
                bracket(ny, yl, &y0, &y1);
                bracket(nz, zl, &z0, &z1);
                double xd = (xl - (double)x0) / ((double)x1 - (double)x0);
                double yd = (yl - (double)y0) / ((double)y1 - (double)y0);
                double zd = (zl - (double)z0) / ((double)z1 - (double)z0);
                double c00 = lerp1(VAT(x0, y0, z0), VAT(x1, y0, z0), xd);
                double c01 = lerp1(VAT(x0, y0, z1), VAT(x1, y0, z1), xd);
                double c10 = lerp1(VAT(x0, y1, z0), VAT(x1, y1, z0), xd);
                double c11 = lerp1(VAT(x0, y1, z1), VAT(x1, y1, z1), xd);
                double c0 = lerp1(c00, c10, yd);
                double c1 = lerp1(c01, c11, yd);
                out[(((size_t)x) * (size_t)sy + (size_t)y) * (size_t)sz + (size_t)z] =
                    lerp1(c0, c1, zd);
            }
#undef VAT
    return 0;
}

int wo_trilerp_resize_basis(const double *v, int64_t vx, int64_t vy, int64_t vz, double *out,
                            int64_t sx, int64_t sy, int64_t sz, int64_t bx, int64_t by, int64_t bz)
{
    return wo_trilerp_resize_basis_zwindow(v, vx, vy, vz, out, sx, sy, 0, sz, bx, by, bz);
}

int wo_trilerp_resize(const double *v, int64_t vx, int64_t vy, int64_t vz, double *out,
                      int64_t sx, int64_t sy, int64_t sz)
{
    return wo_trilerp_resize_basis(v, vx, vy, vz, out, sx, sy, sz, sx, sy, sz);
}
