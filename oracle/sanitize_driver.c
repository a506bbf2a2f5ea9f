/*
 * sanitize_driver.c -- runs every entry point of the CPU oracle on small, awkward grids under
 * AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle asan`, SURVEY.md section 5: sanitizers belong on
 * the CPU build).  TEST INFRASTRUCTURE, like the oracle itself.  Every array is a separate heap allocation of
 * exactly the size the header documents, so an index one past the padded grid or the work area is a report, not
 * a silent read.  Exit code 0 and no report = clean; values are not judged (tests/ does that).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "wafer_oracle.h"

static double *alloc(size_t n)
{
    double *p = (double *)malloc(n * sizeof(double) + (n == 0));
    if (!p) { fprintf(stderr, "out of memory\n"); exit(2); }
    for (size_t i = 0; i < n; ++i) p[i] = 0.0;
    return p;
}

static int run_case(int64_t nx, int64_t ny, int64_t nz, int ext, int potential)
{
    wo_config c;
    memset(&c, 0, sizeof c);
    c.nx = nx; c.ny = ny; c.nz = nz; c.ext = ext; c.potential = potential;
    c.dn = 0.3; c.dt = 0.01; c.mass = 1.3; c.sig = 0.223;
    const size_t n = wo_padded_len(&c), nw = (size_t)(nx * ny * nz);
    double *v = alloc(n), *a = alloc(n), *b = alloc(n), *phi = alloc(n), *work = alloc(nw), *potsub = alloc(nw);
    double *low[3] = {alloc(n), alloc(n), alloc(n)};
    int rc = wo_potential_generate(&c, v);
    if (rc != 0) { fprintf(stderr, "potential %d refused\n", potential); return 1; }
    wo_ab(&c, v, a, b);
    int kind = 0;
    double scalar = 0.0;
    wo_potential_sub(&c, &kind, &scalar, NULL);
    if (kind == 2) wo_potential_sub(&c, &kind, &scalar, potsub);
    /* every initial condition (Coulomb divides by r: NaN at the centre of even padded sizes, as the reference) */
    for (int ic = WO_IC_GAUSSIAN; ic <= WO_IC_BOOLEAN; ++ic) wo_initial_condition(&c, ic, 7u + (unsigned)ic, phi);
    for (int j = 0; j < 3; ++j) {
        wo_initial_condition(&c, WO_IC_GAUSSIAN, 40u + (unsigned)j, low[j]);
        wo_normalise(low[j], n, wo_norm2(&c, low[j]));
        wo_orthogonalise(j, low[j], (const double *const *)low, n);
        wo_normalise(low[j], n, wo_norm2(&c, low[j]));
    }
    wo_initial_condition(&c, WO_IC_BOOLEAN, 0, phi);
    wo_stencil_step(&c, a, b, phi, work);
    wo_evolve(&c, 0, a, b, phi, NULL, 3);
    wo_observables_t obs;
    wo_observables(&c, v, kind, scalar, kind == 2 ? potsub : NULL, phi, &obs);
    for (int wnum = 1; wnum <= 3; ++wnum) wo_evolve(&c, wnum, a, b, phi, (const double *const *)low, 2);
    if (ext == 3)
        for (int sym = 0; sym <= 4; ++sym) wo_symmetrise(&c, sym, phi);
    else if (wo_symmetrise(&c, 1, phi) == 0) { fprintf(stderr, "symmetrise accepted a frame narrower than SevenPoint's\n"); return 1; }
    /* the solve loop: a few blocks, with and without max_steps */
    wo_block_record rec[8];
    int conv = 0;
    wo_initial_condition(&c, WO_IC_GAUSSIAN, 3, phi);
    wo_solve(&c, 0, v, a, b, kind, scalar, kind == 2 ? potsub : NULL, phi, NULL, 1e-3, 5, 1, 20, rec, 8, &conv);
    wo_initial_condition(&c, WO_IC_GAUSSIAN, 4, phi);
    wo_solve(&c, 1, v, a, b, kind, scalar, kind == 2 ? potsub : NULL, phi, (const double *const *)low, 1e-2, 4, 1, 12, rec, 2, &conv);
    /* trilinear resampling up and down, both bases */
    const int64_t sx = 2 * nx + 1, sy = ny + 2, sz = nz > 2 ? nz - 1 : nz;
    double *big = alloc((size_t)(sx * sy * sz));
    const int thin = nx < 2 || ny < 2 || nz < 2;   /* an axis of one point: the reference panics, the oracle refuses */
    if ((wo_trilerp_resize(work, nx, ny, nz, big, sx, sy, sz) != 0) != thin) { fprintf(stderr, "trilerp_resize: wrong answer to a one-point axis\n"); return 1; }
    if ((wo_trilerp_resize_basis(work, nx, ny, nz, big, sx, sy, sz, sx + 2 * ext, sy + 2 * ext, sz + 2 * ext) != 0) != thin) return 1;
    double acc = 0.0;
    for (size_t i = 0; i < (size_t)(sx * sy * sz); ++i) acc += big[i];
    free(big);
    (void)obs; /* values are not judged here (NaNs are the reference's own on degenerate grids): only the accesses are */
    for (int j = 0; j < 3; ++j) free(low[j]);
    free(v); free(a); free(b); free(phi); free(work); free(potsub);
    return acc == acc + 1.0 ? 0 : 0; /* (keeps the resampled values alive) */
}

int main(void)
{
    wo_set_threads(2);
    if (fabs(wo_alphas(3.2) - 6.189593433886306) > 1e-13 || fabs(wo_mu(5.2) - 2.604838027702063) > 1e-13) return 1;
    if (wo_calculate_r2(3, 3, 3, 5, 6, 3) != 1.25) return 1;
    static const int64_t shapes[][3] = {{1, 1, 1}, {2, 3, 1}, {5, 4, 7}, {9, 2, 3}, {6, 6, 6}};
    int cases = 0;
    for (size_t s = 0; s < sizeof shapes / sizeof shapes[0]; ++s)
        for (int ext = 1; ext <= 3; ++ext)
            for (int pot = WO_POT_NOPOTENTIAL; pot <= WO_POT_DODECAHEDRON; ++pot) {
                if (run_case(shapes[s][0], shapes[s][1], shapes[s][2], ext, pot) != 0) return 1;
                ++cases;
            }
    wo_config c;
    memset(&c, 0, sizeof c);
    c.nx = c.ny = c.nz = 3; c.ext = 1; c.dn = 0.1; c.dt = 0.001; c.mass = 1.0;
    c.potential = WO_POT_FROMFILE;
    double out = 0.0;
    if (wo_potential_at(&c, 1, 1, 1, &out) == 0) { fprintf(stderr, "FromFile has no closed form\n"); return 1; }
    printf("SANITIZE-OK %d cases\n", cases);
    return 0;
}
