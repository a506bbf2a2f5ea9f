/*
 * wafer_hip.h -- C ABI of the MI355X engine for Wafer's grid::evolve hot path.
 *
 * Wafer (Rust, /src/grid.rs) has no FFI or plugin interface: `grid` is a
 * private module and evolve/compute_observables/... are private fns.  The
 * boundary is therefore defined by their signatures.
 * Each entry point below names the reference item it replaces.  A Rust host
 * binds these with a plain `extern "C"` block (INTEGRATION.md shows it and
 * the four call sites in grid.rs that change).
 *
 * Conventions
 *  - Every function returns WAFER_OK (0) or a negative wafer_status;
 *    wafer_last_error() gives the message for the calling thread.  (The
 *    reference's fns are infallible and panic; its callers use error_chain
 *    Result<()>, errors.rs -- a host maps non-zero to an ErrorKind.)
 *  - One caller thread per context, non-reentrant (the reference calls these
 *    from its main thread only; parallelism is internal, main.rs:190-196).
 *  - Host arrays are ALWAYS double, C-order [x][y][z] (z contiguous), shape
 *    (nx+2e, ny+2e, nz+2e) with e = ext: exactly ndarray's Array3<R64>
 *    standard layout as the reference holds it (config.rs:224-238), so
 *    `arr.as_ptr()` can be passed straight through.  They describe the GLOBAL
 *    grid even when the context owns only a z-slab of it.
 *  - Device state (phi, V, a, b, pot_sub, w_store) stays resident in HBM
 *    between calls; only wafer_observables, norm2, download and last_evolve_ms block the host.
 */
#ifndef WAFER_HIP_H
#define WAFER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WAFER_ABI_VERSION 1

typedef enum wafer_status {
    WAFER_OK = 0,
    WAFER_ERR_INVALID = -1,      /* bad argument / config (cf. ErrorKind::LargeDt, LargeWavenum) */
    WAFER_ERR_HIP = -2,          /* a HIP runtime call failed */
    WAFER_ERR_NOT_AVAILABLE = -3,/* ErrorKind::PotentialNotAvailable (potential.rs:315-317) */
    WAFER_ERR_STATE = -4,        /* w_store too short / full, potential or phi not set */
    WAFER_ERR_MAX_STEP = -5,     /* ErrorKind::MaxStep (grid.rs:244, errors.rs:111-114) */
    WAFER_ERR_COMM = -6          /* a communication hook failed */
} wafer_status;

/* config.rs:73-104, same order */
typedef enum wafer_potential {
    WAFER_POT_NOPOTENTIAL = 0,
    WAFER_POT_CUBE,
    WAFER_POT_QUADWELL,
    WAFER_POT_PERIODIC,
    WAFER_POT_COULOMB,
    WAFER_POT_COMPLEXCOULOMB,
    WAFER_POT_ELIPTICALCOULOMB,
    WAFER_POT_SIMPLECORNELL,
    WAFER_POT_FULLCORNELL,
    WAFER_POT_HARMONIC,
    WAFER_POT_COMPLEXHARMONIC,
    WAFER_POT_DODECAHEDRON,
    WAFER_POT_FROMFILE,
    WAFER_POT_FROMSCRIPT
} wafer_potential;

/* config.rs:151-170, same order */
typedef enum wafer_initial_condition {
    WAFER_IC_FROMFILE = 0,
    WAFER_IC_GAUSSIAN,
    WAFER_IC_COULOMB,
    WAFER_IC_CONSTANT,
    WAFER_IC_BOOLEAN
} wafer_initial_condition;

/* config.rs:211-239: the value is CentralDifference::ext() */
typedef enum wafer_central_difference {
    WAFER_CD_THREEPOINT = 1,
    WAFER_CD_FIVEPOINT = 2,
    WAFER_CD_SEVENPOINT = 3
} wafer_central_difference;

/* WAFER_F64: the reference's arithmetic.  WAFER_F32: fp32 STORAGE of every array, arithmetic still
 * fp64 in registers (only storage rounding is added).  WAFER_F32_FAST: the ground-state stencil
 * steps also compute in fp32 (1.5x the fp64 step rate, on the three-step kernel; sums, projections and observables stay fp64) --
 * the throughput setting of BASELINE config #5, to be cross-checked against fp64 as that config
 * prescribes. */
typedef enum wafer_dtype { WAFER_F64 = 0, WAFER_F32 = 1, WAFER_F32_FAST = 2 } wafer_dtype;

/* pot_sub: (Option<Array3<R64>>, Option<R64>) of potential.rs:24 */
typedef enum wafer_potsub_kind { WAFER_POTSUB_NONE = 0, WAFER_POTSUB_SCALAR = 1, WAFER_POTSUB_ARRAY = 2 } wafer_potsub_kind;

typedef enum wafer_array_id { WAFER_ARRAY_V = 0, WAFER_ARRAY_A = 1, WAFER_ARRAY_B = 2, WAFER_ARRAY_POTSUB = 3 } wafer_array_id;

/* The subset of Config (config.rs:292-333) the hot path reads, plus the
 * engine's own knobs.  Zero-initialise, then set struct_size = sizeof. */
typedef struct wafer_params {
    uint32_t struct_size;
    uint32_t nx, ny, nz;          /* config.grid.size: GLOBAL work area */
    int32_t central_difference;   /* wafer_central_difference */
    int32_t dtype;                /* wafer_dtype: storage + arithmetic type on the device */
    double dn, dt;                /* config.grid.dn / dt */
    double mass;                  /* config.mass */
    double sig;                   /* config.sig */
    uint32_t max_states;          /* capacity of the device-resident w_store (>= wavemax) */
    int32_t device;               /* HIP device ordinal */
    /* z-slab owned by this context: work planes [z_begin, z_begin+z_count) of
     * the global grid; z_count == 0 means the whole grid.  halo_depth = ghost
     * planes kept on each z side (0 -> ext). */
    uint32_t z_begin, z_count;
    uint32_t halo_depth;
    uint32_t flags;               /* WAFER_FLAG_* */
} wafer_params;

#define WAFER_FLAG_SKIP_DT_CHECK 1u /* do not enforce dt <= dn^2/3 (config.rs:362-365) */
#define WAFER_FLAG_UNPLANNED_DIV 2u /* ignore the verdict of the division plan (wafer_div_plan): every x / (c dn^2 m) of the step
                                     * kernels takes the extra Markstein round, as for a denominator the plan could not clear */

/* grid.rs:17-28, un-normalised, in this order */
typedef struct wafer_observables_t {
    double energy, norm2, v_infinity, r2;
} wafer_observables_t;

/* One row of the convergence table solve() prints (grid.rs:126-221,
 * output.rs:497-521): raw observables at `step`, diff = |E - E_last|. */
typedef struct wafer_block_record {
    uint64_t step;
    double tau;
    wafer_observables_t obs;
    double diff;
} wafer_block_record;

/* output.rs:32-45, 540-547 */
typedef struct wafer_observables_output {
    uint32_t state;
    double energy, binding_energy, r, l_r;
} wafer_observables_output;

typedef struct wafer_ctx wafer_ctx;

/* ---- diagnostics ------------------------------------------------------- */
int wafer_abi_version(void);
const char *wafer_last_error(void);

/* ---- lifetime: replaces the allocations of grid.rs:32-34, 560; potential.rs:101-102 */
int wafer_ctx_create(const wafer_params *params, wafer_ctx **out);
int wafer_ctx_destroy(wafer_ctx *ctx);
int wafer_synchronize(wafer_ctx *ctx);

/* ---- Potentials {v, a, b, pot_sub} (potential.rs:16-25) ----------------- */
/* potential::generate + a/b + pot_sub, on the device (potential.rs:46-62, 101-110, 134-153) */
int wafer_set_potential_builtin(wafer_ctx *ctx, int potential);
/* FromFile / FromScript: v is the padded global array; potsub is NULL, or the
 * UNPADDED nx*ny*nz array when potsub_kind == WAFER_POTSUB_ARRAY */
int wafer_set_potential_host(wafer_ctx *ctx, const double *v, int potsub_kind, double potsub_scalar, const double *potsub);
/* tests / output::potential: padded global array out (POTSUB: unpadded) */
int wafer_download_array(wafer_ctx *ctx, int array_id, double *out);
int wafer_get_potsub(wafer_ctx *ctx, int *kind, double *scalar);
/* replaces pot_sub after the potential has been set: a potential_sub file in ./input takes
 * precedence over the computed value for every potential type (potential.rs:113-131).  potsub is
 * the UNPADDED nx*ny*nz array when kind == WAFER_POTSUB_ARRAY, else ignored. */
int wafer_set_potsub(wafer_ctx *ctx, int kind, double scalar, const double *potsub);
/* the same override from a potential_sub array of another resolution: input::fill_sub_data
 * (input.rs:453-478) resamples it with trilerp_resize onto the work area, basis = (nx, ny, nz).
 * src is an UNPADDED [sx][sy][sz] array. */
int wafer_set_potsub_resampled(wafer_ctx *ctx, const double *src, uint32_t sx, uint32_t sy, uint32_t sz);

/* ---- phi: config::set_initial_conditions (config.rs:577-627), input::wavefunction, output::wavefunction */
int wafer_set_initial_condition(wafer_ctx *ctx, int ic, uint64_t seed);
int wafer_upload_phi(wafer_ctx *ctx, const double *phi);
int wafer_download_phi(wafer_ctx *ctx, double *phi);
/* the work cells of the planes this context OWNS, [nx][ny][z_count] in the reference's axis order
 * (z_count = nz without a slab): what a rank of a decomposed run saves -- its host buffer is the size
 * of its slab, not of the grid */
int wafer_download_phi_owned(wafer_ctx *ctx, double *out);
/* Restart from another resolution: input::fill_data / read_csv's resampling branch with
 * trilerp_resize (input.rs:149-176, 640-656, 667-716).  src is an UNPADDED [sx][sy][sz] array;
 * basis = the `size` the reference builds its linspace from (NULL = the padded target size, which
 * is what the reference's production call passes; its unit test passes the work-area dims). */
int wafer_upload_phi_resampled(wafer_ctx *ctx, const double *src, uint32_t sx, uint32_t sy, uint32_t sz,
                               const uint32_t *basis);
int wafer_set_potential_resampled(wafer_ctx *ctx, const double *src, uint32_t sx, uint32_t sy, uint32_t sz,
                                  const uint32_t *basis);

/* config::symmetrise_wavefunction (config.rs:691-728), applied by the reference to the initial
 * condition (config.rs:625) and to snapshots (grid.rs:138).  Constraint in the order of
 * SymmetryConstraint (config.rs:184-197).  The reference indexes the SevenPoint frame literally
 * (offset 3, extent n + 6) and would run out of bounds on a narrower one: any constraint other
 * than NOT_CONSTRAINED returns WAFER_ERR_INVALID unless central_difference is SevenPoint.  Its
 * quirks are kept: the mirror plane sits half a cell below the centre of the work area and the
 * last work cell along the axis takes the frame's zero.  About z is not available on z-slabs. */
typedef enum wafer_symmetry {
    WAFER_SYM_NOT_CONSTRAINED = 0,
    WAFER_SYM_ABOUT_Z,
    WAFER_SYM_ANTISYM_ABOUT_Z,
    WAFER_SYM_ABOUT_Y,
    WAFER_SYM_ANTISYM_ABOUT_Y
} wafer_symmetry;
int wafer_symmetrise(wafer_ctx *ctx, int constraint);

/* ---- the hot path -------------------------------------------------------- */
/* evolve (grid.rs:544-687): n_steps = config.output.screen_update; like the
 * reference, n_steps == 0 still takes one step.  wnum > 0 renormalises and
 * Gram-Schmidts against w_store[0..wnum) after every step. */
int wafer_evolve(wafer_ctx *ctx, uint32_t wnum, uint64_t n_steps);
/* compute_observables (grid.rs:303-445) */
int wafer_observables(wafer_ctx *ctx, wafer_observables_t *out);
/* get_norm_squared over the work area (grid.rs:454-457) */
int wafer_norm2(wafer_ctx *ctx, double *out);
/* normalise_wavefunction (grid.rs:465-468) */
int wafer_normalise(wafer_ctx *ctx, double norm2);
/* orthogonalise_wavefunction (grid.rs:477-492) */
int wafer_orthogonalise(wafer_ctx *ctx, uint32_t wnum);

/* ---- w_store: Vec<Array3<R64>> of grid.rs:34 ----------------------------- */
int wafer_push_state(wafer_ctx *ctx);                               /* w_store.push(phi), grid.rs:241 */
int wafer_load_state(wafer_ctx *ctx, uint32_t idx, const double *state); /* input::load_wavefunctions, grid.rs:38 */
int wafer_download_state(wafer_ctx *ctx, uint32_t idx, double *state);
int wafer_clone_state_to_phi(wafer_ctx *ctx, uint32_t idx);         /* w_store[wnum-1].clone(), grid.rs:95 */
int wafer_num_states(wafer_ctx *ctx, uint32_t *out);
int wafer_clear_states(wafer_ctx *ctx);

/* ---- solve (grid.rs:50-246) for ONE state, snapshot branch excluded ------ */
/* phi must hold the starting wavefunction.  Writes up to max_records rows,
 * *n_records = rows produced.  Returns WAFER_OK when converged (phi has then
 * been pushed to w_store, grid.rs:239-242) and WAFER_ERR_MAX_STEP otherwise. */
int wafer_solve_state(wafer_ctx *ctx, uint32_t wnum, double tolerance, uint64_t screen_update,
                      int has_max_steps, uint64_t max_steps, wafer_block_record *records,
                      size_t max_records, size_t *n_records, wafer_observables_output *final_out);

/* ---- measurement ---------------------------------------------------------- */
/* HIP-event time of the kernels of the last wafer_evolve call, on the stream
 * they ran on, and the number of steps it took.  Blocks until they finish. */
int wafer_last_evolve_ms(wafer_ctx *ctx, float *ms, uint64_t *steps);
/* name of the stencil kernel a ground-state pass of this context launches (no template arguments): wafer_k_step3_fused,
 * wafer_k_step2_wide (FivePoint), wafer_k_step2_fused, wafer_k_step_lds, wafer_k_step_direct */
const char *wafer_stencil_kernel_name(wafer_ctx *ctx);
/* the template-id of the kernel the last ground-state pass launched, as a profiler prints it (e.g.
 * "wafer_k_step3_fused<double, double, true, 0, true, 1>"); the family name where the family does not record it.  Valid until
 * the next call on this context. */
const char *wafer_stencil_kernel_instance(wafer_ctx *ctx);
/* time steps one launch of that kernel advances in a ground-state evolve (2 for the fused kernel) */
int wafer_stencil_steps_per_launch(wafer_ctx *ctx);
/* choose a stencil kernel variant by index (tuning / A-B runs); -1 = default */
int wafer_set_stencil_variant(wafer_ctx *ctx, int variant);

/* Diagnostic: the device's copy ceiling -- 16 B per lane, `unroll` (1, 2, 4, 8) vectors in flight per
 * lane, grid-stride over blocks_per_cu x CUs workgroups of 256 threads, V -> phi's scratch buffer;
 * GB/s of read + written bytes.  MI355X_MICROARCH.md quotes ~6.3 TB/s for this pattern. */
int wafer_diag_copy_bw(wafer_ctx *ctx, int iters, int unroll, int blocks_per_cu, double *gbps);

/* Diagnostic: integer checksum (sum mod 2^64 of a hash of each cell's bits and its GLOBAL index) of the
 * work cells of global work planes [z_begin, z_begin + z_count) that this context owns.  Order
 * independent, so the checksums of the slabs of a decomposed run must equal those of the same plane
 * ranges of an undecomposed run whenever the bits agree: bench.py's N > 1 parity check. */
int wafer_diag_checksum(wafer_ctx *ctx, uint32_t z_begin, uint32_t z_count, uint64_t *out);

/* Diagnostic: global padded planes [zp_begin, zp_begin + zp_count) of one device array in the reference's layout,
 * out[px][py][zp_count] (z contiguous) -- what wafer_download_array / wafer_download_phi return, restricted to a z-window, for
 * grids whose arrays do not fit the host (1024^3, 2048^3: tests/test_gpu_fullsize.py compares such windows cell by cell with the
 * oracle).  array_id: WAFER_ARRAY_V / _A / _B, or WAFER_ARRAY_PHI (the current wavefunction).  The planes must be among those the
 * context holds (its owned planes + halo_depth ghost planes per side, and the global frame): WAFER_ERR_INVALID otherwise. */
#define WAFER_ARRAY_PHI 4
int wafer_diag_download_window(wafer_ctx *ctx, int array_id, uint32_t zp_begin, uint32_t zp_count, double *out);

/* Diagnostic: which kernel a pass of wafer_evolve(ctx, wnum, .) launches for this context, as one line of key=value pairs ("wnum=0 stencil=1
 * dtype=f64 kernel=wafer_k_step3_fused steps_per_pass=3 ghost_planes_per_pass=3 tile=128x16 v=streamed remainder=wafer_k_step2_fused,wafer_k_step_lds"):
 * the launch path's own predicates, nothing launched.  tools/dispatch_table.py tabulates it (profiles/r06_dispatch_table.md). */
int wafer_diag_dispatch(wafer_ctx *ctx, uint32_t wnum, char *buf, size_t n);

/* Diagnostic: how many passes of the two-excited-steps-per-pass kernel (wafer_stencil_x2.hip.h: ThreePoint fp64, one to three
 * stored states) this context has launched so far -- tests assert that the kernel they mean to test is the one that ran. */
int wafer_diag_x2_passes(wafer_ctx *ctx, uint64_t *out);

/* Diagnostic: the divisions by loop-invariant denominators the host has not planned (the norm in the excited-state
 * transform, projection coefficients) use a hoisted reciprocal with two exact remainders instead of the IEEE
 * sequence (wafer_div_invariant(x, den), wafer_stencil.hip.h).  Draws n_operands (rounded up to 2^18) doubles
 * with uniform sign / significand and biased exponent uniform in [lo_exp, hi_exp] and counts those
 * whose quotient by `den` differs in any bit from the device's IEEE x / den. */
int wafer_diag_div_check(wafer_ctx *ctx, double den, uint64_t seed, uint64_t n_operands, int lo_exp, int hi_exp,
                         uint64_t *mismatches);

/* The division by the run's ONE stencil denominator c dn^2 m (grid.rs:569 / 594 / 626) is planned when the context is created
 * (wafer_amd/csrc/wafer_divplan.h): q = RN(x zh + RN(x zl)) with zh = RN(1/den), zl = RN(1/den - zh) (moved by zl_shift ulps if
 * that gets more operands through) is RN(x/den) except for the few dozen significands x whose quotient lies within 2^-50 ulp of
 * the midpoint of two doubles; the plan enumerates those (n_candidates) and tries each with the device's instruction sequence.
 * checked = 1: all came out as the IEEE quotient, so the three-instruction form is the division for every x (|x/den| >= 2^-960);
 * checked = 0: the kernels add one Markstein round (always correct, two instructions more). */
typedef struct wafer_div_plan_t {
    double den, zh, zl;
    int32_t checked, n_candidates, zl_shift, reserved;
} wafer_div_plan_t;
/* Host only (no GPU needed): the plan for `den`; candidates (may be NULL): up to `cap` of the enumerated significands, as doubles
 * in [2^52, 2^53), *n_written of them. */
int wafer_div_plan(double den, wafer_div_plan_t *out, double *candidates, size_t cap, size_t *n_written);
/* the plan this context's kernels run with */
int wafer_get_div_plan(wafer_ctx *ctx, wafer_div_plan_t *out);
/* Diagnostic: the planned division as the kernels perform it (wafer_div_invariant(x, WaferDen)) with the given plan -- any
 * plan, also one whose `checked` the caller has forced -- on n_random drawn operands (as wafer_diag_div_check) and on the
 * n_operands doubles at `operands` (host memory; may be NULL): the number of quotients that differ from the device's IEEE
 * x / den in any bit, separately. */
int wafer_diag_div_planned(wafer_ctx *ctx, const wafer_div_plan_t *plan, uint64_t seed, uint64_t n_random, int lo_exp, int hi_exp,
                           const double *operands, size_t n_operands, uint64_t *mismatches_random, uint64_t *mismatches_operands);

/* The same plan in fp32, for WAFER_F32_FAST contexts (whose step kernels compute in fp32): q = RN(x zh + RN(x zl)) in float, checked by
 * trying all 2^23 significands on the host (tens of milliseconds; |x/den| >= 2^-100, |den| in [2^-60, 2^60]).  Unchecked: the kernels divide. */
typedef struct wafer_div_plan_f32_t {
    float den, zh, zl;
    int32_t checked, zl_shift;
} wafer_div_plan_f32_t;
int wafer_div_plan_f32(float den, wafer_div_plan_f32_t *out);   /* host only */
/* Diagnostic: the planned fp32 division on EVERY float with a biased exponent in [lo_exp, hi_exp] (all significands, both signs)
 * against the device's IEEE x / den: the number of quotients that differ in any bit. */
int wafer_diag_div_planned_f32(wafer_ctx *ctx, const wafer_div_plan_f32_t *plan, int lo_exp, int hi_exp, uint64_t *mismatches);

/* ---- multi-GPU: communication hooks --------------------------------------- */
/* The engine never links a communication library.  A host that z-slabs the
 * grid over several contexts installs two hooks (RCCL via torch.distributed in
 * wafer_amd/slab.py; ncclSend/ncclRecv + ncclAllReduce in a Rust host):
 *  - halo: copy `bytes` from this rank's send_lo / send_hi (its first / last
 *    `planes` owned planes) into the z-neighbours' ghost planes and fill
 *    recv_lo / recv_hi from theirs.  A NULL pointer means "no neighbour on
 *    that side" (global Dirichlet frame).
 *  - allreduce: in-place sum over ranks of `count` doubles at dev_ptr.
 * Both are called with the hipStream_t the data is ordered on; on return the
 * result must be ordered on that same stream (enqueue, do not block). */
typedef int (*wafer_halo_fn)(void *user, void *send_lo, void *send_hi, void *recv_lo, void *recv_hi,
                             size_t bytes, void *hip_stream);
typedef int (*wafer_allreduce_fn)(void *user, void *dev_ptr, size_t count, void *hip_stream);
int wafer_set_comm_hooks(wafer_ctx *ctx, wafer_halo_fn halo, wafer_allreduce_fn allreduce, void *user);
/* The halo hook may be called with only one direction non-NULL (send_lo + recv_hi, or send_hi + recv_lo: the
 * single-launch pass of wafer_set_overlap mode 2 exchanges the two sides at different times): a hook must
 * tolerate a NULL on a side that has a neighbour, and pair send_lo with the lower neighbour's recv_hi. */
/* z-slabs, how the halo exchange is scheduled (modes 0 .. 6; 3 .. 6 below).  All modes give identical results.
 *  0 = the exchange follows the whole slab's update (nothing overlaps);
 *  1 = boundary planes first, then their exchange, both on a second stream, beside the interior update:
 *      three launches per pass;
 *  2 (default) = ONE launch per three-step pass updates the whole slab as two halves marched outwards from the middle plane;
 *      workgroups count themselves done and the second stream releases each half's exchange as soon as that half
 *      is complete; the ghost planes an exchange fills are announced by a device flag that only the workgroups
 *      reading them poll, shortly before the end of their column.  Ground-state three-step passes with one exchange
 *      per pass; other ground-state passes run as in mode 1, excited-state steps (one plane per side and step) as in
 *      mode 0, which measured faster for them.  The order of the halves alternates from pass to pass,
 *      so every rank must make the same sequence of wafer_evolve calls (as it must anyway).  A workgroup waits at most
 *      WAFER_HV_WAIT_MS (default 20 s) for its ghost planes; one that gives up stores NaN from there on and the next
 *      call that synchronises with the device returns WAFER_ERR_COMM.
 * Setting a mode also resets the pass bookkeeping (which half goes first, which ghost planes are current): after a
 * WAFER_ERR_COMM on any rank, call it on every rank before evolving again. */
int wafer_set_overlap(wafer_ctx *ctx, int mode);
/*  3 = mode 2's single launch with PEER STORES instead of an exchange: the boundary workgroups of a pass store their last
 *      planes into the z-neighbours' ghost planes themselves (the neighbour's buffers mapped here: directly within one process,
 *      through HIP IPC between processes of one node, over xGMI between GPUs) and count themselves into the neighbour's arrival
 *      counter; the neighbour's boundary workgroups poll that counter just before their first ghost-plane load.  No exchange
 *      kernels (RCCL's need whole CUs), no gate kernels, no second stream, no short columns.  Needs wafer_peer_connect on every
 *      rank first and at least 6 owned planes per rank; the host switches all ranks or none.  The first pass of a run of passes
 *      still takes its ghost planes from the halo hook (it is the run's rendezvous).  Other passes as in mode 2. */
/*  4 = PEER COPIES (round 6): every exchange of phi's ghost planes -- the passes' of modes 2 / 1 / 0, the excited-state steps', the ones
 *      observables and wafer_push_state ask for -- is a device copy (hipMemcpyAsync: a copy engine between GPUs, no CU of either) from
 *      this rank's boundary planes INTO the z-neighbour's ghost planes through the mapping wafer_peer_connect holds; the halo hook is
 *      not called for phi at all (the stored states' own ghost planes, exchanged once per change of w_store, keep it; so does the
 *      all-reduce).  The hook's two-sided contract is kept by a rendezvous of four words per link in the neighbours' memory
 *      (one-wave kernels in stream order: "receive k posted" -> wait -> copy -> "copy k landed" -> wait), so a rank's ghost planes are
 *      overwritten only once it would have posted the receive.  Ground-state three-step passes run as mode 2's single launch, without
 *      its short columns (no exchange kernel needs a CU).
 *  5, 6 = peer copies under mode 1's (boundary planes first, three launches per pass) / mode 0's (exchange after the pass) launches:
 *      every kernel that reads ghost planes is launched after the copy that filled them has completed, so nothing rests on what a
 *      RUNNING kernel sees of a peer's writes -- the one assumption mode 4 shares with mode 3 on the consumer side.  (WAFER_COPY_SCHED =
 *      1 / 0 turns mode 4 into these.)  Needs wafer_peer_connect on every rank first; any stencil, storage type and slab
 *      thickness.  After a WAFER_ERR_COMM under this mode the two ends of a link may disagree on their counts: export and connect
 *      again (a fresh export restarts them).  As for mode 3, contexts of ONE process on ONE device must not sit in a device-wide
 *      synchronisation (hipFree, hipDeviceSynchronize) while a neighbour context waits for their signal -- one process per GPU cannot.
 * Peer stores: what a rank publishes about itself, and the connection to its z-neighbours.  wafer_peer_export fills `out`
 * (device addresses valid in this process + HIP IPC handles of the allocations for other processes); the host carries the
 * records to the neighbours (any transport) and calls wafer_peer_connect with the lower / upper neighbour's record (NULL: no
 * neighbour on that side; a record exported by this same process is used by address, without IPC -- several contexts in one
 * process, or a slab whose neighbour is itself; a context of this process on ANOTHER device is used by address after
 * hipDeviceCanAccessPeer / hipDeviceEnablePeerAccess, WAFER_ERR_INVALID where the devices cannot reach each other).  A neighbour
 * that is another context on the SAME device is refused (WAFER_ERR_INVALID) unless WAFER_PEER_SAME_DEVICE=1: its kernels share
 * this one's CUs, and a workgroup polling for a neighbour's stores can then keep that neighbour's kernel from running until the
 * bounded wait gives up -- tests fold ranks onto one GPU on purpose; a real run has one GPU per rank.
 * wafer_peer_disconnect unmaps; wafer_ctx_destroy does it too. */
typedef struct wafer_peer_info {
    uint32_t struct_size;
    uint32_t z_begin, z_count, halo_depth;
    uint64_t pid;                 /* of the exporting process */
    uint64_t phi_addr[2];         /* the two ping-pong buffers (plane 0, row 0) */
    uint64_t flags_addr;          /* arrival counters: [0] lower ghost side, [8] upper (64-bit words, one 64-byte line each) */
    uint64_t phi_alloc_offset[2]; /* byte offset of phi_addr inside its allocation (IPC maps allocations) */
    uint8_t phi_ipc[2][64];       /* hipIpcMemHandle_t */
    uint8_t flags_ipc[64];
    uint64_t process_nonce;       /* drawn once per process: with pid it identifies the exporting process (pids alone repeat across
                                     PID namespaces) -- the by-address shortcut is taken only when both match */
    int32_t device;               /* HIP ordinal of the exporting context's device, as the exporting process numbers them */
    uint32_t reserved;
    uint8_t device_uuid[16];      /* hipDeviceGetUuid of that device: the same GPU whatever the process calls it */
} wafer_peer_info;
int wafer_peer_export(wafer_ctx *ctx, wafer_peer_info *out);
int wafer_peer_connect(wafer_ctx *ctx, const wafer_peer_info *lower, const wafer_peer_info *upper);
int wafer_peer_disconnect(wafer_ctx *ctx);

/* z-slabs, ground state: fused passes per halo exchange.  One fused pass advances K time steps and consumes
 * K * ext ghost planes per side (K = 3 where the three-step kernel applies: ThreePoint, any dtype,
 * with halo_depth >= 3 * ext; else K = 2 -- every rank of a run must be created alike).  With `passes` > 1 the exchange moves K * ext * passes planes at once and
 * the passes in between run unsplit over the owned planes plus the ghost planes that are still valid -- fewer
 * boundary launches, exchanges and stream hops for a few redundant planes.  Needs wafer_params.halo_depth >=
 * K * ext * passes (WAFER_ERR_INVALID otherwise); the default is 1.  All settings give identical results. */
int wafer_set_halo_cycle(wafer_ctx *ctx, int passes);
/* run every kernel on a caller-owned hipStream_t (NULL = the context's own) */
int wafer_set_stream(wafer_ctx *ctx, void *hip_stream);
/* geometry of the local slab, for hosts that need it */
typedef struct wafer_slab_info {
    uint32_t z_begin, z_count, halo_depth, ext;
    uint64_t plane_elems;   /* elements per z-plane of the device layout */
    uint64_t elem_bytes;
} wafer_slab_info;
int wafer_get_slab_info(wafer_ctx *ctx, wafer_slab_info *out);
/* the device the context runs on, as its own properties describe it (printed by the benchmark
 * next to the datasheet HBM peak; the reference has no counterpart) */
typedef struct wafer_device_info {
    char name[64];
    char arch[64];
    uint32_t compute_units;
    uint32_t memory_clock_khz;
    uint32_t memory_bus_bits;
    uint32_t l2_bytes;
    uint64_t total_bytes;
} wafer_device_info;
int wafer_get_device_info(wafer_ctx *ctx, wafer_device_info *out);

#ifdef __cplusplus
}
#endif
#endif
