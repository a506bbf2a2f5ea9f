// wafer_engine_schedules.hip -- which kernel advances what: the stencil variants and their launch geometry, the workgroup
// tables of the three-step kernel, reductions, Gram-Schmidt chains, the excited-state passes, wafer_evolve (grid.rs:544-687),
// compute_observables and the norm / normalise / orthogonalise entry points.
#include "wafer_engine.h"
#include "wafer_elementwise.hip.h"
#include "wafer_stencil_lds.hip.h"
#include "wafer_stencil_fused2.hip.h"

namespace wafer_eng __attribute__((visibility("hidden"))) {
// planes per workgroup so that a launch over `nplanes` has >= target blocks
int pick_zchunk(const wafer_ctx *c, int nplanes, int target_blocks)
{
    // (WAFER_ZCHUNK, but never more than the 64 chunks per column the partial-sum rows are sized for)
    if (c->tune.zchunk > 0) return std::max(c->tune.zchunk, (nplanes + 63) / 64);
    const long long per_layer = (long long)c->bx * c->by;
    long long nch = (target_blocks + per_layer - 1) / per_layer;
    if (nch < 1) nch = 1;
    if (nch > nplanes) nch = nplanes;
    if (nch > 64) nch = 64;
    return (int)((nplanes + nch - 1) / nch);
}

// second-stage reduce of `nq` quantities of `n` partials each into scal[slot..slot+nq)
int reduce_to_scal(wafer_ctx *c, int nq, long long n, int slot, hipStream_t s)
{
    hipLaunchKernelGGL(wafer_k_reduce, dim3(nq), dim3(256), 0, s, c->partials, n,
                       (long long)c->partials_stride, c->scal + slot);
    HIP_TRY(hipGetLastError());
    if (c->allreduce_hook && c->sharded()) {
        if (c->allreduce_hook(c->hook_user, c->scal + slot, (size_t)nq, (void *)s) != 0)
            return fail(WAFER_ERR_COMM, "allreduce hook failed");
    }
    return WAFER_OK;
}

int read_scal(wafer_ctx *c, int slot, int n, double *out, hipStream_t s)
{
    HIP_TRY(hipMemcpyAsync(c->scal_host + slot, c->scal + slot, sizeof(double) * n,
                           hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    TRY(check_hv_err(c));
    for (int q = 0; q < n; ++q) out[q] = c->scal_host[slot + q];
    return WAFER_OK;
}

} // namespace wafer_eng

namespace wafer_eng __attribute__((visibility("hidden"))) {
// ---------------------------------------------------------------------------
// stencil step dispatch
// ---------------------------------------------------------------------------
struct VariantInfo {
    const char *name;
};
static const VariantInfo kVariants[] = {
    {"wafer_k_step_direct"},
    {"wafer_k_step_lds"},
    {"wafer_k_step2_fused"},
    {"wafer_k_step3_fused"},
};
static const int kNumVariants = (int)(sizeof(kVariants) / sizeof(kVariants[0]));
const char *variant_name(int v) { return kVariants[(v >= 0 && v < kNumVariants) ? v : 0].name; }

int default_variant(const wafer_ctx *c)
{
    // (FivePoint on fp32 storage with fp64 arithmetic stayed on the single-step kernel while the two-step kernel on 128 x 16 tiles
    //  was a tie with it, 0.337 against 0.335 ms/step at 512^3; with the planned division, streamed stores and its requests placed
    //  early the two-step kernel takes 0.301 against 0.334)
    // SevenPoint: the two-step kernel exists (variant 2, bit-exact, 128 x 8 tiles) but recomputes phi1 on 14 rows
    // per 8 and is issue-bound: 0.93 ms/step at 512^3 against 0.63 for the single-step kernel on 128 x 16 tiles
    if (c->g.R == 3) return 1;
    // ThreePoint, every type combination: three steps per pass (wafer_stencil_fused3.hip.h); everything else two
    if (c->g.R == 1 && c->tune.fuse3 != 0) return 3;
    return 2;
}

int active_variant(const wafer_ctx *c) { return c->variant >= 0 ? c->variant : default_variant(c); }

// The closed form a kernel may evaluate instead of streaming V (0: none): fp64 contexts whose potential was
// generated from Coulomb / SimpleCornell / Harmonic and whose radii dn .. dn * sqrt(3) (n + 1) / 2 lie inside
// the range of the short reciprocal (wafer_vgen_at); WAFER_VGEN=0 keeps every kernel on the stored array.
int closed_form_vg(const wafer_ctx *c)
{
    const bool r_ok = c->P.dn > 0x1p-300 && c->P.dn * ((double)c->g.nx + c->g.ny + c->g.nz + 3.) < 0x1p300;
    return (!c->f32 && r_ok && c->tune.vgen != 0) ? c->vgen_type : 0;
}
void set_vg_args(const wafer_ctx *c, WaferStepArgs &a)
{
    a.vg_dn = c->P.dn;
    a.vg_mass = c->P.mass;
    a.vg_sig = c->P.sig;
}

// storage / arithmetic types of a launch (wafer_launch.h): WAFER_F32_FAST computes the ground-state stencil steps in
// fp32 as well (sums, projections and observables stay fp64); plain fp32 storage widens to fp64 in registers
int type_combo(const wafer_ctx *c, bool step_kernel)
{
    if (!c->f32) return WAFER_TC_F64;
    return (c->f32_arith && step_kernel) ? WAFER_TC_F32_F32 : WAFER_TC_F32_F64;
}

WaferStepArgs step_args(const wafer_ctx *c, int lz_lo, int lz_hi)
{
    WaferStepArgs a{};
    a.g = c->g;
    a.lz_lo = lz_lo;
    a.lz_hi = lz_hi;
    a.dt = c->P.dt;
    a.target_blocks = c->num_cus;
    a.v_in_range = short_forms(c) ? 1 : 0;
    set_den_args(c, a);
    set_vg_args(c, a);
    return a;
}

template <typename F>
static int dispatch(wafer_ctx *c, F &&f, bool step_kernel = false)
{
    // f(T storage tag, C compute tag, R tag)
    const int R = c->g.R;
    if (c->f32 && c->f32_arith && step_kernel) {
        // WAFER_F32_FAST: the ground-state stencil steps also COMPUTE in fp32 (sums, projections
        // and observables stay fp64)
        if (R == 1) return f(float{}, float{}, std::integral_constant<int, 1>{});
        if (R == 2) return f(float{}, float{}, std::integral_constant<int, 2>{});
        return f(float{}, float{}, std::integral_constant<int, 3>{});
    }
    if (!c->f32) {
        if (R == 1) return f(double{}, double{}, std::integral_constant<int, 1>{});
        if (R == 2) return f(double{}, double{}, std::integral_constant<int, 2>{});
        return f(double{}, double{}, std::integral_constant<int, 3>{});
    }
    // fp32 storage; arithmetic widened to fp64 in registers (the path is HBM-bound)
    if (R == 1) return f(float{}, double{}, std::integral_constant<int, 1>{});
    if (R == 2) return f(float{}, double{}, std::integral_constant<int, 2>{});
    return f(float{}, double{}, std::integral_constant<int, 3>{});
}

int direct_target_blocks(const wafer_ctx *c) { return c->tune.target_blocks > 0 ? c->tune.target_blocks : 4096; }

// one step over local planes [lz_lo, lz_hi); norm: also sum phi'^2 into the partials (more stored states than the
// fused-overlap kernel carries)
int launch_step(wafer_ctx *c, int src, int dst, int lz_lo, int lz_hi, bool norm, hipStream_t s)
{
    if (lz_hi <= lz_lo) return WAFER_OK;
    const int variant = active_variant(c);
    if (kernels_stream_ab(c, variant)) TRY(ensure_ab(c));
    WaferStepArgs a = step_args(c, lz_lo, lz_hi);
    if (variant >= 1) {
        const int tc = type_combo(c, !norm);
        const hipError_t e =
            norm ? wafer_entry_step_lds_excited(tc, c->g.R, c->tune, a, c->phi[src], c->v, c->phi[dst], c->partials, c->partials_stride, 0,
                                                WaferLowPtrs(), s, nullptr, nullptr, 0)
                 : wafer_entry_step_lds(tc, c->g.R, c->tune, a, c->phi[src], c->a, c->b, c->v, c->phi[dst], s, closed_form_vg(c));
        return e == hipSuccess ? WAFER_OK
                               : fail(WAFER_ERR_HIP, "LDS stencil launch failed: %s", hipGetErrorString(hipGetLastError()));
    }
    a.zchunk = pick_zchunk(c, lz_hi - lz_lo, direct_target_blocks(c));
    const dim3 grid(c->bx, c->by, nchunks_of(lz_hi - lz_lo, a.zchunk));
    if ((size_t)grid.x * grid.y * grid.z > c->partials_stride)
        return fail(WAFER_ERR_INVALID, "partials buffer too small");
    return dispatch(c, [&](auto t, auto cc, auto r) {
        using T = decltype(t);
        using C = decltype(cc);
        constexpr int R = decltype(r)::value;
        if (norm)
            hipLaunchKernelGGL((wafer_k_step_direct<T, C, R, true>), grid, dim3(64, 4), 0, s, a, as<T>(c->phi[src]), as<T>(c->a), as<T>(c->b),
                               as<T>(c->phi[dst]), c->partials);
        else
            hipLaunchKernelGGL((wafer_k_step_direct<T, C, R, false>), grid, dim3(64, 4), 0, s, a, as<T>(c->phi[src]), as<T>(c->a), as<T>(c->b),
                               as<T>(c->phi[dst]), c->partials);
        HIP_TRY(hipGetLastError());
        return (int)WAFER_OK;
    }, !norm);
}

// number of partials the norm variant of the last step launch wrote
long long step_partials_count(wafer_ctx *c, int lz_lo, int lz_hi)
{
    if (active_variant(c) >= 1)
        return dispatch(c, [&](auto t, auto, auto r) {
            return (int)wafer_step_lds_excited_blocks<decltype(t), decltype(r)::value>(c->tune, c->g, lz_lo, lz_hi, c->num_cus);
        });
    const int zc = pick_zchunk(c, lz_hi - lz_lo, direct_target_blocks(c));
    return (long long)c->bx * c->by * nchunks_of(lz_hi - lz_lo, zc);
}

// the three-step kernel serves ThreePoint grids (fp64; fp32 storage with either arithmetic) whose rows fill its tiles -- undecomposed, or
// z-slabs created with at least 3 * ext ghost planes; everything else takes the two-step kernel.
// Every rank of a decomposed run must take the same decision (the ranks exchange K * ext planes per K-step pass):
// for a slab it therefore depends only on what all ranks share -- the global nx, ny, the ghost depth the host created
// every context with and the variant -- never on the local slab thickness (slab.partition hands out uneven z_counts
// when nz % world != 0; wafer_ctx_create has already refused a slab thinner than its ghost depth).
bool fuse3_applies(const wafer_ctx *c)
{
    // Small undecomposed grids are launch- and fill-bound and the deeper pipeline costs there: 64^3 4.8 against 4.6 us/step for the
    // two-step kernel, 96^3 6.4 / 6.4; from 128^3 up it wins (7.6 / 8.1, 160^3 14.8 / 16.0, 192^3 19.4 / 20.8, 224^3 24.7 / 28.6 us;
    // round-5 kernels -- with round 3's the crossover was at 256^3).  WAFER_FUSE3_MIN_NY (tests) lifts both thresholds.
    const int ny_env = c->tune.fuse3_min_ny;
    const int min_ny = ny_env >= 0 ? ny_env : 16;
    const long long min_cells = ny_env >= 0 ? 0 : c->tune.fuse3_min_cells;
    if (!(active_variant(c) == 3 && c->g.R == 1 && c->g.ny >= min_ny)) return false;
    if (c->sharded()) return c->g.G >= 3 * c->g.R;
    return (long long)c->g.nx * c->g.ny * c->g.nz >= min_cells;
}

// The two-step kernel: every stencil order in fp64 (SevenPoint on 128 x 8 tiles, a and b formed again at
// the second step: its two seven-plane z-queues leave no registers for an a, b queue); ThreePoint /
// FivePoint on fp32 storage (SevenPoint there spills 200 B per lane and stays on the single-step kernel).
// Slabs need 2 * ext ghost planes (rank-invariant, as above).
bool fuse2_applies(const wafer_ctx *c)
{
    const int R = c->g.R;
    return active_variant(c) >= 2 && (R <= 2 || !c->f32) && (!c->sharded() || c->g.G >= 2 * R);
}

// ---- workgroup tables of the three-step kernel (wafer_stencil_fused3.hip.h), built once per launch shape ---------
int f3_table(wafer_ctx *c, int kind, int lz_lo, int lz_hi, int aux, const wafer_ctx::F3Table **out)
{
    for (const auto &t : c->f3_tables)
        if (t.kind == kind && t.lz_lo == lz_lo && t.lz_hi == lz_hi && t.aux == aux) {
            *out = &t;
            return WAFER_OK;
        }
    int tx_, ty_;
    wafer_step3_tile(type_combo(c, true), &tx_, &ty_);
    const int ntx = (c->g.nx + tx_ - 1) / tx_, nty = (c->g.ny + ty_ - 1) / ty_;
    std::vector<WaferF3Block> host;
    if (kind == F3_PLAIN) {
        wafer_f3_schedule_plain(host, ntx, nty, lz_lo, lz_hi, aux /* planes per workgroup */, c->tune.swz != 0, c->tune.f3_plain_down != 0);
    } else if (kind == F3_MIXED) {
        wafer_f3_schedule_mixed(host, ntx, nty, lz_lo, lz_hi, aux /* short workgroups per tile */);
    } else if (kind == F3_WHOLE) {
        // peer-store pass without a cut: aux bit 0 = marching down, bits 8 / 16 = a neighbour below / above
        const bool need_wait[2] = {(aux & 8) != 0, (aux & 16) != 0};
        wafer_f3_schedule_whole(host, ntx, nty, lz_lo, lz_hi, aux & 1, need_wait, 3 * c->g.R, c->tune.swz != 0);
    } else {
        // the single-launch pass: aux = the half dispatched first.  Both sides wait for their flag whether or not a
        // neighbour exists there: the flag also says that this rank's SEND of the planes about to be overwritten two
        // passes later has completed
        // aux & 4: peer stores (mode 3) -- a side waits only where a neighbour delivers (bits 8: below, 16: above), and no column
        // is cut short: there is no exchange kernel to hand CUs to
        // aux & 32: peer copies (mode 4) -- mode 2's waits (the flag of a side also says that the COPY of the planes about to be
        // overwritten has completed), and no short columns either: the copies need no CU
        const bool peer = (aux & 4) != 0;
        const bool need_wait[2] = {peer ? (aux & 8) != 0 : true, peer ? (aux & 16) != 0 : true};
        const int ntiles = ntx * nty;
        const int nshort = (peer || (aux & 32)) ? 0 : (ntiles >= 64 ? ntiles / 16 : 0);
        wafer_f3_schedule_halves(host, ntx, nty, lz_lo, lz_hi, lz_lo + (lz_hi - lz_lo) / 2, aux & 1, need_wait,
                                 (c->tune.hv_debug & 8) ? 0 : nshort, 4 /* pieces per short column */, 3 * c->g.R /* planes per exchange */, !(aux & 2),
                                 c->tune.hv_debug, c->tune.hv_layout);
    }
    wafer_ctx::F3Table t{};
    t.kind = kind; t.lz_lo = lz_lo; t.lz_hi = lz_hi; t.aux = aux;
    t.nblocks = (int)host.size();
    bool any_up = false, any_down = false;
    for (const auto &k : host) {
        if (k.down & 1) any_down = true;
        else any_up = true;
        if (k.bump >= 0) ++t.nbump[k.bump];
        if (((k.down >> 16) & 3) != 0) ++t.nbump[((k.down >> 16) & 3) - 1];   // whole-column peer passes count on both sides
    }
    t.dir = any_up && any_down ? 0 : (any_down ? 2 : 1);
    HIP_TRY(hipMalloc((void **)&t.dev, sizeof(WaferF3Block) * host.size()));
    hipError_t e = hipMemcpy(t.dev, host.data(), sizeof(WaferF3Block) * host.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(t.dev);
        return fail(WAFER_ERR_HIP, "workgroup table upload failed: %s", hipGetErrorString(e));
    }
    if (c->f3_tables.size() > 64) { // (shapes come from a handful of launch sites; a host cycling through slab shapes must not leak)
        for (auto &old : c->f3_tables) (void)hipFree(old.dev);
        c->f3_tables.clear();
    }
    c->f3_tables.push_back(t);
    *out = &c->f3_tables.back();
    return WAFER_OK;
}

// three fused steps over planes [lz_lo, lz_hi): phi[dst] = step(step(step(phi[src])))
// short_tail: the interior launch of a split slab pass (see wafer_f3_schedule_mixed)
int launch_step3(wafer_ctx *c, int src, int dst, int lz_lo, int lz_hi, hipStream_t s, bool short_tail = false)
{
    if (lz_hi <= lz_lo) return WAFER_OK;
    int tx_, ty_;
    wafer_step3_tile(type_combo(c, true), &tx_, &ty_);
    const int ntx = (c->g.nx + tx_ - 1) / tx_, nty = (c->g.ny + ty_ - 1) / ty_;
    const WaferStepArgs a = step_args(c, lz_lo, lz_hi);
    const wafer_ctx::F3Table *tab = nullptr;
    if (short_tail && lz_hi - lz_lo >= 8 * 4) TRY(f3_table(c, F3_MIXED, lz_lo, lz_hi, 4, &tab));
    else if (c->tune.f3_sched == 1 && !c->sharded() && lz_hi - lz_lo >= 16) TRY(f3_table(c, F3_HALVES, lz_lo, lz_hi, 2 /* no flags, no counters */, &tab));
    else TRY(f3_table(c, F3_PLAIN, lz_lo, lz_hi, wafer_f3_zchunk(c->tune, ntx, nty, lz_hi - lz_lo, c->num_cus), &tab));
    // More workgroups than CUs and whole rounds of them per layer of tiles: one launch per round of CUs (wafer_f3_zchunk says why).  The
    // table hands XCD k a contiguous band in blocks of 8, so a sub-range that starts at a multiple of 8 keeps every workgroup on the XCD the table meant it for.
    const int slots = c->tune.target_blocks > 0 ? c->tune.target_blocks : c->num_cus;
    const int per_launch = (tab->kind == F3_PLAIN && slots % 8 == 0 && wafer_f3_by_rounds(c->tune, (long long)ntx * nty, lz_hi - lz_lo, slots)) ? slots : tab->nblocks;
    for (int first = 0; first < tab->nblocks; first += per_launch) {
        const int nb = std::min(per_launch, tab->nblocks - first);
        if (wafer_entry_step3_fused(type_combo(c, true), c->tune, a, tab->dev + first, nb, WaferF3Sync(), c->phi[src], c->v, c->phi[dst], s, tab->dir) != hipSuccess)
            return fail(WAFER_ERR_HIP, "three-step stencil launch failed: %s", hipGetErrorString(hipGetLastError()));
    }
    c->last_instance_valid = true;
    return WAFER_OK;
}

// two fused steps over planes [lz_lo, lz_hi): phi[dst] = step(step(phi[src]))
// short_tail: the interior launch of a slab -- one long workgroup per tile, except the last 1/16 of
// the tiles, which go as four short workgroups each (see wafer_evolve)
int launch_step2(wafer_ctx *c, int src, int dst, int lz_lo, int lz_hi, hipStream_t s, bool short_tail = false)
{
    if (lz_hi <= lz_lo) return WAFER_OK;
    WaferStepArgs a = step_args(c, lz_lo, lz_hi);
    a.n_long = 0;
    a.nsub = short_tail ? 4 : 0;
    if (kernels_stream_ab(c, 2)) TRY(ensure_ab(c));
    if (wafer_entry_step2_fused(type_combo(c, true), c->g.R, c->tune, a, c->phi[src], c->a, c->b, c->v, c->phi[dst], s) != hipSuccess)
        return fail(WAFER_ERR_HIP, "fused stencil launch failed: %s", hipGetErrorString(hipGetLastError()));
    return WAFER_OK;
}


// elementwise launches (wafer_k_row_op) -----------------------------------------
// OP 0 norm2, 1 dot, 2 normalise (+ dot), 3 axpy (+ dot).  Returns the number of partial sums through *nb.
template <int OP>
static int launch_row_op(wafer_ctx *c, void *phi, const void *lower, const void *next, const double *scal_dev, double imm,
                         hipStream_t s, int *nb)
{
    WaferRowArgs ra;
    ra.g = c->g;
    ra.lz_lo = c->g.G;
    ra.lz_hi = c->g.G + c->g.nzl;
    // eight workgroups per CU, fewer on grids with fewer 1 KiB row segments than that
    const long long segs = (long long)c->g.nzl * c->g.ny * ((c->g.nx + (int)(1024 / c->esz) - 1) / (int)(1024 / c->esz));
    const dim3 grid((unsigned)std::max<long long>(1, std::min<long long>((long long)c->num_cus * 8, (segs + 3) / 4))), block(256);
    *nb = (int)grid.x;
    if ((size_t)grid.x > c->partials_stride) return fail(WAFER_ERR_INVALID, "partials buffer too small");
    return dispatch(c, [&](auto t, auto cc, auto) {
        using T = decltype(t);
        using C = decltype(cc);
        hipLaunchKernelGGL((wafer_k_row_op<T, C, OP>), grid, block, 0, s, ra, as<T>(phi), as<T>(lower), as<T>(next), scal_dev, imm,
                           c->partials);
        HIP_TRY(hipGetLastError());
        return (int)WAFER_OK;
    });
}

// normalise (+ optional overlap with lower) on buffer `buf`; norm2 from scal[slot] or immediate
int launch_normalise(wafer_ctx *c, int buf, const double *norm2_dev, double norm2_imm,
                            void *lower, int out_slot, hipStream_t s)
{
    int nb;
    TRY(launch_row_op<2>(c, c->phi[buf], lower, nullptr, norm2_dev, norm2_imm, s, &nb));
    if (lower) TRY(reduce_to_scal(c, 1, nb, out_slot, s));
    return WAFER_OK;
}

int launch_axpy(wafer_ctx *c, int buf, void *lower, int overlap_slot, void *next, int out_slot,
                       hipStream_t s)
{
    int nb;
    TRY(launch_row_op<3>(c, c->phi[buf], lower, next, c->scal + overlap_slot, 0.0, s, &nb));
    if (next) TRY(reduce_to_scal(c, 1, nb, out_slot, s));
    return WAFER_OK;
}

int launch_dot(wafer_ctx *c, void *phi, void *lower, int out_slot, hipStream_t s)
{
    int nb;
    TRY(launch_row_op<1>(c, phi, lower, nullptr, nullptr, 0.0, s, &nb));
    return reduce_to_scal(c, 1, nb, out_slot, s);
}

// Gram-Schmidt chain on `buf` against states [0,wnum); the first overlap is
// already in scal[1] when first_dot_done.
int gs_chain(wafer_ctx *c, int buf, uint32_t wnum, bool first_dot_done, hipStream_t s)
{
    if (wnum == 0) return WAFER_OK;
    if (!first_dot_done) TRY(launch_dot(c, c->phi[buf], c->states[0], 1, s));
    for (uint32_t l = 0; l < wnum; ++l) {
        void *next = (l + 1 < wnum) ? c->states[l + 1] : nullptr;
        TRY(launch_axpy(c, buf, c->states[l], 1 + (int)l, next, 2 + (int)l, s));
    }
    return WAFER_OK;
}

// Gram matrix of the stored states (lower triangle), recomputed whenever w_store changes.
int recompute_gram(wafer_ctx *c)
{
    c->x2_ready = 0;   // w_store changed: the images M_j and their matrices are rebuilt on demand (ensure_x2)
    const size_t n = c->states.size() < WAFER_MAX_LOW ? c->states.size() : WAFER_MAX_LOW;
    memset(c->gram_host, 0, sizeof c->gram_host);
    for (size_t j = 1; j < n; ++j)
        for (size_t i = 0; i < j; ++i) {
            TRY(launch_dot(c, c->states[j], c->states[i], 13, c->s_main));
            TRY(read_scal(c, 13, 1, &c->gram_host[j * WAFER_MAX_LOW + i], c->s_main));
        }
    HIP_TRY(hipMemcpyAsync(c->gram, c->gram_host, sizeof c->gram_host, hipMemcpyHostToDevice, c->s_main));
    HIP_TRY(hipStreamSynchronize(c->s_main));
    return WAFER_OK;
}

// Excited-state steps with everything fused that can be (wnum <= WAFER_MAX_LOW).
//   step kernel  phi' = step(x), sum phi'^2, t_j = sum l_j phi'   with x = phi (two-pass mode) or
//                x = raw/norm - sum_j l_j s_j formed on load from the previous raw step (one-pass mode)
//   reduce       1 + k scalars (one all-reduce when sharded)
//   apply        phi = phi'/norm - sum_j l_j s_j: after every step (two-pass), or once at the end
// One excited-state stencil launch over local planes [lz_lo, lz_hi): the step, sum phi'^2 and the
// raw overlaps with the stored states; the workgroups' partial sums go to partials[pbase + ...].
// Returns the number of partials written through *nb_out.
int excited_stencil_launch(wafer_ctx *c, int src, int dst, uint32_t wnum, bool transform_on_load, int lz_lo, int lz_hi,
                                  long long pbase, hipStream_t s, long long *nb_out, int zchunk = 0)
{
    const WaferGeom &g = c->g;
    *nb_out = 0;
    if (lz_hi <= lz_lo) return WAFER_OK;
    WaferLowPtrs low;
    for (uint32_t j = 0; j < wnum; ++j) low.p[j] = c->states[j];
    WaferStepArgs a = step_args(c, lz_lo, lz_hi);
    // ONE workgroup per CU (8 waves on a 128x16 tile for k <= 3): every workgroup streams 3 + k
    // arrays a plane ahead, and two per CU overflow the XCD's 4 MB L2, so the halo rows a
    // neighbour just loaded are gone again (512^3, 128x8 tiles: k = 2 1.24 -> 1.13 ms, k = 3
    // 1.45 -> 1.39).  The launcher doubles target_blocks.
    const int target = zchunk > 0 ? -zchunk  // planes per workgroup fixed by the caller (slab interior)
                            : (wnum >= 2 || wafer_excited_nw(c->tune, (int)wnum, c->g.R, c->f32) == 8) ? (c->num_cus + 1) / 2 : c->num_cus;
    a.target_blocks = target;
    const long long nb = dispatch(c, [&](auto t, auto, auto r) {
        return (int)wafer_step_lds_excited_blocks<decltype(t), decltype(r)::value>(c->tune, g, lz_lo, lz_hi, target, (int)wnum, transform_on_load);
    });
    if (pbase + nb > (long long)c->partials_stride) return fail(WAFER_ERR_INVALID, "partials buffer too small");
    if (wafer_entry_step_lds_excited(type_combo(c, false), g.R, c->tune, a, c->phi[src], c->v, c->phi[dst], c->partials + pbase,
                                     c->partials_stride /* the row stride of the partials, too */, (int)wnum, low, s,
                                     transform_on_load ? c->scal : nullptr, c->gram, closed_form_vg(c)) != hipSuccess)
        return fail(WAFER_ERR_HIP, "excited-state stencil launch failed: %s", hipGetErrorString(hipGetLastError()));
    *nb_out = nb;
    return WAFER_OK;
}

// the whole slab in one launch, then the 1 + wnum sums (all-reduced when sharded)
int excited_step_launch(wafer_ctx *c, int src, int dst, uint32_t wnum, bool transform_on_load, hipStream_t s)
{
    long long nb = 0;
    TRY(excited_stencil_launch(c, src, dst, wnum, transform_on_load, c->g.G, c->g.G + c->g.nzl, 0, s, &nb));
    return reduce_to_scal(c, 1 + (int)wnum, nb, 0, s);
}

// z-slabs: the R boundary planes of each side first, on the second stream, their (raw) halo exchange
// behind the interior launch; the sums wait for all three launches
int excited_step_launch_overlapped(wafer_ctx *c, int src, int dst, uint32_t wnum, bool transform_on_load)
{
    const WaferGeom &g = c->g;
    const int R = g.R, lo = g.G, hi = g.G + g.nzl;
    long long nb_lo = 0, nb_hi = 0, nb_in = 0;
    const hipStream_t sb = c->s_aux;
    HIP_TRY(hipEventRecord(c->ev_fork, c->s_main));
    HIP_TRY(hipStreamWaitEvent(c->s_aux, c->ev_fork, 0));
    if (c->has_lo()) TRY(excited_stencil_launch(c, src, dst, wnum, transform_on_load, lo, lo + R, 0, sb, &nb_lo));
    if (c->has_hi()) TRY(excited_stencil_launch(c, src, dst, wnum, transform_on_load, hi - R, hi, nb_lo, sb, &nb_hi));
    HIP_TRY(hipEventRecord(c->ev_bdry, sb));
    TRY(exchange_halo(c, dst, c->s_aux, R));        // enqueued before the interior: its kernels reach the CUs first
    HIP_TRY(hipEventRecord(c->ev_join, c->s_aux));
    HIP_TRY(hipStreamWaitEvent(c->s_main, c->ev_bdry, 0));
    // (one long workgroup per tile here: shorter ones -- the fused ground-state split's answer to CUs
    //  held by the exchange -- cost this kernel more in pipeline refills than the tail they avoid:
    //  k = 1 0.98 vs 1.01 ms, k = 3 1.57 vs 1.53 under an 8-channel RCCL kernel)
    TRY(excited_stencil_launch(c, src, dst, wnum, transform_on_load, c->has_lo() ? lo + R : lo, c->has_hi() ? hi - R : hi,
                               nb_lo + nb_hi, c->s_main, &nb_in));
    HIP_TRY(hipStreamWaitEvent(c->s_main, c->ev_join, 0));
    return reduce_to_scal(c, 1 + (int)wnum, nb_lo + nb_hi + nb_in, 0, c->s_main);
}

// ---- two excited-state steps per pass (wafer_stencil_x2.hip.h) ------------------------------------------------------
// ThreePoint fp64, one to three stored states; z-slabs need two ghost planes (a pass consumes two per side).  Nothing here
// depends on the local slab: what does (the potential inside the short reciprocal's range, two owned planes, memory for the
// images) is the ranks' agreement in x2_agree.
bool x2_applies(const wafer_ctx *c, uint32_t wnum)
{
    // three stored states: the 128 x 8-tile kernel wins where a plane is small enough for the halo rows to stay in the XCDs' L2
    // (-4 % per step at 512 x 512, -2 ... -4 % at 256^2 / 384^2) and loses on 1024 x 1024 planes (+3 ... +5 %,
    // profiles/r04_x2_shapes.log).  The plane extent is the same on every rank of a decomposed run.
    // fp32 storage (round 6): the one-step kernels are bound in the CU there, and the two-step pass wins on wide planes too -- 1024 x 1024 x 128,
    // k = 3: 0.68-0.69 against 0.78 ms/step (profiles/r06_sweep_x2_f32_wide_planes.jsonl)
    const int kmax = c->tune.x2_max_k > 0 ? c->tune.x2_max_k : ((c->f32 || (long long)c->g.nx * c->g.ny <= 300000) ? 3 : 2);
    // (fp32 storage -- dtype f32 and f32fast, whose excited-state steps compute in fp64 -- since round 6: the storage tag of the three-step kernel)
    return c->tune.x2 != 0 && c->tune.one_pass != 0 && c->g.R == 1 && wnum >= 1 && wnum <= 3 && (int)wnum <= kmax &&
           active_variant(c) >= 1 && (!c->sharded() || c->g.G >= 2);
}

// Storage for M_j = A l_j of the first wnum stored states.  Running out of memory here is not an error of the call that asked:
// the one-step path needs none of it (x2_agree).
int alloc_mstates(wafer_ctx *c, uint32_t wnum)
{
    while (c->mstates.size() < wnum) {
        void *slot = nullptr;
        TRY(alloc_grid_array(c, &slot, c->s_main));
        c->mstates.push_back(slot);
    }
    return WAFER_OK;
}

// The two-step pass changes what the ranks of a decomposed run exchange (two planes per pass, 2 + 3k sums), so every rank must
// take it or none.  x2_applies depends on nothing local; what does -- V inside the short reciprocal's range on this slab, two
// owned planes to send, memory for the images M_j -- is agreed on once per potential and number of stored states (a collective:
// every rank reaches its first excited-state wafer_evolve at that level together).  A rank that cannot take the pass makes
// every rank keep the one-step kernels: no error, and nobody is left in a collective.
int x2_agree(wafer_ctx *c, uint32_t wnum, bool *out)
{
    *out = false;
    if (!x2_applies(c, wnum)) return WAFER_OK;
    bool local_ok = short_forms(c) && (!c->sharded() || c->g.nzl >= 2);
    if (local_ok && alloc_mstates(c, wnum) != WAFER_OK) local_ok = false;
    if (!c->sharded()) { *out = local_ok; return WAFER_OK; }
    if (!c->allreduce_hook) return WAFER_OK;
    if (c->x2_agreed[wnum] < 0) {
        c->scal_host[13] = local_ok ? 0.0 : 1.0;
        HIP_TRY(hipMemcpyAsync(c->scal + 13, c->scal_host + 13, sizeof(double), hipMemcpyHostToDevice, c->s_main));
        if (c->allreduce_hook(c->hook_user, c->scal + 13, 1, (void *)c->s_main) != 0) return fail(WAFER_ERR_COMM, "allreduce hook failed");
        double bad = 1.0;
        TRY(read_scal(c, 13, 1, &bad, c->s_main));
        c->x2_agreed[wnum] = bad == 0.0 ? 1 : 0;
    }
    *out = c->x2_agreed[wnum] == 1;
    return WAFER_OK;
}

// M_j = A l_j for the first wnum stored states (one ground-state step of each, grid.rs:568-592) and the matrix <l_j, M_i> of
// the coefficient kernel.  Rebuilt when w_store or the potential changed.  (Storage: alloc_mstates, through x2_agree.)
int ensure_x2(wafer_ctx *c, uint32_t wnum)
{
    if (c->x2_ready >= (int)wnum) return WAFER_OK;
    const WaferGeom &g = c->g;
    TRY(alloc_mstates(c, wnum));
    if (kernels_stream_ab(c, 1)) TRY(ensure_ab(c));
    for (uint32_t j = 0; j < wnum; ++j) {
        // z-slabs: the pass transforms two ghost planes per side, so l_j and M_j must be current there (a stored state
        // carries one ghost plane from wafer_push_state; the second, and M_j's two, come from the neighbours now)
        TRY(exchange_halo_array(c, c->states[j], c->s_main, 2));
        const WaferStepArgs a = step_args(c, g.G, g.G + g.nzl);
        if (wafer_entry_step_lds(type_combo(c, false), g.R, c->tune, a, c->states[j], c->a, c->b, c->v, c->mstates[j], c->s_main, closed_form_vg(c)) != hipSuccess)
            return fail(WAFER_ERR_HIP, "stencil launch (image of a stored state) failed: %s", hipGetErrorString(hipGetLastError()));
        TRY(exchange_halo_array(c, c->mstates[j], c->s_main, 2));
    }
    double host[WAFER_MAX_LOW * WAFER_MAX_LOW];
    memset(host, 0, sizeof host);
    double *amat = host;
    for (uint32_t j = 0; j < wnum; ++j)
        for (uint32_t i = 0; i < wnum; ++i) {   // <l_j, M_i>
            TRY(launch_dot(c, c->mstates[i], c->states[j], 13, c->s_main));
            TRY(read_scal(c, 13, 1, &amat[j * WAFER_MAX_LOW + i], c->s_main));
        }
    HIP_TRY(hipMemcpyAsync(c->x2mat, host, sizeof host, hipMemcpyHostToDevice, c->s_main));
    HIP_TRY(hipStreamSynchronize(c->s_main));
    c->x2_ready = (int)wnum;
    return WAFER_OK;
}

// `pairs` two-step passes from the raw result of a one-step kernel (phi[cur] = A x, its sums in scal[0 .. wnum]), then phi
// materialised: 2 * pairs steps of grid.rs:562-686
int x2_run(wafer_ctx *c, uint32_t wnum, uint64_t pairs, hipStream_t s)
{
    const WaferGeom &g = c->g;
    const int k = (int)wnum, nq = wafer_entry_x2_nsums(k);
    const double *amat = c->x2mat;
    const void *l[3] = {nullptr, nullptr, nullptr}, *m[3] = {nullptr, nullptr, nullptr};
    for (int j = 0; j < k; ++j) { l[j] = c->states[j]; m[j] = c->mstates[j]; }
    if (wafer_entry_x2_coeffs(1, k, c->scal, c->gram, amat, c->x2coef, s) != hipSuccess)
        return fail(WAFER_ERR_HIP, "coefficient kernel launch failed");
    const WaferStepArgs a = step_args(c, g.G, g.G + g.nzl);
    const long long nb = wafer_entry_x2_blocks(type_combo(c, false), c->tune, g, k, closed_form_vg(c), g.G, g.G + g.nzl, c->num_cus);
    if (nb > (long long)c->partials_stride) return fail(WAFER_ERR_INVALID, "partials buffer too small");
    TRY(ensure_halo(c, 2));   // z-slabs: two ghost planes of the raw input per side and pass
    for (uint64_t p = 0; p < pairs; ++p) {
        const int src = c->cur, dst = c->cur ^ 1;
        if (wafer_entry_xstep2(type_combo(c, false), c->tune, a, k, closed_form_vg(c), c->phi[src], c->v, c->phi[dst], c->partials, c->partials_stride, l, m,
                               c->x2coef, s) != hipSuccess)
            return fail(WAFER_ERR_HIP, "two-step excited-state stencil launch failed: %s", hipGetErrorString(hipGetLastError()));
        ++c->x2_passes;
        if (p + 1 < pairs) TRY(exchange_halo(c, dst, s, 2));   // (unsplit: a short exchange takes CUs from a launch that packs them, as for one step per pass)
        TRY(reduce_to_scal(c, nq, nb, X2_SUM_SLOT, s));
        if (wafer_entry_x2_coeffs(2, k, c->scal + X2_SUM_SLOT, c->gram, amat, c->x2coef, s) != hipSuccess)
            return fail(WAFER_ERR_HIP, "coefficient kernel launch failed");
        c->cur = dst;
    }
    // phi = x~ / n_c: the last step's normalisation (grid.rs:679), its norm taken directly as the sum of squares of Y2
    int nap = 0;
    if (wafer_entry_x2_apply(type_combo(c, false), g, g.G, g.G + g.nzl, k, c->phi[c->cur], l, m, c->x2coef, c->partials, c->partials_stride, c->num_cus, s, &nap) != hipSuccess)
        return fail(WAFER_ERR_HIP, "apply launch failed");
    TRY(reduce_to_scal(c, 1, nap, X2_SUM_SLOT, s));
    TRY(launch_normalise(c, c->cur, c->scal + X2_SUM_SLOT, 0.0, nullptr, 0, s));
    c->halo_valid = 0;
    return WAFER_OK;
}

int excited_apply(wafer_ctx *c, int buf, uint32_t wnum, hipStream_t s)
{
    WaferLowPtrs low;
    for (uint32_t j = 0; j < wnum; ++j) low.p[j] = c->states[j];
    return dispatch(c, [&](auto t, auto cc, auto) {
        using T = decltype(t);
        using C = decltype(cc);
        WaferRowArgs ra;
        ra.g = c->g;
        ra.lz_lo = c->g.G;
        ra.lz_hi = c->g.G + c->g.nzl;
        const dim3 grid(c->num_cus * 8), block(256);
        T *p = as<T>(c->phi[buf]);
        switch (wnum) {
        case 1: hipLaunchKernelGGL((wafer_k_gs_apply<T, C, 1>), grid, block, 0, s, ra, p, low, c->scal, c->gram); break;
        case 2: hipLaunchKernelGGL((wafer_k_gs_apply<T, C, 2>), grid, block, 0, s, ra, p, low, c->scal, c->gram); break;
        case 3: hipLaunchKernelGGL((wafer_k_gs_apply<T, C, 3>), grid, block, 0, s, ra, p, low, c->scal, c->gram); break;
        default: hipLaunchKernelGGL((wafer_k_gs_apply<T, C, 4>), grid, block, 0, s, ra, p, low, c->scal, c->gram); break;
        }
        HIP_TRY(hipGetLastError());
        return (int)WAFER_OK;
    });
}


} // namespace wafer_eng

extern "C" {

int wafer_evolve(wafer_ctx *c, uint32_t wnum, uint64_t n_steps)
{
    if (!c) return fail(WAFER_ERR_INVALID, "null context");
    if (!c->have_pot || !c->have_phi) return fail(WAFER_ERR_STATE, "potential and phi must be set before evolve");
    if (wnum > c->states.size()) return fail(WAFER_ERR_STATE, "wnum %u but w_store holds %zu states", wnum, c->states.size());
    if (wnum + 2 > SCAL_SLOTS) return fail(WAFER_ERR_INVALID, "wnum too large");
    HIP_TRY(hipSetDevice(c->P.device));
    RoctxRange range_(wnum ? "wafer_evolve_excited" : "wafer_evolve_ground");
    const WaferGeom &g = c->g;
    const int R = g.R;
    const int lo = g.G, hi = g.G + g.nzl;
    const uint64_t steps = n_steps == 0 ? 1 : n_steps; // grid.rs:682-685
    // two steps per pass where nothing happens between steps (ground state) and, when the grid
    // is sharded, the slab carries 2R ghost planes
    const bool fuse = wnum == 0 && fuse2_applies(c);
    const bool fuse3 = wnum == 0 && fuse3_applies(c);
    // Excited states, two steps per pass: the first two steps (three for an odd count) run one per pass -- whatever the
    // caller hands over (a clone of a stored state, an un-normalised start) is normalised and projected by the reference's own
    // sequence before the regrouped sums take over -- then pairs; phi is materialised after the last pass.
    bool x2 = false;
    if (wnum > 0 && steps >= 4) TRY(x2_agree(c, wnum, &x2));
    const uint64_t x2_head = x2 ? 2 + (steps & 1) : steps;
    if (x2) TRY(ensure_x2(c, wnum));
    HIP_TRY(hipEventRecord(c->ev_start, c->s_main));
    // single-launch passes in flight: their last exchanges have not been waited for by the main stream
    bool hv_active = false, hv_peer = false;
    int hv_depth = 0;
    auto hv_drain = [&]() -> int {
        if (!hv_active) return WAFER_OK;
        if (hv_peer) {
            TRY(peer_drain(c));
        } else {
            HIP_TRY(hipStreamWaitEvent(c->s_main, c->ev_ex[0], 0));
            HIP_TRY(hipStreamWaitEvent(c->s_main, c->ev_ex[1], 0));
        }
        hv_active = false;
        c->halo_valid = hv_depth;
        return WAFER_OK;
    };
    for (uint64_t s = 0; s < steps;) {
        const int src = c->cur, dst = c->cur ^ 1;
        if ((fuse3 && steps - s >= 3) || (fuse && steps - s >= 2)) {
            // K time steps per pass: three on the three-step kernel while at least three remain, else two
            const int K = (fuse3 && steps - s >= 3) ? 3 : 2, H = K * R; // H: ghost planes one pass consumes per side
            auto launch_pass = [&](int zlo, int zhi, hipStream_t st, bool short_tail) {
                return K == 3 ? launch_step3(c, src, dst, zlo, zhi, st, short_tail) : launch_step2(c, src, dst, zlo, zhi, st, short_tail);
            };
            // Deep halos: with E = H * halo_cycle ghost planes exchanged at once, only every halo_cycle-th
            // pass needs boundary-first kernels, an exchange and the event hops around them.  The passes in
            // between run UNSPLIT over the owned planes plus the ghost planes that are still good for one more
            // pass: each fused pass consumes H planes of validity per side (the neighbour computes the
            // same cells from the same values, so the bits agree).  E is a whole number of passes' worth and the same
            // on every rank (the neighbours receive what this one sends).
            const int E = c->sharded() ? std::max(H, std::min(g.G, H * c->halo_cycle) / H * H) : H;
            // Mode 2: the whole slab in one launch (three-step passes with one exchange per pass; every rank takes this
            // branch or none: K, E and H depend on nothing local)
            if (c->sharded() && (c->sched == 2 || c->sched == 3) && K == 3 && E == H) {
                const bool peer = c->sched == 3;
                if (!hv_active) {
                    TRY(ensure_hv(c));
                    // the first pass's ghost planes: a plain exchange in stream order.  (Peer mode: always, also when they are
                    // current -- the collective is the rendezvous that keeps a rank from storing into a neighbour's buffers while
                    // that neighbour is still busy with whatever preceded this call.)
                    // (stream order suffices: my first pass follows my exchange, which completes only when the neighbour's stream has
                    //  reached its own)
                    if (peer) c->halo_valid = 0;
                    TRY(ensure_halo(c, E));
                    hv_active = true;
                    hv_peer = peer;
                    hv_depth = E;
                }
                if (peer) TRY(launch_peer_pass(c, src, dst, E));
                else TRY(launch_halves_pass(c, src, dst, E));
                c->halo_valid = 0;   // (inside the mode; hv_drain restores the invariant)
                c->cur = dst;
                s += K;
                continue;
            }
            TRY(hv_drain());
            if (c->sharded() && c->halo_valid < H) TRY(ensure_halo(c, E));
            if (c->sharded() && c->halo_valid >= 2 * H) {
                const int ext = c->halo_valid - H; // ghost planes still valid after this pass
                TRY(launch_pass(c->has_lo() ? lo - ext : lo, c->has_hi() ? hi + ext : hi, c->s_main, false));
                c->halo_valid = ext;
                c->cur = dst;
                s += K;
                continue;
            }
            const bool split = c->sharded() && c->sched != 0 && g.nzl > 2 * E;
            if (split) {
                // Mode 1.  Second stream: boundary planes, then their exchange.  Main stream: the interior, released
                // by an event recorded after the boundary kernels.  The exchange is enqueued BEFORE the
                // interior launch and needs no event hop, so its kernels reach the CUs first; the interior
                // then fills what is left.  (Without the dependency the interior started first, filled
                // every CU for a whole round, and the boundary kernels -- and the exchange behind them --
                // finished only with the pass; with the exchange merely enqueued second, RCCL's
                // workgroups waited 0.35 ms for CUs: profiles/r01_slab_overlap_timeline.txt.)
                HIP_TRY(hipEventRecord(c->ev_fork, c->s_main));
                HIP_TRY(hipStreamWaitEvent(c->s_aux, c->ev_fork, 0));
                if (c->has_lo()) TRY(launch_pass(lo, lo + E, c->s_aux, false));
                if (c->has_hi()) TRY(launch_pass(hi - E, hi, c->s_aux, false));
                HIP_TRY(hipEventRecord(c->ev_bdry, c->s_aux));
                TRY(exchange_halo(c, dst, c->s_aux, E));
                HIP_TRY(hipEventRecord(c->ev_join, c->s_aux));
                HIP_TRY(hipStreamWaitEvent(c->s_main, c->ev_bdry, 0));
                // The exchange's kernels hold a few CUs for as long as the links need (RCCL's workgroups
                // cannot share a CU with a stencil workgroup).  With one long workgroup per tile every
                // displaced workgroup would add a whole extra round at the end of the pass (measured with
                // an 8-channel RCCL kernel of realistic length: 0.465 ms/step, worse than no overlap).
                // Cutting EVERY tile into four workgroups fixes that at 3 planes of pipeline fill per
                // workgroup (0.396); cutting only the last 1/16 of the tiles -- dispatched last, they
                // fill the holes -- keeps the long workgroups' efficiency.
                TRY(launch_pass(c->has_lo() ? lo + E : lo, c->has_hi() ? hi - E : hi, c->s_main, true));
                HIP_TRY(hipStreamWaitEvent(c->s_main, c->ev_join, 0));
            } else {
                TRY(launch_pass(lo, hi, c->s_main, false));
                TRY(exchange_halo(c, dst, c->s_main, E));
            }
            c->halo_valid = c->sharded() ? E : H;
            c->cur = dst;
            s += K;
            continue;
        }
        TRY(hv_drain());
        TRY(ensure_halo(c, R));
        if (wnum == 0) {
            const bool split = c->sharded() && c->sched != 0 && g.nzl > 2 * R;
            if (split) {
                // boundary planes and their exchange on the second stream, the interior behind an event (as above)
                HIP_TRY(hipEventRecord(c->ev_fork, c->s_main));
                HIP_TRY(hipStreamWaitEvent(c->s_aux, c->ev_fork, 0));
                if (c->has_lo()) TRY(launch_step(c, src, dst, lo, lo + R, false, c->s_aux));
                if (c->has_hi()) TRY(launch_step(c, src, dst, hi - R, hi, false, c->s_aux));
                HIP_TRY(hipEventRecord(c->ev_bdry, c->s_aux));
                TRY(exchange_halo(c, dst, c->s_aux, R));
                HIP_TRY(hipEventRecord(c->ev_join, c->s_aux));
                HIP_TRY(hipStreamWaitEvent(c->s_main, c->ev_bdry, 0));
                TRY(launch_step(c, src, dst, c->has_lo() ? lo + R : lo, c->has_hi() ? hi - R : hi, false, c->s_main));
                HIP_TRY(hipStreamWaitEvent(c->s_main, c->ev_join, 0));
            } else {
                TRY(launch_step(c, src, dst, lo, hi, false, c->s_main));
                TRY(exchange_halo(c, dst, c->s_main, R));
            }
        } else {
            // step + sum phi'^2 (grid.rs:675-678), normalise (:679), Gram-Schmidt (:680)
            if (x2 && s == x2_head) {
                TRY(x2_run(c, wnum, (steps - x2_head) / 2, c->s_main));
                s = steps;
                continue;
            }
            if (wnum <= WAFER_MAX_LOW && active_variant(c) >= 1) {
                // one pass per step: the raw result travels to the next step, which normalises and
                // projects it on load; phi is materialised once after the last step
                const bool one_pass = c->tune.one_pass != 0;
                const bool last = s + 1 == steps;   // (never within the head of a two-steps-per-pass run)
                if (one_pass && s == 0) {
                    hipLaunchKernelGGL(wafer_k_identity_scalars, dim3(1), dim3(64), 0, c->s_main, c->scal, 1 + (int)wnum);
                    HIP_TRY(hipGetLastError());
                }
                // z-slabs, one-pass scheme, not the last step: the raw result's halo exchange hides behind
                // the interior launch (the last step's phi is materialised first and exchanged on demand)
                // (only when asked for by mode 1.  One plane per side and step is a short exchange, and its kernels take CUs
                //  from an interior launch that packs the CUs exactly: the interior ends later by about the exchange's own
                //  duration, and the two thin boundary launches come on top -- bench slab, native RCCL to the same rank,
                //  k = 1: 0.772 ms/step split against 0.718 unsplit (undecomposed 0.643); k = 3: 1.210 against 1.121 (1.033).)
                // The split depends on the LOCAL slab thickness (slab.partition hands out uneven slabs: 3, 2, 2, 2 planes of 9), so
                // both branches call the hooks in the SAME ORDER -- the halo exchange first, the all-reduce of the sums second --
                // or ranks that took different branches would queue a send / receive and a collective on one communicator in
                // different orders and wait for each other for ever (found by tests/fuzz_slabs.py, round 5, with the in-process
                // fabric; RCCL would have hung).
                const bool split = one_pass && !last && c->sharded() && c->sched == 1 && g.nzl > 2 * R;
                if (split) {
                    TRY(excited_step_launch_overlapped(c, src, dst, wnum, one_pass));
                } else if (one_pass && !last) {
                    long long nb = 0;
                    TRY(excited_stencil_launch(c, src, dst, wnum, one_pass, g.G, g.G + g.nzl, 0, c->s_main, &nb));
                    TRY(exchange_halo(c, dst, c->s_main, R));                       // the raw result's planes (stream order: behind the launch)
                    TRY(reduce_to_scal(c, 1 + (int)wnum, nb, 0, c->s_main));
                } else {
                    TRY(excited_step_launch(c, src, dst, wnum, one_pass, c->s_main));
                    if (!one_pass || last) TRY(excited_apply(c, dst, wnum, c->s_main));
                    if (!last || !one_pass) TRY(exchange_halo(c, dst, c->s_main, R));
                }
                c->halo_valid = (one_pass && last) ? 0 : R;
                c->cur = dst;
                s += 1;
                continue;
            }
            TRY(launch_step(c, src, dst, lo, hi, true, c->s_main));
            TRY(reduce_to_scal(c, 1, step_partials_count(c, lo, hi), 0, c->s_main));
            TRY(launch_normalise(c, dst, c->scal + 0, 0.0, c->states[0], 1, c->s_main));
            TRY(gs_chain(c, dst, wnum, true, c->s_main));
            TRY(exchange_halo(c, dst, c->s_main, R));
        }
        c->halo_valid = R;
        c->cur = dst;
        s += 1;
    }
    TRY(hv_drain());
    HIP_TRY(hipEventRecord(c->ev_stop, c->s_main));
    c->last_steps = steps;
    c->timing_valid = true;
    return WAFER_OK;
}

int wafer_last_evolve_ms(wafer_ctx *c, float *ms, uint64_t *steps)
{
    if (!c || !ms) return fail(WAFER_ERR_INVALID, "null argument");
    if (!c->timing_valid) return fail(WAFER_ERR_STATE, "no evolve call to time");
    HIP_TRY(hipSetDevice(c->P.device));
    HIP_TRY(hipEventSynchronize(c->ev_stop));
    TRY(check_hv_err(c));
    HIP_TRY(hipEventElapsedTime(ms, c->ev_start, c->ev_stop));
    if (steps) *steps = c->last_steps;
    return WAFER_OK;
}

int wafer_stencil_steps_per_launch(wafer_ctx *c);
const char *wafer_stencil_kernel_name(wafer_ctx *c)
{
    if (!c) return "";
    int v = active_variant(c);
    const int spl = wafer_stencil_steps_per_launch(c);
    if (v >= 2) v = spl == 3 ? 3 : spl == 2 ? 2 : 1; // what the fused variants fall back to where they do not apply
    // the name of the kernel that is LAUNCHED, not of the variant family: the two-step entry point hands FivePoint on to the
    // 128 x 16-tile kernel (wafer_entry_step2_fused -> wafer_entry_step2_wide unless WAFER_F2_WIDE=0)
    if (v == 2 && c->g.R == 2 && c->tune.f2_wide != 0) return "wafer_k_step2_wide";
    return kVariants[(v >= 0 && v < kNumVariants) ? v : 0].name;
}

// The template-id of the kernel the last ground-state pass launched, as a profiler prints it (e.g.
// "wafer_k_step3_fused<double, double, true, 0, true, 1>"): what bench.py writes into roofline.kernel and matches the committed
// counter figures by.  Falls back to the family name (no template arguments) for the families that do not record theirs.
const char *wafer_stencil_kernel_instance(wafer_ctx *c)
{
    if (!c) return "";
    if (c->last_instance_valid && wafer_stencil_steps_per_launch(c) == 3) {
        wafer_step3_last_instance(c->instance_name, sizeof c->instance_name);
        if (c->instance_name[0]) return c->instance_name;
    }
    if (wafer_stencil_steps_per_launch(c) == 2 && c->g.R == 2 && c->tune.f2_wide != 0) {   // wafer_entry_step2_fused hands FivePoint on
        const int tc = type_combo(c, true);
        snprintf(c->instance_name, sizeof c->instance_name, "wafer_k_step2_wide<%s, %s, %s>",
                 tc == WAFER_TC_F64 ? "double" : tc == WAFER_TC_F32_F64 ? "wafer_f32_wide" : "float", tc == WAFER_TC_F32_F32 ? "float" : "double",
                 short_forms(c) ? "true" : "false");
        return c->instance_name;
    }
    return wafer_stencil_kernel_name(c);
}

int wafer_stencil_steps_per_launch(wafer_ctx *c)
{
    if (!c) return 0;
    if (fuse3_applies(c)) return 3;
    return fuse2_applies(c) ? 2 : 1;
}

// Diagnostic: which kernel a pass of wafer_evolve(ctx, wnum, n) launches for this context, in one line of key=value pairs --
// the SAME predicates the launch path evaluates (fuse3_applies / fuse2_applies / x2_applies / wafer_excited_nw / closed_form_vg),
// nothing launched.  tools/dispatch_table.py tabulates it over stencil x dtype x wnum x grid x slab; tests/test_gpu_configs.py
// holds it against what then ran (wafer_stencil_kernel_instance, wafer_diag_x2_passes).
int wafer_diag_dispatch(wafer_ctx *c, uint32_t wnum, char *buf, size_t n)
{
    if (!c || !buf || n == 0) return fail(WAFER_ERR_INVALID, "null argument");
    const int R = c->g.R;
    const char *dtype = !c->f32 ? "f64" : (c->f32_arith ? "f32fast" : "f32");
    if (wnum == 0) {
        const int K = wafer_stencil_steps_per_launch(c);
        const char *kernel = wafer_stencil_kernel_name(c);
        int tx = 0, ty = 0;
        if (K == 3) wafer_step3_tile(type_combo(c, true), &tx, &ty);
        // what advances the steps a whole pass does not cover (wafer_evolve: three while three remain, then two, then one)
        const char *single = active_variant(c) == 0 ? "wafer_k_step_direct" : "wafer_k_step_lds";
        const char *two = (fuse2_applies(c) && K == 3) ? "wafer_k_step2_fused" : nullptr;
        char tile[32] = "";
        if (tx) snprintf(tile, sizeof tile, " tile=%dx%d", tx, ty);
        snprintf(buf, n, "wnum=0 stencil=%d dtype=%s kernel=%s steps_per_pass=%d ghost_planes_per_pass=%d%s v=%s remainder=%s%s%s", R, dtype, kernel, K, K * R,
                 tile, "streamed", K >= 2 ? (two ? two : single) : "-", (K == 3 && two) ? "," : "", (K == 3 && two) ? single : "");
        return WAFER_OK;
    }
    const int vg = closed_form_vg(c);
    if (x2_applies(c, wnum)) {
        int tx = 0, ty = 0;
        wafer_x2_tile_host(type_combo(c, false), c->tune, (int)wnum, vg, &tx, &ty);
        snprintf(buf, n, "wnum=%u stencil=%d dtype=%s kernel=wafer_k_xstep2 steps_per_pass=2 ghost_planes_per_pass=2 tile=%dx%d v=%s head=wafer_k_step_lds "
                         "condition=every_rank_agrees,n_steps>=4", wnum, R, dtype, tx, ty, vg ? "closed_form" : "streamed");
        return WAFER_OK;
    }
    if (wnum <= WAFER_MAX_LOW && active_variant(c) >= 1) {
        const int nw = wafer_excited_nw(c->tune, (int)wnum, R, c->f32);
        // the closed form is evaluated by the fp64 transform-on-load kernel on 8-wave tiles only (wafer_launch_step_lds_excited)
        const bool cf = vg != 0 && !c->f32 && c->tune.one_pass != 0 && nw == 8 && short_forms(c);
        snprintf(buf, n, "wnum=%u stencil=%d dtype=%s kernel=wafer_k_step_lds nlow=%u steps_per_pass=1 ghost_planes_per_pass=%d tile=128x%d waves=%d v=%s", wnum, R,
                 dtype, wnum, R, nw * 2, nw, cf ? "closed_form" : "streamed");
        return WAFER_OK;
    }
    snprintf(buf, n, "wnum=%u stencil=%d dtype=%s kernel=%s nlow=0 steps_per_pass=1 ghost_planes_per_pass=%d then=wafer_k_row_op(normalise),wafer_k_row_op(gram_schmidt)x%u",
             wnum, R, dtype, active_variant(c) == 0 ? "wafer_k_step_direct" : "wafer_k_step_lds", R, wnum);
    return WAFER_OK;
}

int wafer_set_stencil_variant(wafer_ctx *c, int variant)
{
    if (!c) return fail(WAFER_ERR_INVALID, "null context");
    if (variant >= kNumVariants) return fail(WAFER_ERR_INVALID, "variant %d out of range (have %d)", variant, kNumVariants);
    c->variant = variant;
    return WAFER_OK;
}

// ---- compute_observables (grid.rs:303-445) ------------------------------------------
int wafer_observables(wafer_ctx *c, wafer_observables_t *out)
{
    if (!c || !out) return fail(WAFER_ERR_INVALID, "null argument");
    if (!c->have_pot || !c->have_phi) return fail(WAFER_ERR_STATE, "potential and phi must be set");
    HIP_TRY(hipSetDevice(c->P.device));
    RoctxRange range_("wafer_observables");
    TRY(ensure_halo(c, c->g.R));
    const int R = c->g.R;
    long long nb = 0;
    {
        // the LDS pipeline of the step kernel in its observables mode: 16 B per lane from HBM
        WaferStepArgs sa{};
        sa.g = c->g;
        sa.lz_lo = c->g.G;
        sa.lz_hi = c->g.G + c->g.nzl;
        sa.dt = c->P.dt;
        set_den_args(c, sa);   // grid.rs:314 / 337 / 367: the step's denominator
        sa.target_blocks = c->num_cus;
        sa.v_in_range = short_forms(c) ? 1 : 0;
        sa.potsub_kind = c->potsub_kind;
        sa.potsub_scalar = c->potsub_scalar;
        set_vg_args(c, sa);
        if (wafer_entry_observables_lds(type_combo(c, false), R, c->tune, sa, c->phi[c->cur], c->v, c->potsub, c->partials, c->partials_stride,
                                        c->s_main, &nb, closed_form_vg(c)) != hipSuccess)
            return fail(WAFER_ERR_HIP, "observables launch failed: %s", hipGetErrorString(hipGetLastError()));
    }
    TRY(reduce_to_scal(c, 4, nb, 8, c->s_main));
    double r[4];
    TRY(read_scal(c, 8, 4, r, c->s_main));
    out->energy = r[0];
    out->norm2 = r[1];
    out->v_infinity = (c->potsub_kind == WAFER_POTSUB_NONE) ? 0.0 : r[2]; // grid.rs:425
    out->r2 = r[3];
    return WAFER_OK;
}

int wafer_norm2(wafer_ctx *c, double *out)
{
    if (!c || !out) return fail(WAFER_ERR_INVALID, "null argument");
    if (!c->have_phi) return fail(WAFER_ERR_STATE, "phi not set");
    HIP_TRY(hipSetDevice(c->P.device));
    int nb;
    TRY(launch_row_op<0>(c, c->phi[c->cur], nullptr, nullptr, nullptr, 0.0, c->s_main, &nb));
    TRY(reduce_to_scal(c, 1, nb, 12, c->s_main));
    return read_scal(c, 12, 1, out, c->s_main);
}

int wafer_normalise(wafer_ctx *c, double norm2)
{
    if (!c) return fail(WAFER_ERR_INVALID, "null context");
    if (!c->have_phi) return fail(WAFER_ERR_STATE, "phi not set");
    HIP_TRY(hipSetDevice(c->P.device));
    TRY(launch_normalise(c, c->cur, nullptr, norm2, nullptr, 0, c->s_main));
    c->halo_valid = 0;
    return WAFER_OK;
}

int wafer_orthogonalise(wafer_ctx *c, uint32_t wnum)
{
    if (!c) return fail(WAFER_ERR_INVALID, "null context");
    if (!c->have_phi) return fail(WAFER_ERR_STATE, "phi not set");
    if (wnum > c->states.size()) return fail(WAFER_ERR_STATE, "wnum %u but w_store holds %zu states", wnum, c->states.size());
    if (wnum + 2 > SCAL_SLOTS) return fail(WAFER_ERR_INVALID, "wnum too large");
    HIP_TRY(hipSetDevice(c->P.device));
    TRY(gs_chain(c, c->cur, wnum, false, c->s_main));
    if (wnum) c->halo_valid = 0;
    return WAFER_OK;
}

} // extern "C"
