// wafer_engine.hip -- the context and everything that sets it up or reads it back: arrays, potentials, wavefunction and
// w_store, host <-> device layout conversion, diagnostics.  (Launch logic: wafer_engine_schedules.hip; z-slabs:
// wafer_engine_comm.hip; the solve loop: wafer_engine_solve.hip; shared declarations: wafer_engine.h.)
//
// Host side mirrors the call structure of Wafer's grid.rs (run/solve/evolve/
// compute_observables/normalise/orthogonalise) with device-resident state.
// No CPU fallback exists: every entry point fails loudly if HIP does.
#include "wafer_engine.h"
#include "wafer_elementwise.hip.h"
#include "wafer_setup.hip.h"

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local std::string g_last_error;

int wafer_eng::fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    // HIP keeps the last error per thread until it is read: a failed hipMalloc would otherwise
    // resurface at the next hipGetLastError() after a perfectly good kernel launch
    if (code == WAFER_ERR_HIP) (void)hipGetLastError();
    return code;
}

// (other translation units of the library report through the same thread-local message: wafer_mailbox.hip)
void wafer_set_last_error(const char *msg) { g_last_error = msg ? msg : ""; }

// ---------------------------------------------------------------------------
// host restatement of the two scalar helpers FullCornell needs
// (potential.rs:374-391, 394-398); evaluated once per context.
// ---------------------------------------------------------------------------
static double host_alphas(double mu)
{
    const double nf = 2.0;
    const double b0 = 11. - 2. * nf / 3.;
    const double b1 = 51. - 19. * nf / 3.;
    const double b2 = 2857. - 5033. * nf / 9. + 325. * nf * nf / 27.;
    const double l = 2. * std::log(mu / 2.3);
    const double ll = std::log(l);
    return 4. * WAFER_PI *
           (1. - 2. * b1 * ll / (b0 * b0 * l) +
            4. * b1 * b1 * ((ll - 0.5) * (ll - 0.5) + b2 * b0 / (8. * b1 * b1) - 5.0 / 4.0) /
                (b0 * b0 * b0 * b0 * l * l)) /
           (b0 * l);
}

static double host_mu(double t)
{
    const double nf = 2.0, tc = 0.2;
    return 1.4 * std::sqrt((1. + nf / 6.) * 4. * WAFER_PI * host_alphas(2. * WAFER_PI * t)) * t * tc;
}

namespace wafer_eng __attribute__((visibility("hidden"))) {
int alloc_grid_array(wafer_ctx *c, void **logical, hipStream_t s)
{
    void *raw = nullptr;
    const size_t bytes = (size_t)c->g.total * c->esz;
    HIP_TRY(hipMalloc(&raw, bytes));
    hipError_t e = hipMemsetAsync(raw, 0, bytes, s);
    if (e != hipSuccess) {
        (void)hipFree(raw);
        return fail(WAFER_ERR_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(e));
    }
    *logical = static_cast<char *>(raw) + (size_t)c->g.base_off * c->esz;
    return WAFER_OK;
}
// The stored a, b arrays (potential.rs:101-110) are needed by the kernels that stream them (variant 0,
// WAFER_ABV=0) and by wafer_download_array; everything else forms a, b from V in registers.  They
// are allocated and filled on first use and kept in step with V from then on.
int ensure_ab(wafer_ctx *c)
{
    if (c->a && c->b) return WAFER_OK;
    TRY(alloc_grid_array(c, &c->a, c->s_main));
    TRY(alloc_grid_array(c, &c->b, c->s_main));
    if (c->have_pot) {
        const dim3 grid(c->bx, c->by, c->g.lz), block(64, 4);
        if (c->f32)
            hipLaunchKernelGGL((wafer_k_ab<float>), grid, block, 0, c->s_main, c->g, c->P.dt, as<float>(c->v), as<float>(c->a), as<float>(c->b));
        else
            hipLaunchKernelGGL((wafer_k_ab<double>), grid, block, 0, c->s_main, c->g, c->P.dt, as<double>(c->v), as<double>(c->a), as<double>(c->b));
        HIP_TRY(hipGetLastError());
    }
    return WAFER_OK;
}
// V changed: bring a, b (if they exist) back in step
int refresh_ab(wafer_ctx *c)
{
    if (!c->a || !c->b) return WAFER_OK;
    const dim3 grid(c->bx, c->by, c->g.lz), block(64, 4);
    if (c->f32)
        hipLaunchKernelGGL((wafer_k_ab<float>), grid, block, 0, c->s_main, c->g, c->P.dt, as<float>(c->v), as<float>(c->a), as<float>(c->b));
    else
        hipLaunchKernelGGL((wafer_k_ab<double>), grid, block, 0, c->s_main, c->g, c->P.dt, as<double>(c->v), as<double>(c->a), as<double>(c->b));
    HIP_TRY(hipGetLastError());
    return WAFER_OK;
}
// after V changed: may the kernels that form a, b from V use the short reciprocal?
int check_v_range(wafer_ctx *c)
{
    unsigned long long *d = reinterpret_cast<unsigned long long *>(c->scal + 16);
    unsigned long long init[2] = {~0ull, 0ull}, got[2];
    HIP_TRY(hipMemcpyAsync(d, init, sizeof init, hipMemcpyHostToDevice, c->s_main));
    if (c->f32)
        hipLaunchKernelGGL((wafer_k_v_range<float>), dim3(c->num_cus * 4), dim3(256), 0, c->s_main, as<float>(alloc_base(c, c->v)), c->g.total, c->P.dt, d);
    else
        hipLaunchKernelGGL((wafer_k_v_range<double>), dim3(c->num_cus * 4), dim3(256), 0, c->s_main, as<double>(alloc_base(c, c->v)), c->g.total, c->P.dt, d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(got, d, sizeof got, hipMemcpyDeviceToHost, c->s_main));
    HIP_TRY(hipStreamSynchronize(c->s_main));
    double lo, hi;
    memcpy(&lo, &got[0], 8);
    memcpy(&hi, &got[1], 8);
    c->v_in_range = (lo > 0x1p-400) && (hi < 0x1p400); // a NaN anywhere makes hi a NaN: false
    for (int &a : c->x2_agreed) a = -1;
    return WAFER_OK;
}

} // namespace wafer_eng

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

int wafer_abi_version(void) { return WAFER_ABI_VERSION; }
const char *wafer_last_error(void) { return g_last_error.c_str(); }

int wafer_ctx_create(const wafer_params *p, wafer_ctx **out)
{
    if (!p || !out) return fail(WAFER_ERR_INVALID, "null argument");
    if (p->struct_size != sizeof(wafer_params))
        return fail(WAFER_ERR_INVALID, "wafer_params.struct_size %u != %zu (ABI mismatch)",
                    p->struct_size, sizeof(wafer_params));
    if (p->nx < 1 || p->ny < 1 || p->nz < 1) return fail(WAFER_ERR_INVALID, "grid size must be >= 1");
    if (p->central_difference < 1 || p->central_difference > 3)
        return fail(WAFER_ERR_INVALID, "central_difference must be 1 (Three), 2 (Five) or 3 (SevenPoint)");
    if (p->dtype != WAFER_F64 && p->dtype != WAFER_F32 && p->dtype != WAFER_F32_FAST) return fail(WAFER_ERR_INVALID, "bad dtype");
    if (!(p->dn > 0) || !(p->dt > 0) || !(p->mass > 0)) return fail(WAFER_ERR_INVALID, "dn, dt, mass must be > 0");
    const double stencil_den = wafer_stencil_den(p->central_difference, p->dn, p->mass);
    if (!std::isnormal(stencil_den) || !std::isnormal(1.0 / stencil_den))
        return fail(WAFER_ERR_INVALID, "dn^2 * mass = %g is outside the range of normal doubles (or its reciprocal is)", p->dn * p->dn * p->mass);
    // config.rs:362-365 (ErrorKind::LargeDt)
    if (!(p->flags & WAFER_FLAG_SKIP_DT_CHECK) && p->dt > p->dn * p->dn / 3.)
        return fail(WAFER_ERR_INVALID, "LargeDt: dt must be <= dn^2/3 (config.rs:363)");
    const int R = p->central_difference;
    const uint32_t zc = p->z_count ? p->z_count : p->nz;
    const uint32_t zb = p->z_count ? p->z_begin : 0;
    if (zb + zc > p->nz) return fail(WAFER_ERR_INVALID, "z-slab exceeds the grid");
    const int G = p->halo_depth ? (int)p->halo_depth : R;
    if (G < R) return fail(WAFER_ERR_INVALID, "halo_depth must be >= ext");
    if (zc < p->nz && (int)zc < 2 * R) return fail(WAFER_ERR_INVALID, "a z-slab needs at least 2*ext planes");
    if (zc < p->nz && (int)zc < G)
        return fail(WAFER_ERR_INVALID, "halo_depth %d is deeper than this slab's %u planes: a rank sends its own planes only", G, zc);

    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(WAFER_ERR_HIP, "no HIP device visible: the engine has no CPU path");
    if (p->device < 0 || p->device >= ndev) return fail(WAFER_ERR_INVALID, "device %d out of range", p->device);
    HIP_TRY(hipSetDevice(p->device));
    int cus = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, p->device));

    wafer_ctx *c = new wafer_ctx();
    c->num_cus = cus > 0 ? cus : 256;
    c->P = *p;
    c->f32 = (p->dtype == WAFER_F32 || p->dtype == WAFER_F32_FAST);
    c->f32_arith = (p->dtype == WAFER_F32_FAST);
    c->esz = c->f32 ? 4 : 8;
    c->tune = wafer_tuning_from_env(); // the only place the WAFER_* tuning variables are read
    c->g = wafer_make_geom((int)p->nx, (int)p->ny, (int)p->nz, R, G, (int)zb, (int)zc, (int)c->esz);
    c->bx = (c->g.px + 63) / 64; // covers both the work area and the padded extent
    c->by = (c->g.py + 3) / 4;
    c->div_plan = wafer_divplan_make(stencil_den);
    if (c->f32_arith) c->div_plan_f = wafer_divplan_make_f32((float)stencil_den);
    if (p->flags & WAFER_FLAG_UNPLANNED_DIV) c->div_plan.checked = c->div_plan_f.checked = 0;
    c->overlap_mode = c->sched = (c->tune.overlap >= 0 && c->tune.overlap <= 2) ? c->tune.overlap : 2; // the modes of wafer_set_overlap
    // fused passes per halo exchange: 1 unless the host asks for deep halos (wafer_set_halo_cycle) -- a
    // concentrated exchange outlasts the interior launch it hides behind on anything but a very fast link
    c->halo_cycle = 1;

    auto cleanup_fail = [&](int rc) {
        wafer_ctx_destroy(c);
        return rc;
    };
#define HIP_TRYC(expr)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return cleanup_fail(fail(WAFER_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_))); \
    } while (0)

    HIP_TRYC(hipStreamCreateWithFlags(&c->s_own, hipStreamNonBlocking));
    c->s_main = c->s_own;
    { // boundary planes and their exchange go ahead of the interior update: highest priority
        int prio_lo = 0, prio_hi = 0;
        HIP_TRYC(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        HIP_TRYC(hipStreamCreateWithPriority(&c->s_aux, hipStreamNonBlocking, (c->tune.hv_debug & 64) ? prio_lo : prio_hi));
        HIP_TRYC(hipStreamCreateWithPriority(&c->s_aux2, hipStreamNonBlocking, (c->tune.hv_debug & 64) ? prio_lo : prio_hi));
    }
    HIP_TRYC(hipEventCreate(&c->ev_start));
    HIP_TRYC(hipEventCreate(&c->ev_stop));
    HIP_TRYC(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIP_TRYC(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    HIP_TRYC(hipEventCreateWithFlags(&c->ev_bdry, hipEventDisableTiming));
    for (int a_ = 0; a_ < 2; ++a_) HIP_TRYC(hipEventCreateWithFlags(&c->ev_ex[a_], hipEventDisableTiming));

    // a and b are allocated on first use (ensure_ab): the default kernels form them from V in registers
    void **arrays[] = {&c->phi[0], &c->phi[1], &c->v};
    for (void **arr : arrays)
        if (alloc_grid_array(c, arr, c->s_main) != WAFER_OK) return cleanup_fail(WAFER_ERR_HIP);
    c->partials_stride = std::max<size_t>((size_t)c->bx * c->by * 64 + 1024, (size_t)c->num_cus * 8); // column kernels / row kernels
    // rows: 1 + WAFER_MAX_LOW sums of a one-step excited kernel, 2 + 3k (k <= 3) of a two-step pass
    HIP_TRYC(hipMalloc((void **)&c->partials, sizeof(double) * (2 + 3 * WAFER_MAX_LOW) * c->partials_stride));
    HIP_TRYC(hipMalloc((void **)&c->x2mat, sizeof(double) * 2 * WAFER_MAX_LOW * WAFER_MAX_LOW));
    HIP_TRYC(hipMalloc((void **)&c->x2coef, sizeof(double) * 32));
    HIP_TRYC(hipMalloc((void **)&c->gram, sizeof(double) * WAFER_MAX_LOW * WAFER_MAX_LOW));
    HIP_TRYC(hipMemsetAsync(c->gram, 0, sizeof(double) * WAFER_MAX_LOW * WAFER_MAX_LOW, c->s_main));
    HIP_TRYC(hipMalloc((void **)&c->scal, sizeof(double) * SCAL_SLOTS));
    HIP_TRYC(hipMemsetAsync(c->scal, 0, sizeof(double) * SCAL_SLOTS, c->s_main));
    HIP_TRYC(hipHostMalloc((void **)&c->scal_host, sizeof(double) * SCAL_SLOTS, hipHostMallocDefault));
    HIP_TRYC(hipStreamSynchronize(c->s_main));
#undef HIP_TRYC
    c->kernel_name = variant_name(default_variant(c));
    *out = c;
    return WAFER_OK;
}

int wafer_ctx_destroy(wafer_ctx *c)
{
    if (!c) return WAFER_OK;
    (void)hipSetDevice(c->P.device);
    if (c->s_own) (void)hipStreamSynchronize(c->s_own);
    if (c->s_aux) (void)hipStreamSynchronize(c->s_aux);
    if (c->s_aux2) (void)hipStreamSynchronize(c->s_aux2);
    for (void *p : {c->phi[0], c->phi[1], c->v, c->a, c->b, c->potsub})
        if (p) (void)hipFree(alloc_base(c, p));
    for (void *p : c->states)
        if (p) (void)hipFree(alloc_base(c, p));
    for (void *p : c->mstates)
        if (p) (void)hipFree(alloc_base(c, p));
    if (c->x2mat) (void)hipFree(c->x2mat);
    if (c->x2coef) (void)hipFree(c->x2coef);
    if (c->partials) (void)hipFree(c->partials);
    if (c->scal) (void)hipFree(c->scal);
    if (c->gram) (void)hipFree(c->gram);
    if (c->scal_host) (void)hipHostFree(c->scal_host);
    for (hipEvent_t e : {c->ev_start, c->ev_stop, c->ev_fork, c->ev_join, c->ev_bdry, c->ev_ex[0], c->ev_ex[1]})
        if (e) (void)hipEventDestroy(e);
    for (auto &t : c->f3_tables) (void)hipFree(t.dev);
    (void)wafer_peer_disconnect(c);
    if (c->peer_flags) (void)hipFree(c->peer_flags);
    if (c->peer_dev) (void)hipFree(c->peer_dev);
    if (c->hv_words) (void)hipFree(c->hv_words);
    if (c->hv_err) (void)hipHostFree(c->hv_err);
    if (c->s_own) (void)hipStreamDestroy(c->s_own);
    if (c->s_aux) (void)hipStreamDestroy(c->s_aux);
    if (c->s_aux2) (void)hipStreamDestroy(c->s_aux2);
    delete c;
    return WAFER_OK;
}

int wafer_synchronize(wafer_ctx *c)
{
    if (!c) return fail(WAFER_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->P.device));
    HIP_TRY(hipStreamSynchronize(c->s_aux));
    HIP_TRY(hipStreamSynchronize(c->s_aux2));
    HIP_TRY(hipStreamSynchronize(c->s_main));
    return check_hv_err(c);
}

} // extern "C"

// ---- layout conversion helpers ----------------------------------------------
// Copies the z-range of a global reference-layout host array [sx][sy][szg] that
// this slab holds into a dense staging buffer and transposes it into `dev`.
// (xp0, yp0, zofs): where host element (0,0,0) lands in padded coordinates.
template <bool TO_DEVICE>
static int convert_host_array(wafer_ctx *c, double *host, int sx, int sy, int szg, int xp0, int yp0,
                              int zp0, void *dev)
{
    const WaferGeom &g = c->g;
    // host z index hz corresponds to global padded zp = zp0 + hz; local plane lzp = zp - zp_of(0)
    const int lz_first = g.zp_of(0);
    int hz_lo = lz_first - zp0, hz_hi = lz_first + g.lz - zp0;
    if (hz_lo < 0) hz_lo = 0;
    if (hz_hi > szg) hz_hi = szg;
    const int sz = hz_hi - hz_lo;
    if (sz <= 0) return WAFER_OK;
    const int lzp0 = zp0 + hz_lo - lz_first;
    double *stage = nullptr;
    const size_t rows = (size_t)sx * sy;
    HIP_TRY(hipMalloc((void **)&stage, rows * sz * sizeof(double)));
    WaferXposeArgs a;
    a.g = g;
    a.sx = sx; a.sy = sy; a.sz = sz;
    a.xp0 = xp0; a.yp0 = yp0; a.lzp0 = lzp0;
    const dim3 grid((sx + 31) / 32, sy, (sz + 31) / 32), block(32, 8);
    hipError_t e = hipSuccess;
    // A slab takes sz of the szg values of every host row.  The runtime's pitched copy from / to pageable
    // memory touches the whole span of the host array (measured: 9.4 GB resident for 128 of 1026 planes
    // of an 8 GB memory-mapped file), so slabs gather / scatter their z-range through two pinned chunks
    // instead: a rank's host footprint is its own planes, whatever the size of the global array.
    const bool strided = sz < szg;
    double *pin[2] = {nullptr, nullptr};
    hipEvent_t pev[2] = {nullptr, nullptr};
    const size_t chunk_rows = std::max<size_t>(1, ((size_t)32 << 20) / ((size_t)sz * 8));
    if (strided) {
        for (int b = 0; b < 2 && e == hipSuccess; ++b) {
            e = hipHostMalloc((void **)&pin[b], chunk_rows * sz * sizeof(double), hipHostMallocDefault);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&pev[b], hipEventDisableTiming);
        }
    }
    auto release_pins = [&]() {
        for (int b = 0; b < 2; ++b) {
            if (pin[b]) (void)hipHostFree(pin[b]);
            if (pev[b]) (void)hipEventDestroy(pev[b]);
        }
    };
    if (TO_DEVICE && strided) {
        size_t i = 0;
        for (size_t r0 = 0; r0 < rows && e == hipSuccess; r0 += chunk_rows, ++i) {
            const int b = (int)(i & 1);
            if (i >= 2) e = hipEventSynchronize(pev[b]);
            const size_t n = std::min(chunk_rows, rows - r0);
            for (size_t r = 0; r < n; ++r)
                memcpy(pin[b] + r * sz, host + (r0 + r) * (size_t)szg + hz_lo, (size_t)sz * 8);
            if (e == hipSuccess) e = hipMemcpyAsync(stage + r0 * sz, pin[b], n * sz * sizeof(double), hipMemcpyHostToDevice, c->s_main);
            if (e == hipSuccess) e = hipEventRecord(pev[b], c->s_main);
        }
    } else if (TO_DEVICE) {
        e = hipMemcpy2DAsync(stage, (size_t)sz * 8, host + hz_lo, (size_t)szg * 8, (size_t)sz * 8, rows,
                             hipMemcpyHostToDevice, c->s_main);
    }
    if (TO_DEVICE) {
        if (e == hipSuccess) {
            if (c->f32)
                hipLaunchKernelGGL((wafer_k_transpose<float, true>), grid, block, 0, c->s_main, a, stage, as<float>(dev));
            else
                hipLaunchKernelGGL((wafer_k_transpose<double, true>), grid, block, 0, c->s_main, a, stage, as<double>(dev));
            e = hipGetLastError();
        }
    } else {
        if (c->f32)
            hipLaunchKernelGGL((wafer_k_transpose<float, false>), grid, block, 0, c->s_main, a, stage, as<float>(dev));
        else
            hipLaunchKernelGGL((wafer_k_transpose<double, false>), grid, block, 0, c->s_main, a, stage, as<double>(dev));
        e = hipGetLastError();
        if (e == hipSuccess && strided) {
            for (size_t r0 = 0; r0 < rows && e == hipSuccess; r0 += chunk_rows) {
                const size_t n = std::min(chunk_rows, rows - r0);
                e = hipMemcpyAsync(pin[0], stage + r0 * sz, n * sz * sizeof(double), hipMemcpyDeviceToHost, c->s_main);
                if (e == hipSuccess) e = hipStreamSynchronize(c->s_main);
                for (size_t r = 0; r < n && e == hipSuccess; ++r)
                    memcpy(host + (r0 + r) * (size_t)szg + hz_lo, pin[0] + r * sz, (size_t)sz * 8);
            }
        } else if (e == hipSuccess) {
            e = hipMemcpy2DAsync(host + hz_lo, (size_t)szg * 8, stage, (size_t)sz * 8, (size_t)sz * 8, rows,
                                 hipMemcpyDeviceToHost, c->s_main);
        }
    }
    hipError_t e2 = hipStreamSynchronize(c->s_main);
    release_pins();
    (void)hipFree(stage);
    if (e != hipSuccess || e2 != hipSuccess)
        return fail(WAFER_ERR_HIP, "layout conversion failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return WAFER_OK;
}

static int upload_padded(wafer_ctx *c, const double *host, void *dev)
{
    return convert_host_array<true>(c, const_cast<double *>(host), c->g.px, c->g.py, c->g.pzg, 0, 0, 0, dev);
}
static int download_padded(wafer_ctx *c, double *host, void *dev)
{
    return convert_host_array<false>(c, host, c->g.px, c->g.py, c->g.pzg, 0, 0, 0, dev);
}

// ---- potentials ----------------------------------------------------------------
extern "C" {

static WaferPotArgs pot_args(wafer_ctx *c, int type)
{
    WaferPotArgs a;
    a.g = c->g;
    a.type = type;
    a.dn = c->P.dn; a.dt = c->P.dt; a.mass = c->P.mass; a.sig = c->P.sig;
    const double t = 1.0, xi = 0.0; // potential.rs:252-253
    a.mu_t = host_mu(t);
    a.alphas_2pit = host_alphas(2. * WAFER_PI * t);
    a.xi_coef = 0.07 * std::pow(xi, 0.2);
    a.xi_fac = std::pow(1. + xi, -0.29);
    return a;
}

static int ensure_potsub_array(wafer_ctx *c)
{
    if (!c->potsub) {
        TRY(alloc_grid_array(c, &c->potsub, c->s_main));
    }
    return WAFER_OK;
}

int wafer_set_potential_builtin(wafer_ctx *c, int potential)
{
    if (!c) return fail(WAFER_ERR_INVALID, "null context");
    if (potential < 0 || potential > WAFER_POT_FROMSCRIPT) return fail(WAFER_ERR_INVALID, "unknown potential %d", potential);
    if (potential == WAFER_POT_FROMFILE || potential == WAFER_POT_FROMSCRIPT)
        return fail(WAFER_ERR_NOT_AVAILABLE, "PotentialNotAvailable: FromFile/FromScript need wafer_set_potential_host");
    HIP_TRY(hipSetDevice(c->P.device));
    WaferPotArgs a = pot_args(c, potential);
    const dim3 grid(c->bx, c->by, c->g.lz), block(64, 4);
    if (c->f32)
        hipLaunchKernelGGL((wafer_k_potential<float>), grid, block, 0, c->s_main, a, as<float>(c->v), as<float>(c->a), as<float>(c->b));
    else
        hipLaunchKernelGGL((wafer_k_potential<double>), grid, block, 0, c->s_main, a, as<double>(c->v), as<double>(c->a), as<double>(c->b));
    HIP_TRY(hipGetLastError());
    // pot_sub: potential.rs:134-153 with 326-363
    c->potsub_kind = WAFER_POTSUB_NONE;
    c->potsub_scalar = 0.0;
    if (potential == WAFER_POT_FULLCORNELL) {
        TRY(ensure_potsub_array(c));
        const dim3 g2(c->bx, c->by, c->g.nzl);
        if (c->f32)
            hipLaunchKernelGGL((wafer_k_potsub_fullcornell<float>), g2, block, 0, c->s_main, a, as<float>(c->potsub));
        else
            hipLaunchKernelGGL((wafer_k_potsub_fullcornell<double>), g2, block, 0, c->s_main, a, as<double>(c->potsub));
        HIP_TRY(hipGetLastError());
        c->potsub_kind = WAFER_POTSUB_ARRAY;
    } else {
        double s = 0.0;
        if (potential == WAFER_POT_ELIPTICALCOULOMB) s = 1. / c->P.dn;   // potential.rs:359
        if (potential == WAFER_POT_SIMPLECORNELL) s = 4.0 * c->P.mass;   // potential.rs:360
        if (s > 0.0) { // potential.rs:148-152
            c->potsub_kind = WAFER_POTSUB_SCALAR;
            c->potsub_scalar = s;
        }
    }
    c->have_pot = true;
    c->x2_ready = 0;   // M_j = A l_j follows V
    // the excited-state kernels can evaluate these instead of reading V (wafer_k_step_lds, VG)
    c->vgen_type = (potential == WAFER_POT_COULOMB || potential == WAFER_POT_COMPLEXCOULOMB) ? WAFER_POT_COULOMB
                   : (potential == WAFER_POT_HARMONIC || potential == WAFER_POT_COMPLEXHARMONIC) ? WAFER_POT_HARMONIC
                   : (potential == WAFER_POT_SIMPLECORNELL) ? WAFER_POT_SIMPLECORNELL : 0;
    return check_v_range(c);
}

int wafer_set_potential_host(wafer_ctx *c, const double *v, int potsub_kind, double potsub_scalar, const double *potsub)
{
    if (!c || !v) return fail(WAFER_ERR_INVALID, "null argument");
    if (potsub_kind < 0 || potsub_kind > 2) return fail(WAFER_ERR_INVALID, "bad potsub_kind");
    if (potsub_kind == WAFER_POTSUB_ARRAY && !potsub) return fail(WAFER_ERR_INVALID, "potsub array missing");
    HIP_TRY(hipSetDevice(c->P.device));
    TRY(upload_padded(c, v, c->v));
    TRY(refresh_ab(c));
    c->vgen_type = 0;
    c->potsub_kind = potsub_kind;
    c->potsub_scalar = (potsub_kind == WAFER_POTSUB_SCALAR) ? potsub_scalar : 0.0;
    if (potsub_kind == WAFER_POTSUB_ARRAY) {
        TRY(ensure_potsub_array(c));
        // unpadded [nx][ny][nz] -> work cells (offset R on every axis)
        TRY((convert_host_array<true>(c, const_cast<double *>(potsub), c->g.nx, c->g.ny, c->g.nz, c->g.R, c->g.R, c->g.R, c->potsub)));
    }
    c->have_pot = true;
    c->x2_ready = 0;   // M_j = A l_j follows V
    return check_v_range(c);
}

int wafer_download_array(wafer_ctx *c, int id, double *out)
{
    if (!c || !out) return fail(WAFER_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->P.device));
    switch (id) {
    case WAFER_ARRAY_V: return download_padded(c, out, c->v);
    case WAFER_ARRAY_A: TRY(ensure_ab(c)); return download_padded(c, out, c->a);
    case WAFER_ARRAY_B: TRY(ensure_ab(c)); return download_padded(c, out, c->b);
    case WAFER_ARRAY_POTSUB:
        if (c->potsub_kind != WAFER_POTSUB_ARRAY) return fail(WAFER_ERR_STATE, "pot_sub is not an array");
        return convert_host_array<false>(c, out, c->g.nx, c->g.ny, c->g.nz, c->g.R, c->g.R, c->g.R, c->potsub);
    default: return fail(WAFER_ERR_INVALID, "unknown array id %d", id);
    }
}

int wafer_set_potsub(wafer_ctx *c, int kind, double scalar, const double *potsub)
{
    if (!c) return fail(WAFER_ERR_INVALID, "null context");
    if (!c->have_pot) return fail(WAFER_ERR_STATE, "set the potential first");
    if (kind < 0 || kind > 2) return fail(WAFER_ERR_INVALID, "bad potsub kind");
    if (kind == WAFER_POTSUB_ARRAY && !potsub) return fail(WAFER_ERR_INVALID, "potsub array missing");
    HIP_TRY(hipSetDevice(c->P.device));
    c->potsub_kind = kind;
    c->potsub_scalar = (kind == WAFER_POTSUB_SCALAR) ? scalar : 0.0;
    if (kind == WAFER_POTSUB_ARRAY) {
        TRY(ensure_potsub_array(c));
        TRY((convert_host_array<true>(c, const_cast<double *>(potsub), c->g.nx, c->g.ny, c->g.nz, c->g.R, c->g.R, c->g.R, c->potsub)));
    }
    return WAFER_OK;
}

int wafer_get_potsub(wafer_ctx *c, int *kind, double *scalar)
{
    if (!c || !kind || !scalar) return fail(WAFER_ERR_INVALID, "null argument");
    *kind = c->potsub_kind;
    *scalar = c->potsub_scalar;
    return WAFER_OK;
}

// ---- phi ---------------------------------------------------------------------------
int wafer_set_initial_condition(wafer_ctx *c, int ic, uint64_t seed)
{
    if (!c) return fail(WAFER_ERR_INVALID, "null context");
    if (ic == WAFER_IC_FROMFILE) return fail(WAFER_ERR_NOT_AVAILABLE, "FromFile: use wafer_upload_phi");
    if (ic < WAFER_IC_GAUSSIAN || ic > WAFER_IC_BOOLEAN) return fail(WAFER_ERR_INVALID, "unknown initial condition %d", ic);
    HIP_TRY(hipSetDevice(c->P.device));
    WaferIcArgs a;
    a.g = c->g; a.ic = ic; a.seed = seed;
    a.dn = c->P.dn; a.mass = c->P.mass; a.sig = c->P.sig;
    const dim3 grid(c->bx, c->by, c->g.lz), block(64, 4);
    if (c->f32)
        hipLaunchKernelGGL((wafer_k_initial_condition<float>), grid, block, 0, c->s_main, a, as<float>(c->phi[c->cur]));
    else
        hipLaunchKernelGGL((wafer_k_initial_condition<double>), grid, block, 0, c->s_main, a, as<double>(c->phi[c->cur]));
    HIP_TRY(hipGetLastError());
    c->have_phi = true;
    c->halo_valid = c->g.G; // every ghost plane was generated from global indices
    return WAFER_OK;
}

int wafer_upload_phi(wafer_ctx *c, const double *phi)
{
    if (!c || !phi) return fail(WAFER_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->P.device));
    TRY(upload_padded(c, phi, c->phi[c->cur]));
    c->have_phi = true;
    c->halo_valid = c->g.G; // ghost planes came from the global array
    return WAFER_OK;
}

// config::symmetrise_wavefunction (config.rs:691-728)
int wafer_symmetrise(wafer_ctx *c, int constraint)
{
    if (!c) return fail(WAFER_ERR_INVALID, "null context");
    if (constraint < WAFER_SYM_NOT_CONSTRAINED || constraint > WAFER_SYM_ANTISYM_ABOUT_Y)
        return fail(WAFER_ERR_INVALID, "unknown symmetry constraint %d", constraint);
    if (!c->have_phi) return fail(WAFER_ERR_STATE, "phi not set");
    if (constraint == WAFER_SYM_NOT_CONSTRAINED) return WAFER_OK;
    if (c->g.R != 3)
        return fail(WAFER_ERR_INVALID, "symmetry constraints index the SevenPoint frame (config.rs:702-725); "
                                       "the reference runs out of bounds with central_difference ext %d", c->g.R);
    const int axis = (constraint == WAFER_SYM_ABOUT_Z || constraint == WAFER_SYM_ANTISYM_ABOUT_Z) ? 0 : 1;
    if (axis == 0 && c->g.nzl != c->g.nz)
        return fail(WAFER_ERR_NOT_AVAILABLE, "a mirror about z crosses z-slabs");
    const double sign = (constraint == WAFER_SYM_ANTISYM_ABOUT_Z || constraint == WAFER_SYM_ANTISYM_ABOUT_Y) ? -1.0 : 1.0;
    HIP_TRY(hipSetDevice(c->P.device));
    const int src = c->cur, dst = c->cur ^ 1;
    // the other buffer is scratch between steps: start from zeros so that the frame is the frame
    HIP_TRY(hipMemsetAsync(alloc_base(c, c->phi[dst]), 0, (size_t)c->g.total * c->esz, c->s_main));
    const dim3 grid(c->bx, c->by, c->g.lz), block(64, 4);
    if (c->f32)
        hipLaunchKernelGGL((wafer_k_symmetrise<float>), grid, block, 0, c->s_main, c->g, axis, sign, as<float>(c->phi[src]), as<float>(c->phi[dst]));
    else
        hipLaunchKernelGGL((wafer_k_symmetrise<double>), grid, block, 0, c->s_main, c->g, axis, sign, as<double>(c->phi[src]), as<double>(c->phi[dst]));
    HIP_TRY(hipGetLastError());
    c->cur = dst;
    c->halo_valid = 0;
    return WAFER_OK;
}

// fill_data / read_csv's resampling branch (input.rs:149-176, 640-656, 667-716): `src` is an
// UNPADDED array of another resolution; the work area is filled by trilinear interpolation with
// the reference's basis (the padded target size), the frame is zero.
static int resample_into(wafer_ctx *c, const double *src, uint32_t sx, uint32_t sy, uint32_t sz,
                         const uint32_t *basis, void *dst)
{
    if (sx < 2 || sy < 2 || sz < 2) return fail(WAFER_ERR_INVALID, "resampling needs at least 2 points per axis");
    double *dsrc = nullptr;
    const size_t n = (size_t)sx * sy * sz;
    HIP_TRY(hipMalloc((void **)&dsrc, n * sizeof(double)));
    hipError_t e = hipMemcpyAsync(dsrc, src, n * sizeof(double), hipMemcpyHostToDevice, c->s_main);
    if (e == hipSuccess) e = hipMemsetAsync(alloc_base(c, dst), 0, (size_t)c->g.total * c->esz, c->s_main);
    if (e == hipSuccess) {
        WaferResampleArgs a;
        a.g = c->g;
        a.sx = (int)sx; a.sy = (int)sy; a.sz = (int)sz;
        a.bx = basis ? (int)basis[0] : c->g.px;
        a.by = basis ? (int)basis[1] : c->g.py;
        a.bz = basis ? (int)basis[2] : c->g.pzg;
        const dim3 grid(c->bx, c->by, c->g.lz), block(64, 4);
        if (c->f32) hipLaunchKernelGGL((wafer_k_trilerp<float>), grid, block, 0, c->s_main, a, dsrc, as<float>(dst));
        else hipLaunchKernelGGL((wafer_k_trilerp<double>), grid, block, 0, c->s_main, a, dsrc, as<double>(dst));
        e = hipGetLastError();
    }
    hipError_t e2 = hipStreamSynchronize(c->s_main);
    (void)hipFree(dsrc);
    if (e != hipSuccess || e2 != hipSuccess)
        return fail(WAFER_ERR_HIP, "resample failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return WAFER_OK;
}

int wafer_upload_phi_resampled(wafer_ctx *c, const double *src, uint32_t sx, uint32_t sy, uint32_t sz,
                               const uint32_t *basis)
{
    if (!c || !src) return fail(WAFER_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->P.device));
    TRY(resample_into(c, src, sx, sy, sz, basis, c->phi[c->cur]));
    c->have_phi = true;
    c->halo_valid = c->g.G; // ghost planes were interpolated from the same source
    return WAFER_OK;
}

int wafer_set_potential_resampled(wafer_ctx *c, const double *src, uint32_t sx, uint32_t sy, uint32_t sz,
                                  const uint32_t *basis)
{
    if (!c || !src) return fail(WAFER_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->P.device));
    TRY(resample_into(c, src, sx, sy, sz, basis, c->v));
    TRY(refresh_ab(c));
    c->vgen_type = 0;
    c->potsub_kind = WAFER_POTSUB_NONE; // potential.rs:357-358: FromFile has no pot_sub of its own
    c->potsub_scalar = 0.0;
    c->have_pot = true;
    c->x2_ready = 0;   // M_j = A l_j follows V
    return check_v_range(c);
}

int wafer_set_potsub_resampled(wafer_ctx *c, const double *src, uint32_t sx, uint32_t sy, uint32_t sz)
{
    if (!c || !src) return fail(WAFER_ERR_INVALID, "null argument");
    if (!c->have_pot) return fail(WAFER_ERR_STATE, "set the potential first");
    HIP_TRY(hipSetDevice(c->P.device));
    TRY(ensure_potsub_array(c));
    const uint32_t basis[3] = {(uint32_t)c->g.nx, (uint32_t)c->g.ny, (uint32_t)c->g.nz}; // input.rs:472: target_size
    TRY(resample_into(c, src, sx, sy, sz, basis, c->potsub));
    c->potsub_kind = WAFER_POTSUB_ARRAY;
    c->potsub_scalar = 0.0;
    return WAFER_OK;
}

int wafer_download_phi(wafer_ctx *c, double *phi)
{
    if (!c || !phi) return fail(WAFER_ERR_INVALID, "null argument");
    if (!c->have_phi) return fail(WAFER_ERR_STATE, "phi not set");
    HIP_TRY(hipSetDevice(c->P.device));
    HIP_TRY(hipStreamSynchronize(c->s_aux));
    TRY(download_padded(c, phi, c->phi[c->cur]));
    return check_hv_err(c);   // a wait of a decomposed pass that gave up has poisoned what it stored: say so with the data
}

int wafer_download_phi_owned(wafer_ctx *c, double *out)
{
    if (!c || !out) return fail(WAFER_ERR_INVALID, "null argument");
    if (!c->have_phi) return fail(WAFER_ERR_STATE, "phi not set");
    HIP_TRY(hipSetDevice(c->P.device));
    HIP_TRY(hipStreamSynchronize(c->s_aux));
    // host element (0, 0, 0) = work cell (0, 0, z_begin): padded coordinates (R, R, z_begin + R)
    TRY((convert_host_array<false>(c, out, c->g.nx, c->g.ny, c->g.nzl, c->g.R, c->g.R, c->g.z_begin + c->g.R, c->phi[c->cur])));
    return check_hv_err(c);
}

int wafer_diag_download_window(wafer_ctx *c, int id, uint32_t zp_begin, uint32_t zp_count, double *out)
{
    if (!c || !out) return fail(WAFER_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->P.device));
    const long long lo = c->g.zp_of(0), hi = c->g.zp_of(c->g.lz);   // global padded planes this context holds: [lo, hi)
    if (zp_count == 0 || (long long)zp_begin < std::max(lo, 0LL) || (long long)zp_begin + zp_count > std::min<long long>(hi, c->g.pzg))
        return fail(WAFER_ERR_INVALID, "planes [%u, %u) are not among the padded planes [%lld, %lld) this context holds", zp_begin,
                    zp_begin + zp_count, std::max(lo, 0LL), std::min<long long>(hi, c->g.pzg));
    void *dev = nullptr;
    switch (id) {
    case WAFER_ARRAY_V: dev = c->v; break;
    case WAFER_ARRAY_A: TRY(ensure_ab(c)); dev = c->a; break;
    case WAFER_ARRAY_B: TRY(ensure_ab(c)); dev = c->b; break;
    case WAFER_ARRAY_PHI:
        if (!c->have_phi) return fail(WAFER_ERR_STATE, "phi not set");
        HIP_TRY(hipStreamSynchronize(c->s_aux));
        dev = c->phi[c->cur];
        break;
    default: return fail(WAFER_ERR_INVALID, "unknown array id %d", id);
    }
    // host element (0, 0, hz) = padded cell (0, 0, zp_begin + hz)
    TRY((convert_host_array<false>(c, out, c->g.px, c->g.py, (int)zp_count, 0, 0, (int)zp_begin, dev)));
    return check_hv_err(c);
}

// ---- w_store ------------------------------------------------------------------------
static int new_state_slot(wafer_ctx *c, void **slot)
{
    if (c->states.size() >= c->P.max_states)
        return fail(WAFER_ERR_STATE, "w_store is full (max_states = %u)", c->P.max_states);
    return alloc_grid_array(c, slot, c->s_main);
}

int wafer_push_state(wafer_ctx *c)
{
    if (!c) return fail(WAFER_ERR_INVALID, "null context");
    if (!c->have_phi) return fail(WAFER_ERR_STATE, "phi not set");
    HIP_TRY(hipSetDevice(c->P.device));
    TRY(ensure_halo(c, c->g.R)); // the one-pass excited step reads stored states on ghost planes
    void *slot = nullptr;
    TRY(new_state_slot(c, &slot));
    HIP_TRY(hipMemcpyAsync(alloc_base(c, slot), alloc_base(c, c->phi[c->cur]), (size_t)c->g.total * c->esz, hipMemcpyDeviceToDevice, c->s_main));
    c->states.push_back(slot);
    return recompute_gram(c);
}

int wafer_load_state(wafer_ctx *c, uint32_t idx, const double *state)
{
    if (!c || !state) return fail(WAFER_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->P.device));
    if (idx > c->states.size()) return fail(WAFER_ERR_STATE, "states must be loaded in order");
    if (idx == c->states.size()) {
        void *slot = nullptr;
        TRY(new_state_slot(c, &slot));
        c->states.push_back(slot);
    }
    TRY(upload_padded(c, state, c->states[idx]));
    return recompute_gram(c);
}

int wafer_download_state(wafer_ctx *c, uint32_t idx, double *state)
{
    if (!c || !state) return fail(WAFER_ERR_INVALID, "null argument");
    if (idx >= c->states.size()) return fail(WAFER_ERR_STATE, "no state %u", idx);
    HIP_TRY(hipSetDevice(c->P.device));
    return download_padded(c, state, c->states[idx]);
}

int wafer_clone_state_to_phi(wafer_ctx *c, uint32_t idx)
{
    if (!c) return fail(WAFER_ERR_INVALID, "null context");
    if (idx >= c->states.size()) return fail(WAFER_ERR_STATE, "no state %u", idx);
    HIP_TRY(hipSetDevice(c->P.device));
    HIP_TRY(hipMemcpyAsync(alloc_base(c, c->phi[c->cur]), alloc_base(c, c->states[idx]), (size_t)c->g.total * c->esz, hipMemcpyDeviceToDevice, c->s_main));
    c->have_phi = true;
    c->halo_valid = 0; // stored states carry no ghost-plane guarantee
    return WAFER_OK;
}

int wafer_num_states(wafer_ctx *c, uint32_t *out)
{
    if (!c || !out) return fail(WAFER_ERR_INVALID, "null argument");
    *out = (uint32_t)c->states.size();
    return WAFER_OK;
}

int wafer_clear_states(wafer_ctx *c)
{
    if (!c) return fail(WAFER_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->P.device));
    HIP_TRY(hipStreamSynchronize(c->s_main));
    for (void *p : c->states) (void)hipFree(alloc_base(c, p));
    c->states.clear();
    return recompute_gram(c);
}

// ---- diagnostics ---------------------------------------------------------------------------
// the device's copy ceiling: 16 B per lane, `unroll` (1, 2, 4, 8) vectors in flight per lane, a
// grid-stride loop over blocks_per_cu x CUs workgroups of 256 threads; V -> phi's scratch buffer
int wafer_diag_copy_bw(wafer_ctx *c, int iters, int unroll, int blocks_per_cu, double *gbps)
{
    if (!c || !gbps) return fail(WAFER_ERR_INVALID, "null argument");
    if (iters < 1 || blocks_per_cu < 1 || blocks_per_cu > 64) return fail(WAFER_ERR_INVALID, "iters >= 1, blocks_per_cu in 1..64");
    HIP_TRY(hipSetDevice(c->P.device));
    const long long n16 = (long long)c->g.total * (long long)c->esz / 16;
    const wafer_f4 *src = as<const wafer_f4>(alloc_base(c, c->v));
    wafer_f4 *dst = as<wafer_f4>(alloc_base(c, c->phi[c->cur ^ 1])); // scratch between steps
    const dim3 grid((unsigned)(c->num_cus * blocks_per_cu)), block(256);
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    for (int it = -2; it < iters; ++it) { // two warm-up launches
        if (it == 0) HIP_TRY(hipEventRecord(e0, c->s_main));
        switch (unroll) {
        case 1: hipLaunchKernelGGL((wafer_k_copy16<1>), grid, block, 0, c->s_main, src, dst, n16); break;
        case 2: hipLaunchKernelGGL((wafer_k_copy16<2>), grid, block, 0, c->s_main, src, dst, n16); break;
        case 8: hipLaunchKernelGGL((wafer_k_copy16<8>), grid, block, 0, c->s_main, src, dst, n16); break;
        default: hipLaunchKernelGGL((wafer_k_copy16<4>), grid, block, 0, c->s_main, src, dst, n16); break;
        }
    }
    HIP_TRY(hipEventRecord(e1, c->s_main));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    HIP_TRY(hipMemsetAsync(alloc_base(c, c->phi[c->cur ^ 1]), 0, (size_t)c->g.total * c->esz, c->s_main)); // the scratch buffer's frame
    *gbps = (double)n16 * 16.0 * 2.0 * iters / (ms * 1e-3) / 1e9;
    return WAFER_OK;
}

// position-dependent integer checksum of the work cells of global work planes [z_begin, z_begin + z_count)
// that this context owns (wafer_k_checksum): equal for equal bits, whatever the decomposition
int wafer_diag_x2_passes(wafer_ctx *c, uint64_t *out)
{
    if (!c || !out) return fail(WAFER_ERR_INVALID, "null argument");
    *out = c->x2_passes;
    return WAFER_OK;
}

int wafer_diag_checksum(wafer_ctx *c, uint32_t z_begin, uint32_t z_count, uint64_t *out)
{
    if (!c || !out) return fail(WAFER_ERR_INVALID, "null argument");
    if (!c->have_phi) return fail(WAFER_ERR_STATE, "phi not set");
    HIP_TRY(hipSetDevice(c->P.device));
    HIP_TRY(hipStreamSynchronize(c->s_aux));
    unsigned long long *d = reinterpret_cast<unsigned long long *>(c->scal + 20);
    HIP_TRY(hipMemsetAsync(d, 0, sizeof *d, c->s_main));
    WaferRowArgs ra;
    ra.g = c->g;
    ra.lz_lo = c->g.G;
    ra.lz_hi = c->g.G + c->g.nzl;
    const int lo = (int)z_begin, hi = (int)std::min<uint64_t>((uint64_t)z_begin + z_count, (uint64_t)c->g.nz);
    if (c->f32)
        hipLaunchKernelGGL((wafer_k_checksum<float>), dim3(c->num_cus * 8), dim3(256), 0, c->s_main, ra, as<float>(c->phi[c->cur]), lo, hi, d);
    else
        hipLaunchKernelGGL((wafer_k_checksum<double>), dim3(c->num_cus * 8), dim3(256), 0, c->s_main, ra, as<double>(c->phi[c->cur]), lo, hi, d);
    HIP_TRY(hipGetLastError());
    unsigned long long h = 0;
    HIP_TRY(hipMemcpyAsync(&h, d, sizeof h, hipMemcpyDeviceToHost, c->s_main));
    HIP_TRY(hipStreamSynchronize(c->s_main));
    *out = (uint64_t)h;
    return check_hv_err(c);
}

static int div_check_launch(wafer_ctx *c, const WaferDen<double> &dv, bool planned, uint64_t seed, uint64_t n_random, int lo_exp, int hi_exp,
                            const double *operands, size_t n_operands, uint64_t *bad_random, uint64_t *bad_operands)
{
    if (lo_exp < 0 || hi_exp > 2046 || lo_exp > hi_exp) return fail(WAFER_ERR_INVALID, "biased exponents in 0..2046");
    HIP_TRY(hipSetDevice(c->P.device));
    unsigned long long *d = nullptr;
    double *dx = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 2 * sizeof *d));
    hipError_t e = hipMemsetAsync(d, 0, 2 * sizeof *d, c->s_main);
    const int per_thread = 1024, threads = 256;
    const uint64_t blocks = (n_random + (uint64_t)per_thread * threads - 1) / ((uint64_t)per_thread * threads);
    if (e == hipSuccess && blocks > 0) {
        const dim3 grid((unsigned)std::min<uint64_t>(blocks, 1u << 30));
        if (planned) hipLaunchKernelGGL(wafer_k_div_check<true>, grid, dim3(threads), 0, c->s_main, dv, (unsigned long long)seed, per_thread, lo_exp, hi_exp, d);
        else hipLaunchKernelGGL(wafer_k_div_check<false>, grid, dim3(threads), 0, c->s_main, dv, (unsigned long long)seed, per_thread, lo_exp, hi_exp, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess && operands && n_operands) {
        e = hipMalloc((void **)&dx, n_operands * sizeof(double));
        if (e == hipSuccess) e = hipMemcpyAsync(dx, operands, n_operands * sizeof(double), hipMemcpyHostToDevice, c->s_main);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(wafer_k_div_operands, dim3((unsigned)((n_operands + 255) / 256)), dim3(256), 0, c->s_main, dv, dx,
                               (unsigned long long)n_operands, d + 1);
            e = hipGetLastError();
        }
    }
    unsigned long long h[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, c->s_main);
    hipError_t e2 = hipStreamSynchronize(c->s_main);
    (void)hipFree(d);
    if (dx) (void)hipFree(dx);
    if (e != hipSuccess || e2 != hipSuccess)
        return fail(WAFER_ERR_HIP, "division check failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    if (bad_random) *bad_random = h[0];
    if (bad_operands) *bad_operands = h[1];
    return WAFER_OK;
}

int wafer_diag_div_check(wafer_ctx *c, double den, uint64_t seed, uint64_t n_operands, int lo_exp, int hi_exp,
                         uint64_t *mismatches)
{
    if (!c || !mismatches) return fail(WAFER_ERR_INVALID, "null argument");
    return div_check_launch(c, WaferDen<double>{den, 0.0, 0.0, false}, false, seed, n_operands, lo_exp, hi_exp, nullptr, 0, mismatches, nullptr);
}

static void plan_out(const WaferDivPlan &p, wafer_div_plan_t *out)
{
    out->den = p.den;
    out->zh = p.zh;
    out->zl = p.zl;
    out->checked = p.checked;
    out->n_candidates = p.n_candidates;
    out->zl_shift = p.zl_shift;
    out->reserved = 0;
}

int wafer_div_plan(double den, wafer_div_plan_t *out, double *candidates, size_t cap, size_t *n_written)
{
    if (!out) return fail(WAFER_ERR_INVALID, "null argument");
    plan_out(wafer_divplan_make(den), out);
    if (n_written) *n_written = 0;
    if (candidates && cap) {
        const std::vector<double> cand = wafer_divplan_candidates(den);
        const size_t n = std::min(cap, cand.size());
        std::copy(cand.begin(), cand.begin() + (ptrdiff_t)n, candidates);
        if (n_written) *n_written = n;
    }
    return WAFER_OK;
}

int wafer_get_div_plan(wafer_ctx *c, wafer_div_plan_t *out)
{
    if (!c || !out) return fail(WAFER_ERR_INVALID, "null argument");
    plan_out(c->div_plan, out);
    return WAFER_OK;
}

int wafer_diag_div_planned(wafer_ctx *c, const wafer_div_plan_t *plan, uint64_t seed, uint64_t n_random, int lo_exp, int hi_exp,
                           const double *operands, size_t n_operands, uint64_t *mismatches_random, uint64_t *mismatches_operands)
{
    if (!c || !plan) return fail(WAFER_ERR_INVALID, "null argument");
    return div_check_launch(c, WaferDen<double>{plan->den, plan->zh, plan->zl, plan->checked != 0}, true, seed, n_random, lo_exp, hi_exp, operands,
                            n_operands, mismatches_random, mismatches_operands);
}

int wafer_div_plan_f32(float den, wafer_div_plan_f32_t *out)
{
    if (!out) return fail(WAFER_ERR_INVALID, "null argument");
    const WaferDivPlanF p = wafer_divplan_make_f32(den);
    out->den = p.den;
    out->zh = p.zh;
    out->zl = p.zl;
    out->checked = p.checked;
    out->zl_shift = p.zl_shift;
    return WAFER_OK;
}

int wafer_diag_div_planned_f32(wafer_ctx *c, const wafer_div_plan_f32_t *plan, int lo_exp, int hi_exp, uint64_t *mismatches)
{
    if (!c || !plan || !mismatches) return fail(WAFER_ERR_INVALID, "null argument");
    if (lo_exp < 0 || hi_exp > 254 || lo_exp > hi_exp) return fail(WAFER_ERR_INVALID, "biased exponents in 0..254");
    HIP_TRY(hipSetDevice(c->P.device));
    unsigned long long *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, sizeof *d));
    hipError_t e = hipMemsetAsync(d, 0, sizeof *d, c->s_main);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(wafer_k_div_check_f32, dim3((unsigned)c->num_cus * 16), dim3(256), 0, c->s_main,
                           WaferDen<float>{plan->den, plan->zh, plan->zl, plan->checked != 0}, lo_exp, hi_exp, d);
        e = hipGetLastError();
    }
    unsigned long long h = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d, sizeof h, hipMemcpyDeviceToHost, c->s_main);
    const hipError_t e2 = hipStreamSynchronize(c->s_main);
    (void)hipFree(d);
    if (e != hipSuccess || e2 != hipSuccess) return fail(WAFER_ERR_HIP, "division check failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    *mismatches = h;
    return WAFER_OK;
}

int wafer_set_stream(wafer_ctx *c, void *hip_stream)
{
    if (!c) return fail(WAFER_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->P.device));
    HIP_TRY(hipStreamSynchronize(c->s_main));
    c->s_main = hip_stream ? (hipStream_t)hip_stream : c->s_own;
    return WAFER_OK;
}

int wafer_get_slab_info(wafer_ctx *c, wafer_slab_info *out)
{
    if (!c || !out) return fail(WAFER_ERR_INVALID, "null argument");
    out->z_begin = (uint32_t)c->g.z_begin;
    out->z_count = (uint32_t)c->g.nzl;
    out->halo_depth = (uint32_t)c->g.G;
    out->ext = (uint32_t)c->g.R;
    out->plane_elems = (uint64_t)c->g.plane;
    out->elem_bytes = (uint64_t)c->esz;
    return WAFER_OK;
}

int wafer_get_device_info(wafer_ctx *c, wafer_device_info *out)
{
    if (!c || !out) return fail(WAFER_ERR_INVALID, "null argument");
    hipDeviceProp_t p;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess)
        return fail(WAFER_ERR_HIP, "hipGetDeviceProperties failed");
    memset(out, 0, sizeof *out);
    snprintf(out->name, sizeof out->name, "%s", p.name);
    snprintf(out->arch, sizeof out->arch, "%s", p.gcnArchName);
    out->compute_units = (uint32_t)p.multiProcessorCount;
    out->memory_clock_khz = (uint32_t)p.memoryClockRate;
    out->memory_bus_bits = (uint32_t)p.memoryBusWidth;
    out->l2_bytes = (uint32_t)p.l2CacheSize;
    out->total_bytes = (uint64_t)p.totalGlobalMem;
    return WAFER_OK;
}

} // extern "C"
