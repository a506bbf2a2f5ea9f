// wafer_engine.hip -- context, launch logic and the C ABI of include/wafer_hip.h.
//
// Host side mirrors the call structure of Wafer's grid.rs (run/solve/evolve/
// compute_observables/normalise/orthogonalise) with device-resident state.
// No CPU fallback exists: every entry point fails loudly if HIP does.
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <cfloat>
#include <cstdarg>
#include <initializer_list>
#include <type_traits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/wafer_hip.h"
#include "wafer_elementwise.hip.h"
#include "wafer_geom.h"
#include "wafer_setup.hip.h"
#include "wafer_stencil.hip.h"
#include "wafer_stencil_lds.hip.h"
#include "wafer_stencil_fused2.hip.h"
#include "wafer_stencil_fused3.hip.h"
#include "wafer_launch.h"
#include "wafer_tuning.h"

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int fail(int code, const char *fmt, ...)
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

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(WAFER_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                 \
    } while (0)

#define TRY(expr)                  \
    do {                           \
        int rc_ = (expr);          \
        if (rc_ != WAFER_OK) return rc_; \
    } while (0)

// ---------------------------------------------------------------------------
// roctx ranges: evolve / observables / halo exchange show up by name on a
// rocprofv3 --marker-trace timeline.  The library is looked up at first use (rocprofiler-sdk's roctx,
// then the legacy libroctx64) so that nothing is linked; without it the ranges are no-ops.
// WAFER_ROCTX=0 switches them off.
// ---------------------------------------------------------------------------
#include <dlfcn.h>
namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        const char *e = getenv("WAFER_ROCTX");
        if (e && *e == '0') return;
        for (const char *name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
            pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (push && pop) return;
            push = nullptr;
            pop = nullptr;
        }
    }
};
static Roctx &roctx()
{
    static Roctx r;
    return r;
}
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char *name) : on(roctx().push != nullptr)
    {
        if (on) roctx().push(name);
    }
    ~RoctxRange()
    {
        if (on) roctx().pop();
    }
};
} // namespace

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

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
enum { SCAL_SLOTS = 32 };

struct wafer_ctx {
    wafer_params P;
    WaferGeom g;
    bool f32 = false;       // fp32 storage
    bool f32_arith = false; // ... and fp32 arithmetic in the ground-state stencil steps (WAFER_F32_FAST)
    size_t esz = 8;

    hipStream_t s_main = nullptr, s_aux = nullptr, s_own = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr, ev_fork = nullptr, ev_join = nullptr, ev_bdry = nullptr;

    void *phi[2] = {nullptr, nullptr};
    int cur = 0;
    void *v = nullptr, *a = nullptr, *b = nullptr, *potsub = nullptr;
    std::vector<void *> states;
    // two excited-state steps per pass (wafer_stencil_x2.hip.h): M_j = A l_j of the first x2_ready stored states, the matrix
    // <l_j, M_i> (device) and the load transform's coefficient block
    std::vector<void *> mstates;
    int x2_ready = 0;
    double *x2mat = nullptr, *x2coef = nullptr;
    uint64_t x2_passes = 0;
    int potsub_kind = WAFER_POTSUB_NONE;
    double potsub_scalar = 0.0;
    bool have_pot = false, have_phi = false;
    bool v_in_range = false; // 2^-400 < |1 + dt*V/2| < 2^400 everywhere (wafer_recip's short form is exact)
    int x2_agreed[4] = {-1, -1, -1, -1}; // [k]: every rank can take the two-step excited pass with k stored states (-1: not agreed yet; x2_agree)
    int vgen_type = 0;       // V was generated from this closed form (Coulomb / SimpleCornell / Harmonic), else 0: kernels may re-evaluate it instead of streaming it

    double *partials = nullptr; // [1 + WAFER_MAX_LOW][partials_stride]
    size_t partials_stride = 0;
    double *gram = nullptr;     // WAFER_MAX_LOW^2 doubles, device: G_ji = <state j | state i>
    double gram_host[WAFER_MAX_LOW * WAFER_MAX_LOW] = {0};
    double *scal = nullptr;     // SCAL_SLOTS doubles, device
    double *scal_host = nullptr; // pinned mirror

    // launch geometry shared by the column-marching kernels
    int bx = 0, by = 0;
    int num_cus = 256;

    wafer_halo_fn halo_hook = nullptr;
    wafer_allreduce_fn allreduce_hook = nullptr;
    void *hook_user = nullptr;
    int overlap_mode = 2;   // wafer_set_overlap: 0 exchange after the pass, 1 boundary-first split pass, 2 single-launch half-slab pass
    WaferTuning tune;       // WAFER_* knobs, read once in wafer_ctx_create
    // three-step kernel: workgroup tables by launch shape (device copies), and the words of the single-launch slab pass
    struct F3Table {
        int kind, lz_lo, lz_hi, aux;
        WaferF3Block *dev;
        int nblocks, nbump[2];
        int dir;   // 1: every workgroup marches up, 2: every one down, 0: both occur
    };
    std::vector<F3Table> f3_tables;
    unsigned long long *hv_words = nullptr; // device memory, four 64-byte lines: cnt[0], cnt[1] (finished workgroups per half), flag[0], flag[1]
                                            // (exchanges completed per ghost side, written by the exchange stream)
    unsigned *hv_err = nullptr;             // host memory: set by a workgroup or gate kernel whose wait gave up
    unsigned long long hv_cnt_target[2] = {0, 0}, hv_flag_epoch[2] = {0, 0};
    int hv_first = 0;                       // which half the next single-launch pass dispatches first
    // peer stores (wafer_set_overlap mode 3): the z-neighbours' buffers and arrival counters as mapped here, this context's own
    // counters (their own allocation: peers map it), and how many arrivals each ghost side has been promised so far
    struct PeerSide {
        bool connected = false;
        void *phi[2] = {nullptr, nullptr};
        unsigned long long *flags = nullptr;
        int nzl = 0;
        void *ipc_map[3] = {nullptr, nullptr, nullptr};   // what hipIpcOpenMemHandle returned (to close)
    } peer[2];
    bool peer_ready = false;
    WaferF3Peer *peer_dev = nullptr;            // device copy of what the boundary workgroups need (written by wafer_peer_connect)
    unsigned long long *peer_flags = nullptr;   // [0], [8]: arrivals into the lower / upper ghost planes
    unsigned long long peer_expect[2] = {0, 0};
    hipEvent_t ev_ex[2] = {nullptr, nullptr}; // single-launch pass: the last exchange of each side
    int halo_valid = 0; // ghost planes of phi[cur] (counted from the owned region) known to be current
    int halo_cycle = 1; // fused passes per halo exchange: the exchange moves 2R * halo_cycle planes (<= G), see wafer_evolve

    uint64_t last_steps = 0;
    bool timing_valid = false;
    int variant = -1;
    std::string kernel_name;
    bool last_instance_valid = false;   // a plain three-step launch has run: wafer_step3_last_instance names its instantiation
    char instance_name[160] = {0};

    bool has_lo() const { return g.z_begin > 0; }
    bool has_hi() const { return g.z_begin + g.nzl < g.nz; }
    bool sharded() const { return has_lo() || has_hi(); }
};

// planes per workgroup so that a launch over `nplanes` has >= target blocks
static int pick_zchunk(const wafer_ctx *c, int nplanes, int target_blocks)
{
    if (c->tune.zchunk > 0) return c->tune.zchunk;
    const long long per_layer = (long long)c->bx * c->by;
    long long nch = (target_blocks + per_layer - 1) / per_layer;
    if (nch < 1) nch = 1;
    if (nch > nplanes) nch = nplanes;
    if (nch > 64) nch = 64;
    return (int)((nplanes + nch - 1) / nch);
}

static inline int nchunks_of(int nplanes, int zchunk) { return (nplanes + zchunk - 1) / zchunk; }

template <typename T>
static inline T *as(void *p) { return static_cast<T *>(p); }

// single-launch slab pass: a workgroup or gate kernel that gave up waiting leaves a word in host memory; every call
// that has just synchronised with the device reports it (defined with the pass, below)
static int check_hv_err(wafer_ctx *c);

// Arrays are held as LOGICAL pointers to (plane 0, row 0); the allocation starts
// base_off elements earlier (guard planes / rows, wafer_geom.h).
static inline void *alloc_base(const wafer_ctx *c, void *logical)
{
    return logical ? static_cast<char *>(logical) - (size_t)c->g.base_off * c->esz : nullptr;
}
static int alloc_grid_array(wafer_ctx *c, void **logical, hipStream_t s)
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
template <typename T>
static inline const T *as(const void *p) { return static_cast<const T *>(p); }

// The stored a, b arrays (potential.rs:101-110) are needed by the kernels that stream them (variant 0,
// WAFER_ABV=0) and by wafer_download_array; everything else forms a, b from V in registers.  They
// are allocated and filled on first use and kept in step with V from then on.
static int ensure_ab(wafer_ctx *c)
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
static int refresh_ab(wafer_ctx *c)
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
static bool kernels_stream_ab(const wafer_ctx *c, int variant) { return variant == 0 || c->tune.abv == 0; }

// after V changed: may the kernels that form a, b from V use the short reciprocal?
static int check_v_range(wafer_ctx *c)
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

// second-stage reduce of `nq` quantities of `n` partials each into scal[slot..slot+nq)
static int reduce_to_scal(wafer_ctx *c, int nq, long long n, int slot, hipStream_t s)
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

static int read_scal(wafer_ctx *c, int slot, int n, double *out, hipStream_t s)
{
    HIP_TRY(hipMemcpyAsync(c->scal_host + slot, c->scal + slot, sizeof(double) * n,
                           hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    TRY(check_hv_err(c));
    for (int q = 0; q < n; ++q) out[q] = c->scal_host[slot + q];
    return WAFER_OK;
}

// ---------------------------------------------------------------------------
// halo exchange through the host-installed hook
// ---------------------------------------------------------------------------
// the first / last `planes` owned planes of any grid array (logical pointer) to the z-neighbours' ghost planes
static int exchange_halo_array(wafer_ctx *c, void *array, hipStream_t s, int planes)
{
    if (!c->sharded()) return WAFER_OK;
    if (!c->halo_hook) return fail(WAFER_ERR_COMM, "context owns a z-slab but no halo hook is installed");
    RoctxRange range_("wafer_halo_exchange");
    const WaferGeom &g = c->g;
    if (planes > g.G || planes > g.nzl) return fail(WAFER_ERR_INVALID, "halo exchange deeper than the slab allows");
    char *base = static_cast<char *>(array);
    const size_t plane_b = (size_t)g.plane * c->esz;
    // from row 0 of the first plane to the last padded row of the last plane (guard rows in between ride along)
    const size_t bytes = ((size_t)(planes - 1) * (size_t)g.plane + (size_t)g.py * (size_t)g.pitch) * c->esz;
    void *send_lo = c->has_lo() ? base + (size_t)g.G * plane_b : nullptr;
    void *recv_lo = c->has_lo() ? base + (size_t)(g.G - planes) * plane_b : nullptr;
    void *send_hi = c->has_hi() ? base + (size_t)(g.G + g.nzl - planes) * plane_b : nullptr;
    void *recv_hi = c->has_hi() ? base + (size_t)(g.G + g.nzl) * plane_b : nullptr;
    if (c->halo_hook(c->hook_user, send_lo, send_hi, recv_lo, recv_hi, bytes, (void *)s) != 0)
        return fail(WAFER_ERR_COMM, "halo hook failed");
    return WAFER_OK;
}

static int exchange_halo(wafer_ctx *c, int buf, hipStream_t s, int planes) { return exchange_halo_array(c, c->phi[buf], s, planes); }

// One direction of the exchange (wafer_set_overlap mode 4).  side 0: the LOWEST owned planes go to the lower
// neighbour, the upper neighbour's lowest planes arrive in the UPPER ghost planes; side 1: the mirror image.  Every
// rank calls the same side at the same point of a pass, so the sends and receives pair up.
static int exchange_halo_side(wafer_ctx *c, int buf, hipStream_t s, int planes, int side)
{
    if (!c->halo_hook) return fail(WAFER_ERR_COMM, "context owns a z-slab but no halo hook is installed");
    RoctxRange range_("wafer_halo_exchange");
    const WaferGeom &g = c->g;
    if (planes > g.G || planes > g.nzl) return fail(WAFER_ERR_INVALID, "halo exchange deeper than the slab allows");
    char *base = static_cast<char *>(c->phi[buf]);
    const size_t plane_b = (size_t)g.plane * c->esz;
    const size_t bytes = ((size_t)(planes - 1) * (size_t)g.plane + (size_t)g.py * (size_t)g.pitch) * c->esz;
    void *send_lo = (side == 0 && c->has_lo()) ? base + (size_t)g.G * plane_b : nullptr;
    void *recv_hi = (side == 0 && c->has_hi()) ? base + (size_t)(g.G + g.nzl) * plane_b : nullptr;
    void *send_hi = (side == 1 && c->has_hi()) ? base + (size_t)(g.G + g.nzl - planes) * plane_b : nullptr;
    void *recv_lo = (side == 1 && c->has_lo()) ? base + (size_t)(g.G - planes) * plane_b : nullptr;
    if (!send_lo && !send_hi && !recv_lo && !recv_hi) return WAFER_OK;
    if (c->halo_hook(c->hook_user, send_lo, send_hi, recv_lo, recv_hi, bytes, (void *)s) != 0)
        return fail(WAFER_ERR_COMM, "halo hook failed");
    return WAFER_OK;
}

// makes at least `need` ghost planes of phi[cur] current
static int ensure_halo(wafer_ctx *c, int need)
{
    if (c->sharded() && c->halo_valid < need) {
        TRY(exchange_halo(c, c->cur, c->s_main, need));
        c->halo_valid = need;
    }
    return WAFER_OK;
}

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

static int default_variant(const wafer_ctx *c)
{
    if (c->tune.stencil_variant >= 0) return c->tune.stencil_variant;
    // FivePoint on fp32 storage with fp64 arithmetic: the single-step kernel and the two-step kernel on 128 x 16 tiles
    // (wafer_stencil_fused2w.hip.h) take the same time (512^3: 0.337 against 0.335 ms/step) and the single step needs half the
    // ghost planes on slabs; with fp32 arithmetic as well the two-step kernel wins (0.250 against 0.288)
    if (c->f32 && !c->f32_arith && c->g.R == 2) return 1;
    // SevenPoint: the two-step kernel exists (variant 2, bit-exact, 128 x 8 tiles) but recomputes phi1 on 14 rows
    // per 8 and is issue-bound: 0.93 ms/step at 512^3 against 0.63 for the single-step kernel on 128 x 16 tiles
    if (c->g.R == 3) return 1;
    // ThreePoint, every type combination: three steps per pass (wafer_stencil_fused3.hip.h); everything else two
    if (c->g.R == 1 && c->tune.fuse3 != 0) return 3;
    return 2;
}

static int active_variant(const wafer_ctx *c) { return c->variant >= 0 ? c->variant : default_variant(c); }

// The closed form a kernel may evaluate instead of streaming V (0: none): fp64 contexts whose potential was
// generated from Coulomb / SimpleCornell / Harmonic and whose radii dn .. dn * sqrt(3) (n + 1) / 2 lie inside
// the range of the short reciprocal (wafer_vgen_at); WAFER_VGEN=0 keeps every kernel on the stored array.
static int closed_form_vg(const wafer_ctx *c)
{
    const bool r_ok = c->P.dn > 0x1p-300 && c->P.dn * ((double)c->g.nx + c->g.ny + c->g.nz + 3.) < 0x1p300;
    return (!c->f32 && r_ok && c->tune.vgen != 0) ? c->vgen_type : 0;
}
static void set_vg_args(const wafer_ctx *c, WaferStepArgs &a)
{
    a.vg_dn = c->P.dn;
    a.vg_mass = c->P.mass;
    a.vg_sig = c->P.sig;
}

// storage / arithmetic types of a launch (wafer_launch.h): WAFER_F32_FAST computes the ground-state stencil steps in
// fp32 as well (sums, projections and observables stay fp64); plain fp32 storage widens to fp64 in registers
static int type_combo(const wafer_ctx *c, bool step_kernel)
{
    if (!c->f32) return WAFER_TC_F64;
    return (c->f32_arith && step_kernel) ? WAFER_TC_F32_F32 : WAFER_TC_F32_F64;
}

static WaferStepArgs step_args(const wafer_ctx *c, int lz_lo, int lz_hi)
{
    WaferStepArgs a{};
    a.g = c->g;
    a.lz_lo = lz_lo;
    a.lz_hi = lz_hi;
    a.dt = c->P.dt;
    a.target_blocks = c->num_cus;
    a.v_in_range = c->v_in_range ? 1 : 0;
    const int R = c->g.R;
    const double lead = (R == 1) ? 2. : (R == 2) ? 24. : 360.;
    a.den = lead * c->P.dn * c->P.dn * c->P.mass; // grid.rs:569 / 594 / 626
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

static int direct_target_blocks(const wafer_ctx *c) { return c->tune.target_blocks > 0 ? c->tune.target_blocks : 4096; }

// one step over local planes [lz_lo, lz_hi); norm: also sum phi'^2 into the partials (more stored states than the
// fused-overlap kernel carries)
static int launch_step(wafer_ctx *c, int src, int dst, int lz_lo, int lz_hi, bool norm, hipStream_t s)
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
static long long step_partials_count(wafer_ctx *c, int lz_lo, int lz_hi)
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
static bool fuse3_applies(const wafer_ctx *c)
{
    // Small undecomposed grids are launch- and fill-bound and the deeper pipeline costs there: 50^3 5.8 against 4.5
    // us/step for the two-step kernel, 64^3 6.0 / 4.7, 128^3 9.1 / 8.7; from 256^3 up it wins (40.5 / 41.9 us, 384^3
    // 0.177 / 0.196 ms).  WAFER_FUSE3_MIN_NY (tests) lifts both thresholds.
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
static bool fuse2_applies(const wafer_ctx *c)
{
    const int R = c->g.R;
    return active_variant(c) >= 2 && (R <= 2 || !c->f32) && (!c->sharded() || c->g.G >= 2 * R);
}

// ---- workgroup tables of the three-step kernel (wafer_stencil_fused3.hip.h), built once per launch shape ---------
enum { F3_PLAIN = 0, F3_MIXED = 1, F3_HALVES = 2, F3_WHOLE = 3 };
static int f3_table(wafer_ctx *c, int kind, int lz_lo, int lz_hi, int aux, const wafer_ctx::F3Table **out)
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
        // is cut short: there is no exchange kernel to hand CUs to (WAFER_HV_SHORT_TILES still applies if set)
        const bool peer = (aux & 4) != 0;
        const bool need_wait[2] = {peer ? (aux & 8) != 0 : true, peer ? (aux & 16) != 0 : true};
        const int ntiles = ntx * nty;
        const int nshort = c->tune.hv_short_tiles >= 0 ? c->tune.hv_short_tiles : (peer ? 0 : (ntiles >= 64 ? ntiles / 16 : 0));
        wafer_f3_schedule_halves(host, ntx, nty, lz_lo, lz_hi, lz_lo + (lz_hi - lz_lo) / 2, aux & 1, need_wait,
                                 (c->tune.hv_debug & 8) ? 0 : nshort, c->tune.hv_nsub, 3 * c->g.R /* planes per exchange */, !(aux & 2),
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
static int launch_step3(wafer_ctx *c, int src, int dst, int lz_lo, int lz_hi, hipStream_t s, bool short_tail = false)
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
    if (wafer_entry_step3_fused(type_combo(c, true), c->tune, a, tab->dev, tab->nblocks, WaferF3Sync(), c->phi[src], c->v, c->phi[dst], s, tab->dir) != hipSuccess)
        return fail(WAFER_ERR_HIP, "three-step stencil launch failed: %s", hipGetErrorString(hipGetLastError()));
    c->last_instance_valid = true;
    return WAFER_OK;
}

// two fused steps over planes [lz_lo, lz_hi): phi[dst] = step(step(phi[src]))
// short_tail: the interior launch of a slab -- one long workgroup per tile, except the last 1/16 of
// the tiles, which go as four short workgroups each (see wafer_evolve)
static int launch_step2(wafer_ctx *c, int src, int dst, int lz_lo, int lz_hi, hipStream_t s, bool short_tail = false)
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
static int launch_normalise(wafer_ctx *c, int buf, const double *norm2_dev, double norm2_imm,
                            void *lower, int out_slot, hipStream_t s)
{
    int nb;
    TRY(launch_row_op<2>(c, c->phi[buf], lower, nullptr, norm2_dev, norm2_imm, s, &nb));
    if (lower) TRY(reduce_to_scal(c, 1, nb, out_slot, s));
    return WAFER_OK;
}

static int launch_axpy(wafer_ctx *c, int buf, void *lower, int overlap_slot, void *next, int out_slot,
                       hipStream_t s)
{
    int nb;
    TRY(launch_row_op<3>(c, c->phi[buf], lower, next, c->scal + overlap_slot, 0.0, s, &nb));
    if (next) TRY(reduce_to_scal(c, 1, nb, out_slot, s));
    return WAFER_OK;
}

static int launch_dot(wafer_ctx *c, void *phi, void *lower, int out_slot, hipStream_t s)
{
    int nb;
    TRY(launch_row_op<1>(c, phi, lower, nullptr, nullptr, 0.0, s, &nb));
    return reduce_to_scal(c, 1, nb, out_slot, s);
}

// Gram-Schmidt chain on `buf` against states [0,wnum); the first overlap is
// already in scal[1] when first_dot_done.
static int gs_chain(wafer_ctx *c, int buf, uint32_t wnum, bool first_dot_done, hipStream_t s)
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
static int recompute_gram(wafer_ctx *c)
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
static int excited_stencil_launch(wafer_ctx *c, int src, int dst, uint32_t wnum, bool transform_on_load, int lz_lo, int lz_hi,
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
static int excited_step_launch(wafer_ctx *c, int src, int dst, uint32_t wnum, bool transform_on_load, hipStream_t s)
{
    long long nb = 0;
    TRY(excited_stencil_launch(c, src, dst, wnum, transform_on_load, c->g.G, c->g.G + c->g.nzl, 0, s, &nb));
    return reduce_to_scal(c, 1 + (int)wnum, nb, 0, s);
}

// z-slabs: the R boundary planes of each side first, on the second stream, their (raw) halo exchange
// behind the interior launch; the sums wait for all three launches
static int excited_step_launch_overlapped(wafer_ctx *c, int src, int dst, uint32_t wnum, bool transform_on_load)
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
enum { X2_SUM_SLOT = 18 };   // scal[18 .. 18 + 1 + 2k): the sums of a two-step pass
// ThreePoint fp64, one to three stored states; z-slabs need two ghost planes (a pass consumes two per side).  Nothing here
// depends on the local slab: what does (the potential inside the short reciprocal's range, two owned planes, memory for the
// images) is the ranks' agreement in x2_agree.
static bool x2_applies(const wafer_ctx *c, uint32_t wnum)
{
    // three stored states: the 128 x 8-tile kernel wins where a plane is small enough for the halo rows to stay in the XCDs' L2
    // (-4 % per step at 512 x 512, -2 ... -4 % at 256^2 / 384^2) and loses on 1024 x 1024 planes (+3 ... +5 %,
    // profiles/r04_x2_shapes.log).  The plane extent is the same on every rank of a decomposed run.
    const int kmax = c->tune.x2_max_k > 0 ? c->tune.x2_max_k : ((long long)c->g.nx * c->g.ny <= 300000 ? 3 : 2);
    return c->tune.x2 != 0 && c->tune.one_pass != 0 && !c->f32 && c->g.R == 1 && wnum >= 1 && wnum <= 3 && (int)wnum <= kmax &&
           active_variant(c) >= 1 && (!c->sharded() || c->g.G >= 2);
}

// Storage for M_j = A l_j of the first wnum stored states.  Running out of memory here is not an error of the call that asked:
// the one-step path needs none of it (x2_agree).
static int alloc_mstates(wafer_ctx *c, uint32_t wnum)
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
static int x2_agree(wafer_ctx *c, uint32_t wnum, bool *out)
{
    *out = false;
    if (!x2_applies(c, wnum)) return WAFER_OK;
    bool local_ok = c->v_in_range && (!c->sharded() || c->g.nzl >= 2);
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
static int ensure_x2(wafer_ctx *c, uint32_t wnum)
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
        if (wafer_entry_step_lds(WAFER_TC_F64, g.R, c->tune, a, c->states[j], c->a, c->b, c->v, c->mstates[j], c->s_main, closed_form_vg(c)) != hipSuccess)
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
static int x2_run(wafer_ctx *c, uint32_t wnum, uint64_t pairs, hipStream_t s)
{
    const WaferGeom &g = c->g;
    const int k = (int)wnum, nq = wafer_entry_x2_nsums(k);
    const double *amat = c->x2mat;
    const void *l[3] = {nullptr, nullptr, nullptr}, *m[3] = {nullptr, nullptr, nullptr};
    for (int j = 0; j < k; ++j) { l[j] = c->states[j]; m[j] = c->mstates[j]; }
    if (wafer_entry_x2_coeffs(1, k, c->scal, c->gram, amat, c->x2coef, s) != hipSuccess)
        return fail(WAFER_ERR_HIP, "coefficient kernel launch failed");
    const WaferStepArgs a = step_args(c, g.G, g.G + g.nzl);
    const long long nb = wafer_entry_x2_blocks(c->tune, g, k, closed_form_vg(c), g.G, g.G + g.nzl, c->num_cus);
    if (nb > (long long)c->partials_stride) return fail(WAFER_ERR_INVALID, "partials buffer too small");
    TRY(ensure_halo(c, 2));   // z-slabs: two ghost planes of the raw input per side and pass
    for (uint64_t p = 0; p < pairs; ++p) {
        const int src = c->cur, dst = c->cur ^ 1;
        if (wafer_entry_xstep2(c->tune, a, k, closed_form_vg(c), c->phi[src], c->v, c->phi[dst], c->partials, c->partials_stride, l, m,
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
    if (wafer_entry_x2_apply(g, g.G, g.G + g.nzl, k, c->phi[c->cur], l, m, c->x2coef, c->partials, c->partials_stride, c->num_cus, s, &nap) != hipSuccess)
        return fail(WAFER_ERR_HIP, "apply launch failed");
    TRY(reduce_to_scal(c, 1, nap, X2_SUM_SLOT, s));
    TRY(launch_normalise(c, c->cur, c->scal + X2_SUM_SLOT, 0.0, nullptr, 0, s));
    c->halo_valid = 0;
    return WAFER_OK;
}

static int excited_apply(wafer_ctx *c, int buf, uint32_t wnum, hipStream_t s)
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
    c->g = wafer_make_geom((int)p->nx, (int)p->ny, (int)p->nz, R, G, (int)zb, (int)zc, (int)c->esz);
    c->bx = (c->g.px + 63) / 64; // covers both the work area and the padded extent
    c->by = (c->g.py + 3) / 4;
    c->tune = wafer_tuning_from_env(); // the only place the WAFER_* tuning variables are read
    c->overlap_mode = (c->tune.overlap >= 0 && c->tune.overlap <= 2) ? c->tune.overlap : 2; // the modes of wafer_set_overlap
    // fused passes per halo exchange: 1 unless the host asks for deep halos (wafer_set_halo_cycle) -- a
    // concentrated exchange outlasts the interior launch it hides behind on anything but a very fast link
    c->halo_cycle = std::max(1, c->tune.halo_cycle);
    if (2 * R * c->halo_cycle > G) c->halo_cycle = std::max(1, G / (2 * R));

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
    c->kernel_name = kVariants[(default_variant(c) >= 0 && default_variant(c) < kNumVariants) ? default_variant(c) : 0].name;
    *out = c;
    return WAFER_OK;
}

int wafer_ctx_destroy(wafer_ctx *c)
{
    if (!c) return WAFER_OK;
    (void)hipSetDevice(c->P.device);
    if (c->s_own) (void)hipStreamSynchronize(c->s_own);
    if (c->s_aux) (void)hipStreamSynchronize(c->s_aux);
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
    delete c;
    return WAFER_OK;
}

int wafer_synchronize(wafer_ctx *c)
{
    if (!c) return fail(WAFER_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->P.device));
    HIP_TRY(hipStreamSynchronize(c->s_aux));
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
    return download_padded(c, phi, c->phi[c->cur]);
}

int wafer_download_phi_owned(wafer_ctx *c, double *out)
{
    if (!c || !out) return fail(WAFER_ERR_INVALID, "null argument");
    if (!c->have_phi) return fail(WAFER_ERR_STATE, "phi not set");
    HIP_TRY(hipSetDevice(c->P.device));
    HIP_TRY(hipStreamSynchronize(c->s_aux));
    // host element (0, 0, 0) = work cell (0, 0, z_begin): padded coordinates (R, R, z_begin + R)
    return convert_host_array<false>(c, out, c->g.nx, c->g.ny, c->g.nzl, c->g.R, c->g.R, c->g.z_begin + c->g.R, c->phi[c->cur]);
}

// ---- evolve (grid.rs:544-687) ----------------------------------------------------
// ---- the single-launch pass of a z-slab (wafer_set_overlap mode 2) ----------------------------------------------------
// One launch per three-step pass updates the whole slab as two halves marched outwards from the cut (wafer_f3_schedule_halves).
// A half's workgroups count themselves done (cnt[half], system-scope atomics after their last stores); the exchange stream
// waits for that count and sends the half's boundary planes while the other half -- or the next pass -- computes; the
// ghost planes an exchange fills are announced by flag[side], which the workgroups that read them poll just before their
// first load of a ghost plane, i.e. near the END of their column.  No thin boundary launches, no event hops between the
// streams, one pipeline fill more per tile than an undecomposed slab.
__global__ __launch_bounds__(64) void wafer_k_gate(const unsigned long long *cnt, unsigned long long target, unsigned *err, unsigned max_spins,
                                                   int system_scope = 0)
{
    // one wave, a handful of registers: it shares a CU with a resident stencil workgroup (which leaves 8 VGPRs per SIMD)
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while ((system_scope ? __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                             : __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < target) {
            __builtin_amdgcn_s_sleep(32);
            if (++spins > max_spins) {   // four times what a workgroup waits, so that a late exchange shows as the workgroups' error
                __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
        if (system_scope) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // what the counted workgroups stored is visible to what follows in the stream
    }
}
__global__ __launch_bounds__(64) void wafer_k_post(unsigned long long *flag, unsigned long long value)
{
    if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static int ensure_hv(wafer_ctx *c)
{
    if (c->hv_words) return WAFER_OK;
    HIP_TRY(hipHostMalloc((void **)&c->hv_err, 64, hipHostMallocCoherent | hipHostMallocMapped));
    *c->hv_err = 0;
    HIP_TRY(hipMalloc((void **)&c->hv_words, 4 * 64));
    HIP_TRY(hipMemset(c->hv_words, 0, 4 * 64));
    return WAFER_OK;
}
static unsigned long long *hv_cnt(wafer_ctx *c, int half) { return c->hv_words + half * WAFER_F3_SYNC_STRIDE; }
static unsigned long long *hv_flag(wafer_ctx *c, int side) { return c->hv_words + (2 + side) * WAFER_F3_SYNC_STRIDE; }

// WAFER_HV_WAIT_MS as a spin count (one spin = s_sleep 32 + a poll, about a microsecond)
static unsigned hv_spins(const wafer_ctx *c, int mul)
{
    const long long n = (long long)c->tune.hv_wait_ms * 1000 * mul;
    return (unsigned)(n < 0xffffffffll ? n : 0xffffffffll);
}

// exchange stream: wait until every workgroup of `half` of the current launch has finished
static int hv_gate(wafer_ctx *c, int half)
{
    hipLaunchKernelGGL(wafer_k_gate, dim3(1), dim3(64), 0, c->s_aux, hv_cnt(c, half), c->hv_cnt_target[half], c->hv_err, hv_spins(c, 4), 0);
    HIP_TRY(hipGetLastError());
    return WAFER_OK;
}
// exchange stream: ghost side g has been filled once more (in stream order behind the exchange: its kernels have
// completed, their writes are visible device-wide)
static int hv_post(wafer_ctx *c, int g)
{
    const unsigned long long v = ++c->hv_flag_epoch[g];
    hipLaunchKernelGGL(wafer_k_post, dim3(1), dim3(64), 0, c->s_aux, hv_flag(c, g), v);
    HIP_TRY(hipGetLastError());
    return WAFER_OK;
}

static int check_hv_err(wafer_ctx *c)
{
    if (c->hv_err && *c->hv_err != 0) {
        const unsigned e = *c->hv_err;
        *c->hv_err = 0;
        return fail(WAFER_ERR_COMM, "single-launch slab pass: a %s gave up waiting (halo exchange never completed)",
                    e == 2 ? "gate kernel" : "workgroup");
    }
    return WAFER_OK;
}

// ---- peer stores (wafer_set_overlap mode 3) --------------------------------------------------------------------------
static int ensure_peer_flags(wafer_ctx *c)
{
    if (c->peer_flags) return WAFER_OK;
    // fine-grained where the runtime offers it (coherent for peers without cache maintenance); every access is a system-scope atomic
    void *p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, 2 * 64, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        HIP_TRY(hipMalloc(&p, 2 * 64));
    }
    HIP_TRY(hipMemset(p, 0, 2 * 64));
    c->peer_flags = static_cast<unsigned long long *>(p);
    return WAFER_OK;
}

// The same single launch as launch_halves_pass, but the boundary workgroups deliver their planes themselves (WaferF3Sync::peer).
// need[h]: the arrivals promised to ghost side h by all earlier passes of this context's life; a pass adds one per tile and side.
static int launch_peer_pass(wafer_ctx *c, int src, int dst, int E)
{
    const WaferGeom &g = c->g;
    const int lo = g.G, hi = g.G + g.nzl;
    (void)E;
    const int first = c->hv_first;
    const wafer_ctx::F3Table *tab = nullptr;
    // Whole columns (no cut) where every CU gets a tile of its own; else the two halves (twice the workgroups).  Rank-invariant:
    // the tile count follows nx, ny only.  WAFER_HV_LAYOUT=3 forces the halves.
    int tx_, ty_;
    wafer_step3_tile(type_combo(c, true), &tx_, &ty_);
    const long long ntiles = (long long)((g.nx + tx_ - 1) / tx_) * ((g.ny + ty_ - 1) / ty_);
    const bool whole = (ntiles >= c->num_cus && c->tune.hv_layout != 3) || c->tune.hv_layout == 4;   // (4: always, tests)
    // aux bits: 1 the half dispatched first / the marching direction, 4 peer mode (no short columns), 8 / 16: a neighbour below / above (who waits)
    TRY(f3_table(c, whole ? F3_WHOLE : F3_HALVES, lo, hi, first | 4 | (c->has_lo() ? 8 : 0) | (c->has_hi() ? 16 : 0), &tab));
    WaferF3Sync sy;
    sy.peer = 1;
    sy.cnt = hv_cnt(c, 0);   // (unused in peer mode)
    sy.flag = c->peer_flags;
    sy.err = c->hv_err;
    sy.debug = c->tune.hv_debug;
    sy.max_spins = hv_spins(c, 1);
    sy.peer_dev = c->peer_dev;
    sy.peer_buf = dst;
    for (int h = 0; h < 2; ++h) sy.need[h] = c->peer_expect[h];
    const WaferStepArgs a = step_args(c, lo, hi);
    if (wafer_entry_step3_fused(type_combo(c, true), c->tune, a, tab->dev, tab->nblocks, sy, c->phi[src], c->v, c->phi[dst], c->s_main, tab->dir) != hipSuccess)
        return fail(WAFER_ERR_HIP, "three-step stencil launch failed: %s", hipGetErrorString(hipGetLastError()));
    c->last_instance_valid = true;
    // what this pass's neighbours will deliver: the lower neighbour's upper half (as many boundary workgroups as I have tiles)
    if (c->has_lo()) c->peer_expect[0] += (unsigned long long)tab->nbump[1];
    if (c->has_hi()) c->peer_expect[1] += (unsigned long long)tab->nbump[0];
    c->hv_first ^= 1;
    return WAFER_OK;
}

// main stream: the ghost planes the last peer pass's neighbours deliver have arrived
static int peer_drain(wafer_ctx *c)
{
    for (int h = 0; h < 2; ++h) {
        if (!(h == 0 ? c->has_lo() : c->has_hi())) continue;
        hipLaunchKernelGGL(wafer_k_gate, dim3(1), dim3(64), 0, c->s_main, c->peer_flags + h * WAFER_F3_SYNC_STRIDE, c->peer_expect[h], c->hv_err,
                           hv_spins(c, 4), 1);
        HIP_TRY(hipGetLastError());
    }
    return WAFER_OK;
}

// one three-step pass of the whole slab in ONE launch; the two exchanges follow on the second stream
static int launch_halves_pass(wafer_ctx *c, int src, int dst, int E)
{
    const WaferGeom &g = c->g;
    const int lo = g.G, hi = g.G + g.nzl, mid = lo + g.nzl / 2;
    const int first = c->hv_first;
    const wafer_ctx::F3Table *tab = nullptr;
    TRY(f3_table(c, F3_HALVES, lo, hi, first, &tab));
    WaferF3Sync sy;
    sy.cnt = hv_cnt(c, 0);
    sy.flag = hv_flag(c, 0);
    sy.need[0] = c->hv_flag_epoch[0];   // every exchange enqueued so far
    sy.need[1] = c->hv_flag_epoch[1];
    sy.err = c->hv_err;
    sy.debug = c->tune.hv_debug;
    sy.max_spins = hv_spins(c, 1);
    const WaferStepArgs a = step_args(c, lo, hi);
    if (wafer_entry_step3_fused(type_combo(c, true), c->tune, a, tab->dev, tab->nblocks, sy, c->phi[src], c->v, c->phi[dst], c->s_main, tab->dir) != hipSuccess)
        return fail(WAFER_ERR_HIP, "three-step stencil launch failed: %s", hipGetErrorString(hipGetLastError()));
    c->last_instance_valid = true;
    c->hv_cnt_target[0] += (unsigned long long)tab->nbump[0];
    c->hv_cnt_target[1] += (unsigned long long)tab->nbump[1];
    for (int i = 0; i < 2; ++i) {
        const int half = (first + i) & 1;
        if (!(c->tune.hv_debug & 32)) TRY(hv_gate(c, half));
        // a half thinner than the exchange depth: its side's boundary planes reach into the other half
        if ((half == 0 ? mid - lo : hi - mid) < E) TRY(hv_gate(c, half ^ 1));
        // side 0: the lowest owned planes go down, the upper ghost planes are filled (read by half B); side 1: the mirror image
        TRY(exchange_halo_side(c, dst, c->s_aux, E, half));
        TRY(hv_post(c, half ^ 1));
        HIP_TRY(hipEventRecord(c->ev_ex[half], c->s_aux));
    }
    c->hv_first ^= 1;
    return WAFER_OK;
}

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
            if (c->sharded() && (c->overlap_mode == 2 || c->overlap_mode == 3) && K == 3 && E == H) {
                const bool peer = c->overlap_mode == 3;
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
            const bool split = c->sharded() && c->overlap_mode != 0 && g.nzl > 2 * E;
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
            const bool split = c->sharded() && c->overlap_mode != 0 && g.nzl > 2 * R;
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
                const bool split = one_pass && !last && c->sharded() && c->overlap_mode == 1 && g.nzl > 2 * R;
                if (split) {
                    TRY(excited_step_launch_overlapped(c, src, dst, wnum, one_pass));
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
    return wafer_stencil_kernel_name(c);
}

int wafer_stencil_steps_per_launch(wafer_ctx *c)
{
    if (!c) return 0;
    if (fuse3_applies(c)) return 3;
    return fuse2_applies(c) ? 2 : 1;
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
    const double lead = (R == 1) ? 2. : (R == 2) ? 24. : 360.;
    const double den = lead * c->P.dn * c->P.dn * c->P.mass; // grid.rs:314 / 337 / 367
    long long nb = 0;
    if (c->tune.obs_lds != 0) {
        // the LDS pipeline of the step kernel in its observables mode: 16 B per lane from HBM
        WaferStepArgs sa{};
        sa.g = c->g;
        sa.lz_lo = c->g.G;
        sa.lz_hi = c->g.G + c->g.nzl;
        sa.dt = c->P.dt;
        sa.den = den;
        sa.target_blocks = c->num_cus;
        sa.potsub_kind = c->potsub_kind;
        sa.potsub_scalar = c->potsub_scalar;
        set_vg_args(c, sa);
        if (wafer_entry_observables_lds(type_combo(c, false), R, c->tune, sa, c->phi[c->cur], c->v, c->potsub, c->partials, c->partials_stride,
                                        c->s_main, &nb, closed_form_vg(c)) != hipSuccess)
            return fail(WAFER_ERR_HIP, "observables launch failed: %s", hipGetErrorString(hipGetLastError()));
    } else {
        WaferObsArgs a;
        a.g = c->g;
        a.zchunk = pick_zchunk(c, c->g.nzl, direct_target_blocks(c));
        const dim3 grid(c->bx, c->by, nchunks_of(c->g.nzl, a.zchunk));
        a.nblocks = (long long)c->partials_stride;
        nb = (long long)grid.x * grid.y * grid.z;
        if ((size_t)nb > c->partials_stride) return fail(WAFER_ERR_INVALID, "partials buffer too small");
        a.den = den;
        a.potsub_kind = c->potsub_kind;
        a.potsub_scalar = c->potsub_scalar;
        TRY(dispatch(c, [&](auto t, auto, auto r) {
            using T = decltype(t);
            constexpr int RR = decltype(r)::value;
            hipLaunchKernelGGL((wafer_k_observables<T, RR>), grid, dim3(64, 4), 0, c->s_main, a,
                               as<T>(c->phi[c->cur]), as<T>(c->v), as<T>(c->potsub), c->partials);
            HIP_TRY(hipGetLastError());
            return (int)WAFER_OK;
        }));
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

// ---- solve (grid.rs:50-246) ----------------------------------------------------------
int wafer_solve_state(wafer_ctx *c, uint32_t wnum, double tolerance, uint64_t screen_update,
                      int has_max_steps, uint64_t max_steps, wafer_block_record *records,
                      size_t max_records, size_t *n_records, wafer_observables_output *final_out)
{
    if (!c) return fail(WAFER_ERR_INVALID, "null context");
    if (wnum > c->states.size()) return fail(WAFER_ERR_STATE, "wnum %u but w_store holds %zu states", wnum, c->states.size());
    uint64_t step = 0;
    double last_energy = DBL_MAX; // grid.rs:124
    size_t nrec = 0;
    bool converged = false;
    wafer_observables_t obs;
    for (;;) {
        TRY(wafer_observables(c, &obs));                    // :127
        const double norm_energy = obs.energy / obs.norm2;  // :128
        // R64 panics on NaN in the reference's debug builds (noisy_float); in release it would
        // iterate on NaNs forever.  Report it instead of spinning until max_steps.
        if (!std::isfinite(norm_energy))
            return fail(WAFER_ERR_STATE, "state %u: energy is not finite at step %llu (norm2 = %g): "
                        "the wavefunction vanished or diverged", wnum, (unsigned long long)step, obs.norm2);
        const double tau = (double)step * c->P.dt;          // :129
        TRY(wafer_normalise(c, obs.norm2));                 // :130
        if (wnum > 0) TRY(wafer_orthogonalise(c, wnum));    // :133-135
        const double diff = std::fabs(norm_energy - last_energy); // :161
        if (records && nrec < max_records) {
            records[nrec].step = step;
            records[nrec].tau = tau;
            records[nrec].obs = obs;
            records[nrec].diff = diff;
        }
        ++nrec;
        if (diff < tolerance) { // :162-192
            converged = true;
            break;
        }
        last_energy = norm_energy;                          // :194
        if (has_max_steps && step > max_steps) break;       // :211-213
        TRY(wafer_evolve(c, wnum, screen_update));          // :216
        step += screen_update;                              // :220
    }
    if (n_records) *n_records = nrec;
    if (final_out) { // output.rs:540-547
        const double r_norm = std::sqrt(obs.r2 / obs.norm2);
        final_out->state = wnum;
        final_out->energy = obs.energy / obs.norm2;
        final_out->binding_energy = (obs.energy - obs.v_infinity) / obs.norm2;
        final_out->r = r_norm;
        final_out->l_r = (double)c->P.nx / r_norm;
    }
    if (!converged) return fail(WAFER_ERR_MAX_STEP, "MaxStep: state %u did not converge within max_steps", wnum);
    return wafer_push_state(c); // :239-242
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
    return WAFER_OK;
}

int wafer_diag_div_check(wafer_ctx *c, double den, uint64_t seed, uint64_t n_operands, int lo_exp, int hi_exp,
                         uint64_t *mismatches)
{
    if (!c || !mismatches) return fail(WAFER_ERR_INVALID, "null argument");
    if (lo_exp < 0 || hi_exp > 2046 || lo_exp > hi_exp) return fail(WAFER_ERR_INVALID, "biased exponents in 0..2046");
    HIP_TRY(hipSetDevice(c->P.device));
    unsigned long long *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, sizeof *d));
    hipError_t e = hipMemsetAsync(d, 0, sizeof *d, c->s_main);
    const int per_thread = 1024, threads = 256;
    const uint64_t blocks = (n_operands + (uint64_t)per_thread * threads - 1) / ((uint64_t)per_thread * threads);
    if (e == hipSuccess && blocks > 0) {
        hipLaunchKernelGGL(wafer_k_div_check, dim3((unsigned)std::min<uint64_t>(blocks, 1u << 30)), dim3(threads), 0, c->s_main, den,
                           (unsigned long long)seed, per_thread, lo_exp, hi_exp, d);
        e = hipGetLastError();
    }
    unsigned long long h = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d, sizeof h, hipMemcpyDeviceToHost, c->s_main);
    hipError_t e2 = hipStreamSynchronize(c->s_main);
    (void)hipFree(d);
    if (e != hipSuccess || e2 != hipSuccess)
        return fail(WAFER_ERR_HIP, "division check failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    *mismatches = h;
    return WAFER_OK;
}

// ---- multi-GPU plumbing -----------------------------------------------------------------
int wafer_set_comm_hooks(wafer_ctx *c, wafer_halo_fn halo, wafer_allreduce_fn allreduce, void *user)
{
    if (!c) return fail(WAFER_ERR_INVALID, "null context");
    c->halo_hook = halo;
    c->allreduce_hook = allreduce;
    c->hook_user = user;
    return WAFER_OK;
}

int wafer_set_overlap(wafer_ctx *c, int mode)
{
    if (!c) return fail(WAFER_ERR_INVALID, "null context");
    if (mode < 0 || mode > 3) return fail(WAFER_ERR_INVALID, "overlap mode 0 .. 3");
    if (mode == 3) {
        if (c->sharded() && !c->peer_ready) return fail(WAFER_ERR_STATE, "overlap mode 3 (peer stores) needs wafer_peer_connect first");
        if (c->sharded() && c->g.nzl < 6 * c->g.R) return fail(WAFER_ERR_INVALID, "overlap mode 3 needs at least %d owned planes", 6 * c->g.R);
    }
    c->overlap_mode = mode;
    // a fresh start for the single-launch pass: every rank dispatches the lower half first again and nothing in the ghost
    // planes is taken for current (a host that has just seen WAFER_ERR_COMM on some rank calls this on all of them)
    c->hv_first = 0;
    c->halo_valid = 0;
    return WAFER_OK;
}

// drawn once per process (the by-address shortcut of wafer_peer_connect must not misfire on a pid that another PID namespace
// handed out as well)
static uint64_t process_nonce()
{
    static const uint64_t nonce = [] {
        uint64_t v = 0;
        if (FILE *f = fopen("/dev/urandom", "rb")) {
            if (fread(&v, sizeof v, 1, f) != 1) v = 0;
            fclose(f);
        }
        if (v == 0) v = ((uint64_t)getpid() << 32) ^ (uint64_t)(uintptr_t)&v ^ 0x9e3779b97f4a7c15ull;
        return v;
    }();
    return nonce;
}

int wafer_peer_export(wafer_ctx *c, wafer_peer_info *out)
{
    if (!c || !out) return fail(WAFER_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->P.device));
    TRY(ensure_hv(c));
    TRY(ensure_peer_flags(c));
    memset(out, 0, sizeof *out);
    out->struct_size = (uint32_t)sizeof *out;
    out->z_begin = (uint32_t)c->g.z_begin;
    out->z_count = (uint32_t)c->g.nzl;
    out->halo_depth = (uint32_t)c->g.G;
    out->pid = (uint64_t)getpid();
    out->process_nonce = process_nonce();
    out->device = c->P.device;
    {
        hipUUID u;
        static_assert(sizeof u.bytes == sizeof out->device_uuid, "uuid size");
        HIP_TRY(hipDeviceGetUuid(&u, c->P.device));
        memcpy(out->device_uuid, u.bytes, sizeof out->device_uuid);
    }
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
    for (int b = 0; b < 2; ++b) {
        out->phi_addr[b] = (uint64_t)(uintptr_t)c->phi[b];
        out->phi_alloc_offset[b] = (uint64_t)c->g.base_off * c->esz;
        hipIpcMemHandle_t h;
        // (a handle is only needed by another process; a runtime that cannot export one still serves neighbours in this process)
        if (hipIpcGetMemHandle(&h, alloc_base(c, c->phi[b])) == hipSuccess) memcpy(out->phi_ipc[b], &h, sizeof h);
        else (void)hipGetLastError();
    }
    out->flags_addr = (uint64_t)(uintptr_t)c->peer_flags;
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, c->peer_flags) == hipSuccess) memcpy(out->flags_ipc, &h, sizeof h);
    else (void)hipGetLastError();
    return WAFER_OK;
}

int wafer_peer_disconnect(wafer_ctx *c)
{
    if (!c) return WAFER_OK;
    for (int h = 0; h < 2; ++h) {
        for (void *&m : c->peer[h].ipc_map)
            if (m) { (void)hipIpcCloseMemHandle(m); m = nullptr; }
        c->peer[h] = wafer_ctx::PeerSide();
    }
    c->peer_ready = false;
    if (c->overlap_mode == 3) c->overlap_mode = 2;
    return WAFER_OK;
}

int wafer_peer_connect(wafer_ctx *c, const wafer_peer_info *lower, const wafer_peer_info *upper)
{
    if (!c) return fail(WAFER_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->P.device));
    if ((lower != nullptr) != c->has_lo() || (upper != nullptr) != c->has_hi())
        return fail(WAFER_ERR_INVALID, "wafer_peer_connect: a record is needed exactly for the sides that have a neighbour");
    TRY(ensure_hv(c));
    TRY(ensure_peer_flags(c));
    (void)wafer_peer_disconnect(c);
    const wafer_peer_info *rec[2] = {lower, upper};
    for (int h = 0; h < 2; ++h) {
        const wafer_peer_info *r = rec[h];
        if (!r) continue;
        if (r->struct_size != sizeof *r) return fail(WAFER_ERR_INVALID, "wafer_peer_info.struct_size mismatch");
        if ((int)r->halo_depth != c->g.G) return fail(WAFER_ERR_INVALID, "neighbour was created with another halo_depth");
        // the neighbour must own the planes next to mine
        const bool adjacent = h == 0 ? (int)(r->z_begin + r->z_count) == c->g.z_begin || (int)r->z_begin == c->g.z_begin   // (itself: a self-loop)
                                     : (int)r->z_begin == c->g.z_begin + c->g.nzl || (int)r->z_begin == c->g.z_begin;
        if (!adjacent) return fail(WAFER_ERR_INVALID, "wafer_peer_connect: the %s record is not the z-neighbour's", h == 0 ? "lower" : "upper");
        wafer_ctx::PeerSide &ps = c->peer[h];
        ps.nzl = (int)r->z_count;
        const bool same_process = r->pid == (uint64_t)getpid() && r->process_nonce == process_nonce();
        const bool self_loop = same_process && (int)r->z_begin == c->g.z_begin && r->phi_addr[0] == (uint64_t)(uintptr_t)c->phi[0];
        hipUUID mine;
        HIP_TRY(hipDeviceGetUuid(&mine, c->P.device));
        const bool same_device = memcmp(mine.bytes, r->device_uuid, sizeof mine.bytes) == 0;
        // another context on THIS device shares its CUs: a workgroup that polls for that neighbour's stores can keep the neighbour's
        // kernel from running (tests fold ranks onto one GPU and say so)
        if (same_device && !self_loop && c->tune.peer_same_device == 0)
            return fail(WAFER_ERR_INVALID, "wafer_peer_connect: the %s neighbour is another context on this device (a polling workgroup "
                                           "can starve the kernel it waits for); set WAFER_PEER_SAME_DEVICE=1 to allow it",
                        h == 0 ? "lower" : "upper");
        if (same_process) {
            if (!same_device) {
                // one process, several GPUs: the neighbour's memory must be mapped on this device before a kernel stores into it
                int can = 0;
                HIP_TRY(hipDeviceCanAccessPeer(&can, c->P.device, r->device));
                if (!can) return fail(WAFER_ERR_INVALID, "wafer_peer_connect: device %d cannot access its %s neighbour's device %d",
                                      c->P.device, h == 0 ? "lower" : "upper", r->device);
                const hipError_t pe = hipDeviceEnablePeerAccess(r->device, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled)
                    return fail(WAFER_ERR_HIP, "hipDeviceEnablePeerAccess(%d) failed: %s", r->device, hipGetErrorString(pe));
                (void)hipGetLastError();
            }
            ps.phi[0] = (void *)(uintptr_t)r->phi_addr[0];
            ps.phi[1] = (void *)(uintptr_t)r->phi_addr[1];
            ps.flags = (unsigned long long *)(uintptr_t)r->flags_addr;
        } else {
            for (int b = 0; b < 2; ++b) {
                hipIpcMemHandle_t hd;
                memcpy(&hd, r->phi_ipc[b], sizeof hd);
                HIP_TRY(hipIpcOpenMemHandle(&ps.ipc_map[b], hd, hipIpcMemLazyEnablePeerAccess));
                ps.phi[b] = static_cast<char *>(ps.ipc_map[b]) + r->phi_alloc_offset[b];
            }
            hipIpcMemHandle_t hd;
            memcpy(&hd, r->flags_ipc, sizeof hd);
            HIP_TRY(hipIpcOpenMemHandle(&ps.ipc_map[2], hd, hipIpcMemLazyEnablePeerAccess));
            ps.flags = static_cast<unsigned long long *>(ps.ipc_map[2]);
        }
        ps.connected = true;
    }
    WaferF3Peer host;
    memset(&host, 0, sizeof host);
    for (int h = 0; h < 2; ++h) {
        const wafer_ctx::PeerSide &ps = c->peer[h];
        if (!ps.connected) continue;
        host.out[h][0] = ps.phi[0];
        host.out[h][1] = ps.phi[1];
        // my planes [lo, lo + E) are the lower neighbour's upper ghost planes [G + nzl_n, ...): shift by nzl_n (lo = G);
        // my planes [hi - E, hi) are the upper neighbour's lower ghost planes [G - E, G): shift by -nzl
        host.zshift[h] = h == 0 ? (long long)ps.nzl : -(long long)c->g.nzl;
        host.flag[h] = ps.flags + (1 - h) * WAFER_F3_SYNC_STRIDE;   // what I send down fills the neighbour's UPPER side, and vice versa
    }
    if (!c->peer_dev) HIP_TRY(hipMalloc((void **)&c->peer_dev, sizeof(WaferF3Peer)));
    HIP_TRY(hipMemcpy(c->peer_dev, &host, sizeof host, hipMemcpyHostToDevice));
    c->peer_ready = true;
    return WAFER_OK;
}

int wafer_set_halo_cycle(wafer_ctx *c, int passes)
{
    if (!c) return fail(WAFER_ERR_INVALID, "null context");
    // one fused pass consumes K * ext ghost planes per side: K = 3 where the three-step kernel applies, else 2
    const int per_pass = (fuse3_applies(c) ? 3 : 2) * c->g.R;
    if (passes < 1 || per_pass * passes > c->g.G)
        return fail(WAFER_ERR_INVALID, "halo cycle %d needs %d ghost planes (%d per fused pass), the context has %d (wafer_params.halo_depth)",
                    passes, per_pass * passes, per_pass, c->g.G);
    c->halo_cycle = passes;
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
