// wafer_engine.h -- what the translation units of the host engine share: the context, the error helpers, and the functions one
// unit defines and another calls (namespace wafer_eng, hidden from the library's dynamic symbols).  The C ABI is include/wafer_hip.h.
//   wafer_engine.hip            context, arrays, potentials, wavefunction and w_store, host <-> device layout, diagnostics
//   wafer_engine_schedules.hip  which kernel advances what: variants, workgroup tables, launches, reductions, wafer_evolve, observables
//   wafer_engine_comm.hip       z-slabs: halo exchange through the hooks, the single-launch pass, peer stores
//   wafer_engine_solve.hip      grid.rs:50-246 for one state (host loop)
#pragma once
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
#include "wafer_geom.h"
#include "wafer_divplan.h"
#include "wafer_stencil.hip.h"
#include "wafer_stencil_fused3.hip.h"
#include "wafer_launch.h"
#include "wafer_tuning.h"

namespace wafer_eng __attribute__((visibility("hidden"))) {
// sets the thread's last-error message and returns `code`
int fail(int code, const char *fmt, ...);
}
using namespace wafer_eng;

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
inline Roctx &roctx()
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
    hipStream_t s_aux2 = nullptr;   // the upper side's exchanges of the single-launch pass under the copy transport (launch_halves_pass)
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
    WaferDivPlanF div_plan_f;   // ... in fp32, for WAFER_F32_FAST contexts (checked = 0 elsewhere)
    WaferDivPlan div_plan;   // x / (c dn^2 m) in the step kernels (wafer_divplan.h), made once, at wafer_ctx_create
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
    int overlap_mode = 2;   // wafer_set_overlap: 0 exchange after the pass, 1 boundary-first split pass, 2 single-launch half-slab pass,
                            // 3 peer stores, 4 = COPY transport (below) under the schedule `sched`
    int sched = 2;          // the schedule wafer_evolve follows: overlap_mode, except in mode 4 (tune.copy_sched: 2, 1 or 0)
    // Overlap mode 4: every exchange of phi's ghost planes is a device copy (hipMemcpyAsync: the copy engines between GPUs) from
    // this rank's boundary planes INTO the z-neighbour's ghost planes through the mapping wafer_peer_connect holds, instead of
    // the halo hook.  A rendezvous of four words per link replaces the two-sided semantics of send / recv (wafer_engine_comm.hip,
    // copy_exchange): cp_sent[n] / cp_recv[n] count the exchanges sent to / received from the lower (0) / upper (1) neighbour.
    bool halo_copy = false;
    unsigned long long cp_sent[2] = {0, 0}, cp_recv[2] = {0, 0};
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


// denominators of grid.rs:569 / 594 / 626 (and :314 / 337 / 367)
static inline double wafer_stencil_den(int R, double dn, double mass)
{
    const double lead = (R == 1) ? 2. : (R == 2) ? 24. : 360.;
    return lead * dn * dn * mass;
}
// WaferStepArgs::v_in_range: the kernels may take the short arithmetic forms (wafer_recip for b, three instructions for x / den)
static inline bool short_forms(const wafer_ctx *c) { return c->v_in_range && c->div_plan.checked != 0 && (!c->f32_arith || c->div_plan_f.checked != 0); }
static inline void set_den_args(const wafer_ctx *c, WaferStepArgs &a)
{
    a.den = c->div_plan.den;
    a.den_zh = c->div_plan.zh;
    a.den_zl = c->div_plan.zl;
    a.den_zh_f = c->div_plan_f.checked ? c->div_plan_f.zh : 0.f;
    a.den_zl_f = c->div_plan_f.checked ? c->div_plan_f.zl : 0.f;
}
static inline int nchunks_of(int nplanes, int zchunk) { return (nplanes + zchunk - 1) / zchunk; }
template <typename T>
static inline T *as(void *p) { return static_cast<T *>(p); }
template <typename T>
static inline const T *as(const void *p) { return static_cast<const T *>(p); }
// Arrays are held as LOGICAL pointers to (plane 0, row 0); the allocation starts
// base_off elements earlier (guard planes / rows, wafer_geom.h).
static inline void *alloc_base(const wafer_ctx *c, void *logical)
{
    return logical ? static_cast<char *>(logical) - (size_t)c->g.base_off * c->esz : nullptr;
}
static inline bool kernels_stream_ab(const wafer_ctx *c, int variant) { return variant == 0 || c->tune.abv == 0; }

namespace wafer_eng __attribute__((visibility("hidden"))) {
// ---- wafer_engine.hip
int alloc_grid_array(wafer_ctx *c, void **logical, hipStream_t s);
int ensure_ab(wafer_ctx *c);
int refresh_ab(wafer_ctx *c);
int check_v_range(wafer_ctx *c);
// ---- wafer_engine_schedules.hip
enum { X2_SUM_SLOT = 18 };   // scal[18 .. 18 + 1 + 2k): the sums of a two-step pass
enum { F3_PLAIN = 0, F3_MIXED = 1, F3_HALVES = 2, F3_WHOLE = 3 };
int reduce_to_scal(wafer_ctx *c, int nq, long long n, int slot, hipStream_t s);
int read_scal(wafer_ctx *c, int slot, int n, double *out, hipStream_t s);
int active_variant(const wafer_ctx *c);
int default_variant(const wafer_ctx *c);
const char *variant_name(int v);
int closed_form_vg(const wafer_ctx *c);
int type_combo(const wafer_ctx *c, bool step_kernel);
WaferStepArgs step_args(const wafer_ctx *c, int lz_lo, int lz_hi);
bool fuse3_applies(const wafer_ctx *c);
bool fuse2_applies(const wafer_ctx *c);
int f3_table(wafer_ctx *c, int kind, int lz_lo, int lz_hi, int aux, const wafer_ctx::F3Table **out);
int launch_dot(wafer_ctx *c, void *phi, void *lower, int out_slot, hipStream_t s);
int recompute_gram(wafer_ctx *c);
// ---- wafer_engine_comm.hip
int exchange_halo_array(wafer_ctx *c, void *array, hipStream_t s, int planes);
int exchange_halo(wafer_ctx *c, int buf, hipStream_t s, int planes);
int exchange_halo_side(wafer_ctx *c, int buf, hipStream_t s, int planes, int side);
int copy_exchange(wafer_ctx *c, int buf, hipStream_t s, int planes, bool send_lo, bool send_hi, bool recv_lo, bool recv_hi);
int ensure_halo(wafer_ctx *c, int need);
int ensure_hv(wafer_ctx *c);
int check_hv_err(wafer_ctx *c);
int launch_peer_pass(wafer_ctx *c, int src, int dst, int E);
int launch_halves_pass(wafer_ctx *c, int src, int dst, int E);
int peer_drain(wafer_ctx *c);
}
