// wafer_engine_comm.hip -- the run over z-slabs: halo exchange through the host's hooks, the single-launch pass of overlap mode 2
// (gate / post kernels, flags and counters), peer stores (overlap mode 3: export, connect, the pass, the drain), and the entry
// points that configure them.
#include "wafer_engine.h"

namespace wafer_eng __attribute__((visibility("hidden"))) {
// ---------------------------------------------------------------------------
// halo exchange through the host-installed hook
// ---------------------------------------------------------------------------
// the first / last `planes` owned planes of any grid array (logical pointer) to the z-neighbours' ghost planes
int exchange_halo_array(wafer_ctx *c, void *array, hipStream_t s, int planes)
{
    if (!c->sharded()) return WAFER_OK;
    // overlap mode 4: phi's ghost planes travel as device copies into the neighbours' buffers (the stored states and their images,
    // exchanged once per change of w_store, keep the hook: the peers map each other's two phi buffers only)
    if (c->halo_copy && (array == c->phi[0] || array == c->phi[1]))
        return copy_exchange(c, array == c->phi[0] ? 0 : 1, s, planes, c->has_lo(), c->has_hi(), c->has_lo(), c->has_hi());
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

int exchange_halo(wafer_ctx *c, int buf, hipStream_t s, int planes) { return exchange_halo_array(c, c->phi[buf], s, planes); }

// One direction of the exchange (the single-launch pass of wafer_set_overlap modes 2 and 4).  side 0: the LOWEST owned planes go to the lower
// neighbour, the upper neighbour's lowest planes arrive in the UPPER ghost planes; side 1: the mirror image.  Every
// rank calls the same side at the same point of a pass, so the sends and receives pair up.
int exchange_halo_side(wafer_ctx *c, int buf, hipStream_t s, int planes, int side)
{
    if (c->halo_copy)
        return copy_exchange(c, buf, s, planes, side == 0 && c->has_lo(), side == 1 && c->has_hi(), side == 1 && c->has_lo(), side == 0 && c->has_hi());
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
int ensure_halo(wafer_ctx *c, int need)
{
    if (c->sharded() && c->halo_valid < need) {
        TRY(exchange_halo(c, c->cur, c->s_main, need));
        c->halo_valid = need;
    }
    return WAFER_OK;
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

int ensure_hv(wafer_ctx *c)
{
    if (c->hv_words) return WAFER_OK;
    HIP_TRY(hipHostMalloc((void **)&c->hv_err, 64, hipHostMallocCoherent | hipHostMallocMapped));
    *c->hv_err = 0;
    HIP_TRY(hipMalloc((void **)&c->hv_words, 4 * 64));
    HIP_TRY(hipMemset(c->hv_words, 0, 4 * 64));
    HIP_TRY(hipDeviceSynchronize());   // (the engine's streams are non-blocking: they do not wait for the null stream's memset)
    return WAFER_OK;
}
unsigned long long *hv_cnt(wafer_ctx *c, int half) { return c->hv_words + half * WAFER_F3_SYNC_STRIDE; }
unsigned long long *hv_flag(wafer_ctx *c, int side) { return c->hv_words + (2 + side) * WAFER_F3_SYNC_STRIDE; }

// WAFER_HV_WAIT_MS as a spin count (one spin = s_sleep 32 + a poll, about a microsecond)
unsigned hv_spins(const wafer_ctx *c, int mul)
{
    const long long n = (long long)c->tune.hv_wait_ms * 1000 * mul;
    return (unsigned)(n < 0xffffffffll ? n : 0xffffffffll);
}

// exchange stream: wait until every workgroup of `half` of the current launch has finished
int hv_gate(wafer_ctx *c, int half, hipStream_t sx)
{
    hipLaunchKernelGGL(wafer_k_gate, dim3(1), dim3(64), 0, sx, hv_cnt(c, half), c->hv_cnt_target[half], c->hv_err, hv_spins(c, 4), 0);
    HIP_TRY(hipGetLastError());
    return WAFER_OK;
}
// exchange stream: ghost side g has been filled once more (in stream order behind the exchange: its kernels have
// completed, their writes are visible device-wide)
int hv_post(wafer_ctx *c, int g, hipStream_t sx)
{
    const unsigned long long v = ++c->hv_flag_epoch[g];
    hipLaunchKernelGGL(wafer_k_post, dim3(1), dim3(64), 0, sx, hv_flag(c, g), v);
    HIP_TRY(hipGetLastError());
    return WAFER_OK;
}

int check_hv_err(wafer_ctx *c)
{
    if (c->hv_err && *c->hv_err != 0) {
        const unsigned e = *c->hv_err;
        *c->hv_err = 0;
        return fail(WAFER_ERR_COMM, "single-launch slab pass: a %s gave up waiting (halo exchange never completed)",
                    e == 2 ? "gate kernel" : "workgroup");
    }
    return WAFER_OK;
}

// ---- peer stores (wafer_set_overlap mode 3) and peer copies (mode 4): the words a neighbour writes ------------------------
// One allocation of six 64-byte lines per context, mapped by both z-neighbours (wafer_peer_info::flags_addr / flags_ipc):
//   line 0, 1   mode 3: workgroups arrived in the lower / upper ghost planes
//   line 2, 3   mode 4: exchanges whose copy has LANDED in the lower / upper ghost planes (written by that side's neighbour, behind its copy)
//   line 4, 5   mode 4: exchanges for which the lower / upper neighbour has declared ITS ghost planes facing this rank free to be overwritten
enum { PEER_WORD_ARRIVED = 2, PEER_WORD_CREDIT = 4, PEER_FLAG_LINES = 6 };
int ensure_peer_flags(wafer_ctx *c)
{
    if (c->peer_flags) return WAFER_OK;
    // fine-grained where the runtime offers it (coherent for peers without cache maintenance); every access is a system-scope atomic
    void *p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, PEER_FLAG_LINES * 64, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        HIP_TRY(hipMalloc(&p, PEER_FLAG_LINES * 64));
    }
    HIP_TRY(hipMemset(p, 0, PEER_FLAG_LINES * 64));
    HIP_TRY(hipDeviceSynchronize());   // (as in ensure_hv)
    c->peer_flags = static_cast<unsigned long long *>(p);
    return WAFER_OK;
}

// ---- peer copies (wafer_set_overlap mode 4) ----------------------------------------------------------------------------------
// The halo hook's contract is two-sided: a rank's ghost planes are overwritten only once it has posted the receive, and the call
// returns (in stream order) with its own ghost planes filled.  A copy INTO the neighbour's memory has neither property by itself,
// so every exchange is a rendezvous over the words above, all of it enqueued on the stream the hook would have been called with:
//   1. one one-wave kernel: for every side I receive on, tell that neighbour "receive number k posted" (credit word in ITS memory);
//      THEN, for every side I send to, wait for that neighbour's credit number k (polling my own memory);
//   2. copy my boundary planes into the neighbours' ghost planes (hipMemcpyAsync: a copy engine between GPUs, no CU of either);
//   3. one one-wave kernel: tell each neighbour "copy number k has landed" (arrival word in its memory; stream order puts the store
//      behind the completed copies), then wait for the arrivals on the sides I receive on.
// Every rank grants before it waits, so a chain of ranks cannot lock up; the counts are per link and direction and only ever grow.
// What follows on the stream starts after the arrival and, being another kernel, sees the copied planes the way any kernel sees a
// completed memcpy; the workgroups of a single-launch pass that is already RUNNING read them behind their flag wait and a
// system-scope acquire fence, as they read what RCCL's receive kernel or a neighbour's peer stores wrote (schedule 2; schedules
// 1 and 0 -- WAFER_COPY_SCHED -- start every reader after the copy).  Waits are bounded like the pass's own (WAFER_ERR_COMM).
// One wave does a whole step of the rendezvous: up to two stores ("receive k posted" / "copy k landed" in the neighbours' memory) and
// then up to two bounded waits on this rank's own words.  (One kernel per store and per wait it had been: ten launches and two
// copies per two-sided exchange, each launch a few microseconds of the chain -- profiles/r06_slab_mode4_timeline.txt.)
struct WaferRendezvous {
    unsigned long long *set_word[2] = {nullptr, nullptr};
    unsigned long long set_value[2] = {0, 0};
    const unsigned long long *wait_word[2] = {nullptr, nullptr};
    unsigned long long wait_value[2] = {0, 0};
};
__global__ __launch_bounds__(64) void wafer_k_rendezvous(WaferRendezvous r, unsigned *err, unsigned max_spins)
{
    if (threadIdx.x != 0) return;
    // (relaxed stores: what a word announces -- a completed copy, or nothing at all -- precedes this kernel in stream order; a release
    //  fence here would write back the L2 lines of whatever stencil workgroups share the XCD)
    for (int i = 0; i < 2; ++i)
        if (r.set_word[i]) __hip_atomic_store(r.set_word[i], r.set_value[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    bool waited = false;
    for (int i = 0; i < 2; ++i) {
        if (!r.wait_word[i]) continue;
        waited = true;
        unsigned spins = 0;
        while (__hip_atomic_load(r.wait_word[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < r.wait_value[i]) {
            __builtin_amdgcn_s_sleep(32);
            if (++spins > max_spins) {   // (as wafer_k_gate: the host reports WAFER_ERR_COMM at its next synchronisation)
                __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    if (waited) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // what the neighbour's copy wrote is visible to what follows in the stream
}

int copy_exchange(wafer_ctx *c, int buf, hipStream_t s, int planes, bool send_lo, bool send_hi, bool recv_lo, bool recv_hi)
{
    if (!send_lo && !send_hi && !recv_lo && !recv_hi) return WAFER_OK;
    if (!c->peer_ready) return fail(WAFER_ERR_STATE, "overlap mode 4 (peer copies) needs wafer_peer_connect first");
    RoctxRange range_("wafer_halo_exchange_copy");
    const WaferGeom &g = c->g;
    if (planes > g.G || planes > g.nzl) return fail(WAFER_ERR_INVALID, "halo exchange deeper than the slab allows");
    const size_t plane_b = (size_t)g.plane * c->esz;
    // from row 0 of the first plane to the last padded row of the last plane (guard rows in between ride along)
    const size_t bytes = ((size_t)(planes - 1) * (size_t)g.plane + (size_t)g.py * (size_t)g.pitch) * c->esz;
    char *mine = static_cast<char *>(c->phi[buf]);
    const bool recv[2] = {recv_lo, recv_hi}, send[2] = {send_lo, send_hi};
    auto launch = [&](const WaferRendezvous &r) -> int {
        hipLaunchKernelGGL(wafer_k_rendezvous, dim3(1), dim3(64), 0, s, r, c->hv_err, hv_spins(c, 4));
        HIP_TRY(hipGetLastError());
        return WAFER_OK;
    };
    // the neighbour on side n knows this rank as ITS neighbour on side 1 - n
    WaferRendezvous pre, post;
    for (int n = 0; n < 2; ++n) {
        if (recv[n]) {   // "my receive number k is posted": the credit word facing me in the neighbour's memory
            pre.set_word[n] = c->peer[n].flags + (PEER_WORD_CREDIT + (1 - n)) * WAFER_F3_SYNC_STRIDE;
            pre.set_value[n] = ++c->cp_recv[n];
        }
        if (send[n]) {   // ... and its credit for what I am about to send
            pre.wait_word[n] = c->peer_flags + (PEER_WORD_CREDIT + n) * WAFER_F3_SYNC_STRIDE;
            pre.wait_value[n] = c->cp_sent[n] + 1;
        }
    }
    TRY(launch(pre));
    for (int n = 0; n < 2; ++n) {
        if (!send[n]) continue;
        // my lowest planes fill the lower neighbour's UPPER ghost planes [G + nzl_n, ...); my highest its upper neighbour's LOWER [G - planes, G)
        const char *src = mine + (size_t)(n == 0 ? g.G : g.G + g.nzl - planes) * plane_b;
        char *dst = static_cast<char *>(c->peer[n].phi[buf]) + (size_t)(n == 0 ? g.G + c->peer[n].nzl : g.G - planes) * plane_b;
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
        post.set_word[n] = c->peer[n].flags + (PEER_WORD_ARRIVED + (1 - n)) * WAFER_F3_SYNC_STRIDE;   // "copy number k has landed"
        post.set_value[n] = ++c->cp_sent[n];
    }
    for (int n = 0; n < 2; ++n)
        if (recv[n]) {
            post.wait_word[n] = c->peer_flags + (PEER_WORD_ARRIVED + n) * WAFER_F3_SYNC_STRIDE;
            post.wait_value[n] = c->cp_recv[n];
        }
    TRY(launch(post));
    return WAFER_OK;
}

// The same single launch as launch_halves_pass, but the boundary workgroups deliver their planes themselves (WaferF3Sync::peer).
// need[h]: the arrivals promised to ghost side h by all earlier passes of this context's life; a pass adds one per tile and side.
int launch_peer_pass(wafer_ctx *c, int src, int dst, int E)
{
    const WaferGeom &g = c->g;
    const int lo = g.G, hi = g.G + g.nzl;
    (void)E;
    const int first = c->hv_first;
    const wafer_ctx::F3Table *tab = nullptr;
    // Whole columns (no cut) where every CU gets a tile of its own and the slab is no thicker than 384 planes; else the two halves
    // (twice the workgroups, columns half as long: over long columns the workgroups drift apart and lose each other's halo rows in
    // the L2 -- a 1024 x 1024 x 512 slab: 0.896 ms per step in halves against 0.927 whole, 0.852 undecomposed; at 128 planes whole
    // wins, 0.235 against 0.241).  The tile count follows nx, ny only; the thickness may differ between ranks (uneven partitions),
    // and ranks may then take different layouts: either one delivers one arrival per tile and side and pass, and waits for as many
    // (test_peer_store_pass_uneven_slabs_bit_exact, layout 5).  WAFER_HV_LAYOUT=3 forces the halves, 4 whole columns, 5 = by the
    // thickness alone (tests).
    int tx_, ty_;
    wafer_step3_tile(type_combo(c, true), &tx_, &ty_);
    const long long ntiles = (long long)((g.nx + tx_ - 1) / tx_) * ((g.ny + ty_ - 1) / ty_);
    const bool whole = c->tune.hv_layout == 4 || (c->tune.hv_layout != 3 && (ntiles >= c->num_cus || c->tune.hv_layout == 5) && g.nzl <= c->tune.hv_whole_max);
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
int peer_drain(wafer_ctx *c)
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
int launch_halves_pass(wafer_ctx *c, int src, int dst, int E)
{
    const WaferGeom &g = c->g;
    const int lo = g.G, hi = g.G + g.nzl, mid = lo + g.nzl / 2;
    const int first = c->hv_first;
    const wafer_ctx::F3Table *tab = nullptr;
    // (copy transport: no exchange kernel to hand CUs to, so no column is cut short -- aux bit 32)
    TRY(f3_table(c, F3_HALVES, lo, hi, first | (c->halo_copy ? 32 : 0), &tab));
    WaferF3Sync sy;
    sy.cnt = hv_cnt(c, 0);
    sy.flag = hv_flag(c, 0);
    sy.need[0] = c->hv_flag_epoch[0];   // every exchange enqueued so far
    sy.need[1] = c->hv_flag_epoch[1];
    sy.err = c->hv_err;
    sy.debug = c->tune.hv_debug;
    sy.max_spins = hv_spins(c, 1);
    const WaferStepArgs a = step_args(c, lo, hi);
    // A half thinner than the pass is deep (slabs of fewer than six planes) reads across the other half into the OTHER side's ghost
    // planes, and a workgroup waits for one flag only, its own side's: such a slab launches its pass behind both of the previous
    // pass's exchanges.  (Found in round 6 by the peer copies, the first transport of this pass that is asynchronous in the
    // one-process tests; with RCCL the planes had always arrived long before.)  Local: the exchanges themselves stay where they are.
    if (mid - lo < E || hi - mid < E) {
        HIP_TRY(hipStreamWaitEvent(c->s_main, c->ev_ex[0], 0));
        HIP_TRY(hipStreamWaitEvent(c->s_main, c->ev_ex[1], 0));
    }
    if (wafer_entry_step3_fused(type_combo(c, true), c->tune, a, tab->dev, tab->nblocks, sy, c->phi[src], c->v, c->phi[dst], c->s_main, tab->dir) != hipSuccess)
        return fail(WAFER_ERR_HIP, "three-step stencil launch failed: %s", hipGetErrorString(hipGetLastError()));
    c->last_instance_valid = true;
    c->hv_cnt_target[0] += (unsigned long long)tab->nbump[0];
    c->hv_cnt_target[1] += (unsigned long long)tab->nbump[1];
    for (int i = 0; i < 2; ++i) {
        const int half = (first + i) & 1;
        // The two sides' exchanges go to different neighbours over different links.  Under the copy transport each side has a stream of its
        // own, so that the second half's copy does not queue behind the first's (26.6 MB per side and pass: a few hundred microseconds on
        // a link, two of them in a row are most of a pass; profiles/r06_slab_mode4_timeline.txt): a side's chain of passes stays in order on
        // its stream, the per-link counts are disjoint, and a flag still implies that this rank's copy of the planes about to be overwritten
        // has landed -- the neighbour that posts it has consumed them.  The hook transports (mode 2) keep the one stream they were proved on.
        const hipStream_t sx = (c->halo_copy && half == 1) ? c->s_aux2 : c->s_aux;
        if (!(c->tune.hv_debug & 32)) TRY(hv_gate(c, half, sx));
        // a half thinner than the exchange depth: its side's boundary planes reach into the other half
        if ((half == 0 ? mid - lo : hi - mid) < E) TRY(hv_gate(c, half ^ 1, sx));
        // side 0: the lowest owned planes go down, the upper ghost planes are filled (read by half B); side 1: the mirror image
        TRY(exchange_halo_side(c, dst, sx, E, half));
        TRY(hv_post(c, half ^ 1, sx));
        HIP_TRY(hipEventRecord(c->ev_ex[half], sx));
    }
    c->hv_first ^= 1;
    return WAFER_OK;
}

} // namespace wafer_eng

extern "C" {

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
    if (mode < 0 || mode > 6) return fail(WAFER_ERR_INVALID, "overlap mode 0 .. 6");
    if (mode == 3) {
        if (c->sharded() && !c->peer_ready) return fail(WAFER_ERR_STATE, "overlap mode 3 (peer stores) needs wafer_peer_connect first");
        if (c->sharded() && c->g.nzl < 6 * c->g.R) return fail(WAFER_ERR_INVALID, "overlap mode 3 needs at least %d owned planes", 6 * c->g.R);
    }
    if (mode >= 4) {
        if (c->sharded() && !c->peer_ready) return fail(WAFER_ERR_STATE, "overlap modes 4 .. 6 (peer copies) need wafer_peer_connect first");
        HIP_TRY(hipSetDevice(c->P.device));
        TRY(ensure_hv(c));   // the bounded waits report through hv_err
    }
    c->overlap_mode = mode;
    // 4: peer copies under the single launch (WAFER_COPY_SCHED overrides the schedule); 5 / 6: peer copies under mode 1's / mode 0's
    // launches -- every kernel that reads ghost planes starts after the copy that filled them has completed
    c->halo_copy = mode >= 4 && c->sharded();
    c->sched = mode == 4 ? c->tune.copy_sched : mode == 5 ? 1 : mode == 6 ? 0 : mode;
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
    // a fresh export starts the peer copies' rendezvous (mode 4) from zero on every link of this rank: the host exports on all
    // ranks, carries the records around and connects -- nothing is in flight in between.  (After a WAFER_ERR_COMM under mode 4
    // the counts of a link's two ends may differ: export and connect again.)
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(c->peer_flags + PEER_WORD_ARRIVED * WAFER_F3_SYNC_STRIDE, 0, (PEER_FLAG_LINES - PEER_WORD_ARRIVED) * 64));
    HIP_TRY(hipDeviceSynchronize());   // the zeros are in memory before any neighbour learns the address
    c->cp_sent[0] = c->cp_sent[1] = c->cp_recv[0] = c->cp_recv[1] = 0;
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
    if (c->overlap_mode >= 3) {
        c->overlap_mode = c->sched = 2;
        c->halo_copy = false;
    }
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

} // extern "C"
