// wafer_mailbox.hip -- device-side all-reduce of a few doubles through peer-mapped mailboxes (include/wafer_mailbox.h).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/wafer_hip.h"
#include "../../include/wafer_mailbox.h"

namespace {
enum { SLOT = 16 };   // doubles per (parity, sender): values [0, 14), word 15 = the epoch; two 64-byte lines

struct MailboxDev {
    double *peer[WAFER_MAILBOX_MAX_RANKS];   // every rank's mailbox as mapped here (peer[rank] is this rank's own)
    unsigned *err;        // host memory: 1 + the rank whose contribution never arrived
    unsigned *dead;       // device memory: sticky, set with err
    unsigned max_spins;   // bound of one wait (2^26 spins: several seconds)
    int rank, world;
};
} // namespace

struct wafer_mailbox {
    MailboxDev d{};
    void *own = nullptr;
    void *mapped[WAFER_MAILBOX_MAX_RANKS] = {nullptr};
    unsigned long long epoch = 0;
    int device = 0;
    bool connected = false;
};

extern void wafer_set_last_error(const char *msg);   // wafer_engine.hip

static int mb_fail(const char *what, hipError_t e)
{
    char buf[256];
    snprintf(buf, sizeof buf, "wafer_mailbox: %s: %s", what, hipGetErrorString(e));
    wafer_set_last_error(buf);
    (void)hipGetLastError();
    return WAFER_ERR_HIP;
}

// A wait that gives up must not leave a plausible number behind: the call's results become NaN on this rank, a sticky
// device word (`dead`) makes every later call post and return NaN at once (peers then see NaN instead of timing out one
// by one, and the epoch protocol cannot pair a late contribution with the wrong call), and the host word `err` makes
// wafer_mailbox_allreduce / wafer_mailbox_check report WAFER_ERR_COMM from then on.
__global__ __launch_bounds__(64) void wafer_k_mailbox_allreduce(MailboxDev m, double *__restrict__ data, int n, unsigned long long epoch)
{
    __shared__ double vals[WAFER_MAILBOX_MAX_RANKS][WAFER_MAILBOX_MAX_COUNT];
    __shared__ unsigned gave_up;
    const int lane = threadIdx.x;
    const int par = (int)(epoch & 1);
    const bool dead = __hip_atomic_load(m.dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    if (lane == 0) gave_up = dead ? 1u : 0u;
    __syncthreads();
    if (lane < m.world) {
        // my values, then the epoch behind a release, into rank `lane`'s mailbox (my own included)
        double *dst = m.peer[lane] + (size_t)(par * m.world + m.rank) * SLOT;
        for (int q = 0; q < n; ++q) __hip_atomic_store(dst + q, dead ? __builtin_nan("") : data[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(dst + SLOT - 1), epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (!dead) {
            // every sender's epoch in my own mailbox, then its values
            const double *src = m.peer[m.rank] + (size_t)(par * m.world + lane) * SLOT;
            unsigned spins = 0;
            while (__hip_atomic_load(reinterpret_cast<const unsigned long long *>(src + SLOT - 1), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != epoch) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > m.max_spins) {
                    __hip_atomic_store(m.err, 1u + (unsigned)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(m.dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    break;
                }
            }
            for (int q = 0; q < n; ++q) vals[lane][q] = __hip_atomic_load(src + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __syncthreads();
    if (lane < n) {
        double s = 0.0;
        for (int r = 0; r < m.world; ++r) s += vals[r][lane];   // rank order: the same bits on every rank
        data[lane] = gave_up ? __builtin_nan("") : s;
    }
}

extern "C" {

int wafer_mailbox_create(int rank, int world, int device, wafer_mailbox **out)
{
    if (!out || world < 1 || world > WAFER_MAILBOX_MAX_RANKS || rank < 0 || rank >= world) {
        wafer_set_last_error("wafer_mailbox_create: bad argument (world <= 16)");
        return WAFER_ERR_INVALID;
    }
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return mb_fail("hipSetDevice", e);
    wafer_mailbox *mb = new wafer_mailbox();
    mb->device = device;
    mb->d.rank = rank;
    mb->d.world = world;
    const size_t bytes = sizeof(double) * (2 * WAFER_MAILBOX_MAX_RANKS * SLOT + 8);   // + one line for the sticky word
    // fine-grained device memory where the runtime offers it to IPC (coherent for peers without a cache flush);
    // every access in the kernel is a system-scope atomic either way
    e = hipExtMallocWithFlags(&mb->own, bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipMalloc(&mb->own, bytes);
    }
    if (e != hipSuccess) { delete mb; return mb_fail("hipMalloc", e); }
    e = hipMemset(mb->own, 0, bytes);
    if (e == hipSuccess) e = hipHostMalloc((void **)&mb->d.err, 64, hipHostMallocCoherent | hipHostMallocMapped);
    if (e != hipSuccess) { (void)hipFree(mb->own); delete mb; return mb_fail("set-up", e); }
    *mb->d.err = 0;
    // the sticky word sits behind the two parity buffers of the same allocation
    mb->d.dead = reinterpret_cast<unsigned *>(static_cast<double *>(mb->own) + 2 * WAFER_MAILBOX_MAX_RANKS * SLOT);
    mb->d.max_spins = 1u << 26;
    mb->d.peer[rank] = static_cast<double *>(mb->own);
    if (world == 1) mb->connected = true;
    *out = mb;
    return WAFER_OK;
}

int wafer_mailbox_handle(wafer_mailbox *mb, void *handle_out)
{
    if (!mb || !handle_out) { wafer_set_last_error("wafer_mailbox_handle: null argument"); return WAFER_ERR_INVALID; }
    static_assert(sizeof(hipIpcMemHandle_t) == WAFER_MAILBOX_HANDLE_BYTES, "handle size");
    hipIpcMemHandle_t h;
    hipError_t e = hipSetDevice(mb->device);
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, mb->own);
    if (e != hipSuccess) return mb_fail("hipIpcGetMemHandle", e);
    memcpy(handle_out, &h, sizeof h);
    return WAFER_OK;
}

int wafer_mailbox_connect(wafer_mailbox *mb, const void *all_handles)
{
    if (!mb || !all_handles) { wafer_set_last_error("wafer_mailbox_connect: null argument"); return WAFER_ERR_INVALID; }
    hipError_t e = hipSetDevice(mb->device);
    if (e != hipSuccess) return mb_fail("hipSetDevice", e);
    for (int r = 0; r < mb->d.world; ++r) {
        if (r == mb->d.rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, static_cast<const char *>(all_handles) + (size_t)r * WAFER_MAILBOX_HANDLE_BYTES, sizeof h);
        e = hipIpcOpenMemHandle(&mb->mapped[r], h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return mb_fail("hipIpcOpenMemHandle", e);
        mb->d.peer[r] = static_cast<double *>(mb->mapped[r]);
    }
    mb->connected = true;
    return WAFER_OK;
}

int wafer_mailbox_allreduce(void *mailbox, void *dev_ptr, size_t count, void *hip_stream)
{
    wafer_mailbox *mb = static_cast<wafer_mailbox *>(mailbox);
    if (!mb || !mb->connected || !dev_ptr || count < 1 || count > WAFER_MAILBOX_MAX_COUNT) {
        wafer_set_last_error("wafer_mailbox_allreduce: not connected, or count outside 1..14");
        return 1;
    }
    // an earlier call gave up waiting (its results were NaN): the failure is sticky and every later call says so
    if (wafer_mailbox_check(mb) != WAFER_OK) return 1;
    ++mb->epoch;
    hipLaunchKernelGGL(wafer_k_mailbox_allreduce, dim3(1), dim3(64), 0, static_cast<hipStream_t>(hip_stream), mb->d,
                       static_cast<double *>(dev_ptr), (int)count, mb->epoch);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int wafer_mailbox_check(wafer_mailbox *mb)
{
    if (!mb) return WAFER_OK;
    if (*mb->d.err != 0) {
        char buf[160];
        snprintf(buf, sizeof buf, "wafer_mailbox: rank %d gave up waiting for rank %u's contribution (all-reduce %llu)", mb->d.rank,
                 *mb->d.err - 1, mb->epoch);
        wafer_set_last_error(buf);   // sticky: the mailbox is unusable from here on (its results are NaN)
        return WAFER_ERR_COMM;
    }
    return WAFER_OK;
}

int wafer_mailbox_destroy(wafer_mailbox *mb)
{
    if (!mb) return WAFER_OK;
    (void)hipSetDevice(mb->device);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < mb->d.world; ++r)
        if (mb->mapped[r]) (void)hipIpcCloseMemHandle(mb->mapped[r]);
    if (mb->own) (void)hipFree(mb->own);
    if (mb->d.err) (void)hipHostFree(mb->d.err);
    delete mb;
    return WAFER_OK;
}

} // extern "C"
