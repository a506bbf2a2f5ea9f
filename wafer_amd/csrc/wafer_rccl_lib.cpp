// libwafer_rccl.so -- installs the RCCL hooks of wafer_rccl_hooks.h on a context from any host
// language (Python: wafer_amd.slab.NativeRcclSlabComm).  The ranks share one ncclUniqueId, made by
// rank 0 with wafer_rccl_unique_id and distributed by whatever the host uses for rendezvous.
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/wafer_rccl.h"
#include "wafer_rccl_hooks.h"

static thread_local std::string g_err;
static int fail(const char *what, ncclResult_t r)
{
    g_err = std::string(what) + ": " + ncclGetErrorString(r);
    return 1;
}

extern "C" {

const char *wafer_rccl_last_error(void) { return g_err.c_str(); }
int wafer_rccl_unique_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

int wafer_rccl_unique_id(void *out)
{
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail("ncclGetUniqueId", r);
    memcpy(out, &id, sizeof id);
    return 0;
}

// rank's z-neighbours are rank - 1 and rank + 1 (lower_override / upper_override >= 0 replace them:
// a single rank that is its own neighbour, for tests)
int wafer_rccl_attach(wafer_ctx *ctx, int rank, int world, const void *unique_id, int lower_override, int upper_override,
                      void **handle)
{
    if (!ctx || !unique_id || !handle || world < 1 || rank < 0 || rank >= world) { g_err = "bad argument"; return 1; }
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    wafer_rccl_default_env();
    WaferRcclFabric *f = new WaferRcclFabric();
    ncclResult_t r = ncclCommInitRank(&f->comm, world, id, rank);
    if (r != ncclSuccess) { delete f; return fail("ncclCommInitRank", r); }
    f->lower = lower_override >= 0 ? lower_override : (rank > 0 ? rank - 1 : -1);
    f->upper = upper_override >= 0 ? upper_override : (rank + 1 < world ? rank + 1 : -1);
    if (wafer_set_comm_hooks(ctx, wafer_rccl_halo, wafer_rccl_allreduce, f) != WAFER_OK) {
        g_err = wafer_last_error();
        ncclCommDestroy(f->comm);
        delete f;
        return 1;
    }
    *handle = f;
    return 0;
}

// one small exchange with the neighbours and one all-reduce, so that channel set-up is not part of
// any timed step; `scratch` is >= 4 KiB of device memory, `stream` any stream of the device
int wafer_rccl_warm_up(void *handle, void *scratch, void *stream)
{
    WaferRcclFabric *f = static_cast<WaferRcclFabric *>(handle);
    char *p = static_cast<char *>(scratch);
    if (wafer_rccl_halo(f, f->lower >= 0 ? p : nullptr, f->upper >= 0 ? p + 1024 : nullptr, f->lower >= 0 ? p + 2048 : nullptr,
                        f->upper >= 0 ? p + 3072 : nullptr, 256, stream) != 0) { g_err = "warm-up exchange failed"; return 1; }
    if (wafer_rccl_allreduce(f, p, 8, stream) != 0) { g_err = "warm-up all-reduce failed"; return 1; }
    f->halo_calls = f->reduce_calls = 0;
    return hipStreamSynchronize(static_cast<hipStream_t>(stream)) == hipSuccess ? 0 : 1;
}

// the all-reduce hook served by a connected wafer_mailbox (include/wafer_mailbox.h) from now on; NULL: back to ncclAllReduce
int wafer_rccl_use_mailbox(void *handle, void *mailbox)
{
    WaferRcclFabric *f = static_cast<WaferRcclFabric *>(handle);
    if (!f) { g_err = "null handle"; return 1; }
    f->mailbox = static_cast<wafer_mailbox *>(mailbox);
    return 0;
}

// one all-reduce through the hook the handle serves (mailbox if one is in use, else ncclAllReduce): for hosts that need
// a sum of their own on the same fabric, and for timing the two paths (tools/allreduce_latency.py)
int wafer_rccl_allreduce_now(void *handle, void *dev_ptr, size_t count, void *stream)
{
    if (!handle) { g_err = "null handle"; return 1; }
    return wafer_rccl_allreduce(handle, dev_ptr, count, stream);
}

long wafer_rccl_halo_calls(void *handle) { return static_cast<WaferRcclFabric *>(handle)->halo_calls; }

// what RCCL itself says about the communicator: ncclCommCount / ncclCommUserRank, the z-neighbours in
// use and RCCL's version (bench.py reports them next to the numbers they produced)
int wafer_rccl_comm_info(void *handle, int *nranks, int *rank, int *lower, int *upper, int *version)
{
    WaferRcclFabric *f = static_cast<WaferRcclFabric *>(handle);
    if (!f) { g_err = "null handle"; return 1; }
    int n = 0, r = -1, v = 0;
    ncclResult_t e = ncclCommCount(f->comm, &n);
    if (e == ncclSuccess) e = ncclCommUserRank(f->comm, &r);
    if (e == ncclSuccess) e = ncclGetVersion(&v);
    if (e != ncclSuccess) return fail("ncclCommCount / ncclCommUserRank", e);
    if (nranks) *nranks = n;
    if (rank) *rank = r;
    if (lower) *lower = f->lower;
    if (upper) *upper = f->upper;
    if (version) *version = v;
    return 0;
}

int wafer_rccl_detach(wafer_ctx *ctx, void *handle)
{
    WaferRcclFabric *f = static_cast<WaferRcclFabric *>(handle);
    if (ctx) (void)wafer_set_comm_hooks(ctx, nullptr, nullptr, nullptr);
    if (f) {
        ncclCommDestroy(f->comm);
        delete f;
    }
    return 0;
}

} // extern "C"
