// The engine's two communication hooks (include/wafer_hip.h) served by RCCL's C API: grouped
// ncclSend / ncclRecv of the boundary planes to the z-neighbours and an in-place ncclAllReduce of
// the reduction scalars, both enqueued on the hipStream_t the engine passes (nothing blocks the
// host).  Shared by the native host (wafer_rccl_host.cpp) and by libwafer_rccl.so, which lets a
// Python host install the same hooks (wafer_amd.slab.NativeRcclSlabComm).
#pragma once
#include <cstddef>
#include <cstdlib>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "../../include/wafer_mailbox.h"

// RCCL's send / recv kernels take whole CUs away from the stencil for as long as the links are busy
// (their workgroups cannot share a CU with a stencil workgroup) and RCCL launches 64 of them by
// default for the four transfers of a pass.  Eight channels still move two 1024^2 planes faster than
// a link can and leave the CUs to the interior launch.  Call before the communicator is created;
// a value chosen by the user wins.
static inline void wafer_rccl_default_env() { setenv("NCCL_MAX_P2P_NCHANNELS", "8", 0); }

struct WaferRcclFabric {
    ncclComm_t comm = nullptr;
    int lower = -1, upper = -1; // z-neighbour ranks, -1 = the global Dirichlet frame
    long halo_calls = 0, reduce_calls = 0;
    // optional: the device-side all-reduce of libwafer_hip.so (include/wafer_mailbox.h) for the path's few doubles;
    // ncclAllReduce stays the default and serves whatever the mailbox does not take
    wafer_mailbox *mailbox = nullptr;
};

static inline int wafer_rccl_halo(void *user, void *send_lo, void *send_hi, void *recv_lo, void *recv_hi, size_t bytes,
                                  void *stream)
{
    WaferRcclFabric *f = static_cast<WaferRcclFabric *>(user);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (ncclGroupStart() != ncclSuccess) return 1;
    // receives first, then sends: between one pair of ranks they match in posting order.  When ONE rank is both neighbours
    // (a ring of one or two slabs: the self-neighbour test) the sends go upper first, so that the lower ghost planes receive
    // the neighbour's upper boundary planes -- the same ring a one-direction call (send_lo + recv_hi) and the peer stores close
    const bool ring = f->lower >= 0 && f->lower == f->upper;
    bool ok = true;
    if (recv_lo) ok = ok && ncclRecv(recv_lo, bytes, ncclChar, f->lower, f->comm, s) == ncclSuccess;
    if (recv_hi) ok = ok && ncclRecv(recv_hi, bytes, ncclChar, f->upper, f->comm, s) == ncclSuccess;
    if (ring && send_hi) ok = ok && ncclSend(send_hi, bytes, ncclChar, f->upper, f->comm, s) == ncclSuccess;
    if (send_lo) ok = ok && ncclSend(send_lo, bytes, ncclChar, f->lower, f->comm, s) == ncclSuccess;
    if (!ring && send_hi) ok = ok && ncclSend(send_hi, bytes, ncclChar, f->upper, f->comm, s) == ncclSuccess;
    if (ncclGroupEnd() != ncclSuccess || !ok) return 1;
    ++f->halo_calls;
    return 0;
}

static inline int wafer_rccl_allreduce(void *user, void *dev_ptr, size_t count, void *stream)
{
    WaferRcclFabric *f = static_cast<WaferRcclFabric *>(user);
    ++f->reduce_calls;
    if (f->mailbox && count <= WAFER_MAILBOX_MAX_COUNT) return wafer_mailbox_allreduce(f->mailbox, dev_ptr, count, stream);
    return ncclAllReduce(dev_ptr, dev_ptr, count, ncclDouble, ncclSum, f->comm, static_cast<hipStream_t>(stream)) == ncclSuccess ? 0 : 1;
}
