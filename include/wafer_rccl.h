/* libwafer_rccl.so -- the engine's two communication hooks (wafer_hip.h, wafer_set_comm_hooks)
 * served by RCCL's C API: grouped ncclSend / ncclRecv of the boundary planes to the z-neighbours and
 * an in-place ncclAllReduce of the reduction scalars, enqueued on the hipStream_t the engine passes.
 * The reference is a single-process program and has no counterpart; this is what its Rust host
 * would link for the multi-GPU leg.  Source: wafer_amd/csrc/wafer_rccl_lib.cpp,
 * wafer_rccl_hooks.h.  All functions return 0 on success. */
#ifndef WAFER_RCCL_H
#define WAFER_RCCL_H

#include "wafer_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

const char *wafer_rccl_last_error(void);
/* sizeof(ncclUniqueId); rank 0 fills a buffer of that size and hands it to every rank */
int wafer_rccl_unique_id_bytes(void);
int wafer_rccl_unique_id(void *out);
/* ncclCommInitRank + wafer_set_comm_hooks.  The z-neighbours of `rank` are rank - 1 and rank + 1;
 * lower_override / upper_override >= 0 replace them (a single rank that is its own neighbour: tests). */
int wafer_rccl_attach(wafer_ctx *ctx, int rank, int world, const void *unique_id, int lower_override, int upper_override,
                      void **handle);
/* one small exchange and all-reduce so that channel set-up is outside any timed step;
 * scratch: >= 4 KiB of device memory, stream: any stream of the device */
int wafer_rccl_warm_up(void *handle, void *scratch, void *stream);
/* from now on the all-reduce hook is served by a connected wafer_mailbox (wafer_mailbox.h: device-side, no RCCL kernel on
 * the critical path of an excited-state step); NULL restores ncclAllReduce */
int wafer_rccl_use_mailbox(void *handle, void *mailbox);
/* one in-place all-reduce of `count` doubles through whatever serves the hook (mailbox or ncclAllReduce), enqueued on stream */
int wafer_rccl_allreduce_now(void *handle, void *dev_ptr, size_t count, void *stream);
long wafer_rccl_halo_calls(void *handle);
/* ncclCommCount, ncclCommUserRank, the z-neighbour ranks in use (-1 = none), ncclGetVersion */
int wafer_rccl_comm_info(void *handle, int *nranks, int *rank, int *lower, int *upper, int *version);
int wafer_rccl_detach(wafer_ctx *ctx, void *handle);

#ifdef __cplusplus
}
#endif
#endif
