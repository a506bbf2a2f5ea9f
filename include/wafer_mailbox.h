/*
 * wafer_mailbox.h -- a device-side all-reduce for the handful of doubles the path's global sums need
 * (SURVEY.md section 7 "hard parts", 8e collectives (2): 1 + k doubles per excited-state step, 4 per
 * compute_observables, on the critical path of every step).  Part of libwafer_hip.so; no communication
 * library involved.
 *
 * Every rank owns a mailbox in its own device memory and maps every other rank's through HIP IPC (one node:
 * peers over xGMI, or several ranks on one GPU).  One all-reduce is ONE one-wave kernel per rank on the
 * stream the data is ordered on: lane r stores this rank's values, then the call's epoch, into rank r's mailbox
 * (system-scope stores, the epoch behind a release); the same lanes then poll this rank's own mailbox until
 * every sender's epoch has arrived, and the sums are formed in rank order -- the same bits on every rank --
 * and written in place.  Two buffers alternate by epoch parity: a rank can be at most one all-reduce ahead of
 * another (it needs the other's contribution to finish).  Waits are bounded (2^26 spins, several seconds); a wait that
 * gives up turns that call's results into NaN on the rank, makes every later call of the rank post and return NaN, and
 * is reported by wafer_mailbox_check and by every later wafer_mailbox_allreduce (non-zero return): the failure is
 * sticky, a mailbox that timed out once is not used again.
 *
 * The reference has no counterpart (one process); a Rust host would bind these five functions next to
 * wafer_set_comm_hooks.  wafer_mailbox_allreduce has the signature of wafer_allreduce_fn, `user` = the mailbox.
 */
#ifndef WAFER_MAILBOX_H
#define WAFER_MAILBOX_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WAFER_MAILBOX_MAX_RANKS 16
#define WAFER_MAILBOX_MAX_COUNT 14      /* doubles per all-reduce */
#define WAFER_MAILBOX_HANDLE_BYTES 64   /* sizeof(hipIpcMemHandle_t) */

typedef struct wafer_mailbox wafer_mailbox;

/* allocates this rank's mailbox on `device` (the current device is set) */
int wafer_mailbox_create(int rank, int world, int device, wafer_mailbox **out);
/* the IPC handle of this rank's mailbox: WAFER_MAILBOX_HANDLE_BYTES bytes, to be gathered over all ranks by the host */
int wafer_mailbox_handle(wafer_mailbox *mb, void *handle_out);
/* all_handles: world x WAFER_MAILBOX_HANDLE_BYTES bytes in rank order; maps every other rank's mailbox */
int wafer_mailbox_connect(wafer_mailbox *mb, const void *all_handles);
/* in-place sum over ranks of `count` (<= WAFER_MAILBOX_MAX_COUNT) doubles at dev_ptr, enqueued on hip_stream;
 * every rank must make the same sequence of calls.  Signature of wafer_allreduce_fn (wafer_hip.h). */
int wafer_mailbox_allreduce(void *mailbox, void *dev_ptr, size_t count, void *hip_stream);
/* WAFER_ERR_COMM (and a message through wafer_last_error) once a kernel has given up waiting; sticky */
int wafer_mailbox_check(wafer_mailbox *mb);
int wafer_mailbox_destroy(wafer_mailbox *mb);

#ifdef __cplusplus
}
#endif
#endif
