"""z-slab decomposition of the grid over one process per GPU.

The reference has no distributed path (its only trace is the comment "without
mpi, this is just update interior", grid.rs:551).  The stencil reaches `ext`
planes along z, every other operation is elementwise or a global sum, so the
grid shards into contiguous z-slabs with
  - one nearest-neighbour halo exchange of `ext` planes per step, and
  - scalar all-reduces for norm^2 / overlaps / observables.
a, b, V, pot_sub and the stored states are sharded identically and never move.

The engine (C ABI) does not link a communication library; it calls the two
hooks installed here.  `SlabComm` implements them on torch.distributed, so the
same code runs on RCCL over xGMI (backend "nccl", device tensors aliasing the
engine's HBM buffers) and on gloo with CPU tensors (tests/test_slab_gloo.py).
"""
from __future__ import annotations

from typing import Optional


def partition(nz: int, world: int, rank: int):
    """Contiguous balanced split of nz work planes: -> (z_begin, z_count)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(nz, world)
    count = base + (1 if rank < rem else 0)
    begin = rank * base + min(rank, rem)
    return begin, count


class SlabComm:
    """Neighbour halo exchange and scalar all-reduce on torch.distributed.

    Rank r owns planes partition(nz, world, r); its lower neighbour (smaller z)
    is rank r-1, its upper neighbour rank r+1; ranks 0 and world-1 face the
    global Dirichlet frame on their outer side and exchange nothing there.
    """

    def __init__(self, rank: int, world: int, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.world, self.group = rank, world, group

    @property
    def lower(self) -> Optional[int]:
        return self.rank - 1 if self.rank > 0 else None

    @property
    def upper(self) -> Optional[int]:
        return self.rank + 1 if self.rank + 1 < self.world else None

    def exchange(self, send_lo, send_hi, recv_lo, recv_hi):
        """send_lo -> lower neighbour's recv_hi, send_hi -> upper neighbour's
        recv_lo (tensors, or None where there is no neighbour).  Returns the
        list of outstanding works (already waited on for stream ordering)."""
        dist = self.dist
        ops = []
        # receives first, then sends; NCCL groups them, gloo posts them asynchronously
        # (a buffer may also be None on a side that HAS a neighbour: the engine's half-slab schedule,
        #  wafer_set_overlap mode 2, exchanges one direction at a time)
        if self.lower is not None and recv_lo is not None:
            ops.append(dist.P2POp(dist.irecv, recv_lo, self.lower, self.group))
        if self.upper is not None and recv_hi is not None:
            ops.append(dist.P2POp(dist.irecv, recv_hi, self.upper, self.group))
        if self.lower is not None and send_lo is not None:
            ops.append(dist.P2POp(dist.isend, send_lo, self.lower, self.group))
        if self.upper is not None and send_hi is not None:
            ops.append(dist.P2POp(dist.isend, send_hi, self.upper, self.group))
        if not ops:
            return []
        works = dist.batch_isend_irecv(ops)
        for w in works:
            w.wait()  # nccl: orders the current stream after the transfer; gloo: blocks
        return works

    def allreduce(self, t):
        """in-place sum over ranks"""
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t


class _DeviceView:
    """Exposes a raw device address range through __cuda_array_interface__ so
    torch can alias the engine's HBM buffers without owning them."""

    def __init__(self, ptr: int, count: int, typestr: str):
        self.__cuda_array_interface__ = {
            "shape": (count,), "typestr": typestr, "data": (int(ptr), False), "version": 3,
            "strides": None,
        }


class TorchSlabComm(SlabComm):
    """Installs the engine's halo / allreduce hooks on a wafer_amd.Context.

    The hooks receive raw device addresses plus the hipStream_t the data is
    ordered on; the collective is enqueued under that stream (ExternalStream),
    so RCCL waits for the boundary kernels and the stream waits for RCCL --
    nothing blocks the host, and the interior update running on the engine's
    other stream overlaps the transfer.
    """

    def __init__(self, ctx, rank: int, world: int, device, group=None):
        super().__init__(rank, world, group)
        import torch
        self.torch = torch
        self.device = device
        self.ctx = ctx
        self._tensors = {}
        self._streams = {}
        self.halo_calls = 0   # calls of the halo hook so far (overlap mode 4 -- peer copies -- must leave it alone)
        ctx.set_comm_hooks(self._counted_halo_hook, self._allreduce_hook)

    def warm_up(self):
        """Establish the neighbour connections and the all-reduce ring before anything is timed
        (RCCL builds its point-to-point channels lazily, at the first send / receive)."""
        torch = self.torch
        mk = lambda: torch.zeros(256, dtype=torch.uint8, device=self.device)  # noqa: E731
        self.exchange(mk() if self.lower is not None else None, mk() if self.upper is not None else None,
                      mk() if self.lower is not None else None, mk() if self.upper is not None else None)
        self.allreduce(torch.zeros(8, dtype=torch.float64, device=self.device))
        torch.cuda.synchronize(self.device)

    def _stream(self, ptr):
        s = self._streams.get(ptr)
        if s is None:
            s = self.torch.cuda.ExternalStream(int(ptr), device=self.device)
            self._streams[ptr] = s
        return s

    def _bytes(self, ptr, nbytes):
        if not ptr:
            return None
        key = (int(ptr), int(nbytes), "u1")
        t = self._tensors.get(key)
        if t is None:
            t = self.torch.as_tensor(_DeviceView(ptr, nbytes, "|u1"), device=self.device)
            self._tensors[key] = t
        return t

    def _doubles(self, ptr, count):
        key = (int(ptr), int(count), "f8")
        t = self._tensors.get(key)
        if t is None:
            t = self.torch.as_tensor(_DeviceView(ptr, count, "<f8"), device=self.device)
            self._tensors[key] = t
        return t

    def _counted_halo_hook(self, *a):
        self.halo_calls += 1
        return self._halo_hook(*a)

    def _halo_hook(self, send_lo, send_hi, recv_lo, recv_hi, nbytes, stream):
        with self.torch.cuda.stream(self._stream(stream)):
            self.exchange(self._bytes(send_lo, nbytes), self._bytes(send_hi, nbytes),
                          self._bytes(recv_lo, nbytes), self._bytes(recv_hi, nbytes))
        return 0

    def _allreduce_hook(self, ptr, count, stream):
        with self.torch.cuda.stream(self._stream(stream)):
            self.allreduce(self._doubles(ptr, count))
        return 0


def connect_peers(ctx, rank: int, world: int, group=None) -> bool:
    """Peer stores (wafer_set_overlap mode 3): every rank publishes its wafer_peer_info record (device addresses + HIP IPC
    handles of its two phi buffers and its arrival counters), the records travel through torch.distributed (any backend), and
    each rank maps its z-neighbours'.  Collective; returns True when EVERY rank connected (the mode is for all ranks or none)
    -- False leaves the context as it was, e.g. where the runtime cannot export or map the handles."""
    import torch.distributed as dist
    err, rec = "", b""
    try:
        rec = ctx.peer_export()
    except Exception as e:  # noqa: BLE001
        err = repr(e)
    recs = [None] * world
    if world > 1:
        dist.all_gather_object(recs, (err, rec), group=group)
    else:
        recs = [(err, rec)]
    ok = not any(e for e, _r in recs)
    if ok:
        try:
            ctx.peer_connect(recs[rank - 1][1] if rank > 0 else None, recs[rank + 1][1] if rank + 1 < world else None)
        except Exception as e:  # noqa: BLE001
            err = repr(e)
        outcome = [None] * world
        if world > 1:
            dist.all_gather_object(outcome, err, group=group)
        else:
            outcome = [err]
        ok = not any(outcome)
    if not ok:
        try:
            ctx.peer_disconnect()
        except Exception:  # noqa: BLE001
            pass
    return ok


def overlap_modes_agree(ctx, rank: int, world: int, mode: int, ref_mode: int, steps: int = 15, group=None, device="cpu") -> bool:
    """Collective self-check before a schedule is trusted on a fabric it has never run on (peer stores between GPUs: nothing
    in this repository has crossed a link): the same `steps` ground-state steps from the Boolean initial condition under overlap
    `mode` and under `ref_mode` (an exchange through the halo hook), the checksum of this rank's own planes after each -- every
    cell's bits -- and True only if they are equal on EVERY rank.  A WAFER_ERR_COMM in either run counts as disagreement.  The
    potential must be set; phi is left at the Boolean initial condition and the overlap mode at `ref_mode`."""
    import torch
    import torch.distributed as dist
    from .engine import WaferError
    sums, bad = [], False

    def any_rank(flag: bool) -> bool:
        if world <= 1:
            return flag
        t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return float(t[0]) != 0.0

    for m in (mode, ref_mode):
        # a schedule this rank's context refuses (mode 3 on a slab thinner than 6 ext planes, no peers connected) must be known to
        # EVERY rank before any of them evolves: evolve's halo hook posts sends and receives to the neighbours, and a rank that
        # skipped it would sit in the barrier below while they wait for its planes
        refused = False
        try:
            ctx.set_overlap(m)
            ctx.set_initial_condition("Boolean")
        except WaferError:
            refused = True
        if any_rank(refused):
            bad = True
            sums.append(None)
            continue
        try:
            ctx.evolve(0, steps)
            ctx.synchronize()
            sums.append(ctx.checksum(ctx.params.z_begin, ctx.params.z_count))
        except WaferError:
            bad = True
            sums.append(None)
        if world > 1:   # the same collectives on every rank whatever happened here
            dist.barrier(group=group)
    bad = bad or sums[0] != sums[1]
    if world > 1:
        t = torch.tensor([1.0 if bad else 0.0], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        bad = float(t[0]) != 0.0
    try:
        ctx.set_overlap(ref_mode)
        ctx.set_initial_condition("Boolean")
    except WaferError:
        pass
    return not bad


def time_overlap_schedules(ctx, candidates, rank: int, device="cpu", group=None, run_in: int = 9, steps: int = 42, log=None,
                           device_sync=None) -> dict:
    """Collective.  Times every halo schedule of `candidates` -- (overlap mode, fused passes per exchange) pairs -- over a run-in
    of `run_in` and `steps` further ground-state steps and returns {(mode, cycle): ms per step on the slowest rank} for those that
    worked on EVERY rank.  Every rank makes the same sequence of collective calls whatever happens on it: a schedule that this
    rank's context refuses, or whose bounded waits give up on this fabric (WAFER_ERR_COMM -- a neighbour's planes never arrived;
    reported when the rank next synchronises with its device), is agreed on right after the phase in which it happened and
    dropped on all ranks; nobody walks on into a barrier that pairs with somebody else's all-reduce.  phi is left undefined
    (the caller re-initialises); after a failure the pass bookkeeping is reset on every rank (set_overlap(0))."""
    import time
    import torch
    import torch.distributed as dist
    from .engine import WaferError
    say = log or (lambda msg: None)

    def any_rank(flag: bool) -> bool:
        t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return float(t[0]) != 0.0

    def phase(n_steps: int, what: str):
        bad = False
        t0 = time.perf_counter()
        try:
            ctx.evolve(0, n_steps)
            ctx.synchronize()
            if device_sync is not None:
                device_sync()
        except WaferError as e:
            say(f"rank {rank}: halo schedule {what} failed in the set-up trial: {e}")
            bad = True
        dt = time.perf_counter() - t0
        dist.barrier(group=group)
        tt = torch.tensor([dt, 1.0 if bad else 0.0], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=group)
        return float(tt[0]), float(tt[1]) != 0.0

    trial = {}
    for mode, cycle in candidates:
        what = f"{mode} (cycle {cycle})"
        refused = False
        try:
            ctx.set_overlap(mode)
            ctx.set_halo_cycle(cycle)
        except WaferError as e:
            say(f"rank {rank}: halo schedule {what} refused: {e}")
            refused = True
        failed = any_rank(refused)
        if not failed:
            _, failed = phase(run_in, what)      # the run-in: rendezvous, first exchanges, table uploads
        if not failed:
            seconds, failed = phase(steps, what)
        if failed:
            try:
                ctx.set_overlap(0)                    # resets the pass bookkeeping on every rank
                ctx.set_initial_condition("Boolean")  # whatever the failed passes left behind
            except WaferError:
                pass
            continue
        trial[(mode, cycle)] = seconds / steps * 1e3
    return trial


class MailboxAllReduce:
    """include/wafer_mailbox.h through ctypes: the device-side all-reduce of libwafer_hip.so -- every rank's mailbox
    mapped into every other rank through HIP IPC, one one-wave kernel per call, sums in rank order (the same bits on
    every rank).  The IPC handles are gathered with torch.distributed (any backend).  Serves the all-reduce hook of
    any of the comm classes below (`mailbox=True`); RCCL's ncclAllReduce stays the default."""

    def __init__(self, rank: int, world: int, device_index: int, group=None):
        import ctypes as C
        import torch.distributed as dist
        from .engine import load_library
        self._L = L = load_library()
        L.wafer_mailbox_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.wafer_mailbox_handle.argtypes = [C.c_void_p, C.c_void_p]
        L.wafer_mailbox_connect.argtypes = [C.c_void_p, C.c_void_p]
        L.wafer_mailbox_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.wafer_mailbox_check.argtypes = [C.c_void_p]
        L.wafer_mailbox_destroy.argtypes = [C.c_void_p]
        L.wafer_last_error.restype = C.c_char_p
        self._mb = C.c_void_p()
        # every rank reports how its own set-up went BEFORE anything depends on a peer: a rank whose mailbox cannot be
        # created must not leave the others waiting in the gather
        err, raw = "", b"\0" * 64
        if L.wafer_mailbox_create(rank, world, device_index, C.byref(self._mb)) != 0:
            err = L.wafer_last_error().decode()
        else:
            h = C.create_string_buffer(64)
            if L.wafer_mailbox_handle(self._mb, h) != 0:
                err = L.wafer_last_error().decode()
            raw = h.raw
        gathered = [None] * world
        if world > 1:
            dist.all_gather_object(gathered, (err, raw), group=group)
        else:
            gathered = [(err, raw)]
        bad = [(r, e) for r, (e, _h) in enumerate(gathered) if e]
        if not bad:
            blob = C.create_string_buffer(b"".join(hh for _e, hh in gathered), 64 * world)
            if L.wafer_mailbox_connect(self._mb, blob) != 0:
                err = L.wafer_last_error().decode()
            if world > 1:   # ... and nobody starts using mailboxes that a peer could not map
                dist.all_gather_object(gathered, (err, b""), group=group)
                bad = [(r, e) for r, (e, _h) in enumerate(gathered) if e]
            elif err:
                bad = [(rank, err)]
        if bad:
            self.close()
            raise RuntimeError("wafer_mailbox set-up failed on rank(s) " + "; ".join(f"{r}: {e}" for r, e in bad))

    @property
    def handle(self):
        return self._mb

    def allreduce(self, ptr: int, count: int, stream: int) -> int:
        return int(self._L.wafer_mailbox_allreduce(self._mb, ptr, count, stream))

    def check(self):
        if self._L.wafer_mailbox_check(self._mb) != 0:
            raise RuntimeError(self._L.wafer_last_error().decode())

    def close(self):
        if self._mb:
            self._L.wafer_mailbox_destroy(self._mb)
            self._mb = None


class HostStagedSlabComm(TorchSlabComm):
    """The same hooks over a CPU process group (gloo): halo planes and scalars are
    staged through host buffers (plain pageable tensors and blocking copies:
    torch's pinned-memory cache would remember the engine's streams past the
    context's lifetime).  For fabrics without device-aware
    collectives, and for running the multi-process path with several ranks on
    ONE GPU (tests/test_gpu_multiprocess.py) -- RCCL refuses two ranks per
    device.  The engine's arithmetic is untouched: only bytes move differently.
    """

    def __init__(self, ctx, rank: int, world: int, device, group=None, mailbox: bool = False):
        super().__init__(ctx, rank, world, device, group)
        self._host = {}
        # the all-reduce hook served on the device (wafer_mailbox.h) instead of through host staging
        self.mailbox = MailboxAllReduce(rank, world, device.index or 0, group) if mailbox else None

    def warm_up(self):
        torch = self.torch
        mk = lambda: torch.zeros(256, dtype=torch.uint8)  # noqa: E731
        self.exchange(mk() if self.lower is not None else None, mk() if self.upper is not None else None,
                      mk() if self.lower is not None else None, mk() if self.upper is not None else None)
        self.allreduce(torch.zeros(8, dtype=torch.float64))

    def _staging(self, tag, nbytes, dtype):
        key = (tag, int(nbytes), dtype)
        t = self._host.get(key)
        if t is None:
            t = self.torch.empty(int(nbytes), dtype=dtype)
            self._host[key] = t
        return t

    def _halo_hook(self, send_lo, send_hi, recv_lo, recv_hi, nbytes, stream):
        torch = self.torch
        st = self._stream(stream)
        u8 = torch.uint8
        host = {}
        with torch.cuda.stream(st):
            for tag, ptr in (("slo", send_lo), ("shi", send_hi)):
                if ptr:
                    host[tag] = self._staging(tag, nbytes, u8)
                    host[tag].copy_(self._bytes(ptr, nbytes))  # ordered on st, blocks until on the host
            for tag, ptr in (("rlo", recv_lo), ("rhi", recv_hi)):
                if ptr:
                    host[tag] = self._staging(tag, nbytes, u8)
            self.exchange(host.get("slo"), host.get("shi"), host.get("rlo"), host.get("rhi"))
            for tag, ptr in (("rlo", recv_lo), ("rhi", recv_hi)):
                if ptr:
                    self._bytes(ptr, nbytes).copy_(host[tag])
            st.synchronize()
        return 0

    def _allreduce_hook(self, ptr, count, stream):
        if self.mailbox is not None:
            return self.mailbox.allreduce(ptr, count, stream)
        torch = self.torch
        st = self._stream(stream)
        with torch.cuda.stream(st):
            h = self._staging("scal", count, torch.float64)
            h.copy_(self._doubles(ptr, count))
            self.allreduce(h)
            self._doubles(ptr, count).copy_(h)
            st.synchronize()
        return 0


class NativeRcclSlabComm:
    """The engine's hooks served by RCCL's C API directly (wafer_amd/csrc/wafer_rccl_hooks.h through
    libwafer_rccl.so): grouped ncclSend / ncclRecv and ncclAllReduce on the engine's own streams, no
    Python in the hook path.  torch.distributed is used only to hand rank 0's ncclUniqueId to the
    other ranks.  Measured on one GPU against a rank that is its own neighbour: 0.367 ms/step at the
    1024x1024x128 bench slab, 0.386 for TorchSlabComm (tools/slab_overhead.py, wafer-hip-slabs --self).

    `self_neighbours=True` (tests): world of one rank whose z-neighbours are itself.
    """

    @staticmethod
    def precheck(rank: int):
        """Phase 1, LOCAL and free of collectives: load libwafer_rccl.so, bind it, and on rank 0 make
        the ncclUniqueId.  Returns (library, id buffer); raises on any failure.  make_slab_comm
        all-reduces whether every rank got through before anyone enters the collective phase 2, so a
        rank with a missing or unloadable library can no longer leave the others in a broadcast."""
        import ctypes as C
        import os
        import torch  # noqa: F401 -- before librccl: the process-wide RCCL / HIP runtime are torch's (same sonames)
        # RCCL's send / recv kernels take whole CUs away from the stencil for as long as the links are
        # busy (their workgroups cannot share a CU with a stencil workgroup); left alone RCCL launches
        # 64 of them for the four transfers of a pass.  8 channels still move the 2 x 17.7 MB of a
        # 1024^2 plane pair faster than a link can (215 GB/s to the same GPU) and leave the CUs to the
        # interior launch.  Respected only if the user has not chosen otherwise.
        os.environ.setdefault("NCCL_MAX_P2P_NCHANNELS", "8")
        here = os.path.dirname(os.path.abspath(__file__))
        path = os.path.join(here, "libwafer_rccl.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: build it with `python -m wafer_amd.build`")
        L = C.CDLL(path)
        L.wafer_rccl_last_error.restype = C.c_char_p
        L.wafer_rccl_attach.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.wafer_rccl_warm_up.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.wafer_rccl_detach.argtypes = [C.c_void_p, C.c_void_p]
        L.wafer_rccl_halo_calls.argtypes = [C.c_void_p]
        L.wafer_rccl_halo_calls.restype = C.c_long
        ip = C.POINTER(C.c_int)
        L.wafer_rccl_comm_info.argtypes = [C.c_void_p, ip, ip, ip, ip, ip]
        uid = C.create_string_buffer(L.wafer_rccl_unique_id_bytes())
        if rank == 0 and L.wafer_rccl_unique_id(uid) != 0:
            raise RuntimeError("wafer_rccl_unique_id: " + L.wafer_rccl_last_error().decode())
        return L, uid

    def __init__(self, ctx, rank: int, world: int, device, group=None, self_neighbours: bool = False, prechecked=None,
                 mailbox: bool = False):
        """Phase 2, COLLECTIVE (broadcast of rank 0's id, ncclCommInitRank): every rank of `group` must
        call it, which make_slab_comm guarantees by agreeing on phase 1 first."""
        import ctypes as C
        import torch
        self.torch, self.ctx, self.rank, self.world, self.device = torch, ctx, rank, world, device
        L, uid = prechecked if prechecked is not None else self.precheck(rank)
        self._L = L
        n = L.wafer_rccl_unique_id_bytes()
        if world > 1:
            import torch.distributed as dist
            box = [uid.raw]
            dist.broadcast_object_list(box, src=0, group=group)
            uid = C.create_string_buffer(box[0], n)
        self._handle = C.c_void_p()
        with torch.cuda.device(device):
            rc = L.wafer_rccl_attach(ctx.handle, rank, world, uid, 0 if self_neighbours else -1,
                                     0 if self_neighbours else -1, C.byref(self._handle))
        if rc != 0:
            self._handle = None
            raise RuntimeError("wafer_rccl_attach: " + L.wafer_rccl_last_error().decode())
        self.mailbox = None
        if mailbox:   # the scalar all-reduces on the device (wafer_mailbox.h); halo planes stay with RCCL
            self.mailbox = MailboxAllReduce(rank, 1 if self_neighbours else world, device.index or 0, group)
            L.wafer_rccl_use_mailbox.argtypes = [C.c_void_p, C.c_void_p]
            if L.wafer_rccl_use_mailbox(self._handle, self.mailbox.handle) != 0:
                raise RuntimeError("wafer_rccl_use_mailbox: " + L.wafer_rccl_last_error().decode())

    def warm_up(self):
        torch = self.torch
        scratch = torch.zeros(4096, dtype=torch.uint8, device=self.device)
        stream = torch.cuda.current_stream(self.device)
        if self._L.wafer_rccl_warm_up(self._handle, scratch.data_ptr(), stream.cuda_stream) != 0:
            raise RuntimeError("wafer_rccl_warm_up: " + self._L.wafer_rccl_last_error().decode())

    def halo_calls(self) -> int:
        return int(self._L.wafer_rccl_halo_calls(self._handle))

    def pick_allreduce(self, group=None, calls: int = 200) -> dict:
        """Which all-reduce serves the path's few doubles (1 + k per excited-state step, 2 + 3k per two-step pass, 4 per
        observables) is decided by measurement on the fabric at hand: `calls` back-to-back 4-double all-reduces through
        ncclAllReduce, then through the device-side mailboxes (include/wafer_mailbox.h: HIP-IPC-mapped, one one-wave kernel per
        call) -- the slowest rank's time decides, and every rank switches or none.  Collective.  Returns the two timings (us
        per call) and the choice; a mailbox that cannot be set up or does not sum correctly leaves ncclAllReduce in place."""
        import ctypes as C
        import torch
        import torch.distributed as dist
        L = self._L
        L.wafer_rccl_allreduce_now.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.wafer_rccl_use_mailbox.argtypes = [C.c_void_p, C.c_void_p]
        world = 1 if self.world <= 1 else self.world
        stream = torch.cuda.Stream(device=self.device)

        def slowest(x: float) -> float:
            if self.world <= 1:
                return x
            t = torch.tensor([x], dtype=torch.float64, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            return float(t[0])

        def timed():
            """-> (us per call, every call returned 0 and the sum is right)"""
            t = torch.ones(4, dtype=torch.float64, device=self.device)
            rcs = []
            with torch.cuda.stream(stream):
                for _ in range(10):
                    rcs.append(L.wafer_rccl_allreduce_now(self._handle, t.data_ptr(), 4, stream.cuda_stream))
                stream.synchronize()
                t.fill_(1.0)
                rcs.append(L.wafer_rccl_allreduce_now(self._handle, t.data_ptr(), 4, stream.cuda_stream))
                stream.synchronize()
                good = bool(torch.all(t == float(world)).item())
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(calls):
                    rcs.append(L.wafer_rccl_allreduce_now(self._handle, t.data_ptr(), 4, stream.cuda_stream))
                e1.record(stream)
                stream.synchronize()
            # (a mailbox whose wait gave up stays dead and says so through the return code of every later call)
            return e0.elapsed_time(e1) * 1e3 / calls, good and not any(rcs)

        out = {}
        had = self.mailbox
        # every decision below is taken by ALL ranks together (slowest() is an all-reduce): a rank that raised, skipped timed() or
        # kept another transport on its own would leave the others inside their 200 collective calls
        detach_failed = L.wafer_rccl_use_mailbox(self._handle, None) != 0
        if slowest(1.0 if detach_failed else 0.0) != 0.0:
            raise RuntimeError("wafer_rccl_use_mailbox(None) failed on some rank: " + L.wafer_rccl_last_error().decode())
        us_nccl, good_nccl = timed()
        out["ncclAllReduce_us"] = slowest(us_nccl)
        nccl_ok = slowest(0.0 if good_nccl else 1.0) == 0.0
        if not nccl_ok:
            out["ncclAllReduce_error"] = "a call failed or summed wrongly on some rank"
        mb = had
        if mb is None:
            try:
                mb = MailboxAllReduce(self.rank, world, self.device.index or 0, group)
            except Exception as e:  # noqa: BLE001 -- raised on every rank together (MailboxAllReduce gathers the outcome)
                out["mailbox_error"] = repr(e)
                mb = None
        if mb is not None:
            attached = L.wafer_rccl_use_mailbox(self._handle, mb.handle) == 0
            if slowest(0.0 if attached else 1.0) != 0.0:      # some rank could not attach: nobody times the mailbox
                out["mailbox_error"] = "wafer_rccl_use_mailbox failed on some rank"
                us_mb, good = float("inf"), False
            else:
                us_mb, good = timed()
            bad = slowest(0.0 if good else 1.0)
            out["mailbox_us"] = slowest(us_mb)
            if bad == 0.0 and (out["mailbox_us"] < out["ncclAllReduce_us"] or not nccl_ok):
                self.mailbox = mb
                out["chosen"] = "mailbox"
                return out
            L.wafer_rccl_use_mailbox(self._handle, None)
            mb.close()          # whoever made it: a mailbox that lost (or died) serves nobody, and close() could no longer reach it
            self.mailbox = None
        if not nccl_ok:   # (agreed above: every rank raises) -- never continue on an all-reduce that has just summed wrongly
            raise RuntimeError("pick_allreduce: ncclAllReduce failed its own check on some rank and no working mailbox exists: "
                               + str(out))
        out["chosen"] = "ncclAllReduce"
        return out

    def info(self) -> dict:
        """what RCCL reports for the communicator the hooks use (ncclCommCount, ncclCommUserRank,
        ncclGetVersion), the z-neighbours in use and the channel limit in force"""
        import ctypes as C
        import os
        v = [C.c_int(-1) for _ in range(5)]
        if self._L.wafer_rccl_comm_info(self._handle, *[C.byref(x) for x in v]) != 0:
            raise RuntimeError("wafer_rccl_comm_info: " + self._L.wafer_rccl_last_error().decode())
        return {"rccl_ranks": v[0].value, "rccl_rank": v[1].value, "lower": v[2].value, "upper": v[3].value,
                "rccl_version": v[4].value, "NCCL_MAX_P2P_NCHANNELS": os.environ.get("NCCL_MAX_P2P_NCHANNELS")}

    def close(self):
        if getattr(self, "_handle", None):
            self.ctx.synchronize()
            self._L.wafer_rccl_detach(self.ctx.handle, self._handle)
            self._handle = None
            if getattr(self, "mailbox", None) is not None:
                self.mailbox.close()
                self.mailbox = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_slab_comm(ctx, rank: int, world: int, device, transport: Optional[str] = None):
    """Installs the communication hooks of `ctx` and returns (comm, name).

    transport: "native" (default; RCCL's C API through libwafer_rccl.so), "torch" (TorchSlabComm:
    RCCL through torch.distributed) or "host" (HostStagedSlabComm over gloo).  Defaults to
    $WAFER_TRANSPORT.  If the native hooks cannot be installed on some rank, every rank falls back
    to "torch" together (the decision is all-reduced).  torch.distributed must be initialised.
    """
    import os
    import sys
    import torch
    import torch.distributed as dist
    transport = transport or os.environ.get("WAFER_TRANSPORT", "native")
    if transport == "host":
        return HostStagedSlabComm(ctx, rank, world, device), "host-staged gloo"
    if transport == "native":
        def agreed(ok: int) -> bool:
            flag = torch.tensor([ok], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag.item()) == 1

        def complain(e):
            print(f"wafer_amd.slab: native RCCL hooks unavailable on rank {rank} ({e!r}); falling back to torch.distributed",
                  file=sys.stderr, flush=True)

        # phase 1: local checks only (library present and loadable, rank 0's unique id) -- agreed on
        # BEFORE anything collective, so a one-sided failure cannot strand the other ranks
        pre = None
        try:
            pre = NativeRcclSlabComm.precheck(rank)
        except Exception as e:  # noqa: BLE001 -- any failure means "use the torch path"
            complain(e)
        if agreed(1 if pre is not None else 0):
            # phase 2: id broadcast + ncclCommInitRank on every rank, then agree on the outcome
            comm = None
            try:
                comm = NativeRcclSlabComm(ctx, rank, world, device, prechecked=pre,
                                          mailbox=os.environ.get("WAFER_MAILBOX", "0") not in ("", "0"))
            except Exception as e:  # noqa: BLE001
                complain(e)
            if agreed(1 if comm is not None else 0):
                return comm, "RCCL (native hooks)"
            if comm is not None:
                comm.close()
    return TorchSlabComm(ctx, rank, world, device), "RCCL (torch.distributed hooks)"
