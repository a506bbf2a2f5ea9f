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
        if self.lower is not None:
            ops.append(dist.P2POp(dist.irecv, recv_lo, self.lower, self.group))
        if self.upper is not None:
            ops.append(dist.P2POp(dist.irecv, recv_hi, self.upper, self.group))
        if self.lower is not None:
            ops.append(dist.P2POp(dist.isend, send_lo, self.lower, self.group))
        if self.upper is not None:
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
        ctx.set_comm_hooks(self._halo_hook, self._allreduce_hook)

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

    def _halo_hook(self, send_lo, send_hi, recv_lo, recv_hi, nbytes, stream):
        with self.torch.cuda.stream(self._stream(stream)):
            self.exchange(self._bytes(send_lo, nbytes), self._bytes(send_hi, nbytes),
                          self._bytes(recv_lo, nbytes), self._bytes(recv_hi, nbytes))
        return 0

    def _allreduce_hook(self, ptr, count, stream):
        with self.torch.cuda.stream(self._stream(stream)):
            self.allreduce(self._doubles(ptr, count))
        return 0


class HostStagedSlabComm(TorchSlabComm):
    """The same hooks over a CPU process group (gloo): halo planes and scalars are
    staged through host buffers (plain pageable tensors and blocking copies:
    torch's pinned-memory cache would remember the engine's streams past the
    context's lifetime).  For fabrics without device-aware
    collectives, and for running the multi-process path with several ranks on
    ONE GPU (tests/test_gpu_multiprocess.py) -- RCCL refuses two ranks per
    device.  The engine's arithmetic is untouched: only bytes move differently.
    """

    def __init__(self, ctx, rank: int, world: int, device, group=None):
        super().__init__(ctx, rank, world, device, group)
        self._host = {}

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
        torch = self.torch
        st = self._stream(stream)
        with torch.cuda.stream(st):
            h = self._staging("scal", count, torch.float64)
            h.copy_(self._doubles(ptr, count))
            self.allreduce(h)
            self._doubles(ptr, count).copy_(h)
            st.synchronize()
        return 0
