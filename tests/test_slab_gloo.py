"""The N > 1 path on CPU: wafer_amd.slab's partition + halo exchange +
all-reduce on torch.distributed/gloo (world_size 2 and 3), driving a z-slab
mini-engine built from the ORACLE's per-slab kernels, compared bit for bit
with the undecomposed oracle.  (On the GPU box the same SlabComm methods move
device tensors over RCCL; the engine-side slab logic is covered by
tests/test_gpu_slab.py.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_partition():
    from wafer_amd.slab import partition
    for nz in (1, 7, 16, 128, 1024):
        for world in (1, 2, 3, 8):
            if world > nz:
                continue
            parts = [partition(nz, world, r) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == nz
            for (b0, c0), (b1, _) in zip(parts, parts[1:]):
                assert b0 + c0 == b1
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
    with pytest.raises(ValueError):
        partition(8, 2, 2)


def _worker(rank, world, port, shape, ext, wnum, steps, out_dir, one_sided=False):
    from oracle import wafer_oracle as wo
    from wafer_amd.slab import SlabComm, partition
    wo.set_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        nx, ny, nz = shape
        e = ext
        gcfg = wo.Config(nx, ny, nz, ext=e, potential="Harmonic", dn=0.25, dt=0.006, mass=1.0)
        v = wo.potential_generate(gcfg)
        a, b = wo.ab(gcfg, v)
        rng = np.random.default_rng(5)
        phi = np.zeros(gcfg.padded_shape)
        phi[e:-e, e:-e, e:-e] = rng.standard_normal(gcfg.work_shape)
        lowers = []
        for i in range(wnum):
            l = np.zeros(gcfg.padded_shape)
            l[e:-e, e:-e, e:-e] = np.random.default_rng(50 + i).standard_normal(gcfg.work_shape)
            wo.orthogonalise(i, l, lowers)
            wo.normalise(l, wo.norm2(gcfg, l))
            lowers.append(l)

        zb, zc = partition(nz, world, rank)
        comm = SlabComm(rank, world)
        lcfg = wo.Config(nx, ny, zc, ext=e, potential="Harmonic", dn=gcfg.dn, dt=gcfg.dt, mass=gcfg.mass)
        cut = lambda arr: np.ascontiguousarray(arr[:, :, zb:zb + zc + 2 * e])  # owned planes + e ghosts per side
        la, lb, lphi = cut(a), cut(b), cut(phi)
        llow = [cut(l) for l in lowers]

        def halo_exchange(p):
            t = lambda arr: torch.from_numpy(np.ascontiguousarray(arr))
            send_lo, send_hi = t(p[:, :, e:2 * e]), t(p[:, :, zc:zc + e])
            recv_lo, recv_hi = torch.empty_like(send_lo), torch.empty_like(send_hi)
            if one_sided:
                # the engine's half-slab schedule (wafer_set_overlap mode 2): one direction at a time -- what goes
                # down arrives in the lower neighbour's UPPER ghost planes, then the mirror image
                comm.exchange(send_lo if comm.lower is not None else None, None, None,
                              recv_hi if comm.upper is not None else None)
                comm.exchange(None, send_hi if comm.upper is not None else None,
                              recv_lo if comm.lower is not None else None, None)
            else:
                comm.exchange(send_lo if comm.lower is not None else None,
                              send_hi if comm.upper is not None else None,
                              recv_lo if comm.lower is not None else None,
                              recv_hi if comm.upper is not None else None)
            if comm.lower is not None:
                p[:, :, :e] = recv_lo.numpy()
            if comm.upper is not None:
                p[:, :, zc + e:] = recv_hi.numpy()

        def gsum(x):
            t = torch.tensor([x], dtype=torch.float64)
            comm.allreduce(t)
            return float(t[0])

        own = (slice(e, -e), slice(e, -e), slice(e, zc + e))
        for _ in range(steps):
            lphi[own] = wo.stencil_step(lcfg, la, lb, lphi)          # grid.rs:563-673 on the slab
            if wnum > 0:                                             # grid.rs:674-681
                n2 = gsum(wo.norm2(lcfg, lphi))
                lphi[own] = lphi[own] / np.sqrt(n2)
                for l in llow:
                    s = gsum(float(np.sum(l[own] * lphi[own])))
                    lphi[own] = lphi[own] - l[own] * s
            halo_exchange(lphi)
        np.save(os.path.join(out_dir, f"slab_{rank}.npy"), lphi[:, :, e:zc + e])
        if rank == 0:
            wo.set_threads(2)
            wo.evolve(gcfg, wnum, a, b, phi, lowers, steps)
            np.save(os.path.join(out_dir, "global.npy"), phi)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,ext,wnum,one_sided", [
    (2, (9, 8, 12), 1, 0, False),
    (2, (8, 9, 13), 2, 0, False),
    (3, (7, 6, 19), 3, 0, False),
    (2, (8, 8, 10), 1, 2, False),
    (3, (7, 6, 17), 1, 0, True),     # one direction per exchange call, three ranks: a middle rank sends and receives in both
    (2, (8, 8, 10), 2, 1, True),
])
def test_slab_evolve_matches_global(tmp_path, world, shape, ext, wnum, one_sided):
    steps = 6
    mp.spawn(_worker, args=(world, _free_port(), shape, ext, wnum, steps, str(tmp_path), one_sided), nprocs=world, join=True)
    from wafer_amd.slab import partition
    want = np.load(tmp_path / "global.npy")
    e = ext
    got = np.concatenate([np.load(tmp_path / f"slab_{r}.npy") for r in range(world)], axis=2)
    assert got.shape == (shape[0] + 2 * e, shape[1] + 2 * e, shape[2])
    if wnum == 0:
        assert np.array_equal(got, want[:, :, e:-e])   # halo exchange is exact
    else:  # global sums are associated differently per decomposition
        assert np.allclose(got, want[:, :, e:-e], rtol=0, atol=1e-13)


# ---------------------------------------------------------------------------------------------
# make_slab_comm's two-phase agreement (slab.py:306) with REAL ranks on gloo: a rank whose
# local pre-check fails (library missing on its node) must take every rank to the torch.distributed
# hooks together, before anything collective of the native path has started -- nobody left in a broadcast
# ---------------------------------------------------------------------------------------------
class _FakeCtx:
    """just enough of wafer_amd.Context for TorchSlabComm's constructor"""
    handle = None

    def set_comm_hooks(self, halo, allreduce):
        self.hooks = (halo, allreduce)


def _fallback_worker(rank, world, port, failing_rank, phase, out_dir):
    from wafer_amd import slab
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                            timeout=__import__("datetime").timedelta(seconds=60))
    try:
        real_pre = slab.NativeRcclSlabComm.precheck
        calls = {"phase2": 0}

        def pre(r):
            if phase == 1 and r == failing_rank:
                raise ImportError("simulated: libwafer_rccl.so is missing on this node")
            return ("library", "id")            # never used: phase 2 is replaced below

        class Native(slab.NativeRcclSlabComm):
            precheck = staticmethod(pre)

            def __init__(self, ctx, r, w, device, group=None, self_neighbours=False, prechecked=None, mailbox=False):
                calls["phase2"] += 1
                box = [b"id"]                   # the collective of the real phase 2: every rank must get here or none
                dist.broadcast_object_list(box, src=0, group=group)
                if phase == 2 and r == failing_rank:
                    raise RuntimeError("simulated: ncclCommInitRank failed")
                self._handle = None

            def close(self):
                pass

        slab.NativeRcclSlabComm = Native
        comm, name = slab.make_slab_comm(_FakeCtx(), rank, world, torch.device("cpu"))
        ok = isinstance(comm, slab.TorchSlabComm) and "torch.distributed" in name
        if phase == 1:
            ok = ok and calls["phase2"] == 0   # nobody entered the collective phase
        else:
            ok = ok and calls["phase2"] == 1
        dist.barrier()                           # every rank is still in step
        with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
            f.write("1" if ok else "0")
        slab.NativeRcclSlabComm.precheck = real_pre
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("phase,failing_rank", [(1, 1), (1, 0), (2, 1)])
def test_native_hook_fallback_is_agreed_by_all_ranks(tmp_path, phase, failing_rank):
    world = 2
    mp.spawn(_fallback_worker, args=(world, _free_port(), failing_rank, phase, str(tmp_path)), nprocs=world, join=True)
    assert [open(tmp_path / f"ok{r}").read() for r in range(world)] == ["1"] * world


# ---- the set-up trial of the halo schedules (bench.py --gpus N): a schedule that fails on ONE rank must be dropped on ALL,
# ---- with every rank making the same collective calls (the old loop paired one rank's barrier with the others' all-reduce)
class _TrialCtx:
    """what slab.time_overlap_schedules needs of a context; `fail` = (mode, where) makes this rank's context fail there"""

    def __init__(self, fail=None):
        self.fail, self.mode, self.log = fail, 0, []

    def _maybe(self, where):
        from wafer_amd.engine import WaferError
        if self.fail and self.fail[0] == self.mode and self.fail[1] == where:
            raise WaferError(-6, f"simulated failure in {where}")

    def set_overlap(self, mode):
        self.mode = mode
        self.log.append(("overlap", mode))
        self._maybe("set_overlap")

    def set_halo_cycle(self, cycle):
        pass

    def evolve(self, wnum, steps):
        self.log.append(("evolve", self.mode, steps))
        self._maybe(f"evolve{steps}")

    def synchronize(self):
        self._maybe("synchronize")

    def set_initial_condition(self, name):
        self.log.append(("ic", name))


def _trial_worker(rank, world, port, failing_rank, fail, out_dir):
    from wafer_amd import slab
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                            timeout=__import__("datetime").timedelta(seconds=60))
    try:
        ctx = _TrialCtx(fail if rank == failing_rank else None)
        trial = slab.time_overlap_schedules(ctx, [(3, 1), (2, 1), (1, 1), (0, 1)], rank, device="cpu", run_in=9, steps=42)
        dist.barrier()                           # every rank is still in step: a desynchronised collective would hang or throw before here
        gathered = [None] * world
        dist.all_gather_object(gathered, sorted(trial))
        ok = all(g == gathered[0] for g in gathered) and (fail[0], 1) not in trial and len(trial) == 3
        # the rank that did NOT fail must not have run the failed schedule's remaining phases
        evolves = [e for e in ctx.log if e[0] == "evolve" and e[1] == fail[0]]
        ok = ok and len(evolves) == {"set_overlap": 0, "evolve9": 1, "synchronize": 1, "evolve42": 2}[fail[1]]
        ok = ok and ("overlap", 0) in ctx.log and ("ic", "Boolean") in ctx.log    # bookkeeping reset on every rank
        with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
            f.write("1" if ok else "0")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fail", [(3, "set_overlap"), (3, "evolve9"), (2, "synchronize"), (1, "evolve42")])
@pytest.mark.parametrize("failing_rank", [0, 2])
def test_schedule_trial_drops_a_failed_schedule_on_every_rank(tmp_path, fail, failing_rank):
    """slab.time_overlap_schedules over three gloo ranks with a context that fails on ONE rank -- refused by set_overlap, in
    the run-in, at the synchronisation (where a bounded wait that gave up is reported), in the timed phase: every rank returns
    the same three surviving schedules, none hangs, none runs phases the others skipped"""
    world = 3
    mp.spawn(_trial_worker, args=(world, _free_port(), failing_rank, fail, str(tmp_path)), nprocs=world, join=True)
    assert [open(tmp_path / f"ok{r}").read() for r in range(world)] == ["1"] * world


class _AgreeCtx(_TrialCtx):
    """... and of slab.overlap_modes_agree: a checksum that depends on the overlap mode on the rank told to disagree"""

    def __init__(self, disagree, fail=None):
        super().__init__(fail)
        self.disagree = disagree

        class P:
            z_begin, z_count = 0, 8
        self.params = P()

    def checksum(self, z_begin, z_count):
        return 1234 + (self.mode if self.disagree else 0)


def _agree_worker(rank, world, port, bad_rank, kind, out_dir):
    from wafer_amd import slab
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                            timeout=__import__("datetime").timedelta(seconds=60))
    try:
        mine = rank == bad_rank
        ctx = _AgreeCtx(disagree=mine and kind == "bits", fail=(3, "synchronize") if mine and kind == "error" else None)
        got = slab.overlap_modes_agree(ctx, rank, world, 3, 2, steps=15, device="cpu")
        dist.barrier()
        want = kind == "fine"
        with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
            f.write("1" if (got == want and ctx.mode == 2) else "0")    # every rank hears the same answer and ends on the reference mode
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["fine", "bits", "error"])
def test_peer_store_self_check_is_agreed_by_all_ranks(tmp_path, kind):
    """slab.overlap_modes_agree (what bench.py and wafer_amd.run ask before overlap mode 3 is used): equal checksums on every
    rank -> True everywhere; other bits on ONE rank, or a WAFER_ERR_COMM there -> False everywhere, nobody left in a collective"""
    world = 3
    mp.spawn(_agree_worker, args=(world, _free_port(), 1, kind, str(tmp_path)), nprocs=world, join=True)
    assert [open(tmp_path / f"ok{r}").read() for r in range(world)] == ["1"] * world
