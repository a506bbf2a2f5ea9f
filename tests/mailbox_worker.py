"""One rank of the device-side all-reduce test (tests/test_gpu_multiprocess.py; torch.distributed.run, gloo rendezvous,
all ranks on the box's one GPU -- their mailboxes reach each other through HIP IPC exactly as they would over xGMI).

  * wafer_mailbox_allreduce against the sum formed on the host, for every count 1..14, many calls in a row without any
    host synchronisation in between (the double buffering by epoch parity), bitwise identical on every rank;
  * excited-state steps on z-slabs with the all-reduce hook served by the mailbox, against the same run with the hook
    served through gloo: every cell within 1e-12, norm2 within 1e-12;
  * microseconds per all-reduce, HIP events around 200 back-to-back calls.

Prints "MAILBOX-OK <world> us_per_allreduce=<x>" on rank 0."""
import ctypes as C
import dataclasses
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import wafer_amd as wa
    from wafer_amd.slab import HostStagedSlabComm, MailboxAllReduce, partition

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)

    # ---- the primitive ----
    mb = MailboxAllReduce(rank, world, 0)
    stream = torch.cuda.Stream(device=dev)
    rng = np.random.default_rng(1234)           # the same stream of numbers on every rank
    for count in list(range(1, 15)) * 3:
        vals = rng.standard_normal((world, count)) * 10.0 ** rng.integers(-8, 8, size=(world, 1))
        want = np.zeros(count)
        for r in range(world):                  # rank order, as the kernel sums
            want = want + vals[r]
        with torch.cuda.stream(stream):
            t = torch.tensor(vals[rank], dtype=torch.float64, device=dev)
            assert mb.allreduce(t.data_ptr(), count, stream.cuda_stream) == 0
            got = t.cpu().numpy()
        assert np.array_equal(got, want), (count, got, want)
    # many calls in flight, no host synchronisation in between
    with torch.cuda.stream(stream):
        t = torch.full((4,), float(rank + 1), dtype=torch.float64, device=dev)
        for _ in range(10):
            assert mb.allreduce(t.data_ptr(), 4, stream.cuda_stream) == 0
        stream.synchronize()
        s0 = world * (world + 1) / 2.0
        assert np.array_equal(t.cpu().numpy(), np.full(4, s0 * world ** 9)), t
        # latency
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t.fill_(1e-30)
        dist.barrier()
        e0.record(stream)
        for _ in range(200):
            mb.allreduce(t.data_ptr(), 4, stream.cuda_stream)
        e1.record(stream)
        stream.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 200
    mb.check()
    mb.close()

    # ---- excited-state steps on slabs: mailbox hook against the gloo hook ----
    shape, wnum = (72, 40, 50), 2
    whole = wa.Params(*shape, dn=0.25, dt=0.006, central_difference=1, max_states=wnum)
    zb, zc = partition(shape[2], world, rank)
    mine = dataclasses.replace(whole, z_begin=zb, z_count=zc)

    def run(mailbox):
        with wa.Context(mine) as ctx:
            comm = HostStagedSlabComm(ctx, rank, world, dev, mailbox=mailbox)
            ctx.set_potential("Harmonic")
            for j in range(wnum):
                ctx.set_initial_condition("Gaussian", seed=40 + j)
                ctx.normalise(ctx.norm2())
                ctx.orthogonalise(j)
                ctx.normalise(ctx.norm2())
                ctx.push_state()
            ctx.set_initial_condition("Gaussian", seed=7)
            ctx.evolve(wnum, 6)
            out = ctx.download_phi(), ctx.norm2(), ctx.observables()
            if comm.mailbox is not None:
                comm.mailbox.check()
                comm.mailbox.close()
            return out
    a, b = run(True), run(False)
    ext = 1
    mine_a, mine_b = a[0][:, :, zb + ext:zb + zc + ext], b[0][:, :, zb + ext:zb + zc + ext]
    assert float(np.max(np.abs(mine_a - mine_b))) <= 1e-12 * max(1.0, float(np.max(np.abs(mine_b))))
    assert abs(a[1] - b[1]) <= 1e-12 * abs(b[1])
    for k in b[2]:
        assert abs(a[2][k] - b[2][k]) <= 1e-12 * max(1.0, abs(b[2][k])), k
    all_us = [None] * world
    dist.all_gather_object(all_us, us)
    if rank == 0:
        print(f"MAILBOX-OK {world} us_per_allreduce={max(all_us):.2f}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
