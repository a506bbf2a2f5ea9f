"""RCCL itself on the one GPU of the box: a world of ONE rank ("nccl" backend = RCCL) whose
neighbours are itself, so every halo plane really travels through ncclSend / ncclRecv and every
scalar through ncclAllReduce -- from tensors that alias the engine's HBM
(__cuda_array_interface__) and under the engine's own streams (ExternalStream), exactly as
wafer_amd.slab.TorchSlabComm issues them on an 8-GPU node.  The slab is a middle slab; with itself
as both neighbours the received ghost planes are its own boundary planes (a z-mirror), so the
reference result is the same run with the hook served by plain device-to-device copies.
Bit-for-bit equality => the RCCL transport moved the right bytes at the right time.

Prints "RCCL-OK" on success."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import wafer_amd as wa
    from wafer_amd.slab import TorchSlabComm

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29433")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    class SelfNeighbours(TorchSlabComm):
        lower = 0   # both z-neighbours are this rank
        upper = 0

    shape, ext, pl = (136, 40, 96), int(os.environ.get("WAFER_TEST_EXT", "1")), 24
    par = wa.Params(*shape, dn=0.2, dt=0.004, central_difference=ext, z_begin=2 * pl, z_count=pl,
                    halo_depth=2 * ext, max_states=2)

    def run(install):
        with wa.Context(par) as ctx:
            keep = install(ctx)
            ctx.set_potential("Harmonic")
            ctx.set_initial_condition("Gaussian", seed=3)
            ctx.evolve(0, 9)                      # fused passes + one odd trailing step
            ground = ctx.download_phi()
            n2 = ctx.norm2()                      # all-reduce (world of one: identity)
            ctx.normalise(n2)
            ctx.push_state()
            ctx.set_initial_condition("Gaussian", seed=4)
            ctx.evolve(1, 6)                      # excited state: scalars all-reduced every step
            obs = ctx.observables()
            del keep
            return ground, ctx.download_phi(), n2, obs

    calls = {"halo": 0, "allreduce": 0}

    def rccl(ctx):
        comm = SelfNeighbours(ctx, 0, 1, dev)
        comm.warm_up()
        halo, allreduce = comm._halo_hook, comm._allreduce_hook

        def h(*a):
            calls["halo"] += 1
            return halo(*a)

        def r(*a):
            calls["allreduce"] += 1
            return allreduce(*a)
        ctx.set_comm_hooks(h, r)
        return comm

    hip = None
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                hip = C.CDLL(line.split()[-1])
                break
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]

    def copies(ctx):
        def halo(slo, shi, rlo, rhi, nbytes, stream):
            # what the self-neighbour exchange delivers: the first receive posted (lower ghost
            # planes) matches the first send posted (lower boundary planes), then upper <- upper
            assert hip.hipMemcpyAsync(rlo, slo, nbytes, 3, stream) == 0
            assert hip.hipMemcpyAsync(rhi, shi, nbytes, 3, stream) == 0
            return 0
        ctx.set_comm_hooks(halo, lambda ptr, count, stream: 0)
        return None

    got = run(rccl)
    want = run(copies)
    assert calls["halo"] >= 5 + 6 and calls["allreduce"] >= 6, calls
    assert np.array_equal(got[0], want[0]), "ground-state slab differs"
    assert np.array_equal(got[1], want[1]), "excited-state slab differs"
    assert got[2] == want[2] and got[3] == want[3]
    assert np.isfinite(got[1]).all() and abs(got[3]["norm2"]) > 0
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("RCCL-OK", calls, flush=True)


if __name__ == "__main__":
    main()
