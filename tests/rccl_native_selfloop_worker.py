"""wafer_amd.slab.NativeRcclSlabComm on the one GPU of the box: libwafer_rccl.so's hooks (RCCL's C
API, torch's RCCL library in this process) with the rank as its own z-neighbour, against the same
run with device copies -- bit for bit, ground state (fused, overlapped) and an excited-state block.
Prints "NATIVE-OK"."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import wafer_amd as wa
    from wafer_amd.slab import NativeRcclSlabComm

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    shape, ext, pl = (136, 40, 96), 1, 24
    par = wa.Params(*shape, dn=0.2, dt=0.004, central_difference=ext, z_begin=2 * pl, z_count=pl, halo_depth=2 * ext, max_states=2)

    def run(install):
        with wa.Context(par) as ctx:
            keep = install(ctx)
            ctx.set_potential("Harmonic")
            ctx.set_initial_condition("Gaussian", seed=3)
            ctx.evolve(0, 9)
            ground = ctx.download_phi()
            n2 = ctx.norm2()
            ctx.normalise(n2)
            ctx.push_state()
            ctx.set_initial_condition("Gaussian", seed=4)
            ctx.evolve(1, 6)
            out = ground, ctx.download_phi(), n2, ctx.observables()
            if keep is not None:
                calls = keep.halo_calls()
                keep.close()
                return out + (calls,)
            return out + (0,)

    hip = None
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                hip = C.CDLL(line.split()[-1])
                break
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]

    def native(ctx):
        comm = NativeRcclSlabComm(ctx, 0, 1, dev, self_neighbours=True)
        comm.warm_up()
        # the scalar all-reduce picked by measurement (ncclAllReduce against the device-side mailbox): whichever wins, the sums
        # below must not change
        choice = comm.pick_allreduce(calls=50)
        assert choice["chosen"] in ("mailbox", "ncclAllReduce") and choice["ncclAllReduce_us"] > 0, choice
        print("all-reduce:", choice, flush=True)
        return comm

    def copies(ctx):
        def halo(slo, shi, rlo, rhi, nbytes, stream):
            # one rank as both neighbours: wafer_rccl_halo closes a ring (lower ghost planes <- upper boundary planes)
            if rlo:
                assert hip.hipMemcpyAsync(rlo, shi, nbytes, 3, stream) == 0
            if rhi:
                assert hip.hipMemcpyAsync(rhi, slo, nbytes, 3, stream) == 0
            return 0
        ctx.set_comm_hooks(halo, lambda ptr, count, stream: 0)
        return None

    got, want = run(native), run(copies)
    assert got[4] >= 11, got[4]
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert got[2] == want[2] and got[3] == want[3] and np.isfinite(got[1]).all()
    print("NATIVE-OK halo_calls", got[4], flush=True)


if __name__ == "__main__":
    main()
