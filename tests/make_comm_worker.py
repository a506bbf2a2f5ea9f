"""wafer_amd.slab.make_slab_comm in a one-rank RCCL process group: the native hooks by default,
the agreed fall-back to the torch.distributed hooks when they cannot be installed.  Prints "COMM-OK"."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import wafer_amd as wa
    from wafer_amd import slab

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    par = wa.Params(64, 32, 48, dn=0.2, dt=0.004, z_begin=16, z_count=16, halo_depth=2)
    with wa.Context(par) as ctx:
        comm, name = slab.make_slab_comm(ctx, 0, 1, dev)
        assert isinstance(comm, slab.NativeRcclSlabComm) and "native" in name, name
        comm.warm_up()
        comm.close()

        class Broken(slab.NativeRcclSlabComm):
            def __init__(self, *a, **k):
                raise RuntimeError("simulated: libwafer_rccl.so cannot attach")
        real, slab.NativeRcclSlabComm = slab.NativeRcclSlabComm, Broken
        try:
            comm, name = slab.make_slab_comm(ctx, 0, 1, dev)
        finally:
            slab.NativeRcclSlabComm = real
        assert isinstance(comm, slab.TorchSlabComm) and "torch.distributed" in name, name
        comm.warm_up()
        # a LOCAL failure (library missing on this node) is agreed on before anything collective
        real_pre = slab.NativeRcclSlabComm.precheck

        def broken_pre(rank):
            raise ImportError("simulated: libwafer_rccl.so is missing")
        slab.NativeRcclSlabComm.precheck = staticmethod(broken_pre)
        try:
            comm, name = slab.make_slab_comm(ctx, 0, 1, dev)
        finally:
            slab.NativeRcclSlabComm.precheck = staticmethod(real_pre)
        assert isinstance(comm, slab.TorchSlabComm) and "torch.distributed" in name, name
        comm, name = slab.make_slab_comm(ctx, 0, 1, dev, "torch")
        assert isinstance(comm, slab.TorchSlabComm)
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("COMM-OK", flush=True)


if __name__ == "__main__":
    main()
