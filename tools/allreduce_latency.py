#!/usr/bin/env python3
"""Microseconds per all-reduce of 4 doubles on ONE GPU, one process: ncclAllReduce through the native hook library (a
communicator of one rank: the floor of RCCL's path -- kernel launch + its own set-up, no peer) against the device-side
mailbox all-reduce (include/wafer_mailbox.h) of one rank.  HIP events around 500 back-to-back calls on one stream.
The two- and three-process figures of the mailbox (ranks on one GPU through HIP IPC) come from
tests/test_gpu_multiprocess.py::test_device_side_allreduce_between_processes."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import wafer_amd
    from wafer_amd.slab import NativeRcclSlabComm
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29477")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    par = wafer_amd.Params(64, 64, 64, dn=0.1, dt=1e-3, z_begin=16, z_count=16)
    out = {}
    for name, mailbox in (("ncclAllReduce_one_rank", False), ("wafer_mailbox_one_rank", True)):
        with wafer_amd.Context(par) as ctx:
            comm = NativeRcclSlabComm(ctx, 0, 1, dev, self_neighbours=True, mailbox=mailbox)
            comm.warm_up()
            L = comm._L
            L.wafer_rccl_allreduce_now.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
            stream = torch.cuda.Stream(device=dev)
            t = torch.ones(4, dtype=torch.float64, device=dev)
            with torch.cuda.stream(stream):
                for _ in range(20):
                    L.wafer_rccl_allreduce_now(comm._handle, t.data_ptr(), 4, stream.cuda_stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(500):
                    L.wafer_rccl_allreduce_now(comm._handle, t.data_ptr(), 4, stream.cuda_stream)
                e1.record(stream)
                stream.synchronize()
            out[name + "_us"] = e0.elapsed_time(e1) * 1e3 / 500
            comm.close()
    dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
