#!/usr/bin/env python3
"""A few overlapped fused passes on the bench slab with a loopback hook (or, --rccl, native RCCL to this same rank), for
`rocprofv3 --kernel-trace`:
    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python3 tools/slab_trace.py [--rccl] [--mode 2]
tools/slab_trace.py --parse out/.../t_kernel_trace.csv   prints the per-pass timeline."""
import ctypes as C
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse(path):
    rows = [r for r in csv.DictReader(open(path)) if "step2_fused" in r["Kernel_Name"] or "step3_fused" in r["Kernel_Name"]
            or "ccl" in r["Kernel_Name"].lower() or "copyBuffer" in r["Kernel_Name"] or "wafer_k_gate" in r["Kernel_Name"]
            or "wafer_k_post" in r["Kernel_Name"] or "wafer_k_rendezvous" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-(40 if any('wafer_k_rendezvous' in r['Kernel_Name'] for r in rows) else 24):]
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        print(f"{s / 1e3:10.1f} us -> {e / 1e3:10.1f} us  ({(e - s) / 1e3:7.1f} us)  {r['Kernel_Name'][:48]}")


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--parse":
        return parse(sys.argv[2])
    import wafer_amd
    wafer_amd.load_library()
    hip = None
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                hip = C.CDLL(line.split()[-1])
                break
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]

    def halo(slo, shi, rlo, rhi, nbytes, stream):
        if rlo and shi:
            hip.hipMemcpyAsync(rlo, shi, nbytes, 3, stream)
        if rhi and slo:
            hip.hipMemcpyAsync(rhi, slo, nbytes, 3, stream)
        return 0
    mode = int(sys.argv[sys.argv.index("--mode") + 1]) if "--mode" in sys.argv else 2
    par = wafer_amd.Params(1024, 1024, 1024, dn=0.02, dt=8e-5, mass=2.35, sig=0.223, z_begin=512, z_count=128, halo_depth=3)
    with wafer_amd.Context(par) as ctx:
        if mode >= 3:   # peer stores / peer copies: the slab as its own neighbour
            rec = ctx.peer_export()
            ctx.peer_connect(rec, rec)
        ctx.set_overlap(mode)
        comm = None
        if "--rccl" in sys.argv:
            import torch
            from wafer_amd.slab import NativeRcclSlabComm
            comm = NativeRcclSlabComm(ctx, 0, 1, torch.device("cuda", 0), self_neighbours=True)
            comm.warm_up()
        else:
            ctx.set_comm_hooks(halo, lambda p, n, s: 0)
        ctx.set_potential("SimpleCornell")
        ctx.set_initial_condition("Boolean")
        ctx.evolve(0, 42)
        ctx.synchronize()
        if comm is not None:
            comm.close()


if __name__ == "__main__":
    main()
