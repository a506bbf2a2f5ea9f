#!/usr/bin/env python3
"""Compute-side cost of the z-slab decomposition on ONE GPU: a middle slab of the N-GPU bench
grid (1024 x 1024 x 128 owned planes, 2*ext ghost planes, boundary-first overlap on the second
stream) against the same 2^27 points as an undecomposed grid.  The halo hook is a loopback
(device-to-device copy of the slab's own boundary planes on the hook's stream, same bytes as a
neighbour would send), so what is measured is everything except the fabric: split launches,
redundant ghost-plane updates, stream fork/join.

    python tools/slab_overhead.py [--steps 100] [--planes 128] [--cd 1]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wafer_amd  # noqa: E402


def hip_runtime():
    wafer_amd.load_library()
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                return C.CDLL(line.split()[-1])
    raise RuntimeError("no HIP runtime mapped")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--planes", type=int, default=128)
    ap.add_argument("--xy", type=int, default=1024)
    ap.add_argument("--cd", type=int, default=1)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32", "f32fast"], help="storage / arithmetic of the contexts (config #5: f32)")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--wnum", type=int, default=0, help="excited-state steps against this many stored states")
    ap.add_argument("--modes", default="2,1,0", help="halo schedules to time (wafer_set_overlap modes; 4 = peer copies, its schedule from WAFER_COPY_SCHED)")
    ap.add_argument("--cycles", default="1", help="fused passes per halo exchange to time (wafer_set_halo_cycle): e.g. 1,2,3")
    ap.add_argument("--torch-hooks", action="store_true", help="with --rccl: also the torch.distributed hooks")
    ap.add_argument("--per-pass", type=int, default=0, help="ghost planes one fused pass consumes (default: 3 for --cd 1 = the three-step kernel, else 2*cd)")
    ap.add_argument("--rccl", action="store_true",
                    help="also serve the hook with RCCL send/recv to this same rank (wafer_amd.slab.TorchSlabComm, "
                         "world of one): adds the host cost of the Python hook + batch_isend_irecv per pass")
    args = ap.parse_args()
    hip = hip_runtime()
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    n, pl, ext = args.xy, args.planes, args.cd
    kw = dict(dn=0.02, dt=8e-5, mass=2.35, sig=0.223, central_difference=ext, max_states=max(1, args.wnum), dtype=args.dtype)
    out = {}

    def run(par, hooks, overlap=True):
        with wafer_amd.Context(par) as ctx:
            if hooks:
                ctx.set_comm_hooks(*hooks)
                if int(overlap) in (3, 4):   # peer stores / peer copies, the slab as its own neighbour on both sides
                    rec = ctx.peer_export()
                    ctx.peer_connect(rec, rec)
                ctx.set_overlap(overlap)
                ctx.set_halo_cycle(max(1, par.halo_depth // (args.per_pass or (3 if ext == 1 else 2 * ext))))
            ctx.set_potential("SimpleCornell")
            for j in range(args.wnum):
                ctx.set_initial_condition("Gaussian", seed=j + 1)
                ctx.normalise(ctx.norm2())
                ctx.push_state()
            ctx.set_initial_condition("Boolean")
            ctx.evolve(args.wnum, 100)
            ctx.synchronize()
            best = None
            for _ in range(5):   # median of 5 (clock ramp, other tenants)
                ctx.evolve(args.wnum, args.steps)
                ms, k = ctx.last_evolve_ms()
                best = sorted((best or []) + [ms / k])
            return best[len(best) // 2]

    out["undecomposed_ms_per_step"] = run(wafer_amd.Params(n, n, pl, **kw), None)
    calls = {"halo": 0, "bytes": 0}

    def halo(slo, shi, rlo, rhi, nbytes, stream):
        # my own boundary planes stand in for the neighbours': same sizes, same stream ordering
        if rlo:
            assert hip.hipMemcpyAsync(rlo, shi, nbytes, 3, stream) == 0
        if rhi:
            assert hip.hipMemcpyAsync(rhi, slo, nbytes, 3, stream) == 0
        calls["halo"] += 1
        calls["bytes"] = nbytes
        return 0

    def allreduce(ptr, count, stream):
        return 0

    cycles = [int(c) for c in args.cycles.split(',')]

    def mid_params(cycle):
        per_pass = args.per_pass or (3 if ext == 1 else 2 * ext)
        return wafer_amd.Params(n, n, pl * args.world, z_begin=pl * (args.world // 2), z_count=pl, halo_depth=per_pass * cycle, **kw)

    def tag(cycle):
        return "" if cycle == 1 else f"_cycle{cycle}"

    mid = mid_params(1)
    for cycle in cycles:
        for overlap in [int(m) for m in args.modes.split(',')]:
            calls["halo"] = 0
            out[f"slab_ms_per_step_overlap_{int(overlap)}{tag(cycle)}"] = run(mid_params(cycle), (halo, allreduce), overlap)
    if args.rccl:
        import torch
        import torch.distributed as dist
        from wafer_amd.slab import TorchSlabComm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29455")
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

        class SelfNeighbours(TorchSlabComm):
            lower = 0
            upper = 0

        for overlap in ([int(m) for m in args.modes.split(',')] if args.torch_hooks else []):
            with wafer_amd.Context(mid) as ctx:
                comm = SelfNeighbours(ctx, 0, 1, dev)
                comm.warm_up()
                ctx.set_overlap(overlap)
                ctx.set_potential("SimpleCornell")
                for j in range(args.wnum):
                    ctx.set_initial_condition("Gaussian", seed=j + 1)
                    ctx.normalise(ctx.norm2())
                    ctx.push_state()
                ctx.set_initial_condition("Boolean")
                ctx.evolve(args.wnum, 100)
                ctx.synchronize()
                ts = []
                for _ in range(5):
                    ctx.evolve(args.wnum, args.steps)
                    ms, k = ctx.last_evolve_ms()
                    ts.append(ms / k)
                out[f"slab_rccl_self_ms_per_step_overlap_{int(overlap)}"] = sorted(ts)[2]
                del comm
        from wafer_amd.slab import NativeRcclSlabComm
        for cycle, overlap in [(cy, int(m)) for cy in cycles for m in args.modes.split(',')]:
            with wafer_amd.Context(mid_params(cycle)) as ctx:
                comm = NativeRcclSlabComm(ctx, 0, 1, dev, self_neighbours=True)
                comm.warm_up()
                if int(overlap) in (3, 4):
                    rec = ctx.peer_export()
                    ctx.peer_connect(rec, rec)
                ctx.set_overlap(overlap)
                ctx.set_halo_cycle(cycle)
                ctx.set_potential("SimpleCornell")
                for j in range(args.wnum):
                    ctx.set_initial_condition("Gaussian", seed=j + 1)
                    ctx.normalise(ctx.norm2())
                    ctx.push_state()
                ctx.set_initial_condition("Boolean")
                ctx.evolve(args.wnum, 100)
                ctx.synchronize()
                ts = []
                for _ in range(5):
                    ctx.evolve(args.wnum, args.steps)
                    ms, k = ctx.last_evolve_ms()
                    ts.append(ms / k)
                out[f"slab_native_rccl_self_ms_per_step_overlap_{int(overlap)}{tag(cycle)}"] = sorted(ts)[2]
                comm.close()
        torch.cuda.synchronize()
        dist.destroy_process_group()
    out["halo_calls_per_step_last_loopback_run"] = calls["halo"] / (100 + 5 * args.steps)
    out["halo_bytes_per_direction_per_call"] = calls["bytes"]
    first = int(args.modes.split(',')[0])
    out["slab_over_undecomposed"] = out[f"slab_ms_per_step_overlap_{first}"] / out["undecomposed_ms_per_step"]
    out["grid"] = [n, n, pl]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
