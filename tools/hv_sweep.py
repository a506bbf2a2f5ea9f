#!/usr/bin/env python3
"""Experiments on the single-launch slab pass (wafer_set_overlap mode 2) at the bench slab, 1024 x 1024 x 128 of 1024^3,
one GPU, loopback hook (device copies of the slab's own boundary planes): every variant is timed in ONE process,
interleaved, medians of 5.  Variants are WAFER_* settings read at context creation (wafer_tuning.h).

    python tools/hv_sweep.py [--steps 60] [--rccl]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wafer_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--rccl", action="store_true")
    ap.add_argument("--configs", default="")
    args = ap.parse_args()
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
            assert hip.hipMemcpyAsync(rlo, shi, nbytes, 3, stream) == 0
        if rhi and slo:
            assert hip.hipMemcpyAsync(rhi, slo, nbytes, 3, stream) == 0
        return 0

    n, pl, world = 1024, 128, 8
    kw = dict(dn=0.02, dt=8e-5, mass=2.35, sig=0.223, central_difference=1, max_states=1)
    comm_ctx = {}
    if args.rccl:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29466")
        os.environ.setdefault("NCCL_MAX_P2P_NCHANNELS", "8")
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        comm_ctx["dev"] = dev

    def noop_halo(slo, shi, rlo, rhi, nbytes, stream):
        return 0

    def make(env, slab, mode, hook=None):
        for k in list(os.environ):
            if k.startswith("WAFER_") and k not in ("WAFER_PRELOAD_TORCH",):
                del os.environ[k]
        os.environ.update(env)
        par = (wafer_amd.Params(n, n, pl * world, z_begin=pl * (world // 2), z_count=pl, halo_depth=3, **kw) if slab
               else wafer_amd.Params(n, n, pl, **kw))
        ctx = wafer_amd.Context(par)
        comm = None
        if slab:
            if args.rccl:
                from wafer_amd.slab import NativeRcclSlabComm
                comm = NativeRcclSlabComm(ctx, 0, 1, comm_ctx["dev"], self_neighbours=True)
                comm.warm_up()
            else:
                ctx.set_comm_hooks(hook or halo, lambda p, c, s: 0)
            if mode in (3, 4):   # peer stores / peer copies, the slab as its own neighbour
                rec = ctx.peer_export()
                ctx.peer_connect(rec, rec)
            ctx.set_overlap(mode)
        ctx.set_potential("SimpleCornell")
        ctx.set_initial_condition("Boolean")
        ctx.evolve(0, 60)
        ctx.synchronize()
        return ctx, comm

    variants = [
        ("mode2_noop_hook", {}, True, 2, noop_halo),
        ("mode2_noop_hook_no_bump_no_gate", {"WAFER_HV_DEBUG": "32"}, True, 2, noop_halo),
        ("mode2_no_bump_no_gate", {"WAFER_HV_DEBUG": "32"}, True, 2),
        ("mode2_aux_low_priority", {"WAFER_HV_DEBUG": "64"}, True, 2),
        ("undecomposed", {}, False, 0),
        ("undecomposed_halves_schedule", {"WAFER_F3_SCHED": "1"}, False, 0),
        ("undecomposed_halves_schedule_no_shorts", {"WAFER_F3_SCHED": "1", "WAFER_HV_DEBUG": "8"}, False, 0),
        ("undecomposed_halves_schedule_xcd_order", {"WAFER_F3_SCHED": "1", "WAFER_HV_DEBUG": "24"}, False, 0),
        ("mode1", {}, True, 1),
        ("mode0", {}, True, 0),
        ("mode2", {}, True, 2),
        ("mode2_no_acquire", {"WAFER_HV_DEBUG": "4"}, True, 2),
        ("mode3", {}, True, 3),
        ("mode3_halves", {"WAFER_HV_LAYOUT": "3"}, True, 3),
        ("mode4", {}, True, 4),                                       # peer copies under the single launch on two halves
        ("mode4_boundary_first", {"WAFER_COPY_SCHED": "1"}, True, 4),
        ("mode4_exchange_after", {"WAFER_COPY_SCHED": "0"}, True, 4),
        ("mode2_no_shorts", {"WAFER_HV_DEBUG": "8"}, True, 2),
        ("mode2_xcd_order_no_shorts", {"WAFER_HV_DEBUG": "24"}, True, 2),
    ]
    if args.rccl:   # (with the exchange stream at normal priority RCCL's workgroups never win a CU from the queued stencil
        #  workgroups: the waiting workgroups time out after tens of seconds -- not something to spend GPU time on)
        variants = [v for v in variants if "low_priority" not in v[0] and "noop" not in v[0] and "no_bump" not in v[0]]
    if args.configs:
        want = set(args.configs.split(","))
        variants = [v for v in variants if v[0] in want]
    live = []
    for v in variants:
        name, env, slab, mode = v[:4]
        try:
            live.append((name, *make(env, slab, mode, v[4] if len(v) > 4 else None)))
        except Exception as e:  # noqa: BLE001
            print(json.dumps({"variant": name, "error": repr(e)}), flush=True)
    times = {name: [] for name, _, _ in live}
    for _ in range(5):
        for name, ctx, comm in live:
            ctx.evolve(0, args.steps)
            ms, k = ctx.last_evolve_ms()
            times[name].append(ms / k)
    for name, ctx, comm in live:
        t = sorted(times[name])
        print(json.dumps({"variant": name, "ms_per_step_median": t[2], "min": t[0], "max": t[-1]}), flush=True)
    for name, ctx, comm in live:
        if comm is not None:
            comm.close()
        ctx.close()


if __name__ == "__main__":
    main()
