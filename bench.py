#!/usr/bin/env python3
"""Benchmark of the grid::evolve hot path (BASELINE.json metric: grid-point
updates/sec on 512^3 fp64, achieved HBM GB/s vs peak).

    python bench.py --gpus N --steps K --warmup W

A "step" is ONE imaginary-time step (grid.rs:562-686, wnum = 0) of the whole
grid.  N = 1: the 512^3 fp64 ThreePoint Coulomb grid of BASELINE configs[2],
ground-state evolve, potential / phi generated in HBM (no host traffic in the
timed region).  N > 1 (launched by torch.distributed.run, one rank per GPU):
the grid is (1024, 1024, 128*N) -- 2^27 points per GPU, i.e. configs[3]'s
1024^3 at N = 8 -- z-slab decomposed with RCCL halo exchange; "weak" scaling.

Prints ONE JSON line on rank 0 (contract in the task statement) with the extra
objects "roofline" and "cpu_baseline".
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
BYTES_PER_UPDATE = {"f64": 32, "f32": 16, "f32fast": 16}  # phi, a, b in + phi' out


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32", "f32fast"])
    ap.add_argument("--grid", default=None, help="override: NX,NY,NZ (global work area)")
    ap.add_argument("--cd", type=int, default=1, help="central difference ext: 1/2/3")
    ap.add_argument("--potential", default="Coulomb")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU time of the oracle sample")
    ap.add_argument("--variant", type=int, default=-1, help="stencil kernel variant (-1 = default)")
    return ap.parse_args()


def physical_cores() -> int:
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(min(n, len(os.sched_getaffinity(0))))
    except Exception:
        pass
    return max(1, len(os.sched_getaffinity(0)))


def cpu_baseline(shape, ext, potential, dn, dt, mass, target_seconds):
    """The oracle (kind "port": a C restatement of Wafer's rayon path, same pass
    structure: stencil into work, copy back) timed on this host's cores on a
    bounded number of steps of the SAME grid.  Reported, not the target."""
    from oracle import wafer_oracle as wo
    cores = physical_cores()  # Wafer's own thread rule, main.rs:190-196
    wo.set_threads(cores)
    cfg = wo.Config(*shape, ext=ext, potential=potential, dn=dn, dt=dt, mass=mass)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    del v
    phi = wo.initial_condition(cfg, "Boolean")
    t0 = time.perf_counter()
    wo.evolve(cfg, 0, a, b, phi, [], 1)
    t1 = time.perf_counter() - t0
    steps = int(max(2, min(200, target_seconds / max(t1, 1e-6))))
    t0 = time.perf_counter()
    wo.evolve(cfg, 0, a, b, phi, [], steps)
    dt_s = time.perf_counter() - t0
    pts = shape[0] * shape[1] * shape[2]
    return {
        "value": pts * steps / dt_s,
        "unit": "updates/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{steps} steps of the same {shape[0]}x{shape[1]}x{shape[2]} fp64 {potential} grid "
                  f"(oracle/wafer_oracle.c wo_evolve, OpenMP, {dt_s:.1f} s)",
    }


def pmc_traffic(kernel_name: str):
    """HBM bytes per launch from the committed rocprofv3 --pmc summary, if any."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        for k, v in d.get("kernels", {}).items():
            if k in kernel_name or kernel_name in k:
                return v.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def main():
    # RCCL caches its parameters at first use, which is torch's own communicator: the channel limit the
    # slab transport wants (wafer_rccl_hooks.h) has to be in the environment before that
    os.environ.setdefault("NCCL_MAX_P2P_NCHANNELS", "8")
    args = parse_args()
    # stdout carries exactly ONE line, the JSON result: native libraries that write to the C stdout
    # (RCCL prints a version banner there, flushed at exit) are sent to stderr instead
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = args.gpus
    if world > 1 and world != n_gpus:
        raise SystemExit(f"--gpus {n_gpus} but WORLD_SIZE={world}")

    import torch
    import wafer_amd

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    # WAFER_BENCH_TRANSPORT=host: halo planes staged through host memory over gloo, ranks folded
    # onto the GPUs present -- only for exercising the N > 1 leg on a one-GPU box
    # (tests/test_gpu_multiprocess.py); the default is RCCL, one GPU per rank, the hooks served by
    # RCCL's C API directly (WAFER_TRANSPORT=torch: through torch.distributed)
    host_transport = os.environ.get("WAFER_BENCH_TRANSPORT", "rccl") == "host"
    # a launcher that narrows each rank's view to its own GPU (ROCR_/HIP_VISIBLE_DEVICES per rank) leaves
    # one visible device with index 0; the host-staged test transport folds ranks onto the GPUs present
    if host_transport or local_rank >= torch.cuda.device_count():
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)

    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if host_transport:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    ext = args.cd
    if args.grid:
        shape = tuple(int(s) for s in args.grid.split(","))
    elif n_gpus == 1:
        shape = (512, 512, 512)
    else:
        shape = (1024, 1024, 128 * n_gpus)
    if n_gpus == 1:
        dn, dt, mass = 0.05, 5e-4, 1.0
        potential = args.potential
    else:
        dn, dt, mass = 0.02, 8e-5, 2.35           # config #4: SimpleCornell, sig 0.223
        potential = "SimpleCornell" if args.potential == "Coulomb" else args.potential

    nz = shape[2]
    z_begin, z_count = 0, 0
    if world > 1:
        from wafer_amd import slab
        z_begin, z_count = slab.partition(nz, world, rank)
    par = wafer_amd.Params(shape[0], shape[1], shape[2], dn=dn, dt=dt, mass=mass, sig=0.223,
                           central_difference=ext, dtype=args.dtype, max_states=1, device=local_rank,
                           z_begin=z_begin, z_count=z_count,
                           halo_depth=2 * ext if world > 1 else 0)  # 2*ext ghost planes: two fused steps per exchange
    ctx = wafer_amd.Context(par)
    if args.variant >= 0:
        ctx.set_stencil_variant(args.variant)
    comm, transport_name = None, None
    if world > 1:
        comm, transport_name = slab.make_slab_comm(ctx, rank, world, torch.device("cuda", local_rank),
                                                   "host" if host_transport else None)
        comm.warm_up()   # RCCL channel set-up is not part of any step
    ctx.set_potential(potential)
    ctx.set_initial_condition("Boolean")   # deterministic, "good for benchmarks" (config.rs:168)
    ctx.synchronize()

    def barrier():
        # this rank's queued work first, then every rank's, then whatever the barrier itself queued
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # the box itself next to the 8 TB/s datasheet peak: the bandwidth its own
    # properties imply and what a flat streaming kernel reaches on the same buffers.  Measured
    # during set-up, before the warm-up steps (it also brings the clocks up from idle).
    device_info = None
    if True:   # every rank (the same set-up work everywhere); rank 0 reports
        try:
            di = ctx.device_info()
            mclk_khz, width = di["memory_clock_khz"], di["memory_bus_bits"]
            device_info = {
                "name": di["name"], "arch": di["arch"], "compute_units": di["compute_units"],
                "hbm_GB": round(di["total_bytes"] / 2**30, 1),
                "memory_clock_MHz": mclk_khz / 1e3, "memory_bus_bits": width,
                # HBM3E moves 4 bits per pin per reported memory clock (2 GHz -> 8 Gb/s/pin):
                # 8192 pins x 8 Gb/s = 8.19 TB/s, the datasheet's "8 TB/s"
                "hbm_GBps_from_props": 4.0 * mclk_khz * 1e3 * width / 8 / 1e9,
                "measured_stream_GBps_1r1w": round(ctx.stream_bandwidth(1, 300), 1),
                "measured_stream_GBps_3r1w": round(ctx.stream_bandwidth(3, 300), 1),
            }
        except Exception as e:  # informational
            device_info = {"error": repr(e)}

    ctx.set_initial_condition("Boolean")   # the streaming kernel used phi's second buffer as scratch
    # N > 1: the halo exchange hides behind the interior update (mode 1: boundary planes and exchange on
    # a second stream; mode 2: boundary planes in-stream, only the exchange on the second stream; mode 3:
    # as 1 with the streams swapping roles every pass) or
    # follows the whole slab's update (mode 0).  Which is fastest depends on the fabric, which this code
    # has never seen: all four are timed over a few untimed set-up steps and every rank takes the mode
    # that is fastest for the slowest rank (the default, 1, unless another wins by more than 2 %).
    overlap_choice = None
    if dist is not None and os.environ.get("WAFER_OVERLAP", "") == "" and args.steps >= 8:
        trial = {}
        for mode in (1, 2, 3, 0):
            ctx.set_overlap(mode)
            ctx.evolve(0, 8)
            barrier()
            t_ = time.perf_counter()
            ctx.evolve(0, 40)
            barrier()
            tt = torch.tensor([time.perf_counter() - t_], dtype=torch.float64,
                              device="cpu" if host_transport else f"cuda:{local_rank}")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            trial[mode] = float(tt[0]) / 40 * 1e3
        best = min(trial, key=lambda m: trial[m] * (1.0 if m == 1 else 1.02))
        ctx.set_overlap(best)
        overlap_choice = {"mode": best, "ms_per_step": {"1_overlap": trial[1], "2_overlap_boundary_in_stream": trial[2],
                                                        "3_overlap_alternating_streams": trial[3], "0_no_overlap": trial[0]}}
        ctx.set_initial_condition("Boolean")
    if args.warmup > 0:
        ctx.evolve(0, args.warmup)
    barrier()
    t0 = time.perf_counter()
    ctx.evolve(0, args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, ksteps = ctx.last_evolve_ms()     # HIP events on the engine's own stream
    if dist is not None:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64,
                         device="cpu" if host_transport else f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])

    pts_total = shape[0] * shape[1] * shape[2]
    pts_rank = shape[0] * shape[1] * (z_count if z_count else shape[2])
    value = pts_total * args.steps / elapsed
    bpu = BYTES_PER_UPDATE[args.dtype]
    spl = ctx.steps_per_launch()                  # the fused kernel advances two steps per launch
    launch_s = (kernel_ms / 1e3) / max(1, ksteps) * spl
    achieved = pts_rank * bpu * spl / launch_s / 1e9
    kname = ctx.stencil_kernel_name()
    traffic = pmc_traffic(kname) if (n_gpus == 1 and not args.grid and args.dtype == "f64" and ext == 1) else None

    result = {
        "metric": "grid_point_updates_per_sec",
        "value": value,
        "unit": "updates/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,   # the reference publishes no number for this metric
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {
            "workload": f"{shape[0]}x{shape[1]}x{shape[2]} {potential} potential, "
                        f"{ {1: 'ThreePoint', 2: 'FivePoint', 3: 'SevenPoint'}[ext]} stencil, ground-state "
                        f"imaginary-time evolve (grid.rs:544-687), Boolean initial condition"
                        + ("" if n_gpus == 1 else f", z-slabs of {shape[2] // n_gpus} planes per GPU, "
                           + f"halo exchange: {transport_name}"),
            "grid": list(shape),
            "points_per_gpu": pts_rank,
            "parallelism": f"zslab{n_gpus}",
            "kernel": kname,
            **({"halo_overlap": overlap_choice} if overlap_choice else {}),
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic,
            "kernel": kname,
            "avg_launch_ms": launch_s * 1e3,
            "steps_per_launch": spl,
            "algorithmic_bytes_per_launch": pts_rank * bpu * spl,
            # what the kernel really moved per second (PMC traffic / launch time), next to the accounting figure
            "traffic_GBps": (traffic / launch_s / 1e9) if traffic else None,
        },
        "device": device_info,
    }
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline(shape, ext, potential, dn, dt, mass, args.cpu_seconds)
        except Exception as e:  # the baseline is reported, never required
            result["cpu_baseline"] = {"value": None, "unit": "updates/s", "cores": physical_cores(),
                                      "kind": "port", "sample": f"failed: {e!r}"}
    if dist is not None:   # the process group goes first: its work objects refer to the engine's streams
        ctx.synchronize()
        if hasattr(comm, "close"):
            comm.close()       # the native hooks' communicator
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    if rank == 0:
        result_out.write(json.dumps(result) + "\n")
        result_out.flush()


if __name__ == "__main__":
    main()
