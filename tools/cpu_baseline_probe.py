#!/usr/bin/env python3
"""How fast can this HOST run the oracle's evolve (bench.py's cpu_baseline leg), and what decides it?

Each configuration runs in a process of its own (the OpenMP runtime reads OMP_PROC_BIND / OMP_PLACES once, when it
loads): threads x {unbound, bound: OMP_PROC_BIND=spread OMP_PLACES=cores} x {portable build, -march=native build}, on the
bench grid (512^3 fp64 Coulomb, ThreePoint), a few steps each, next to the host's OpenMP copy bandwidth on the same
threads.  One JSON line per configuration on stdout.  No GPU is touched.

    python tools/cpu_baseline_probe.py [--grid 512] [--steps 4] [--threads 8,16,32,64,128]
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(grid, steps, threads, native):
    sys.path.insert(0, ROOT)
    from oracle import wafer_oracle as wo
    if native:
        assert wo.use_native()
    wo.set_threads(threads)
    cfg = wo.Config(grid, grid, grid, ext=1, potential="Coulomb", dn=0.05, dt=5e-4, mass=1.0, sig=0.223)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    del v
    phi = wo.initial_condition(cfg, "Boolean")
    wo.evolve(cfg, 0, a, b, phi, [], 1)
    t0 = time.perf_counter()
    wo.evolve(cfg, 0, a, b, phi, [], steps)
    dt = time.perf_counter() - t0
    rate = grid ** 3 * steps / dt
    print(json.dumps({"grid": grid, "threads": threads, "native": native, "bind": os.environ.get("OMP_PROC_BIND", ""),
                      "placement": wo.thread_placement(), "updates_per_s": rate, "effective_GBps_48B": rate * 48 / 1e9,
                      "host_copy_GBps": wo.host_copy_gbps(1 << 30, 3), "seconds": dt}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--threads", default="")
    ap.add_argument("--one", default="")
    args = ap.parse_args()
    if args.one:
        t, n = args.one.split(",")
        one(args.grid, args.steps, int(t), n == "1")
        return
    ncpu = len(os.sched_getaffinity(0))
    threads = [int(x) for x in args.threads.split(",")] if args.threads else sorted({max(1, ncpu // d) for d in (1, 2, 4, 8, 16)})
    try:
        print(json.dumps({"cpus_in_affinity_mask": ncpu, "cgroup_cpu_max": open("/sys/fs/cgroup/cpu.max").read().strip()}), flush=True)
    except OSError:
        print(json.dumps({"cpus_in_affinity_mask": ncpu}), flush=True)
    for native in (0, 1):
        for bind in (0, 1):
            for t in threads:
                env = dict(os.environ)
                for k in ("OMP_PROC_BIND", "OMP_PLACES"):
                    env.pop(k, None)
                if bind:
                    env.update(OMP_PROC_BIND="spread", OMP_PLACES="cores")
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--grid", str(args.grid), "--steps", str(args.steps),
                                    "--one", f"{t},{native}"], env=env, capture_output=True, text=True)
                sys.stdout.write(r.stdout if r.returncode == 0 else json.dumps({"threads": t, "native": native, "bind": bind, "error": r.stderr[-400:]}) + "\n")
                sys.stdout.flush()


if __name__ == "__main__":
    main()
