#!/usr/bin/env python3
"""A/B sweep of the stencil kernel variants and launch parameters on one GPU.
Interleaved rounds in ONE process (cdna_hip_programming.md rule 24); reports
median and min of the HIP-event time per step.

    python tools/stencil_sweep.py --grid 512,512,512 --rounds 5 --steps 20 \
        --configs "v=0" "v=1" "v=1,zchunk=32" "v=1,blocks=4096"
"""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wafer_amd  # noqa: E402


def parse_cfg(s):
    d = {}
    for kv in s.split(","):
        if kv:
            k, v = kv.split("=")
            d[k] = v
    return d


ENV_KEYS = (("WAFER_ZCHUNK", "zchunk"), ("WAFER_TARGET_BLOCKS", "blocks"), ("WAFER_LDS_RY", "ry"), ("WAFER_XCD_SWIZZLE", "xcd"),
            ("WAFER_NT", "nt"), ("WAFER_ABV", "abv"),
            ("WAFER_VGEN", "vgen"), ("WAFER_XF_NW", "xfnw"), ("WAFER_XF_DEEP", "deep"),
            ("WAFER_F3_SCHED", "sched"), ("WAFER_X2", "x2"), ("WAFER_X2_RY", "x2ry"), ("WAFER_X2_MAX_K", "x2k"), ("WAFER_F3_XS", "xs"), ("WAFER_F2_WIDE", "f2w"), ("WAFER_F3_PLAIN_DOWN", "down"), ("WAFER_F3_ROUNDS", "rounds"))


def make_ctx(par, cfg, args):
    """the tuning variables are read once, when a context is created (wafer_tuning.h): one context per configuration"""
    for env, key in ENV_KEYS:
        if key in cfg:
            os.environ[env] = cfg[key]
        else:
            os.environ.pop(env, None)
    ctx = wafer_amd.Context(par)
    ctx.set_stencil_variant(int(cfg.get("v", -1)))
    ctx.set_potential(args.potential)
    for i in range(args.wnum):
        ctx.set_initial_condition("Gaussian", seed=i + 1)
        ctx.normalise(ctx.norm2())
        ctx.push_state()
    ctx.set_initial_condition("Boolean")
    return ctx


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="512,512,512")
    ap.add_argument("--cd", type=int, default=1)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--wnum", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--configs", nargs="+", default=["v=0", "v=1"])
    ap.add_argument("--potential", default="Coulomb")
    ap.add_argument("--dn", type=float, default=0.05)
    ap.add_argument("--dt", type=float, default=5e-4)
    ap.add_argument("--mass", type=float, default=1.0)
    ap.add_argument("--sig", type=float, default=0.223)
    args = ap.parse_args()
    nx, ny, nz = (int(s) for s in args.grid.split(","))
    par = wafer_amd.Params(nx, ny, nz, dn=args.dn, dt=args.dt, mass=args.mass, sig=args.sig,
                           central_difference=args.cd, dtype=args.dtype, max_states=max(1, args.wnum))
    bpu = {"f64": 32, "f32": 16, "f32fast": 16}[args.dtype]
    if True:
        cfgs = [parse_cfg(s) for s in args.configs]
        ctxs = [make_ctx(par, cfg, args) for cfg in cfgs]
        times = [[] for _ in cfgs]
        for r in range(args.rounds + 1):
            for i, ctx in enumerate(ctxs):
                ctx.evolve(args.wnum, args.steps)
                ms, steps = ctx.last_evolve_ms()
                if r > 0:  # round 0 is warm-up
                    times[i].append(ms / steps)
                if args.wnum == 0:
                    ctx.set_initial_condition("Boolean")  # keep values in range over long sweeps
        for ctx in ctxs:
            ctx.close()
        pts = nx * ny * nz
        for s, t in zip(args.configs, times):
            med, mn = statistics.median(t), min(t)
            print(json.dumps({"config": s, "ms_per_step_median": round(med, 4), "ms_per_step_min": round(mn, 4),
                              "Gupdates_per_s": round(pts / med / 1e6, 2),
                              "GBps_algorithmic": round(pts * bpu / med / 1e6, 1),
                              "frac_of_8TBps": round(pts * bpu / med / 1e6 / 8000, 4)}), flush=True)


if __name__ == "__main__":
    main()
