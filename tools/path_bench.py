#!/usr/bin/env python3
"""Times every kernel family of the path on one GPU at a given grid:
ground-state step, excited-state steps (wnum 1..3), observables, norm2,
normalise, orthogonalise.  Reports ms and the fraction of the HBM roofline at
the algorithmic byte counts of SURVEY.md 8(d)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wafer_amd

ap = argparse.ArgumentParser()
ap.add_argument("--grid", default="512,512,512")
ap.add_argument("--cd", type=int, default=1)
ap.add_argument("--dtype", default="f64")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--rounds", type=int, default=3, help="evolve calls per wnum; the median is reported")
args = ap.parse_args()
nx, ny, nz = (int(s) for s in args.grid.split(","))
pts = nx * ny * nz
es = {"f64": 8, "f32": 4}[args.dtype]
par = wafer_amd.Params(nx, ny, nz, dn=0.05, dt=5e-4, central_difference=args.cd, dtype=args.dtype, max_states=3)


def report(name, ms, bytes_per_pt):
    gbps = pts * bytes_per_pt / ms / 1e6
    print(json.dumps({"op": name, "ms": round(ms, 4), "algorithmic_B_per_pt": bytes_per_pt,
                      "GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / 8000, 4)}), flush=True)


with wafer_amd.Context(par) as ctx:
    ctx.set_potential("Coulomb")
    for i in range(3):
        ctx.set_initial_condition("Gaussian", seed=i + 1)
        ctx.normalise(ctx.norm2())
        ctx.push_state()
    ctx.set_initial_condition("Boolean")
    ctx.evolve(0, 600)   # clocks up (a cold device runs the first launches 10-40 % slow)
    ctx.set_initial_condition("Boolean")
    for wnum, b in ((0, 4 * es), (1, 10 * es), (2, 14 * es), (3, 18 * es)):
        ctx.evolve(wnum, 4)
        per = []
        for _ in range(args.rounds):
            ctx.evolve(wnum, args.steps)
            ms, steps = ctx.last_evolve_ms()
            per.append(ms / steps)
        report(f"evolve wnum={wnum}", sorted(per)[len(per) // 2], b)
        ctx.set_initial_condition("Boolean")

    def timed(fn, reps=10):
        fn(); ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    report("observables (incl. host sync)", timed(ctx.observables), 2 * es)
    report("norm2 (incl. host sync)", timed(ctx.norm2), es)
    report("normalise", timed(lambda: ctx.normalise(1.0)), 2 * es)
    report("orthogonalise(3)", timed(lambda: ctx.orthogonalise(3)), (2 + 4 + 4 + 3) * es)
