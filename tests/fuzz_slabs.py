#!/usr/bin/env python3
"""Randomised sweep of the DECOMPOSED ground-state path against the oracle (a development aid next to the fixed cases of
tests/test_gpu_slab.py, and run by it in a process of its own): random grids, stencil orders, storage types, 2 - 4 uneven
z-slabs as contexts of this process (device copies stand in for the fabric), every overlap mode the shape allows incl. peer
stores (both pass layouts) and peer copies (mode 4, its three schedules, thin slabs), deep halos, several evolve calls with step counts that leave every kind of remainder, potentials
incl. the ones whose formula singles out z; then excited-state steps (one and two per pass, k = 1 .. 3, thin and uneven slabs).  fp64: the assembled slabs equal the ORACLE bit for bit (FullCornell: device libm,
1e-12) and the all-reduced observables agree to 1e-11; fp32 storage: the slabs equal one context.
    N=40 SEED=3 python tests/fuzz_slabs.py"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("WAFER_PEER_SAME_DEVICE", "1")
import wafer_amd as wa
from oracle import wafer_oracle as wo
from test_gpu_slab import run_slabs, assemble
wo.set_threads(8)
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
bad = 0
for it in range(int(os.environ.get("N", "30"))):
    ext = int(rng.choice([1, 1, 1, 2, 3]))
    world = int(rng.integers(2, 5))
    dtype = str(rng.choice(["f64", "f64", "f64", "f32", "f32fast"]))
    nx = int(rng.choice([40, 64, 128, 130, 136, 200, 256, 264]))
    ny = int(rng.choice([16, 17, 24, 32, 33, 40, 48]))
    depth = int(rng.choice([3, 6])) * ext if ext == 1 else int(rng.choice([2, 4])) * ext
    # overlap modes: 3 (peer stores) is the ThreePoint three-step pass on slabs of six planes or more; 4 (peer copies) is a transport and
    # serves every pass and thickness, under each of its three schedules
    mode = int(rng.choice([0, 1, 2, 4, 4, 5, 6] + ([3, 3] if ext == 1 and depth >= 3 else [])))
    nz = int(rng.integers(max(world * depth, world * 2 * ext, 6 * world if mode == 3 else 0), 70))
    pot, kw = [("Coulomb", {}), ("SimpleCornell", dict(mass=2.35, sig=0.223)), ("QuadWell", {}), ("Harmonic", {}), ("FullCornell", dict(mass=2.35, sig=0.223)),
               ("Periodic", {})][int(rng.integers(0, 6))]
    calls = [int(rng.integers(1, 14)) for _ in range(int(rng.integers(1, 4)))]
    cycle = int(rng.choice([1, 2])) if (ext == 1 and depth == 6 and mode in (0, 1, 4, 5, 6)) or (ext > 1 and depth == 4 * ext and mode in (0, 1, 4, 5, 6)) else 1
    os.environ["WAFER_FUSE3_MIN_NY"] = "1"
    os.environ["WAFER_HV_LAYOUT"] = str(rng.choice([0, 3, 4]))
    os.environ["WAFER_ZCHUNK"] = str(rng.choice([0, 0, 3, 7]))
    os.environ["WAFER_COPY_SCHED"] = str(rng.choice([2, 2, 1, 0]))
    params = dict(dn=0.2, dt=0.004, mass=1.0, sig=1.0)
    params.update(kw)
    tag = (nx, ny, nz, ext, world, dtype, pot, calls, mode, cycle, depth, os.environ["WAFER_HV_LAYOUT"], os.environ["WAFER_ZCHUNK"], os.environ["WAFER_COPY_SCHED"])
    try:
        base = wa.Params(nx, ny, nz, central_difference=ext, dtype=dtype, halo_depth=depth, **params)

        def body(ctx, rank):
            ctx.set_overlap(mode)
            if cycle > 1:
                ctx.set_halo_cycle(cycle)
            ctx.set_potential(pot)
            ctx.set_initial_condition("Boolean")
            for n in calls:
                ctx.evolve(0, n)
            return ctx.download_phi(), ctx.observables()

        res, fab = run_slabs(wa, base, world, body, connect=True if mode >= 4 else None)
        if mode >= 4 and any(fab.halo_calls):
            raise RuntimeError(f"peer copies went through the halo hook: {fab.halo_calls}")
        got = assemble(base, world, [r[0] for r in res])
        if dtype == "f64":
            cfg = wo.Config(nx, ny, nz, ext=ext, potential=pot, **params)
            v = wo.potential_generate(cfg); a, b = wo.ab(cfg, v)
            want = wo.initial_condition(cfg, "Boolean")
            for n in calls:
                wo.evolve(cfg, 0, a, b, want, [], n)
            wobs = wo.observables(cfg, v, want, wo.potential_sub(cfg))
            exact = pot not in ("FullCornell", "Periodic")
            ok = np.array_equal(got, want) if exact else np.allclose(got, want, rtol=0, atol=1e-12)
            ok2 = all(abs(r[1][k] - wobs[k]) <= 1e-11 * max(1.0, abs(wobs[k])) for r in res for k in wobs)
        else:
            with wa.Context(wa.Params(nx, ny, nz, central_difference=ext, dtype=dtype, **params)) as ctx:
                ctx.set_potential(pot); ctx.set_initial_condition("Boolean")
                for n in calls:
                    ctx.evolve(0, n)
                want = ctx.download_phi(); wobs = ctx.observables()
            ok = np.array_equal(got, want)
            ok2 = all(abs(r[1][k] - wobs[k]) <= 1e-11 * max(1.0, abs(wobs[k])) for r in res for k in wobs)
        if not (ok and ok2):
            bad += 1
            print("MISMATCH", tag, ok, ok2, flush=True)
    except Exception as e:
        bad += 1
        print("ERROR", tag, repr(e)[:300], flush=True)
# excited states on slabs: renormalise + Gram-Schmidt every step (grid.rs:674-681), one and two steps per pass, the 1 + k / 2 + 3k sums
# all-reduced -- against one context (1e-12 per cell: the sums associate per slab) and against the oracle (1e-10)
import sys as _sys
_sys.setswitchinterval(1e-4)
for it in range(int(os.environ.get("N", "30")) // 3):
    ext = int(rng.choice([1, 1, 1, 2, 3]))
    world = int(rng.integers(2, 5))
    wnum = int(rng.integers(1, 4))
    nx = int(rng.choice([24, 40, 64, 128, 130, 140]))
    ny = int(rng.choice([8, 16, 17, 24, 40]))
    depth = int(rng.choice([2, 3])) if ext == 1 else ext
    nz = int(rng.integers(max(world * depth, world * 2 * ext) + 1, 44))
    pot = str(rng.choice(["Harmonic", "Coulomb", "SimpleCornell", "Cube"]))
    calls = [int(rng.integers(1, 10)) for _ in range(int(rng.integers(1, 3)))]
    mode = int(rng.choice([0, 1, 2, 4, 5, 6]))
    os.environ["WAFER_COPY_SCHED"] = str(rng.choice([2, 1, 0]))
    os.environ["WAFER_X2_MAX_K"] = "3"
    os.environ["WAFER_X2"] = str(rng.choice([1, 1, 0]))
    os.environ["WAFER_VGEN"] = str(rng.choice([1, 0]))
    xdtype = str(rng.choice(["f64", "f64", "f32"]))   # fp32 storage: the one- and (round 6) two-step kernels on float arrays, fp64 arithmetic
    tag = ("excited", nx, ny, nz, ext, world, wnum, pot, calls, mode, depth, os.environ["WAFER_X2"], os.environ["WAFER_VGEN"], xdtype)
    try:
        params = dict(dn=0.25, dt=0.006, mass=1.3, sig=0.3)
        base = wa.Params(nx, ny, nz, central_difference=ext, max_states=wnum, halo_depth=depth, dtype=xdtype, **params)
        single = wa.Params(nx, ny, nz, central_difference=ext, max_states=wnum, dtype=xdtype, **params)

        def body(ctx, rank=0):
            ctx.set_overlap(mode)
            ctx.set_potential(pot)
            for j in range(wnum):      # orthonormalised random stored states, identical on every slab (the start is keyed by the global cell index)
                ctx.set_initial_condition("Gaussian", seed=40 + 7 * it + j)
                ctx.normalise(ctx.norm2()); ctx.orthogonalise(j); ctx.normalise(ctx.norm2())
                ctx.push_state()
            ctx.set_initial_condition("Gaussian", seed=7 + it)
            start = ctx.download_phi()
            getattr(ctx, "rendezvous", lambda: None)()   # (mode 4 with the ranks as contexts of one process on one device: test_gpu_slab.run_slabs)
            for n in calls:
                ctx.evolve(wnum, n)
            return ctx.download_phi(), ctx.norm2(), start, [ctx.download_state(j) for j in range(wnum)]

        with wa.Context(single) as ctx:
            want, want_n2, start, lowers = body(ctx)
        res, _ = run_slabs(wa, base, world, body, connect=True if mode >= 4 else None)
        got = assemble(base, world, [r[0] for r in res])
        scale = max(1.0, float(np.max(np.abs(want))))
        # (fp32 storage: the slabs' sums differ in their last bits, and a cell's rounding to float can then fall the other way: a float ulp)
        bar1, bar2 = (1e-12, 1e-10) if xdtype == "f64" else (5e-7, 1e-5)
        ok = float(np.max(np.abs(got - want))) / scale <= bar1 and all(abs(r[1] - want_n2) <= max(bar1, 1e-12) * max(1.0, abs(want_n2)) for r in res)
        cfg = wo.Config(nx, ny, nz, ext=ext, potential=pot, **params)
        v = wo.potential_generate(cfg); a, b = wo.ab(cfg, v)
        ref = start.copy()
        for n in calls:
            wo.evolve(cfg, wnum, a, b, ref, lowers, n)
        ok2 = float(np.max(np.abs(got - ref))) / max(1.0, float(np.max(np.abs(ref)))) <= bar2
        if not (ok and ok2):
            bad += 1
            print("MISMATCH", tag, ok, ok2, float(np.max(np.abs(got - want))), float(np.max(np.abs(got - ref))), flush=True)
    except Exception as e:
        bad += 1
        print("ERROR", tag, repr(e)[:300], flush=True)
for name in ("WAFER_X2_MAX_K", "WAFER_X2", "WAFER_VGEN"):
    os.environ.pop(name, None)
for name in ("WAFER_FUSE3_MIN_NY", "WAFER_HV_LAYOUT", "WAFER_ZCHUNK", "WAFER_COPY_SCHED"):
    os.environ.pop(name, None)
print("slab fuzz done,", int(os.environ.get("N", "30")), "cases, bad =", bad)
sys.exit(1 if bad else 0)
