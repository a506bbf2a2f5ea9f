#!/usr/bin/env python3
"""Randomised parity sweep of the HIP path against the oracle (a development aid next to the
fixed cases of tests/test_gpu_parity.py): random shapes incl. 1-cell axes, stencil orders,
potentials, step counts and kernel variants; ground state bit for bit, excited states (random
stored states, Gram-Schmidt every step) to 1e-10; on fp32 storage (one and two steps per pass) to the fp32-storage bar.   N=200 SEED=3 python tests/fuzz_parity.py   (it lives under tests/ because it drives the oracle)"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import wafer_amd as wa
from oracle import wafer_oracle as wo
from gpu_common import make_pair, random_phi
wo.set_threads(8)
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
bad = 0
for it in range(int(os.environ.get("N", "60"))):
    ext = int(rng.integers(1, 4))
    shape = tuple(int(x) for x in rng.choice([1, 2, 3, 5, 8, 15, 16, 17, 31, 33, 64, 65, 127, 129, 130, 200, 257], size=3))
    if rng.random() < 0.5:
        shape = (shape[0], int(rng.integers(1, 40)), int(rng.integers(1, 40)))
    whole = rng.random() < 0.3   # grids made of whole 128 x 16 tiles: the three-step kernel's exact-store / ring / single-direction code
    if whole:
        shape = (int(rng.choice([128, 256])), int(rng.choice([16, 32, 48])), int(rng.integers(1, 40)))
        ext = 1
    pot = str(rng.choice(["Harmonic", "Coulomb", "SimpleCornell", "NoPotential", "Cube"]))
    steps = int(rng.integers(1, 13 if whole else 9))
    variant = 3 if whole else int(rng.choice([-1, 0, 1, 2, 3]))
    os.environ["WAFER_FUSE3_MIN_NY"] = str(rng.choice([1, 16]))   # the three-step kernel also on grids thinner than its tile
    for name, choices in (("WAFER_ZCHUNK", ["0", "1", "2", "3", "5", "7"]), ("WAFER_F3_PLAIN_DOWN", ["0", "1"]), ("WAFER_F3_XS", ["1", "1", "0"]),
                          ("WAFER_F3_SCHED", ["0", "0", "1"])):
        os.environ[name] = str(rng.choice(choices)) if whole else "0" if name != "WAFER_F3_XS" else "1"
    dtype = "f64"
    os.environ["WAFER_F2_WIDE"] = str(rng.choice([1, 1, 0]))   # FivePoint two steps per pass: the 128 x 16-tile kernel / the one with helper waves
    if ext == 2 and not whole:
        os.environ["WAFER_ZCHUNK"] = str(rng.choice([0, 1, 2, 3, 5]))
    try:
        cfg, par = make_pair(shape, ext=ext, potential=pot, dn=0.2, dt=0.004, mass=1.3, sig=0.3, dtype=dtype)
        v = wo.potential_generate(cfg); a, b = wo.ab(cfg, v)
        phi = random_phi(cfg, seed=it)
        want = phi.copy(); wo.evolve(cfg, 0, a, b, want, [], steps)
        with wa.Context(par) as ctx:
            ctx.set_stencil_variant(variant)
            ctx.set_potential(pot); ctx.upload_phi(phi); ctx.evolve(0, steps)
            got = ctx.download_phi()
            ok = np.array_equal(got, want, equal_nan=True)
            o1, o2 = ctx.observables(), wo.observables(cfg, v, want, wo.potential_sub(cfg))
            ok2 = all(abs(o1[k]-o2[k]) <= 1e-11*max(1.0, abs(o2[k])) or (np.isnan(o1[k]) and np.isnan(o2[k])) for k in o2)
        if not (ok and ok2):
            bad += 1
            print("MISMATCH", shape, ext, pot, steps, variant, ok, ok2, flush=True)
    except Exception as e:
        bad += 1
        print("ERROR", shape, ext, pot, steps, variant, repr(e)[:200], flush=True)
# fp32 storage (fp64 or fp32 step arithmetic): no oracle to the bit, but every kernel family must give the single-step kernel's bits --
# the three-step kernel (float in HBM, double in the CU), the FivePoint two-step kernel on 128 x 16 tiles, the two-step kernel
for it in range(int(os.environ.get("N", "60")) // 3):
    ext = int(rng.integers(1, 3))
    shape = (int(rng.choice([64, 128, 130, 200, 256, 257, 300, 512])), int(rng.choice([1, 8, 16, 17, 32, 37, 48])), int(rng.integers(1, 30)))
    dtype = str(rng.choice(["f32", "f32fast"]))
    steps = int(rng.integers(1, 12))
    pot = str(rng.choice(["Harmonic", "Coulomb", "SimpleCornell", "Cube"]))
    os.environ["WAFER_FUSE3_MIN_NY"] = "1"
    os.environ["WAFER_F2_WIDE"] = "1"
    got = {}
    try:
        for variant in (3 if ext == 1 else 2, 1):
            os.environ["WAFER_ZCHUNK"] = str(rng.choice([0, 1, 2, 3, 5])) if variant != 1 else "0"
            os.environ["WAFER_F3_PLAIN_DOWN"] = str(rng.choice([0, 1])) if variant == 3 else "0"
            with wa.Context(wa.Params(*shape, dn=0.2, dt=0.004, mass=1.3, sig=0.3, central_difference=ext, dtype=dtype)) as ctx:
                ctx.set_stencil_variant(variant)
                ctx.set_potential(pot); ctx.set_initial_condition("Gaussian", seed=it + 1); ctx.evolve(0, steps)
                got[variant] = ctx.download_phi()
        a_, b_ = got.values()
        if not np.array_equal(a_, b_, equal_nan=True):
            bad += 1
            print("MISMATCH fp32 storage", shape, ext, dtype, pot, steps, float(np.max(np.abs(a_ - b_))), flush=True)
    except Exception as e:
        bad += 1
        print("ERROR fp32 storage", shape, ext, dtype, pot, steps, repr(e)[:200], flush=True)
# excited states: normalise + modified Gram-Schmidt after every step (grid.rs:674-681)
for it in range(int(os.environ.get("N", "60")) // 2):
    ext = int(rng.integers(1, 4))
    shape = tuple(int(x) for x in rng.choice([2, 3, 5, 8, 15, 16, 17, 31, 33, 64, 65, 129], size=3))
    wnum = int(rng.integers(1, 6))
    steps = int(rng.integers(1, 5))
    two = rng.random() < 0.4     # the two-steps-per-pass kernels (four steps or more, up to three stored states), whole tiles or not
    if two:
        shape = (int(rng.choice([128, 256, 130, 64])), int(rng.choice([8, 16, 24, 32, 17])), int(rng.integers(2, 30)))
        ext, wnum, steps = 1, int(rng.integers(1, 4)), int(rng.integers(4, 10))
    os.environ["WAFER_F3_PLAIN_DOWN"] = "0"
    os.environ["WAFER_X2_MAX_K"] = "3"
    os.environ["WAFER_X2_RY"] = str(rng.choice([0, 1, 2]))
    os.environ["WAFER_ZCHUNK"] = str(rng.choice([0, 1, 2, 3, 5])) if two else "0"
    os.environ["WAFER_F3_XS"] = str(rng.choice([1, 1, 0]))
    pot = str(rng.choice(["Harmonic", "Coulomb", "SimpleCornell", "NoPotential"]))
    one_pass = 1 if two else int(rng.integers(0, 2))
    os.environ["WAFER_ONE_PASS"] = str(one_pass)
    os.environ["WAFER_VGEN"] = str(rng.integers(0, 2))      # closed-form V in the kernel / the stored array
    os.environ["WAFER_XF_DEEP"] = str(rng.integers(0, 2))   # staging pipeline / plain prefetch
    try:
        cfg, par = make_pair(shape, ext=ext, potential=pot, dn=0.2, dt=0.004, mass=1.3, sig=0.3, max_states=wnum)
        v = wo.potential_generate(cfg); a, b = wo.ab(cfg, v)
        lowers = []
        for j in range(wnum):   # orthonormalised random states, as a converged w_store would be
            l = random_phi(cfg, seed=1000 + 10 * it + j)
            wo.normalise(l, wo.norm2(cfg, l)); wo.orthogonalise(j, l, lowers); wo.normalise(l, wo.norm2(cfg, l))
            lowers.append(l)
        phi = random_phi(cfg, seed=it)
        want = phi.copy(); wo.evolve(cfg, wnum, a, b, want, lowers, steps)
        with wa.Context(par) as ctx:
            ctx.set_potential(pot)
            for l in lowers:
                ctx.upload_phi(l); ctx.push_state()
            ctx.upload_phi(phi); ctx.evolve(wnum, steps)
            got = ctx.download_phi()
        scale = max(1e-300, float(np.max(np.abs(want))))
        if not np.allclose(got, want, rtol=0, atol=1e-10 * scale):
            bad += 1
            print("MISMATCH excited", shape, ext, pot, wnum, steps, one_pass, float(np.max(np.abs(got - want))) / scale, flush=True)
    except Exception as e:
        bad += 1
        print("ERROR excited", shape, ext, pot, wnum, steps, one_pass, repr(e)[:200], flush=True)
# excited states on fp32 STORAGE (round 6: the two-steps-per-pass kernels on a storage tag): no reference to the bit -- against the fp64 oracle
# (5e-6 of the state's largest value) and against the one-step fp32-storage kernels (2e-6), random tiles / z-chunks / tile heights / store counts
for it in range(int(os.environ.get("N", "60")) // 4):
    shape = (int(rng.choice([128, 256, 130, 64, 200])), int(rng.choice([8, 16, 24, 32, 17, 5])), int(rng.integers(2, 30)))
    wnum, steps = int(rng.integers(1, 4)), int(rng.integers(4, 13))
    dtype = str(rng.choice(["f32", "f32fast"]))
    os.environ["WAFER_X2_MAX_K"] = "3"
    os.environ["WAFER_X2_RY"] = str(rng.choice([0, 1, 2]))
    os.environ["WAFER_ZCHUNK"] = str(rng.choice([0, 1, 2, 3, 5]))
    os.environ["WAFER_F3_XS"] = str(rng.choice([1, 1, 0]))
    os.environ["WAFER_ONE_PASS"] = "1"
    pot = str(rng.choice(["Harmonic", "Coulomb", "SimpleCornell", "Cube"]))
    try:
        cfg, par = make_pair(shape, ext=1, potential=pot, dn=0.2, dt=0.004, mass=1.3, sig=0.3, max_states=wnum, dtype=dtype)
        r32 = lambda x: np.ascontiguousarray(x.astype(np.float32).astype(np.float64))
        v = r32(wo.potential_generate(cfg)); a, b = wo.ab(cfg, v)
        lowers = []
        for j in range(wnum):
            l = random_phi(cfg, seed=2000 + 10 * it + j)
            wo.normalise(l, wo.norm2(cfg, l)); wo.orthogonalise(j, l, lowers); wo.normalise(l, wo.norm2(cfg, l))
            lowers.append(r32(l))
        phi = r32(random_phi(cfg, seed=it))
        want = phi.copy(); wo.evolve(cfg, wnum, a, b, want, lowers, steps)
        got = {}
        for x2 in ("1", "0"):
            os.environ["WAFER_X2"] = x2
            with wa.Context(par) as ctx:
                ctx.set_potential(pot)
                for j, l in enumerate(lowers):
                    ctx.load_state(j, l)
                ctx.upload_phi(phi); ctx.evolve(wnum, steps)
                got[x2] = (ctx.download_phi(), ctx.x2_passes())
        scale = max(1e-300, float(np.max(np.abs(want))))
        e1, e0, e10 = (float(np.max(np.abs(got["1"][0] - want))) / scale, float(np.max(np.abs(got["0"][0] - want))) / scale,
                       float(np.max(np.abs(got["1"][0] - got["0"][0]))) / scale)
        if not (e1 <= 5e-6 and e0 <= 5e-6 and e10 <= 2e-6 and got["1"][1] > 0 and got["0"][1] == 0):
            bad += 1
            print("MISMATCH excited fp32 storage", shape, dtype, pot, wnum, steps, e1, e0, e10, got["1"][1], got["0"][1], flush=True)
    except Exception as e:
        bad += 1
        print("ERROR excited fp32 storage", shape, dtype, pot, wnum, steps, repr(e)[:200], flush=True)
os.environ.pop("WAFER_X2", None)
os.environ.pop("WAFER_ONE_PASS", None)
os.environ.pop("WAFER_VGEN", None)
os.environ.pop("WAFER_XF_DEEP", None)
os.environ.pop("WAFER_FUSE3_MIN_NY", None)
for name in ("WAFER_ZCHUNK", "WAFER_F3_PLAIN_DOWN", "WAFER_F3_XS", "WAFER_F3_SCHED", "WAFER_X2_MAX_K", "WAFER_X2_RY", "WAFER_F2_WIDE"):
    os.environ.pop(name, None)
print("fuzz done, bad =", bad)
sys.exit(1 if bad else 0)
