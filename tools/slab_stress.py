#!/usr/bin/env python3
"""Repeats one decomposed ground-state case (ranks as contexts of this process, tests/test_gpu_slab.py's harness) many times and
reports which planes differ when a repetition does -- for chasing rare races of the overlap modes.
    python tools/slab_stress.py MODE REPS [world nx ny nz steps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("WAFER_PEER_SAME_DEVICE", "1")
os.environ.setdefault("WAFER_FUSE3_MIN_NY", "1")
import wafer_amd as wa  # noqa: E402
from tests.test_gpu_slab import run_slabs, assemble  # noqa: E402
from wafer_amd.slab import partition  # noqa: E402

mode, reps = int(sys.argv[1]), int(sys.argv[2])
world, nx, ny, nz, steps = (int(a) for a in sys.argv[3:8]) if len(sys.argv) >= 8 else (3, 140, 17, 37, 7)
shape = (nx, ny, nz)
calls = (steps, 5, 9)
with wa.Context(wa.Params(*shape, dn=0.2, dt=0.004, mass=1.0, central_difference=1)) as ctx:
    ctx.set_potential("Coulomb")
    ctx.set_initial_condition("Boolean")
    for n in calls:
        ctx.evolve(0, n)
    want = ctx.download_phi()
base = wa.Params(*shape, dn=0.2, dt=0.004, mass=1.0, central_difference=1, halo_depth=3)
bad = 0
for rep in range(reps):
    def body(ctx, rank):
        ctx.set_overlap(mode)
        ctx.set_potential("Coulomb")
        ctx.set_initial_condition("Boolean")
        for n in calls:
            ctx.evolve(0, n)
        return ctx.download_phi()
    try:
        res, fabric = run_slabs(wa, base, world, body, connect=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(f"rep {rep}: {e!r}"[:300], flush=True)
        continue
    got = assemble(base, world, res)
    if not np.array_equal(got, want):
        bad += 1
        planes = sorted(set(np.argwhere(got != want)[:, 2].tolist()))
        owners = [partition(nz, world, r) for r in range(world)]
        print(f"rep {rep}: {int(np.sum(got != want))} cells differ on padded planes {planes}; slabs (z_begin, z_count) {owners}", flush=True)
print(f"slab stress: mode {mode}, {reps} repetitions of {world} x {shape}, bad = {bad}")
sys.exit(1 if bad else 0)
