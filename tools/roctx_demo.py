#!/usr/bin/env python3
"""A short solve-shaped run for `rocprofv3 --marker-trace --kernel-trace`: the engine's roctx ranges
(wafer_evolve_ground / wafer_evolve_excited / wafer_observables / wafer_halo_exchange) bracket the kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wafer_amd
with wafer_amd.Context(wafer_amd.Params(256, 256, 256, dn=0.05, dt=5e-4, max_states=2)) as ctx:
    ctx.set_potential("Coulomb")
    ctx.set_initial_condition("Boolean")
    for _ in range(2):
        obs = ctx.observables()
        ctx.normalise(obs["norm2"])
        ctx.evolve(0, 12)
    ctx.push_state()
    ctx.set_initial_condition("Constant")
    obs = ctx.observables()
    ctx.normalise(obs["norm2"])
    ctx.orthogonalise(1)
    ctx.evolve(1, 6)
    ctx.synchronize()
