#!/usr/bin/env python3
"""Measured streaming ceilings of this GPU (flat 16 B/lane kernels over 512^3 fp64 buffers)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wafer_amd
par = wafer_amd.Params(512, 512, 512, dn=0.05, dt=5e-4)
with wafer_amd.Context(par) as ctx:
    ctx.set_potential("Coulomb"); ctx.set_initial_condition("Boolean")
    out = {f"{n}r1w_GBps": round(ctx.stream_bandwidth(n, 30), 1) for n in (1, 2, 3)}
    print(json.dumps(out))
