#!/usr/bin/env python3
"""Bit-for-bit check of the hoisted-reciprocal division (wafer_div_invariant) against the device's IEEE
division over many random operands.   python tools/div_check.py [log2_operands_per_den=36]"""
import json
import sys
import time

sys.path.insert(0, ".")
import wafer_amd  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 36
dens = {"512^3 bench (2*0.05^2*1)": 2 * 0.05 ** 2, "1024^3 Cornell (2*0.02^2*2.35)": 2 * 0.02 ** 2 * 2.35,
        "FivePoint (24*0.05^2)": 24 * 0.05 ** 2, "SevenPoint (360*0.05^2)": 360 * 0.05 ** 2, "a norm (0.73105857863000487)": 0.73105857863000487}
out = {}
with wafer_amd.Context(wafer_amd.Params(8, 8, 8, dn=0.2, dt=0.004)) as ctx:
    for name, den in dens.items():
        t0, bad = time.time(), 0
        for chunk in range(1 << max(0, lg - 31)):
            bad += ctx.div_check(den, 1 << min(lg, 31), lo_exp=64, hi_exp=1983, seed=1000 + chunk)
        out[name] = {"operands": 1 << lg, "mismatches": bad, "seconds": round(time.time() - t0, 2)}
print(json.dumps(out, indent=1))
