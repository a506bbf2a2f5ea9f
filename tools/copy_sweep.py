#!/usr/bin/env python3
"""The device's copy ceiling (wafer_diag_copy_bw) over vectors in flight per lane and workgroups per
CU: prints one JSON line per setting."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wafer_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
with wafer_amd.Context(wafer_amd.Params(n, n, n, dn=0.05, dt=5e-4, max_states=1)) as ctx:
    ctx.set_potential("Coulomb")
    ctx.set_initial_condition("Boolean")
    for unroll in (1, 2, 4, 8):
        for bpc in (1, 2, 4, 8, 16, 32):
            print(json.dumps({"kernel": "wafer_k_copy16", "unroll": unroll, "blocks_per_cu": bpc,
                              "GBps": round(ctx.copy_bandwidth(100, unroll, bpc), 1)}), flush=True)
