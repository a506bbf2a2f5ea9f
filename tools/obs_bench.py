#!/usr/bin/env python3
"""HIP-event style timing of compute_observables / norm2 at 512^3 (host sync included, median of 20)."""
import json, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wafer_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
with wafer_amd.Context(wafer_amd.Params(n, n, n, dn=0.05, dt=5e-4, max_states=1)) as ctx:
    ctx.set_potential("Coulomb")
    ctx.set_initial_condition("Boolean")
    for name, fn in (("observables", ctx.observables), ("norm2", ctx.norm2)):
        fn(); ts = []
        for _ in range(20):
            ctx.synchronize(); t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
        print(json.dumps({"op": name, "ms_median_incl_host_sync": round(statistics.median(ts), 4), "ms_min": round(min(ts), 4),
                          "env": {k: v for k, v in os.environ.items() if k.startswith("WAFER_")}}))
