#!/usr/bin/env python3
"""Where a plane iteration of the three-step kernel spends its cycles, per wave (a -DWAFER_DIAG=1 build:
bash tools/build_alt.sh stamp "-DWAFER_DIAG=1"; WAFER_HIP_LIB=.../alt_stamp/libwafer_hip.so python3 tools/f3_stamps.py).
Shares, not times: the stamps' own waits forbid overlaps the real kernel has."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wafer_amd  # noqa: E402
from wafer_amd import engine  # noqa: E402

SEG = ["issue+stage", "level1 main", "level1 extra", "level2", "level3+stores", "barrier", "wait+rotate", "-"]


def main():
    n = int(os.environ.get("N", "512"))
    par = wafer_amd.Params(n, n, n, dn=0.05, dt=5e-4, mass=1.0, max_states=1)
    with wafer_amd.Context(par) as ctx:
        ctx.set_stencil_variant(3)
        ctx.set_potential("Coulomb")
        ctx.set_initial_condition("Boolean")
        ctx.evolve(0, 30)
        lib = engine.load_library()
        buf = (C.c_ulonglong * 64)()
        rc = lib.wafer_debug_f3_stamps(buf)
        assert rc == 0, rc
    rows = [[buf[w * 8 + k] for k in range(8)] for w in range(8)]
    out = {"grid": n, "segments": SEG[:7], "per_wave_cycles": rows}
    for w, r in enumerate(rows):
        tot = sum(r) or 1
        print("wave %d  total %9d  " % (w, tot) + "  ".join("%s %4.1f%%" % (SEG[k], 100.0 * r[k] / tot) for k in range(7)))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
