#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc runs (FETCH_SIZE in one pass, WRITE_SIZE in another:
they do not fit one pass on gfx950, MI355X_MICROARCH.md 'rocprofv3 PMC slots')
into profiles/pmc_traffic.json: HBM-side bytes per launch of each kernel.

Corrections applied exactly as MI355X_MICROARCH.md section HBM prescribes:
  - both counters are in KiB -> x 1024;
  - on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced
    streaming read (16 B/lane) -> x 2 on the read side;
  - WRITE_SIZE reads exact for 16 B/lane streaming stores.
Infinity-Cache hits are counted by these L2-fabric-side counters, so `traffic`
is an UPPER bound on what HBM moved.

    python tools/pmc_summary.py <dir_with_fetch_run> <dir_with_write_run> [out.json]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def collect(root, counter):
    vals = defaultdict(list)
    files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {root}")
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == counter:
                    vals[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return vals


def main():
    fetch_dir, write_dir = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    fetch, write = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(fetch) | set(write)):
        f = fetch.get(name, [])
        w = write.get(name, [])
        f_avg = sum(f) / len(f) if f else 0.0
        w_avg = sum(w) / len(w) if w else 0.0
        kernels[name] = {
            "launches_fetch_pass": len(f), "launches_write_pass": len(w),
            "FETCH_SIZE_KiB_avg_raw": f_avg, "WRITE_SIZE_KiB_avg_raw": w_avg,
            "read_bytes_per_launch": 2.0 * f_avg * 1024.0,
            "write_bytes_per_launch": w_avg * 1024.0,
            "hbm_bytes_per_launch": 2.0 * f_avg * 1024.0 + w_avg * 1024.0,
        }
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from wafer_amd.provenance import kernel_sources_sha16
    doc = {
        # the kernel sources these bytes were measured on: bench.py labels roofline.traffic stale when they have changed since
        "kernel_sources_sha16": kernel_sources_sha16(),
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py at N=1",
        "corrections": "KiB->bytes x1024; FETCH_SIZE x2 (gfx950 wide coalesced reads); WRITE_SIZE exact; "
                       "Infinity-Cache hits are included, so this bounds HBM traffic from above",
        "kernels": kernels,
    }
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1)
    for k, v in kernels.items():
        if v["hbm_bytes_per_launch"] > 1e6:
            print(f"{k[:70]:70s} read {v['read_bytes_per_launch']/1e9:8.3f} GB  write {v['write_bytes_per_launch']/1e9:8.3f} GB")


if __name__ == "__main__":
    main()
