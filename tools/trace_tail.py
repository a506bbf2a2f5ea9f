#!/usr/bin/env python3
"""Prints the last N kernels of a rocprofv3 kernel trace with start / end relative times and the gap
to the previous kernel:  python tools/trace_tail.py <kernel_trace.csv> [N]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    gap = "" if prev_end is None else f"gap {(s - prev_end) / 1e3:7.1f} us"
    print(f"{s / 1e3:10.1f} -> {e / 1e3:10.1f} us ({(e - s) / 1e3:8.1f} us) {gap:18s} {r['Kernel_Name'][:60]}")
    prev_end = e
