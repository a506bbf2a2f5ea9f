#!/usr/bin/env python3
"""Per-kernel averages of arbitrary rocprofv3 --pmc counters.
    python tools/pmc_counters.py <dir> [<dir> ...] [--match substr] > out.json
Every *counter_collection.csv under the directories is read; counter values of the dispatches of
one kernel are averaged (per launch)."""
import csv, glob, json, os, sys
from collections import defaultdict

dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
match = None
if "--match" in sys.argv:
    match = sys.argv[sys.argv.index("--match") + 1]
    dirs = [d for d in dirs if d != match]
acc = defaultdict(lambda: defaultdict(list))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if match and match not in k:
                continue
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k[:140]: {c: {"launches": len(v), "avg": sum(v) / len(v)} for c, v in cs.items()} for k, cs in acc.items()}
print(json.dumps(out, indent=1))
