#!/usr/bin/env python3
"""Secondary measurement rows ("Five/Seven as secondary rows"; BASELINE configs #2, #4's grid on one GPU, the
fp32 path): for each configuration ONE bench.py line, the rocprofv3 --kernel-trace --stats summary of the same command
and the HBM-side traffic of its dominant kernel from two --pmc passes (FETCH_SIZE, WRITE_SIZE; corrected as
MI355X_MICROARCH.md section HBM prescribes: KiB -> bytes, FETCH_SIZE x 2 on gfx950).  Writes gpurun_out/<tag>/rows.jsonl and
one kernel-stats CSV per row; copy what is to be judged into profiles/.

    python3 tools/secondary_rows.py <outdir> [row ...]
"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROWS = {
    "threepoint_f64_512": [],
    "fivepoint_f64_512": ["--cd", "2"],
    "sevenpoint_f64_512": ["--cd", "3"],
    "threepoint_f32_storage_512": ["--dtype", "f32"],
    "threepoint_f32fast_512": ["--dtype", "f32fast"],   # fp32 storage and fp32 step arithmetic: config #5's throughput setting, on the three-step kernel
    "config2_harmonic_256": ["--grid", "256,256,256", "--potential", "Harmonic"],
    "threepoint_f64_1024": ["--grid", "1024,1024,1024", "--potential", "SimpleCornell", "--no-parity"],
    "threepoint_f64_384": ["--grid", "384,384,384"],   # 72 tiles per layer: the z-chunk policy's case (wafer_pick_zchunk)
}
COMMON = ["--no-cpu-baseline", "--no-excited"]
SHORT = ["--steps", "60", "--warmup", "6", "--preheat", "0", "--no-parity"]


def run(cmd, **kw):
    return subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, **kw)


def pmc(outdir, counter, args):
    d = os.path.join(outdir, counter.lower())
    run(["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--", "python3", "bench.py", *COMMON, *SHORT, *args])
    vals = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter:
                vals.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
        os.remove(f)
    return {k: sum(v) / len(v) for k, v in vals.items()}


def main():
    out = sys.argv[1]
    names = sys.argv[2:] or list(ROWS)
    os.makedirs(out, exist_ok=True)
    os.environ.setdefault("TMPDIR", "/tmp")
    for name in names:
        args = ROWS[name]
        rec = {"row": name, "command": "python3 bench.py " + " ".join(COMMON + args)}
        r = run(["python3", "bench.py", *COMMON, *args])
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            rec["error"] = f"bench.py rc {r.returncode}: {r.stderr[-400:]}"
            print(json.dumps(rec), flush=True)
            continue
        b = json.loads(lines[-1])
        rec["bench"] = {k: b[k] for k in ("metric", "value", "ms_per_step", "dtype", "steps", "warmup", "roofline", "parity", "config") if k in b}
        # the same command under rocprofv3 --kernel-trace --stats
        d = os.path.join(out, name + "_stats")
        r = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", "python3", "bench.py", *COMMON, *args])
        stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
        if stats:
            keep = os.path.join(out, f"kernel_stats_{name}.csv")
            os.replace(stats[0], keep)
            rows = list(csv.DictReader(open(keep)))
            rows.sort(key=lambda x: -float(x.get("TotalDurationNs", x.get("Total_Duration_Ns", 0)) or 0))
            top = rows[0]
            rec["dominant_kernel"] = {"name": top.get("Name", top.get("Kernel_Name", ""))[:120], "calls": int(top.get("Calls", 0)),
                                      "avg_us": float(top.get("AverageNs", top.get("Average_Ns", 0))) / 1e3,
                                      "stats_csv": os.path.basename(keep)}
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            os.remove(f)
        # HBM-side traffic of that kernel
        kname = rec.get("dominant_kernel", {}).get("name", "")
        fetch, write = pmc(out, "FETCH_SIZE", args), pmc(out, "WRITE_SIZE", args)
        for k in fetch:
            if kname and (kname[:60] in k or k[:60] in kname):
                rd, wr = 2.0 * fetch[k] * 1024.0, write.get(k, 0.0) * 1024.0
                rec["pmc"] = {"kernel": k[:120], "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                              "corrections": "KiB -> bytes x 1024; FETCH_SIZE x 2 (gfx950 wide coalesced reads); WRITE_SIZE exact"}
                if rec.get("dominant_kernel", {}).get("avg_us"):
                    rec["pmc"]["traffic_GBps_at_rocprof_avg"] = (rd + wr) / (rec["dominant_kernel"]["avg_us"] * 1e-6) / 1e9
                break
        with open(os.path.join(out, "rows.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")
        print(json.dumps({k: rec.get(k) for k in ("row", "dominant_kernel", "pmc")}) + "  ms/step " + str(rec["bench"]["ms_per_step"]), flush=True)


if __name__ == "__main__":
    main()
