#!/usr/bin/env python3
"""VGPRs / scratch / LDS / occupancy of every kernel of the engine, from hipcc's own remarks.

    cd wafer_amd/csrc && hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 --cuda-device-only -c \
        wafer_engine.hip -o /tmp/eng.o -Rpass-analysis=kernel-resource-usage > /tmp/resusage.txt 2>&1
    python tools/resource_usage.py /tmp/resusage.txt [substring ...]
"""
import re
import subprocess
import sys


def main():
    txt = open(sys.argv[1]).read()
    want = sys.argv[2:]
    blocks = re.split(r"remark: Function Name: ", txt)[1:]
    names = [b.split(" ")[0] for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    for b, dn in zip(blocks, dem):
        if want and not all(w in dn for w in want):
            continue

        def g(k):
            m = re.search(k + r": (\d+)", b)
            return m.group(1) if m else "?"
        scratch, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
        print(f"vgpr {g('VGPRs'):>3} agpr {g('AGPRs'):>3} scratch {scratch:>4} occ {occ} lds {lds:>6}  {dn[:150]}")


if __name__ == "__main__":
    main()
