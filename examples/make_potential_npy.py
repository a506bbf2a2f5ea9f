#!/usr/bin/env python3
"""Writes a user potential as the framed .npy that wafer_amd.run memory-maps (and wafer-hip --convert
produces from the reference's formats): float64, C order, shape (n + 2e,)*3 with e zero cells around
the work area (e = 1 / 2 / 3 for Three / Five / SevenPoint).  Plane by plane, so the array never has
to fit in memory.

    python examples/make_potential_npy.py N OUT.npy [--ext 1] [--dn 0.01]

The potential here is the symmetric Poschl-Teller well of the reference's gen_potential.py example,
V = -(l(l+1)/2) (sech^2 x + sech^2 y + sech^2 z), l = 6, on the reference's cell-centred coordinates.
"""
import argparse

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("n", type=int)
    ap.add_argument("out")
    ap.add_argument("--ext", type=int, default=1)
    ap.add_argument("--dn", type=float, default=0.01)
    ap.add_argument("--lam", type=float, default=6.0)
    a = ap.parse_args()
    n, e = a.n, a.ext
    extent = (a.dn * n - a.dn) / 2.0
    s = np.linspace(-extent, extent, n)
    well = -(a.lam * (a.lam + 1.0)) / 2.0 / np.cosh(s) ** 2          # one axis' term
    out = np.lib.format.open_memmap(a.out, mode="w+", dtype=np.float64, shape=(n + 2 * e,) * 3)
    yz = well[:, None] + well[None, :]
    for i in range(n):                                                # x is the slowest axis
        out[i + e, e:-e, e:-e] = well[i] + yz
    out.flush()


if __name__ == "__main__":
    main()
