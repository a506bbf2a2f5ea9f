#!/usr/bin/env python3
"""Writes BASELINE config #5's user potential as the reference's own input file: a 64^3 array in ./input/potential.csv
(`i,j,k,data` rows over the unpadded array, output.rs:148-165), which `wafer-hip` (like Wafer: input.rs:149-176) reads and
trilinearly resamples to the configured grid -- ON THE DEVICE here (input.rs:667-716 -> wafer_k_trilerp).

    python examples/make_potential_64.py ./input/potential.csv [--n 64]
    wafer_amd/wafer-hip -c examples/fromfile_2048_f32.yaml

The potential is the anisotropic Poschl-Teller well the GPU tests use (tests/test_gpu_fullsize.py::file_source,
tests/test_gpu_configs.py: the reference's gen_potential.py:45-60 in spirit)."""
import argparse
import os

import numpy as np


def source(n_src=64):
    ax = (np.arange(n_src) - (n_src - 1) / 2) * (12.8 / n_src)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    return -3.0 / np.cosh(0.6 * np.sqrt(X * X + Y * Y + 2.0 * Z * Z)) ** 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--n", type=int, default=64)
    a = ap.parse_args()
    src = source(a.n)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    I, J, K = np.meshgrid(*[np.arange(a.n)] * 3, indexing="ij")
    np.savetxt(a.out, np.column_stack([I.ravel(), J.ravel(), K.ravel(), src.ravel()]), fmt=["%d", "%d", "%d", "%.17g"], delimiter=",")


if __name__ == "__main__":
    main()
