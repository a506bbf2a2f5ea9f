#!/usr/bin/env python3
"""Writes tests/golden/oracle_fixtures.npz: small inputs and the oracle's outputs for every
function of the hot path (SURVEY.md 8c "fixtures to commit").  The fixtures freeze the oracle
(tests/test_oracle_fixtures.py re-derives them bit for bit on CPU) and let the GPU parity tests
compare the HIP path with committed numbers instead of a freshly built checker
(tests/test_gpu_fixtures.py).  Inputs come from numpy's PCG64 with the seeds below; run

    python tests/golden/make_oracle_fixtures.py

after a deliberate change of the oracle only.  Each array is <= 256 kB."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import wafer_oracle as wo  # noqa: E402

CASES = [  # name, shape, ext, potential, dn, dt, mass, sig
    ("harmonic_3pt", (14, 11, 17), 1, "Harmonic", 0.3, 0.012, 1.0, 1.0),
    ("coulomb_5pt", (12, 13, 10), 2, "Coulomb", 0.25, 0.004, 1.3, 1.0),
    ("cornell_7pt", (10, 12, 11), 3, "SimpleCornell", 0.2, 0.002, 2.35, 0.223),
    ("fullcornell_3pt", (9, 10, 12), 1, "FullCornell", 0.2, 0.004, 1.4, 0.223),
]


def random_phi(cfg, seed):
    e = cfg.ext
    phi = np.zeros(cfg.padded_shape)
    phi[e:-e, e:-e, e:-e] = np.random.default_rng(seed).standard_normal(cfg.work_shape)
    return phi


def build():
    wo.set_threads(2)
    out = {}
    for idx, (name, shape, ext, pot, dn, dt, mass, sig) in enumerate(CASES):
        cfg = wo.Config(*shape, ext=ext, potential=pot, dn=dn, dt=dt, mass=mass, sig=sig)
        v = wo.potential_generate(cfg)
        a, b = wo.ab(cfg, v)
        kind, scalar, arr = wo.potential_sub(cfg)
        phi0 = random_phi(cfg, 100 + idx)
        lowers = []
        for j in range(2):   # an orthonormal store, as converged lower states are
            l = random_phi(cfg, 200 + 10 * idx + j)
            wo.normalise(l, wo.norm2(cfg, l))
            wo.orthogonalise(j, l, lowers)
            wo.normalise(l, wo.norm2(cfg, l))
            lowers.append(l)
        obs0 = wo.observables(cfg, v, phi0, (kind, scalar, arr))
        ground = phi0.copy()
        wo.evolve(cfg, 0, a, b, ground, [], 5)
        excited = phi0.copy()
        wo.evolve(cfg, 2, a, b, excited, lowers, 3)
        gs = phi0.copy()
        wo.normalise(gs, wo.norm2(cfg, gs))
        wo.orthogonalise(2, gs, lowers)
        out[f"{name}/params"] = np.array([*shape, ext, dn, dt, mass, sig], dtype=np.float64)
        out[f"{name}/v"], out[f"{name}/a"], out[f"{name}/b"] = v, a, b
        out[f"{name}/potsub_kind_scalar"] = np.array([kind, scalar])
        if arr is not None:
            out[f"{name}/potsub"] = arr
        out[f"{name}/phi0"] = phi0
        out[f"{name}/lower0"], out[f"{name}/lower1"] = lowers
        out[f"{name}/observables0"] = np.array([obs0["energy"], obs0["norm2"], obs0["v_infinity"], obs0["r2"]])
        out[f"{name}/norm2_0"] = np.array([wo.norm2(cfg, phi0)])
        out[f"{name}/ground_5steps"] = ground
        out[f"{name}/excited_wnum2_3steps"] = excited
        out[f"{name}/normalised_orthogonalised"] = gs
        for ic in ("Boolean", "Constant"):
            out[f"{name}/ic_{ic}"] = wo.initial_condition(cfg, ic)
    return out


def main():
    out = build()
    assert all(v.nbytes <= 256 * 1024 for v in out.values())
    np.savez_compressed(os.path.join(HERE, "oracle_fixtures.npz"), **out)
    print(len(out), "arrays,", sum(v.nbytes for v in out.values()) // 1024, "KiB uncompressed")


if __name__ == "__main__":
    main()
