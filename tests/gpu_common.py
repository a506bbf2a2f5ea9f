"""Shared helpers of the -m gpu parity tests: build the same problem in the CPU
oracle and in the HIP engine (through the C ABI) from one description."""
import numpy as np

from oracle import wafer_oracle as wo
import wafer_amd


def make_pair(shape, ext=1, potential="Harmonic", dn=0.2, dt=0.004, mass=1.0, sig=1.0, dtype="f64",
              max_states=4, **kw):
    cfg = wo.Config(*shape, ext=ext, potential=potential, dn=dn, dt=dt, mass=mass, sig=sig)
    par = wafer_amd.Params(*shape, dn=dn, dt=dt, mass=mass, sig=sig, central_difference=ext,
                           dtype=dtype, max_states=max_states, **kw)
    return cfg, par


def random_phi(cfg, seed=0):
    """work area ~ N(0,1), zero Dirichlet frame"""
    e = cfg.ext
    rng = np.random.default_rng(seed)
    phi = np.zeros(cfg.padded_shape)
    phi[e:-e, e:-e, e:-e] = rng.standard_normal(cfg.work_shape)
    return phi


def ulp_diff(a, b):
    """max distance in units of the last place between two float64 arrays"""
    a = np.ascontiguousarray(a, dtype=np.float64).view(np.int64)
    b = np.ascontiguousarray(b, dtype=np.float64).view(np.int64)
    a = np.where(a < 0, np.int64(-(2 ** 63)) - a, a)
    b = np.where(b < 0, np.int64(-(2 ** 63)) - b, b)
    return int(np.max(np.abs(a - b))) if a.size else 0
