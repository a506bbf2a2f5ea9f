"""ctypes loader for the CPU oracle (oracle/libwafer_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under wafer_amd/ may import this module.

Arrays are numpy float64, C-order [x][y][z] with shape (nx+2e, ny+2e, nz+2e),
exactly the reference's ndarray layout (config.rs:224-238, grid.rs:505-534).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libwafer_oracle.so")

POTENTIALS = [
    "NoPotential", "Cube", "QuadWell", "Periodic", "Coulomb", "ComplexCoulomb",
    "ElipticalCoulomb", "SimpleCornell", "FullCornell", "Harmonic", "ComplexHarmonic",
    "Dodecahedron", "FromFile", "FromScript",
]
INITIAL_CONDITIONS = ["FromFile", "Gaussian", "Coulomb", "Constant", "Boolean"]


class _Config(C.Structure):
    _fields_ = [
        ("nx", C.c_int64), ("ny", C.c_int64), ("nz", C.c_int64),
        ("ext", C.c_int32), ("potential", C.c_int32),
        ("dn", C.c_double), ("dt", C.c_double), ("mass", C.c_double), ("sig", C.c_double),
    ]


class _Obs(C.Structure):
    _fields_ = [("energy", C.c_double), ("norm2", C.c_double),
                ("v_infinity", C.c_double), ("r2", C.c_double)]


class _Record(C.Structure):
    _fields_ = [("step", C.c_uint64), ("tau", C.c_double), ("energy", C.c_double),
                ("norm2", C.c_double), ("v_infinity", C.c_double), ("r2", C.c_double),
                ("diff", C.c_double)]


def build(force: bool = False) -> str:
    """Compile the oracle with its Makefile (gcc); returns the .so path."""
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("wafer_oracle.c", "wafer_oracle.h", "Makefile")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libwafer_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None
_NATIVE_PATH = os.path.join(_HERE, "libwafer_oracle_native.so")
_native = False


def use_native() -> bool:
    """The timed baseline leg of bench.py: rebuild the same source -O3 -march=native -ffp-contract=off
    ON THIS HOST (make native) and route every later call through it.  Contraction stays off, so the bits are the portable
    build's.  Returns False (and changes nothing) where the build fails."""
    global _lib, _native
    if _native:
        return True
    try:
        subprocess.check_call(["make", "-C", _HERE, "-B", "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        L = _bind(C.CDLL(_NATIVE_PATH))
    except Exception:  # noqa: BLE001 -- no compiler, or a host gcc does not know: the portable build serves
        return False
    _lib, _native = L, True
    return True


def is_native() -> bool:
    return _native


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = _bind(C.CDLL(_LIB_PATH))
    return _lib


def _bind(L):
    if True:
        dp = C.POINTER(C.c_double)
        cp = C.POINTER(_Config)
        L.wo_padded_len.restype = C.c_size_t
        L.wo_padded_len.argtypes = [cp]
        L.wo_calculate_r2.restype = C.c_double
        L.wo_calculate_r2.argtypes = [C.c_int64] * 6
        L.wo_alphas.restype = C.c_double
        L.wo_alphas.argtypes = [C.c_double]
        L.wo_mu.restype = C.c_double
        L.wo_mu.argtypes = [C.c_double]
        L.wo_potential_generate.argtypes = [cp, dp]
        L.wo_ab.argtypes = [cp, dp, dp, dp]
        L.wo_ab.restype = None
        L.wo_potential_sub.argtypes = [cp, C.POINTER(C.c_int), dp, dp]
        L.wo_initial_condition.argtypes = [cp, C.c_int, C.c_uint64, dp]
        L.wo_norm2.restype = C.c_double
        L.wo_norm2.argtypes = [cp, dp]
        L.wo_normalise.restype = None
        L.wo_normalise.argtypes = [dp, C.c_size_t, C.c_double]
        L.wo_orthogonalise.restype = None
        L.wo_orthogonalise.argtypes = [C.c_int, dp, C.POINTER(dp), C.c_size_t]
        L.wo_symmetrise.restype = C.c_int
        L.wo_symmetrise.argtypes = [cp, C.c_int, dp]
        L.wo_observables.restype = None
        L.wo_observables.argtypes = [cp, dp, C.c_int, C.c_double, dp, dp, C.POINTER(_Obs)]
        L.wo_evolve.restype = None
        L.wo_evolve.argtypes = [cp, C.c_int, dp, dp, dp, C.POINTER(dp), C.c_uint64]
        L.wo_stencil_step.restype = None
        L.wo_stencil_step.argtypes = [cp, dp, dp, dp, dp]
        L.wo_solve.restype = C.c_size_t
        L.wo_solve.argtypes = [cp, C.c_int, dp, dp, dp, C.c_int, C.c_double, dp, dp,
                               C.POINTER(dp), C.c_double, C.c_uint64, C.c_int, C.c_uint64,
                               C.POINTER(_Record), C.c_size_t, C.POINTER(C.c_int)]
        L.wo_trilerp_resize.restype = C.c_int
        L.wo_trilerp_resize.argtypes = [dp, C.c_int64, C.c_int64, C.c_int64, dp,
                                        C.c_int64, C.c_int64, C.c_int64]
        L.wo_trilerp_resize_basis.restype = C.c_int
        L.wo_trilerp_resize_basis.argtypes = [dp] + [C.c_int64] * 3 + [dp] + [C.c_int64] * 6
        L.wo_potential_generate_zwindow.argtypes = [cp, C.c_int64, C.c_int64, dp]
        L.wo_initial_condition_zwindow.argtypes = [cp, C.c_int, C.c_uint64, C.c_int64, C.c_int64, dp]
        L.wo_ab_n.restype = None
        L.wo_ab_n.argtypes = [C.c_double, dp, dp, dp, C.c_size_t]
        L.wo_trilerp_resize_basis_zwindow.restype = C.c_int
        L.wo_trilerp_resize_basis_zwindow.argtypes = [dp] + [C.c_int64] * 3 + [dp] + [C.c_int64] * 7
        L.wo_set_threads.argtypes = [C.c_int]
        L.wo_get_threads.restype = C.c_int
        L.wo_host_copy_gbps.restype = C.c_double
        L.wo_host_copy_gbps.argtypes = [C.c_size_t, C.c_int]
        L.wo_thread_placement.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]
    return L


def _dp(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _store_ptrs(w_store):
    dp = C.POINTER(C.c_double)
    arr = (dp * max(1, len(w_store)))()
    for i, w in enumerate(w_store):
        arr[i] = _dp(w)
    return arr


@dataclass
class Config:
    """The subset of config.rs:292-333 the hot path reads."""
    nx: int
    ny: int
    nz: int
    ext: int = 1              # 1/2/3 = Three/Five/SevenPoint
    potential: str = "Harmonic"
    dn: float = 0.1
    dt: float = 1e-3
    mass: float = 1.0
    sig: float = 1.0

    def c(self) -> _Config:
        return _Config(self.nx, self.ny, self.nz, self.ext, POTENTIALS.index(self.potential),
                       self.dn, self.dt, self.mass, self.sig)

    @property
    def padded_shape(self):
        e = self.ext
        return (self.nx + 2 * e, self.ny + 2 * e, self.nz + 2 * e)

    @property
    def work_shape(self):
        return (self.nx, self.ny, self.nz)


def set_threads(n: int) -> None:
    lib().wo_set_threads(int(n))


def get_threads() -> int:
    return lib().wo_get_threads()


def host_copy_gbps(nbytes: int, iters: int = 5) -> float:
    """read + written GB/s of an OpenMP copy between two first-touched buffers of `nbytes` on the current threads"""
    return float(lib().wo_host_copy_gbps(int(nbytes), int(iters)))


def thread_placement() -> dict:
    """the OpenMP runtime's binding policy and place count (what OMP_PROC_BIND / OMP_PLACES resolved to)"""
    pb, npl = C.c_int(0), C.c_int(0)
    lib().wo_thread_placement(C.byref(pb), C.byref(npl))
    names = {0: "false", 1: "true", 2: "master", 3: "close", 4: "spread"}
    return {"proc_bind": names.get(pb.value, str(pb.value)), "places": npl.value}


def calculate_r2(idx, size) -> float:
    return lib().wo_calculate_r2(idx[0], idx[1], idx[2], size[0], size[1], size[2])


def alphas(mu: float) -> float:
    return lib().wo_alphas(mu)


def mu(t: float) -> float:
    return lib().wo_mu(t)


def potential_generate(cfg: Config) -> np.ndarray:
    v = np.zeros(cfg.padded_shape)
    c = cfg.c()
    if lib().wo_potential_generate(C.byref(c), _dp(v)):
        raise ValueError("PotentialNotAvailable")
    return v


def ab(cfg: Config, v: np.ndarray):
    a = np.empty_like(v)
    b = np.empty_like(v)
    c = cfg.c()
    lib().wo_ab(C.byref(c), _dp(v), _dp(a), _dp(b))
    return a, b


def potential_sub(cfg: Config):
    """-> (kind, scalar, array|None); kind 0 none / 1 scalar / 2 unpadded array."""
    c = cfg.c()
    kind = C.c_int(0)
    scalar = C.c_double(0.0)
    lib().wo_potential_sub(C.byref(c), C.byref(kind), C.byref(scalar), None)
    arr = None
    if kind.value == 2:
        arr = np.zeros(cfg.work_shape)
        lib().wo_potential_sub(C.byref(c), C.byref(kind), C.byref(scalar), _dp(arr))
    return kind.value, scalar.value, arr


def initial_condition(cfg: Config, ic: str, seed: int = 0) -> np.ndarray:
    phi = np.zeros(cfg.padded_shape)
    c = cfg.c()
    if lib().wo_initial_condition(C.byref(c), INITIAL_CONDITIONS.index(ic), seed, _dp(phi)):
        raise ValueError("unsupported initial condition " + ic)
    return phi


def norm2(cfg: Config, phi: np.ndarray) -> float:
    c = cfg.c()
    return lib().wo_norm2(C.byref(c), _dp(phi))


def normalise(phi: np.ndarray, n2: float) -> None:
    lib().wo_normalise(_dp(phi), phi.size, n2)


def orthogonalise(wnum: int, phi: np.ndarray, w_store) -> None:
    lib().wo_orthogonalise(wnum, _dp(phi), _store_ptrs(w_store), phi.size)


SYMMETRY = ["NotConstrained", "AboutZ", "AntisymAboutZ", "AboutY", "AntisymAboutY"]  # config.rs:184-197


def symmetrise(cfg: Config, kind: str, phi: np.ndarray) -> None:
    """config::symmetrise_wavefunction (config.rs:691-728), in place"""
    rc = lib().wo_symmetrise(C.byref(cfg.c()), SYMMETRY.index(kind), _dp(phi))
    if rc:
        raise ValueError("symmetry constraints index the SevenPoint frame (config.rs:702-725): ext must be 3")


def observables(cfg: Config, v, phi, potsub=(0, 0.0, None)):
    c = cfg.c()
    o = _Obs()
    kind, scalar, arr = potsub
    lib().wo_observables(C.byref(c), _dp(v), kind, scalar, _dp(arr) if arr is not None else None,
                         _dp(phi), C.byref(o))
    return dict(energy=o.energy, norm2=o.norm2, v_infinity=o.v_infinity, r2=o.r2)


def evolve(cfg: Config, wnum: int, a, b, phi, w_store, steps: int) -> None:
    c = cfg.c()
    lib().wo_evolve(C.byref(c), wnum, _dp(a), _dp(b), _dp(phi), _store_ptrs(w_store), steps)


def stencil_step(cfg: Config, a, b, phi) -> np.ndarray:
    c = cfg.c()
    work = np.zeros(cfg.work_shape)
    lib().wo_stencil_step(C.byref(c), _dp(a), _dp(b), _dp(phi), _dp(work))
    return work


def solve(cfg: Config, wnum: int, v, a, b, phi, w_store, tolerance: float, screen_update: int,
          max_steps=None, potsub=(0, 0.0, None), max_records: int = 100000):
    """grid.rs:50-246 for one state; phi is updated in place.
    -> (records: list[dict], converged: bool)"""
    c = cfg.c()
    recs = (_Record * max_records)()
    conv = C.c_int(0)
    kind, scalar, arr = potsub
    n = lib().wo_solve(C.byref(c), wnum, _dp(v), _dp(a), _dp(b), kind, scalar,
                       _dp(arr) if arr is not None else None, _dp(phi), _store_ptrs(w_store),
                       tolerance, screen_update, 0 if max_steps is None else 1,
                       0 if max_steps is None else int(max_steps), recs, max_records,
                       C.byref(conv))
    out = []
    for i in range(min(n, max_records)):
        r = recs[i]
        out.append(dict(step=r.step, tau=r.tau, energy=r.energy, norm2=r.norm2,
                        v_infinity=r.v_infinity, r2=r.r2, diff=r.diff))
    return out, bool(conv.value)


def trilerp_resize(v: np.ndarray, size, basis=None) -> np.ndarray:
    """input.rs:667-716.  `basis` = the `size` argument the reference builds its
    linspace from (defaults to the output dims, as in the reference's unit test;
    the production call passes the padded target size)."""
    v = np.ascontiguousarray(v, dtype=np.float64)
    out = np.zeros(tuple(size))
    basis = tuple(size) if basis is None else tuple(basis)
    if lib().wo_trilerp_resize_basis(_dp(v), v.shape[0], v.shape[1], v.shape[2], _dp(out),
                                     size[0], size[1], size[2], basis[0], basis[1], basis[2]) != 0:
        raise ValueError("trilerp_resize: every axis of the source needs at least two points (the reference panics)")
    return out


# ---- z-windows of grids whose arrays do not fit the host (wafer_oracle.h) ---------------------------------------------------
def potential_generate_zwindow(cfg: Config, zp0: int, zcount: int) -> np.ndarray:
    """potential.rs:46-62 on the global padded planes [zp0, zp0 + zcount): (px, py, zcount)"""
    px, py, _ = cfg.padded_shape
    v = np.zeros((px, py, zcount))
    rc = lib().wo_potential_generate_zwindow(C.byref(cfg.c()), zp0, zcount, _dp(v))
    if rc:
        raise ValueError("PotentialNotAvailable" if rc == 1 else "window outside the padded grid")
    return v


def initial_condition_zwindow(cfg: Config, ic: str, zp0: int, zcount: int, seed: int = 0) -> np.ndarray:
    px, py, _ = cfg.padded_shape
    phi = np.zeros((px, py, zcount))
    rc = lib().wo_initial_condition_zwindow(C.byref(cfg.c()), INITIAL_CONDITIONS.index(ic), seed, zp0, zcount, _dp(phi))
    if rc:
        raise ValueError("unsupported initial condition " + ic if rc == 1 else "window outside the padded grid")
    return phi


def ab_n(dt: float, v: np.ndarray):
    """potential.rs:101-110, elementwise over any array"""
    a, b = np.empty_like(v), np.empty_like(v)
    lib().wo_ab_n(dt, _dp(v), _dp(a), _dp(b), v.size)
    return a, b


def trilerp_resize_zwindow(v: np.ndarray, size_xy, zbegin: int, zcount: int, basis) -> np.ndarray:
    """planes [zbegin, zbegin + zcount) of trilerp_resize(v, (sx, sy, .), basis): (sx, sy, zcount)"""
    v = np.ascontiguousarray(v, dtype=np.float64)
    out = np.zeros((size_xy[0], size_xy[1], zcount))
    if lib().wo_trilerp_resize_basis_zwindow(_dp(v), v.shape[0], v.shape[1], v.shape[2], _dp(out), size_xy[0], size_xy[1],
                                             zbegin, zcount, basis[0], basis[1], basis[2]) != 0:
        raise ValueError("trilerp_resize: every axis of the source needs at least two points (the reference panics)")
    return out


def evolve_zwindow(cfg: Config, zp0: int, a: np.ndarray, b: np.ndarray, phi: np.ndarray, steps: int, storage=None):
    """`steps` ground-state steps (grid.rs:544-687) of the window phi = global padded planes [zp0, zp0 + zcount), in place, as a
    grid of its own: wo_evolve on a config with nz = zcount - 2 ext (the stencil has no notion of position; a, b are the
    window's).  Returns (lo, hi): the window planes [lo, hi) that equal the global run's afterwards -- a window end inside the
    grid loses ext planes per step, one that is the global frame none.  `storage` (np.float32): the result of every step is
    rounded to that type, which is what a device array of that type holds (dtype "f32": fp32 storage, fp64 arithmetic)."""
    e = cfg.ext
    zcount = phi.shape[2]
    pz = cfg.nz + 2 * e
    assert phi.shape[:2] == cfg.padded_shape[:2] and (zp0 == 0 or zp0 >= e) and zcount > 2 * e
    local = Config(cfg.nx, cfg.ny, zcount - 2 * e, ext=e, potential=cfg.potential, dn=cfg.dn, dt=cfg.dt, mass=cfg.mass, sig=cfg.sig)
    c = local.c()
    for _ in range(steps):
        lib().wo_evolve(C.byref(c), 0, _dp(a), _dp(b), _dp(phi), _store_ptrs([]), 1)
        if storage is not None:
            phi[...] = phi.astype(storage).astype(np.float64)
    lo = 0 if zp0 == 0 else steps * e
    hi = zcount if zp0 + zcount == pz else zcount - steps * e
    return lo, hi
