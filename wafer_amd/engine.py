"""ctypes binding of include/wafer_hip.h.

`Context` mirrors the reference's private functions in src/grid.rs one to one
(evolve, compute_observables, get_norm_squared, normalise_wavefunction,
orthogonalise_wavefunction, solve) with device-resident state.  Host arrays are
numpy float64 in the reference's layout: C-order [x][y][z], shape (n + 2*ext).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# WAFER_HIP_LIB: another build of the same library (same-box A/B runs of two kernel versions, tools/gpu_batch.sh ab_prev);
# it is still a HIP build of this engine -- there is no other implementation to point this at
_LIB = os.environ.get("WAFER_HIP_LIB") or os.path.join(_HERE, "libwafer_hip.so")

POTENTIALS = [
    "NoPotential", "Cube", "QuadWell", "Periodic", "Coulomb", "ComplexCoulomb",
    "ElipticalCoulomb", "SimpleCornell", "FullCornell", "Harmonic", "ComplexHarmonic",
    "Dodecahedron", "FromFile", "FromScript",
]  # config.rs:73-104
SYMMETRY = ["NotConstrained", "AboutZ", "AntisymAboutZ", "AboutY", "AntisymAboutY"]  # config.rs:184-197
INITIAL_CONDITIONS = ["FromFile", "Gaussian", "Coulomb", "Constant", "Boolean"]  # config.rs:151-170
CENTRAL_DIFFERENCE = {"ThreePoint": 1, "FivePoint": 2, "SevenPoint": 3}  # config.rs:224-238

WAFER_OK = 0
WAFER_ERR_MAX_STEP = -5
FLAG_SKIP_DT_CHECK = 1
FLAG_UNPLANNED_DIV = 2

EXPORTS = [
    "wafer_abi_version", "wafer_last_error", "wafer_ctx_create", "wafer_ctx_destroy",
    "wafer_synchronize", "wafer_set_potential_builtin", "wafer_set_potential_host",
    "wafer_download_array", "wafer_get_potsub", "wafer_set_initial_condition", "wafer_upload_phi",
    "wafer_download_phi", "wafer_upload_phi_resampled", "wafer_set_potential_resampled", "wafer_evolve", "wafer_observables", "wafer_norm2", "wafer_normalise",
    "wafer_orthogonalise", "wafer_push_state", "wafer_load_state", "wafer_download_state",
    "wafer_clone_state_to_phi", "wafer_num_states", "wafer_clear_states", "wafer_solve_state",
    "wafer_last_evolve_ms", "wafer_stencil_kernel_name", "wafer_stencil_kernel_instance", "wafer_stencil_steps_per_launch", "wafer_set_stencil_variant",
    "wafer_set_comm_hooks", "wafer_set_overlap", "wafer_set_stream", "wafer_get_slab_info",
    "wafer_get_device_info", "wafer_set_potsub", "wafer_set_potsub_resampled", "wafer_symmetrise", "wafer_download_phi_owned", "wafer_diag_div_check", "wafer_div_plan", "wafer_get_div_plan", "wafer_diag_div_planned", "wafer_div_plan_f32", "wafer_diag_div_planned_f32",
    "wafer_diag_copy_bw", "wafer_diag_checksum", "wafer_diag_download_window", "wafer_diag_dispatch", "wafer_set_halo_cycle", "wafer_diag_x2_passes",
    "wafer_peer_export", "wafer_peer_connect", "wafer_peer_disconnect",
]


class _DivPlan(C.Structure):
    """wafer_div_plan_t (include/wafer_hip.h)"""
    _fields_ = [("den", C.c_double), ("zh", C.c_double), ("zl", C.c_double), ("checked", C.c_int32), ("n_candidates", C.c_int32),
                ("zl_shift", C.c_int32), ("reserved", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


class _DivPlanF32(C.Structure):
    """wafer_div_plan_f32_t (include/wafer_hip.h)"""
    _fields_ = [("den", C.c_float), ("zh", C.c_float), ("zl", C.c_float), ("checked", C.c_int32), ("zl_shift", C.c_int32)]


class WaferError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"wafer_hip error {code}: {msg}")
        self.code = code


class _PeerInfo(C.Structure):
    """wafer_peer_info (include/wafer_hip.h)"""
    _fields_ = [
        ("struct_size", C.c_uint32), ("z_begin", C.c_uint32), ("z_count", C.c_uint32), ("halo_depth", C.c_uint32),
        ("pid", C.c_uint64), ("phi_addr", C.c_uint64 * 2), ("flags_addr", C.c_uint64), ("phi_alloc_offset", C.c_uint64 * 2),
        ("phi_ipc", (C.c_uint8 * 64) * 2), ("flags_ipc", C.c_uint8 * 64),
        ("process_nonce", C.c_uint64), ("device", C.c_int32), ("reserved", C.c_uint32), ("device_uuid", C.c_uint8 * 16),
    ]


class _Params(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("nx", C.c_uint32), ("ny", C.c_uint32), ("nz", C.c_uint32),
        ("central_difference", C.c_int32), ("dtype", C.c_int32),
        ("dn", C.c_double), ("dt", C.c_double), ("mass", C.c_double), ("sig", C.c_double),
        ("max_states", C.c_uint32), ("device", C.c_int32),
        ("z_begin", C.c_uint32), ("z_count", C.c_uint32), ("halo_depth", C.c_uint32),
        ("flags", C.c_uint32),
    ]


class _Obs(C.Structure):
    _fields_ = [("energy", C.c_double), ("norm2", C.c_double), ("v_infinity", C.c_double),
                ("r2", C.c_double)]


class _Record(C.Structure):
    _fields_ = [("step", C.c_uint64), ("tau", C.c_double), ("obs", _Obs), ("diff", C.c_double)]


class _ObsOut(C.Structure):
    _fields_ = [("state", C.c_uint32), ("energy", C.c_double), ("binding_energy", C.c_double),
                ("r", C.c_double), ("l_r", C.c_double)]


class _DeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("arch", C.c_char * 64), ("compute_units", C.c_uint32),
                ("memory_clock_khz", C.c_uint32), ("memory_bus_bits", C.c_uint32), ("l2_bytes", C.c_uint32),
                ("total_bytes", C.c_uint64)]


class _SlabInfo(C.Structure):
    _fields_ = [("z_begin", C.c_uint32), ("z_count", C.c_uint32), ("halo_depth", C.c_uint32),
                ("ext", C.c_uint32), ("plane_elems", C.c_uint64), ("elem_bytes", C.c_uint64)]


HALO_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                      C.c_size_t, C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

_lib = None


def library_path() -> str:
    return _LIB


def load_library():
    """Loads libwafer_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process.  PyTorch's ROCm wheels bundle their own libamdhip64.so.7 /
    # libhsa-runtime64 with the SAME sonames as /opt/rocm's, so whichever loads first serves
    # every later user.  If this library pulled in /opt/rocm's first, a later torch.cuda init
    # fails ("No HIP GPUs are available") and slab.py could not alias the engine's buffers and
    # streams.  Importing torch first makes its runtime the process-wide one.  A host without
    # torch (a Rust binary, WAFER_PRELOAD_TORCH=0) simply links /opt/rocm's.
    if os.environ.get("WAFER_PRELOAD_TORCH", "1") != "0":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(_LIB):
        raise ImportError(
            f"{_LIB} is missing: build it with `python -m wafer_amd.build` "
            "(hipcc --offload-arch=gfx950). The engine has no CPU fallback.")
    L = C.CDLL(_LIB)
    vp, dp = C.c_void_p, C.POINTER(C.c_double)
    L.wafer_abi_version.restype = C.c_int
    L.wafer_last_error.restype = C.c_char_p
    L.wafer_ctx_create.argtypes = [C.POINTER(_Params), C.POINTER(vp)]
    L.wafer_ctx_destroy.argtypes = [vp]
    L.wafer_synchronize.argtypes = [vp]
    L.wafer_set_potential_builtin.argtypes = [vp, C.c_int]
    L.wafer_set_potential_host.argtypes = [vp, dp, C.c_int, C.c_double, dp]
    L.wafer_download_array.argtypes = [vp, C.c_int, dp]
    L.wafer_get_potsub.argtypes = [vp, C.POINTER(C.c_int), dp]
    L.wafer_set_potsub.argtypes = [vp, C.c_int, C.c_double, dp]
    L.wafer_set_potsub_resampled.argtypes = [vp, dp, C.c_uint32, C.c_uint32, C.c_uint32]
    L.wafer_symmetrise.argtypes = [vp, C.c_int]
    L.wafer_download_phi_owned.argtypes = [vp, dp]
    L.wafer_diag_div_check.argtypes = [vp, C.c_double, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
    L.wafer_div_plan.argtypes = [C.c_double, C.POINTER(_DivPlan), dp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.wafer_get_div_plan.argtypes = [vp, C.POINTER(_DivPlan)]
    L.wafer_diag_div_planned.argtypes = [vp, C.POINTER(_DivPlan), C.c_uint64, C.c_uint64, C.c_int, C.c_int, dp, C.c_size_t,
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.wafer_div_plan_f32.argtypes = [C.c_float, C.POINTER(_DivPlanF32)]
    L.wafer_diag_div_planned_f32.argtypes = [vp, C.POINTER(_DivPlanF32), C.c_int, C.c_int, C.POINTER(C.c_uint64)]
    L.wafer_diag_copy_bw.argtypes = [vp, C.c_int, C.c_int, C.c_int, dp]
    L.wafer_diag_checksum.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
    L.wafer_diag_x2_passes.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.wafer_diag_dispatch.argtypes = [vp, C.c_uint32, C.c_char_p, C.c_size_t]
    L.wafer_diag_download_window.argtypes = [vp, C.c_int, C.c_uint32, C.c_uint32, dp]
    L.wafer_peer_export.argtypes = [vp, C.POINTER(_PeerInfo)]
    L.wafer_peer_connect.argtypes = [vp, C.POINTER(_PeerInfo), C.POINTER(_PeerInfo)]
    L.wafer_peer_disconnect.argtypes = [vp]
    L.wafer_set_initial_condition.argtypes = [vp, C.c_int, C.c_uint64]
    L.wafer_upload_phi.argtypes = [vp, dp]
    L.wafer_download_phi.argtypes = [vp, dp]
    u32p = C.POINTER(C.c_uint32)
    L.wafer_upload_phi_resampled.argtypes = [vp, dp, C.c_uint32, C.c_uint32, C.c_uint32, u32p]
    L.wafer_set_potential_resampled.argtypes = [vp, dp, C.c_uint32, C.c_uint32, C.c_uint32, u32p]
    L.wafer_evolve.argtypes = [vp, C.c_uint32, C.c_uint64]
    L.wafer_observables.argtypes = [vp, C.POINTER(_Obs)]
    L.wafer_norm2.argtypes = [vp, dp]
    L.wafer_normalise.argtypes = [vp, C.c_double]
    L.wafer_orthogonalise.argtypes = [vp, C.c_uint32]
    L.wafer_push_state.argtypes = [vp]
    L.wafer_load_state.argtypes = [vp, C.c_uint32, dp]
    L.wafer_download_state.argtypes = [vp, C.c_uint32, dp]
    L.wafer_clone_state_to_phi.argtypes = [vp, C.c_uint32]
    L.wafer_num_states.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.wafer_clear_states.argtypes = [vp]
    L.wafer_solve_state.argtypes = [vp, C.c_uint32, C.c_double, C.c_uint64, C.c_int, C.c_uint64,
                                    C.POINTER(_Record), C.c_size_t, C.POINTER(C.c_size_t),
                                    C.POINTER(_ObsOut)]
    L.wafer_last_evolve_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
    L.wafer_stencil_kernel_name.argtypes = [vp]
    L.wafer_stencil_kernel_name.restype = C.c_char_p
    L.wafer_stencil_kernel_instance.argtypes = [vp]
    L.wafer_stencil_kernel_instance.restype = C.c_char_p
    L.wafer_stencil_steps_per_launch.argtypes = [vp]
    L.wafer_set_stencil_variant.argtypes = [vp, C.c_int]
    L.wafer_set_comm_hooks.argtypes = [vp, HALO_FN, ALLREDUCE_FN, vp]
    L.wafer_set_overlap.argtypes = [vp, C.c_int]
    L.wafer_set_halo_cycle.argtypes = [vp, C.c_int]
    L.wafer_set_stream.argtypes = [vp, vp]
    L.wafer_get_slab_info.argtypes = [vp, C.POINTER(_SlabInfo)]
    L.wafer_get_device_info.argtypes = [vp, C.POINTER(_DeviceInfo)]
    if L.wafer_abi_version() != 1:
        raise ImportError("libwafer_hip.so ABI version mismatch")
    _lib = L
    return L


@dataclass
class Params:
    """The subset of Config (config.rs:292-333) the hot path reads + engine knobs."""
    nx: int
    ny: int
    nz: int
    dn: float
    dt: float
    mass: float = 1.0
    sig: float = 1.0
    central_difference: int = 1      # 1/2/3 or "ThreePoint"/"FivePoint"/"SevenPoint"
    dtype: str = "f64"               # "f64" | "f32" (fp32 storage, fp64 arithmetic) | "f32fast" (+ fp32 step arithmetic)
    max_states: int = 4
    device: int = 0
    z_begin: int = 0
    z_count: int = 0                 # 0 = whole grid
    halo_depth: int = 0              # 0 = ext
    skip_dt_check: bool = False
    unplanned_div: bool = False      # WAFER_FLAG_UNPLANNED_DIV: x / (c dn^2 m) always with the extra Markstein round

    @property
    def ext(self) -> int:
        cd = self.central_difference
        return CENTRAL_DIFFERENCE[cd] if isinstance(cd, str) else int(cd)

    @property
    def padded_shape(self):
        e = self.ext
        return (self.nx + 2 * e, self.ny + 2 * e, self.nz + 2 * e)

    @property
    def work_shape(self):
        return (self.nx, self.ny, self.nz)

    def c(self) -> _Params:
        return _Params(C.sizeof(_Params), self.nx, self.ny, self.nz, self.ext,
                       {"f64": 0, "f32": 1, "f32fast": 2}[self.dtype], self.dn, self.dt, self.mass, self.sig,
                       self.max_states, self.device, self.z_begin, self.z_count, self.halo_depth,
                       (FLAG_SKIP_DT_CHECK if self.skip_dt_check else 0) | (FLAG_UNPLANNED_DIV if self.unplanned_div else 0))


def div_plan(den: float, max_candidates: int = 4096):
    """(plan, candidates): wafer_div_plan -- host only, no GPU: how the step kernels will divide by `den`, and the significands
    (doubles in [2^52, 2^53)) the plan had to try"""
    L = load_library()
    p = _DivPlan()
    cand = np.zeros(max_candidates)
    n = C.c_size_t(0)
    rc = L.wafer_div_plan(den, C.byref(p), _dp(cand), cand.size, C.byref(n))
    if rc != 0:
        raise WaferError(rc, L.wafer_last_error().decode())
    return p, cand[:n.value].copy()


def div_plan_f32(den: float) -> _DivPlanF32:
    """wafer_div_plan_f32 -- host only: the fp32 plan of x / den (all 2^23 significands tried)"""
    L = load_library()
    p = _DivPlanF32()
    rc = L.wafer_div_plan_f32(den, C.byref(p))
    if rc != 0:
        raise WaferError(rc, L.wafer_last_error().decode())
    return p


def _dp(a: np.ndarray):
    if a.dtype != np.float64 or not a.flags["C_CONTIGUOUS"]:
        raise TypeError("host arrays must be C-contiguous float64")
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Context:
    """Device-resident solver state for one grid (or one z-slab of it)."""

    def __init__(self, params: Params):
        self._L = load_library()
        self.params = params
        self._h = C.c_void_p()
        p = params.c()
        self._check(self._L.wafer_ctx_create(C.byref(p), C.byref(self._h)))
        self._hooks = None

    # -- plumbing -----------------------------------------------------------------
    def _check(self, rc: int) -> None:
        if rc != WAFER_OK:
            raise WaferError(rc, self._L.wafer_last_error().decode())

    def close(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            self._L.wafer_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def handle(self):
        return self._h

    def synchronize(self) -> None:
        self._check(self._L.wafer_synchronize(self._h))

    # -- Potentials (potential.rs:16-25, 75-175) --------------------------------------
    def set_potential(self, name: str) -> None:
        self._check(self._L.wafer_set_potential_builtin(self._h, POTENTIALS.index(name)))

    def set_potential_host(self, v: np.ndarray, potsub_kind: int = 0, potsub_scalar: float = 0.0,
                           potsub: np.ndarray | None = None) -> None:
        assert v.shape == self.params.padded_shape
        self._check(self._L.wafer_set_potential_host(
            self._h, _dp(v), potsub_kind, potsub_scalar, _dp(potsub) if potsub is not None else None))

    def download_array(self, which: str) -> np.ndarray:
        idx = {"v": 0, "a": 1, "b": 2, "potsub": 3}[which]
        out = np.zeros(self.params.work_shape if which == "potsub" else self.params.padded_shape)
        self._check(self._L.wafer_download_array(self._h, idx, _dp(out)))
        return out

    def download_window(self, which: str, zp_begin: int, zp_count: int) -> np.ndarray:
        """global padded planes [zp_begin, zp_begin + zp_count) of "v" / "a" / "b" / "phi" in the reference's layout:
        (px, py, zp_count) -- download_array / download_phi restricted to a z-window (wafer_diag_download_window)"""
        px, py, _ = self.params.padded_shape
        out = np.zeros((px, py, zp_count))
        self._check(self._L.wafer_diag_download_window(self._h, {"v": 0, "a": 1, "b": 2, "phi": 4}[which], zp_begin, zp_count, _dp(out)))
        return out

    def set_potsub(self, kind: int, scalar: float = 0.0, potsub: np.ndarray | None = None) -> None:
        """override pot_sub (a potential_sub file in ./input, potential.rs:113-131)"""
        if potsub is not None:
            assert potsub.shape == self.params.work_shape
        self._check(self._L.wafer_set_potsub(self._h, kind, scalar, _dp(potsub) if potsub is not None else None))

    def set_potsub_resampled(self, src: np.ndarray) -> None:
        """pot_sub from an array of another resolution (input::fill_sub_data, input.rs:453-478)"""
        src = np.ascontiguousarray(src, dtype=np.float64)
        assert src.ndim == 3
        self._check(self._L.wafer_set_potsub_resampled(self._h, _dp(src), *src.shape))

    def potsub(self):
        kind, scalar = C.c_int(0), C.c_double(0.0)
        self._check(self._L.wafer_get_potsub(self._h, C.byref(kind), C.byref(scalar)))
        return kind.value, scalar.value

    # -- phi ----------------------------------------------------------------------------
    def set_initial_condition(self, name: str, seed: int = 0) -> None:
        self._check(self._L.wafer_set_initial_condition(self._h, INITIAL_CONDITIONS.index(name), seed))

    def symmetrise(self, constraint: str) -> None:
        """config::symmetrise_wavefunction (config.rs:691-728); SevenPoint only, like the reference"""
        self._check(self._L.wafer_symmetrise(self._h, SYMMETRY.index(constraint)))

    def div_check(self, den: float, n_operands: int, lo_exp: int = 64, hi_exp: int = 1983, seed: int = 1) -> int:
        """operands (of n_operands random ones) whose hoisted-reciprocal quotient by `den` differs from the
        IEEE division in any bit (wafer_diag_div_check)"""
        bad = C.c_uint64(0)
        self._check(self._L.wafer_diag_div_check(self._h, den, seed, n_operands, lo_exp, hi_exp, C.byref(bad)))
        return int(bad.value)

    def div_plan(self) -> _DivPlan:
        """the plan of x / (c dn^2 m) this context's kernels run with (wafer_get_div_plan)"""
        p = _DivPlan()
        self._check(self._L.wafer_get_div_plan(self._h, C.byref(p)))
        return p

    def div_planned_check(self, plan: _DivPlan, n_random: int = 0, operands=None, lo_exp: int = 64, hi_exp: int = 1983, seed: int = 1):
        """(mismatches among n_random drawn operands, mismatches among `operands`): the planned division as the kernels perform
        it against the device's IEEE x / den, bit for bit (wafer_diag_div_planned)"""
        ops = np.ascontiguousarray(operands, dtype=np.float64) if operands is not None else np.zeros(0)
        bad_r, bad_o = C.c_uint64(0), C.c_uint64(0)
        self._check(self._L.wafer_diag_div_planned(self._h, C.byref(plan), seed, n_random, lo_exp, hi_exp,
                                                   _dp(ops) if ops.size else None, ops.size, C.byref(bad_r), C.byref(bad_o)))
        return int(bad_r.value), int(bad_o.value)

    def div_planned_check_f32(self, plan: _DivPlanF32, lo_exp: int = 27, hi_exp: int = 227) -> int:
        """floats (ALL significands, both signs, biased exponents lo_exp .. hi_exp) whose planned fp32 quotient differs from the
        device's IEEE x / den in any bit (wafer_diag_div_planned_f32)"""
        bad = C.c_uint64(0)
        self._check(self._L.wafer_diag_div_planned_f32(self._h, C.byref(plan), lo_exp, hi_exp, C.byref(bad)))
        return int(bad.value)

    def download_phi_owned(self) -> np.ndarray:
        """the work cells of the planes this context owns, (nx, ny, z_count)"""
        p = self.params
        out = np.zeros((p.nx, p.ny, p.z_count if p.z_count else p.nz))
        self._check(self._L.wafer_download_phi_owned(self._h, _dp(out)))
        return out

    def upload_phi(self, phi: np.ndarray) -> None:
        assert phi.shape == self.params.padded_shape
        self._check(self._L.wafer_upload_phi(self._h, _dp(phi)))

    @staticmethod
    def _basis(basis):
        return None if basis is None else (C.c_uint32 * 3)(*basis)

    def upload_phi_resampled(self, src: np.ndarray, basis=None) -> None:
        """input.rs:667-716: trilinear resample of an unpadded array of another resolution"""
        self._check(self._L.wafer_upload_phi_resampled(self._h, _dp(src), *src.shape, self._basis(basis)))

    def set_potential_resampled(self, src: np.ndarray, basis=None) -> None:
        self._check(self._L.wafer_set_potential_resampled(self._h, _dp(src), *src.shape, self._basis(basis)))

    def download_phi(self, out: np.ndarray | None = None) -> np.ndarray:
        if out is None:
            out = np.zeros(self.params.padded_shape)
        self._check(self._L.wafer_download_phi(self._h, _dp(out)))
        return out

    # -- hot path (grid.rs) ---------------------------------------------------------------
    def evolve(self, wnum: int, steps: int) -> None:
        """grid.rs:544-687"""
        self._check(self._L.wafer_evolve(self._h, wnum, steps))

    def observables(self) -> dict:
        """grid.rs:303-445 (un-normalised)"""
        o = _Obs()
        self._check(self._L.wafer_observables(self._h, C.byref(o)))
        return dict(energy=o.energy, norm2=o.norm2, v_infinity=o.v_infinity, r2=o.r2)

    def norm2(self) -> float:
        """grid.rs:454-457"""
        v = C.c_double(0.0)
        self._check(self._L.wafer_norm2(self._h, C.byref(v)))
        return v.value

    def normalise(self, norm2: float) -> None:
        """grid.rs:465-468"""
        self._check(self._L.wafer_normalise(self._h, norm2))

    def orthogonalise(self, wnum: int) -> None:
        """grid.rs:477-492"""
        self._check(self._L.wafer_orthogonalise(self._h, wnum))

    # -- w_store ---------------------------------------------------------------------------
    def push_state(self) -> None:
        self._check(self._L.wafer_push_state(self._h))

    def load_state(self, idx: int, state: np.ndarray) -> None:
        assert state.shape == self.params.padded_shape
        self._check(self._L.wafer_load_state(self._h, idx, _dp(state)))

    def download_state(self, idx: int) -> np.ndarray:
        out = np.zeros(self.params.padded_shape)
        self._check(self._L.wafer_download_state(self._h, idx, _dp(out)))
        return out

    def clone_state_to_phi(self, idx: int) -> None:
        self._check(self._L.wafer_clone_state_to_phi(self._h, idx))

    def num_states(self) -> int:
        n = C.c_uint32(0)
        self._check(self._L.wafer_num_states(self._h, C.byref(n)))
        return n.value

    def clear_states(self) -> None:
        self._check(self._L.wafer_clear_states(self._h))

    # -- solve (grid.rs:50-246) --------------------------------------------------------------
    def solve_state(self, wnum: int, tolerance: float, screen_update: int, max_steps=None,
                    max_records: int = 100000):
        """-> (records, final, converged).  Raises on errors other than MaxStep."""
        recs = (_Record * max_records)()
        n = C.c_size_t(0)
        fin = _ObsOut()
        rc = self._L.wafer_solve_state(self._h, wnum, tolerance, screen_update,
                                       0 if max_steps is None else 1,
                                       0 if max_steps is None else int(max_steps), recs, max_records,
                                       C.byref(n), C.byref(fin))
        if rc not in (WAFER_OK, WAFER_ERR_MAX_STEP):
            self._check(rc)
        out = []
        for i in range(min(n.value, max_records)):
            r = recs[i]
            out.append(dict(step=r.step, tau=r.tau, energy=r.obs.energy, norm2=r.obs.norm2,
                            v_infinity=r.obs.v_infinity, r2=r.obs.r2, diff=r.diff))
        final = dict(state=fin.state, energy=fin.energy, binding_energy=fin.binding_energy,
                     r=fin.r, l_r=fin.l_r)
        return out, final, rc == WAFER_OK

    # -- measurement ----------------------------------------------------------------------------
    def last_evolve_ms(self):
        ms, steps = C.c_float(0.0), C.c_uint64(0)
        self._check(self._L.wafer_last_evolve_ms(self._h, C.byref(ms), C.byref(steps)))
        return ms.value, steps.value

    def stencil_kernel_name(self) -> str:
        return self._L.wafer_stencil_kernel_name(self._h).decode()

    def stencil_kernel_instance(self) -> str:
        """template-id of the kernel the last ground-state pass launched, as a profiler prints it"""
        return self._L.wafer_stencil_kernel_instance(self._h).decode()

    def steps_per_launch(self) -> int:
        return int(self._L.wafer_stencil_steps_per_launch(self._h))

    def set_stencil_variant(self, variant: int) -> None:
        self._check(self._L.wafer_set_stencil_variant(self._h, variant))

    def copy_bandwidth(self, iters: int = 50, unroll: int = 4, blocks_per_cu: int = 1) -> float:
        """measured GB/s (read + written) of a 16 B-per-lane device copy: the device's own ceiling"""
        v = C.c_double(0.0)
        self._check(self._L.wafer_diag_copy_bw(self._h, iters, unroll, blocks_per_cu, C.byref(v)))
        return v.value

    def dispatch(self, wnum: int = 0) -> dict:
        """which kernel a pass of evolve(wnum, .) launches for this context (wafer_diag_dispatch): {"kernel": ..., "steps_per_pass": ...,
        "ghost_planes_per_pass": ..., "tile": ..., "v": "streamed" | "closed_form", ...} -- the launch path's own predicates"""
        buf = C.create_string_buffer(512)
        self._check(self._L.wafer_diag_dispatch(self._h, wnum, buf, len(buf)))
        out = dict(kv.split("=", 1) for kv in buf.value.decode().split())
        for k in ("wnum", "stencil", "steps_per_pass", "ghost_planes_per_pass", "nlow", "waves"):
            if k in out:
                out[k] = int(out[k])
        return out

    def x2_passes(self) -> int:
        """passes of the two-excited-steps-per-pass kernel launched so far (include/wafer_hip.h)"""
        v = C.c_uint64(0)
        self._check(self._L.wafer_diag_x2_passes(self._h, C.byref(v)))
        return int(v.value)

    def checksum(self, z_begin: int = 0, z_count: int | None = None) -> int:
        """position-dependent integer checksum of the owned work cells of global planes
        [z_begin, z_begin + z_count): equal bits <=> equal checksums, whatever the decomposition"""
        v = C.c_uint64(0)
        n = self.params.nz if z_count is None else z_count
        self._check(self._L.wafer_diag_checksum(self._h, z_begin, n, C.byref(v)))
        return int(v.value)

    # -- multi-GPU ---------------------------------------------------------------------------------
    def set_comm_hooks(self, halo, allreduce) -> None:
        """halo(send_lo, send_hi, recv_lo, recv_hi, nbytes, stream) and
        allreduce(dev_ptr, count, stream): python callables taking raw device
        addresses (ints or None); they must return 0 on success."""
        def _halo(_user, slo, shi, rlo, rhi, nbytes, stream):
            try:
                return int(halo(slo, shi, rlo, rhi, nbytes, stream) or 0)
            except Exception as e:  # never unwind through C
                print("wafer halo hook failed:", repr(e), flush=True)
                return 1

        def _allreduce(_user, ptr, count, stream):
            try:
                return int(allreduce(ptr, count, stream) or 0)
            except Exception as e:
                print("wafer allreduce hook failed:", repr(e), flush=True)
                return 1

        self._hooks = (HALO_FN(_halo), ALLREDUCE_FN(_allreduce))  # keep alive
        self._check(self._L.wafer_set_comm_hooks(self._h, self._hooks[0], self._hooks[1], None))

    def peer_export(self) -> bytes:
        """this context's wafer_peer_info record (for the z-neighbours' peer_connect), as bytes any transport can carry"""
        info = _PeerInfo()
        self._check(self._L.wafer_peer_export(self._h, C.byref(info)))
        return bytes(info)

    def peer_connect(self, lower: bytes | None, upper: bytes | None) -> None:
        """map the z-neighbours' buffers and arrival counters (records from their peer_export; None: no neighbour on that side):
        the precondition of set_overlap(3)"""
        recs = [_PeerInfo.from_buffer_copy(r) if r is not None else None for r in (lower, upper)]
        self._check(self._L.wafer_peer_connect(self._h, C.byref(recs[0]) if recs[0] is not None else None,
                                               C.byref(recs[1]) if recs[1] is not None else None))

    def peer_disconnect(self) -> None:
        self._check(self._L.wafer_peer_disconnect(self._h))

    def set_overlap(self, enabled) -> None:
        """halo schedule of a z-slab: False / 0 (exchange after the pass), True / 1 (boundary planes first, three launches
        per pass), 2 (the default: one launch per three-step pass, exchanges released by completion counters), 3 (the same
        launch with peer stores into the neighbours' ghost planes instead of an exchange: after peer_connect);
        include/wafer_hip.h"""
        self._check(self._L.wafer_set_overlap(self._h, int(enabled)))

    def set_halo_cycle(self, passes: int) -> None:
        """fused passes per halo exchange of a z-slab (needs halo_depth >= K * ext * passes, K = steps per fused pass)"""
        self._check(self._L.wafer_set_halo_cycle(self._h, int(passes)))

    def set_stream(self, stream_ptr: int | None) -> None:
        self._check(self._L.wafer_set_stream(self._h, stream_ptr))

    def device_info(self) -> dict:
        d = _DeviceInfo()
        self._check(self._L.wafer_get_device_info(self._h, C.byref(d)))
        out = {k: getattr(d, k) for k, _ in _DeviceInfo._fields_}
        out["name"], out["arch"] = d.name.decode(), d.arch.decode()
        return out

    def slab_info(self) -> dict:
        s = _SlabInfo()
        self._check(self._L.wafer_get_slab_info(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in _SlabInfo._fields_}
