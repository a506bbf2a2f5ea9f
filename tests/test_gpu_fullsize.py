"""BASELINE configs #4 and #5 at their FULL single-device sizes: 1024^3 fp64 (43 GB of device arrays) and 2048^3 with
fp32 storage (182 GB -- skipped unless the GPU has that much free).  More than 2^32 elements per array, so every index in the
kernels has to be 64-bit clean.  Three kinds of check, none of which needs a host array of the grid's size:

1. CELL BY CELL against the oracle on z-WINDOWS of the grid (round 6).  The oracle evaluates the potential / the initial
   condition / the trilinear resample on the global padded planes [zp0, zp0 + W) only (wo_*_zwindow: the full-array functions
   ARE these with the window [0, pz)) and evolves the window as a grid of its own; after s steps its planes [s ext, W - s ext)
   are the global run's (a window end that is the global frame stays valid).  The engine returns the same planes through
   wafer_diag_download_window.  Config #4: 1024^3 fp64 SimpleCornell with its real parameters, V / a / b and phi after 3 + 2
   steps (one three-step pass + the two-step remainder) BIT EXACT at z = 0, the middle and the top, every x / y edge included.
   Config #5: 2048^3 fp32 storage, V from a 64^3 source through the device resampler (input.rs:149-176, 667-716): windows whose
   element offsets lie below 2^32, straddle 2^32 and exceed 2^33 equal (float) of the oracle's trilerp_resize, and phi after
   three steps equals the oracle's with every step's result rounded to float (fp32 storage, fp64 arithmetic) bit for bit.
2. Closed forms on an empty potential (below): absolute values of sums no window can give.
3. Path equalities (fused passes against single steps) as CHECKSUMS over every cell's bits (wafer_diag_checksum), on the
   configs' real potentials.

Boolean initial condition (config.rs:676-683: 1 where all three PADDED indices are odd) and no
potential (a = b = 1): with k = dt / (2 dn^2 m) one step (grid.rs:568-592) gives, exactly,
    1 - 6k                on the N^3/8 cells that held a 1,
    2k  (k next to the far frame: that neighbour is a frame zero)   on cells with ONE even index,
    0                     elsewhere,
so  sum phi'^2 = (N/2)^3 fl(1-6k)^2 + 3 (N/2)^2 [(N/2-1) fl(2k)^2 + fl(k)^2],  fl = rounding to
the storage type.  The sums before the step are exact integers."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def free_gib():
    import torch
    free, _total = torch.cuda.mem_get_info(0)
    return free / 2**30


def closed_forms(n, dn, dt, mass, store, arith32=False):
    k = dt / (2.0 * dn * dn * mass)
    half = n // 2
    # the engine computes w*a + b*dt*S/den in fp64 and rounds to the storage type; f32fast computes in fp32 as well:
    # every operand (dt, den) and every product / quotient / sum rounded to fp32 (grid.rs:580-589, left to right)
    ar = np.float32 if arith32 else np.float64
    den = ar(2.0 * dn * dn * mass)
    upd = lambda w, S: ar(ar(w) * ar(1.0)) + ar(ar(ar(1.0) * ar(dt)) * ar(S)) / den
    c1 = float(store(upd(1.0, -6.0)))
    c2 = float(store(upd(0.0, 2.0)))
    c3 = float(store(upd(0.0, 1.0)))
    norm2_0 = float(half) ** 3
    norm2_1 = half ** 3 * c1 ** 2 + 3 * half ** 2 * ((half - 1) * c2 ** 2 + c3 ** 2)
    # r^2 observable (grid.rs:428-437): WORK index i = 0, 2, 4, ... of the odd padded indices, centre (N+1)/2
    i = np.arange(0, n, 2, dtype=np.float64)
    r2_0 = 3 * half ** 2 * float(np.sum((i - (n + 1) / 2.0) ** 2))
    # energy (grid.rs:312-405) with V = 0: -sum w S / den, S = -6 on the occupied cells
    energy_0 = half ** 3 * 6.0 / (2.0 * dn * dn * mass)
    return k, norm2_0, norm2_1, r2_0, energy_0


def file_source(n_src=64):
    """the 64^3 user potential of config #5 (an anisotropic Poschl-Teller well, gen_potential.py:45-60 in spirit)"""
    ax = (np.arange(n_src) - (n_src - 1) / 2) * (12.8 / n_src)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    return np.ascontiguousarray(-3.0 / np.cosh(0.6 * np.sqrt(X * X + Y * Y + 2.0 * Z * Z)) ** 2)


def set_real_potential(ctx, potential):
    if potential == "file":
        ctx.set_potential_resampled(file_source())      # basis = the padded target size: the reference's production call
    else:
        ctx.set_potential(potential)


def run_case(wa, n, dtype, dn, dt, mass, potential, sig=1.0):
    """closed forms on V = 0, then the path equalities on the config's own potential, as checksums of every cell's bits
    (the single-step kernel run here is the LDS one with a, b from V: no stored a, b arrays)"""
    store = np.float64 if dtype == "f64" else np.float32
    k, n0, n1, r2, e0 = closed_forms(n, dn, dt, mass, store, arith32=dtype == "f32fast")
    par = wa.Params(n, n, n, dn=dn, dt=dt, mass=mass, sig=sig, dtype=dtype, max_states=1)
    with wa.Context(par) as ctx:
        ctx.set_potential("NoPotential")
        ctx.set_initial_condition("Boolean")
        obs = ctx.observables()
        assert obs["norm2"] == n0 and ctx.norm2() == n0                 # integer sums: exact
        assert obs["r2"] == pytest.approx(r2, rel=1e-13)
        assert obs["energy"] == pytest.approx(e0, rel=1e-13) and obs["v_infinity"] == 0.0
        ctx.set_stencil_variant(1)                                      # one single step
        ctx.evolve(0, 1)
        assert ctx.norm2() == pytest.approx(n1, rel=1e-12)

        # from here on the config's own potential, and every cell's bits
        set_real_potential(ctx, potential)
        nz = n
        thirds = [(0, nz // 3), (nz // 3, nz // 3), (2 * (nz // 3), nz - 2 * (nz // 3))]
        sums = lambda: tuple(ctx.checksum(z0, zc) for z0, zc in thirds)    # three plane ranges: a wrong plane is located
        ctx.set_initial_condition("Boolean")
        ctx.evolve(0, 1)
        ctx.evolve(0, 1)
        two_single = sums()
        ctx.evolve(0, 1)
        three_single = sums()
        assert len(set(two_single + three_single)) == 6                  # not a degenerate checksum
        # the default path from the same start equals two single steps (two steps cannot fill a three-step pass: the
        # fused kernels' own remainder handling runs here)
        ctx.set_stencil_variant(-1)
        ctx.set_initial_condition("Boolean")
        ctx.evolve(0, 2)
        assert ctx.stencil_kernel_name() == "wafer_k_step3_fused"    # every dtype (fp32 storage with fp64 arithmetic since round 5)
        assert sums() == two_single
        two_norm = ctx.norm2()
        # ... and three fused steps (ThreePoint: one pass of the three-step kernel) equal three single steps
        ctx.set_initial_condition("Boolean")
        ctx.evolve(0, 3)
        assert sums() == three_single
        ctx.set_initial_condition("Boolean")
        ctx.evolve(0, 2)
        # linearity: twice the wavefunction, four times the norm, bit for bit (powers of two)
        ctx.normalise(0.25)                                             # phi / sqrt(1/4) = 2 phi
        assert ctx.norm2() == 4.0 * two_norm
        ctx.evolve(0, 6)
        ms, steps = ctx.last_evolve_ms()
        assert np.isfinite(ctx.norm2()) and steps == 6
        return ms / steps


# ---------------------------------------------------------------------------------------------------------------------------
# cell by cell against the oracle, on z-windows
# ---------------------------------------------------------------------------------------------------------------------------
W = 16    # planes per window: 3 + 2 ThreePoint steps leave W - 10 valid planes inside the grid, W - 5 at its frame


def embed_work_window(cfg, work, zp0, zc):
    """the padded planes [zp0, zp0 + zc) of an array whose WORK planes [max(zp0 - e, 0), ...) are `work` (nx, ny, .): zero frame"""
    e = cfg.ext
    px, py, pz = cfg.padded_shape
    out = np.zeros((px, py, zc))
    lo, hi = max(zp0, e), min(zp0 + zc, pz - e)            # padded planes that are work planes
    out[e:-e, e:-e, lo - zp0:hi - zp0] = work
    return out


def window_check(wa, wo, n, dtype, potential, dn, dt, mass, sig, windows=None, steps=(3, 2), check_ab=True):
    """V (a, b) and phi after sum(steps) ground-state steps from the Boolean start on z-windows of an n^3 ThreePoint grid, cell by
    cell against the oracle: fp64 bit exact; fp32 storage bit exact against the oracle with V and every step's result rounded to
    float.  windows: first padded planes (None: bottom, middle, top)."""
    wo.set_threads(min(16, len(os.sched_getaffinity(0))))
    e = 1
    f32 = dtype != "f64"
    cfg = wo.Config(n, n, n, ext=e, potential="NoPotential" if potential == "file" else potential, dn=dn, dt=dt, mass=mass, sig=sig)
    par = wa.Params(n, n, n, dn=dn, dt=dt, mass=mass, sig=sig, dtype=dtype, max_states=1)
    pz = n + 2 * e
    if windows is None:
        windows = [0, pz // 2 - W // 2, pz - W]
    src = file_source() if potential == "file" else None
    total = sum(steps)
    with wa.Context(par) as ctx:
        set_real_potential(ctx, potential)
        ctx.set_initial_condition("Boolean")
        want = {}
        for zp0 in windows:
            if src is not None:   # fill_data -> trilerp_resize onto the work area with the padded size as basis (input.rs:156-173)
                lo, hi = max(zp0, e), min(zp0 + W, pz - e)
                v = embed_work_window(cfg, wo.trilerp_resize_zwindow(src, (n, n), lo - e, hi - lo, (pz, pz, pz)), zp0, W)
            else:
                v = wo.potential_generate_zwindow(cfg, zp0, W)
            if f32:
                v = v.astype(np.float32).astype(np.float64)       # what a float array holds
            got_v = ctx.download_window("v", zp0, W)
            assert np.array_equal(got_v, v), f"V differs on padded planes [{zp0}, {zp0 + W})"
            a, b = wo.ab_n(dt, v)                                  # the kernels form a, b from the stored V, in fp64
            if check_ab and not f32:
                assert np.array_equal(ctx.download_window("a", zp0, W), a) and np.array_equal(ctx.download_window("b", zp0, W), b)
            phi = wo.initial_condition_zwindow(cfg, "Boolean", zp0, W)
            assert np.array_equal(ctx.download_window("phi", zp0, W), phi)
            lo, hi = wo.evolve_zwindow(cfg, zp0, a, b, phi, total, storage=np.float32 if f32 else None)
            assert hi - lo >= W - 2 * total
            want[zp0] = (lo, hi, phi)
        for s in steps:
            ctx.evolve(0, s)
        name = ctx.stencil_kernel_name()
        worst = 0
        for zp0, (lo, hi, phi) in want.items():
            got = ctx.download_window("phi", zp0, W)
            assert np.any(got[:, :, lo:hi] != 0.0)
            bad = got[:, :, lo:hi] != phi[:, :, lo:hi]
            if bad.any():
                i, j, k = np.argwhere(bad)[0]
                worst = max(worst, int(bad.sum()))
                raise AssertionError(f"{int(bad.sum())} cells of planes [{zp0 + lo}, {zp0 + hi}) differ from the oracle; first at "
                                     f"({i}, {j}, {zp0 + lo + k}): {got[i, j, lo + k]!r} against {phi[i, j, lo + k]!r}")
        return name


@pytest.mark.parametrize("n", [96])
def test_window_oracle_small_grid_agrees_with_full_arrays(wa_mod, n):
    """the window machinery itself, where the whole array still fits: windows of the oracle equal slices of its full arrays
    (tests/test_oracle_windows.py on the CPU), and here the engine's window download equals slices of its full download"""
    from oracle import wafer_oracle as wo
    par = wa_mod.Params(n, n - 8, n + 5, dn=0.02, dt=8e-5, mass=2.35, sig=0.223, max_states=1)
    with wa_mod.Context(par) as ctx:
        ctx.set_potential("SimpleCornell")
        ctx.set_initial_condition("Boolean")
        ctx.evolve(0, 5)
        full, v = ctx.download_phi(), ctx.download_array("v")
        for zp0, zc in ((0, 7), (40, 16), (n + 7 - 9, 9)):
            assert np.array_equal(ctx.download_window("phi", zp0, zc), full[:, :, zp0:zp0 + zc])
            assert np.array_equal(ctx.download_window("v", zp0, zc), v[:, :, zp0:zp0 + zc])
        with pytest.raises(wa_mod.WaferError):
            ctx.download_window("phi", n + 7 - 3, 4)


@pytest.mark.parametrize("dtype,potential", [("f64", "SimpleCornell"), ("f32", "file"), ("f64", "file"), ("f32", "SimpleCornell")])
def test_windows_against_the_oracle_at_256_cubed(wa_mod, dtype, potential):
    """the full-size window tests at a size that takes a second: the same code path, every dtype / potential combination"""
    from oracle import wafer_oracle as wo
    name = window_check(wa_mod, wo, 256, dtype, potential, 0.02, 8e-5, 2.35 if potential != "file" else 1.0, 0.223)
    assert name == "wafer_k_step3_fused"


def test_config4_1024_cubed_simplecornell_windows_bit_exact(wa_mod):
    """BASELINE config #4 at FULL size on its REAL potential (SimpleCornell, m = 2.35, sig = 0.223, dn = 0.02,
    dt = 8e-5), one device: V, a, b and phi after 3 + 2 steps (one three-step pass + the two-step remainder) bit exact against
    the windowed oracle at z = 0, the middle (the potential's centre and its r < dn branch) and the top; every x / y edge is in
    the windows (whole planes)."""
    if free_gib() < 70:
        pytest.skip("needs 70 GiB of free device memory")
    from oracle import wafer_oracle as wo
    name = window_check(wa_mod, wo, 1024, "f64", "SimpleCornell", 0.02, 8e-5, 2.35, 0.223)
    assert name == "wafer_k_step3_fused"


def planes_by_element_offset(wa, n, dtype):
    """first padded planes of three windows of an n^3 ThreePoint array: element offsets below 2^32, straddling 2^32, beyond 2^33
    (the last where the array is that long, else the top of the grid)"""
    par = wa.Params(n, n, n, dn=0.01, dt=2e-5, dtype=dtype, max_states=1)
    e = 1
    # geometry without a context: wafer_geom.h (plane stride = (py + 2 gy) * pitch; gz = 3 ext guard planes precede plane 0)
    esz = 8 if dtype == "f64" else 4
    align = 128 // esz
    tile = 8 * align
    pitch = -(-((align - e) + e + -(-n // tile) * tile + 2 * e) // align) * align
    plane = (n + 2 * e + 2 * (16 + 3 * e)) * pitch
    gz = 3 * e
    pz = n + 2 * e
    at = lambda off: off // plane - gz                   # the padded plane that holds element `off` of the allocation
    out = [0]
    if at(2 ** 32) + W // 2 < pz:
        out.append(max(e, at(2 ** 32) - W // 2))         # planes on both sides of element 2^32
    out.append(pz - W if at(2 ** 33) + W >= pz else at(2 ** 33) + 3)
    return out, plane


def test_config5_2048_cubed_file_potential_fp32_windows(wa_mod):
    """BASELINE config #5 as written, at FULL size on one device: a 64^3 user potential through the device's trilinear
    resampler (wafer_set_potential_resampled = input::fill_data -> trilerp_resize, input.rs:149-176, 667-716) onto 2048^3 with
    fp32 storage.  V on planes whose element offsets are < 2^32, straddle 2^32 and exceed 2^33 equals (float) of the oracle's
    trilerp_resize cell by cell, and phi after three steps (one three-step pass) from the Boolean start equals the windowed
    oracle's with every step rounded to float -- bit for bit."""
    if free_gib() < 200:
        pytest.skip("needs 200 GiB of free device memory")
    from oracle import wafer_oracle as wo
    windows, plane = planes_by_element_offset(wa_mod, 2048, "f32")
    assert (windows[1] + 3 + 3) * plane < 2 ** 32 < (windows[1] + 3 + W) * plane and (windows[2] + 3) * plane > 2 ** 33
    name = window_check(wa_mod, wo, 2048, "f32", "file", 0.01, 2e-5, 1.0, 0.6, windows=windows + [2050 - W], steps=(3,))
    assert name == "wafer_k_step3_fused"


def test_config4_grid_1024_cubed_fp64(wa_mod):
    """1024^3 fp64 (config #4's grid, here on ONE device): 1.1e9 cells, 8.9 GB per array"""
    if free_gib() < 60:
        pytest.skip("needs 60 GiB of free device memory")
    ms = run_case(wa_mod, 1024, "f64", 0.02, 8e-5, 2.35, "SimpleCornell", sig=0.223)
    assert ms < 20.0


def test_config5_grid_2048_cubed_fp32(wa_mod):
    """2048^3 with fp32 storage (config #5's grid on ONE device): 8.6e9 cells, 36 GB per array,
    element offsets beyond 2^33"""
    if free_gib() < 200:
        pytest.skip("needs 200 GiB of free device memory")
    ms = run_case(wa_mod, 2048, "f32", 0.01, 2e-5, 1.0, "file")
    assert ms < 200.0


def test_config5_grid_2048_cubed_f32fast(wa_mod):
    """the same grid with fp32 arithmetic in the stencil steps as well (`f32fast`, the three-step kernel on 256 x 16 tiles):
    the closed forms hold with every product rounded to fp32"""
    if free_gib() < 200:
        pytest.skip("needs 200 GiB of free device memory")
    ms = run_case(wa_mod, 2048, "f32fast", 0.01, 2e-5, 1.0, "file")
    assert ms < 200.0


def test_config5_grid_2048_cubed_fp64_cross_check_size(wa_mod):
    """2048^3 in fp64 on ONE device (config #5's fp64 cross-check size: 71 GB per array).  It fits
    because the default kernels form a, b from V in registers and the stored a, b arrays are only
    allocated on demand: phi x 2 + V = 214 GB of the 288 GB."""
    if free_gib() < 235:
        pytest.skip("needs 235 GiB of free device memory")
    ms = run_case(wa_mod, 2048, "f64", 0.01, 2e-5, 1.0, "file")
    assert ms < 200.0


@pytest.fixture(scope="module")
def wa_mod():
    import wafer_amd
    wafer_amd.load_library()
    return wafer_amd
