"""BASELINE configs #4 and #5 at their FULL single-device sizes, checked against closed forms
that need no host array: 1024^3 fp64 (43 GB of device arrays) and 2048^3 with fp32 storage
(182 GB -- skipped unless the GPU has that much free).  More than 2^32 elements per array, so
every index in the kernels has to be 64-bit clean.

Boolean initial condition (config.rs:676-683: 1 where all three PADDED indices are odd) and no
potential (a = b = 1): with k = dt / (2 dn^2 m) one step (grid.rs:568-592) gives, exactly,
    1 - 6k                on the N^3/8 cells that held a 1,
    2k  (k next to the far frame: that neighbour is a frame zero)   on cells with ONE even index,
    0                     elsewhere,
so  sum phi'^2 = (N/2)^3 fl(1-6k)^2 + 3 (N/2)^2 [(N/2-1) fl(2k)^2 + fl(k)^2],  fl = rounding to
the storage type.  The sums before the step are exact integers."""
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


def run_case(wa, n, dtype, dn, dt, mass):
    """(the single-step kernel run here is the LDS one with a, b from V: no stored a, b arrays)"""
    store = np.float64 if dtype == "f64" else np.float32
    k, n0, n1, r2, e0 = closed_forms(n, dn, dt, mass, store, arith32=dtype == "f32fast")
    par = wa.Params(n, n, n, dn=dn, dt=dt, mass=mass, dtype=dtype, max_states=1)
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
        # the default path from the same start equals two single steps (two steps cannot fill a three-step pass: the
        # fused kernels' own remainder handling runs here)
        ctx.set_initial_condition("Boolean")
        ctx.evolve(0, 1)
        ctx.evolve(0, 1)
        two_single = ctx.norm2()
        ctx.set_stencil_variant(-1)
        ctx.set_initial_condition("Boolean")
        ctx.evolve(0, 2)
        assert ctx.stencil_kernel_name() == "wafer_k_step3_fused"    # every dtype (fp32 storage with fp64 arithmetic since round 5)
        assert ctx.norm2() == two_single
        # ... and three fused steps (fp64 ThreePoint: one pass of the three-step kernel) equal three single steps
        ctx.set_stencil_variant(1)
        ctx.set_initial_condition("Boolean")
        ctx.evolve(0, 3)
        three_single = ctx.norm2()
        ctx.set_stencil_variant(-1)
        ctx.set_initial_condition("Boolean")
        ctx.evolve(0, 3)
        assert ctx.norm2() == three_single
        ctx.set_initial_condition("Boolean")
        ctx.evolve(0, 2)
        # linearity: twice the wavefunction, four times the norm, bit for bit (powers of two)
        ctx.normalise(0.25)                                             # phi / sqrt(1/4) = 2 phi
        assert ctx.norm2() == 4.0 * two_single
        ctx.evolve(0, 6)
        ms, steps = ctx.last_evolve_ms()
        assert np.isfinite(ctx.norm2()) and steps == 6
        return ms / steps


def test_config4_grid_1024_cubed_fp64(wa_mod):
    """1024^3 fp64 (config #4's grid, here on ONE device): 1.1e9 cells, 8.9 GB per array"""
    if free_gib() < 60:
        pytest.skip("needs 60 GiB of free device memory")
    ms = run_case(wa_mod, 1024, "f64", 0.02, 8e-5, 2.35)
    assert ms < 20.0


def test_config5_grid_2048_cubed_fp32(wa_mod):
    """2048^3 with fp32 storage (config #5's grid on ONE device): 8.6e9 cells, 36 GB per array,
    element offsets beyond 2^33"""
    if free_gib() < 200:
        pytest.skip("needs 200 GiB of free device memory")
    ms = run_case(wa_mod, 2048, "f32", 0.01, 2e-5, 1.0)
    assert ms < 200.0


def test_config5_grid_2048_cubed_f32fast(wa_mod):
    """the same grid with fp32 arithmetic in the stencil steps as well (`f32fast`, the three-step kernel on 256 x 16 tiles):
    the closed forms hold with every product rounded to fp32"""
    if free_gib() < 200:
        pytest.skip("needs 200 GiB of free device memory")
    ms = run_case(wa_mod, 2048, "f32fast", 0.01, 2e-5, 1.0)
    assert ms < 200.0


def test_config5_grid_2048_cubed_fp64_cross_check_size(wa_mod):
    """2048^3 in fp64 on ONE device (config #5's fp64 cross-check size: 71 GB per array).  It fits
    because the default kernels form a, b from V in registers and the stored a, b arrays are only
    allocated on demand: phi x 2 + V = 214 GB of the 288 GB."""
    if free_gib() < 235:
        pytest.skip("needs 235 GiB of free device memory")
    ms = run_case(wa_mod, 2048, "f64", 0.01, 2e-5, 1.0)
    assert ms < 200.0


@pytest.fixture(scope="module")
def wa_mod():
    import wafer_amd
    wafer_amd.load_library()
    return wafer_amd
