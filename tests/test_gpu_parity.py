"""Parity of the HIP engine (called through the C ABI) with the CPU oracle on
the same inputs.  Bars: fp64 stencil steps, normalise, a/b,
algebraic potentials and layout conversion are BIT-EXACT; global sums agree to
rel 1e-12 (the reference's own rayon sums are order-nondeterministic)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests.gpu_common import make_pair, random_phi, ulp_diff  # noqa: E402

REL_SUM = 1e-12  # tolerance on global reductions, fp64


@pytest.fixture(scope="module")
def wo():
    from oracle import wafer_oracle
    wafer_oracle.build()
    return wafer_oracle


@pytest.fixture(scope="module")
def wa():
    import wafer_amd
    wafer_amd.load_library()
    return wafer_amd


SHAPES = [(17, 17, 17), (64, 64, 64), (65, 33, 20), (130, 9, 7), (3, 2, 5), (1, 1, 1)]


# ---------------------------------------------------------------- layout / setup
@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("ext", [1, 2, 3])
def test_upload_download_roundtrip(wa, shape, ext):
    cfg, par = make_pair(shape, ext=ext)
    phi = np.random.default_rng(1).standard_normal(cfg.padded_shape)
    with wa.Context(par) as ctx:
        ctx.upload_phi(phi)
        assert np.array_equal(ctx.download_phi(), phi)


ALGEBRAIC = ["NoPotential", "Cube", "QuadWell", "Coulomb", "ComplexCoulomb", "ElipticalCoulomb",
             "SimpleCornell", "Harmonic", "ComplexHarmonic", "Dodecahedron"]


@pytest.mark.parametrize("pot", ALGEBRAIC + ["Periodic", "FullCornell"])
@pytest.mark.parametrize("shape,ext", [((20, 14, 18), 1), ((9, 12, 16), 2), ((16, 16, 16), 3)])
def test_builtin_potentials_and_ab(wo, wa, pot, shape, ext):
    """potential.rs:46-62, 101-110, 188-319; z is a special axis for QuadWell /
    ElipticalCoulomb / FullCornell, hence the anisotropic shapes"""
    cfg, par = make_pair(shape, ext=ext, potential=pot, dn=0.13, dt=0.003, mass=1.7, sig=0.223)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    kind, scalar, arr = wo.potential_sub(cfg)
    with wa.Context(par) as ctx:
        ctx.set_potential(pot)
        gv, ga, gb = ctx.download_array("v"), ctx.download_array("a"), ctx.download_array("b")
        if pot in ALGEBRAIC:  # + - * / sqrt only: bit exact
            assert np.array_equal(gv, v)
            assert np.array_equal(ga, a) and np.array_equal(gb, b)
        else:  # sin / exp come from different libms: a few ulp
            assert np.allclose(gv, v, rtol=1e-14, atol=1e-15, equal_nan=True)
            assert np.allclose(gb, b, rtol=1e-14, equal_nan=True)
            assert np.allclose(ga, a, rtol=1e-14, equal_nan=True)
        gkind, gscalar = ctx.potsub()
        assert gkind == kind and gscalar == scalar
        if kind == 2:
            assert np.allclose(ctx.download_array("potsub"), arr, rtol=1e-14, equal_nan=True)


def test_host_potential_upload(wo, wa):
    """FromFile/FromScript path: V uploaded in the reference layout, a/b derived on device"""
    cfg, par = make_pair((12, 10, 14), ext=2, dt=0.002)
    v = np.random.default_rng(5).standard_normal(cfg.padded_shape)
    a, b = wo.ab(cfg, v)
    potsub = np.random.default_rng(6).standard_normal(cfg.work_shape)
    with wa.Context(par) as ctx:
        ctx.set_potential_host(v, 2, 0.0, potsub)
        assert np.array_equal(ctx.download_array("v"), v)
        assert np.array_equal(ctx.download_array("a"), a)
        assert np.array_equal(ctx.download_array("b"), b)
        assert np.array_equal(ctx.download_array("potsub"), potsub)
        with pytest.raises(wa.WaferError):
            ctx.set_potential("FromFile")  # ErrorKind::PotentialNotAvailable


@pytest.mark.parametrize("ic", ["Boolean", "Constant", "Gaussian", "Coulomb"])
@pytest.mark.parametrize("ext", [1, 3])
def test_initial_conditions(wo, wa, ic, ext):
    cfg, par = make_pair((11, 14, 9), ext=ext, mass=0.8, sig=0.7)
    want = wo.initial_condition(cfg, ic, seed=42)
    with wa.Context(par) as ctx:
        ctx.set_initial_condition(ic, seed=42)
        got = ctx.download_phi()
    if ic in ("Boolean", "Constant"):
        assert np.array_equal(got, want)
    else:
        assert np.allclose(got, want, rtol=1e-12, atol=1e-13, equal_nan=True)


def test_trilinear_resample(wo, wa, ref_vectors):
    """input.rs:667-716 on the device: the reference's 64-value table (its unit test's
    basis = the view's dims) and the production basis (padded target size), bit exact"""
    g = ref_vectors["interpolation"]
    src = np.array(g["source"]).reshape(g["source_shape"])
    n = g["target_shape"]
    with wa.Context(wa.Params(*n, dn=0.1, dt=1e-3)) as ctx:
        ctx.upload_phi_resampled(src, basis=n)
        got = ctx.download_phi()
        assert np.array_equal(got[1:-1, 1:-1, 1:-1].ravel(), np.array(g["expected"]))
        assert got.sum() == got[1:-1, 1:-1, 1:-1].sum()          # frame untouched
    rng = np.random.default_rng(1)
    src = rng.standard_normal((7, 5, 9))
    for ext, shape in ((1, (20, 13, 33)), (3, (9, 30, 12))):
        with wa.Context(wa.Params(*shape, dn=0.1, dt=1e-3, central_difference=ext)) as ctx:
            ctx.upload_phi_resampled(src)                         # basis = padded target size
            want = wo.trilerp_resize(src, shape, basis=tuple(s + 2 * ext for s in shape))
            assert np.array_equal(ctx.download_phi()[ext:-ext, ext:-ext, ext:-ext], want)
            ctx.set_potential_resampled(src)
            v = ctx.download_array("v")
            assert np.array_equal(v[ext:-ext, ext:-ext, ext:-ext], want)
            assert np.array_equal(ctx.download_array("b"), 1. / (1. + 1e-3 * v / 2.))


# ---------------------------------------------------------------- the stencil step
@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("ext", [1, 2, 3])
@pytest.mark.parametrize("shape", SHAPES)
def test_ground_state_steps_bit_exact(wo, wa, shape, ext, variant):
    """grid.rs:544-687 with wnum = 0: every cell of phi after 1 and after 10 steps"""
    cfg, par = make_pair(shape, ext=ext, potential="Harmonic", dn=0.2, dt=0.004, mass=1.0)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    phi = random_phi(cfg, seed=11)
    with wa.Context(par) as ctx:
        ctx.set_stencil_variant(variant)
        ctx.set_potential("Harmonic")
        ctx.upload_phi(phi)
        done = 0
        for steps in (1, 9):
            wo.evolve(cfg, 0, a, b, phi, [], steps)
            ctx.evolve(0, steps)
            done += steps
            got = ctx.download_phi()
            assert ulp_diff(got, phi) == 0, f"after {done} steps"


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("zchunk", ["1", "3", "1000"])
def test_zchunk_independence(wo, wa, variant, zchunk, monkeypatch):
    """the z-chunking of a launch never changes a bit"""
    monkeypatch.setenv("WAFER_ZCHUNK", zchunk)
    cfg, par = make_pair((40, 21, 13), ext=2, potential="Coulomb", dn=0.1, dt=0.002)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    phi = random_phi(cfg, seed=2)
    with wa.Context(par) as ctx:
        ctx.set_stencil_variant(variant)
        ctx.set_potential("Coulomb")
        ctx.upload_phi(phi)
        ctx.evolve(0, 5)
        wo.evolve(cfg, 0, a, b, phi, [], 5)
        assert ulp_diff(ctx.download_phi(), phi) == 0
        obs, want = ctx.observables(), wo.observables(cfg, v, phi)
        for k in want:
            assert obs[k] == pytest.approx(want[k], rel=REL_SUM, abs=1e-300)


@pytest.mark.parametrize("opts", [dict(WAFER_ABV="1"), dict(WAFER_ABV="1", WAFER_NT="0"), dict(WAFER_NT="0"),
                                  dict(WAFER_LDS_RY="4"), dict(WAFER_LDS_RY="4", WAFER_ABV="1"),
                                  dict(WAFER_XCD_SWIZZLE="0"), dict(WAFER_TARGET_BLOCKS="7")])
@pytest.mark.parametrize("ext", [1, 2, 3])
def test_lds_kernel_options_bit_exact(wo, wa, ext, opts, monkeypatch):
    """every tuning knob of the LDS kernel (a/b formed from V in registers,
    non-temporal streams, tile height, XCD map, launch size) leaves every bit alone"""
    for k, v in opts.items():
        monkeypatch.setenv(k, v)
    cfg, par = make_pair((150, 37, 29), ext=ext, potential="SimpleCornell", dn=0.1, dt=0.002, mass=2.35, sig=0.223)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    phi = random_phi(cfg, seed=4)
    with wa.Context(par) as ctx:
        ctx.set_stencil_variant(1)
        ctx.set_potential("SimpleCornell")
        ctx.upload_phi(phi)
        ctx.evolve(0, 7)
        wo.evolve(cfg, 0, a, b, phi, [], 7)
        assert ulp_diff(ctx.download_phi(), phi) == 0


@pytest.mark.parametrize("steps", [2, 7, 10])
@pytest.mark.parametrize("ext", [1, 2, 3])
@pytest.mark.parametrize("shape", SHAPES + [(150, 37, 29), (257, 20, 11)])
def test_fused_two_step_kernel_bit_exact(wo, wa, shape, ext, steps):
    """variant 2 (two time steps per pass over HBM, a/b formed from V): the same
    bits as `steps` single reference steps -- even and odd counts, ragged tiles,
    grids smaller than a tile, every frame cell still exactly zero"""
    cfg, par = make_pair(shape, ext=ext, potential="Coulomb", dn=0.2, dt=0.004, mass=1.0)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    phi = random_phi(cfg, seed=12)
    with wa.Context(par) as ctx:
        ctx.set_stencil_variant(2)
        ctx.set_potential("Coulomb")
        ctx.upload_phi(phi)
        ctx.evolve(0, steps)
        wo.evolve(cfg, 0, a, b, phi, [], steps)
        assert ulp_diff(ctx.download_phi(), phi) == 0


@pytest.mark.parametrize("wide", ["1", "0"])
@pytest.mark.parametrize("zchunk", ["", "1", "3", "7"])
@pytest.mark.parametrize("shape,steps", [((150, 37, 29), 6), ((257, 20, 11), 5), ((128, 16, 9), 4), ((129, 33, 40), 7), ((300, 50, 7), 2),
                                         ((5, 4, 3), 4), ((1, 1, 1), 3), ((64, 70, 2), 8), ((256, 32, 21), 6)])
def test_five_point_two_step_kernels_bit_exact(wo, wa, shape, steps, zchunk, wide, monkeypatch):
    """FivePoint, two steps per pass: the 128 x 16-tile kernel in the three-step kernel's structure (wafer_stencil_fused2w.hip.h:
    eight even waves, round 5) and the kernel with dedicated helper waves it replaces (WAFER_F2_WIDE=0) against the oracle, every
    cell's bits (grid.rs:593-624): ragged tiles, grids smaller than a tile and of whole tiles, one-plane / odd z-chunks, even and
    odd step counts, every frame cell still exactly zero"""
    monkeypatch.setenv("WAFER_F2_WIDE", wide)
    if zchunk:
        monkeypatch.setenv("WAFER_ZCHUNK", zchunk)
    cfg, par = make_pair(shape, ext=2, potential="Coulomb", dn=0.2, dt=0.004, mass=1.0)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    phi = random_phi(cfg, seed=12)
    with wa.Context(par) as ctx:
        ctx.set_stencil_variant(2)
        assert ctx.stencil_kernel_name() == ("wafer_k_step2_wide" if wide == "1" else "wafer_k_step2_fused") and ctx.steps_per_launch() == 2
        ctx.set_potential("Coulomb")
        ctx.upload_phi(phi)
        ctx.evolve(0, steps)
        wo.evolve(cfg, 0, a, b, phi, [], steps)
        got = ctx.download_phi()
        assert ulp_diff(got, phi) == 0
        assert not got[:2].any() and not got[-2:].any() and not got[:, :2].any() and not got[:, :, -2:].any()


@pytest.mark.parametrize("dtype", ["f32", "f32fast"])
@pytest.mark.parametrize("shape,steps", [((200, 37, 29), 8), ((300, 20, 18), 7), ((256, 32, 21), 6), ((512, 16, 9), 4)])
def test_five_point_two_step_kernel_fp32_storage_gives_the_single_step_kernels_bits(wa, dtype, shape, steps, monkeypatch):
    """the same kernel on fp32 storage -- with fp64 arithmetic (float in HBM, double in the CU, every level rounded to float) and
    with fp32 arithmetic (256 x 16 tiles) -- against the single-step kernel on the same storage: every cell's bits"""
    out = {}
    for variant, zchunk in ((2, ""), (2, "5"), (1, "")):
        if zchunk:
            monkeypatch.setenv("WAFER_ZCHUNK", zchunk)
        else:
            monkeypatch.delenv("WAFER_ZCHUNK", raising=False)
        par = wa.Params(*shape, dn=0.2, dt=0.004, mass=1.3, central_difference=2, dtype=dtype)
        with wa.Context(par) as ctx:
            ctx.set_stencil_variant(variant)
            ctx.set_potential("Coulomb")
            ctx.set_initial_condition("Gaussian", seed=5)
            ctx.evolve(0, steps)
            out[(variant, zchunk)] = ctx.download_phi()
    assert np.array_equal(out[(2, "")], out[(1, "")])
    assert np.array_equal(out[(2, "5")], out[(1, "")])


@pytest.mark.parametrize("steps", [3, 7, 11, 12])
@pytest.mark.parametrize("shape", SHAPES + [(150, 37, 29), (257, 20, 11), (128, 16, 9), (129, 33, 40), (300, 50, 7)])
def test_fused_three_step_kernel_bit_exact(wo, wa, shape, steps, monkeypatch):
    """variant 3 (THREE ThreePoint steps per pass over HBM, wafer_stencil_fused3.hip.h): the same bits as
    `steps` single reference steps -- counts that leave a two-step and a single-step remainder, ragged
    tiles, grids smaller than a tile (forced onto the kernel), every frame cell still exactly zero"""
    monkeypatch.setenv("WAFER_FUSE3_MIN_NY", "1")
    cfg, par = make_pair(shape, ext=1, potential="Coulomb", dn=0.2, dt=0.004, mass=1.0)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    phi = random_phi(cfg, seed=12)
    with wa.Context(par) as ctx:
        ctx.set_stencil_variant(3)
        assert ctx.stencil_kernel_name() == "wafer_k_step3_fused" and ctx.steps_per_launch() == 3
        ctx.set_potential("Coulomb")
        ctx.upload_phi(phi)
        ctx.evolve(0, steps)
        wo.evolve(cfg, 0, a, b, phi, [], steps)
        got = ctx.download_phi()
        assert ulp_diff(got, phi) == 0
        assert not got[0].any() and not got[-1].any() and not got[:, 0].any() and not got[:, :, -1].any()


@pytest.mark.parametrize("zchunk", ["1", "2", "3", "5", "1000"])
def test_fused3_zchunk_independence(wo, wa, zchunk, monkeypatch):
    monkeypatch.setenv("WAFER_ZCHUNK", zchunk)
    monkeypatch.setenv("WAFER_FUSE3_MIN_NY", "1")
    cfg, par = make_pair((70, 21, 13), ext=1, potential="Harmonic", dn=0.1, dt=0.002)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    phi = random_phi(cfg, seed=2)
    with wa.Context(par) as ctx:
        ctx.set_stencil_variant(3)
        ctx.set_potential("Harmonic")
        ctx.upload_phi(phi)
        ctx.evolve(0, 6)
        wo.evolve(cfg, 0, a, b, phi, [], 6)
        assert ulp_diff(ctx.download_phi(), phi) == 0


@pytest.mark.parametrize("sched", ["0", "1", "down"])
@pytest.mark.parametrize("xs", ["0", "1"])
@pytest.mark.parametrize("zchunk", ["1", "2", "5", "1000"])
@pytest.mark.parametrize("shape", [(128, 16, 5), (256, 32, 11), (128, 48, 23)])
def test_fused3_exact_store_count_variant(wo, wa, shape, zchunk, xs, sched, monkeypatch):
    """grids made of whole 128 x 16 tiles run the three-step kernel that issues two stores in EVERY plane iteration (while the
    pipeline fills they go to the column's first plane, which the first real store overwrites; WAFER_F3_XS=0: the stores
    inside conditions): the oracle's bits either way, for every z-chunking, marching up and (WAFER_F3_SCHED=1: the
    two-halves schedule on an undecomposed grid) down"""
    monkeypatch.setenv("WAFER_ZCHUNK", zchunk)
    monkeypatch.setenv("WAFER_F3_XS", xs)
    # "down": the plain schedule marching every column downwards -- the single-direction instantiation of the kernel that the
    # whole-column peer-store passes use every other pass (ring queues only march up: the downward copy keeps the shifts)
    monkeypatch.setenv("WAFER_F3_SCHED", "0" if sched == "down" else sched)
    monkeypatch.setenv("WAFER_F3_PLAIN_DOWN", "1" if sched == "down" else "0")
    monkeypatch.setenv("WAFER_FUSE3_MIN_NY", "1")
    cfg, par = make_pair(shape, ext=1, potential="Coulomb", dn=0.2, dt=0.004, mass=1.0)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    phi = random_phi(cfg, seed=31)
    with wa.Context(par) as ctx:
        ctx.set_stencil_variant(3)
        ctx.set_potential("Coulomb")
        ctx.upload_phi(phi)
        ctx.evolve(0, 9)
        wo.evolve(cfg, 0, a, b, phi, [], 9)
        got = ctx.download_phi()
        assert ulp_diff(got, phi) == 0
        assert not got[0].any() and not got[-1].any() and not got[:, 0].any() and not got[:, :, -1].any()


def test_fused3_thousand_steps_64cubed_and_default_dispatch(wo, wa, monkeypatch):
    """1000 steps (333 three-step passes + one single step) at 64^3 (forced onto the kernel) against the
    oracle; the three-step kernel is what a ThreePoint fp64 context of 1.5 M cells or more runs by default,
    the two-step kernel what small grids, FivePoint and slabs with two ghost planes run"""
    with wa.Context(wa.Params(64, 64, 64, dn=0.2, dt=8e-3)) as ctx:      # small: launch-bound, the two-step kernel is faster
        assert ctx.stencil_kernel_name() == "wafer_k_step2_fused"
    for shape in ((256, 256, 192), (128, 128, 128)):
        with wa.Context(wa.Params(*shape, dn=0.2, dt=8e-3, max_states=1)) as ctx:
            assert ctx.stencil_kernel_name() == "wafer_k_step3_fused" and ctx.steps_per_launch() == 3
    monkeypatch.setenv("WAFER_FUSE3_MIN_NY", "1")
    cfg, par = make_pair((64, 64, 64), ext=1, potential="Harmonic", dn=0.2, dt=8e-3, mass=1.0)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    phi = wo.initial_condition(cfg, "Boolean")
    wo.evolve(cfg, 0, a, b, phi, [], 1000)
    with wa.Context(par) as ctx:
        assert ctx.stencil_kernel_name() == "wafer_k_step3_fused" and ctx.steps_per_launch() == 3
        ctx.set_potential("Harmonic")
        ctx.set_initial_condition("Boolean")
        ctx.evolve(0, 1000)
        assert ulp_diff(ctx.download_phi(), phi) == 0
    for kw, name in ((dict(central_difference=2), "wafer_k_step2_wide"), (dict(z_begin=16, z_count=16, halo_depth=2), "wafer_k_step2_fused")):
        with wa.Context(wa.Params(64, 64, 64, dn=0.2, dt=8e-3, **kw)) as ctx:
            assert ctx.stencil_kernel_name() == name and ctx.steps_per_launch() == 2   # the kernel that is launched, not its family
    with wa.Context(wa.Params(64, 64, 64, dn=0.2, dt=8e-3, dtype="f32")) as ctx:   # fp32 storage, fp64 arithmetic: the three-step kernel too (round 5)
        assert ctx.stencil_kernel_name() == "wafer_k_step3_fused" and ctx.steps_per_launch() == 3


@pytest.mark.parametrize("zchunk", ["1", "2", "5", "1000"])
def test_fused_zchunk_independence(wo, wa, zchunk, monkeypatch):
    monkeypatch.setenv("WAFER_ZCHUNK", zchunk)
    cfg, par = make_pair((70, 21, 13), ext=1, potential="Harmonic", dn=0.1, dt=0.002)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    phi = random_phi(cfg, seed=2)
    with wa.Context(par) as ctx:
        ctx.set_stencil_variant(2)
        ctx.set_potential("Harmonic")
        ctx.upload_phi(phi)
        ctx.evolve(0, 6)
        wo.evolve(cfg, 0, a, b, phi, [], 6)
        assert ulp_diff(ctx.download_phi(), phi) == 0


def test_fused_thousand_steps_64cubed(wo, wa):
    cfg, par = make_pair((64, 64, 64), ext=1, potential="Harmonic", dn=0.2, dt=8e-3, mass=1.0)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    phi = wo.initial_condition(cfg, "Boolean")
    wo.evolve(cfg, 0, a, b, phi, [], 1000)
    with wa.Context(par) as ctx:
        ctx.set_stencil_variant(2)
        ctx.set_potential("Harmonic")
        ctx.set_initial_condition("Boolean")
        ctx.evolve(0, 1000)
        assert ulp_diff(ctx.download_phi(), phi) == 0


@pytest.mark.parametrize("variant", [1, 2])
def test_extreme_potential_takes_the_full_division(wo, wa, variant):
    """a, b are formed from V in registers with a shortened reciprocal that is exact only for
    2^-400 < |1 + dt V/2| < 2^400; the engine checks the potential and otherwise keeps the full
    fp64 division.  A potential with 1 + dt V/2 = 2^-500 and 2^+600 cells must still match the
    oracle bit for bit."""
    cfg, par = make_pair((34, 20, 18), ext=1, dn=0.2, dt=0.004)
    v = np.random.default_rng(3).standard_normal(cfg.padded_shape)
    v[5, 6, 7] = (2.0 ** -500 - 1.0) * 2 / cfg.dt      # 1 + dt V/2 = 2^-500 (after rounding: tiny)
    v[9, 3, 4] = (2.0 ** 600) * 2 / cfg.dt
    a, b = wo.ab(cfg, v)
    phi = random_phi(cfg, seed=21)
    with wa.Context(par) as ctx:
        ctx.set_stencil_variant(variant)
        ctx.set_potential_host(v)
        ctx.upload_phi(phi)
        ctx.evolve(0, 4)
        wo.evolve(cfg, 0, a, b, phi, [], 4)
        got = ctx.download_phi()
        assert np.array_equal(got, phi, equal_nan=True)


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("scale,exact", [(1e-280, True), (1.0, True), (1e150, True), (1e300, True),
                                         (1e-295, False), (1e-303, False), (1e-306, False), (4e-320, False)])
def test_division_by_the_invariant_denominator_over_the_exponent_range(wo, wa, variant, scale, exact):
    """wafer_div_invariant: the hoisted-reciprocal division is the IEEE quotient, bit for bit, wherever
    Markstein's theorem holds -- wavefunctions from 1e-280 to 1e300 -- and within one unit in the last
    place per step below (|b dt S| < 2^-960, where the remainders become subnormal): after 4 steps
    of a sign-alternating field the ABSOLUTE difference stays below 1e-319, a few thousand subnormal
    quanta, down to subnormal wavefunctions.  Zero, infinite and NaN operands:
    test_extreme_potential_takes_the_full_division."""
    cfg, par = make_pair((37, 22, 19), ext=1, dn=0.2, dt=0.004)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    phi = random_phi(cfg, seed=33) * scale
    with wa.Context(par) as ctx:
        ctx.set_stencil_variant(variant)
        ctx.set_potential("Harmonic")
        ctx.upload_phi(phi)
        ctx.evolve(0, 4)
        wo.evolve(cfg, 0, a, b, phi, [], 4)
        got = ctx.download_phi()
    assert np.isfinite(phi).all() and np.abs(phi).max() > 0
    if exact:
        assert ulp_diff(got, phi) == 0
    else:
        assert float(np.max(np.abs(got - phi))) <= 1e-319


@pytest.mark.parametrize("ext,variant", [(2, 2), (2, 1), (3, 1)])
@pytest.mark.parametrize("scale", [1e-280, 1.0, 1e290])
def test_power_of_two_coefficients_as_fused_multiply_adds_over_the_exponent_range(wo, wa, ext, variant, scale):
    """wafer_fma_pow2: the FivePoint / SevenPoint sums take t + 16 x (t + 2 x) as one fused multiply-add -- the product is exact, so
    these are the reference's bits from tiny to huge wavefunctions (up to 2^1019, where the separate product would overflow)"""
    cfg, par = make_pair((37, 22, 19), ext=ext, dn=0.2, dt=0.002)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    phi = random_phi(cfg, seed=34) * scale
    with wa.Context(par) as ctx:
        ctx.set_stencil_variant(variant)
        ctx.set_potential("Harmonic")
        ctx.upload_phi(phi)
        ctx.evolve(0, 4)
        wo.evolve(cfg, 0, a, b, phi, [], 4)
        got = ctx.download_phi()
    assert np.isfinite(phi).all() and np.abs(phi).max() > 0
    assert ulp_diff(got, phi) == 0


def test_hoisted_division_equals_the_ieee_division_on_2e10_operands(wa):
    """wafer_div_invariant against the device's own IEEE x / den, bit for bit: 2^31 random operands
    (uniform significand and sign, exponents 2^-959 ... 2^+960) for each of ten denominators -- the
    c dn^2 m of the configurations in use, norms, and awkward significands (all ones, 1 + ulp)."""
    dens = [2 * 0.05 ** 2 * 1.0, 2 * 0.02 ** 2 * 2.35, 24 * 0.2 ** 2 * 1.3, 360 * 0.01 ** 2 * 0.7, 1.0,
            float(np.nextafter(2.0, 0.0)), float(np.nextafter(1.0, 2.0)), 3.0, 0.9999999999999432, 1.7320508075688772e-3]
    with wa.Context(wa.Params(8, 8, 8, dn=0.2, dt=0.004)) as ctx:
        for i, den in enumerate(dens):
            assert ctx.div_check(den, 1 << 31, lo_exp=64, hi_exp=1983, seed=100 + i) == 0, den
        # the counter is live: below the theorem's range some quotients differ (by one ulp)
        assert ctx.div_check(dens[0], 1 << 24, lo_exp=1, hi_exp=40, seed=7) > 0


@pytest.mark.parametrize("shape,steps", [((256, 128, 400), 7), ((256, 192, 500), 9), ((128, 256, 1153), 6)])
def test_three_step_kernel_one_launch_per_round_of_cus(wo, wa, shape, steps, monkeypatch):
    """a layer of tiles that is whole rounds of CUs and columns longer than 384 planes (1024^3 on one GPU): the columns are cut and
    every round of CUs is a launch of its own (wafer_f3_by_rounds) -- here with 8 'CUs' (WAFER_TARGET_BLOCKS) and 16 tiles per layer;
    the same bits as the oracle, and as one launch over the uncut columns"""
    monkeypatch.setenv("WAFER_TARGET_BLOCKS", "8")
    cfg, par = make_pair(shape, ext=1, potential="Coulomb", dn=0.2, dt=0.004)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    phi0 = random_phi(cfg, seed=5)
    got = {}
    for rounds in ("-1", "0"):
        monkeypatch.setenv("WAFER_F3_ROUNDS", rounds)
        with wa.Context(par) as ctx:
            ctx.set_stencil_variant(3)
            ctx.set_potential("Coulomb")
            ctx.upload_phi(phi0)
            ctx.evolve(0, steps)
            assert ctx.stencil_kernel_name() == "wafer_k_step3_fused"
            got[rounds] = ctx.download_phi()
    phi = phi0.copy()
    wo.evolve(cfg, 0, a, b, phi, [], steps)
    assert ulp_diff(got["-1"], phi) == 0 and ulp_diff(got["0"], phi) == 0


@pytest.mark.parametrize("shape,steps", [((256, 128, 400), 6), ((128, 256, 777), 5)])
def test_five_point_kernel_one_launch_per_round_of_cus(wo, wa, shape, steps, monkeypatch):
    """the FivePoint two-step kernel under the same launch rule (it derives its tile from blockIdx: a round carries its offset into the
    whole schedule): 8 'CUs', 16 tiles per layer, columns cut to at most 384 planes; the oracle's bits, with the rule and without"""
    monkeypatch.setenv("WAFER_TARGET_BLOCKS", "8")
    cfg, par = make_pair(shape, ext=2, potential="Coulomb", dn=0.2, dt=0.002)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    phi0 = random_phi(cfg, seed=6)
    got = {}
    for rounds in ("-1", "0"):
        monkeypatch.setenv("WAFER_F3_ROUNDS", rounds)
        with wa.Context(par) as ctx:
            ctx.set_potential("Coulomb")
            ctx.upload_phi(phi0)
            ctx.evolve(0, steps)
            assert ctx.stencil_kernel_instance().startswith("wafer_k_step2_wide<double")
            got[rounds] = ctx.download_phi()
    phi = phi0.copy()
    wo.evolve(cfg, 0, a, b, phi, [], steps)
    assert ulp_diff(got["-1"], phi) == 0 and ulp_diff(got["0"], phi) == 0


DENS_NEEDING_A_MOVED_ZL = [0.007395769697490762, 0.1078657875904072, 0.20222586000144446]   # tests/test_div_plan.py


def candidate_operands(cand):
    """the plan's significands over the exponent range, both signs"""
    ops = [np.ldexp(cand, e) for e in (-52, -900, -400, 0, 300, 900)]
    ops = np.concatenate(ops)
    return np.concatenate([ops, -ops])


def test_planned_division_equals_the_ieee_division(wa):
    """the three-instruction x / (c dn^2 m) of the step kernels (wafer_div_invariant(x, WaferDen), planned by wafer_div_plan)
    against the device's IEEE division, bit for bit: 2^31 random operands per denominator, and the operands that matter --
    every significand whose quotient comes within 2^-50 ulp of a rounding boundary (the plan's candidates), over the exponent
    range and with both signs"""
    from wafer_amd import engine
    dens = [2 * 0.05 ** 2 * 1.0, 2 * 0.02 ** 2 * 2.35, 24 * 0.2 ** 2 * 1.3, 360 * 0.01 ** 2 * 0.7, 2 * 0.2 ** 2, 3.0,
            float(np.nextafter(2.0, 0.0)), float(np.nextafter(1.0, 2.0)), 1.7320508075688772e-3] + DENS_NEEDING_A_MOVED_ZL
    with wa.Context(wa.Params(8, 8, 8, dn=0.2, dt=0.004)) as ctx:
        mine = ctx.div_plan()
        assert mine.den == 2 * 0.2 * 0.2 * 1.0 and mine.checked == 1
        for i, den in enumerate(dens):
            plan, cand = engine.div_plan(den)
            assert plan.checked == 1, den
            ops = candidate_operands(cand) if cand.size else None
            assert ctx.div_planned_check(plan, 1 << 31, ops, seed=300 + i) == (0, 0), den
            # without the plan's verdict: the extra round, the same bits
            plan.checked = 0
            assert ctx.div_planned_check(plan, 1 << 28, ops, seed=400 + i) == (0, 0), den
        # below the range the plan speaks for (x zl subnormal) some quotients differ by an ulp: the counter is live
        plan, _ = engine.div_plan(dens[0])
        assert ctx.div_planned_check(plan, 1 << 24, None, lo_exp=1, hi_exp=40, seed=7)[0] > 0


def test_the_plan_catches_what_the_device_gets_wrong(wa):
    """divisors whose RN(1/den - zh) is NOT good enough: forced onto the three-instruction form with that zl the device does
    return a wrong last bit for some of the plan's candidates -- the operands the plan exists to try -- while 2^31 random
    operands show nothing; with the zl the plan settled on, or with the extra round, every one of them is right"""
    from wafer_amd import engine
    with wa.Context(wa.Params(8, 8, 8, dn=0.2, dt=0.004)) as ctx:
        for i, den in enumerate(DENS_NEEDING_A_MOVED_ZL):
            plan, cand = engine.div_plan(den)
            ops = candidate_operands(cand)
            assert plan.zl_shift != 0 and ctx.div_planned_check(plan, 0, ops) == (0, 0)
            unmoved = engine._DivPlan(den, plan.zh, float(np.nextafter(plan.zl, -np.inf if plan.zl_shift > 0 else np.inf)), 1, 0, 0, 0)
            assert abs(plan.zl_shift) == 1
            bad_random, bad_ops = ctx.div_planned_check(unmoved, 1 << 31, ops, seed=500 + i)
            assert bad_ops > 0 and bad_random == 0, (den, bad_random, bad_ops)
            unmoved.checked = 0
            assert ctx.div_planned_check(unmoved, 1 << 28, ops, seed=600 + i) == (0, 0)


def test_planned_fp32_division_on_every_float(wa):
    """WAFER_F32_FAST: the three-instruction fp32 division of the step kernels against the device's IEEE x / den on EVERY float of
    201 binades (all 2^23 significands, both signs: 6.7e9 operands per denominator), bit for bit; below 2^-100 (x zl subnormal) and
    without the plan's verdict the kernels' behaviour is the division itself"""
    from wafer_amd import engine
    with wa.Context(wa.Params(8, 8, 8, dn=0.2, dt=0.004, dtype="f32fast")) as ctx:
        for den in (2 * 0.05 ** 2, 2 * 0.02 ** 2 * 2.35, 24 * 0.2 ** 2 * 1.3, 360 * 0.01 ** 2 * 0.7, 2 * 0.2 ** 2, 3.0, 0.06):
            plan = engine.div_plan_f32(den)
            assert plan.checked == 1
            assert ctx.div_planned_check_f32(plan, 27, 227) == 0, den
            plan.checked = 0
            assert ctx.div_planned_check_f32(plan, 1, 254) == 0     # unchecked: x / den itself, everywhere
        plan = engine.div_plan_f32(0.06)
        assert plan.zl_shift != 0
        unmoved = engine._DivPlanF32(plan.den, plan.zh, float(np.nextafter(np.float32(plan.zl), np.float32(-np.inf if plan.zl_shift > 0 else np.inf))), 1, 0)
        assert ctx.div_planned_check_f32(unmoved, 27, 227) > 0      # the host's exhaustive check is not vacuous either


def test_planned_fp32_division_below_its_checked_range(wa):
    """the fp32 plan is checked for |x / den| >= 2^-100 only.  Below (biased exponents 1 .. 26 of x for denominators near 1:
    x zl subnormal) the three instructions may differ from the IEEE quotient in the last bit.  Measured here, on every float of those
    26 binades: how often, and that the range the kernels' parity contract names (27 .. 227) stays clean.  f32fast is a tolerance mode
    (DESIGN.md section 3); values below 8e-31 do not reach its bar."""
    from wafer_amd import engine
    out = {}
    with wa.Context(wa.Params(8, 8, 8, dn=0.2, dt=0.004, dtype="f32fast")) as ctx:
        for den in (2 * 0.05 ** 2, 2 * 0.2 ** 2, 3.0, 0.06):
            plan = engine.div_plan_f32(den)
            assert plan.checked == 1 and ctx.div_planned_check_f32(plan, 27, 227) == 0
            bad = ctx.div_planned_check_f32(plan, 1, 26)
            out[den] = bad / (26 * 2.0 ** 24)
            assert out[den] < 0.5            # (a last-bit difference on some operands, never most of them)
    print("fraction of floats below the checked range whose planned quotient differs in the last bit:", out)
    log = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(log):
        import json
        with open(os.path.join(log, "f32_division_below_range.json"), "w") as f:
            json.dump({str(k): v for k, v in out.items()}, f)


def grid_spacing_whose_divisor_needs_a_moved_zl(lead, mass):
    from wafer_amd import engine
    rng = np.random.default_rng(3)
    for _ in range(4000):
        dn = float(rng.uniform(0.05, 0.4))
        if engine.div_plan(lead * dn * dn * mass)[0].zl_shift != 0:
            return dn
    raise AssertionError("no such grid spacing found")


@pytest.mark.parametrize("mode", ["moved_zl", "unplanned"])
@pytest.mark.parametrize("ext,variant,steps", [(1, 3, 7), (1, 2, 5), (1, 1, 3), (2, 2, 5), (3, 1, 2)])
def test_step_kernels_with_an_awkward_divisor_and_without_the_plan(wo, wa, ext, variant, steps, mode, monkeypatch):
    """the step kernels against the oracle (which divides) where the plan has work to do: a grid spacing whose c dn^2 m needs its
    zl moved, and WAFER_FLAG_UNPLANNED_DIV (every division with the extra round -- the instantiations a divisor the plan cannot
    clear would run: VIR = false, b by the full division as well)"""
    monkeypatch.setenv("WAFER_FUSE3_MIN_NY", "1")
    lead = {1: 2.0, 2: 24.0, 3: 360.0}[ext]
    dn = grid_spacing_whose_divisor_needs_a_moved_zl(lead, 1.3) if mode == "moved_zl" else 0.2
    cfg, par = make_pair((150, 37, 29), ext=ext, potential="Coulomb", dn=dn, dt=dn * dn / 10, mass=1.3, unplanned_div=(mode == "unplanned"))
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    phi = random_phi(cfg, seed=21)
    with wa.Context(par) as ctx:
        plan = ctx.div_plan()
        assert plan.checked == (0 if mode == "unplanned" else 1) and (plan.zl_shift != 0) == (mode == "moved_zl")
        ctx.set_stencil_variant(variant)
        ctx.set_potential("Coulomb")
        ctx.upload_phi(phi)
        ctx.evolve(0, steps)
        if variant == 3 and ext == 1:
            assert ctx.stencil_kernel_instance().startswith("wafer_k_step3_fused<double, double, %s," % ("false" if mode == "unplanned" else "true"))
        obs = ctx.observables()
        wo.evolve(cfg, 0, a, b, phi, [], steps)
        assert ulp_diff(ctx.download_phi(), phi) == 0
        want = wo.observables(cfg, v, phi)
        for k in want:   # (the energy integrand divides by the same denominator)
            assert obs[k] == pytest.approx(want[k], rel=REL_SUM, abs=1e-300)


def test_evolve_zero_steps_takes_one(wo, wa):
    """grid.rs:682-685"""
    cfg, par = make_pair((8, 8, 8))
    phi = random_phi(cfg)
    with wa.Context(par) as ctx:
        ctx.set_potential("Harmonic")
        ctx.upload_phi(phi)
        ctx.evolve(0, 0)
        one = ctx.download_phi()
        ctx.upload_phi(phi)
        ctx.evolve(0, 1)
        assert np.array_equal(one, ctx.download_phi())
        assert not np.array_equal(one, phi)


def test_thousand_steps_64cubed(wo, wa):
    """BASELINE config #1 (64^3 harmonic, Boolean IC): a full screen_update block"""
    cfg, par = make_pair((64, 64, 64), ext=1, potential="Harmonic", dn=0.2, dt=8e-3, mass=1.0)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    phi = wo.initial_condition(cfg, "Boolean")
    wo.evolve(cfg, 0, a, b, phi, [], 1000)
    with wa.Context(par) as ctx:
        ctx.set_potential("Harmonic")
        ctx.set_initial_condition("Boolean")
        ctx.evolve(0, 1000)
        assert ulp_diff(ctx.download_phi(), phi) == 0
        obs, want = ctx.observables(), wo.observables(cfg, v, phi)
        for k in ("energy", "norm2", "r2"):
            assert obs[k] == pytest.approx(want[k], rel=REL_SUM)
        assert obs["v_infinity"] == 0.0 == want["v_infinity"]


# ---------------------------------------------------------------- reductions & helpers
def _rule(shape, rule):
    i, j, k = np.meshgrid(*[np.arange(s, dtype=np.float64) for s in shape], indexing="ij")
    return np.ascontiguousarray({"i+j+k": i + j + k, "-(i+j+k)": -i - j - k, "i*j*k": i * j * k}[rule])


def _embed(work, ext):
    out = np.zeros(tuple(s + 2 * ext for s in work.shape))
    out[ext:-ext, ext:-ext, ext:-ext] = work
    return out


def test_reference_unit_vectors_through_hip(wa, ref_vectors):
    """the reference's own known answers (grid.rs:721-799), computed by the HIP path"""
    g = ref_vectors["norm2"]  # 70070 on the work area of a (5,8,7) array
    e = g["ext"]
    work = tuple(s - 2 * e for s in g["shape"])
    with wa.Context(wa.Params(*work, dn=0.1, dt=1e-3)) as ctx:
        ctx.upload_phi(_rule(g["shape"], g["phi_rule"]))
        assert abs(ctx.norm2() - g["expected"]) < g["eps"]
    g = ref_vectors["wfn_normalise"]  # whole (3,2,5) array / sqrt(1.23): embed as a work area
    with wa.Context(wa.Params(*g["shape"], dn=0.1, dt=1e-3)) as ctx:
        ctx.upload_phi(_embed(_rule(g["shape"], g["phi_rule"]), 1))
        ctx.normalise(g["norm2"])
        got = ctx.download_phi()[1:-1, 1:-1, 1:-1]
        assert np.allclose(got, _rule(g["shape"], g["phi_rule"]) / g["expected_divisor"], atol=g["tol"])
        assert np.array_equal(got, _rule(g["shape"], g["phi_rule"]) / np.sqrt(g["norm2"]))
    g = ref_vectors["gram_schmidt"]  # ground = i+j+k, phi = -ground on 2^3
    with wa.Context(wa.Params(*g["shape"], dn=0.1, dt=1e-3)) as ctx:
        ctx.load_state(0, _embed(_rule(g["shape"], g["lower_rule"]), 1))
        ctx.upload_phi(_embed(_rule(g["shape"], g["phi_rule"]), 1))
        ctx.orthogonalise(1)
        got = ctx.download_phi()[1:-1, 1:-1, 1:-1].ravel()
        assert np.allclose(got, g["expected"], atol=g["tol"])
        assert np.array_equal(got, np.array(g["expected"]))


@pytest.mark.parametrize("ext", [1, 2, 3])
@pytest.mark.parametrize("pot", ["Harmonic", "SimpleCornell", "FullCornell"])
def test_observables(wo, wa, ext, pot):
    """grid.rs:303-445 incl. scalar and array pot_sub and the work-index r^2"""
    cfg, par = make_pair((23, 18, 31), ext=ext, potential=pot, dn=0.15, dt=0.003, mass=1.4, sig=0.223)
    v = wo.potential_generate(cfg)
    potsub = wo.potential_sub(cfg)
    phi = random_phi(cfg, seed=3)
    want = wo.observables(cfg, v, phi, potsub)
    with wa.Context(par) as ctx:
        ctx.set_potential(pot)
        ctx.upload_phi(phi)
        got = ctx.observables()
    for k in want:
        if np.isnan(want[k]):  # FullCornell's pot_sub is NaN at r = 0 for odd grids (potential.rs:335)
            assert np.isnan(got[k])
        else:
            assert got[k] == pytest.approx(want[k], rel=REL_SUM, abs=1e-300), k


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("kind", ["AboutZ", "AntisymAboutZ", "AboutY", "AntisymAboutY"])
@pytest.mark.parametrize("shape", [(70, 9, 12), (12, 16, 7)])
def test_symmetrise(wo, wa, kind, shape, dtype):
    """wafer_symmetrise against config::symmetrise_wavefunction's loops (config.rs:691-728)"""
    cfg, par = make_pair(shape, ext=3, dtype=dtype)
    phi = random_phi(cfg, seed=21)
    if dtype == "f32":
        phi = phi.astype(np.float32).astype(np.float64)
    want = phi.copy()
    wo.symmetrise(cfg, kind, want)
    with wa.Context(par) as ctx:
        ctx.upload_phi(phi)
        ctx.symmetrise("NotConstrained")
        assert np.array_equal(ctx.download_phi(), phi)
        ctx.symmetrise(kind)
        got = ctx.download_phi()
        assert np.array_equal(got, want)
        n2 = ctx.norm2()                      # the engine keeps working on the swapped buffer
        assert n2 == pytest.approx(wo.norm2(cfg, want), rel=1e-6 if dtype == "f32" else REL_SUM)
        ctx.set_potential("Harmonic")
        ctx.evolve(0, 2)                      # and the frame of that buffer is a frame
        a, b = wo.ab(cfg, wo.potential_generate(cfg))
        if dtype == "f64":
            wo.evolve(cfg, 0, a, b, want, [], 2)
            assert np.array_equal(ctx.download_phi(), want)


def test_symmetrise_refuses_what_the_reference_cannot_index(wa):
    with wa.Context(wa.Params(8, 8, 8, dn=0.1, dt=0.001, central_difference=2)) as ctx:
        ctx.set_initial_condition("Boolean")
        ctx.symmetrise("NotConstrained")
        with pytest.raises(wa.WaferError, match="SevenPoint"):
            ctx.symmetrise("AboutY")
    with wa.Context(wa.Params(8, 8, 16, dn=0.1, dt=0.001, central_difference=3, z_begin=0, z_count=8)) as ctx:
        ctx.set_initial_condition("Boolean")
        ctx.symmetrise("AntisymAboutY")       # local to every z-plane
        with pytest.raises(wa.WaferError, match="z-slabs"):
            ctx.symmetrise("AboutZ")


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_potsub_override(wo, wa, kind):
    """wafer_set_potsub: a potential_sub file replaces the computed pot_sub for any potential
    (potential.rs:113-131); v_infinity follows grid.rs:408-427"""
    cfg, par = make_pair((20, 17, 22), ext=2, potential="SimpleCornell", dn=0.15, dt=0.003, mass=1.4, sig=0.223)
    v = wo.potential_generate(cfg)
    phi = random_phi(cfg, seed=4)
    arr = np.random.default_rng(12).standard_normal(cfg.work_shape) if kind == 2 else None
    want = wo.observables(cfg, v, phi, (kind, 0.75 if kind == 1 else 0.0, arr))
    with wa.Context(par) as ctx:
        with pytest.raises(wa.WaferError):
            ctx.set_potsub(1, 0.75)          # no potential yet
        ctx.set_potential("SimpleCornell")
        assert ctx.potsub()[0] == 1          # computed: the singular 4m term
        ctx.set_potsub(kind, 0.75, arr)
        assert ctx.potsub() == (kind, 0.75 if kind == 1 else 0.0)
        ctx.upload_phi(phi)
        got = ctx.observables()
    for k in want:
        assert got[k] == pytest.approx(want[k], rel=REL_SUM, abs=1e-300), k
    assert (want["v_infinity"] == 0.0) == (kind == 0)


def test_potsub_array_of_another_resolution(wo, wa):
    """input::fill_sub_data (input.rs:453-478): a potential_sub array whose dims differ from the
    grid is resampled with trilerp_resize onto (nx, ny, nz) -- basis = that target size, no frame --
    bit exact against the oracle, on one context and on z-slabs"""
    cfg, par = make_pair((18, 21, 24), ext=2, potential="FullCornell", dn=0.15, dt=0.003, mass=1.4, sig=0.223)
    src = np.random.default_rng(5).standard_normal((7, 9, 5))
    want = wo.trilerp_resize(src, cfg.work_shape)
    v = wo.potential_generate(cfg)
    phi = random_phi(cfg, seed=4)
    obs = wo.observables(cfg, v, phi, (2, 0.0, want))
    with wa.Context(par) as ctx:
        with pytest.raises(wa.WaferError):
            ctx.set_potsub_resampled(src)   # no potential yet
        ctx.set_potential("FullCornell")
        ctx.set_potsub_resampled(src)
        assert ctx.potsub() == (2, 0.0)
        assert np.array_equal(ctx.download_array("potsub"), want)
        ctx.upload_phi(phi)
        got = ctx.observables()
    for k in obs:
        assert got[k] == pytest.approx(obs[k], rel=REL_SUM, abs=1e-300), k
    import dataclasses
    for z0, zc in ((0, 10), (10, 14)):
        with wa.Context(dataclasses.replace(par, z_begin=z0, z_count=zc)) as ctx:
            ctx.set_potential("FullCornell")
            ctx.set_potsub_resampled(src)
            assert np.array_equal(ctx.download_array("potsub")[:, :, z0:z0 + zc], want[:, :, z0:z0 + zc])


def test_norm_normalise_orthogonalise(wo, wa):
    cfg, par = make_pair((19, 22, 17), ext=2)
    phi = random_phi(cfg, seed=8)
    lowers = [random_phi(cfg, seed=20 + i) for i in range(3)]
    for l in lowers:
        wo.normalise(l, wo.norm2(cfg, l))
    with wa.Context(par) as ctx:
        for i, l in enumerate(lowers):
            ctx.load_state(i, l)
        assert ctx.num_states() == 3
        ctx.upload_phi(phi)
        n2 = ctx.norm2()
        assert n2 == pytest.approx(wo.norm2(cfg, phi), rel=REL_SUM)
        n2 = wo.norm2(cfg, phi)
        ctx.normalise(n2)
        wo.normalise(phi, n2)
        assert ulp_diff(ctx.download_phi(), phi) == 0  # true division, bit exact
        ctx.orthogonalise(3)
        wo.orthogonalise(3, phi, lowers)
        assert np.allclose(ctx.download_phi(), phi, rtol=0, atol=1e-14)
        for i, l in enumerate(lowers):
            assert np.array_equal(ctx.download_state(i), l)
        with pytest.raises(wa.WaferError):
            ctx.orthogonalise(4)  # w_store too short


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("wnum", [1, 2, 3])
def test_excited_state_evolve(wo, wa, wnum, variant):
    """grid.rs:674-681: per-step renormalise + modified Gram-Schmidt"""
    cfg, par = make_pair((24, 20, 28), ext=1, potential="Harmonic", dn=0.3, dt=0.01)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    lowers = []
    for i in range(wnum):  # an orthonormal set, as converged states would be
        l = random_phi(cfg, seed=30 + i)
        wo.orthogonalise(i, l, lowers)
        wo.normalise(l, wo.norm2(cfg, l))
        lowers.append(l)
    phi = random_phi(cfg, seed=40)
    with wa.Context(par) as ctx:
        ctx.set_stencil_variant(variant)
        ctx.set_potential("Harmonic")
        for i, l in enumerate(lowers):
            ctx.load_state(i, l)
        ctx.upload_phi(phi)
        ctx.evolve(wnum, 25)
        wo.evolve(cfg, wnum, a, b, phi, lowers, 25)
        got = ctx.download_phi()
        assert np.allclose(got, phi, rtol=0, atol=1e-13)
        # normalise comes BEFORE the projection (grid.rs:679-680), so norm2 = 1 - sum s_l^2
        assert ctx.norm2() == pytest.approx(wo.norm2(cfg, phi), rel=1e-12)
        for l in lowers:
            assert abs(np.sum(l * got)) < 1e-13


@pytest.mark.parametrize("wnum", [1, 3])
def test_excited_state_evolve_without_the_division_plan(wo, wa, wnum):
    """WAFER_FLAG_UNPLANNED_DIV (what a denominator the plan cannot clear would run): the excited-state steps keep to the one-step
    kernels -- the two-steps-per-pass kernel exists for the short arithmetic forms only -- and agree with the oracle as ever"""
    cfg, par = make_pair((130, 36, 40), ext=1, potential="Coulomb", dn=0.3, dt=0.01, unplanned_div=True)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    lowers = _orthonormal_store(wo, cfg, wnum)
    phi = random_phi(cfg, seed=41)
    with wa.Context(par) as ctx:
        assert ctx.div_plan().checked == 0
        ctx.set_potential("Coulomb")
        for i, l in enumerate(lowers):
            ctx.load_state(i, l)
        ctx.upload_phi(phi)
        ctx.evolve(wnum, 12)
        assert ctx.x2_passes() == 0
        wo.evolve(cfg, wnum, a, b, phi, lowers, 12)
        assert np.allclose(ctx.download_phi(), phi, rtol=0, atol=1e-13)
        assert ctx.norm2() == pytest.approx(wo.norm2(cfg, phi), rel=1e-12)
    par.unplanned_div = False
    with wa.Context(par) as ctx:   # ... and with the plan the same grid does take two steps per pass
        ctx.set_potential("Coulomb")
        for i, l in enumerate(lowers):
            ctx.load_state(i, l)
        ctx.upload_phi(random_phi(cfg, seed=41))
        ctx.evolve(wnum, 12)
        assert ctx.x2_passes() > 0
        assert np.allclose(ctx.download_phi(), phi, rtol=0, atol=1e-13)


@pytest.mark.parametrize("wnum", [2, 4, 5])
def test_excited_state_evolve_nonorthogonal_store(wo, wa, wnum):
    """stored states that are NOT orthonormal: the one-pass Gram-Schmidt (raw
    overlaps + Gram-matrix recurrence) must still reproduce the reference's
    sequential modified Gram-Schmidt; wnum = 5 exceeds the fused kernel's
    capacity and takes the kernel-per-projection path"""
    cfg, par = make_pair((26, 19, 23), ext=2, potential="Coulomb", dn=0.25, dt=0.006, max_states=5)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    lowers = []
    for i in range(wnum):
        l = random_phi(cfg, seed=60 + i) + (0.4 * lowers[0] if lowers else 0.0)   # correlated on purpose
        wo.normalise(l, wo.norm2(cfg, l))
        lowers.append(np.ascontiguousarray(l))
    phi = random_phi(cfg, seed=70)
    with wa.Context(par) as ctx:
        ctx.set_potential("Coulomb")
        for i, l in enumerate(lowers):
            ctx.load_state(i, l)
        ctx.upload_phi(phi)
        ctx.evolve(wnum, 6)
        wo.evolve(cfg, wnum, a, b, phi, lowers, 6)
        assert np.allclose(ctx.download_phi(), phi, rtol=0, atol=2e-13)
        assert ctx.norm2() == pytest.approx(wo.norm2(cfg, phi), rel=1e-11)


def _orthonormal_store(wo, cfg, wnum, seed=30, correlated=0.0):
    lowers = []
    for i in range(wnum):
        l = random_phi(cfg, seed=seed + i)
        if correlated and lowers:
            l = l + correlated * lowers[0]          # deliberately NOT orthogonal to the first
        else:
            wo.orthogonalise(i, l, lowers)
        wo.normalise(l, wo.norm2(cfg, l))
        lowers.append(np.ascontiguousarray(l))
    return lowers


X2_SHAPES = [(24, 20, 28), (150, 37, 29), (129, 17, 9), (65, 33, 20), (3, 2, 5), (260, 8, 40)]   # (more cells than stored states)


@pytest.mark.parametrize("steps", [4, 7, 12])
@pytest.mark.parametrize("potential", ["Harmonic", "Coulomb", "SimpleCornell", "Cube"])
@pytest.mark.parametrize("wnum", [1, 2, 3])
def test_two_excited_steps_per_pass_vs_oracle(wo, wa, wnum, potential, steps, monkeypatch):
    """wafer_stencil_x2.hip.h against the reference's sequence (grid.rs:562-686, wnum > 0: step, norm^2, normalise, modified
    Gram-Schmidt after EVERY step): the pass regroups the sums by linearity, so the bar is the excited-state one, 1e-13 per
    cell.  Closed-form V (Harmonic, Coulomb, SimpleCornell) and streamed V (Cube); even and odd step counts; ragged tiles,
    grids smaller than a tile.  (WAFER_X2_MAX_K=3: the default keeps three stored states on the one-step kernel.)"""
    monkeypatch.setenv("WAFER_X2_MAX_K", "3")
    for shape in X2_SHAPES:
        cfg, par = make_pair(shape, ext=1, potential=potential, dn=0.3, dt=0.01, mass=1.3, sig=0.4)
        v = wo.potential_generate(cfg)
        a, b = wo.ab(cfg, v)
        lowers = _orthonormal_store(wo, cfg, wnum)
        phi = random_phi(cfg, seed=40)
        with wa.Context(par) as ctx:
            ctx.set_potential(potential)
            for i, l in enumerate(lowers):
                ctx.load_state(i, l)
            ctx.upload_phi(phi)
            ctx.evolve(wnum, steps)
            assert ctx.x2_passes() == (steps - 2 - steps % 2) // 2, "the two-step kernel did not run"
            wo.evolve(cfg, wnum, a, b, phi, lowers, steps)
            got = ctx.download_phi()
            assert np.allclose(got, phi, rtol=0, atol=1e-13), (shape, float(np.max(np.abs(got - phi))))
            assert ctx.norm2() == pytest.approx(wo.norm2(cfg, phi), rel=1e-12)
            # a second call continues from the materialised phi
            ctx.evolve(wnum, 6)
            wo.evolve(cfg, wnum, a, b, phi, lowers, 6)
            assert np.allclose(ctx.download_phi(), phi, rtol=0, atol=1e-13)


@pytest.mark.parametrize("dtype", ["f32", "f32fast"])
@pytest.mark.parametrize("wnum", [1, 2, 3])
def test_two_excited_steps_per_pass_on_fp32_storage(wo, wa, wnum, dtype, monkeypatch):
    """round 6: wafer_k_xstep2 on the storage tag of the three-step kernel (dtype f32, and f32fast, whose excited-state steps compute
    in fp64 too): the raw pass result, V, the stored states and their images are float in HBM, queues / LDS / sums double.  No
    bit-level reference exists for it (the regrouped sums differ from the one-step kernel's by construction, and every pass rounds
    what it stores to float): held to the fp32-storage bar -- per cell within 3e-6 of the largest value of the fp64 ORACLE's state
    after 12 steps, and no further from it than the one-step fp32-storage kernels are; ragged tiles, whole tiles, a grid smaller
    than a tile, every stored-state count; norm to 1e-5."""
    monkeypatch.setenv("WAFER_X2_MAX_K", "3")
    r32 = lambda x: np.ascontiguousarray(x.astype(np.float32).astype(np.float64))
    for shape in [(150, 37, 29), (128, 32, 20), (24, 20, 28)]:
        cfg, par = make_pair(shape, ext=1, potential="Coulomb", dn=0.3, dt=0.01, mass=1.3, sig=0.4, dtype=dtype)
        v = r32(wo.potential_generate(cfg))              # what a float array holds
        a, b = wo.ab(cfg, v)
        lowers = [r32(l) for l in _orthonormal_store(wo, cfg, wnum)]
        phi0 = r32(random_phi(cfg, seed=40))
        want = phi0.copy()
        wo.evolve(cfg, wnum, a, b, want, lowers, 12)
        got = {}
        for x2 in ("1", "0"):
            monkeypatch.setenv("WAFER_X2", x2)
            with wa.Context(par) as ctx:
                ctx.set_potential("Coulomb")
                for i, l in enumerate(lowers):
                    ctx.load_state(i, l)
                ctx.upload_phi(phi0)
                ctx.evolve(wnum, 12)
                assert ctx.x2_passes() == (5 if x2 == "1" else 0), "the two-step kernel did not run" if x2 == "1" else "it ran"
                got[x2] = (ctx.download_phi(), ctx.norm2())
        scale = float(np.max(np.abs(want)))
        err = {k: float(np.max(np.abs(g[0] - want))) / scale for k, g in got.items()}
        assert err["1"] <= 3e-6 and err["0"] <= 3e-6, (shape, err)
        assert err["1"] <= 3.0 * err["0"] + 2e-7, (shape, err)            # regrouping costs no accuracy worth the name
        for g in got.values():
            assert g[1] == pytest.approx(wo.norm2(cfg, want), rel=1e-5)


@pytest.mark.parametrize("ry", ["1", "2"])
@pytest.mark.parametrize("zchunk", ["1", "2", "3", "5", "1000"])
def test_two_excited_steps_per_pass_zchunking_and_tile_heights(wo, wa, zchunk, ry, monkeypatch):
    """every z-chunking (the chunks overlap by a plane of Y1 on each side and split the sums differently) and both tile
    heights for one stored state (128 x 16: RY = 2, 128 x 8: RY = 1)"""
    monkeypatch.setenv("WAFER_ZCHUNK", zchunk)
    monkeypatch.setenv("WAFER_X2_RY", ry)
    cfg, par = make_pair((140, 35, 23), ext=1, potential="Coulomb", dn=0.25, dt=0.008)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    for wnum in (1, 2):
        lowers = _orthonormal_store(wo, cfg, wnum, seed=50)
        phi = random_phi(cfg, seed=57)
        with wa.Context(par) as ctx:
            ctx.set_potential("Coulomb")
            for i, l in enumerate(lowers):
                ctx.load_state(i, l)
            ctx.upload_phi(phi)
            ctx.evolve(wnum, 10)
            assert ctx.x2_passes() == 4
            wo.evolve(cfg, wnum, a, b, phi, lowers, 10)
            assert np.allclose(ctx.download_phi(), phi, rtol=0, atol=1e-13)


@pytest.mark.parametrize("wnum", [2, 3])
def test_two_excited_steps_per_pass_nonorthogonal_store(wo, wa, wnum, monkeypatch):
    """stored states that are NOT orthonormal (overlaps of order one in every step): the regrouped sums -- Gram matrix,
    <l_j, A l_i>, <A l_i, A l_j> -- must still reproduce the reference's sequential modified Gram-Schmidt"""
    monkeypatch.setenv("WAFER_X2_MAX_K", "3")
    cfg, par = make_pair((40, 33, 26), ext=1, potential="SimpleCornell", dn=0.25, dt=0.006, mass=2.0, sig=0.3)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    lowers = _orthonormal_store(wo, cfg, wnum, seed=60, correlated=0.4)
    phi = random_phi(cfg, seed=70)
    with wa.Context(par) as ctx:
        ctx.set_potential("SimpleCornell")
        for i, l in enumerate(lowers):
            ctx.load_state(i, l)
        ctx.upload_phi(phi)
        ctx.evolve(wnum, 16)
        assert ctx.x2_passes() == 7
        wo.evolve(cfg, wnum, a, b, phi, lowers, 16)
        assert np.allclose(ctx.download_phi(), phi, rtol=0, atol=2e-13)
        assert ctx.norm2() == pytest.approx(wo.norm2(cfg, phi), rel=1e-11)


def test_two_excited_steps_per_pass_follows_store_and_potential_changes(wo, wa):
    """the images M_j = A l_j and their matrices are rebuilt when w_store or V changes; a start that IS a stored state (the
    reference's clone, grid.rs:95) goes through the one-step head and stays finite; long run against the one-step path"""
    cfg, par = make_pair((48, 40, 36), ext=1, potential="Harmonic", dn=0.3, dt=0.01)
    lowers = _orthonormal_store(wo, cfg, 2, seed=80)
    with wa.Context(par) as ctx:
        ctx.set_potential("Harmonic")
        ctx.load_state(0, lowers[0])
        ctx.upload_phi(random_phi(cfg, seed=81))
        ctx.evolve(1, 8)
        n0 = ctx.x2_passes()
        assert n0 == 3
        # a second stored state, another potential: both must be picked up
        ctx.load_state(1, lowers[1])
        ctx.set_potential("Coulomb")
        cfg2, _ = make_pair((48, 40, 36), ext=1, potential="Coulomb", dn=0.3, dt=0.01)
        v = wo.potential_generate(cfg2)
        a, b = wo.ab(cfg2, v)
        phi = random_phi(cfg, seed=82)
        ctx.upload_phi(phi)
        ctx.evolve(2, 9)
        assert ctx.x2_passes() == n0 + 3
        wo.evolve(cfg2, 2, a, b, phi, lowers, 9)
        assert np.allclose(ctx.download_phi(), phi, rtol=0, atol=1e-13)
        # the clone start of solve (grid.rs:95, 130-135): normalised, projected to rounding noise, then evolved
        ctx.clone_state_to_phi(1)
        ctx.normalise(ctx.norm2())
        ctx.orthogonalise(2)
        ctx.evolve(2, 40)
        got = ctx.download_phi()
        assert np.all(np.isfinite(got)) and abs(ctx.norm2() - 1.0) < 1e-4   # 1 - sum s_j^2: normalise precedes the projection (grid.rs:679-680)
        for l in lowers:
            assert abs(np.sum(l * got)) < 1e-12


@pytest.mark.parametrize("ext", [1, 2, 3])
@pytest.mark.parametrize("wnum", [1, 3, 4])
def test_one_pass_excited_step_equals_two_pass(wa, wnum, ext, monkeypatch):
    """the one-pass excited-state step (previous raw result normalised and projected on load)
    performs the two-pass scheme's operations on the same operands: identical bits"""
    shape = (70, 21, 19)
    out = {}
    monkeypatch.setenv("WAFER_X2", "0")   # (two STEPS per pass regroup the sums: compared with the oracle above, not bit for bit)
    for mode in ("0", "1"):
        monkeypatch.setenv("WAFER_ONE_PASS", mode)
        par = wa.Params(*shape, dn=0.2, dt=0.004, central_difference=ext, max_states=4)
        with wa.Context(par) as ctx:
            ctx.set_potential("Coulomb")
            for i in range(wnum):
                ctx.set_initial_condition("Gaussian", seed=80 + i)
                ctx.normalise(ctx.norm2())
                ctx.push_state()
            ctx.set_initial_condition("Gaussian", seed=90)
            ctx.evolve(wnum, 7)
            ctx.evolve(wnum, 1)
            out[mode] = (ctx.download_phi(), ctx.norm2())
    assert np.array_equal(out["0"][0], out["1"][0]) and out["0"][1] == out["1"][1]


@pytest.mark.parametrize("potential", ["Coulomb", "SimpleCornell", "Harmonic", "ComplexCoulomb", "ComplexHarmonic"])
@pytest.mark.parametrize("ext,x2", [(1, "0"), (2, "0"), (3, "0"), (1, "1")])   # two steps per pass: ThreePoint only
@pytest.mark.parametrize("wnum", [1, 2, 3])
def test_closed_form_potential_evaluated_in_the_excited_step_kernel(wa, wnum, ext, potential, x2, monkeypatch):
    """the excited-state step kernels evaluate Coulomb / SimpleCornell / Harmonic (potential.rs:221-229,
    241-249, 270-274) per cell instead of streaming the stored V: the same function that filled the
    array, so identical bits per cell and identical sums (WAFER_VGEN=0 streams V); odd axes put a cell
    at r = 0 (the r < dn clamp), the ragged shape leaves partial tiles"""
    shape = (133, 21, 19)
    out = {}
    # x2 = 0: one step per pass; x2 = 1: two (wafer_stencil_x2.hip.h) -- there the comparison needs the SAME tile height on
    # both sides (the partial sums of a pass follow the tiles, and the default gives a streamed V the lower tile)
    monkeypatch.setenv("WAFER_X2", x2)
    monkeypatch.setenv("WAFER_X2_RY", "2")
    monkeypatch.setenv("WAFER_X2_MAX_K", "3")
    for mode in ("0", "1"):
        monkeypatch.setenv("WAFER_VGEN", mode)
        par = wa.Params(*shape, dn=0.2, dt=0.004, mass=1.3, sig=0.223, central_difference=ext, max_states=4)
        with wa.Context(par) as ctx:
            ctx.set_potential(potential)
            for i in range(wnum):
                ctx.set_initial_condition("Gaussian", seed=80 + i)
                ctx.normalise(ctx.norm2())
                ctx.push_state()
            ctx.set_initial_condition("Gaussian", seed=90)
            ctx.evolve(wnum, 6)
            ctx.evolve(wnum, 1)
            out[mode] = [ctx.download_phi(), ctx.norm2(), ctx.observables()]
            # the single-step ground-state kernel and compute_observables take the closed form as well
            ctx.set_stencil_variant(1)
            ctx.evolve(0, 3)
            out[mode] += [ctx.download_phi(), ctx.observables()]
    assert np.array_equal(out["0"][0], out["1"][0]) and out["0"][1] == out["1"][1]
    assert out["0"][2] == out["1"][2]
    assert np.array_equal(out["0"][3], out["1"][3]) and out["0"][4] == out["1"][4]


def test_closed_form_potential_is_dropped_when_the_potential_is_replaced(wo, wa):
    """a host potential uploaded AFTER a built-in one must be the one the excited-state kernels use"""
    cfg, par = make_pair((24, 20, 28), ext=1, potential="Harmonic", dn=0.3, dt=0.01)
    v = wo.potential_generate(cfg) * 1.75 + 0.3
    a, b = wo.ab(cfg, v)
    low = random_phi(cfg, seed=31)
    wo.normalise(low, wo.norm2(cfg, low))
    phi = random_phi(cfg, seed=41)
    with wa.Context(par) as ctx:
        ctx.set_potential("Harmonic")
        ctx.set_potential_host(v)
        ctx.load_state(0, low)
        ctx.upload_phi(phi)
        ctx.evolve(1, 9)
        wo.evolve(cfg, 1, a, b, phi, [low], 9)
        assert np.allclose(ctx.download_phi(), phi, rtol=0, atol=1e-13)


def test_solve_matches_oracle(wo, wa):
    """grid.rs:50-246: same block table (step, tau, E, diff) and stop step for the
    ground state and two excited states.  Excited states start here from a fresh
    O(1) guess on both sides (the reference's from-disk branch, grid.rs:70): its
    other start, a clone of the previous state (grid.rs:95), is annihilated by
    Gram-Schmidt down to rounding noise and regrows from that noise, which no two
    implementations (or two runs of the reference: rayon sums) reproduce."""
    cfg, par = make_pair((32, 32, 32), ext=1, potential="Harmonic", dn=0.4, dt=0.032, mass=1.0)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    with wa.Context(par) as ctx:
        ctx.set_potential("Harmonic")
        store, energies = [], []
        for wnum in range(3):
            phi = wo.initial_condition(cfg, "Gaussian", seed=3 + wnum)
            ctx.upload_phi(phi)
            want, conv = wo.solve(cfg, wnum, v, a, b, phi, store, 1e-9, 100, max_steps=100000)
            got, final, gconv = ctx.solve_state(wnum, 1e-9, 100, max_steps=100000)
            assert conv and gconv
            assert abs(len(got) - len(want)) <= 1  # the stop test is |dE| < 1e-9 on sums
            for g, w in zip(got, want):
                assert g["step"] == w["step"] and g["tau"] == w["tau"]
                assert g["energy"] / g["norm2"] == pytest.approx(w["energy"] / w["norm2"], abs=2e-9)
                assert np.sqrt(g["r2"] / g["norm2"]) == pytest.approx(np.sqrt(w["r2"] / w["norm2"]), rel=1e-7)
            assert final["energy"] == pytest.approx(want[-1]["energy"] / want[-1]["norm2"], abs=2e-9)
            assert final["state"] == wnum and final["l_r"] == pytest.approx(32 / final["r"])
            assert ctx.num_states() == wnum + 1
            store.append(phi.copy())
            energies.append(final["energy"])
        assert energies[0] == pytest.approx(1.5, abs=0.02)
        assert energies[1] == pytest.approx(2.5, abs=0.04) and energies[2] == pytest.approx(2.5, abs=0.04)


def test_solve_from_clone_of_previous_state(wa):
    """the reference's default excited-state start (grid.rs:95): clone, Gram-Schmidt
    to noise, regrow.  Own trajectory; only the eigenvalue is checked."""
    par = wa.Params(32, 32, 32, dn=0.4, dt=0.032, mass=1.0, max_states=2)
    with wa.Context(par) as ctx:
        ctx.set_potential("Harmonic")
        ctx.set_initial_condition("Gaussian", seed=3)
        _, f0, c0 = ctx.solve_state(0, 1e-9, 100, max_steps=100000)
        ctx.clone_state_to_phi(0)
        try:
            _, f1, c1 = ctx.solve_state(1, 1e-9, 100, max_steps=100000)
        except wa.WaferError as e:   # the documented hazard: the clone cancelled to exactly zero
            assert "not finite" in str(e)
            return
        assert c0 and c1
        assert f0["energy"] == pytest.approx(1.5, abs=0.02) and f1["energy"] == pytest.approx(2.5, abs=0.04)


def test_solve_max_steps(wa):
    """ErrorKind::MaxStep (grid.rs:211-213, 244): tested after the step, with >"""
    par = wa.Params(16, 16, 16, dn=0.4, dt=0.032)
    with wa.Context(par) as ctx:
        ctx.set_potential("Harmonic")
        ctx.set_initial_condition("Boolean")
        recs, final, conv = ctx.solve_state(0, 1e-300, 10, max_steps=25)
        assert not conv
        assert [r["step"] for r in recs] == [0, 10, 20, 30]  # 30 > 25 stops it
        assert ctx.num_states() == 0


def test_caller_stream_state_store_and_slab_info(wo, wa):
    """wafer_set_stream (the engine runs on a caller-owned hipStream_t), wafer_clear_states,
    wafer_get_slab_info, wafer_get_device_info"""
    import torch
    cfg, par = make_pair((70, 30, 41), ext=1, potential="Coulomb", max_states=2)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    want = random_phi(cfg, seed=2)
    phi = want.copy()
    wo.evolve(cfg, 0, a, b, want, [], 7)
    with wa.Context(par) as ctx:
        mine = torch.cuda.Stream()
        ctx.set_stream(mine.cuda_stream)
        ctx.set_potential("Coulomb")
        ctx.upload_phi(phi)
        ctx.evolve(0, 7)
        mine.synchronize()                       # everything was enqueued on the caller's stream
        assert np.array_equal(ctx.download_phi(), want)
        ctx.set_stream(None)                     # back to the context's own stream
        ctx.evolve(0, 1)
        info = ctx.slab_info()
        assert (info["z_begin"], info["z_count"], info["halo_depth"], info["ext"], info["elem_bytes"]) == (0, 41, 1, 1, 8)
        assert info["plane_elems"] >= (30 + 2) * (70 + 2)
        dev = ctx.device_info()
        assert dev["compute_units"] > 0 and dev["total_bytes"] > 0 and dev["arch"].startswith("gfx")
        ctx.push_state()
        ctx.push_state()
        assert ctx.num_states() == 2
        with pytest.raises(wa.WaferError):
            ctx.push_state()                     # max_states = 2
        ctx.clear_states()
        assert ctx.num_states() == 0
        ctx.push_state()
        assert ctx.num_states() == 1


def test_out_of_memory_fails_loudly_and_leaves_the_device_usable(wa):
    """a grid no device holds: wafer_ctx_create reports the HIP error, frees what it had taken,
    and the stale error does not resurface in the next context"""
    import torch
    free0 = torch.cuda.mem_get_info(0)[0]
    with pytest.raises(wa.WaferError, match="memory"):
        wa.Context(wa.Params(4096, 4096, 4096, dn=0.05, dt=5e-4))
    assert torch.cuda.mem_get_info(0)[0] >= free0 - (1 << 30)
    with wa.Context(wa.Params(40, 40, 40, dn=0.2, dt=0.004)) as ctx:
        ctx.set_potential("Harmonic")
        ctx.set_initial_condition("Boolean")
        ctx.evolve(0, 4)
        assert np.isfinite(ctx.norm2())


def test_config_validation(wa):
    with pytest.raises(wa.WaferError):  # ErrorKind::LargeDt, config.rs:362-365
        wa.Context(wa.Params(8, 8, 8, dn=0.1, dt=0.1))
    with pytest.raises(wa.WaferError):
        wa.Context(wa.Params(8, 8, 8, dn=0.1, dt=1e-3, central_difference=4))
    with wa.Context(wa.Params(8, 8, 8, dn=0.1, dt=1e-3, max_states=1)) as ctx:
        with pytest.raises(wa.WaferError):
            ctx.evolve(0, 1)  # nothing set yet
        ctx.set_potential("NoPotential")
        ctx.set_initial_condition("Constant")
        ctx.push_state()
        with pytest.raises(wa.WaferError):
            ctx.push_state()  # w_store full


# ---------------------------------------------------------------- fp32 storage path
def test_f32_path_tracks_f64(wo, wa):
    """config #5's cross-check at a size the test can afford: same potential in
    fp64 and fp32 storage, relative energy error <= 1e-5, |norm2 - 1| <= 1e-5"""
    shape = (48, 48, 48)
    res = {}
    for dtype in ("f64", "f32", "f32fast"):
        _, par = make_pair(shape, ext=1, potential="Harmonic", dn=0.27, dt=0.0145, dtype=dtype)
        with wa.Context(par) as ctx:
            ctx.set_potential("Harmonic")
            ctx.set_initial_condition("Boolean")
            recs, final, conv = ctx.solve_state(0, 1e-7, 200, max_steps=40000)
            assert conv
            ctx.clone_state_to_phi(0)
            res[dtype] = (final["energy"], ctx.norm2())
    assert res["f32"][0] == pytest.approx(res["f64"][0], rel=1e-5)
    assert res["f32"][1] == pytest.approx(1.0, abs=1e-5)
    assert res["f64"][1] == pytest.approx(1.0, abs=1e-12)
    # fp32 arithmetic in the stencil steps as well (WAFER_F32_FAST): same bars
    assert res["f32fast"][0] == pytest.approx(res["f64"][0], rel=1e-5)
    assert res["f32fast"][1] == pytest.approx(1.0, abs=1e-5)


@pytest.mark.parametrize("ext", [1, 2])
@pytest.mark.parametrize("dtype", ["f32", "f32fast"])
def test_f32_storage_every_kernel_gives_the_same_bits(wa, dtype, ext):
    """fp32 storage is not a bit-for-bit path against the fp64 oracle, but it must not depend on WHICH kernel advances the
    steps: the fused two-step kernel against the single-step kernel, every cell's bits (round 3: the fused kernel used to
    hand its second step fp32-rounded a, b).  (The plain kernel, variant 0, streams the STORED a, b arrays -- fp32 like every
    stored array -- and is a different, equally legitimate reading of "fp32 storage"; it is not part of this comparison.)"""
    shape = (200, 37, 29)
    out = {}
    for variant in (2, 1):
        par = wa.Params(*shape, dn=0.2, dt=0.004, mass=1.3, central_difference=ext, dtype=dtype)
        with wa.Context(par) as ctx:
            ctx.set_stencil_variant(variant)
            ctx.set_potential("Coulomb")
            ctx.set_initial_condition("Gaussian", seed=5)
            ctx.evolve(0, 8)
            out[variant] = ctx.download_phi()
    assert np.array_equal(out[2], out[1])


@pytest.mark.parametrize("sched", ["up", "halves", "down"])
@pytest.mark.parametrize("shape,steps", [((200, 37, 29), 9), ((300, 20, 18), 7), ((257, 33, 40), 12), ((64, 16, 9), 6),
                                         ((128, 16, 5), 9), ((256, 32, 11), 10), ((128, 48, 23), 11)])   # the last three: whole 128 x 16 tiles (exact store counts, ring queues)
def test_fp32_storage_three_step_kernel_gives_the_single_step_kernels_bits(wa, shape, steps, sched, monkeypatch):
    """dtype f32 -- fp32 STORAGE, fp64 arithmetic: config #5's -- on the three-step kernel (round 5: float in HBM, the fp64
    kernel's registers and LDS, every level's result rounded to float before it is used) against the single-step kernel and
    the two-step kernel on the same storage: every cell's bits; ragged tiles and grids of whole tiles (exact store counts and
    ring queues), z-chunks, marching up / down / as two halves, step counts with a two-step and a one-step remainder"""
    monkeypatch.setenv("WAFER_FUSE3_MIN_NY", "1")
    monkeypatch.setenv("WAFER_F3_SCHED", "1" if sched == "halves" else "0")
    monkeypatch.setenv("WAFER_F3_PLAIN_DOWN", "1" if sched == "down" else "0")
    out = {}
    for variant, zchunk in ((3, ""), (3, "5"), (3, "2"), (2, ""), (1, "")):
        if zchunk:
            monkeypatch.setenv("WAFER_ZCHUNK", zchunk)
        else:
            monkeypatch.delenv("WAFER_ZCHUNK", raising=False)
        par = wa.Params(*shape, dn=0.2, dt=0.004, mass=1.3, central_difference=1, dtype="f32")
        with wa.Context(par) as ctx:
            ctx.set_stencil_variant(variant)
            ctx.set_potential("Coulomb")
            ctx.set_initial_condition("Gaussian", seed=5)
            if variant == 3:
                assert ctx.stencil_kernel_name() == "wafer_k_step3_fused" and ctx.steps_per_launch() == 3
            ctx.evolve(0, steps)
            if variant == 3:
                assert ctx.stencil_kernel_instance().startswith("wafer_k_step3_fused<wafer_f32_wide, double, ")
            out[(variant, zchunk)] = ctx.download_phi()
    for key in ((3, ""), (3, "5"), (3, "2"), (2, "")):
        assert np.array_equal(out[key], out[(1, "")]), key


@pytest.mark.parametrize("shape,steps", [((200, 37, 29), 9), ((300, 20, 18), 7), ((257, 33, 40), 12), ((64, 16, 9), 6)])
def test_all_fp32_three_step_kernel_gives_the_single_step_kernels_bits(wa, shape, steps, monkeypatch):
    """f32fast (fp32 storage AND fp32 step arithmetic) on the three-step kernel -- 256 x 16 tiles, a and b carried between the
    levels in fp32, which is this path's arithmetic type -- against the single-step fp32 kernel: every cell's bits, ragged
    tiles, z-chunks, an odd step count (a trailing two-step pass / single step)"""
    monkeypatch.setenv("WAFER_FUSE3_MIN_NY", "1")
    out = {}
    for variant, zchunk in ((3, ""), (3, "5"), (1, "")):
        if zchunk:
            monkeypatch.setenv("WAFER_ZCHUNK", zchunk)
        else:
            monkeypatch.delenv("WAFER_ZCHUNK", raising=False)
        par = wa.Params(*shape, dn=0.2, dt=0.004, mass=1.3, central_difference=1, dtype="f32fast")
        with wa.Context(par) as ctx:
            ctx.set_stencil_variant(variant)
            ctx.set_potential("Coulomb")
            ctx.set_initial_condition("Gaussian", seed=5)
            if variant == 3:
                assert ctx.stencil_kernel_name() == "wafer_k_step3_fused"
            ctx.evolve(0, steps)
            out[(variant, zchunk)] = ctx.download_phi()
    assert np.array_equal(out[(3, "")], out[(1, "")])
    assert np.array_equal(out[(3, "5")], out[(1, "")])


def test_config5_flow_file_potential_fp32_vs_fp64(wa):
    """BASELINE config #5 in miniature: a user potential given at low resolution (as a file would
    hold it), trilinearly upsampled on the device (input.rs:667-716), solved with fp32 storage and
    cross-checked against fp64: relative energy error <= 1e-5, |norm2 - 1| <= 1e-5."""
    n_src, n = 16, 64
    ax = (np.arange(n_src) - (n_src - 1) / 2) * (12.8 / n_src)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    src = np.ascontiguousarray(-3.0 / np.cosh(0.6 * np.sqrt(X * X + Y * Y + 2.0 * Z * Z)) ** 2)  # anisotropic Poschl-Teller well
    res = {}
    for dtype in ("f64", "f32", "f32fast"):
        par = wa.Params(n, n, n, dn=0.2, dt=0.008, mass=1.0, dtype=dtype, max_states=1)
        with wa.Context(par) as ctx:
            ctx.set_potential_resampled(src)
            ctx.set_initial_condition("Boolean")
            recs, final, conv = ctx.solve_state(0, 1e-7, 200, max_steps=60000)
            assert conv
            ctx.clone_state_to_phi(0)
            res[dtype] = (final["energy"], ctx.norm2())
    assert res["f64"][0] < -0.5                      # bound state of the well
    assert res["f32"][0] == pytest.approx(res["f64"][0], rel=1e-5)
    assert res["f32"][1] == pytest.approx(1.0, abs=1e-5) and res["f64"][1] == pytest.approx(1.0, abs=1e-12)
    assert res["f32fast"][0] == pytest.approx(res["f64"][0], rel=1e-5)
    assert res["f32fast"][1] == pytest.approx(1.0, abs=1e-5)


@pytest.mark.parametrize("ext", [1, 2, 3])
def test_f32fast_steps_track_fp64(wo, wa, ext):
    """WAFER_F32_FAST: 40 ground-state steps with fp32 stencil arithmetic stay within fp32 rounding
    of the fp64 oracle (per cell 2e-5 of the largest value), on every stencil order"""
    cfg, par = make_pair((70, 33, 40), ext=ext, potential="Coulomb", dn=0.2, dt=0.004, dtype="f32fast")
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    phi = random_phi(cfg, seed=5).astype(np.float32).astype(np.float64)
    want = phi.copy()
    wo.evolve(cfg, 0, a, b, want, [], 40)
    with wa.Context(par) as ctx:
        ctx.set_potential("Coulomb")
        ctx.upload_phi(phi)
        ctx.evolve(0, 40)
        got = ctx.download_phi()
        n2 = ctx.norm2()
    assert np.max(np.abs(got - want)) <= 2e-5 * np.max(np.abs(want))
    assert n2 == pytest.approx(wo.norm2(cfg, want), rel=1e-5)


def test_f32_excited_state_path(wa):
    """fp32 storage through the fused excited-state step (raw overlaps + one apply pass)"""
    res = {}
    for dtype in ("f64", "f32"):
        _, par = make_pair((40, 36, 44), ext=1, potential="Harmonic", dn=0.3, dt=0.018, dtype=dtype, max_states=2)
        with wa.Context(par) as ctx:
            ctx.set_potential("Harmonic")
            ctx.set_initial_condition("Gaussian", seed=5)
            _, f0, c0 = ctx.solve_state(0, 1e-7, 100, max_steps=40000)
            ctx.set_initial_condition("Gaussian", seed=6)
            _, f1, c1 = ctx.solve_state(1, 1e-7, 100, max_steps=40000)
            assert c0 and c1
            res[dtype] = (f0["energy"], f1["energy"])
    assert res["f32"][0] == pytest.approx(res["f64"][0], rel=1e-5)
    assert res["f32"][1] == pytest.approx(res["f64"][1], rel=2e-5)
    assert res["f64"][1] == pytest.approx(2.5, abs=0.05)


# ---------------------------------------------------------------- full BASELINE size
@pytest.mark.skipif(os.environ.get("WAFER_SKIP_BIG") == "1", reason="WAFER_SKIP_BIG=1")
def test_full_size_512_properties(wa):
    """512^3 fp64 (the headline size): size-independent properties.
    V = 0: the discrete sine mode decays by exactly (1 - dt E) per step;
    norm2 follows; Gram-Schmidt leaves <l|phi> = 0 and idempotent normalise."""
    n = 512
    par = wa.Params(n, n, n, dn=0.05, dt=5e-4, mass=1.0, max_states=1)
    ax = np.sin(np.pi * np.arange(1, n + 1) / (n + 1))
    E = 3 * (1 - np.cos(np.pi / (n + 1))) / (par.mass * par.dn ** 2)
    with wa.Context(par) as ctx:
        ctx.set_potential("NoPotential")
        phi = np.zeros(par.padded_shape)
        phi[1:-1, 1:-1, 1:-1] = ax[:, None, None] * ax[None, :, None] * ax[None, None, :]
        ctx.upload_phi(phi)
        n0 = ctx.norm2()
        assert n0 == pytest.approx(((n + 1) / 2) ** 3, rel=1e-12)
        obs = ctx.observables()
        assert obs["energy"] == pytest.approx(E * obs["norm2"], rel=1e-9)
        ctx.evolve(0, 20)
        assert ctx.norm2() == pytest.approx(n0 * (1 - par.dt * E) ** 40, rel=1e-11)
        got = ctx.download_phi()
        assert np.allclose(got, (1 - par.dt * E) ** 20 * phi, rtol=0, atol=1e-13)
        assert not got[0].any() and not got[:, 0].any() and not got[:, :, -1].any()  # frame stays 0
        ctx.push_state()
        ctx.set_initial_condition("Boolean")
        ctx.normalise(ctx.norm2())
        assert ctx.norm2() == pytest.approx(1.0, abs=1e-12)
        # Gram-Schmidt at the headline size (grid.rs:477-492): the stored state is the evolved sine mode
        # `got` (not normalised: <l|l> = c), so one projection leaves <l|phi'> = s (1 - c) with s = <l|phi>,
        # and a second one multiplies that by (1 - c) again -- modified Gram-Schmidt exactly as the
        # reference applies it, whatever the norm of the stored state
        c = float(np.sum(got * got))
        b0 = ctx.download_phi()
        s = float(np.sum(got * b0))
        del b0
        ctx.orthogonalise(1)
        p1 = ctx.download_phi()
        assert float(np.sum(got * p1)) == pytest.approx(s * (1 - c), rel=1e-9, abs=1e-13 * abs(s) * c)
        ctx.orthogonalise(1)
        p2 = ctx.download_phi()
        assert float(np.sum(got * p2)) == pytest.approx(s * (1 - c) ** 2, rel=1e-9, abs=1e-13 * abs(s) * c * c)
        del p1, p2
        # ... and with an orthoNORMAL store the projection annihilates the overlap and is idempotent
        ctx.clear_states()
        lower = got / np.sqrt(c)
        ctx.load_state(0, lower)
        ctx.set_initial_condition("Boolean")
        ctx.normalise(ctx.norm2())
        ctx.orthogonalise(1)
        p1 = ctx.download_phi()
        assert abs(float(np.sum(lower * p1))) < 1e-13 * float(np.sqrt(np.sum(p1 * p1)))
        ctx.orthogonalise(1)
        p2 = ctx.download_phi()
        assert ulp_diff(p1[1:-1, 1:-1, 1:-1], p2[1:-1, 1:-1, 1:-1]) <= 1 or np.max(np.abs(p2 - p1)) < 1e-16
        assert not p2[0].any() and not p2[:, -1].any()                  # the frame is still zero
        del got, phi, lower, p1, p2


@pytest.mark.skipif(os.environ.get("WAFER_SKIP_BIG") == "1", reason="WAFER_SKIP_BIG=1")
def test_full_size_512_coulomb_three_steps_bit_exact(wo, wa):
    """BASELINE config #3's grid and potential: three steps of the 512^3
    Coulomb problem, every cell compared with the oracle"""
    cfg, par = make_pair((512, 512, 512), ext=1, potential="Coulomb", dn=0.05, dt=5e-4, mass=1.0)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    phi = wo.initial_condition(cfg, "Boolean")
    wo.evolve(cfg, 0, a, b, phi, [], 3)
    with wa.Context(par) as ctx:
        ctx.set_potential("Coulomb")
        ctx.set_initial_condition("Boolean")
        ctx.evolve(0, 3)
        got = ctx.download_phi()
        assert np.array_equal(got, phi)
        obs, want = ctx.observables(), wo.observables(cfg, v, phi)
        for k in ("energy", "norm2", "r2"):
            assert obs[k] == pytest.approx(want[k], rel=REL_SUM)


@pytest.mark.parametrize("seed", [21, 22])
def test_randomised_parity_sweep(seed):
    """tests/fuzz_parity.py: random shapes (1-cell axes included), stencil orders, potentials, kernel
    variants and step counts; ground state bit for bit, excited states with random stores to 1e-10"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_parity.py")], capture_output=True, text=True,
                       env=dict(os.environ, N="80", SEED=str(seed)), timeout=600)
    assert r.returncode == 0 and "bad = 0" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
