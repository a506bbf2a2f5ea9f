"""The oracle's z-window functions (oracle/wafer_oracle.h, round 6) against its full-array functions: the windows that
tests/test_gpu_fullsize.py compares 1024^3 / 2048^3 runs with are slices of what the full-array oracle would return.  The
full-array functions are implemented AS the window [0, pz), so this pins the window arithmetic (offsets, valid planes), not a
second copy of the formulas."""
import numpy as np
import pytest

from oracle import wafer_oracle as wo


@pytest.mark.parametrize("ext", [1, 2, 3])
@pytest.mark.parametrize("potential", ["SimpleCornell", "FullCornell", "QuadWell"])
def test_windows_are_slices_of_the_full_arrays(ext, potential):
    wo.set_threads(4)
    cfg = wo.Config(20, 18, 40, ext=ext, potential=potential, dn=0.3, dt=0.005, mass=2.35, sig=0.223)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    pz = cfg.padded_shape[2]
    for ic in ("Boolean", "Gaussian", "Coulomb", "Constant"):
        full = wo.initial_condition(cfg, ic, seed=5)
        for zp0, zc in ((0, 20), (10, 26), (pz - 22, 22), (0, pz)):
            # (the Coulomb start is 0/0 at the centre of an even padded grid, config.rs:655-666: NaN in both)
            assert np.array_equal(wo.initial_condition_zwindow(cfg, ic, zp0, zc, seed=5), full[:, :, zp0:zp0 + zc], equal_nan=True)
    for zp0, zc in ((0, 20), (10, 26), (pz - 22, 22)):
        vw = wo.potential_generate_zwindow(cfg, zp0, zc)
        assert np.array_equal(vw, v[:, :, zp0:zp0 + zc])
        aw, bw = wo.ab_n(cfg.dt, vw)
        assert np.array_equal(aw, a[:, :, zp0:zp0 + zc]) and np.array_equal(bw, b[:, :, zp0:zp0 + zc])
    with pytest.raises(ValueError):
        wo.potential_generate_zwindow(cfg, pz - 3, 4)


@pytest.mark.parametrize("ext,steps", [(1, 5), (1, 3), (2, 3), (3, 2)])
def test_a_window_evolved_on_its_own_equals_the_global_run_on_its_valid_planes(ext, steps):
    wo.set_threads(4)
    cfg = wo.Config(20, 18, 40, ext=ext, potential="SimpleCornell", dn=0.3, dt=0.005, mass=2.35, sig=0.223)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    phi = wo.initial_condition(cfg, "Boolean")
    ref = phi.copy()
    wo.evolve(cfg, 0, a, b, ref, [], steps)
    pz = cfg.padded_shape[2]
    for zp0, zc in ((0, 20), (10, 26), (pz - 22, 22)):
        aw, bw = wo.ab_n(cfg.dt, wo.potential_generate_zwindow(cfg, zp0, zc))
        pw = wo.initial_condition_zwindow(cfg, "Boolean", zp0, zc)
        lo, hi = wo.evolve_zwindow(cfg, zp0, aw, bw, pw, steps)
        assert (lo == 0) == (zp0 == 0) and (hi == zc) == (zp0 + zc == pz) and hi - lo >= zc - 2 * steps * ext
        assert np.array_equal(pw[:, :, lo:hi], ref[:, :, zp0 + lo:zp0 + hi])
        if lo > 0:      # ... and the planes declared invalid really are (the bound is tight)
            assert not np.array_equal(pw[:, :, lo - 1], ref[:, :, zp0 + lo - 1])
        if hi < zc:
            assert not np.array_equal(pw[:, :, hi], ref[:, :, zp0 + hi])


def test_storage_rounding_between_steps_is_applied_to_every_step():
    wo.set_threads(4)
    cfg = wo.Config(12, 10, 30, ext=1, potential="Harmonic", dn=0.3, dt=0.005)
    v = wo.potential_generate(cfg).astype(np.float32).astype(np.float64)
    a, b = wo.ab(cfg, v)
    phi = wo.initial_condition(cfg, "Boolean")
    ref = phi.copy()
    for _ in range(3):
        wo.evolve(cfg, 0, a, b, ref, [], 1)
        ref = ref.astype(np.float32).astype(np.float64)
    pw = phi.copy()
    lo, hi = wo.evolve_zwindow(cfg, 0, a, b, pw, 3, storage=np.float32)
    assert (lo, hi) == (0, 32) and np.array_equal(pw, ref)


def test_trilerp_window_is_a_slice_of_the_full_resample():
    wo.set_threads(4)
    src = np.random.default_rng(1).standard_normal((9, 7, 8))
    full = wo.trilerp_resize(src, (30, 28, 33), basis=(32, 30, 35))
    for z0, zc in ((0, 5), (11, 9), (28, 5)):
        assert np.array_equal(wo.trilerp_resize_zwindow(src, (30, 28), z0, zc, (32, 30, 35)), full[:, :, z0:z0 + zc])
