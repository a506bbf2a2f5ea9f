"""The BASELINE configurations at (or next to) their prescribed sizes, each held to the oracle or to a
size-independent property (VERDICT r01 "close the parity holes"):
  #2  256^3 harmonic oscillator, ground state            -- 10 steps, every cell bit for bit
  #3  512^3 Coulomb, excited states (Gram-Schmidt)       -- 3 stored states x 2 steps vs the oracle
  #5  file potential, fp32 path vs fp64 cross-check      -- 512^3, the potential read from a FILE by
                                                            the driver and resampled on the device
plus every built-in potential generated ON THE DEVICE against numpy forms written from the text of
potential.rs (tests/test_oracle_physics.py), which neither the oracle's C nor the HIP kernels saw."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests.gpu_common import make_pair  # noqa: E402
from tests.test_oracle_physics import ALL_POTENTIALS, np_potential, np_potsub  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIG = pytest.mark.skipif(os.environ.get("WAFER_SKIP_BIG") == "1", reason="WAFER_SKIP_BIG=1")


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


@pytest.mark.parametrize("pot", ALL_POTENTIALS)
@pytest.mark.parametrize("shape,ext", [((12, 9, 17), 1), ((7, 10, 8), 2), ((16, 16, 16), 3), ((33, 21, 19), 1)])
def test_device_potentials_vs_numpy_forms_of_the_rs_text(wa, pot, shape, ext):
    """wafer_k_potential / wafer_k_ab / wafer_k_potsub_fullcornell against potential.rs:188-319 restated in
    numpy -- the third, independent reading (odd sizes put a cell on r = 0: the clamps and FullCornell's 0/0)"""
    dn, mass, sig, dt = 0.13, 2.35, 0.223, 1e-3
    par = wa.Params(*shape, dn=dn, dt=dt, mass=mass, sig=sig, central_difference=ext)
    want = np_potential(pot, shape, ext, dn, mass, sig)
    with wa.Context(par) as ctx:
        ctx.set_potential(pot)
        v, a, b = ctx.download_array("v"), ctx.download_array("a"), ctx.download_array("b")
        kind, scalar = ctx.potsub()
        arr = ctx.download_array("potsub") if kind == 2 else None
    if pot in ("Periodic", "FullCornell"):     # device sin / exp / log against numpy's
        assert np.allclose(v, want, rtol=1e-13, atol=1e-13)
    else:
        assert np.array_equal(v, want)
    bw = 1. / (1. + dt * v / 2.)
    assert np.array_equal(b, bw) and np.array_equal(a, (1. - dt * v / 2.) * bw)
    wk, wv = np_potsub(pot, shape, dn, mass, sig)
    assert kind == {"none": 0, "scalar": 1, "array": 2}[wk]
    if wk == "scalar":
        assert scalar == wv
    if wk == "array":
        assert (np.isnan(arr) == np.isnan(wv)).all()
        ok = ~np.isnan(wv)
        assert np.allclose(arr[ok], wv[ok], rtol=1e-13)


@BIG
def test_config2_256_cubed_harmonic_ten_steps_bit_exact(wo, wa):
    """BASELINE config #2: 256^3 harmonic oscillator, ground state, fp64 (dn 0.05, dt 5e-4)"""
    cfg, par = make_pair((256, 256, 256), ext=1, potential="Harmonic", dn=0.05, dt=5e-4, mass=1.0)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    phi = wo.initial_condition(cfg, "Boolean")
    wo.evolve(cfg, 0, a, b, phi, [], 10)
    with wa.Context(par) as ctx:
        assert ctx.stencil_kernel_name() == "wafer_k_step3_fused"
        ctx.set_potential("Harmonic")
        ctx.set_initial_condition("Boolean")
        ctx.evolve(0, 10)
        got = ctx.download_phi()
        assert np.array_equal(got, phi)
        obs, want = ctx.observables(), wo.observables(cfg, v, phi)
        for k in ("energy", "norm2", "r2"):
            assert obs[k] == pytest.approx(want[k], rel=1e-12)
        # the solve loop's first block on the same state: E / norm2 of the oracle's own record
        assert obs["energy"] / obs["norm2"] == pytest.approx(want["energy"] / want["norm2"], rel=1e-12)


@BIG
@pytest.mark.parametrize("shape", [(384, 320, 448), (320, 384, 200)])
def test_grids_between_the_powers_of_two_seven_steps_bit_exact(wo, wa, shape):
    """tile counts that do not divide the CUs (72 / 60 tiles per layer): the z-chunk policy picks 7 / 4 chunks per column
    (wafer_pick_zchunk; rounds 1-2 left a second round of 32 workgroups at 384^3), x- and y-ragged tiles, the frame cells
    that are no longer fetched; two three-step passes and a single step against the oracle"""
    cfg, par = make_pair(shape, ext=1, potential="Coulomb", dn=0.05, dt=5e-4, mass=1.0)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    phi = wo.initial_condition(cfg, "Boolean")
    wo.evolve(cfg, 0, a, b, phi, [], 7)
    with wa.Context(par) as ctx:
        assert ctx.stencil_kernel_name() == "wafer_k_step3_fused"
        ctx.set_potential("Coulomb")
        ctx.set_initial_condition("Boolean")
        ctx.evolve(0, 7)
        assert np.array_equal(ctx.download_phi(), phi)
        obs, want = ctx.observables(), wo.observables(cfg, v, phi)
        for k in ("energy", "norm2", "r2"):
            assert obs[k] == pytest.approx(want[k], rel=1e-12)


@BIG
@pytest.mark.parametrize("potential,k", [("Coulomb", 3), ("Coulomb", 1), ("Coulomb", 2), ("SimpleCornell", 1), ("SimpleCornell", 2),
                                         ("SimpleCornell", 3)])
def test_config3_512_cubed_stored_states_two_excited_steps(wo, wa, potential, k, monkeypatch):
    """BASELINE config #3 where its wall time goes (93 % of its steps are excited-state steps): 512^3, k stored
    states, two steps of grid.rs:674-681 (step, renormalise, modified Gram-Schmidt) through the PRODUCTION kernels --
    closed-form V evaluated per cell, raw staging pipeline -- against the oracle: every cell to 1e-13, the sums to
    1e-12.  Coulomb (config #3's potential) and SimpleCornell (config #4's), k = 1, 2, 3.  The stored states are the
    lowest box modes (exactly orthogonal, normalised), phi another one plus a Boolean grid and parts of the stored ones.
    Then the same state through the two-steps-per-pass kernels (wafer_stencil_x2.hip.h): six more steps (two one-step, two
    passes) and seven more (three one-step, two passes), each against the oracle at the same bar."""
    monkeypatch.setenv("WAFER_X2_MAX_K", "3")   # (the default keeps k = 3 on the one-step kernel)
    n = 512
    cfg, par = make_pair((n, n, n), ext=1, potential=potential, dn=0.05, dt=5e-4, mass=1.0, sig=0.223, max_states=3)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    s = [np.sin(np.pi * m * np.arange(1, n + 1) / (n + 1)) * np.sqrt(2.0 / (n + 1)) for m in (1, 2, 3)]

    def mode(mx, my, mz):
        out = np.zeros(cfg.padded_shape)
        out[1:-1, 1:-1, 1:-1] = s[mx - 1][:, None, None] * s[my - 1][None, :, None] * s[mz - 1][None, None, :]
        return out
    lowers = [mode(1, 1, 1), mode(2, 1, 1), mode(1, 1, 2)][:k]
    phi = wo.initial_condition(cfg, "Boolean") * 1e-3 + mode(1, 2, 1) + 0.3 * lowers[0] - 0.2 * lowers[-1]
    with wa.Context(par) as ctx:
        ctx.set_potential(potential)
        for i, l in enumerate(lowers):
            ctx.load_state(i, l)
        ctx.upload_phi(phi)
        for steps, passes in ((2, 0), (6, 2), (7, 2)):
            before = ctx.x2_passes()
            ctx.evolve(k, steps)
            assert ctx.x2_passes() - before == passes
            got = ctx.download_phi()
            n2 = ctx.norm2()
            obs = ctx.observables()
            wo.evolve(cfg, k, a, b, phi, lowers, steps)
            assert np.max(np.abs(got - phi)) <= 1e-13, steps
            assert n2 == pytest.approx(wo.norm2(cfg, phi), rel=1e-12)
            want = wo.observables(cfg, v, phi)
            for key in ("energy", "norm2", "r2"):
                assert obs[key] == pytest.approx(want[key], rel=1e-12)
            for l in lowers:       # orthogonal to every stored state after the step's Gram-Schmidt
                assert abs(float(np.sum(l * got))) < 1e-13
            del got
    del phi, lowers, a, b, v


@BIG
@pytest.mark.parametrize("ext,kernel", [(2, "wafer_k_step2_wide"), (3, "wafer_k_step_lds")])
def test_five_and_seven_point_256_cubed_four_steps_bit_exact(wo, wa, ext, kernel):
    """the default kernels of the wider stencils at a size where every workgroup marches a long column: FivePoint on the
    two-step kernel on 128 x 16 tiles, SevenPoint on the single-step LDS kernel, 256^3 Coulomb x 4 steps, every cell's bits
    against the oracle (grid.rs:593-663); dt = 0.2 dn^2 keeps both inside the true forward-Euler bound"""
    n = 256
    cfg, par = make_pair((n, n, n), ext=ext, potential="Coulomb", dn=0.05, dt=5e-4, mass=1.0)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    phi = wo.initial_condition(cfg, "Boolean")
    wo.evolve(cfg, 0, a, b, phi, [], 4)
    with wa.Context(par) as ctx:
        assert ctx.stencil_kernel_name() == kernel
        ctx.set_potential("Coulomb")
        ctx.set_initial_condition("Boolean")
        ctx.evolve(0, 4)
        got = ctx.download_phi()
        assert np.array_equal(got, phi)
        obs, want = ctx.observables(), wo.observables(cfg, v, phi)
        for key in ("energy", "norm2", "r2"):
            assert obs[key] == pytest.approx(want[key], rel=1e-12)


@BIG
@pytest.mark.parametrize("x2", ["0", "1"])
@pytest.mark.parametrize("potential", ["Coulomb", "SimpleCornell"])
def test_512_cubed_excited_steps_closed_form_and_staging_pipeline_bit_identical(wa, potential, x2, monkeypatch):
    """the production kernels of BASELINE config #3's excited states at full size -- potential evaluated per cell
    (VG), raw staging pipeline with full-vector stores (DEEP) -- against the kernels that stream the stored V on the
    plain prefetch: the checksum of every cell's bits after 5 steps against k = 1, 2, 3 stored states, and the sums"""
    n = 512
    got = {}
    # x2 = 1: the same comparison through the two-steps-per-pass kernels (closed form against streamed V on the same tiles:
    # WAFER_X2_RY=2 keeps one stored state on 128 x 16 tiles on both sides)
    monkeypatch.setenv("WAFER_X2", x2)
    monkeypatch.setenv("WAFER_X2_RY", "2")
    monkeypatch.setenv("WAFER_X2_MAX_K", "3")
    for mode in ("plain", "production"):
        if mode == "plain":
            monkeypatch.setenv("WAFER_VGEN", "0")
            monkeypatch.setenv("WAFER_XF_DEEP", "0")
        else:
            monkeypatch.delenv("WAFER_VGEN", raising=False)
            monkeypatch.delenv("WAFER_XF_DEEP", raising=False)
        par = wa.Params(n, n, n, dn=0.05, dt=5e-4, mass=1.0, sig=0.223, max_states=3)
        out = []
        with wa.Context(par) as ctx:
            ctx.set_potential(potential)
            for i in range(3):
                ctx.set_initial_condition("Gaussian", seed=11 + i)
                ctx.normalise(ctx.norm2())
                ctx.push_state()
            for k in (1, 2, 3):
                ctx.set_initial_condition("Gaussian", seed=20 + k)
                ctx.evolve(k, 4)
                ctx.evolve(k, 1)
                out.append((ctx.checksum(), ctx.norm2()))
        got[mode] = out
    assert got["plain"] == got["production"]


@BIG
def test_config5_512_cubed_file_potential_fp32_vs_fp64(tmp_path):
    """BASELINE config #5's cross-check at the size SURVEY.md 8d prescribes: the SAME user potential --
    a 64^3 array in a FILE (./input/potential.csv, the reference's `i,j,k,data` rows), read by the
    driver and trilinearly resampled to 512^3 ON THE DEVICE (input.rs:149-176, 667-716) -- solved in
    fp64, with fp32 storage (fp64 arithmetic) and with fp32 storage + fp32 arithmetic in the stencil steps (f32fast):
    relative energy error <= 1e-5, the same number of blocks to converge."""
    cli = os.path.join(ROOT, "wafer_amd", "wafer-hip")
    n_src, n = 64, 512
    ax = (np.arange(n_src) - (n_src - 1) / 2) * (12.8 / n_src)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    src = -3.0 / np.cosh(0.6 * np.sqrt(X * X + Y * Y + 2.0 * Z * Z)) ** 2      # anisotropic Poschl-Teller well (gen_potential.py:45-60 in spirit)
    inp = tmp_path / "input"
    inp.mkdir()
    I, J, K = np.meshgrid(*[np.arange(n_src)] * 3, indexing="ij")
    np.savetxt(inp / "potential.csv", np.column_stack([I.ravel(), J.ravel(), K.ravel(), src.ravel()]),
               fmt=["%d", "%d", "%d", "%.17g"], delimiter=",")
    res = {}
    for dtype in ("f64", "f32", "f32fast"):
        (tmp_path / f"{dtype}.yaml").write_text(f"""project_name: "config5 {dtype}"
grid:
    size:
        x: {n}
        y: {n}
        z: {n}
    dn: 0.025
    dt: 1.25e-4
tolerance: 1e-6
central_difference: ThreePoint
max_steps: 400000
wavenum: 0
wavemax: 0
potential: FromFile
mass: 1.0
init_condition: Boolean
sig: 1.0
init_symmetry: NotConstrained
output:
    screen_update: 1000
    file_type: Csv
    save_wavefns: false
    save_potential: false
gpu:
    dtype: {dtype}
""")
        r = subprocess.run([cli, "-c", str(tmp_path / f"{dtype}.yaml"), "--output-dir", str(tmp_path / f"out_{dtype}"),
                            "--input-dir", str(inp)], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        assert "Interpolating from [66, 66, 66] to requested size of [514, 514, 514]" in r.stderr
        e = float(re.search(r"Ground state energy = ([0-9.eE+-]+)", r.stdout).group(1))
        od = tmp_path / f"out_{dtype}" / os.listdir(tmp_path / f"out_{dtype}")[0]
        rows = [l for l in r.stdout.splitlines() if re.match(r"^\s+│\s*[0-9.]+ │", l)]
        res[dtype] = (e, len(rows), od)
    assert res["f64"][0] < -0.5                                       # a bound state of the well
    # fp32 STORAGE (fp64 arithmetic) and the all-fp32 ground-state steps (`f32fast`: config #5's throughput setting) both
    # have to hold the prescribed cross-check
    for dtype in ("f32", "f32fast"):
        assert res[dtype][0] == pytest.approx(res["f64"][0], rel=1e-5), dtype
        assert abs(res[dtype][1] - res["f64"][1]) <= 1, dtype         # the same number of blocks to converge


DISPATCH_CASES = [  # (shape, ext, dtype, slab kwargs)
    ((128, 128, 128), 1, "f64", {}), ((64, 64, 64), 1, "f64", {}), ((128, 128, 128), 1, "f32", {}), ((128, 128, 128), 1, "f32fast", {}),
    ((128, 64, 64), 2, "f64", {}), ((128, 64, 64), 2, "f32", {}), ((128, 64, 64), 3, "f64", {}), ((128, 64, 64), 3, "f32", {}),
    ((128, 128, 128), 1, "f64", dict(z_begin=32, z_count=32, halo_depth=3)), ((128, 128, 128), 1, "f64", dict(z_begin=32, z_count=32, halo_depth=2)),
    ((128, 64, 64), 2, "f64", dict(z_begin=16, z_count=16, halo_depth=4)), ((128, 64, 64), 2, "f64", dict(z_begin=16, z_count=16, halo_depth=2)),
]


@pytest.mark.parametrize("shape,ext,dtype,slab", DISPATCH_CASES)
def test_dispatch_description_is_what_runs(wa, shape, ext, dtype, slab):
    """wafer_diag_dispatch (what tools/dispatch_table.py tabulates into profiles/r06_dispatch_table.md) against what then ran: the
    kernel family and the steps per pass of a ground-state evolve (wafer_stencil_kernel_instance, wafer_stencil_steps_per_launch),
    and whether excited-state steps took the two-steps-per-pass kernel (wafer_diag_x2_passes) -- on undecomposed grids and on a
    slab as its own neighbour, for every stencil and storage type"""
    par = wa.Params(*shape, dn=0.2, dt=0.004, central_difference=ext, dtype=dtype, max_states=3, **slab)
    with wa.Context(par) as ctx:
        if slab:   # a self-loop: the slab's own planes stand in for the neighbours' (what is checked is the dispatch, not the physics)
            import ctypes as C
            from tests.test_gpu_slab import _loaded_hip_runtime
            hip = C.CDLL(_loaded_hip_runtime())
            hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]

            def halo(slo, shi, rlo, rhi, nbytes, stream):
                if rlo and shi:
                    assert hip.hipMemcpyAsync(rlo, shi, nbytes, 3, stream) == 0
                if rhi and slo:
                    assert hip.hipMemcpyAsync(rhi, slo, nbytes, 3, stream) == 0
                return 0
            ctx.set_comm_hooks(halo, lambda ptr, count, stream: 0)
        ctx.set_potential("Coulomb")
        d = ctx.dispatch(0)
        assert d["stencil"] == ext and d["dtype"] == dtype and d["kernel"] == ctx.stencil_kernel_name()
        assert d["steps_per_pass"] == ctx.steps_per_launch() and d["ghost_planes_per_pass"] == d["steps_per_pass"] * ext
        ctx.set_initial_condition("Boolean")
        ctx.evolve(0, 6)
        assert ctx.stencil_kernel_instance().startswith(d["kernel"])            # the kernel that ran
        if d["kernel"] == "wafer_k_step3_fused":
            assert d["tile"] == ("256x16" if dtype == "f32fast" else "128x16")
        for k in (1, 2, 3):
            ctx.set_initial_condition("Gaussian", seed=k)
            ctx.normalise(ctx.norm2())
            ctx.orthogonalise(k - 1)
            ctx.normalise(ctx.norm2())
            ctx.push_state()
            if dtype == "f32fast":
                continue
            d = ctx.dispatch(k)
            before = ctx.x2_passes()
            ctx.set_initial_condition("Gaussian", seed=10 + k)
            ctx.evolve(k, 8)
            assert np.isfinite(ctx.norm2())
            took_x2 = ctx.x2_passes() > before
            assert took_x2 == (d["kernel"] == "wafer_k_xstep2"), (k, d)
            assert d["steps_per_pass"] == (2 if took_x2 else 1)
            if took_x2:
                # (fp32 storage since round 6, there on the tall tile at every k: the stored states' LDS queue is float)
                assert ext == 1 and dtype in ("f64", "f32") and d["tile"] == ("128x16" if (k <= 2 or dtype == "f32") else "128x8")
            else:
                assert d["kernel"] == "wafer_k_step_lds" and d["nlow"] == k and d["waves"] in (4, 8)
        d5 = ctx.dispatch(5)                                     # more stored states than the fused overlaps carry (WAFER_MAX_LOW = 4)
        assert d5["kernel"] == "wafer_k_step_lds" and d5["nlow"] == 0 and "gram_schmidt" in d5["then"]


def test_committed_dispatch_table_is_current(wa):
    """profiles/r06_dispatch_table.json (tools/dispatch_table.py) still says what the library says, row by row, for the grids that
    cost nothing to create"""
    import json
    path = os.path.join(ROOT, "profiles", "r06_dispatch_table.json")
    if not os.path.exists(path):
        pytest.skip("table not generated yet")
    table = [r for r in json.load(open(path)) if "error" not in r and r["grid"] in ("64^3", "256^3")]
    assert len(table) > 60
    shapes = {"64^3": (64, 64, 64), "256^3": (256, 256, 256)}
    ctxs = {}
    try:
        for r in table:
            key = (r["grid"], r["stencil"], r["dtype"])
            if key not in ctxs:
                ctxs[key] = wa.Context(wa.Params(*shapes[r["grid"]], dn=0.05, dt=5e-4, central_difference=r["stencil"], dtype=r["dtype"], max_states=1))
                ctxs[key].set_potential("Coulomb")
            now = ctxs[key].dispatch(r["wnum"])
            assert {k: v for k, v in r.items() if k != "grid"} == now, (r, now)
    finally:
        for c in ctxs.values():
            c.close()
