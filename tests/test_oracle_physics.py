"""Pins for the oracle functions no reference unit test covers (evolve,
compute_observables, a/b, potentials, initial conditions): discrete analytic
eigenpairs, an independent numpy restatement, continuum known answers and the
survey's indicative wafer.yaml run.  CPU only."""
import numpy as np
import pytest

COEFF = {  # grid.rs:582-588 / 608-620 / 642-659: offsets -> weights, and the lead of the denominator
    1: ({1: 1.0}, 6.0, 2.0),
    2: ({1: 16.0, 2: -1.0}, 90.0, 24.0),
    3: ({1: 270.0, 2: -27.0, 3: 2.0}, 1470.0, 360.0),
}


def np_stencil_sum(phi, e):
    """independent vectorised form of the bracketed sum S on the work area"""
    w, centre, _ = COEFF[e]
    n = [s - 2 * e for s in phi.shape]
    c = phi[e:e + n[0], e:e + n[1], e:e + n[2]]
    s = -centre * c
    for off, wt in w.items():
        for ax in range(3):
            for sgn in (+1, -1):
                sl = [slice(e, e + n[0]), slice(e, e + n[1]), slice(e, e + n[2])]
                sl[ax] = slice(e + sgn * off, e + sgn * off + n[ax])
                s = s + wt * phi[tuple(sl)]
    return s


def np_step(cfg, a, b, phi):
    e = cfg.ext
    den = COEFF[e][2] * cfg.dn * cfg.dn * cfg.mass
    c = phi[e:-e, e:-e, e:-e]
    return c * a[e:-e, e:-e, e:-e] + b[e:-e, e:-e, e:-e] * cfg.dt * np_stencil_sum(phi, e) / den


def sine_mode(cfg, n):
    e = cfg.ext
    phi = np.zeros(cfg.padded_shape)
    ax = [np.sin(np.pi * n[d] * np.arange(1, N + 1) / (N + 1)) for d, N in enumerate(cfg.work_shape)]
    phi[e:-e, e:-e, e:-e] = ax[0][:, None, None] * ax[1][None, :, None] * ax[2][None, None, :]
    return phi


@pytest.mark.parametrize("shape,mode", [((9, 7, 11), (1, 1, 1)), ((12, 12, 12), (2, 1, 3)), ((8, 5, 6), (3, 2, 1))])
def test_threepoint_sine_eigenpair(oracle, shape, mode):
    """V=0, R=1: prod sin(pi n i/(N+1)) is an exact eigenvector of the Dirichlet
    operator with E = sum (1-cos(pi n/(N+1)))/(m dn^2) (SURVEY 8c pin 1)."""
    cfg = oracle.Config(*shape, ext=1, potential="NoPotential", dn=0.3, dt=0.01, mass=1.7)
    v = oracle.potential_generate(cfg)
    a, b = oracle.ab(cfg, v)
    assert np.all(a == 1.0) and np.all(b == 1.0)
    phi = sine_mode(cfg, mode)
    E = sum((1 - np.cos(np.pi * n / (N + 1))) for n, N in zip(mode, shape)) / (cfg.mass * cfg.dn ** 2)
    obs = oracle.observables(cfg, v, phi)
    assert obs["energy"] == pytest.approx(E * obs["norm2"], rel=1e-12)
    work = oracle.stencil_step(cfg, a, b, phi)
    assert np.allclose(work, (1 - cfg.dt * E) * phi[1:-1, 1:-1, 1:-1], rtol=0, atol=1e-14)
    before = phi.copy()
    oracle.evolve(cfg, 0, a, b, phi, [], 5)
    assert np.allclose(phi, (1 - cfg.dt * E) ** 5 * before, rtol=0, atol=1e-13)


@pytest.mark.parametrize("ext", [1, 2, 3])
@pytest.mark.parametrize("pot", ["Harmonic", "Coulomb", "SimpleCornell"])
def test_step_and_observables_vs_numpy(oracle, ext, pot):
    """one stencil pass and the four sums against an independent numpy form"""
    rng = np.random.default_rng(7 + ext)
    cfg = oracle.Config(11, 8, 13, ext=ext, potential=pot, dn=0.2, dt=0.004, mass=1.3, sig=0.223)
    v = oracle.potential_generate(cfg)
    a, b = oracle.ab(cfg, v)
    assert np.allclose(b, 1 / (1 + cfg.dt * v / 2), rtol=1e-15)
    assert np.allclose(a, (1 - cfg.dt * v / 2) / (1 + cfg.dt * v / 2), rtol=1e-15)
    e = ext
    phi = np.zeros(cfg.padded_shape)
    phi[e:-e, e:-e, e:-e] = rng.standard_normal(cfg.work_shape)
    work = oracle.stencil_step(cfg, a, b, phi)
    assert np.allclose(work, np_step(cfg, a, b, phi), rtol=1e-12, atol=1e-13)
    kind, scalar, arr = oracle.potential_sub(cfg)
    obs = oracle.observables(cfg, v, phi, (kind, scalar, arr))
    c = phi[e:-e, e:-e, e:-e]
    den = COEFF[e][2] * cfg.dn ** 2 * cfg.mass
    assert obs["energy"] == pytest.approx(np.sum(v[e:-e, e:-e, e:-e] * c * c - c * np_stencil_sum(phi, e) / den), rel=1e-11)
    assert obs["norm2"] == pytest.approx(np.sum(c * c), rel=1e-13)
    i, j, k = np.meshgrid(*[np.arange(n) - (n + 1) / 2 for n in cfg.work_shape], indexing="ij")
    assert obs["r2"] == pytest.approx(np.sum(c * c * (i * i + j * j + k * k)), rel=1e-13)
    if pot == "SimpleCornell":
        assert kind == 1 and scalar == 4 * cfg.mass
        assert obs["v_infinity"] == pytest.approx(scalar * np.sum(c * c), rel=1e-13)
    else:
        assert kind == 0 and obs["v_infinity"] == 0.0


def test_evolve_equals_repeated_steps_bitwise(oracle):
    cfg = oracle.Config(10, 9, 8, ext=2, potential="Harmonic", dn=0.2, dt=0.004, mass=1.0)
    v = oracle.potential_generate(cfg)
    a, b = oracle.ab(cfg, v)
    phi = oracle.initial_condition(cfg, "Boolean")
    ref = phi.copy()
    for _ in range(7):
        ref[2:-2, 2:-2, 2:-2] = oracle.stencil_step(cfg, a, b, ref)
    oracle.evolve(cfg, 0, a, b, phi, [], 7)
    assert np.array_equal(phi, ref)
    # screen_update = 0 still takes one step (grid.rs:682-685)
    p0 = oracle.initial_condition(cfg, "Boolean")
    p1 = p0.copy()
    oracle.evolve(cfg, 0, a, b, p0, [], 0)
    oracle.evolve(cfg, 0, a, b, p1, [], 1)
    assert np.array_equal(p0, p1)


def test_boolean_and_constant_ic(oracle):
    cfg = oracle.Config(6, 5, 7, ext=2)
    phi = oracle.initial_condition(cfg, "Boolean")
    i, j, k = np.meshgrid(*[np.arange(s) for s in cfg.padded_shape], indexing="ij")
    want = ((i % 2) * (j % 2) * (k % 2)).astype(float)
    want[:2] = want[-2:] = 0
    want[:, :2] = want[:, -2:] = 0
    want[:, :, :2] = want[:, :, -2:] = 0
    assert np.array_equal(phi, want)


def test_potential_centres_and_special_axes(oracle):
    """potentials use the PADDED index with the unpadded centre (potential.rs:52-53, 366-371)"""
    cfg = oracle.Config(8, 6, 10, ext=2, potential="Harmonic", dn=0.5)
    v = oracle.potential_generate(cfg)
    i, j, k = np.meshgrid(*[np.arange(s, dtype=float) for s in cfg.padded_shape], indexing="ij")
    r2 = (i - 4.5) ** 2 + (j - 3.5) ** 2 + (k - 5.5) ** 2
    assert np.array_equal(v, (0.5 * np.sqrt(r2)) ** 2 / 2)
    cfg.potential = "QuadWell"  # z-special: potential.rs:202-211
    v = oracle.potential_generate(cfg)
    inside = (i > 2) & (i <= 6) & (j > 1) & (j <= 4) & (k > 3) & (k <= 6)
    assert np.array_equal(v, np.where(inside, -10.0, 0.0))
    cfg.potential = "ElipticalCoulomb"
    v = oracle.potential_generate(cfg)
    r = 0.5 * np.sqrt((i - 4.5) ** 2 + (j - 3.5) ** 2 + ((k - 5.5) * 2) ** 2)
    assert np.array_equal(v, np.where(r < 0.5, 0.0, -1.0 / r + 1.0 / 0.5))
    cfg.potential = "FromFile"
    with pytest.raises(ValueError):
        oracle.potential_generate(cfg)


def test_harmonic_spectrum(oracle):
    """continuum pin: 3D oscillator (m = omega = 1): 1.5 then 2.5, to O(dn^2)"""
    cfg = oracle.Config(32, 32, 32, ext=1, potential="Harmonic", dn=0.4, dt=0.032, mass=1.0)
    v = oracle.potential_generate(cfg)
    a, b = oracle.ab(cfg, v)
    phi = oracle.initial_condition(cfg, "Gaussian", seed=3)
    store = []
    energies = []
    for wnum in range(2):
        if wnum:
            phi = store[-1].copy()  # grid.rs:95
        recs, conv = oracle.solve(cfg, wnum, v, a, b, phi, store, 1e-9, 100, max_steps=200000)
        assert conv
        energies.append(recs[-1]["energy"] / recs[-1]["norm2"])
        assert oracle.norm2(cfg, phi) == pytest.approx(1.0, abs=1e-12)
        store.append(phi.copy())
    assert energies[0] == pytest.approx(1.5, abs=0.02)
    assert energies[1] == pytest.approx(2.5, abs=0.04)
    assert abs(np.sum(store[0] * store[1])) < 1e-10


def test_wafer_yaml_ground_state(oracle):
    """the shipped wafer.yaml (50^3 Harmonic, Boolean IC): SURVEY section 6 /
    BASELINE.md section 5 give step 18000, E0 = 3.56925, r_rms ~ 16.09 from an
    independent numpy restatement of grid.rs:50-246."""
    cfg = oracle.Config(50, 50, 50, ext=1, potential="Harmonic", dn=0.01, dt=3e-5, mass=15.9994)
    v = oracle.potential_generate(cfg)
    a, b = oracle.ab(cfg, v)
    phi = oracle.initial_condition(cfg, "Boolean")
    recs, conv = oracle.solve(cfg, 0, v, a, b, phi, [], 1e-4, 1000)
    assert conv and recs[-1]["step"] == 18000
    assert recs[-1]["energy"] / recs[-1]["norm2"] == pytest.approx(3.56925, abs=5e-6)
    assert np.sqrt(recs[-1]["r2"] / recs[-1]["norm2"]) == pytest.approx(16.09, abs=5e-3)
    # analytic: 3(1-cos(pi/51))/(m dn^2) + <V>
    assert recs[-1]["energy"] / recs[-1]["norm2"] == pytest.approx(3.55639 + 0.01275, abs=2e-4)


@pytest.mark.parametrize("kind", ["AboutZ", "AntisymAboutZ", "AboutY", "AntisymAboutY"])
@pytest.mark.parametrize("shape", [(5, 6, 7), (4, 7, 6)])
def test_symmetrise_follows_the_reference_loops(oracle, kind, shape):
    wo = oracle
    """config.rs:691-728 restated twice: the oracle's literal loops against an independent numpy
    evaluation of what those loops do -- cells up to h = (3+n)/2 times sign, cells above mirrored
    about n+4 (half a cell below the centre), the last work cell takes the frame's zero."""
    cfg = wo.Config(*shape, ext=3, potential="Harmonic", dn=0.1, dt=0.001, mass=1.0)
    rng = np.random.default_rng(3)
    phi = np.zeros(cfg.padded_shape)
    phi[3:-3, 3:-3, 3:-3] = rng.standard_normal(shape)
    got = phi.copy()
    wo.symmetrise(cfg, kind, got)
    sign = -1.0 if kind.startswith("Antisym") else 1.0
    axis = 2 if kind.endswith("Z") else 1
    n = shape[axis]
    h = (3 + n) // 2
    want = phi.copy()
    src = np.moveaxis(phi, axis, 0)
    dst = np.moveaxis(want, axis, 0)      # a view: writes land in `want`
    for s_ in range(3, 3 + n + 1):
        t = s_ if s_ <= h else n + 4 - s_
        if t == s_:
            dst[s_] = sign * src[s_]
        elif t >= 3:
            dst[s_] = sign * (sign * src[t])
        else:
            dst[s_] = sign * src[t]
    assert np.array_equal(got, want)
    work = np.moveaxis(got, axis, 0)[3:3 + n]
    assert not work[-1].any()                                  # last work cell along the axis: zero
    assert np.array_equal(work[n - 2], sign * sign * np.moveaxis(phi, axis, 0)[3])   # s = n+1 mirrors s = 3
    # applying it twice changes nothing more for the symmetric kinds
    again = got.copy()
    wo.symmetrise(cfg, kind, again)
    if sign > 0:
        assert np.array_equal(again, got)


def test_symmetrise_needs_the_seven_point_frame(oracle):
    wo = oracle
    cfg = wo.Config(5, 5, 5, ext=1, potential="Harmonic", dn=0.1, dt=0.001, mass=1.0)
    with pytest.raises(ValueError):
        wo.symmetrise(cfg, "AboutZ", np.zeros(cfg.padded_shape))
    phi = np.ones(cfg.padded_shape)
    wo.symmetrise(cfg, "NotConstrained", phi)                  # does nothing, any frame
    assert (phi == 1).all()
