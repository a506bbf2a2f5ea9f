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


@pytest.mark.parametrize("ext,wnum", [(1, 1), (1, 3), (2, 2), (3, 1)])
@pytest.mark.parametrize("pot", ["Harmonic", "Coulomb"])
def test_excited_state_evolve_vs_numpy_form_of_the_rs_text(oracle, ext, wnum, pot):
    """grid.rs:674-681 read a third time (neither the oracle's C nor the engine's HIP): after EVERY step the
    norm squared over the WORK area (get_norm_squared, :454-457), the whole array divided by its square root
    (normalise_wavefunction, :465-468: normalise comes first), then for each stored state IN ORDER the overlap
    with the ALREADY UPDATED w summed over the whole array and w -= lower * overlap
    (orthogonalise_wavefunction, :477-492: modified Gram-Schmidt).  The stored states are deliberately not
    orthogonal to each other, so classical and modified Gram-Schmidt -- or projection before normalisation --
    are told apart (asserted below)."""
    rng = np.random.default_rng(100 * ext + wnum)
    cfg = oracle.Config(9, 7, 8, ext=ext, potential=pot, dn=0.3, dt=0.01, mass=1.3)
    v = oracle.potential_generate(cfg)
    a, b = oracle.ab(cfg, v)
    e = ext

    def rand():
        out = np.zeros(cfg.padded_shape)
        out[e:-e, e:-e, e:-e] = rng.standard_normal(cfg.work_shape)
        return out
    lowers = []
    for j in range(wnum):
        l = rand() + (0.5 * lowers[0] if lowers else 0.0)      # correlated on purpose
        l /= np.sqrt(np.sum(l * l))
        lowers.append(np.ascontiguousarray(l))
    phi = rand()
    want = phi.copy()
    for _ in range(6):
        want[e:-e, e:-e, e:-e] = np_step(cfg, a, b, want)       # the stencil pass and the copy back (:562-673)
        norm2 = np.sum(want[e:-e, e:-e, e:-e] ** 2)             # :675-678
        want = want / np.sqrt(norm2)                            # :679
        for l in lowers[:wnum]:                                 # :680
            want = want - l * np.sum(l * want)
    got = phi.copy()
    oracle.evolve(cfg, wnum, a, b, got, lowers, 6)
    assert np.allclose(got, want, rtol=0, atol=1e-13)
    # the orderings this must NOT be: classical Gram-Schmidt (all overlaps from the same w) ...
    if wnum >= 2:
        cgs = phi.copy()
        for _ in range(6):
            cgs[e:-e, e:-e, e:-e] = np_step(cfg, a, b, cgs)
            cgs = cgs / np.sqrt(np.sum(cgs[e:-e, e:-e, e:-e] ** 2))
            cgs = cgs - sum(l * np.sum(l * cgs) for l in lowers[:wnum])
        assert np.max(np.abs(cgs - want)) > 1e-8      # five orders above the bar of the comparison above
    # ... or the projection before the normalisation
    pfirst = phi.copy()
    for _ in range(6):
        pfirst[e:-e, e:-e, e:-e] = np_step(cfg, a, b, pfirst)
        for l in lowers[:wnum]:
            pfirst = pfirst - l * np.sum(l * pfirst)
        pfirst = pfirst / np.sqrt(np.sum(pfirst[e:-e, e:-e, e:-e] ** 2))
    assert np.max(np.abs(pfirst - want)) > 1e-8


@pytest.mark.parametrize("wnum,max_steps", [(0, None), (1, None), (0, 40)])
def test_solve_loop_vs_numpy_form_of_the_rs_text(oracle, wnum, max_steps):
    """grid.rs:122-221 read a third time: per block -- observables of the CURRENT phi (energy / norm2 is the row's
    energy, tau = step * dt), normalise by THAT norm2, Gram-Schmidt if wnum > 0, the convergence test on
    |E - E_last| BEFORE anything evolves (so the first row can never converge: E_last starts at f64::MAX), the
    max_steps test as `step > max_steps` AFTER the convergence test, then screen_update steps of evolve.
    Rows (step, tau, E, diff), the stop step and the final phi against the oracle's wo_solve."""
    rng = np.random.default_rng(5 + wnum)
    cfg = oracle.Config(10, 9, 11, ext=1, potential="Harmonic", dn=0.4, dt=0.03, mass=1.0)
    v = oracle.potential_generate(cfg)
    a, b = oracle.ab(cfg, v)
    e, su, tol = 1, 20, 1e-7
    den = COEFF[e][2] * cfg.dn ** 2 * cfg.mass

    def rand():
        out = np.zeros(cfg.padded_shape)
        out[e:-e, e:-e, e:-e] = rng.standard_normal(cfg.work_shape)
        return out
    lowers = []
    if wnum:
        l = sine_mode(cfg, (1, 1, 1))
        lowers = [l / np.sqrt(np.sum(l * l))]
    phi0 = rand()
    # --- the numpy reading
    phi, step, last, rows, conv = phi0.copy(), 0, np.finfo(np.float64).max, [], False
    while True:
        c = phi[e:-e, e:-e, e:-e]
        norm2 = np.sum(c * c)
        energy = np.sum(v[e:-e, e:-e, e:-e] * c * c - c * np_stencil_sum(phi, e) / den)   # :325-332, :405-407
        E = energy / norm2
        phi = phi / np.sqrt(norm2)                                                          # :130
        for l in lowers[:wnum]:                                                            # :133-135
            phi = phi - l * np.sum(l * phi)
        diff = abs(E - last)                                                               # :161
        rows.append((step, step * cfg.dt, E, diff))
        if diff < tol:
            conv = True
            break
        last = E
        if max_steps is not None and step > max_steps:                                     # :209-211
            break
        for _ in range(su):                                                                # evolve: :544-687
            phi[e:-e, e:-e, e:-e] = np_step(cfg, a, b, phi)
            if wnum > 0:
                phi = phi / np.sqrt(np.sum(phi[e:-e, e:-e, e:-e] ** 2))
                for l in lowers[:wnum]:
                    phi = phi - l * np.sum(l * phi)
        step += su
    # --- the oracle
    got = phi0.copy()
    recs, oconv = oracle.solve(cfg, wnum, v, a, b, got, lowers, tol, su, max_steps=max_steps)
    assert oconv == conv and len(recs) == len(rows)
    if max_steps is not None:
        assert not conv and rows[-1][0] == 60          # 0, 20, 40 pass `step > 40`; the block at step 60 stops
    for r, (st, tau, E, diff) in zip(recs, rows):
        assert r["step"] == st and r["tau"] == pytest.approx(tau, rel=1e-15, abs=0)
        assert r["energy"] / r["norm2"] == pytest.approx(E, rel=1e-11)
        assert r["diff"] == pytest.approx(diff, rel=1e-6, abs=1e-12)
    assert np.allclose(got, phi, rtol=0, atol=1e-11)


def test_boolean_and_constant_ic(oracle):
    cfg = oracle.Config(6, 5, 7, ext=2)
    phi = oracle.initial_condition(cfg, "Boolean")
    i, j, k = np.meshgrid(*[np.arange(s) for s in cfg.padded_shape], indexing="ij")
    want = ((i % 2) * (j % 2) * (k % 2)).astype(float)
    want[:2] = want[-2:] = 0
    want[:, :2] = want[:, -2:] = 0
    want[:, :, :2] = want[:, :, -2:] = 0
    assert np.array_equal(phi, want)


@pytest.mark.parametrize("shape,ext", [((6, 5, 7), 2), ((8, 8, 8), 1), ((5, 7, 9), 3)])
def test_coulomb_and_constant_ic_vs_numpy_form_of_the_rs_text(oracle, shape, ext):
    """config.rs:638-665 (generate_coulomb) read from the text: the centre is init_size / 2 of the PADDED array
    (not the (n + 1) / 2 of the potentials), r = dn * sqrt(dx^2 + dy^2 + dz^2), the four hydrogen-like terms;
    an even padded size puts a cell at r = 0 where cos(theta) = 0 / 0 makes the value NaN in the reference too --
    unless the frame zeroes it (config.rs:597-622), which it does not for the centre.  Constant is 0.1 inside the
    frame (config.rs:594)."""
    mass, dn = 1.3, 0.25
    cfg = oracle.Config(*shape, ext=ext, potential="NoPotential", dn=dn, dt=0.004, mass=mass)
    ps = cfg.padded_shape
    i, j, k = np.meshgrid(*[np.arange(n, dtype=float) for n in ps], indexing="ij")
    dx, dy, dz = i - ps[0] / 2.0, j - ps[1] / 2.0, k - ps[2] / 2.0
    with np.errstate(invalid="ignore", divide="ignore"):
        r = dn * np.sqrt(dx ** 2 + dy ** 2 + dz ** 2)
        costheta = dn * dz / r
        cosphi = dn * dx / r
        mr2 = np.exp(-mass * r / 2.0)
        want = (np.exp(-mass * r) + (2.0 - mass * r) * mr2 + mass * r * mr2 * costheta
                + mass * r * mr2 * np.sqrt(1.0 - costheta ** 2) * cosphi)
    e = ext
    for arr in (want,):
        arr[:e] = arr[-e:] = 0
        arr[:, :e] = arr[:, -e:] = 0
        arr[:, :, :e] = arr[:, :, -e:] = 0
    got = oracle.initial_condition(cfg, "Coulomb")
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.isnan(want).sum() == (1 if all(n % 2 == 0 for n in ps) else 0)
    ok = ~np.isnan(want)
    assert np.allclose(got[ok], want[ok], rtol=1e-13, atol=1e-15)
    const = oracle.initial_condition(cfg, "Constant")
    inner = np.zeros(ps)
    inner[e:-e, e:-e, e:-e] = 0.1
    assert np.array_equal(const, inner)


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


# ---------------------------------------------------------------------------------------------
# Independent numpy forms of ALL built-in potentials, written from the text of
# /root/reference/src/potential.rs:188-319 (+ :326-398) -- not from the oracle's C and not from the
# engine's HIP, which share an author.  Vectorised over index grids, so a slip that the C and HIP
# siblings (oracle/wafer_oracle.c:158-243, wafer_setup.hip.h:52-118) have in common would show here.
# ---------------------------------------------------------------------------------------------
def np_alphas(mu):
    """potential.rs:374-391"""
    nf = 2.0
    b0 = 11. - 2. * nf / 3.
    b1 = 51. - 19. * nf / 3.
    b2 = 2857. - 5033. * nf / 9. + 325. * nf * nf / 27.
    r = 2.3
    l = 2. * np.log(mu / r)
    return (4. * np.pi * (1. - 2. * b1 * np.log(l) / (b0 * b0 * l)
                          + 4. * b1 * b1 * ((np.log(l) - 0.5) * (np.log(l) - 0.5) + b2 * b0 / (8. * b1 * b1) - 5.0 / 4.0)
                          / (b0 * b0 * b0 * b0 * l * l)) / (b0 * l))


def np_mu(t):
    """potential.rs:394-398"""
    nf, tc = 2.0, 0.2
    return 1.4 * np.sqrt((1. + nf / 6.) * 4. * np.pi * np_alphas(2. * np.pi * t)) * t * tc


def np_potential(name, n, ext, dn, mass, sig):
    """V on the PADDED index grid (potential.rs:46-62 calls potential() with padded indices; every
    centre is (n + 1) / 2 of the UNPADDED size, potential.rs:366-371)."""
    nx, ny, nz = n
    shape = (nx + 2 * ext, ny + 2 * ext, nz + 2 * ext)
    ix, iy, iz = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")     # integer indices
    fx, fy, fz = ix.astype(float), iy.astype(float), iz.astype(float)
    dx, dy, dz = fx - (nx + 1.) / 2., fy - (ny + 1.) / 2., fz - (nz + 1.) / 2.
    r2 = dx * dx + dy * dy + dz * dz
    with np.errstate(divide="ignore", invalid="ignore"):
        if name == "NoPotential":
            return np.zeros(shape)
        if name in ("Cube", "QuadWell"):      # :191-210, usize arithmetic: integer division
            zlo, zhi = (nz // 4, 3 * nz // 4) if name == "Cube" else (3 * nz // 8, 5 * nz // 8)
            inside = ((ix > nx // 4) & (ix <= 3 * nx // 4) & (iy > ny // 4) & (iy <= 3 * ny // 4)
                      & (iz > zlo) & (iz <= zhi))
            return np.where(inside, -10.0, 0.0)
        if name == "Periodic":                # :211-220
            t = np.sin(2. * np.pi * (fx - 1.) / (nx - 1.)) ** 2
            t = t * np.sin(2. * np.pi * (fy - 1.) / (ny - 1.)) ** 2
            t = t * np.sin(2. * np.pi * (fz - 1.) / (nz - 1.)) ** 2
            return -t + 1.
        if name in ("Coulomb", "ComplexCoulomb"):   # :221-229
            r = dn * np.sqrt(r2)
            return np.where(r < dn, -1. / dn, -1. / r)
        if name == "ElipticalCoulomb":        # :230-240
            dz2 = dz * 2.
            r = dn * np.sqrt(dx * dx + dy * dy + dz2 * dz2)
            return np.where(r < dn, 0.0, -1. / r + 1. / dn)
        if name == "SimpleCornell":           # :241-249
            r = dn * np.sqrt(r2)
            return np.where(r < dn, 4. * mass, (-0.5 * (4. / 3.)) / r + sig * r + 4. * mass)
        if name == "FullCornell":             # :250-269, t = 1, xi = 0
            t, xi = 1.0, 0.0
            r = dn * np.sqrt(r2)
            md = np_mu(t) * (1. + (0.07 * xi ** 0.2) * (1. - dn * dn * dz * dz / (r * r))) * ((1. + xi) ** -0.29)
            full = ((-np_alphas(2. * np.pi * t) * (4. / 3.)) * np.exp(-md * r) / r + sig * (1. - np.exp(-md * r)) / md
                    - (0.8 * sig) / (4. * mass * mass * r) + 4. * mass)
            return np.where(r < dn, 4. * mass, full)
        if name in ("Harmonic", "ComplexHarmonic"):   # :270-274
            r = dn * np.sqrt(r2)
            return r * r / 2.
        if name == "Dodecahedron":            # :275-312: twelve half-spaces in normalised coordinates
            x, y, z = dx / ((nx - 1.) / 2.), dy / ((ny - 1.) / 2.), dz / ((nz - 1.) / 2.)
            A, B, Cc = 12.70820393249937, 11.210068307552588, 14.674169922690343
            D, E, F = 5.605034153776295, 3.23606797749979, 1.2360679774997896
            G, H, I_ = 4.23606797749979, 5.23606797749979, 18.1382715378281
            J, K, Lc = 3.464101615137755, 9.06913576891405, 15.70820393249937
            M, N_, O = 9.70820393249937, 5.605034153776294, 6.47213595499958
            P, Q, S_ = 25.41640786499874, 1.7320508075688772, 8.47213595499958
            inside = ((A + B * x >= Cc * z) & (B * x <= A + Cc * z)
                      & (D * (E * x - F * z) <= 6. * (G + H * y))
                      & (I_ * x + J * z <= A)
                      & (K * x + Lc * y <= A + J * z)
                      & (M * y <= A + N_ * x + Cc * z)
                      & (A + N_ * x + M * y + Cc * z >= 0.)
                      & (Lc * y + J * z <= A + K * x)
                      & (D * (-O * x - F * z) <= P)
                      & (J * z <= K * x + 3. * (G + H * y))
                      & (Q * (E * x + S_ * z) <= 3. * (G + E * y))
                      & (N_ * x + M * y + Cc * z <= A))
            return np.where(inside, -100.0, 0.0)
    raise ValueError(name)


def np_potsub(name, n, dn, mass, sig):
    """pot_sub (potential.rs:134-153, 326-363): ('array', A) on the UNPADDED index grid for FullCornell,
    ('scalar', s) when s > 0, else ('none', None)"""
    if name == "FullCornell":                 # :326-340 -- note the grouping differs from potential()'s md
        nx, ny, nz = n
        ix, iy, iz = np.meshgrid(*[np.arange(s, dtype=float) for s in n], indexing="ij")
        dx, dy, dz = ix - (nx + 1.) / 2., iy - (ny + 1.) / 2., iz - (nz + 1.) / 2.
        r = dn * np.sqrt(dx * dx + dy * dy + dz * dz)
        t, xi = 1.0, 0.0
        with np.errstate(divide="ignore", invalid="ignore"):
            md = np_mu(t) * 1. + (0.07 * xi ** 0.2) * (1. - dn * dn * dz * dz / (r * r)) * ((1. + xi) ** -0.29)
        return "array", sig / md + 4. * mass
    s = {"ElipticalCoulomb": 1. / dn, "SimpleCornell": 4.0 * mass}.get(name, 0.0)
    return ("scalar", s) if s > 0.0 else ("none", None)


ALL_POTENTIALS = ["NoPotential", "Cube", "QuadWell", "Periodic", "Coulomb", "ComplexCoulomb", "ElipticalCoulomb",
                  "SimpleCornell", "FullCornell", "Harmonic", "ComplexHarmonic", "Dodecahedron"]


def test_reference_vectors_of_the_numpy_forms(ref_vectors):
    """the numpy forms themselves against the reference's own unit-test values (potential.rs:445-454)"""
    assert np_alphas(3.2) == pytest.approx(6.189593433886306, abs=1e-14)
    assert np_mu(5.2) == pytest.approx(2.604838027702063, abs=1e-14)


@pytest.mark.parametrize("ext", [1, 2, 3])
@pytest.mark.parametrize("shape", [(12, 9, 17), (16, 16, 16), (7, 10, 8)])
@pytest.mark.parametrize("pot", ALL_POTENTIALS)
def test_every_builtin_potential_vs_numpy_form_of_the_rs_text(oracle, pot, shape, ext):
    """all 12 closed forms, anisotropic even / odd grids (odd sizes put a cell on r = 0: the r < dn
    clamps of Coulomb / Cornell and FullCornell's 0/0), every frame width; a, b as potential.rs:101-110"""
    dn, mass, sig, dt = 0.13, 2.35, 0.223, 1e-3
    cfg = oracle.Config(*shape, ext=ext, potential=pot, dn=dn, dt=dt, mass=mass, sig=sig)
    v = oracle.potential_generate(cfg)
    want = np_potential(pot, shape, ext, dn, mass, sig)
    assert v.shape == want.shape
    if pot in ("Periodic", "FullCornell"):     # libm sin / exp / log against numpy's: an ulp or two
        assert np.allclose(v, want, rtol=1e-13, atol=1e-13)
    else:
        assert np.array_equal(v, want)
    assert np.isfinite(v).all()
    a, b = oracle.ab(cfg, v)
    bw = 1. / (1. + dt * v / 2.)
    assert np.array_equal(b, bw) and np.array_equal(a, (1. - dt * v / 2.) * bw)
    kind, scalar, arr = oracle.potential_sub(cfg)
    wk, wv = np_potsub(pot, shape, dn, mass, sig)
    assert kind == {"none": 0, "scalar": 1, "array": 2}[wk]
    if wk == "scalar":
        assert scalar == wv
    if wk == "array":
        assert arr.shape == tuple(shape)
        both_nan = np.isnan(arr) & np.isnan(wv)       # r = 0 on odd grids: 0 * NaN (hazard kept)
        assert np.allclose(arr[~both_nan], wv[~both_nan], rtol=1e-13) and (np.isnan(arr) == np.isnan(wv)).all()


def test_potential_clamps_are_exercised():
    """the grids above really hit the special branches the forms were written for"""
    v = np_potential("Coulomb", (7, 9, 11), 1, 0.13, 1.0, 1.0)
    assert np.isfinite(v).all() and v[4, 5, 6] == -1. / 0.13   # the cell with r = 0 < dn (padded index = centre (n+1)/2)
    assert (v == -1. / 0.13).sum() == 7                     # ... and its six neighbours at r = dn exactly, where -1/r is the same number
    v = np_potential("SimpleCornell", (7, 9, 11), 2, 0.13, 2.35, 0.223)
    assert (v == 4 * 2.35).sum() == 1
    v = np_potential("FullCornell", (7, 9, 11), 1, 0.13, 2.35, 0.223)
    assert np.isfinite(v).all() and (v == 4 * 2.35).sum() == 1   # md is NaN there, the clamp wins
    k, a = np_potsub("FullCornell", (7, 9, 11), 0.13, 2.35, 0.223)
    assert np.isnan(a).sum() == 1
    d = np_potential("Dodecahedron", (24, 24, 24), 1, 0.1, 1.0, 1.0)
    assert 0.05 < (d == -100.0).mean() < 0.5                # a solid body, neither empty nor everything
    c, q = np_potential("Cube", (10, 13, 17), 1, 0.1, 1.0, 1.0), np_potential("QuadWell", (10, 13, 17), 1, 0.1, 1.0, 1.0)
    assert (c == -10).sum() == 5 * 6 * 8 and (q == -10).sum() == 5 * 6 * 4   # usize division: 10/4=2, 30/4=7 ...


# ---------------------------------------------------------------------------------------------
# Dense-matrix pin (SURVEY.md 8c pin 2): at N <= 8 the Dirichlet operator of every stencil order
# is built as an explicit matrix, cell by cell from the coefficients of grid.rs:582-588 / 608-620 /
# 642-659, and one evolve step and the energy sum are compared with plain matrix algebra.
# ---------------------------------------------------------------------------------------------
def dense_laplacian(shape, e):
    """L with (L phi)[cell] = the bracketed sum S at that work cell, zero frame outside"""
    w, centre, _ = COEFF[e]
    nx, ny, nz = shape
    idx = lambda i, j, k: (i * ny + j) * nz + k   # noqa: E731
    L = np.zeros((nx * ny * nz,) * 2)
    for i in range(nx):
        for j in range(ny):
            for k in range(nz):
                row = idx(i, j, k)
                L[row, row] = -centre
                for off, wt in w.items():
                    for d in (-off, off):
                        for ax in range(3):
                            c = [i, j, k]
                            c[ax] += d
                            if 0 <= c[0] < nx and 0 <= c[1] < ny and 0 <= c[2] < nz:   # else: the Dirichlet frame's zero
                                L[row, idx(*c)] += wt
    return L


@pytest.mark.parametrize("ext", [1, 2, 3])
@pytest.mark.parametrize("shape,pot", [((6, 5, 7), "Harmonic"), ((8, 8, 8), "Coulomb"), ((4, 7, 5), "NoPotential")])
def test_dense_matrix_step_and_rayleigh_quotient(oracle, ext, shape, pot):
    e = ext
    cfg = oracle.Config(*shape, ext=e, potential=pot, dn=0.3, dt=0.004, mass=1.3)
    v = oracle.potential_generate(cfg)
    a, b = oracle.ab(cfg, v)
    rng = np.random.default_rng(11 * e + shape[0])
    phi = np.zeros(cfg.padded_shape)
    phi[e:-e, e:-e, e:-e] = rng.standard_normal(shape)
    x = phi[e:-e, e:-e, e:-e].reshape(-1)
    L = dense_laplacian(shape, e)
    assert np.array_equal(L, L.T)                                   # the discrete operator is symmetric
    den = COEFF[e][2] * cfg.dn * cfg.dn * cfg.mass
    vw, aw, bw = (arr[e:-e, e:-e, e:-e].reshape(-1) for arr in (v, a, b))
    # one evolve step = (diag(a) + dt/den diag(b) L) x      (grid.rs:580-589)
    want = aw * x + bw * cfg.dt * (L @ x) / den
    got = phi.copy()
    oracle.evolve(cfg, 0, a, b, got, [], 1)
    assert np.allclose(got[e:-e, e:-e, e:-e].reshape(-1), want, rtol=0, atol=1e-12 * np.abs(want).max())
    assert not got[:e].any() and not got[:, :, -e:].any()          # the frame is never written
    # energy sum = x^T (diag(V) - L/den) x = x^T H x        (grid.rs:325-332)
    H = np.diag(vw) - L / den
    obs = oracle.observables(cfg, v, phi)
    assert obs["energy"] == pytest.approx(x @ H @ x, rel=1e-11)
    assert obs["norm2"] == pytest.approx(x @ x, rel=1e-13)
    # the lowest eigenvalue of H bounds the Rayleigh quotient from below; k steps of evolve from the
    # dense ground state leave its direction unchanged to O(dt^2)
    evals, evecs = np.linalg.eigh(H)
    assert obs["energy"] / obs["norm2"] >= evals[0] - 1e-9
    g = np.zeros(cfg.padded_shape)
    g[e:-e, e:-e, e:-e] = evecs[:, 0].reshape(shape)
    og = oracle.observables(cfg, v, g)
    assert og["energy"] / og["norm2"] == pytest.approx(evals[0], rel=1e-10, abs=1e-10)
