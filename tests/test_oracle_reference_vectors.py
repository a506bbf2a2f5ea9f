"""The CPU oracle against every known-answer vector the reference's own unit
tests hold for this path.  CPU only."""
import numpy as np


def _rule(shape, rule):
    i, j, k = np.meshgrid(*[np.arange(s, dtype=np.float64) for s in shape], indexing="ij")
    return np.ascontiguousarray({"i+j+k": i + j + k, "-(i+j+k)": -i - j - k, "i*j*k": i * j * k}[rule])


def test_gram_schmidt(oracle, ref_vectors):
    g = ref_vectors["gram_schmidt"]  # grid.rs:721-746
    lower = _rule(g["shape"], g["lower_rule"])
    phi = _rule(g["shape"], g["phi_rule"])
    oracle.orthogonalise(1, phi, [lower])
    assert np.allclose(phi.ravel(), g["expected"], atol=g["tol"], rtol=0)
    # the vector is exactly representable, so the oracle must hit it exactly
    assert np.array_equal(phi.ravel(), np.array(g["expected"]))


def test_work_area_dims(oracle, ref_vectors):
    g = ref_vectors["work_area"]  # grid.rs:748-778
    cfg = oracle.Config(*g["expected_dims"], ext=g["ext"])
    assert list(cfg.padded_shape) == g["shape"]
    # mut_work_area: filling the work area leaves a zero frame; Constant IC does exactly that
    phi = oracle.initial_condition(cfg, "Constant")
    e = g["ext"]
    inner = phi[e:-e, e:-e, e:-e]
    assert inner.shape == tuple(g["expected_dims"]) and np.all(inner == 0.1)
    assert phi.sum() == inner.sum()


def test_norm2(oracle, ref_vectors):
    g = ref_vectors["norm2"]  # grid.rs:780-786
    e = g["ext"]
    cfg = oracle.Config(*[s - 2 * e for s in g["shape"]], ext=e)
    phi = _rule(g["shape"], g["phi_rule"])
    assert abs(oracle.norm2(cfg, phi) - g["expected"]) < g["eps"]


def test_wfn_normalise(oracle, ref_vectors):
    g = ref_vectors["wfn_normalise"]  # grid.rs:788-799
    phi = _rule(g["shape"], g["phi_rule"])
    want = _rule(g["shape"], g["phi_rule"]) / g["expected_divisor"]
    oracle.normalise(phi, g["norm2"])
    assert np.allclose(phi, want, atol=g["tol"], rtol=0)
    assert np.array_equal(phi, _rule(g["shape"], g["phi_rule"]) / np.sqrt(g["norm2"]))


def test_distance_squared(oracle, ref_vectors):
    g = ref_vectors["distance_squared"]  # potential.rs:434-443
    assert abs(oracle.calculate_r2(g["idx"], g["size"]) - g["expected"]) < g["eps"]


def test_running_coupling(oracle, ref_vectors):
    g = ref_vectors["running_coupling"]  # potential.rs:445-449
    assert abs(oracle.alphas(g["mu"]) - g["expected"]) < g["eps"]


def test_debye_screening_mass(oracle, ref_vectors):
    g = ref_vectors["debye_screening_mass"]  # potential.rs:450-454
    assert abs(oracle.mu(g["t"]) - g["expected"]) < g["eps"]


def test_interpolation(oracle, ref_vectors):
    g = ref_vectors["interpolation"]  # input.rs:732-824, assert_eq! => bit exact
    src = np.array(g["source"]).reshape(g["source_shape"])
    out = oracle.trilerp_resize(src, g["target_shape"])
    assert np.array_equal(out.ravel(), np.array(g["expected"]))


def test_interpolation_production_basis(oracle):
    """the reference's production call builds the basis from the PADDED target size while
    filling the unpadded work view (input.rs:156-173, 640-656): the first N of N+bb samples"""
    rng = np.random.default_rng(0)
    src = rng.standard_normal((5, 4, 6))
    n, bb = (9, 7, 11), 2
    out = oracle.trilerp_resize(src, n, basis=tuple(s + bb for s in n))
    full = oracle.trilerp_resize(src, tuple(s + bb for s in n))      # the whole padded-size basis
    assert np.array_equal(out, full[:n[0], :n[1], :n[2]])
    # independent numpy form of the same rule
    def axis(m, count):
        pos = np.arange(count) * ((m - 1) / (count - 1))
        lo = np.minimum(np.floor(pos).astype(int), m - 2)
        return lo, pos - lo
    (x0, xd), (y0, yd), (z0, zd) = [axis(m, c + bb) for m, c in zip(src.shape, n)]
    x0, xd, y0, yd, z0, zd = x0[:n[0]], xd[:n[0]], y0[:n[1]], yd[:n[1]], z0[:n[2]], zd[:n[2]]
    X0, Y0, Z0 = np.ix_(x0, y0, z0)
    XD, YD, ZD = np.ix_(xd, yd, zd)
    lerp = lambda a, b, t: a * (1 - t) + b * t
    c00 = lerp(src[X0, Y0, Z0], src[X0 + 1, Y0, Z0], XD)
    c01 = lerp(src[X0, Y0, Z0 + 1], src[X0 + 1, Y0, Z0 + 1], XD)
    c10 = lerp(src[X0, Y0 + 1, Z0], src[X0 + 1, Y0 + 1, Z0], XD)
    c11 = lerp(src[X0, Y0 + 1, Z0 + 1], src[X0 + 1, Y0 + 1, Z0 + 1], XD)
    want = lerp(lerp(c00, c10, YD), lerp(c01, c11, YD), ZD)
    assert np.allclose(out, want, rtol=1e-13, atol=1e-14)
