"""CPU: the algebra behind wafer_stencil_x2.hip.h (two excited-state steps per pass), in numpy on top of the oracle's single
steps, against the oracle's own sequence (grid.rs:562-686 with wnum > 0: step, norm^2, normalise, modified Gram-Schmidt after
EVERY step).  What the kernel relies on, each held here without a GPU:

  * linearity: the second step's normalisation and projection can be applied by the NEXT pass's load transform, with
    M_j = A l_j stored once per state;
  * scale invariance: the state may be carried up to a positive factor (the next pass's first division removes it), so the
    norm of the second raw step is never needed between passes -- only when phi is materialised;
  * the scalars a pass needs follow from 1 + 2k sums taken DURING the previous pass (sum Y1^2, sum l_j Y1, sum l_j Z), the
    Gram matrix and <l_j, M_i>.
"""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def wo():
    from oracle import wafer_oracle
    wafer_oracle.build()
    return wafer_oracle


def _field(rng, shape, ext):
    w = np.zeros(tuple(n + 2 * ext for n in shape))
    w[ext:-ext, ext:-ext, ext:-ext] = rng.standard_normal(shape)
    return w


@pytest.mark.parametrize("k", [1, 2, 3])
@pytest.mark.parametrize("orthonormal", [True, False])
def test_two_steps_per_pass_regrouping_reproduces_the_reference_sequence(wo, k, orthonormal):
    shape, ext = (22, 18, 16), 1
    cfg = wo.Config(*shape, ext=ext, potential="Coulomb", dn=0.05, dt=5e-4, mass=1.0, sig=0.223)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    rng = np.random.default_rng(7)
    dot = lambda x, y: float(np.sum(x * y))   # noqa: E731

    def A(w):   # one ground-state step: the linear operator of grid.rs:568-592
        w = w.copy()
        wo.evolve(cfg, 0, a, b, w, [], 1)
        return w

    lows = []
    for j in range(k):
        w = _field(rng, shape, ext)
        for _ in range(20):
            w = A(w)
        if orthonormal:
            for l in lows:
                w -= l * dot(l, w)
        elif lows:
            w += 0.4 * np.sqrt(dot(w, w)) * lows[0]      # deliberately correlated
        w /= np.sqrt(dot(w, w))
        lows.append(w)
    phi0 = _field(rng, shape, ext)
    head, pairs = 2, 9
    want = phi0.copy()
    wo.evolve(cfg, k, a, b, want, lows, head + 2 * pairs)

    G = np.array([[dot(lows[j], lows[i]) for i in range(k)] for j in range(k)])
    M = [A(l) for l in lows]
    amat = np.array([[dot(lows[j], M[i]) for i in range(k)] for j in range(k)])     # <l_j, M_i>

    def mgs(n, t):     # the reference's sequential overlaps from raw ones (wafer_k_gs_apply's recurrence)
        s = np.zeros(k)
        for j in range(k):
            s[j] = t[j] / n - sum(s[i] * G[j][i] for i in range(j))
        return s

    # head: the reference's own sequence, one step per pass; the last one leaves the RAW step and its sums
    x = phi0.copy()
    if head > 1:
        wo.evolve(cfg, k, a, b, x, lows, head - 1)
    raw = A(x)
    n = np.sqrt(dot(raw, raw))
    s = mgs(n, [dot(l, raw) for l in lows])
    w0, sb, sc = 1.0 / n, np.zeros(k), s       # kind-1 coefficients: x = raw / n - sum s_j l_j
    buf = raw
    for _ in range(pairs):
        xt = buf * w0 - sum(sb[j] * M[j] for j in range(k)) - sum(sc[j] * lows[j] for j in range(k))   # the load transform
        y1 = A(xt)
        z = A(y1)
        # the 1 + 2k sums of the pass
        syy, sly, slz = dot(y1, y1), [dot(l, y1) for l in lows], [dot(l, z) for l in lows]
        nb = np.sqrt(syy)
        sb = mgs(nb, sly)
        t2 = [slz[j] / nb - sum(sb[i] * amat[j][i] for i in range(k)) for j in range(k)]
        sc = mgs(1.0, t2)                      # sigma_j = n_c s^c_j: not divided by the (unknown) norm
        w0 = 1.0 / nb
        buf = z
    # materialise: the last step's norm taken directly
    u = buf * w0 - sum(sb[j] * M[j] for j in range(k))
    got = (u - sum(sc[j] * lows[j] for j in range(k))) / np.sqrt(dot(u, u))
    assert np.max(np.abs(got - want)) < 1e-14, float(np.max(np.abs(got - want)))
    assert dot(got, got) == pytest.approx(dot(want, want), rel=1e-13)


def test_scale_of_the_input_drops_out_after_one_step(wo):
    """the reference's state after a step does not depend on the scale of the step's input: (A c x) / |A c x| = (A x) / |A x|,
    and the overlaps with the stored states scale the same way"""
    shape, ext = (12, 10, 14), 1
    cfg = wo.Config(*shape, ext=ext, potential="Harmonic", dn=0.3, dt=0.01, mass=1.0, sig=1.0)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    rng = np.random.default_rng(3)
    low = _field(rng, shape, ext)
    low /= np.sqrt(np.sum(low * low))
    x = _field(rng, shape, ext)
    one, other = x.copy(), 0.37 * x
    wo.evolve(cfg, 1, a, b, one, [low], 1)
    wo.evolve(cfg, 1, a, b, other, [low], 1)
    assert np.max(np.abs(one - other)) < 1e-15
