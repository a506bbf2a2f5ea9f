"""The plan of x / (c dn^2 m) (wafer_amd/csrc/wafer_divplan.h, wafer_div_plan of the C ABI): host code, no GPU.

The step kernels form q = RN(x zh + RN(x zl)) and the plan vouches that this is RN(x / den) for EVERY x after trying only the
few dozen significands whose quotient comes close enough to a rounding boundary.  Three things are checked here:
  * the enumeration of those significands is complete: in 8- and 9-bit arithmetic, every divisor and every operand tried,
    no operand outside the candidate set ever fails (the same program ran for 10 to 12 bits when the plan was written);
  * the library's candidates for doubles are the ones exact integer arithmetic in Python finds;
  * the library's verdict is what exact rational arithmetic says about those candidates, for the denominators of the
    configurations in use and for random ones -- including some whose first zl fails.
"""
import math
import random
from fractions import Fraction as F

import numpy as np
import pytest

import wafer_amd
from wafer_amd import engine


def rn_bits(fr, p):
    """a Fraction rounded to p significant bits, ties to even, unbounded exponent"""
    if fr == 0:
        return F(0)
    s = 1 if fr > 0 else -1
    a = abs(fr)
    e = a.numerator.bit_length() - a.denominator.bit_length() - p
    while a / F(2) ** e >= (1 << p):
        e += 1
    while a / F(2) ** e < (1 << (p - 1)):
        e -= 1
    q = a / F(2) ** e
    fl = q.numerator // q.denominator
    rem = q - fl
    if rem > F(1, 2) or (rem == F(1, 2) and (fl & 1)):
        fl += 1
    return s * fl * F(2) ** e


def candidates(D, p, K):
    """significands X in [2^(p-1), 2^p) with |X 2^s - M D| <= K for an odd M in [2^p, 2^(p+1)): wafer_divplan.h's (*)"""
    t = (D & -D).bit_length() - 1
    if (1 << t) > K:
        return set()
    Dp = D >> t
    out = set()
    for s in (p, p + 1):
        mb = s - t
        mod = 1 << mb
        inv = pow(Dp, -1, mod)
        for k in range(-K, K + 1):
            if k == 0 or k % (1 << t):
                continue
            M = (-(k // (1 << t)) * inv) % mod
            while M < (1 << (p + 1)):
                if M >= (1 << p) and (M & 1):
                    num = M * D + k
                    if num % (1 << s) == 0:
                        X = num >> s
                        if (1 << (p - 1)) <= X < (1 << p) and ((s == p) == (X >= D)):
                            out.add(X)
                M += mod
    return out


@pytest.mark.parametrize("p", [8, 9])
def test_no_operand_outside_the_candidate_set_fails_in_small_arithmetic(p):
    """every divisor, every operand, zl as rounded and moved by up to two ulps: the operands for which
    RN(x zh + RN(x zl)) != RN(x / den) all lie in the candidate set of (*) with K = 12"""
    failures = 0
    for D in range(1 << (p - 1), 1 << p):
        den = F(D)
        zh = rn_bits(1 / den, p)
        zl0 = rn_bits(1 / den - zh, p)
        cand = candidates(D, p, 12)
        for shift in (0, 1, -1, 2, -2):
            if zl0 == 0 and shift:
                continue
            ulp = F(2) ** (math.floor(math.log2(abs(zl0))) - (p - 1)) if zl0 != 0 else F(0)
            zl = zl0 + shift * ulp
            for X in range(1 << (p - 1), 1 << p):
                x = F(X)
                q = rn_bits(x * zh + rn_bits(x * zl, p), p)
                if q != rn_bits(x / den, p):
                    failures += 1
                    assert X in cand, (p, D, X, shift)
                    # ... and the extra Markstein round repairs it
                    r = rn_bits(x - q * den, p)
                    assert rn_bits(q + r * zh, p) == rn_bits(x / den, p), (p, D, X, shift)
    assert failures > 0   # the test is live: some (divisor, operand) pairs do fail without the check


def exact_q(x, zh, zl):
    t = F(x) * F(zl)
    t = t.numerator / t.denominator          # int / int: correctly rounded
    v = F(x) * F(zh) + F(t)
    return v.numerator / v.denominator


def exact_div(x, den):
    v = F(x) / F(den)
    return v.numerator / v.denominator


DENS_IN_USE = [2 * 0.05 ** 2 * 1.0, 2 * 0.02 ** 2 * 2.35, 24 * 0.05 ** 2, 360 * 0.05 ** 2, 2 * 0.1 ** 2, 24 * 0.2 ** 2 * 1.3,
               360 * 0.01 ** 2 * 0.7, 2 * 0.2 ** 2, 2 * 0.01 ** 2 * 0.5, 0.5, 3.0, 1.0, float(np.nextafter(2.0, 0.0)),
               float(np.nextafter(1.0, 2.0)), 1.7320508075688772e-3]


# divisors whose RN(1/den - zh) leaves a candidate on the wrong side of its boundary (found by a random search)
DENS_NEEDING_A_MOVED_ZL = [0.007395769697490762, 0.1078657875904072, 0.20222586000144446]


def python_plan(den):
    m, e = math.frexp(den)
    D = int(m * (1 << 53))
    cand = sorted(candidates(D, 53, 32))
    zh = 1.0 / den
    e1 = F(1) - F(zh) * F(den)
    zl0 = (e1 / F(den)).numerator / (e1 / F(den)).denominator
    for shift in (0, 1, -1, 2, -2):
        zl = zl0
        for _ in range(abs(shift)):
            zl = float(np.nextafter(zl, math.inf if shift > 0 else -math.inf))
        if shift and zl0 == 0.0:
            break
        if all(exact_q(s * float(X), zh, zl) == exact_div(s * float(X), den) for X in cand for s in (1.0, -1.0)):
            return cand, zh, zl, 1, shift
    return cand, zh, zl0, 0, 0


def test_the_library_plans_what_exact_arithmetic_plans():
    rng = random.Random(5)
    dens = DENS_IN_USE + DENS_NEEDING_A_MOVED_ZL + [rng.uniform(1, 2) * 2.0 ** rng.randint(-30, 10) for _ in range(300)]
    shifted = unchecked = 0
    for den in dens:
        plan, cand = engine.div_plan(den)
        want_cand, zh, zl, checked, shift = python_plan(den)
        assert sorted(int(c) for c in cand) == want_cand, den
        assert plan.n_candidates == len(want_cand)
        assert (plan.den, plan.zh, plan.checked, plan.zl_shift) == (den, zh, checked, shift), (den, plan.as_dict())
        assert plan.zl == zl, den
        shifted += shift != 0
        unchecked += not checked
    assert all(engine.div_plan(d)[0].checked == 1 for d in DENS_IN_USE)
    assert shifted >= 3         # some divisors need their zl moved: the search is exercised
    assert unchecked <= 1       # ... and with it next to none is left to the extra round


def test_the_extra_round_repairs_every_candidate_of_a_divisor_whose_first_zl_fails():
    """divisors for which RN(1/den - zh) does not get every candidate through: the failing operands exist (the check is
    not vacuous), and q + RN(x - q den) zh rounds to the IEEE quotient for each of them (Markstein's theorem: q is faithful)"""
    for den in DENS_NEEDING_A_MOVED_ZL:
        plan, cand = engine.div_plan(den)
        assert plan.zl_shift != 0 and plan.checked == 1
        zh = plan.zh
        e1 = F(1) - F(zh) * F(den)
        zl0 = (e1 / F(den)).numerator / (e1 / F(den)).denominator
        bad = [s * x for x in cand for s in (1.0, -1.0) if exact_q(s * x, zh, zl0) != exact_div(s * x, den)]
        assert bad
        for x in bad:
            q = exact_q(x, zh, zl0)
            r = F(x) - F(q) * F(den)
            r = r.numerator / r.denominator
            v = F(q) + F(r) * F(zh)
            assert v.numerator / v.denominator == exact_div(x, den)


def test_divisors_without_a_plan():
    for den in (0.0, math.inf, math.nan, 5e-324, 1e-310):
        plan, cand = engine.div_plan(den)
        assert plan.checked == 0 and cand.size == 0
    # outside [2^-900, 2^900] the enumeration's premise fails (zl subnormal above ~2^969; x zh near overflow for a tiny den): no
    # verdict, the kernels keep the Markstein round -- except for powers of two, whose zl is 0 (wafer_divplan.h:88)
    for den in (3.0 * 2.0 ** 1000, 1.7 * 2.0 ** 970, 1.3 * 2.0 ** -1000, -(5.0 / 3.0) * 2.0 ** 905):
        plan, cand = engine.div_plan(den)
        assert plan.checked == 0 and plan.den == den and plan.zh == 1.0 / den
    for den in (2.0 ** 1000, 2.0 ** -1000):
        plan, cand = engine.div_plan(den)
        assert (plan.checked, plan.zl) == (1, 0.0)
    for den in (1.3 * 2.0 ** 899, 1.3 * 2.0 ** -899):      # inside the window: planned as before
        assert engine.div_plan(den)[0].n_candidates > 0
    # a power of two: 1/den is a double, zl = 0, nothing can go wrong
    plan, cand = engine.div_plan(0.125)
    assert (plan.zh, plan.zl, plan.checked, cand.size) == (8.0, 0.0, 1, 0)
    assert "wafer_div_plan" in wafer_amd.engine.EXPORTS


def test_the_fp32_plan_against_every_significand():
    """WAFER_F32_FAST contexts: q = RN(x zh + RN(x zl)) in float.  The library tries all 2^23 significands; so does numpy here (a
    float product is exact in float64, the sum of a 48-bit product and a float 2^-24 below it too, one rounding to float32 = fmaf)"""
    X = np.arange(1 << 23, 1 << 24, dtype=np.float32)

    def all_right(den, zh, zl):
        t = (X * np.float32(zl)).astype(np.float32)
        q = (X.astype(np.float64) * np.float64(np.float32(zh)) + t.astype(np.float64)).astype(np.float32)
        return bool(np.array_equal(q, X / np.float32(den)))

    rng = random.Random(9)
    dens = [2 * 0.05 ** 2, 2 * 0.02 ** 2 * 2.35, 24 * 0.05 ** 2, 360 * 0.05 ** 2, 0.08, 0.5, 3.0] + [rng.uniform(1, 2) * 2.0 ** rng.randint(-20, 5) for _ in range(12)]
    moved = 0
    for den in dens:
        p = engine.div_plan_f32(den)
        den32 = np.float32(den)
        assert p.den == den32 and p.zh == np.float32(1) / den32
        zl0 = np.float32((1.0 - float(p.zh) * float(den32)) / float(den32))   # exact remainder (float64 holds it), one division
        assert p.checked == 1, den
        assert all_right(den, p.zh, p.zl)
        if p.zl_shift == 0:
            assert p.zl == zl0
        else:
            moved += 1
            assert not all_right(den, p.zh, zl0)      # the move was needed
    for den in (0.0, math.inf, math.nan, 1e-30, 1e30):
        assert engine.div_plan_f32(den).checked == 0
