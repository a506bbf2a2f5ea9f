// The plan of x / den for the run's one stencil denominator (host side; the device side is WaferDen / wafer_div_invariant in
// wafer_stencil.hip.h, which says what the plan is for).  Plain C++, no HIP: wafer_div_plan() of the C ABI serves it without a GPU.
//
// The device forms  q = RN(x*zh + RN(x*zl))  with zh = RN(1/den), zl = RN(1/den - zh) moved by at most two ulps.  Before its
// rounding that value is
//     x/den + x*(zh + zl - 1/den) + x*zl*e2,   |e2| <= 2^-53.
// |1/den - zh| <= 2^-53 / |den|, so ulp(zl) <= 2^-105 / |den| and |zh + zl - 1/den| <= 2.5 ulp(zl) <= 5 * 2^-106 / |den|;
// |x*zl*e2| <= 1.01 * 2^-106 |x/den|.  With Q = x/den in [2^e, 2^(e+1)) and ulp = 2^(e-52) the value is Q + delta,
// |delta| < 6.01 * 2^-106 * 2^(e+1) ~ 3 * 2^-52 ulp (2^-52 ulp with zl unmoved).  RN(Q + delta) differs from RN(Q) only if the
// midpoint m of two neighbouring doubles lies within |delta| of Q.  In integers -- den = D * 2^d, x = X * 2^a, m = M * 2^c with
// D, X in [2^52, 2^53) and M ODD in [2^53, 2^54), ulp = 2^(c+1); c + d = a - s with s = 53 where X >= D and s = 54 where X < D --
//     |X * 2^s - M * D| <= |delta| / ulp * 2 D < 12.02 * 2^-54 * 2^54 ~ 12                                    (*)
// a nonzero integer k (zero would make a 53-bit D divisible by 2^53).  For every k the congruence M * D = -k (mod 2^s) has
// at most a handful of odd solutions M in range, each gives at most one X: these X (and -X; the exponent of x plays no part
// while nothing is subnormal) are the only operands whose quotient can round wrongly.  The plan enumerates them for |k| <= 32
// -- more than twice what (*) asks for -- and runs the device's very instruction sequence on each against the IEEE division.
// The enumeration was checked exhaustively in 8- to 12-bit arithmetic (every divisor, every operand: no operand outside the
// candidate set ever fails; tests/test_div_plan.py repeats it for 8 and 9 bits) and against exact rational arithmetic for doubles.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

struct WaferDivPlan {
    double den = 0.0, zh = 0.0, zl = 0.0;
    int checked = 0;       // 1: the three-instruction form returns RN(x/den) for every x; 0: the device adds a Markstein round
    int n_candidates = 0;  // significands that had to be tried
    int zl_shift = 0;      // ulps zl was moved from RN(1/den - zh) to get every candidate through
};

// the device's sequence, on the host (std::fma is correctly rounded; the library is built with -ffp-contract=off)
static inline double wafer_divplan_q(double x, double zh, double zl)
{
    const double t = x * zl;
    return std::fma(x, zh, t);
}
// ... and with the extra round of an unchecked divisor
static inline double wafer_divplan_q_round(double x, double den, double zh, double zl)
{
    const double q = wafer_divplan_q(x, zh, zl);
    const double r = std::fma(-q, den, x);
    return std::fma(r, zh, q);
}

// the significands X in [2^52, 2^53) (as doubles) whose quotient by den comes within 2^-49 ulp of a rounding boundary
static inline std::vector<double> wafer_divplan_candidates(double den, int K = 32)
{
    std::vector<double> out;
    if (!(std::isfinite(den)) || den == 0.0) return out;
    int e2 = 0;
    const double m = std::frexp(std::fabs(den), &e2);
    const uint64_t D = (uint64_t)std::ldexp(m, 53);   // (subnormal den: fewer bits, D < 2^52 -- no plan then)
    if (D < (1ull << 52)) return out;
    int t = 0;
    while (!((D >> t) & 1)) ++t;
    if ((1 << (t < 30 ? t : 30)) > K) return out;   // every k of (*) would have to be a multiple of 2^t: none is
    const uint64_t Dodd = D >> t;
    uint64_t inv = Dodd;   // Newton: the inverse of an odd number modulo 2^64
    for (int i = 0; i < 6; ++i) inv *= 2 - Dodd * inv;
    for (int s = 53; s <= 54; ++s) {
        const int mb = s - t;   // M is fixed modulo 2^mb
        const uint64_t mod_mask = (1ull << mb) - 1;
        for (int k = -K; k <= K; ++k) {
            if (k == 0 || k % (1 << t) != 0) continue;
            const int64_t kp = k / (1 << t);
            uint64_t M = ((uint64_t)(-kp) * inv) & mod_mask;
            for (; M < (1ull << 54); M += (1ull << mb)) {
                if (M < (1ull << 53) || !(M & 1)) continue;
                const __int128 num = (__int128)((unsigned __int128)M * D) + k;
                if (num & (((__int128)1 << s) - 1)) continue;
                const uint64_t X = (uint64_t)(num >> s);
                if (X < (1ull << 52) || X >= (1ull << 53)) continue;
                if ((s == 53) != (X >= D)) continue;
                out.push_back((double)X);
            }
        }
    }
    return out;
}

static inline WaferDivPlan wafer_divplan_make(double den)
{
    WaferDivPlan p;
    p.den = den;
    p.zh = 1.0 / den;
    const double zl0 = std::fma(-p.zh, den, 1.0) / den;   // RN(1/den - zh): the remainder 1 - zh*den is exact
    p.zl = zl0;
    if (!std::isnormal(den) || !std::isnormal(p.zh)) {
        // zero / infinite / NaN / subnormal divisors (or reciprocals): no plan, zl = 0 makes the first product exact and the extra
        // round + v_div_fixup see to the rest
        p.zl = 0.0;
        return p;
    }
    // The candidate set rests on |zh + zl - 1/den| <= 2.5 ulp(zl) with zl at FULL precision, and the trial runs significands at
    // one exponent.  For |den| above ~2^969 zl is subnormal (its relative error far above 2^-106), and for a tiny den x*zh nears
    // overflow for operands the trial never sees: outside [2^-900, 2^900] -- or with a subnormal non-zero zl -- no verdict:
    // `checked` stays 0 and the kernels take the Markstein round, which is RN whatever the plan found.
    {
        int e = 0;
        (void)std::frexp(den, &e);
        if (zl0 != 0.0 && (e < -900 || e > 900 || !std::isnormal(zl0))) return p;   // (zl0 == 0: 1/den is a double, x*zh IS the quotient)
    }
    const std::vector<double> cand = wafer_divplan_candidates(den);
    p.n_candidates = (int)cand.size();
    static const int shifts[] = {0, 1, -1, 2, -2};
    for (int sh : shifts) {
        double zl = zl0;
        for (int i = 0; i < std::abs(sh); ++i) zl = std::nextafter(zl, sh > 0 ? INFINITY : -INFINITY);
        if (sh != 0 && zl0 == 0.0) break;   // 1/den is a double: nothing to move
        bool ok = true;
        for (double X : cand) {
            for (int sg = 0; sg < 2 && ok; ++sg) {
                const double x = sg ? -X : X;
                ok = wafer_divplan_q(x, p.zh, zl) == x / den;
            }
            if (!ok) break;
        }
        if (ok) {
            p.zl = zl;
            p.zl_shift = sh;
            p.checked = 1;
            return p;
        }
    }
    return p;
}

// ---- the same plan in fp32, for the contexts whose step kernels compute in fp32 (WAFER_F32_FAST): q = RN(x*zh + RN(x*zl)) in float.
// No enumeration needed: all 2^23 significands are tried (both signs by symmetry; the exponent of x plays no part while x*zl is a
// normal float, |x/den| >= 2^-100).  ~10 ms per zl tried, once per context.
struct WaferDivPlanF {
    float den = 0.f, zh = 0.f, zl = 0.f;
    int checked = 0;
    int zl_shift = 0;
};

static inline float wafer_divplan_qf(float x, float zh, float zl)
{
    const float t = x * zl;
    return std::fmaf(x, zh, t);
}

static inline WaferDivPlanF wafer_divplan_make_f32(float den)
{
    WaferDivPlanF p;
    p.den = den;
    p.zh = 1.0f / den;
    const float zl0 = std::fmaf(-p.zh, den, 1.0f) / den;
    p.zl = zl0;
    // the scaling below must stay inside the normal floats: |den| in [2^-60, 2^60]
    if (!std::isnormal(den) || !(std::fabs(den) >= 0x1p-60f && std::fabs(den) <= 0x1p60f)) {
        p.zl = 0.f;
        return p;
    }
    static const int shifts[] = {0, 1, -1, 2, -2};
    for (int sh : shifts) {
        float zl = zl0;
        for (int i = 0; i < std::abs(sh); ++i) zl = std::nextafterf(zl, sh > 0 ? INFINITY : -INFINITY);
        if (sh != 0 && zl0 == 0.f) break;
        bool ok = true;
        for (uint32_t X = 1u << 23; X < (1u << 24) && ok; ++X) {
            const float x = (float)X;
            ok = wafer_divplan_qf(x, p.zh, zl) == x / den;
        }
        if (ok) {
            p.zl = zl;
            p.zl_shift = sh;
            p.checked = 1;
            return p;
        }
    }
    return p;
}
