// wafer_divplan.h (the host side of the planned division) under -fsanitize=address,undefined: tests/test_sanitizers.py builds and runs this.
#include <cstdio>
#include <limits>
#include <random>
#include "../wafer_amd/csrc/wafer_divplan.h"

int main()
{
    const double inf = std::numeric_limits<double>::infinity();
    const double awkward[] = {0.0, -0.0, inf, -inf, std::nan(""), 5e-324, 1e-310, 2.2250738585072014e-308, 1.7976931348623157e308, 1.0, 0.125, 3.0,
                              -0.005, 0.005000000000000001, 0.0018800000000000002, 0.007395769697490762, 0.1078657875904072, 0.20222586000144446,
                              1.0 + 0x1p-20, 1.5, 0x1.fffffffffffffp0, 0x1.0000000000001p0, 0x1p-1000, 0x1p1000, 6.0 * 0x1p40};
    int checked = 0, total = 0;
    for (double den : awkward) {
        const WaferDivPlan p = wafer_divplan_make(den);
        checked += p.checked;
        ++total;
        if (p.checked)   // what `checked` promises, on the candidates themselves
            for (double X : wafer_divplan_candidates(den))
                if (wafer_divplan_q(X, p.zh, p.zl) != X / den || wafer_divplan_q(-X, p.zh, p.zl) != -X / den) return 2;
    }
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> mant(1.0, 2.0);
    for (int i = 0; i < 2000; ++i) {
        const double den = std::ldexp(mant(rng), (int)(rng() % 200) - 100) * ((rng() & 1) ? 1.0 : -1.0);
        const WaferDivPlan p = wafer_divplan_make(den);
        checked += p.checked;
        ++total;
        for (double X : wafer_divplan_candidates(den))   // the extra round is right whatever the plan found
            if (wafer_divplan_q_round(X, den, p.zh, p.zl) != X / den) return 3;
    }
    std::printf("DIVPLAN-OK %d of %d checked\n", checked, total);
    return 0;
}
