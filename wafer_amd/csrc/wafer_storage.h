// Storage type against register type of the fused ground-state kernels.
#pragma once
#include <type_traits>

// What the first template argument of the fused kernels (wafer_k_step3_fused, wafer_k_step2_wide) stands for.  A plain type: stored, queued and staged as that type.
// wafer_f32_wide: fp32 STORAGE with fp64 arithmetic (dtype WAFER_F32: config #5's) -- the arrays in HBM are float, 8 bytes per lane
// and request; everything inside the CU (z-queues, LDS rings, the carried a and b * dt) is double, exactly the fp64 kernel's, and
// every level's result is rounded to float before it is used or stored: the bits of three single fp32-storage steps.  (Queues and
// LDS in float would halve the LDS traffic but put a conversion in front of every operand: 4 more vector instructions per update
// where this form has 2.7, and the fp64 kernel is bound by the in-CU pipeline, not by bytes.)
struct wafer_f32_wide {};
template <typename T> struct WaferF3Store { using S = T; using Q = T; };
template <> struct WaferF3Store<wafer_f32_wide> { using S = float; using Q = double; };
template <typename SV, typename QV, int N>
__device__ __forceinline__ QV wafer_f3_widen(const SV &x)
{
    if constexpr (std::is_same<SV, QV>::value) return x;
    else {
        QV r;
#pragma unroll
        for (int v = 0; v < N; ++v) r[v] = x[v];
        return r;
    }
}

