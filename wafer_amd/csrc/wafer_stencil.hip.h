// Stencil arithmetic and the evolve / observables kernels (gfx950).
//
// Arithmetic contract: the per-point expressions keep the association the
// reference spells (grid.rs:580-589, 606-621, 640-660, 325-332) and the
// library is built with -ffp-contract=off, so an fp64 stencil step is
// bit-for-bit what rustc emits for the same inputs (IEEE add/mul/div, no FMA).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "wafer_geom.h"
#include "wafer_tuning.h"

#define WAFER_WAVE 64

// ---- wave / block sum of one double (fixed order => run-to-run deterministic)
__device__ __forceinline__ double wafer_wave_sum(double v)
{
#pragma unroll
    for (int off = WAFER_WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, WAFER_WAVE);
    return v;
}

// Sums `v` over a block of NW waves; result valid in thread 0.  `red` is NW doubles of LDS.
template <int NW>
__device__ __forceinline__ double wafer_block_sum(double v, double *red, int tid)
{
    v = wafer_wave_sum(v);
    const int wave = tid / WAFER_WAVE, lane = tid % WAFER_WAVE;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0.0;
    if (tid == 0) {
#pragma unroll
        for (int w = 0; w < NW; ++w) s += red[w];
    }
    return s;
}

// The multi-step kernels' result stores: streamed (non-temporal) -- nobody reads the new wavefunction before the next launch.
// -DWAFER_PLAIN_STORES: A/B builds.
template <typename VT>
__device__ __forceinline__ void wafer_store_result(VT *p, VT v)
{
#ifdef WAFER_PLAIN_STORES
    *p = v;
#else
    __builtin_nontemporal_store(v, p);
#endif
}

// ---- a value of the neighbouring lane by DPP (gfx9 wave shifts), for x neighbours that the lane next door holds in registers.
// wafer_lane_below(own, edge): lane i gets lane i-1's `own`, lane 0 keeps `edge`; wafer_lane_above: lane i gets lane i+1's,
// lane 63 keeps `edge`.  Two v_mov_b32 with a DPP control per double (the shifts exist for 32-bit operands only).
template <int CTRL>
__device__ __forceinline__ double wafer_lane_shift(double own, double edge)
{
    int lo = __double2loint(own), hi = __double2hiint(own);
    const int elo = __double2loint(edge), ehi = __double2hiint(edge);
    lo = __builtin_amdgcn_update_dpp(elo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(ehi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wafer_lane_below(double own, double edge) { return wafer_lane_shift<0x138>(own, edge); } // wave_shr:1
__device__ __forceinline__ double wafer_lane_above(double own, double edge) { return wafer_lane_shift<0x130>(own, edge); } // wave_shl:1
__device__ __forceinline__ float wafer_lane_below(float own, float edge)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(own), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float wafer_lane_above(float own, float edge)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(own), 0x130, 0xf, 0xf, false));
}

// RN(c * x + t) for a power-of-two c: the same bits as RN(RN(c * x) + t), the product being exact
__device__ __forceinline__ double wafer_fma_pow2(double c, double x, double t) { return __builtin_fma(c, x, t); }
__device__ __forceinline__ float wafer_fma_pow2(float c, float x, float t) { return __builtin_fmaf(c, x, t); }

// ---- the bracketed central-difference sum S --------------------------------
// xs/ys/zs hold the 2R+1 values along each axis, index R is the centre w.
template <typename T, int R>
__device__ __forceinline__ T wafer_stencil_sum(const T *xs, const T *ys, const T *zs, T w)
{
    if constexpr (R == 1) { // grid.rs:582-588
        return xs[2] + xs[0] + ys[2] + ys[0] + zs[2] + zs[0] - T(6) * w;
    } else if constexpr (R == 2) { // grid.rs:608-620
        // The coefficient 16 is a power of two: 16 * x is exact, so RN(t + 16 * x) in one fused multiply-add is the reference's
        // RN(t + RN(16 * x)) bit for bit (|x| < 2^1019; the library is built with -ffp-contract=off, these six are spelled out):
        // 20 -> 14 instructions per sum.
        T t = wafer_fma_pow2(T(16), xs[3], -xs[4]);
        t = wafer_fma_pow2(T(16), xs[1], t) - xs[0] - ys[4];
        t = wafer_fma_pow2(T(16), ys[3], t);
        t = wafer_fma_pow2(T(16), ys[1], t) - ys[0] - zs[4];
        t = wafer_fma_pow2(T(16), zs[3], t);
        t = wafer_fma_pow2(T(16), zs[1], t) - zs[0];
        return t - T(90) * w;
    } else { // grid.rs:642-659
        // (the coefficient 2 likewise: 37 -> 31)
        T t = T(2) * xs[6] - T(27) * xs[5] + T(270) * xs[4] + T(270) * xs[2] - T(27) * xs[1];
        t = wafer_fma_pow2(T(2), xs[0], t);
        t = wafer_fma_pow2(T(2), ys[6], t) - T(27) * ys[5] + T(270) * ys[4] + T(270) * ys[2] - T(27) * ys[1];
        t = wafer_fma_pow2(T(2), ys[0], t);
        t = wafer_fma_pow2(T(2), zs[6], t) - T(27) * zs[5] + T(270) * zs[4] + T(270) * zs[2] - T(27) * zs[1];
        t = wafer_fma_pow2(T(2), zs[0], t);
        return t - T(1470) * w;
    }
}

// 1 / x for the in-register a, b (potential.rs:104-110).  With `in_range` -- the engine has
// checked 2^-400 < |1 + dt*V/2| < 2^400 over the whole potential -- hipcc's own expansion of the
// fp64 division (v_div_scale x2, v_rcp, 4 fma, mul, fma, v_div_fmas, v_div_fixup) shortens to
// v_rcp + 6 fma WITH THE SAME BITS: v_div_scale returns both operands unchanged for a numerator
// of 1.0 and such a denominator, 1.0 * r is exact, and v_div_fixup passes a finite normal result
// through.  `in_range` is a kernel argument, so the choice is a scalar branch.
__device__ __forceinline__ double wafer_recip(double x, bool in_range)
{
    if (in_range) {
        const double rcp = __builtin_amdgcn_rcp(x);
        const double f0 = __builtin_fma(-x, rcp, 1.0);
        const double f1 = __builtin_fma(rcp, f0, rcp);
        const double f2 = __builtin_fma(-x, f1, 1.0);
        const double r = __builtin_fma(f1, f2, f1);
        const double f4 = __builtin_fma(-x, r, 1.0);
        return __builtin_fma(f4, r, r);
    }
    return 1.0 / x;
}
__device__ __forceinline__ float wafer_recip(float x, bool) { return 1.0f / x; }

// x / den for a loop-invariant divisor the host has NOT planned (norms, projection coefficients).  The fp64 form hoists
// y = RN(1/den) (the compiler moves the one IEEE division out of the loop) and refines q = x*y with two exact remainders: after
// the first, q is a faithful quotient; Markstein's theorem (correctly rounded reciprocal + faithful quotient + exact remainder by
// FMA) makes the second RN(x/den) -- the bits of the IEEE division, in 5 full-rate instructions instead of the 11 of the division
// sequence with its quarter-rate v_rcp_f64.  v_div_fixup restores the IEEE results for zero / infinite / NaN operands.
// Outside the theorem: |x| < 2^-960 (the remainder is no longer exact in the subnormal range; the quotient is then within one
// ulp) -- wavefunction values below 1e-289.  -DWAFER_IEEE_DIV keeps the division.
template <typename T>
__device__ __forceinline__ T wafer_div_invariant(T x, T den)
{
    return x / den;
}
#ifndef WAFER_IEEE_DIV
template <>
__device__ __forceinline__ double wafer_div_invariant<double>(double x, double den)
{
    const double y = 1.0 / den;
    double q = x * y;
    double r = __builtin_fma(-q, den, x);
    q = __builtin_fma(r, y, q);
    r = __builtin_fma(-q, den, x);
    q = __builtin_fma(r, y, q);
    return __builtin_amdgcn_div_fixup(q, den, x);
}
#endif

// x / den for the run's ONE stencil denominator c*dn^2*m (grid.rs:569 / 594 / 626), which the host plans when the context is
// created (wafer_divplan.h, wafer_div_plan): zh = RN(1/den) and zl ~ RN(1/den - zh) arrive as kernel arguments (scalar
// registers), and q = RN(x*zh + RN(x*zl)) -- a multiplication and a fused multiply-add (Brisebarre, Muller, Raina 2004) -- is x/den
// with an error below 2^-52 ulp BEFORE its one rounding: it is RN(x/den) unless x/den lies that close to the midpoint of two
// neighbouring doubles.  For a given den only a few dozen significands x come that close (solutions of X*2^s - M*den_mant = k,
// |k| small: the plan enumerates them all), and the plan tries every one of them, with the very instructions below, against the
// IEEE division: `checked` says that they all came out right -- then the three instructions give the bits of the division for
// EVERY x (|x/den| >= 2^-960; v_div_fixup again for zero / infinite / NaN operands).  Nine divisors in ten pass with the first
// zl tried, the plan may move zl by an ulp or two to get the rest through, and a divisor that still fails (none seen) runs with
// one Markstein round more (q is faithful whatever the plan found, so that round yields RN(x/den) by the theorem above).
// 3 or 5 instructions against 6: the step kernels are bound by their vector issue slots as much as by HBM (profiles/NOTES.md,
// round 5), 16 divisions per wave and plane in the three-step kernel.
template <typename T>
struct WaferDen {
    T den;
};
template <>
struct WaferDen<double> {
    double den, zh, zl;
    bool checked;
};
template <>
struct WaferDen<float> {   // (WAFER_F32_FAST: the same plan in fp32, every significand tried on the host -- wafer_divplan_make_f32)
    float den, zh, zl;
    bool checked;
};
template <typename T>
__device__ __forceinline__ T wafer_div_invariant(T x, const WaferDen<T> &d)
{
#ifdef WAFER_DIV_UNPLANNED   // (A/B builds: the hoisted reciprocal with two remainders, what every division was before the plan)
    return wafer_div_invariant<T>(x, d.den);
#endif
#ifndef WAFER_IEEE_DIV
    if constexpr (std::is_same_v<T, double>) {
        double q = __builtin_fma(x, d.zh, x * d.zl);
        if (!d.checked) {   // (a template argument in the multi-step kernels)
            const double r = __builtin_fma(-q, d.den, x);
            q = __builtin_fma(r, d.zh, q);
        }
        return __builtin_amdgcn_div_fixup(q, d.den, x);
    } else if constexpr (std::is_same_v<T, float>) {
        // 3 instructions against the 11 of the fp32 division sequence (v_div_scale x 2, v_rcp, 4 fma, mul, v_div_fmas, v_div_fixup)
        if (d.checked) return __builtin_amdgcn_div_fixupf(__builtin_fmaf(x, d.zh, x * d.zl), d.den, x);
        return x / d.den;
    } else
#endif
        return x / d.den;
}

// grid.rs:580-589: *work = w*pa + pb*dt*S/denominator
template <typename T>
__device__ __forceinline__ T wafer_update(T w, T pa, T pb, T dt, T S, const WaferDen<T> &den)
{
    return w * pa + wafer_div_invariant<T>(pb * dt * S, den);
}

#define WAFER_MAX_LOW 4 // stored states whose overlaps ride along with the excited-state step
struct WaferLowPtrs {
    const void *p[WAFER_MAX_LOW] = {nullptr, nullptr, nullptr, nullptr};
};

struct WaferStepArgs {
    WaferGeom g;
    int lz_lo, lz_hi;   // local planes [lz_lo, lz_hi) to update
    int zchunk;         // planes marched by one workgroup
    int target_blocks;  // workgroups a launch should aim for (the device's CU count)
    int v_in_range;     // the short arithmetic forms are exact for this run: 2^-400 < |1 + dt*V/2| < 2^400 everywhere (wafer_recip)
                        // AND the plan of x / den is checked (WaferDen)
    // fused kernel, mixed launch (slab interiors): the first n_long workgroups march all of
    // [lz_lo, lz_hi) for one tile each, the remaining tiles are cut into nsub workgroups of zchunk
    // planes; nsub <= 1: every tile is cut into workgroups of zchunk planes
    int n_long, nsub;
    double dt, den;
    double den_zh = 0.0, den_zl = 0.0;   // the plan of x / den (WaferDen, wafer_divplan.h); "checked" travels in v_in_range
    // a launch that is one round of a longer schedule (wafer_f3_by_rounds): its first workgroup's index in the schedule and the
    // schedule's length (0: the launch is the schedule) -- kernels that derive their tile from blockIdx (wafer_k_step2_wide)
    int block0 = 0, nblocks_all = 0;
    float den_zh_f = 0.f, den_zl_f = 0.f;   // the fp32 plan (contexts whose step kernels compute in fp32; 0: none)
    // observables mode of the LDS kernel (NLOW = -2): wafer_potsub_kind and the scalar pot_sub
    int potsub_kind = 0;
    double potsub_scalar = 0.0;
    // kernels that form V from its closed form instead of streaming it (template parameter VG of
    // wafer_k_step_lds): the parameters potential.rs:188-274 reads
    double vg_dn = 0.0, vg_mass = 0.0, vg_sig = 0.0;
};

// short_forms: WaferStepArgs::v_in_range, as the kernel knows it -- a template argument (VIR) in the multi-step kernels, where the
// extra round is then compiled in or out; tested at run time in the single-step ones (the compiler turns that into a select)
template <typename T>
__device__ __forceinline__ WaferDen<T> wafer_den(const WaferStepArgs &a, bool short_forms)
{
    if constexpr (std::is_same_v<T, double>) return WaferDen<double>{a.den, a.den_zh, a.den_zl, short_forms};
    else if constexpr (std::is_same_v<T, float>) return WaferDen<float>{(float)a.den, a.den_zh_f, a.den_zl_f, short_forms && a.den_zh_f != 0.f};
    else return WaferDen<T>{(T)a.den};
}

// ---------------------------------------------------------------------------
// Variant 0: z-marching, register queue along z, x/y neighbours straight from
// global memory (L1/L2).  One work point per lane.  Reference-quality fallback
// and the baseline the LDS-tiled variants are measured against.
// grid: (ceil(nx/64), ceil(ny/4), nchunks), block (64,4).
// If NORM, partials[linear block id] receives the block's sum of out^2.
// ---------------------------------------------------------------------------
template <typename T, typename C, int R, bool NORM>
__global__ __launch_bounds__(256) void wafer_k_step_direct(WaferStepArgs a, const T *__restrict__ phi,
                                                           const T *__restrict__ pa,
                                                           const T *__restrict__ pb, T *__restrict__ out,
                                                           double *__restrict__ partials)
{
    __shared__ double red[4];
    const WaferGeom &g = a.g;
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int j = blockIdx.y * 4 + threadIdx.y;
    const int zs = a.lz_lo + blockIdx.z * a.zchunk;
    const int ze = min(zs + a.zchunk, a.lz_hi);
    const bool active = (i < g.nx) && (j < g.ny);
    const C dt = (C)a.dt;
    const WaferDen<C> den = wafer_den<C>(a, a.v_in_range != 0);
    double acc = 0.0;
    if (active) {
        const long long col = (long long)(j + R) * g.pitch + g.xoff + (i + R);
        const T *p = phi + col;
        C zq[2 * R + 1];
#pragma unroll
        for (int m = 1; m <= 2 * R; ++m) zq[m] = (C)p[(long long)(zs + m - 1 - R) * g.plane];
        for (int z = zs; z < ze; ++z) {
#pragma unroll
            for (int m = 0; m < 2 * R; ++m) zq[m] = zq[m + 1];
            const long long o = (long long)z * g.plane;
            zq[2 * R] = (C)p[o + (long long)R * g.plane];
            C xs[2 * R + 1], ys[2 * R + 1];
#pragma unroll
            for (int d = -R; d <= R; ++d) {
                xs[d + R] = (d == 0) ? zq[R] : (C)p[o + d];
                ys[d + R] = (d == 0) ? zq[R] : (C)p[o + (long long)d * g.pitch];
            }
            const C w = zq[R];
            const C S = wafer_stencil_sum<C, R>(xs, ys, zq, w);
            const C r = wafer_update<C>(w, (C)pa[col + o], (C)pb[col + o], dt, S, den);
            const T rs = (T)r;
            out[col + o] = rs;
            if constexpr (NORM) acc += (double)rs * (double)rs;
        }
    }
    if constexpr (NORM) {
        const int tid = threadIdx.y * 64 + threadIdx.x;
        const double s = wafer_block_sum<4>(acc, red, tid);
        if (tid == 0)
            partials[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = s;
    }
}
