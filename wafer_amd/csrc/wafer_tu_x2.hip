// translation unit: two excited-state steps per pass (ThreePoint fp64; wafer_stencil_x2.hip.h)
#include "wafer_launch.h"
#include "wafer_stencil_x2.hip.h"

static WaferX2Ptrs x2_ptrs(int k, const void *const *l, const void *const *m)
{
    WaferX2Ptrs st;
    for (int j = 0; j < k && j < WAFER_X2_MAX_LOW; ++j) {
        st.l[j] = l[j];
        st.m[j] = m[j];
    }
    return st;
}

// tc: WAFER_TC_F64, or WAFER_TC_F32_F64 = fp32 storage with fp64 arithmetic (the arrays are float; vg must be 0)
hipError_t wafer_entry_xstep2(int tc, const WaferTuning &t, const WaferStepArgs &a, int k, int vg, const void *phi, const void *pv, void *out,
                              double *partials, size_t partials_cap, const void *const *l, const void *const *m, const double *coef,
                              hipStream_t s)
{
    if (k < 1 || k > WAFER_X2_MAX_LOW || a.v_in_range == 0) return hipErrorInvalidValue;
    if (tc == WAFER_TC_F32_F64)
        return vg != 0 ? hipErrorInvalidValue
                       : wafer_launch_xstep2<wafer_f32_wide>(t, a, k, 0, static_cast<const float *>(phi), static_cast<const float *>(pv), static_cast<float *>(out),
                                                             partials, partials_cap, x2_ptrs(k, l, m), coef, s);
    if (tc != WAFER_TC_F64) return hipErrorInvalidValue;
    return wafer_launch_xstep2<double>(t, a, k, vg, static_cast<const double *>(phi), static_cast<const double *>(pv), static_cast<double *>(out),
                                       partials, partials_cap, x2_ptrs(k, l, m), coef, s);
}

hipError_t wafer_entry_x2_coeffs(int kind, int k, const double *sums, const double *gram, const double *amat, double *coef, hipStream_t s)
{
    hipLaunchKernelGGL(wafer_k_x2_coeffs, dim3(1), dim3(64), 0, s, kind, k, sums, gram, amat, coef);
    return hipGetLastError();
}

hipError_t wafer_entry_x2_apply(int tc, const WaferGeom &g, int lz_lo, int lz_hi, int k, void *phi, const void *const *l, const void *const *m,
                                const double *coef, double *partials, size_t partials_cap, int num_cus, hipStream_t s, int *nblocks_out)
{
    WaferRowArgs ra;
    ra.g = g;
    ra.lz_lo = lz_lo;
    ra.lz_hi = lz_hi;
    if (tc == WAFER_TC_F32_F64)
        return wafer_launch_x2_apply<wafer_f32_wide>(ra, k, static_cast<float *>(phi), x2_ptrs(k, l, m), coef, partials, partials_cap, num_cus, s, nblocks_out);
    if (tc != WAFER_TC_F64) return hipErrorInvalidValue;
    return wafer_launch_x2_apply<double>(ra, k, static_cast<double *>(phi), x2_ptrs(k, l, m), coef, partials, partials_cap, num_cus, s, nblocks_out);
}

long long wafer_entry_x2_blocks(int tc, const WaferTuning &t, const WaferGeom &g, int k, int vg, int lz_lo, int lz_hi, int target_blocks)
{
    return wafer_x2_blocks(t, g, k, vg, lz_lo, lz_hi, target_blocks, tc == WAFER_TC_F32_F64);
}

int wafer_entry_x2_nsums(int k) { return wafer_x2_nsums(k); }
void wafer_x2_tile_host(int tc, const WaferTuning &t, int k, int vg, int *tx, int *ty) { wafer_x2_tile(t, k, vg, tx, ty, tc == WAFER_TC_F32_F64); }
