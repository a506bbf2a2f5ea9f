// translation unit: two fused FivePoint ground-state steps on 128 x 16 tiles (wafer_stencil_fused2w.hip.h)
#include "wafer_launch.h"
#include "wafer_stencil_fused2w.hip.h"

hipError_t wafer_entry_step2_wide(int tc, const WaferTuning &t, const WaferStepArgs &a, const void *phi, const void *pv, void *out, hipStream_t s)
{
    switch (tc) {
    case WAFER_TC_F64:
        return wafer_launch_step2_wide<double, double>(t, a, static_cast<const double *>(phi), static_cast<const double *>(pv), static_cast<double *>(out), s);
    case WAFER_TC_F32_F64:
        return wafer_launch_step2_wide<wafer_f32_wide, double>(t, a, static_cast<const float *>(phi), static_cast<const float *>(pv), static_cast<float *>(out), s);
    case WAFER_TC_F32_F32:
        return wafer_launch_step2_wide<float, float>(t, a, static_cast<const float *>(phi), static_cast<const float *>(pv), static_cast<float *>(out), s);
    default:
        return hipErrorInvalidValue;
    }
}
