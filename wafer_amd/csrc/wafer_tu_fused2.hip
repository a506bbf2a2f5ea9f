// translation unit: two fused ground-state steps
#include "wafer_launch.h"
#include "wafer_stencil_fused2.hip.h"

template <typename T, typename C>
static hipError_t f2_r(int R, const WaferTuning &t, const WaferStepArgs &a, const void *phi, const void *pa, const void *pb, const void *pv,
                       void *out, hipStream_t s)
{
    const T *p = static_cast<const T *>(phi), *a_ = static_cast<const T *>(pa), *b_ = static_cast<const T *>(pb), *v_ = static_cast<const T *>(pv);
    T *o = static_cast<T *>(out);
    switch (R) {
    case 1: return wafer_launch_step2_fused<T, C, 1>(t, a, p, a_, b_, v_, o, s);
    case 2: return wafer_launch_step2_fused<T, C, 2>(t, a, p, a_, b_, v_, o, s);
    case 3: return wafer_launch_step2_fused<T, C, 3>(t, a, p, a_, b_, v_, o, s);
    default: return hipErrorInvalidValue;
    }
}

hipError_t wafer_entry_step2_fused(int tc, int R, const WaferTuning &t, const WaferStepArgs &a, const void *phi, const void *pa,
                                   const void *pb, const void *pv, void *out, hipStream_t s)
{
    // FivePoint: the 128 x 16-tile kernel in the three-step kernel's structure (wafer_stencil_fused2w.hip.h); WAFER_F2_WIDE=0: this unit's
    if (R == 2 && t.f2_wide != 0) return wafer_entry_step2_wide(tc, t, a, phi, pv, out, s);
    switch (tc) {
    case WAFER_TC_F64: return f2_r<double, double>(R, t, a, phi, pa, pb, pv, out, s);
    case WAFER_TC_F32_F64: return f2_r<float, double>(R, t, a, phi, pa, pb, pv, out, s);
    case WAFER_TC_F32_F32: return f2_r<float, float>(R, t, a, phi, pa, pb, pv, out, s);
    default: return hipErrorInvalidValue;
    }
}
