// translation unit: single ground-state step and compute_observables on the LDS pipeline
#include "wafer_launch.h"
#include "wafer_stencil_lds.hip.h"

template <typename T, typename C>
static hipError_t step_r(int R, const WaferTuning &t, const WaferStepArgs &a, const void *phi, const void *pa, const void *pb,
                         const void *pv, void *out, hipStream_t s, int vg)
{
    const T *p = static_cast<const T *>(phi), *a_ = static_cast<const T *>(pa), *b_ = static_cast<const T *>(pb), *v_ = static_cast<const T *>(pv);
    T *o = static_cast<T *>(out);
    switch (R) {
    case 1: return wafer_launch_step_lds<T, C, 1>(t, a, p, a_, b_, v_, o, s, vg);
    case 2: return wafer_launch_step_lds<T, C, 2>(t, a, p, a_, b_, v_, o, s, vg);
    case 3: return wafer_launch_step_lds<T, C, 3>(t, a, p, a_, b_, v_, o, s, vg);
    default: return hipErrorInvalidValue;
    }
}

hipError_t wafer_entry_step_lds(int tc, int R, const WaferTuning &t, const WaferStepArgs &a, const void *phi, const void *pa,
                                const void *pb, const void *pv, void *out, hipStream_t s, int vg)
{
    switch (tc) {
    case WAFER_TC_F64: return step_r<double, double>(R, t, a, phi, pa, pb, pv, out, s, vg);
    case WAFER_TC_F32_F64: return step_r<float, double>(R, t, a, phi, pa, pb, pv, out, s, vg);
    case WAFER_TC_F32_F32: return step_r<float, float>(R, t, a, phi, pa, pb, pv, out, s, vg);
    default: return hipErrorInvalidValue;
    }
}

template <typename T>
static hipError_t obs_r(int R, const WaferTuning &t, const WaferStepArgs &a, const void *phi, const void *pv, const void *potsub,
                        double *partials, size_t cap, hipStream_t s, long long *nb, int vg)
{
    const T *p = static_cast<const T *>(phi), *v_ = static_cast<const T *>(pv), *ps = static_cast<const T *>(potsub);
    switch (R) {
    case 1: return wafer_launch_observables_lds<T, 1>(t, a, p, v_, ps, partials, cap, s, nb, vg);
    case 2: return wafer_launch_observables_lds<T, 2>(t, a, p, v_, ps, partials, cap, s, nb, vg);
    case 3: return wafer_launch_observables_lds<T, 3>(t, a, p, v_, ps, partials, cap, s, nb, vg);
    default: return hipErrorInvalidValue;
    }
}

hipError_t wafer_entry_observables_lds(int tc, int R, const WaferTuning &t, const WaferStepArgs &a, const void *phi, const void *pv,
                                       const void *potsub, double *partials, size_t partials_cap, hipStream_t s,
                                       long long *nblocks_out, int vg)
{
    return tc == WAFER_TC_F64 ? obs_r<double>(R, t, a, phi, pv, potsub, partials, partials_cap, s, nblocks_out, vg)
                              : obs_r<float>(R, t, a, phi, pv, potsub, partials, partials_cap, s, nblocks_out, vg);
}
