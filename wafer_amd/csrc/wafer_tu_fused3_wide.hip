// translation unit: three fused ground-state steps on fp32 STORAGE with fp64 arithmetic (dtype WAFER_F32; wafer_f32_wide in
// wafer_stencil_fused3.hip.h) -- a unit of its own so that it compiles beside the fp64 / all-fp32 instantiations
#include "wafer_launch.h"
#include "wafer_stencil_fused3.hip.h"

hipError_t wafer_entry_step3_fused_wide(const WaferTuning &t, const WaferStepArgs &a, const WaferF3Block *table, int nblocks,
                                        const WaferF3Sync &sy, const void *phi, const void *pv, void *out, hipStream_t s, int dir)
{
    return wafer_launch_step3_fused<wafer_f32_wide, double>(t, a, table, nblocks, sy, static_cast<const float *>(phi), static_cast<const float *>(pv),
                                                            static_cast<float *>(out), s, dir);
}
