// translation unit: three fused ground-state steps (ThreePoint fp64)
#include "wafer_launch.h"
#include "wafer_stencil_fused3.hip.h"

hipError_t wafer_entry_step3_fused(const WaferTuning &t, const WaferStepArgs &a, const WaferF3Block *table, int nblocks,
                                   const WaferF3Sync &sy, const double *phi, const double *pv, double *out, hipStream_t s)
{
    return wafer_launch_step3_fused<double, double>(t, a, table, nblocks, sy, phi, pv, out, s);
}
