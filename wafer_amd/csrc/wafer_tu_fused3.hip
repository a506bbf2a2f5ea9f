// translation unit: three fused ground-state steps (ThreePoint; fp64, and fp32 storage with fp32 step arithmetic; fp32 storage with
// fp64 arithmetic: wafer_tu_fused3_wide.hip)
#include <cstdio>
#include "wafer_launch.h"
#include "wafer_stencil_fused3.hip.h"

hipError_t wafer_entry_step3_fused(int tc, const WaferTuning &t, const WaferStepArgs &a, const WaferF3Block *table, int nblocks,
                                   const WaferF3Sync &sy, const void *phi, const void *pv, void *out, hipStream_t s, int dir)
{
    switch (tc) {
    case WAFER_TC_F64:
        return wafer_launch_step3_fused<double, double>(t, a, table, nblocks, sy, static_cast<const double *>(phi), static_cast<const double *>(pv),
                                                        static_cast<double *>(out), s, dir);
    case WAFER_TC_F32_F32:   // a and b ride between the levels in the arithmetic type: fp32 here, as in the two-step kernel
        return wafer_launch_step3_fused<float, float>(t, a, table, nblocks, sy, static_cast<const float *>(phi), static_cast<const float *>(pv),
                                                      static_cast<float *>(out), s, dir);
    case WAFER_TC_F32_F64:   // fp32 storage, fp64 arithmetic: float in HBM, the fp64 kernel's registers and LDS (its own unit: wafer_tu_fused3_wide.hip)
        return wafer_entry_step3_fused_wide(t, a, table, nblocks, sy, phi, pv, out, s, dir);
    default:
        return hipErrorInvalidValue;
    }
}

// tile of the kernel for a type combination (the host builds the workgroup tables from it)
void wafer_step3_tile(int tc, int *tx, int *ty)
{
    if (tc == WAFER_TC_F32_F32) { *tx = WaferF3Cfg<float>::TX; *ty = WaferF3Cfg<float>::TY; }
    else { *tx = WaferF3Cfg<double>::TX; *ty = WaferF3Cfg<double>::TY; }   // (fp32 storage with fp64 arithmetic runs the fp64 kernel's tile)
}

void wafer_step3_last_instance(char *buf, size_t n)
{
    const WaferF3Instance &li = wafer_f3_last_instance();
    if (n == 0) return;
    buf[0] = 0;
    if (li.tsize == 0) return;
    snprintf(buf, n, "wafer_k_step3_fused<%s, %s, %s, %d, %s, %d>", li.tsize == 8 ? "double" : li.tsize == -4 ? "wafer_f32_wide" : "float", li.csize == 8 ? "double" : "float",
             li.vir ? "true" : "false", li.mode, li.xs ? "true" : "false", li.dir);
}

#if WAFER_DIAG & 1
// diagnostic builds only: the per-wave cycle sums of the last launch's stamped workgroup
extern "C" int wafer_debug_f3_stamps(unsigned long long *host_out)
{
    (void)hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(wafer_f3_stamp_buf), sizeof(unsigned long long) * 8 * WAFER_F3_NSTAMP);
}
#endif
