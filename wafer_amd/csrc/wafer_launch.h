// Non-template entry points of the kernel families.  Each family is compiled in its own translation unit
// (wafer_tu_*.hip) so that the device code builds in parallel and an edit to one kernel rebuilds one unit; the engine
// (wafer_engine.hip) holds the host logic and the small elementwise / set-up kernels.
//
// tc: storage / arithmetic types -- 0: fp64 / fp64, 1: fp32 storage / fp64 arithmetic, 2: fp32 / fp32 (ground-state
// steps of WAFER_F32_FAST only).  R: CentralDifference::ext().  Array pointers are the engine's logical pointers
// (plane 0, row 0) of that storage type.
#pragma once
#include <hip/hip_runtime.h>
#include "wafer_stencil.hip.h"
#include "wafer_tuning.h"

struct WaferF3Block;
struct WaferF3Sync;

enum { WAFER_TC_F64 = 0, WAFER_TC_F32_F64 = 1, WAFER_TC_F32_F32 = 2 };

// one ground-state step on the LDS pipeline (wafer_stencil_lds.hip.h)
hipError_t wafer_entry_step_lds(int tc, int R, const WaferTuning &t, const WaferStepArgs &a, const void *phi, const void *pa,
                                const void *pb, const void *pv, void *out, hipStream_t s, int vg);
// one excited-state step with nlow raw overlaps (nlow == 0: the norm only); one translation unit per stencil order
#define WAFER_DECL_EXCITED(R_)                                                                                                          \
    hipError_t wafer_entry_step_lds_excited_r##R_(int tc, const WaferTuning &t, const WaferStepArgs &a, const void *phi, const void *pv, \
                                                   void *out, double *partials, size_t partials_cap, int nlow, const WaferLowPtrs &low,   \
                                                   hipStream_t s, const double *xscal, const double *xgram, int vg);
WAFER_DECL_EXCITED(1)
WAFER_DECL_EXCITED(2)
WAFER_DECL_EXCITED(3)
#undef WAFER_DECL_EXCITED
static inline hipError_t wafer_entry_step_lds_excited(int tc, int R, const WaferTuning &t, const WaferStepArgs &a, const void *phi,
                                                      const void *pv, void *out, double *partials, size_t partials_cap, int nlow,
                                                      const WaferLowPtrs &low, hipStream_t s, const double *xscal, const double *xgram, int vg)
{
    switch (R) {
    case 1: return wafer_entry_step_lds_excited_r1(tc, t, a, phi, pv, out, partials, partials_cap, nlow, low, s, xscal, xgram, vg);
    case 2: return wafer_entry_step_lds_excited_r2(tc, t, a, phi, pv, out, partials, partials_cap, nlow, low, s, xscal, xgram, vg);
    case 3: return wafer_entry_step_lds_excited_r3(tc, t, a, phi, pv, out, partials, partials_cap, nlow, low, s, xscal, xgram, vg);
    default: return hipErrorInvalidValue;
    }
}
// compute_observables on the LDS pipeline (storage type only: the sums are fp64)
hipError_t wafer_entry_observables_lds(int tc, int R, const WaferTuning &t, const WaferStepArgs &a, const void *phi, const void *pv,
                                       const void *potsub, double *partials, size_t partials_cap, hipStream_t s,
                                       long long *nblocks_out, int vg);
// two fused ground-state steps
hipError_t wafer_entry_step2_fused(int tc, int R, const WaferTuning &t, const WaferStepArgs &a, const void *phi, const void *pa,
                                   const void *pb, const void *pv, void *out, hipStream_t s);
// ... FivePoint on 128 x 16 tiles, eight even waves (wafer_stencil_fused2w.hip.h)
hipError_t wafer_entry_step2_wide(int tc, const WaferTuning &t, const WaferStepArgs &a, const void *phi, const void *pv, void *out, hipStream_t s);
// three fused ground-state steps (ThreePoint; every type combination), table-driven
// dir: 1 = every workgroup of the table marches up, 2 = every one down, 0 = both occur (picks the kernel that carries only the
// copy of the plane loop it needs)
hipError_t wafer_entry_step3_fused(int tc, const WaferTuning &t, const WaferStepArgs &a, const WaferF3Block *table, int nblocks,
                                   const WaferF3Sync &sy, const void *phi, const void *pv, void *out, hipStream_t s, int dir = 0);
hipError_t wafer_entry_step3_fused_wide(const WaferTuning &t, const WaferStepArgs &a, const WaferF3Block *table, int nblocks,
                                        const WaferF3Sync &sy, const void *phi, const void *pv, void *out, hipStream_t s, int dir);
void wafer_step3_tile(int tc, int *tx, int *ty);
// the template-id of the instantiation this thread's last wafer_entry_step3_fused launched, as rocprofv3 prints it
// ("wafer_k_step3_fused<double, double, true, 0, true, 1>"); empty before the first launch
void wafer_step3_last_instance(char *buf, size_t n);

// two excited-state steps per pass (ThreePoint, fp64 or fp32 storage with fp64 arithmetic -- tc --, 1 <= k <= 3 stored states; wafer_stencil_x2.hip.h): out = A A x with x the
// load transform of phi by `coef`; l / m: the stored states and their images M_j = A l_j; the 1 + 2k sums of the pass go to
// partials[q * partials_cap + workgroup]
hipError_t wafer_entry_xstep2(int tc, const WaferTuning &t, const WaferStepArgs &a, int k, int vg, const void *phi, const void *pv, void *out,
                              double *partials, size_t partials_cap, const void *const *l, const void *const *m, const double *coef,
                              hipStream_t s);
// the transform's coefficients from the sums of the last pass (kind 2) or of the last one-step kernel (kind 1)
hipError_t wafer_entry_x2_coeffs(int kind, int k, const double *sums, const double *gram, const double *amat, double *coef, hipStream_t s);
// phi materialised in place after the last pass, up to the last step's norm: its square is summed into partials[0 .. *nblocks_out)
hipError_t wafer_entry_x2_apply(int tc, const WaferGeom &g, int lz_lo, int lz_hi, int k, void *phi, const void *const *l, const void *const *m,
                                const double *coef, double *partials, size_t partials_cap, int num_cus, hipStream_t s, int *nblocks_out);
long long wafer_entry_x2_blocks(int tc, const WaferTuning &t, const WaferGeom &g, int k, int vg, int lz_lo, int lz_hi, int target_blocks);
int wafer_entry_x2_nsums(int k);
void wafer_x2_tile_host(int tc, const WaferTuning &t, int k, int vg, int *tx, int *ty);   // the kernel's tile for k stored states
