// translation unit: excited-state step kernels, ext = 3 (one unit per stencil order: they are the bulk of the device code)
#define WAFER_TU_EXCITED_R 3
#include "wafer_tu_excited.inc"
