// The row-vectorised walk of the elementwise kernels (wafer_elementwise.hip.h, wafer_stencil_x2.hip.h).
#pragma once
#include "wafer_geom.h"

// ---------------------------------------------------------------------------
// Row-vectorised elementwise kernels: each wave walks 1 KiB row segments
// (16 B per lane, 128 B-aligned: wafer_geom.h) of the work area in memory order.
// ---------------------------------------------------------------------------
struct WaferRowArgs {
    WaferGeom g;
    int lz_lo, lz_hi;
};

template <typename T> struct WaferRowVec;
template <> struct WaferRowVec<double> { static constexpr int N = 2; typedef double __attribute__((ext_vector_type(2))) type; };
template <> struct WaferRowVec<float> { static constexpr int N = 4; typedef float __attribute__((ext_vector_type(4))) type; };

// rows [0, (lz_hi - lz_lo) * ny) dealt over the waves of the grid, four waves per workgroup; inside: `rowp`, the element
// offset of the row's first work cell
#define WAFER_ROW_WALK_BEGIN(a, g)                                                                                  \
    {                                                                                                               \
        const int rows_total_ = ((a).lz_hi - (a).lz_lo) * (g).ny, stride_ = (int)gridDim.x * 4;                     \
        int row_ = (int)blockIdx.x * 4 + wave;                                                                      \
        int y_ = row_ % (g).ny, z_ = row_ / (g).ny;                                                                 \
        const int sy_ = stride_ % (g).ny, sz_ = stride_ / (g).ny;                                                   \
        for (; row_ < rows_total_; row_ += stride_) {                                                               \
            const long long rowp = (long long)((a).lz_lo + z_) * (g).plane + (long long)(y_ + (g).R) * (g).pitch + (g).xoff + (g).R;
#define WAFER_ROW_WALK_END(g)                                                                                       \
            y_ += sy_;                                                                                              \
            z_ += sz_;                                                                                              \
            if (y_ >= (g).ny) { y_ -= (g).ny; ++z_; }                                                               \
        }                                                                                                           \
    }

