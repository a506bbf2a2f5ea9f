// Norm / normalise / Gram-Schmidt kernels and the fixed-order partial reducer.
//
// Fusion plan for one excited-state step at level k:
//   step kernel (+ sum phi'^2)                      32 B/pt
//   normalise fused with dot_0   (r phi, l0; w phi)  24 B/pt
//   axpy_i fused with dot_{i+1}  (r phi, li, li+1; w phi) 32 B/pt each
//   last axpy                    (r phi, l; w phi)   24 B/pt
// Scalars travel between kernels through device memory (`scal`), never the host.
#pragma once
#include <hip/hip_runtime.h>
#include "wafer_geom.h"
#include "wafer_stencil.hip.h"

struct WaferEwArgs {
    WaferGeom g;
    int lz_lo, lz_hi; // local planes to touch
    int zchunk;
};

// Common thread -> column mapping of the elementwise kernels:
// grid (ceil(nx/64), ceil(ny/4), nchunks), block (64,4), work cells only
// (frame cells are zero in every array, so the reference's whole-padded-array
// passes (grid.rs:467, 483-490) and these work-area passes agree).
#define WAFER_EW_PROLOGUE(R_)                                                          \
    const WaferGeom &g = a.g;                                                          \
    const int i = blockIdx.x * 64 + threadIdx.x;                                       \
    const int j = blockIdx.y * 4 + threadIdx.y;                                        \
    const int zs = a.lz_lo + blockIdx.z * a.zchunk;                                    \
    const int ze = min(zs + a.zchunk, a.lz_hi);                                        \
    const bool active = (i < g.nx) && (j < g.ny);                                      \
    const long long col = (long long)(j + (R_)) * g.pitch + g.xoff + (i + (R_));       \
    const int tid = threadIdx.y * 64 + threadIdx.x;                                    \
    const size_t blin = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;

// get_norm_squared (grid.rs:454-457)
template <typename T>
__global__ __launch_bounds__(256) void wafer_k_norm2(WaferEwArgs a, const T *__restrict__ phi,
                                                     double *__restrict__ partials)
{
    __shared__ double red[4];
    WAFER_EW_PROLOGUE(g.R)
    double acc = 0.0;
    if (active)
        for (int z = zs; z < ze; ++z) {
            const double el = (double)phi[col + (long long)z * g.plane];
            acc += el * el;
        }
    const double s = wafer_block_sum<4>(acc, red, tid);
    if (tid == 0) partials[blin] = s;
}

// normalise_wavefunction (grid.rs:465-468): el /= sqrt(norm2), fused with the
// first Gram-Schmidt overlap sum_l0 (grid.rs:482-487) when `lower` != nullptr.
// norm2 comes from *norm2_dev if non-null, else norm2_imm.
template <typename T, typename C>
__global__ __launch_bounds__(256) void wafer_k_normalise_dot(WaferEwArgs a, T *__restrict__ phi,
                                                             const double *__restrict__ norm2_dev,
                                                             double norm2_imm,
                                                             const T *__restrict__ lower,
                                                             double *__restrict__ partials)
{
    __shared__ double red[4];
    WAFER_EW_PROLOGUE(g.R)
    const double norm2 = norm2_dev ? *norm2_dev : norm2_imm;
    const C norm = (C)sqrt(norm2);
    double acc = 0.0;
    if (active)
        for (int z = zs; z < ze; ++z) {
            const long long p = col + (long long)z * g.plane;
            const T r = (T)((C)phi[p] / norm);
            phi[p] = r;
            if (lower) acc += (double)((C)lower[p] * (C)r);
        }
    if (lower) {
        const double s = wafer_block_sum<4>(acc, red, tid);
        if (tid == 0) partials[blin] = s;
    }
}

// Gram-Schmidt projection *w -= lower * overlap_sum (grid.rs:488-490), fused
// with the NEXT lower state's overlap (modified Gram-Schmidt order is kept:
// the next overlap is taken with the already-projected phi).
template <typename T, typename C>
__global__ __launch_bounds__(256) void wafer_k_axpy_dot(WaferEwArgs a, T *__restrict__ phi,
                                                        const T *__restrict__ lower,
                                                        const double *__restrict__ overlap_dev,
                                                        const T *__restrict__ next,
                                                        double *__restrict__ partials)
{
    __shared__ double red[4];
    WAFER_EW_PROLOGUE(g.R)
    const C s = (C)(*overlap_dev);
    double acc = 0.0;
    if (active)
        for (int z = zs; z < ze; ++z) {
            const long long p = col + (long long)z * g.plane;
            const T r = (T)((C)phi[p] - (C)lower[p] * s);
            phi[p] = r;
            if (next) acc += (double)((C)next[p] * (C)r);
        }
    if (next) {
        const double t = wafer_block_sum<4>(acc, red, tid);
        if (tid == 0) partials[blin] = t;
    }
}

// plain overlap sum_l (lower * phi), used by wafer_orthogonalise's first state
template <typename T, typename C>
__global__ __launch_bounds__(256) void wafer_k_dot(WaferEwArgs a, const T *__restrict__ phi,
                                                   const T *__restrict__ lower,
                                                   double *__restrict__ partials)
{
    __shared__ double red[4];
    WAFER_EW_PROLOGUE(g.R)
    double acc = 0.0;
    if (active)
        for (int z = zs; z < ze; ++z) {
            const long long p = col + (long long)z * g.plane;
            acc += (double)((C)lower[p] * (C)phi[p]);
        }
    const double s = wafer_block_sum<4>(acc, red, tid);
    if (tid == 0) partials[blin] = s;
}

// Fixed-order second stage: out[q] = sum of partials[q*stride .. q*stride+n).
// One 256-thread block per quantity; strided serial sums, then an LDS tree.
__global__ __launch_bounds__(256) void wafer_k_reduce(const double *__restrict__ partials,
                                                      long long n, long long stride,
                                                      double *__restrict__ out)
{
    __shared__ double sh[256];
    const double *p = partials + (size_t)blockIdx.x * stride;
    double s = 0.0;
    for (long long q = threadIdx.x; q < n; q += 256) s += p[q];
    sh[threadIdx.x] = s;
    __syncthreads();
#pragma unroll
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = sh[0];
}

// ---------------------------------------------------------------------------
// Diagnostics: flat streaming kernels over the same buffers, to measure the
// HBM ceiling of this device for (a) a copy and (b) the stencil's stream mix
// (3 reads + 1 write per element) without any neighbour traffic.
// ---------------------------------------------------------------------------
typedef float __attribute__((ext_vector_type(4))) wafer_f4;

template <int NREAD>
__global__ __launch_bounds__(256) void wafer_k_stream(const wafer_f4 *__restrict__ r0,
                                                      const wafer_f4 *__restrict__ r1,
                                                      const wafer_f4 *__restrict__ r2,
                                                      wafer_f4 *__restrict__ w, long long n16)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        wafer_f4 v = r0[i];
        if constexpr (NREAD >= 2) v += r1[i];
        if constexpr (NREAD >= 3) v += r2[i];
        w[i] = v;
    }
}
