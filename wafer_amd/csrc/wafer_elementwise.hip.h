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

// Fixed-order second stage: out[q] = sum of partials[q*stride .. q*stride+n).
// One 256-thread block per quantity; strided serial sums, then an LDS tree.
static __global__ __launch_bounds__(256) void wafer_k_reduce(const double *__restrict__ partials,
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

#include "wafer_rowwalk.h"

// The once-per-block elementwise passes of solve (grid.rs:126-135) and of the kernel-per-projection
// excited-state path, all on the row-vectorised walk (16 B per lane):
//   OP 0  get_norm_squared (grid.rs:454-457):            partial sums of phi^2
//   OP 1  overlap sum(lower * phi) (grid.rs:482-487)
//   OP 2  normalise_wavefunction (grid.rs:465-468): phi /= sqrt(norm2), fused with the first overlap
//         sum(lower * phi) when `lower` is given; norm2 = *scal_dev if non-null, else imm
//   OP 3  Gram-Schmidt projection phi -= lower * s (grid.rs:488-490), s = *scal_dev, fused with the NEXT
//         state's overlap (taken with the already-projected phi: modified Gram-Schmidt order is kept)
// Work cells only: frame cells are zero in every array, so the reference's whole-padded-array passes
// and these work-area passes agree.  One partial per workgroup, summed in a fixed order.
template <typename T, typename C, int OP>
__global__ __launch_bounds__(256) void wafer_k_row_op(WaferRowArgs a, T *__restrict__ phi, const T *__restrict__ lower,
                                                      const T *__restrict__ next, const double *__restrict__ scal_dev,
                                                      double imm, double *__restrict__ partials)
{
    using VT = typename WaferRowVec<T>::type;
    constexpr int VEC = WaferRowVec<T>::N;
    __shared__ double red[4];
    const WaferGeom &g = a.g;
    const double sval = scal_dev ? *scal_dev : imm;
    const C coef = (OP == 2) ? (C)sqrt(sval) : (C)sval;
    const T *dotwith = (OP == 1 || OP == 2) ? lower : (OP == 3 ? next : nullptr);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nsegx = (g.nx + 64 * VEC - 1) / (64 * VEC);
    const int wlim = g.pitch - g.xoff - g.R;
    double acc = 0.0;
    // A wave walks whole rows, (y, z) advanced by addition: the walk used to be over 1 KiB segments with two 64-bit
    // divisions per segment to find (x, y, z) -- a hundred-odd instructions beside one load and one store (0.76 ms for the
    // projection's 3.2 GB at 512^3, round 2).
    WAFER_ROW_WALK_BEGIN(a, g)
    for (int xs = 0; xs < nsegx; ++xs) {
        const int xi = xs * 64 * VEC + lane * VEC;
        if (xi >= wlim || xi >= g.nx) continue;
        const long long p = rowp + xi;
        VT w = *reinterpret_cast<const VT *>(phi + p);
        if constexpr (OP == 2) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) w[v] = (T)wafer_div_invariant<C>((C)w[v], coef);
        }
        if constexpr (OP == 3) {
            const VT l = __builtin_nontemporal_load(reinterpret_cast<const VT *>(lower + p));
#pragma unroll
            for (int v = 0; v < VEC; ++v) w[v] = (T)((C)w[v] - (C)l[v] * coef);
        }
        if constexpr (OP >= 2) {
            if (xi + VEC <= g.nx) {
                *reinterpret_cast<VT *>(phi + p) = w;
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    if (xi + v < g.nx) phi[p + v] = w[v];
            }
        }
        if constexpr (OP == 0) {
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                if (xi + v < g.nx) acc += (double)w[v] * (double)w[v];
        } else {
            if (dotwith) {
                const VT d = __builtin_nontemporal_load(reinterpret_cast<const VT *>(dotwith + p));
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    if (xi + v < g.nx) acc += (double)((C)d[v] * (C)w[v]);
            }
        }
    }
    WAFER_ROW_WALK_END(g)
    if (OP <= 1 || dotwith) {
        const double s = wafer_block_sum<4>(acc, red, threadIdx.x);
        if (threadIdx.x == 0) partials[blockIdx.x] = s;
    }
}

// Normalise + the whole modified Gram-Schmidt chain of one excited-state step
// (grid.rs:679-680) in ONE pass.  The step kernel has left in scal[]:
//   scal[0] = sum phi'^2,  scal[1+j] = t_j = sum l_j * phi'   (un-normalised phi').
// With norm = sqrt(scal[0]) and the Gram matrix G_ji = sum l_j * l_i of the
// stored states (kept current by the engine), the reference's sequential overlaps
//   s_j = sum l_j * (phi'/norm - sum_{i<j} l_i s_i)            (grid.rs:482-487)
// are s_j = t_j/norm - sum_{i<j} s_i G_ji  -- modified Gram-Schmidt exactly, not
// classical: only the association of the sums differs (rel ~1e-16, inside the
// 1e-12 bar on reductions).  Per cell the reference's own operation order is kept:
// w /= norm (true division), then w -= l_j * s_j for j = 0..k-1.
template <typename T, typename C, int NLOW>
__global__ __launch_bounds__(256) void wafer_k_gs_apply(WaferRowArgs a, T *__restrict__ phi, WaferLowPtrs low,
                                                        const double *__restrict__ scal,
                                                        const double *__restrict__ gram)
{
    using VT = typename WaferRowVec<T>::type;
    constexpr int VEC = WaferRowVec<T>::N;
    const WaferGeom &g = a.g;
    const C norm = (C)sqrt(scal[0]);
    C sj[NLOW];
#pragma unroll
    for (int j = 0; j < NLOW; ++j) {
        double s = scal[1 + j] / (double)norm;
#pragma unroll
        for (int i = 0; i < j; ++i) s -= (double)sj[i] * gram[j * WAFER_MAX_LOW + i];
        sj[j] = (C)s;
    }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nsegx = (g.nx + 64 * VEC - 1) / (64 * VEC);
    const int wlim = g.pitch - g.xoff - g.R;
    WAFER_ROW_WALK_BEGIN(a, g)
    for (int xs = 0; xs < nsegx; ++xs) {
        const int xi = xs * 64 * VEC + lane * VEC;
        if (xi >= wlim || xi >= g.nx) continue;
        const long long p = rowp + xi;
        VT w = *reinterpret_cast<const VT *>(phi + p);
        VT l[NLOW];
#pragma unroll
        for (int j = 0; j < NLOW; ++j) l[j] = __builtin_nontemporal_load(reinterpret_cast<const VT *>(static_cast<const T *>(low.p[j]) + p));
        VT r;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            C x = wafer_div_invariant<C>((C)w[v], norm); // grid.rs:467
#pragma unroll
            for (int j = 0; j < NLOW; ++j) x = x - (C)l[j][v] * sj[j]; // grid.rs:488-490
            r[v] = (T)x;
        }
        if (xi + VEC <= g.nx) {
            *reinterpret_cast<VT *>(phi + p) = r;
        } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                if (xi + v < g.nx) phi[p + v] = r[v];
        }
    }
    WAFER_ROW_WALK_END(g)
}

// scal[0] = 1, scal[1..n) = 0: the scalars for which the load transform / wafer_k_gs_apply is the
// identity (x/1 - l*0), i.e. "phi is already normalised and projected"
static __global__ void wafer_k_identity_scalars(double *__restrict__ scal, int n)
{
    const int i = threadIdx.x;
    if (i < n) scal[i] = (i == 0) ? 1.0 : 0.0;
}

// ---------------------------------------------------------------------------
// Diagnostic: the device's copy ceiling (MI355X_MICROARCH.md: ~6.3 TB/s for a float4 copy): every lane moves
// U independent 16 B vectors per trip of a grid-stride loop -- U loads in flight before the first
// store -- with a grid of a few 256-thread workgroups per CU.
// ---------------------------------------------------------------------------
typedef float __attribute__((ext_vector_type(4))) wafer_f4;

template <int U>
__global__ __launch_bounds__(256) void wafer_k_copy16(const wafer_f4 *__restrict__ src, wafer_f4 *__restrict__ dst, long long n16)
{
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        wafer_f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], dst + i + u * stride);
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

// ---- diagnostic: position-dependent checksum of the work cells of global planes [kz_lo, kz_hi) that
// this slab owns: sum (mod 2^64) of mix(bits(cell) ^ mix(global linear index)).  An integer sum is
// order independent, so a slab-decomposed run and an undecomposed one must give the same value plane
// range by plane range -- a bit-exactness test that moves 8 bytes instead of the grid.
__device__ __forceinline__ unsigned long long wafer_hash64(unsigned long long z)
{
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
template <typename T>
__global__ __launch_bounds__(256) void wafer_k_checksum(WaferRowArgs a, const T *__restrict__ phi, int kz_lo, int kz_hi,
                                                        unsigned long long *__restrict__ out)
{
    const WaferGeom &g = a.g;
    const int lz0 = max(a.lz_lo, kz_lo - g.z_begin + g.G), lz1 = min(a.lz_hi, kz_hi - g.z_begin + g.G);
    unsigned long long acc = 0;
    const long long rows = (long long)max(0, lz1 - lz0) * g.ny;
    for (long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long long)gridDim.x * 4) {
        const int y = (int)(row % g.ny), z = lz0 + (int)(row / g.ny);
        const long long kg = (long long)g.z_begin + (z - g.G);
        const long long p0 = (long long)z * g.plane + (long long)(y + g.R) * g.pitch + g.xoff + g.R;
        const unsigned long long lin0 = ((unsigned long long)kg * (unsigned long long)g.ny + (unsigned long long)y) * (unsigned long long)g.nx;
        for (int x = threadIdx.x & 63; x < g.nx; x += 64) {
            unsigned long long bits;
            if constexpr (sizeof(T) == 8) bits = (unsigned long long)__double_as_longlong((double)phi[p0 + x]);
            else bits = (unsigned long long)__float_as_uint((float)phi[p0 + x]);
            acc += wafer_hash64(bits ^ wafer_hash64(lin0 + (unsigned long long)x + 0x9e3779b97f4a7c15ull));
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

// ---- diagnostic: wafer_div_invariant against the IEEE division ------------------------------------
// Every thread draws `per_thread` operands x from a counter-based generator (splitmix64 of the global
// operand index and the seed): uniform significand and sign, biased exponent uniform in [lo_exp, hi_exp].
// Counts the operands for which wafer_div_invariant(x, den) and x / den differ in any bit.
template <bool PLANNED>
static __global__ __launch_bounds__(256) void wafer_k_div_check(WaferDen<double> dv, unsigned long long seed, int per_thread, int lo_exp,
                                                         int hi_exp, unsigned long long *__restrict__ mismatches)
{
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long bad = 0;
    for (int i = 0; i < per_thread; ++i) {
        unsigned long long z = seed + (tid * (unsigned long long)per_thread + (unsigned long long)i) * 0x9e3779b97f4a7c15ull;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        z ^= z >> 31;
        const unsigned long long e = (unsigned long long)lo_exp + (z >> 53) % (unsigned long long)(hi_exp - lo_exp + 1);
        const unsigned long long bits = (z & 0x800fffffffffffffull) | (e << 52);
        const double x = __longlong_as_double((long long)bits);
        const double q_fast = PLANNED ? wafer_div_invariant<double>(x, dv) : wafer_div_invariant<double>(x, dv.den);
        const double q_ieee = x / dv.den;
        bad += (unsigned long long)(__double_as_longlong(q_fast) != __double_as_longlong(q_ieee));
    }
    if (bad) atomicAdd(mismatches, bad);
}
// ... the fp32 plan on EVERY float of the biased exponents [lo_exp, hi_exp]: all 2^23 significands, both signs
static __global__ __launch_bounds__(256) void wafer_k_div_check_f32(WaferDen<float> dv, int lo_exp, int hi_exp, unsigned long long *__restrict__ mismatches)
{
    const unsigned long long n = (unsigned long long)(hi_exp - lo_exp + 1) << 24;   // exponent, sign, significand
    unsigned long long bad = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned int bits = ((unsigned int)(i & 0x7fffffu)) | ((unsigned int)(lo_exp + (int)(i >> 24)) << 23) | ((unsigned int)((i >> 23) & 1u) << 31);
        const float x = __uint_as_float(bits);
        const float q_fast = wafer_div_invariant<float>(x, dv);
        const float q_ieee = x / dv.den;
        bad += (unsigned long long)(__float_as_uint(q_fast) != __float_as_uint(q_ieee));
    }
    if (bad) atomicAdd(mismatches, bad);
}
// ... the planned division on a list of operands (the plan's candidates, scaled over the exponent range by the caller)
static __global__ __launch_bounds__(256) void wafer_k_div_operands(WaferDen<double> dv, const double *__restrict__ x, unsigned long long n,
                                                            unsigned long long *__restrict__ mismatches)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double q_fast = wafer_div_invariant<double>(x[i], dv);
    const double q_ieee = x[i] / dv.den;
    if (__double_as_longlong(q_fast) != __double_as_longlong(q_ieee)) atomicAdd(mismatches, 1ull);
}
