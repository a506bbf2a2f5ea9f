// Setup kernels: built-in potentials + a/b, pot_sub, initial conditions and
// the layout transposes between the reference's [x][y][z] host arrays and the
// device's [z][y][x] slabs.  None of these is on the per-step path; they exist
// so that 1024^3 / 2048^3 slabs are generated in HBM instead of crossing PCIe.
#pragma once
#include <hip/hip_runtime.h>
#include "wafer_geom.h"

#define WAFER_PI 3.14159265358979323846264338327950288

struct WaferPotArgs {
    WaferGeom g;
    int type;            // wafer_potential
    double dn, dt, mass, sig;
    // FullCornell constants, evaluated on the host exactly as potential.rs:252-260, 264
    double mu_t;         // mu(t), t = 1
    double alphas_2pit;  // alphas(2*pi*t)
    double xi_coef;      // 0.07 * xi^0.2, xi = 0
    double xi_fac;       // (1 + xi)^-0.29
};

// potential.rs:366-371
__device__ __forceinline__ double wafer_r2(int ix, int iy, int iz, int nx, int ny, int nz)
{
    const double dx = (double)ix - ((double)nx + 1.) / 2.;
    const double dy = (double)iy - ((double)ny + 1.) / 2.;
    const double dz = (double)iz - ((double)nz + 1.) / 2.;
    return dx * dx + dy * dy + dz * dz;
}

// potential.rs:283-308, the reference's grouping of every half-space test
__device__ __forceinline__ bool wafer_in_dodecahedron(double x, double y, double z)
{
    const double A = 12.70820393249937, B = 11.210068307552588, C = 14.674169922690343;
    const double D = 5.605034153776295, D2 = 5.605034153776294;
    const double Gg = 3.23606797749979, H = 1.2360679774997896;
    const double P = 4.23606797749979, Q = 5.23606797749979;
    const double S = 18.1382715378281, T = 3.464101615137755;
    const double U = 9.06913576891405, W = 15.70820393249937, Y = 9.70820393249937;
    const double Z2 = 6.47213595499958, K = 25.41640786499874;
    const double R3 = 1.7320508075688772, E = 8.47213595499958;
    return (A + B * x >= C * z) && (B * x <= A + C * z) &&
           (D * (Gg * x - H * z) <= 6. * (P + Q * y)) && (S * x + T * z <= A) &&
           (U * x + W * y <= A + T * z) && (Y * y <= A + D2 * x + C * z) &&
           (A + D2 * x + Y * y + C * z >= 0.) && (W * y + T * z <= A + U * x) &&
           (D * (-Z2 * x - H * z) <= K) && (T * z <= U * x + 3. * (P + Q * y)) &&
           (R3 * (Gg * x + E * z) <= 3. * (P + Gg * y)) && (D2 * x + Y * y + C * z <= A);
}

// potential.rs:188-314 at PADDED global index (ix,iy,iz)
__device__ __forceinline__ double wafer_potential_at(const WaferPotArgs &a, int ix, int iy, int iz)
{
    const int nx = a.g.nx, ny = a.g.ny, nz = a.g.nz;
    switch (a.type) {
    case 1: // Cube :192-201
        return ((ix > nx / 4 && ix <= 3 * nx / 4) && (iy > ny / 4 && iy <= 3 * ny / 4) &&
                (iz > nz / 4 && iz <= 3 * nz / 4)) ? -10.0 : 0.0;
    case 2: // QuadWell :202-211
        return ((ix > nx / 4 && ix <= 3 * nx / 4) && (iy > ny / 4 && iy <= 3 * ny / 4) &&
                (iz > 3 * nz / 8 && iz <= 5 * nz / 8)) ? -10.0 : 0.0;
    case 3: { // Periodic :212-220
        const double sx = sin(2. * WAFER_PI * ((double)ix - 1.) / ((double)nx - 1.));
        const double sy = sin(2. * WAFER_PI * ((double)iy - 1.) / ((double)ny - 1.));
        const double sz = sin(2. * WAFER_PI * ((double)iz - 1.) / ((double)nz - 1.));
        double temp = sx * sx;
        temp *= sy * sy;
        temp *= sz * sz;
        return -temp + 1.;
    }
    case 4:
    case 5: { // Coulomb / ComplexCoulomb :221-229
        const double r = a.dn * sqrt(wafer_r2(ix, iy, iz, nx, ny, nz));
        return (r < a.dn) ? -1. / a.dn : -1. / r;
    }
    case 6: { // ElipticalCoulomb :230-240
        const double dx = (double)ix - ((double)nx + 1.) / 2.;
        const double dy = (double)iy - ((double)ny + 1.) / 2.;
        const double dz = ((double)iz - ((double)nz + 1.) / 2.) * 2.;
        const double r = a.dn * sqrt(dx * dx + dy * dy + dz * dz);
        return (r < a.dn) ? 0.0 : -1. / r + 1. / a.dn;
    }
    case 7: { // SimpleCornell :241-249
        const double r = a.dn * sqrt(wafer_r2(ix, iy, iz, nx, ny, nz));
        if (r < a.dn) return 4. * a.mass;
        return (-0.5 * (4. / 3.)) / r + a.sig * r + 4. * a.mass;
    }
    case 8: { // FullCornell :250-269
        const double dz = (double)iz - ((double)nz + 1.) / 2.;
        const double r = a.dn * sqrt(wafer_r2(ix, iy, iz, nx, ny, nz));
        const double md = a.mu_t * (1. + a.xi_coef * (1. - a.dn * a.dn * dz * dz / (r * r))) * a.xi_fac;
        if (r < a.dn) return 4. * a.mass;
        const double screen = exp(-md * r);
        return (-a.alphas_2pit * (4. / 3.)) * screen / r + a.sig * (1. - screen) / md -
               (0.8 * a.sig) / (4. * a.mass * a.mass * r) + 4. * a.mass;
    }
    case 9:
    case 10: { // Harmonic / ComplexHarmonic :270-274
        const double r = a.dn * sqrt(wafer_r2(ix, iy, iz, nx, ny, nz));
        return r * r / 2.;
    }
    case 11: { // Dodecahedron :275-314
        const double dx = (double)ix - ((double)nx + 1.) / 2.;
        const double dy = (double)iy - ((double)ny + 1.) / 2.;
        const double dz = (double)iz - ((double)nz + 1.) / 2.;
        const double x = dx / (((double)nx - 1.) / 2.);
        const double y = dy / (((double)ny - 1.) / 2.);
        const double z = dz / (((double)nz - 1.) / 2.);
        return wafer_in_dodecahedron(x, y, z) ? -100. : 0.0;
    }
    default: // NoPotential :191
        return 0.0;
    }
}

// potential::generate (potential.rs:46-62) + ancillary arrays (potential.rs:101-110)
// over every padded cell of the slab (ghost planes included).
// grid (ceil(px/64), ceil(py/4), lz), block (64,4).
template <typename T>
__global__ __launch_bounds__(256) void wafer_k_potential(WaferPotArgs a, T *__restrict__ v,
                                                         T *__restrict__ pa, T *__restrict__ pb)
{
    const WaferGeom &g = a.g;
    const int xp = blockIdx.x * 64 + threadIdx.x;
    const int yp = blockIdx.y * 4 + threadIdx.y;
    const int lzp = blockIdx.z;
    const int zp = g.zp_of(lzp);
    if (xp >= g.px || yp >= g.py || zp < 0 || zp >= g.pzg) return;
    const double vv = wafer_potential_at(a, xp, yp, zp);
    const double bb = 1. / (1. + a.dt * vv / 2.);
    const double aa = (1. - a.dt * vv / 2.) * bb;
    const long long p = g.at(lzp, yp, xp);
    v[p] = (T)vv;
    if (pa) { // the a, b arrays exist only once a kernel that streams them has been asked for
        pa[p] = (T)aa;
        pb[p] = (T)bb;
    }
}

// a, b from an uploaded V (potential.rs:101-110)
template <typename T>
__global__ __launch_bounds__(256) void wafer_k_ab(WaferGeom g, double dt, const T *__restrict__ v,
                                                  T *__restrict__ pa, T *__restrict__ pb)
{
    const int xp = blockIdx.x * 64 + threadIdx.x;
    const int yp = blockIdx.y * 4 + threadIdx.y;
    const int lzp = blockIdx.z;
    const int zp = g.zp_of(lzp);
    if (xp >= g.px || yp >= g.py || zp < 0 || zp >= g.pzg) return;
    const long long p = g.at(lzp, yp, xp);
    const double vv = (double)v[p];
    const double bb = 1. / (1. + dt * vv / 2.);
    pa[p] = (T)((1. - dt * vv / 2.) * bb);
    pb[p] = (T)bb;
}

// potential_sub_idx for FullCornell (potential.rs:326-341) on the UNPADDED
// work index, stored at the work cell of the padded device layout.
template <typename T>
__global__ __launch_bounds__(256) void wafer_k_potsub_fullcornell(WaferPotArgs a, T *__restrict__ ps)
{
    const WaferGeom &g = a.g;
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int j = blockIdx.y * 4 + threadIdx.y;
    const int kl = blockIdx.z;
    if (i >= g.nx || j >= g.ny || kl >= g.nzl) return;
    const int k = g.z_begin + kl;
    const double dz = (double)k - ((double)g.nz + 1.) / 2.;
    const double r = a.dn * sqrt(wafer_r2(i, j, k, g.nx, g.ny, g.nz));
    const double md = a.mu_t * 1. + a.xi_coef * (1. - a.dn * a.dn * dz * dz / (r * r)) * a.xi_fac;
    ps[g.at(g.lzp_of_work(kl), j + g.R, i + g.R)] = (T)(a.sig / md + 4. * a.mass);
}

// ---- initial conditions (config.rs:577-683) -----------------------------------
__device__ __forceinline__ unsigned long long wafer_mix64(unsigned long long z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

struct WaferIcArgs {
    WaferGeom g;
    int ic;              // wafer_initial_condition
    unsigned long long seed;
    double dn, mass, sig;
};

// Writes EVERY padded cell of the slab: the chosen profile inside the work
// area, zero on the Dirichlet frame (config.rs:597-622).
template <typename T>
__global__ __launch_bounds__(256) void wafer_k_initial_condition(WaferIcArgs a, T *__restrict__ phi)
{
    const WaferGeom &g = a.g;
    const int xp = blockIdx.x * 64 + threadIdx.x;
    const int yp = blockIdx.y * 4 + threadIdx.y;
    const int lzp = blockIdx.z;
    const int zp = g.zp_of(lzp);
    if (xp >= g.px || yp >= g.py) return;
    double val = 0.0;
    const bool in_grid = zp >= 0 && zp < g.pzg;
    const bool frame = !in_grid || xp < g.R || xp >= g.px - g.R || yp < g.R || yp >= g.py - g.R ||
                       zp < g.R || zp >= g.pzg - g.R;
    if (!frame) {
        switch (a.ic) {
        case 1: { // Gaussian (config.rs:636-642): the engine's own counter RNG keyed by
                  // the reference-layout padded linear index (thread_rng there is unseeded)
            const unsigned long long c =
                ((unsigned long long)xp * (unsigned long long)g.py + (unsigned long long)yp) *
                    (unsigned long long)g.pzg + (unsigned long long)zp;
            const unsigned long long h1 = wafer_mix64(a.seed ^ wafer_mix64(2 * c));
            const unsigned long long h2 = wafer_mix64(a.seed ^ wafer_mix64(2 * c + 1));
            const double u1 = ((double)(h1 >> 11) + 1.0) * (1.0 / 9007199254740992.0);
            const double u2 = (double)(h2 >> 11) * (1.0 / 9007199254740992.0);
            val = a.sig * (sqrt(-2.0 * log(u1)) * cos(2.0 * WAFER_PI * u2));
            break;
        }
        case 2: { // Coulomb-like (config.rs:650-669); centre = padded size / 2
            const double dx = (double)xp - ((double)g.px / 2.);
            const double dy = (double)yp - ((double)g.py / 2.);
            const double dz = (double)zp - ((double)g.pzg / 2.);
            const double r = a.dn * sqrt(dx * dx + dy * dy + dz * dz);
            const double costheta = a.dn * dz / r;
            const double cosphi = a.dn * dx / r;
            const double mr2 = exp(-a.mass * r / 2.);
            val = exp(-a.mass * r) + (2. - a.mass * r) * mr2 + a.mass * r * mr2 * costheta +
                  a.mass * r * mr2 * sqrt(1. - costheta * costheta) * cosphi;
            break;
        }
        case 3: // Constant (config.rs:593)
            val = 0.1;
            break;
        default: // Boolean (config.rs:676-683) on padded indices
            val = (double)((xp & 1) * (yp & 1) * (zp & 1));
            break;
        }
    }
    phi[g.at(lzp, yp, xp)] = (T)val;
}

// ---- symmetry constraints (config.rs:691-728) ---------------------------------------------------
// The reference walks the SevenPoint frame in place and in ascending order, so cells above the
// mirror plane read cells the same pass has already multiplied by `sign`.  Restated per cell from
// the OLD values (out != in, no ordering between threads): along the constrained axis, padded
// coordinate s in [3, 3 + n), h = (3 + n) / 2, t = n + 4 - s,
//   s <= h or t == s : sign * old[s]
//   t >= 3           : sign * (sign * old[t])
//   t <  3           : sign * old[t]            (t is a frame cell: zero)
// The other cells the reference touches (x frame, the frame row / plane at 3 + n) hold zeros and
// are left alone.  axis 0: z (device plane index), 1: y.
template <typename T>
__global__ __launch_bounds__(256) void wafer_k_symmetrise(WaferGeom g, int axis, double sign, const T *__restrict__ in,
                                                          T *__restrict__ out)
{
    const int xp = blockIdx.x * 64 + threadIdx.x;
    const int yp = blockIdx.y * 4 + threadIdx.y;
    const int lzp = blockIdx.z;
    const int zp = g.zp_of(lzp);
    if (xp < g.R || xp >= g.px - g.R || yp < g.R || yp >= g.py - g.R || zp < g.R || zp >= g.pzg - g.R) return;
    const int n = axis == 0 ? g.nz : g.ny;
    const int s = axis == 0 ? zp : yp;
    const int h = (3 + n) / 2, t = n + 4 - s;
    double v;
    if (s <= h || t == s) {
        v = sign * (double)in[g.at(lzp, yp, xp)];
    } else {
        const double src = axis == 0 ? (double)in[g.at(lzp + (t - s), yp, xp)] : (double)in[g.at(lzp, t, xp)];
        v = t >= 3 ? sign * (sign * src) : sign * src;
    }
    out[g.at(lzp, yp, xp)] = (T)v;
}

// ---- layout transposes ----------------------------------------------------------
// `dense` is a double array in the reference's C-order [sx][sy][sz]; its element
// (x,y,z) corresponds to device cell (lzp0+z, yp0+y, xp0+x).
struct WaferXposeArgs {
    WaferGeom g;
    int sx, sy, sz;
    int xp0, yp0, lzp0;
};

// grid (ceil(sx/32), sy, ceil(sz/32)), block (32,8)
template <typename T, bool TO_DEVICE>
__global__ __launch_bounds__(256) void wafer_k_transpose(WaferXposeArgs a, double *__restrict__ dense,
                                                         T *__restrict__ dev)
{
    __shared__ double tile[32][33];
    const WaferGeom &g = a.g;
    const int x0 = blockIdx.x * 32, y = blockIdx.y, z0 = blockIdx.z * 32;
    if constexpr (TO_DEVICE) {
#pragma unroll
        for (int r = 0; r < 32; r += 8) { // rows of the tile = x, fast index = z
            const int x = x0 + threadIdx.y + r, z = z0 + threadIdx.x;
            if (x < a.sx && z < a.sz)
                tile[threadIdx.y + r][threadIdx.x] =
                    dense[((size_t)x * a.sy + y) * (size_t)a.sz + z];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 32; r += 8) { // rows = z, fast index = x
            const int z = z0 + threadIdx.y + r, x = x0 + threadIdx.x;
            if (x < a.sx && z < a.sz)
                dev[g.at(a.lzp0 + z, a.yp0 + y, a.xp0 + x)] = (T)tile[threadIdx.x][threadIdx.y + r];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 32; r += 8) {
            const int z = z0 + threadIdx.y + r, x = x0 + threadIdx.x;
            if (x < a.sx && z < a.sz)
                tile[threadIdx.x][threadIdx.y + r] =
                    (double)dev[g.at(a.lzp0 + z, a.yp0 + y, a.xp0 + x)];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 32; r += 8) {
            const int x = x0 + threadIdx.y + r, z = z0 + threadIdx.x;
            if (x < a.sx && z < a.sz)
                dense[((size_t)x * a.sy + y) * (size_t)a.sz + z] = tile[threadIdx.y + r][threadIdx.x];
        }
    }
}

// ---- trilinear resample (input.rs:667-716) -----------------------------------------------
// `src` is a dense double array [sx][sy][sz] (reference layout).  Fills the WORK
// cells of `dst` (device layout); the sample position of work cell (i,j,k) is
// linspace(0, s-1, basis)[i] per axis, exactly as the reference computes it
// (ndarray linspace: a + i*(b-a)/(n-1); bracket = first integer > position).
struct WaferResampleArgs {
    WaferGeom g;
    int sx, sy, sz;     // source dims
    int bx, by, bz;     // basis sizes (the reference passes the padded target size)
};

__device__ __forceinline__ void wafer_bracket(int n, double look, int *lo, int *hi)
{
    // (0..n).position(|q| q as f64 > look): the first integer above `look`, if below n
    const int q = (int)floor(look) + 1;
    if (q < n) {
        *lo = q - 1;
        *hi = q;
    } else {
        *lo = n - 1;
        *hi = n;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void wafer_k_trilerp(WaferResampleArgs a, const double *__restrict__ src,
                                                       T *__restrict__ dst)
{
    const WaferGeom &g = a.g;
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int j = blockIdx.y * 4 + threadIdx.y;
    const int lzp = blockIdx.z;                 // every local plane, ghost planes included
    const int k = g.z_begin + (lzp - g.G);      // global work-z
    if (i >= g.nx || j >= g.ny || k < 0 || k >= g.nz) return;
    const int nx = a.sx - 1, ny = a.sy - 1, nz = a.sz - 1;
    const double stx = a.bx > 1 ? ((double)nx - 0.) / (double)(a.bx - 1) : 0.;
    const double sty = a.by > 1 ? ((double)ny - 0.) / (double)(a.by - 1) : 0.;
    const double stz = a.bz > 1 ? ((double)nz - 0.) / (double)(a.bz - 1) : 0.;
    const double xl = 0. + stx * (double)i, yl = 0. + sty * (double)j, zl = 0. + stz * (double)k;
    int x0, x1, y0, y1, z0, z1;
    wafer_bracket(nx, xl, &x0, &x1);
    wafer_bracket(ny, yl, &y0, &y1);
    wafer_bracket(nz, zl, &z0, &z1);
    const double xd = (xl - (double)x0) / ((double)x1 - (double)x0);
    const double yd = (yl - (double)y0) / ((double)y1 - (double)y0);
    const double zd = (zl - (double)z0) / ((double)z1 - (double)z0);
    auto at = [&](int x, int y, int z) { return src[((size_t)x * a.sy + y) * (size_t)a.sz + z]; };
    auto op = [](double c0, double c1, double d) { return c0 * (1. - d) + c1 * d; };
    const double c00 = op(at(x0, y0, z0), at(x1, y0, z0), xd);
    const double c01 = op(at(x0, y0, z1), at(x1, y0, z1), xd);
    const double c10 = op(at(x0, y1, z0), at(x1, y1, z0), xd);
    const double c11 = op(at(x0, y1, z1), at(x1, y1, z1), xd);
    const double c0 = op(c00, c10, yd);
    const double c1 = op(c01, c11, yd);
    dst[g.at(lzp, j + g.R, i + g.R)] = (T)op(c0, c1, zd);
}

// min / max of |1 + dt*V/2| over a whole allocation (guard cells hold V = 0 -> 1), as the bit
// patterns of the non-negative doubles (their unsigned order is their numeric order; NaN sorts
// above +inf).  out[0] = min (start at ~0ull), out[1] = max (start at 0).
template <typename T>
__global__ __launch_bounds__(256) void wafer_k_v_range(const T *__restrict__ v, long long n, double dt,
                                                       unsigned long long *__restrict__ out)
{
    unsigned long long lo = ~0ull, hi = 0ull;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const double x = fabs(1. + dt * (double)v[i] / 2.);
        const unsigned long long b = (unsigned long long)__double_as_longlong(x);
        lo = b < lo ? b : lo;
        hi = b > hi ? b : hi;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long l2 = __shfl_down(lo, off, 64), h2 = __shfl_down(hi, off, 64);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&out[0], lo);
        atomicMax(&out[1], hi);
    }
}
