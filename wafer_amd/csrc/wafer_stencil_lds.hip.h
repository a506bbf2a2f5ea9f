// LDS-tiled, z-streaming stencil step for gfx950 (the hot kernel).
//
// One workgroup = 4 wavefronts = one (TX x TY) tile of the x-y plane, marched
// along z over `zchunk` planes:
//   - each lane owns VEC = 16 B / sizeof(T) consecutive x cells on RY rows and
//     keeps their z-column (2R+1 planes + one prefetched plane) in registers,
//     so every phi cell is fetched from HBM once, as a 16-byte-per-lane,
//     128-byte-aligned wave load (rows start on a 128 B boundary: wafer_geom.h);
//   - the centre plane of the tile (+ R halo rows / columns) sits in a
//     double-buffered LDS tile; x and y neighbours come from LDS (or from the
//     lane's own registers when they are in its RY x VEC patch);
//   - a, b are streamed straight into registers one plane ahead;
//   - one s_barrier per plane.
// HBM traffic per update: 4*sizeof(T) (phi, a, b in; phi' out) + halo re-reads
// that hit L2 / Infinity Cache.  No MFMA: 12-42 flop per 32 B is bandwidth bound.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "wafer_geom.h"
#include "wafer_stencil.hip.h"
#include "wafer_setup.hip.h"

template <typename T> struct WaferVec;
template <> struct WaferVec<double> { static constexpr int N = 2; typedef double __attribute__((ext_vector_type(2))) type; };
template <> struct WaferVec<float> { static constexpr int N = 4; typedef float __attribute__((ext_vector_type(4))) type; };

template <typename T, int R, int RY_, int NW_ = 4>
struct WaferLdsCfg {
    static constexpr int VEC = WaferVec<T>::N;
    static constexpr int NW = NW_;                // wavefronts per workgroup, stacked along y (4, or 8: 16-row tiles)
    static constexpr int NT = NW * 64;            // threads per workgroup
    static constexpr int RY = RY_;                // rows per lane
    static constexpr int TX = 64 * VEC;           // tile width  (one wave-wide 1 KiB row segment)
    static constexpr int TY = NW * RY;            // tile height
    static constexpr int HX = ((R + VEC - 1) / VEC) * VEC; // halo columns kept per side (VEC aligned)
    static constexpr int LP = TX + 2 * HX;        // LDS row pitch (elements)
    static constexpr int LROWS = TY + 2 * R;
    static constexpr int TILE = LROWS * LP;       // elements per LDS buffer
    static constexpr int NHALO_X = 2 * R * TY;    // halo-column cells per plane
    static constexpr int HALO_X_ITERS = (NHALO_X + NT - 1) / NT; // cells are dealt over all lanes of the workgroup
    static constexpr int HALO_ROWS_PER_WAVE = (2 * R + NW - 1) / NW;
};

// launch-time tuning knobs (tools/stencil_sweep.py drives them through the environment)
struct WaferLdsOpts {
    int ry;   // rows per lane: 2 or 4 -> tile height 8, 16
    int swz;  // XCD-aware workgroup -> tile mapping
    int nt;   // non-temporal a/b loads and phi' stores
    int pad;  // extra dynamic LDS bytes per workgroup (caps workgroups per CU; tuning only)
    int abv;  // form a, b from V in registers instead of streaming them
};
// R: stencil reach of the kernel the options are for (the SevenPoint single-step kernel defaults to
// 4 rows per lane: 6 halo rows per 16 instead of per 8 -- 0.60 vs 0.63 ms/step at 512^3)
static inline WaferLdsOpts wafer_lds_opts(const WaferTuning &t, int R = 1)
{
    WaferLdsOpts o{R == 3 ? 4 : 2, t.swz, t.nt >= 0 ? t.nt : 1, 0, t.abv};
    if (t.lds_ry) o.ry = t.lds_ry;
    // abv < 0: kernel default (both the single-step and the fused kernel form a, b from V)
    return o;
}

// Planes per workgroup.  Measured on MI355X (profiles/r01_sweep_*.jsonl): the step
// kernel streams fastest with ONE workgroup per CU marching a long z-column
// (256 workgroups at 512^3: 0.73 ms/step) and loses 5-12 % when the same work
// is cut into 2-8x more, shorter columns -- more concurrent streams than CUs
// only add DRAM/L2 contention.  So z is chunked only as far as needed to give
// every CU a workgroup.
template <typename T, int R>
static inline int wafer_lds_zchunk(const WaferTuning &t, const WaferGeom &g, int nplanes, int ry, int target_blocks)
{
    using Cfg = WaferLdsCfg<T, R, 1>;
    const int TY = Cfg::NW * ry;
    if (t.zchunk > 0) return t.zchunk;
    if (target_blocks < 0) return -target_blocks < nplanes ? -target_blocks : nplanes; // the caller fixed the chunk length
    const long long per_layer = (long long)((g.nx + Cfg::TX - 1) / Cfg::TX) * ((g.ny + TY - 1) / TY);
    // two workgroups per CU: with a, b formed from V the kernel does more arithmetic per byte and
    // a second resident workgroup hides it (0.539 vs 0.574 ms at 512^3, profiles/r01_sweep_e_512.jsonl)
    const long long target = t.target_blocks > 0 ? t.target_blocks : 2 * (target_blocks > 0 ? target_blocks : 256);
    return wafer_pick_zchunk(per_layer, nplanes, target, R + 2);
}

template <typename T, int R>
static inline long long wafer_step_lds_blocks(const WaferTuning &t, const WaferGeom &g, int lz_lo, int lz_hi, int target_blocks)
{
    using Cfg = WaferLdsCfg<T, R, 1>;
    const int ry = wafer_lds_opts(t, R).ry;
    const int TY = Cfg::NW * ry;
    const int zc = wafer_lds_zchunk<T, R>(t, g, lz_hi - lz_lo, ry, target_blocks);
    return (long long)((g.nx + Cfg::TX - 1) / Cfg::TX) * ((g.ny + TY - 1) / TY) *
           ((lz_hi - lz_lo + zc - 1) / zc);
}

// streamed-once data (a, b, phi') can bypass the caches' retention
template <bool NT, typename VT>
__device__ __forceinline__ VT wafer_ld_stream(const VT *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT, typename VT>
__device__ __forceinline__ void wafer_st_stream(VT *p, VT v)
{
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// ---- closed-form potentials evaluated in the step kernel (template parameter VG) -----------------
// sqrt of r^2 = dx^2 + dy^2 + dz^2 (potential.rs:366-371): the argument is zero or lies in
// [0.75, 3 (n+1)^2 / 4], so the scaling branch of hipcc's correctly rounded fp64 square root
// (arguments below 2^-767) never applies; what is left of its expansion is spelled here -- v_rsq_f64,
// the coupled Goldschmidt step and two Newton corrections of the root -- and gives the same bits
// (tests: the kernel's V against the stored array, cell by cell, through identical phi).
__device__ __forceinline__ double wafer_sqrt_r2(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return x == 0.0 ? x : g;
}

// potential.rs:221-229 / 241-249 / 270-274 at PADDED global index (ix, iy, iz): the expressions of
// wafer_potential_at (wafer_setup.hip.h) with the square root above and, for Coulomb, -1/r as the
// negated short reciprocal (IEEE division is sign-symmetric; wafer_recip's short form holds for
// 2^-400 < r < 2^400, which the engine checks on dn and the grid size before it selects VG).
template <int VG>
__device__ __forceinline__ double wafer_vgen_at(const WaferPotArgs &a, int ix, int iy, int iz)
{
    const double r = a.dn * wafer_sqrt_r2(wafer_r2(ix, iy, iz, a.g.nx, a.g.ny, a.g.nz));
    if constexpr (VG == 4) { // Coulomb / ComplexCoulomb
        return -wafer_recip((r < a.dn) ? a.dn : r, true);
    } else if constexpr (VG == 7) { // SimpleCornell
        const double far = (-0.5 * (4. / 3.)) / r + a.sig * r + 4. * a.mass;
        return (r < a.dn) ? 4. * a.mass : far;
    } else { // Harmonic / ComplexHarmonic
        static_assert(VG == 9, "closed forms: Coulomb (4), SimpleCornell (7), Harmonic (9)");
        return r * r / 2.;
    }
}

// ABV: `pa` is the potential V and a, b are formed in registers exactly as
// potential.rs:104-110 does (same expressions => the same bits as the stored
// arrays), which removes one of the four HBM streams: 24 B instead of 32 B of
// traffic per update, at the price of one more fp64 division.
// NLOW: -1 = plain step; >= 0 = the excited-state step: also accumulates
// sum(phi'^2) (grid.rs:675-678) and the raw overlaps t_j = sum(l_j * phi') with
// NLOW stored states in the same pass (partials[q * pstride + workgroup]),
// from which wafer_k_gs_apply forms the modified Gram-Schmidt coefficients.
// XF (excited states only): `phi` holds the RAW result of the previous step (un-normalised,
// un-projected) and every cell is normalised and Gram-Schmidt-projected as it is loaded,
//   x = phi/norm - sum_j l_j s_j        (grid.rs:467, 488-490; the operations of wafer_k_gs_apply)
// with norm and s_j formed from the previous step's scalars xscal[0..NLOW] and the Gram matrix.
// That folds the apply pass into the next step: (3+k)*8 B per update instead of (5+2k)*8 B.
// NLOW = -2: compute_observables (grid.rs:303-445) on the same pipeline -- `pa` is V, nothing is
// written, and the four work-area sums (energy integrand V w^2 - w S / den, w^2, w^2 pot_sub,
// w^2 r^2 with the WORK-area index, grid.rs:429-435) go to partials[q * pstride + workgroup];
// an array pot_sub rides in low.p[0].  16 B per lane from HBM (round 1's scalar-load kernel is gone).
// VG != 0 (fp64, with ABV): V is not streamed at all.  The potential is one of the closed forms of
// potential.rs:188-274 (VG = its wafer_potential number: Coulomb, SimpleCornell, Harmonic) and every
// lane evaluates wafer_potential_at -- the function that filled the stored array, so the same bits --
// for the cells it updates: 8 B per update less through the L2 <-> fabric path, which is what bounds
// the excited-state kernels ((3+k)*8 -> (2+k)*8 B), paid for with a square root and a division per
// cell out of the VALU time those kernels spend waiting.
// VIRT: a.v_in_range known at compile time (1 / 0) instead of tested per cell (-1): no scalar branch
// inside the update, so the RY x VEC cells of a lane share one basic block and their chains interleave.
// DEEP (with XF): every prefetched value -- the lane's own cells of phi and of the stored states for plane
// z+R+2, the halo rows / columns of plane z+3 -- is requested RAW into a staging set at the END of iteration z,
// behind that iteration's stores and right after the set's previous contents (requested one iteration earlier)
// were transformed.  No register holding a load in flight is copied (a rotation would force the wait into the
// iteration that issued the load), so the loads have a whole iteration, barrier included, to land.  Loads and
// stores complete in order on gfx9 and share one counter: with the requests behind the stores, the only
// operations younger than a staging set when it is consumed are the next iteration's stores -- and DEEP kernels
// store every row of the tile as one full vector, cells outside the work area as the zero they hold anyway
// (frame, pad and guard cells are zeros in every array: wafer_geom.h), so that their number is known to the
// compiler on every path (the first iteration included) and the wait is exact.
template <typename T, typename C, int R, int RY, int NLOW, bool NT, bool ABV, bool XF = false, int NW = 4, int VG = 0, int VIRT = -1, bool DEEP = false>
__global__ __launch_bounds__(NW * 64) void wafer_k_step_lds(WaferStepArgs a, int ntx, int nty, int swz,
                                                        const T *__restrict__ phi,
                                                        const T *__restrict__ pa,
                                                        const T *__restrict__ pb, T *__restrict__ out,
                                                        double *__restrict__ partials, long long pstride,
                                                        WaferLowPtrs low,
                                                        const double *__restrict__ xscal = nullptr,
                                                        const double *__restrict__ xgram = nullptr)
{
    constexpr bool NORM = NLOW >= 0;
    constexpr bool OBS = NLOW == -2;
    // (full zero-masked stores in EVERY kernel of this family were measured -- same box, alternating libraries:
    //  ThreePoint / FivePoint / fp32-storage single-step -1 %, SevenPoint +1.5 % -- and not adopted)
    constexpr int NL = NLOW > 0 ? NLOW : 0;
    static_assert(!XF || NL > 0, "transform-on-load needs stored states");
    static_assert(!OBS || (ABV && std::is_same<C, double>::value), "observables: V in pa's slot, fp64 sums");
    static_assert(VG == 0 || (ABV && std::is_same<T, double>::value), "closed-form V: fp64 storage, a and b formed in registers");
    static_assert(!DEEP || XF, "the two-plane prefetch is built for the transform-on-load kernels");
    using Cfg = WaferLdsCfg<T, R, RY, NW>;
    using VT = typename WaferVec<T>::type;
    constexpr int VEC = Cfg::VEC, TX = Cfg::TX, TY = Cfg::TY, HX = Cfg::HX, LP = Cfg::LP;
    __shared__ __attribute__((aligned(16))) T lds[2 * Cfg::TILE];
    __shared__ double red[NW];

    const WaferGeom &g = a.g;
    // tile coordinates: x fastest, then y, then z-chunk
    // Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2).
    // With swz the tiles an XCD works on form one contiguous range of the
    // (x, y, z-chunk) order, so neighbouring tiles -- which re-read each other's
    // halo rows -- share an L2.  Bijective for any grid size; speed only.
    int bid = blockIdx.x;
    if (swz) {
        const int n = gridDim.x, q = n >> 3, r = n & 7, k = bid & 7;
        bid = k * q + min(k, r) + (bid >> 3);
    }
    const int tx_i = bid % ntx;
    const int ty_i = (bid / ntx) % nty;
    const int tz_i = bid / (ntx * nty);
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;

    const int x0 = tx_i * TX;            // work-x of the tile's first column
    const int y0 = ty_i * TY;            // work-y of the tile's first row
    const int xl = lane * VEC;           // lane's first column inside the tile
    const int yl = wave * RY;            // lane's first row inside the tile
    const int xi = x0 + xl;              // work-x of the lane's first cell
    const int zs = a.lz_lo + tz_i * a.zchunk;
    const int ze = min(zs + a.zchunk, a.lz_hi);

    // Loads carry no bounds predicates: whole tiles, R halo rows / columns and the planes just
    // outside the slab lie in the allocation's zero guard zone (wafer_geom.h).
    bool rowin[RY];       // the row is a work row (wave-uniform)
    long long rowoff[RY]; // element offset of (row r, lane's first cell) inside a plane
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        const int y = y0 + yl + r;
        rowin[r] = y < g.ny;
        rowoff[r] = (long long)(y + R) * g.pitch + g.xoff + R + xi;
    }

    // halo rows this wave fetches: halo row h in [0,2R): h<R is row y0-R+h, else row y0+TY+(h-R).
    // A wave without a halo row (h >= 2R) requests its own first row a second time instead: the request is a cache hit
    // behind the wave's own request of that row, its value is never written to the tile, and every wave runs the same
    // instruction stream (the staging pipeline's waits count requests).  Before round 3 these waves requested rows
    // y0+TY+R.. of the tile BELOW, six whole rows per tile and plane with 8 waves: hits while that tile's workgroup was
    // resident on the same XCD, HBM reads otherwise -- 4.7 % of the reads at 512^3 (profiles/r03_halo_attribution.json).
    long long hrow_off[Cfg::HALO_ROWS_PER_WAVE];
    int hrow_lds[Cfg::HALO_ROWS_PER_WAVE];
#pragma unroll
    for (int q = 0; q < Cfg::HALO_ROWS_PER_WAVE; ++q) {
        const int h = wave + q * Cfg::NW;
        const int ly = (h < R) ? h : TY + h;          // LDS row (0..R-1 above, TY+R.. below)
        const int yp = y0 + ly;                        // padded y (= work y - R + R)
        hrow_off[q] = h < 2 * R ? (long long)yp * g.pitch + g.xoff + R + xi : rowoff[0];
        hrow_lds[q] = ly * LP + HX + xl;
    }
    // halo-column cells: cell c in [0, 2R*TY): row = c / (2R), k = c % (2R);
    // k<R: column x0-1-k, else column x0+TX+(k-R).
    // A cell outside the work area (the Dirichlet frame and the pad cells behind it, zeros that no kernel writes) is not
    // fetched: its 128-byte line holds nothing else anyone reads, so every such request was an HBM read of its own -- 32 lines
    // per plane and row of tiles, 6.2 % of the reads at 512^3.  The lane requests the tile's own edge cell of that row instead
    // (a hit) and the value is replaced by the zero it stands for.
    long long hcol_off[Cfg::HALO_X_ITERS];
    int hcol_lds[Cfg::HALO_X_ITERS];
    bool hcol_frame[Cfg::HALO_X_ITERS];
#pragma unroll
    for (int q = 0; q < Cfg::HALO_X_ITERS; ++q) {
        const int cidx = min(tid + q * Cfg::NT, Cfg::NHALO_X - 1);     // surplus lanes repeat the last cell
        const int row = cidx / (2 * R), k = cidx % (2 * R);
        const int xw = (k < R) ? (x0 - 1 - k) : (x0 + TX + (k - R)); // work x, may be -R..nx+R-1
        const int y = y0 + row;
        hcol_frame[q] = xw < 0 || xw >= g.nx;
        hcol_off[q] = (long long)(y + R) * g.pitch + g.xoff + R + (hcol_frame[q] ? ((k < R) ? x0 : x0 + TX - 1) : xw);
        hcol_lds[q] = (row + R) * LP + ((k < R) ? (HX - 1 - k) : (HX + TX + (k - R)));
    }

    const C dt = (C)a.dt;
    const WaferDen<C> den = wafer_den<C>(a, VIRT < 0 ? a.v_in_range != 0 : VIRT != 0);
    [[maybe_unused]] WaferPotArgs vgen;   // only the fields wafer_potential_at reads for these types
    if constexpr (VG != 0) {
        vgen.g = g;
        vgen.type = VG;
        vgen.dn = a.vg_dn; vgen.dt = a.dt; vgen.mass = a.vg_mass; vgen.sig = a.vg_sig;
        vgen.mu_t = vgen.alphas_2pit = vgen.xi_coef = vgen.xi_fac = 0.0;
    }
    VT zero;
#pragma unroll
    for (int v = 0; v < VEC; ++v) zero[v] = T(0);

    // ---- transform-on-load coefficients (XF) -------------------------------------------------
    C xnorm = C(1);
    C xsj[NL > 0 ? NL : 1];
    if constexpr (XF) {
        xnorm = (C)sqrt(xscal[0]);
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            double sj = xscal[1 + j] / (double)xnorm;
#pragma unroll
            for (int i = 0; i < j; ++i) sj -= (double)xsj[i] * xgram[j * WAFER_MAX_LOW + i];
            xsj[j] = (C)sj;
        }
    }
    // loads one VEC group / one cell of phi at element offset `off`, transformed if XF;
    // lkeep (may be null) receives the stored states' values at the same cells
    // x = phi/norm - sum_j l_j s_j on one VEC group (grid.rs:467, 488-490)
    auto xform_vec = [&](VT w, const VT *l) -> VT {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            C x = wafer_div_invariant<C>((C)w[v], xnorm);
#pragma unroll
            for (int j = 0; j < NL; ++j) x = x - (C)l[j][v] * xsj[j];
            w[v] = (T)x;
        }
        return w;
    };
    auto xform_cell = [&](T w, const T *l) -> T {
        C x = wafer_div_invariant<C>((C)w, xnorm);
#pragma unroll
        for (int j = 0; j < NL; ++j) x = x - (C)l[j] * xsj[j];
        return (T)x;
    };
    auto load_vec = [&](long long off, VT *lkeep) -> VT {
        VT w = *reinterpret_cast<const VT *>(phi + off);
        if constexpr (XF) {
            VT l[NL];
#pragma unroll
            for (int j = 0; j < NL; ++j) l[j] = *reinterpret_cast<const VT *>(static_cast<const T *>(low.p[j]) + off);
            w = xform_vec(w, l);
            if (lkeep) {
#pragma unroll
                for (int j = 0; j < NL; ++j) lkeep[j] = l[j];
            }
        }
        return w;
    };
    auto load_cell = [&](long long off) -> T {
        T w = phi[off];
        if constexpr (XF) {
            C x = wafer_div_invariant<C>((C)w, xnorm);
#pragma unroll
            for (int j = 0; j < NL; ++j) x = x - (C)static_cast<const T *>(low.p[j])[off] * xsj[j];
            w = (T)x;
        }
        return w;
    };

    // ---- prologue: z-queue for plane zs, LDS tile of plane zs, prefetches
    VT q[2 * R + 1][RY];
    // XF: stored states at the lane's own cells, planes z .. z+R (+ prefetch), for the overlaps
    VT lq[XF ? R + 2 : 1][RY][NL > 0 ? NL : 1];
#pragma unroll
    for (int m = 0; m <= 2 * R; ++m)
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            VT keep[NL > 0 ? NL : 1];
            q[m][r] = load_vec((long long)(zs - R + m) * g.plane + rowoff[r], (XF && m >= R) ? keep : nullptr);
            if constexpr (XF) {
                if (m >= R) {
#pragma unroll
                    for (int j = 0; j < NL; ++j) lq[m - R][r][j] = keep[j];
                }
            }
        }
    VT ab_a[RY], ab_b[RY];
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        if constexpr (VG == 0) ab_a[r] = wafer_ld_stream<NT>(reinterpret_cast<const VT *>(pa + (long long)zs * g.plane + rowoff[r]));
        if constexpr (!ABV)
            ab_b[r] = wafer_ld_stream<NT>(reinterpret_cast<const VT *>(pb + (long long)zs * g.plane + rowoff[r]));
    }
    {
        T *tile = lds + (zs & 1) * Cfg::TILE;
#pragma unroll
        for (int r = 0; r < RY; ++r)
            *reinterpret_cast<VT *>(tile + (yl + r + R) * LP + HX + xl) = q[R][r];
#pragma unroll
        for (int qq = 0; qq < Cfg::HALO_ROWS_PER_WAVE; ++qq)
            if (wave + qq * Cfg::NW < 2 * R)
                *reinterpret_cast<VT *>(tile + hrow_lds[qq]) = load_vec((long long)zs * g.plane + hrow_off[qq], nullptr);
#pragma unroll
        for (int qq = 0; qq < Cfg::HALO_X_ITERS; ++qq)
            if (tid + qq * Cfg::NT < Cfg::NHALO_X)
                tile[hcol_lds[qq]] = hcol_frame[qq] ? (T)0 : load_cell((long long)zs * g.plane + hcol_off[qq]);
    }
    // halo of plane zs+1, held in registers until it is written at iteration zs
    VT hrow_nxt[Cfg::HALO_ROWS_PER_WAVE];
    T hcol_nxt[Cfg::HALO_X_ITERS];
    {
#pragma unroll
        for (int qq = 0; qq < Cfg::HALO_ROWS_PER_WAVE; ++qq)
            hrow_nxt[qq] = load_vec((long long)(zs + 1) * g.plane + hrow_off[qq], nullptr);
#pragma unroll
        for (int qq = 0; qq < Cfg::HALO_X_ITERS; ++qq)
            hcol_nxt[qq] = hcol_frame[qq] ? (T)0 : load_cell((long long)(zs + 1) * g.plane + hcol_off[qq]);
    }
    // DEEP: the raw staging set
    [[maybe_unused]] VT raw_w[RY];
    [[maybe_unused]] VT raw_l[NL > 0 ? NL : 1][RY];
    [[maybe_unused]] VT raw_hw[Cfg::HALO_ROWS_PER_WAVE];
    [[maybe_unused]] VT raw_hl[NL > 0 ? NL : 1][Cfg::HALO_ROWS_PER_WAVE];
    [[maybe_unused]] T raw_cw[Cfg::HALO_X_ITERS];
    [[maybe_unused]] T raw_cl[NL > 0 ? NL : 1][Cfg::HALO_X_ITERS];
    // requests a staging set: the halo rows / columns of plane zh and the lane's own cells of plane zw
    auto issue_raw = [&](int zh, int zw) {
        const long long zno = (long long)zh * g.plane;
#pragma unroll
        for (int qq = 0; qq < Cfg::HALO_ROWS_PER_WAVE; ++qq) {
            raw_hw[qq] = *reinterpret_cast<const VT *>(phi + zno + hrow_off[qq]);
#pragma unroll
            for (int j = 0; j < NL; ++j) raw_hl[j][qq] = *reinterpret_cast<const VT *>(static_cast<const T *>(low.p[j]) + zno + hrow_off[qq]);
        }
#pragma unroll
        for (int qq = 0; qq < Cfg::HALO_X_ITERS; ++qq) {
            raw_cw[qq] = phi[zno + hcol_off[qq]];
#pragma unroll
            for (int j = 0; j < NL; ++j) raw_cl[j][qq] = static_cast<const T *>(low.p[j])[zno + hcol_off[qq]];
        }
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const long long off = (long long)zw * g.plane + rowoff[r];
            // (requesting the rows that no neighbouring tile reads as halo rows non-temporally, so that the edge rows
            //  outlive them in the XCD's L2, changed nothing: 0.633 / 0.837 / 1.060 ms either way)
            raw_w[r] = *reinterpret_cast<const VT *>(phi + off);
#pragma unroll
            for (int j = 0; j < NL; ++j) raw_l[j][r] = *reinterpret_cast<const VT *>(static_cast<const T *>(low.p[j]) + off);
        }
    };
    if constexpr (DEEP) issue_raw(zs + 2, zs + R + 1); // transformed at the end of iteration zs
    __syncthreads();

    double acc = 0.0;
    double acc_t[NL > 0 ? NL : 1];
#pragma unroll
    for (int j = 0; j < NL; ++j) acc_t[j] = 0.0;
    double ob_e = 0.0, ob_n = 0.0, ob_v = 0.0, ob_r = 0.0; // OBS: the four sums of grid.rs:405-437
    for (int z = zs; z < ze; ++z) {
        const bool more = z + 1 < ze;   // wave-uniform
        const long long zo = (long long)z * g.plane;
        // ---- 1. prefetch: phi plane z+R+1, a/b plane z+1, halo of plane z+2
        VT pre[RY], pre_a[RY], pre_b[RY];
        VT hrow_pre[Cfg::HALO_ROWS_PER_WAVE];
        T hcol_pre[Cfg::HALO_X_ITERS];
        if constexpr (DEEP) {
            if constexpr (VG == 0) { // V of plane z+1 (copied at the end of this iteration)
#pragma unroll
                for (int r = 0; r < RY; ++r) pre_a[r] = wafer_ld_stream<NT>(reinterpret_cast<const VT *>(pa + zo + g.plane + rowoff[r]));
            }
        } else {
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            {
                VT keep[NL > 0 ? NL : 1];
                pre[r] = load_vec(zo + (long long)(R + 1) * g.plane + rowoff[r], XF ? keep : nullptr);
                if constexpr (XF) {
#pragma unroll
                    for (int j = 0; j < NL; ++j) lq[R + 1][r][j] = keep[j];
                }
            }
            if constexpr (VG == 0) pre_a[r] = wafer_ld_stream<NT>(reinterpret_cast<const VT *>(pa + zo + g.plane + rowoff[r]));
            if constexpr (!ABV) pre_b[r] = wafer_ld_stream<NT>(reinterpret_cast<const VT *>(pb + zo + g.plane + rowoff[r]));
        }
#pragma unroll
        for (int qq = 0; qq < Cfg::HALO_ROWS_PER_WAVE; ++qq)
            hrow_pre[qq] = load_vec(zo + 2 * g.plane + hrow_off[qq], nullptr);
#pragma unroll
        for (int qq = 0; qq < Cfg::HALO_X_ITERS; ++qq)
            hcol_pre[qq] = hcol_frame[qq] ? (T)0 : load_cell(zo + 2 * g.plane + hcol_off[qq]);
        }

        // stored states at this plane (only the cells this lane updates)
        VT lw[NL > 0 ? NL : 1][RY];
#pragma unroll
        for (int j = 0; j < NL; ++j)
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                if constexpr (XF) lw[j][r] = lq[0][r][j];   // already loaded when the plane entered the pipeline
                else lw[j][r] = wafer_ld_stream<NT>(reinterpret_cast<const VT *>(static_cast<const T *>(low.p[j]) + zo + rowoff[r]));
            }
        // ---- 2. stage plane z+1 into the other LDS buffer
        if (more) {
            T *nt = lds + ((z + 1) & 1) * Cfg::TILE;
#pragma unroll
            for (int r = 0; r < RY; ++r)
                *reinterpret_cast<VT *>(nt + (yl + r + R) * LP + HX + xl) = q[R + 1][r];
#pragma unroll
            for (int qq = 0; qq < Cfg::HALO_ROWS_PER_WAVE; ++qq)
                if (wave + qq * Cfg::NW < 2 * R) *reinterpret_cast<VT *>(nt + hrow_lds[qq]) = hrow_nxt[qq];
#pragma unroll
            for (int qq = 0; qq < Cfg::HALO_X_ITERS; ++qq)
                if (tid + qq * Cfg::NT < Cfg::NHALO_X) nt[hcol_lds[qq]] = hcol_nxt[qq];
        }

        // ---- 3. update plane z
        const T *ct = lds + (z & 1) * Cfg::TILE;
        [[maybe_unused]] const double ob_dz = (double)(g.z_begin + (z - g.G)) - ((double)g.nz + 1.) / 2.;
        VT resq[RY];
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            VT res;
            [[maybe_unused]] VT psub = zero;
            if constexpr (OBS) {
                if (a.potsub_kind == 2) psub = wafer_ld_stream<NT>(reinterpret_cast<const VT *>(static_cast<const T *>(low.p[0]) + zo + rowoff[r]));
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                C xs[2 * R + 1], ys[2 * R + 1], zz[2 * R + 1];
                const C w = (C)q[R][r][v];
#pragma unroll
                for (int d = -R; d <= R; ++d) {
                    zz[d + R] = (C)q[R + d][r][v];
                    if (d == 0) {
                        xs[R] = w;
                        ys[R] = w;
                    } else {
                        xs[d + R] = (v + d >= 0 && v + d < VEC) ? (C)q[R][r][(v + d + VEC) % VEC]
                                                               : (C)ct[(yl + r + R) * LP + HX + xl + v + d];
                        ys[d + R] = (r + d >= 0 && r + d < RY) ? (C)q[R][(r + d + RY) % RY][v]
                                                             : (C)ct[(yl + r + R + d) * LP + HX + xl + v];
                    }
                }
                const C S = wafer_stencil_sum<C, R>(xs, ys, zz, w);
                if constexpr (OBS) {
                    if (rowin[r] && xi + v < g.nx) {
                        double vv;
                        if constexpr (VG != 0) vv = wafer_vgen_at<VG>(vgen, xi + v + R, y0 + yl + r + R, g.zp_of(z));
                        else vv = (double)ab_a[r][v];
                        ob_e += vv * w * w - wafer_div_invariant<double>(w * S, wafer_den<double>(a, a.v_in_range != 0)); // grid.rs:325-332 (the bits of the IEEE quotient)
                        ob_n += w * w;                      // grid.rs:407
                        if (a.potsub_kind == 2) ob_v += w * w * (double)psub[v];      // grid.rs:410-418
                        else if (a.potsub_kind == 1) ob_v += w * w * a.potsub_scalar; // grid.rs:419-424
                        // potential::calculate_r2 on the WORK-AREA index (grid.rs:429-435, potential.rs:366-371)
                        const double dx = (double)(xi + v) - ((double)g.nx + 1.) / 2.;
                        const double dy = (double)(y0 + yl + r) - ((double)g.ny + 1.) / 2.;
                        ob_r += w * w * (dx * dx + dy * dy + ob_dz * ob_dz); // grid.rs:428-437
                    }
                    res[v] = T(0);
                    continue;
                }
                C ca, cb;
                if constexpr (ABV) { // potential.rs:104-110
                    C vv;
                    if constexpr (VG != 0) vv = (C)wafer_vgen_at<VG>(vgen, xi + v + R, y0 + yl + r + R, g.zp_of(z)); // potential.rs:46-62
                    else vv = (C)ab_a[r][v];
                    cb = wafer_recip(C(1) + dt * vv / C(2), VIRT < 0 ? a.v_in_range != 0 : VIRT != 0);
                    ca = (C(1) - dt * vv / C(2)) * cb;
                } else {
                    ca = (C)ab_a[r][v];
                    cb = (C)ab_b[r][v];
                }
                res[v] = (T)wafer_update<C>(w, ca, cb, dt, S, den);
            }
            resq[r] = res;
        }
        // sums: cells outside the work area contribute an exact zero through a select, not a branch (the
        // update of all RY x VEC cells above stays one basic block); per accumulator the order is unchanged
        if constexpr (NORM) {
#pragma unroll
            for (int r = 0; r < RY; ++r)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const C m = (rowin[r] && xi + v < g.nx) ? (C)resq[r][v] : C(0);
                    acc += (double)m * (double)m;
#pragma unroll
                    for (int j = 0; j < NL; ++j) acc_t[j] += (double)((C)lw[j][r][v] * m);
                }
        }
        if constexpr (DEEP) {
            // one full vector per row, always: cells outside the work area get the zero they hold (see DEEP above)
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                VT o = resq[r];
#pragma unroll
                for (int v = 0; v < VEC; ++v) o[v] = (rowin[r] && xi + v < g.nx) ? o[v] : T(0);
                wafer_st_stream<NT>(reinterpret_cast<VT *>(out + zo + rowoff[r]), o);
            }
        } else if constexpr (!OBS) {
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                if (!rowin[r]) continue;
                T *dst = out + zo + rowoff[r];
                if (xi + VEC <= g.nx) {
                    wafer_st_stream<NT>(reinterpret_cast<VT *>(dst), resq[r]);
                } else {
#pragma unroll
                    for (int v = 0; v < VEC; ++v)
                        if (xi + v < g.nx) dst[v] = resq[r][v];
                }
            }
        }
        if constexpr (DEEP) {
            // the staging set requested one iteration ago: own cells of plane z+R+1, halo of plane z+2
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                VT l[NL > 0 ? NL : 1];
#pragma unroll
                for (int j = 0; j < NL; ++j) {
                    l[j] = raw_l[j][r];
                    lq[R + 1][r][j] = l[j];
                }
                pre[r] = xform_vec(raw_w[r], l);
            }
#pragma unroll
            for (int qq = 0; qq < Cfg::HALO_ROWS_PER_WAVE; ++qq) {
                VT l[NL > 0 ? NL : 1];
#pragma unroll
                for (int j = 0; j < NL; ++j) l[j] = raw_hl[j][qq];
                hrow_nxt[qq] = xform_vec(raw_hw[qq], l);
            }
#pragma unroll
            for (int qq = 0; qq < Cfg::HALO_X_ITERS; ++qq) {
                T l[NL > 0 ? NL : 1];
#pragma unroll
                for (int j = 0; j < NL; ++j) l[j] = raw_cl[j][qq];
                hcol_nxt[qq] = hcol_frame[qq] ? (T)0 : xform_cell(raw_cw[qq], l);
            }
            // (neither the optimiser may sink these transforms towards their uses nor the scheduler lift the
            //  requests below above them: the staging registers would then be live twice and the compiler falls
            //  back to copying loads in flight, i.e. to waiting for them in the iteration that issued them)
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                asm volatile("" : "+v"(pre[r]));
#pragma unroll
                for (int j = 0; j < NL; ++j) asm volatile("" : "+v"(lq[R + 1][r][j]));
            }
#pragma unroll
            for (int qq = 0; qq < Cfg::HALO_ROWS_PER_WAVE; ++qq) asm volatile("" : "+v"(hrow_nxt[qq]));
#pragma unroll
            for (int qq = 0; qq < Cfg::HALO_X_ITERS; ++qq) asm volatile("" : "+v"(hcol_nxt[qq]));
            __builtin_amdgcn_sched_barrier(0);
            // ... and the same registers take the next set, behind this iteration's stores: halo of plane z+3, own
            // cells of plane z+R+2
            issue_raw(z + 3, z + R + 2);
        }
        __syncthreads();
        // ---- 4. rotate the register pipeline
#pragma unroll
        for (int m = 0; m < 2 * R; ++m)
#pragma unroll
            for (int r = 0; r < RY; ++r) q[m][r] = q[m + 1][r];
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            q[2 * R][r] = pre[r];
            if constexpr (VG == 0) ab_a[r] = pre_a[r];
            if constexpr (!ABV) ab_b[r] = pre_b[r];
        }
        if constexpr (!DEEP) {
#pragma unroll
            for (int qq = 0; qq < Cfg::HALO_ROWS_PER_WAVE; ++qq) hrow_nxt[qq] = hrow_pre[qq];
#pragma unroll
            for (int qq = 0; qq < Cfg::HALO_X_ITERS; ++qq) hcol_nxt[qq] = hcol_pre[qq];
        }
        if constexpr (XF) {
#pragma unroll
            for (int m = 0; m <= R; ++m)
#pragma unroll
                for (int r = 0; r < RY; ++r)
#pragma unroll
                    for (int j = 0; j < NL; ++j) lq[m][r][j] = lq[m + 1][r][j];
        }
    }
    if constexpr (OBS) {
        const double sums[4] = {ob_e, ob_n, ob_v, ob_r};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double s = wafer_block_sum<NW>(sums[q], red, tid);
            if (tid == 0) partials[(size_t)q * pstride + blockIdx.x] = s;
        }
    }
    if constexpr (NORM) {
        const double s = wafer_block_sum<NW>(acc, red, tid);
        if (tid == 0) partials[blockIdx.x] = s;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const double t = wafer_block_sum<NW>(acc_t[j], red, tid);
            if (tid == 0) partials[(size_t)(1 + j) * pstride + blockIdx.x] = t;
        }
    }
}

template <typename T, typename C, int R, int RY, int NLOW, bool NT, bool ABV, bool XF = false, int NW = 4, int VG = 0, int VIRT = -1, bool DEEP = false>
static inline hipError_t wafer_launch_step_lds_ry(const WaferTuning &t, WaferStepArgs a, const WaferLdsOpts &o, const T *phi,
                                                  const T *pa, const T *pb, T *out, double *partials,
                                                  size_t partials_cap, hipStream_t s,
                                                  const WaferLowPtrs &low = WaferLowPtrs(),
                                                  const double *xscal = nullptr, const double *xgram = nullptr)
{
    using Cfg = WaferLdsCfg<T, R, RY, NW>;
    const WaferGeom &g = a.g;
    a.zchunk = wafer_lds_zchunk<T, R>(t, g, a.lz_hi - a.lz_lo, RY * (NW / 4), a.target_blocks); // tile height NW * RY
    const int ntx = (g.nx + Cfg::TX - 1) / Cfg::TX;
    const int nty = (g.ny + Cfg::TY - 1) / Cfg::TY;
    const int ntz = (a.lz_hi - a.lz_lo + a.zchunk - 1) / a.zchunk;
    const long long nblocks = (long long)ntx * nty * ntz;
    if ((NLOW >= 0 || NLOW == -2) && (size_t)nblocks > partials_cap) return hipErrorInvalidValue;
    hipLaunchKernelGGL((wafer_k_step_lds<T, C, R, RY, NLOW, NT, ABV, XF, NW, VG, VIRT, DEEP>), dim3((unsigned)nblocks), dim3(Cfg::NT), (size_t)o.pad, s,
                       a, ntx, nty, o.swz, phi, pa, pb, out, partials, (long long)partials_cap, low, xscal, xgram);
    return hipGetLastError();
}

// Waves per workgroup of the excited-state step kernels: with one to three stored states 8 waves on a
// 128x16 tile, one workgroup per CU (half the halo rows of phi and of every stored state per tile: at
// 512^3 k = 1 0.87 -> 0.83 ms, k = 2 1.13 -> 1.09, k = 3 1.40 -> 1.31) -- WHERE THAT KERNEL FITS ITS REGISTERS: the 8-wave
// instantiations of the wide stencils with many stored states spill, and a reload from scratch waits with vmcnt(0), every
// prefetch in flight included.  Measured at 512^3 (tools/excited_nw_probe.sh, 8 against 4 waves, ms per step): fp64 SevenPoint
// k = 3 3.67 / 1.89, FivePoint k = 3 1.53 / 1.51; fp32 storage FivePoint k = 2 1.55 / 0.86, k = 3 2.43 / 1.13, SevenPoint
// k = 1 / 2 / 3 1.59 / 0.92, 2.60 / 1.22, 3.68 / 1.52 -- those take the 4-wave 128x8 kernel; everything else is faster on
// 8 waves (fp64 SevenPoint k = 2 1.38 / 1.52 even with 44-176 B of scratch).  WAFER_XF_NW = 4 / 8 forces either.
static inline int wafer_excited_nw(const WaferTuning &t, int nlow, int R, bool f32_storage)
{
    if (nlow < 1 || nlow > 3) return 4;
    if (t.xf_nw == 4 || t.xf_nw == 8) return t.xf_nw;
    if (!f32_storage) return (R == 1 || nlow <= 2) ? 8 : 4;
    return (R == 1 || (R == 2 && nlow == 1)) ? 8 : 4;
}

// excited-state step with `nlow` raw overlaps fused in (fixed tuning: RY 2, NT, a/b from V)
template <typename T, typename C, int R>
static inline hipError_t wafer_launch_step_lds_excited(const WaferTuning &t, WaferStepArgs a, const T *phi, const T *pv, T *out,
                                                       double *partials, size_t partials_cap, int nlow,
                                                       const WaferLowPtrs &low, hipStream_t s,
                                                       const double *xscal = nullptr, const double *xgram = nullptr, int vg = 0)
{
    WaferLdsOpts o = wafer_lds_opts(t);
    o.ry = 2;
    if constexpr (std::is_same<T, double>::value && std::is_same<C, double>::value) {
        // closed-form V in the kernel (fp64, transform-on-load, 8-wave tiles): one HBM stream fewer
        if (vg != 0 && xscal && a.v_in_range != 0 && wafer_excited_nw(t, nlow, R, !std::is_same<T, double>::value) == 8) {
            // the raw staging pipeline (DEEP) where the registers allow it (FivePoint from k = 2 and SevenPoint spill); WAFER_XF_DEEP=0: off
            const bool deep = t.xf_deep != 0;
            // (two workgroups per CU for k = 1 -- 128 VGPRs, 28 B/lane of scratch -- measured: 0.797 against 0.686 ms;
            //  twice the concurrent footprint in the XCD's L2, as without the closed form)
#define WAFER_VG_CASE(NLOW_, VG_)                                                                                          \
    if (nlow == NLOW_ && vg == VG_) {                                                                                      \
        if (deep && (R == 1 || (R == 2 && NLOW_ <= 1)))                                                                    \
            return wafer_launch_step_lds_ry<T, C, R, 2, NLOW_, true, true, true, 8, VG_, 1, (R == 1 || (R == 2 && NLOW_ <= 1))>(t, a, o, phi, pv, pv, out, partials, partials_cap, s, low, xscal, xgram); \
        return wafer_launch_step_lds_ry<T, C, R, 2, NLOW_, true, true, true, 8, VG_, 1>(t, a, o, phi, pv, pv, out, partials, partials_cap, s, low, xscal, xgram); \
    }
            WAFER_VG_CASE(1, 4) WAFER_VG_CASE(2, 4) WAFER_VG_CASE(3, 4)
            WAFER_VG_CASE(1, 7) WAFER_VG_CASE(2, 7) WAFER_VG_CASE(3, 7)
            WAFER_VG_CASE(1, 9) WAFER_VG_CASE(2, 9) WAFER_VG_CASE(3, 9)
#undef WAFER_VG_CASE
        }
    }
    if (xscal) { // transform-on-load: phi is the raw previous step
        if (wafer_excited_nw(t, nlow, R, !std::is_same<T, double>::value) == 8) { // 128x16 tiles, 8 waves: half the halo rows per array
            if constexpr (R == 1 && std::is_same<T, double>::value) { // streamed V on the raw staging pipeline (DEEP) where it fits the registers
                if (t.xf_deep != 0) {
                    if (nlow == 1) return wafer_launch_step_lds_ry<T, C, R, 2, 1, true, true, true, 8, 0, -1, true>(t, a, o, phi, pv, pv, out, partials, partials_cap, s, low, xscal, xgram);
                    if (nlow == 2) return wafer_launch_step_lds_ry<T, C, R, 2, 2, true, true, true, 8, 0, -1, true>(t, a, o, phi, pv, pv, out, partials, partials_cap, s, low, xscal, xgram);
                    if (nlow == 3) return wafer_launch_step_lds_ry<T, C, R, 2, 3, true, true, true, 8, 0, -1, true>(t, a, o, phi, pv, pv, out, partials, partials_cap, s, low, xscal, xgram);
                }
            }
            switch (nlow) {
            case 1: return wafer_launch_step_lds_ry<T, C, R, 2, 1, true, true, true, 8>(t, a, o, phi, pv, pv, out, partials, partials_cap, s, low, xscal, xgram);
            case 2: return wafer_launch_step_lds_ry<T, C, R, 2, 2, true, true, true, 8>(t, a, o, phi, pv, pv, out, partials, partials_cap, s, low, xscal, xgram);
            case 3: return wafer_launch_step_lds_ry<T, C, R, 2, 3, true, true, true, 8>(t, a, o, phi, pv, pv, out, partials, partials_cap, s, low, xscal, xgram);
            default: return hipErrorInvalidValue;
            }
        }
        switch (nlow) {
        case 1: return wafer_launch_step_lds_ry<T, C, R, 2, 1, true, true, true>(t, a, o, phi, pv, pv, out, partials, partials_cap, s, low, xscal, xgram);
        case 2: return wafer_launch_step_lds_ry<T, C, R, 2, 2, true, true, true>(t, a, o, phi, pv, pv, out, partials, partials_cap, s, low, xscal, xgram);
        case 3: return wafer_launch_step_lds_ry<T, C, R, 2, 3, true, true, true>(t, a, o, phi, pv, pv, out, partials, partials_cap, s, low, xscal, xgram);
        case 4: return wafer_launch_step_lds_ry<T, C, R, 2, 4, true, true, true>(t, a, o, phi, pv, pv, out, partials, partials_cap, s, low, xscal, xgram);
        default: return hipErrorInvalidValue;
        }
    }
    if (wafer_excited_nw(t, nlow, R, !std::is_same<T, double>::value) == 8) { // the same tiles as the transform-on-load kernel: same partial sums, same bits
        switch (nlow) {
        case 1: return wafer_launch_step_lds_ry<T, C, R, 2, 1, true, true, false, 8>(t, a, o, phi, pv, pv, out, partials, partials_cap, s, low);
        case 2: return wafer_launch_step_lds_ry<T, C, R, 2, 2, true, true, false, 8>(t, a, o, phi, pv, pv, out, partials, partials_cap, s, low);
        case 3: return wafer_launch_step_lds_ry<T, C, R, 2, 3, true, true, false, 8>(t, a, o, phi, pv, pv, out, partials, partials_cap, s, low);
        default: return hipErrorInvalidValue;
        }
    }
    switch (nlow) {
    case 0: return wafer_launch_step_lds_ry<T, C, R, 2, 0, true, true>(t, a, o, phi, pv, pv, out, partials, partials_cap, s, low);
    case 1: return wafer_launch_step_lds_ry<T, C, R, 2, 1, true, true>(t, a, o, phi, pv, pv, out, partials, partials_cap, s, low);
    case 2: return wafer_launch_step_lds_ry<T, C, R, 2, 2, true, true>(t, a, o, phi, pv, pv, out, partials, partials_cap, s, low);
    case 3: return wafer_launch_step_lds_ry<T, C, R, 2, 3, true, true>(t, a, o, phi, pv, pv, out, partials, partials_cap, s, low);
    case 4: return wafer_launch_step_lds_ry<T, C, R, 2, 4, true, true>(t, a, o, phi, pv, pv, out, partials, partials_cap, s, low);
    default: return hipErrorInvalidValue;
    }
}

// compute_observables on the LDS pipeline: 8 waves on 128x16 tiles (ThreePoint / FivePoint), 4 waves on
// 128x8 (SevenPoint).  *nblocks_out = partial sums written per quantity.
template <typename T, int R>
static inline hipError_t wafer_launch_observables_lds(const WaferTuning &t, WaferStepArgs a, const T *phi, const T *pv, const T *potsub,
                                                      double *partials, size_t partials_cap, hipStream_t s, long long *nblocks_out,
                                                      int vg = 0)
{
    WaferLdsOpts o = wafer_lds_opts(t);
    o.ry = 2;
    constexpr int NW = R <= 2 ? 8 : 4;
    using Cfg = WaferLdsCfg<T, R, 2, NW>;
    WaferLowPtrs low;
    low.p[0] = potsub;
    // (two workgroups per CU -- 110 VGPRs at 512 threads: 0.51 -> 0.37 ms at 512^3 incl. the host sync)
    const int zc = wafer_lds_zchunk<T, R>(t, a.g, a.lz_hi - a.lz_lo, 2 * (NW / 4), a.target_blocks);
    *nblocks_out = (long long)((a.g.nx + Cfg::TX - 1) / Cfg::TX) * ((a.g.ny + Cfg::TY - 1) / Cfg::TY) * ((a.lz_hi - a.lz_lo + zc - 1) / zc);
    // (the closed-form V of the step kernels, template parameter VG, was measured here too: with two workgroups
    //  per CU the kernel is short of issue slots, not of bytes -- 0.40 against 0.37 ms at 512^3 -- so V is streamed)
    (void)vg;
    return wafer_launch_step_lds_ry<T, double, R, 2, -2, true, true, false, NW>(t, a, o, phi, pv, pv, nullptr, partials, partials_cap, s, low);
}

template <typename T, int R>
static inline long long wafer_step_lds_excited_blocks(const WaferTuning &t, const WaferGeom &g, int lz_lo, int lz_hi, int target_blocks,
                                                      int nlow = 0, bool xf = false)
{
    using Cfg = WaferLdsCfg<T, R, 2>;
    (void)xf;
    const int mul = (wafer_excited_nw(t, nlow, R, !std::is_same<T, double>::value) == 8) ? 2 : 1;
    const int TY = Cfg::TY * mul;
    const int zc = wafer_lds_zchunk<T, R>(t, g, lz_hi - lz_lo, 2 * mul, target_blocks);
    return (long long)((g.nx + Cfg::TX - 1) / Cfg::TX) * ((g.ny + TY - 1) / TY) * ((lz_hi - lz_lo + zc - 1) / zc);
}

template <typename T, typename C, int R>
static inline hipError_t wafer_launch_step_lds(const WaferTuning &t, WaferStepArgs a, const T *phi, const T *pa, const T *pb,
                                               const T *pv, T *out, hipStream_t s, int vg = 0)
{
    WaferLdsOpts o = wafer_lds_opts(t, R);
    if (o.abv < 0) o.abv = 1;
    double *partials = nullptr;
    const size_t partials_cap = 0;
    if constexpr (std::is_same<T, double>::value && std::is_same<C, double>::value) {
        // closed-form V (fp64, the default tuning of each stencil order): 16 B per update instead of 24
        const bool dflt = t.lds_ry == 0 && o.abv != 0 && o.nt != 0 && a.v_in_range != 0;
        // (ThreePoint 0.546 -> 0.422 ms/step at 512^3, FivePoint 0.556 -> 0.475; SevenPoint, 4 rows per lane on 4
        //  waves, is short of issue slots: 0.617 -> 0.634 with Coulomb, so it keeps streaming V)
        {
            if (vg != 0 && dflt && R <= 2) {   // (SevenPoint: the closed form measured slower than the stream, profiles/NOTES.md)
                a.target_blocks = (a.target_blocks + 1) / 2; // one workgroup per CU, as below
                if (vg == 4) return wafer_launch_step_lds_ry<T, C, R, 2, -1, true, true, false, 8, 4, 1>(t, a, o, phi, pv, pb, out, partials, partials_cap, s);
                if (vg == 7) return wafer_launch_step_lds_ry<T, C, R, 2, -1, true, true, false, 8, 7, 1>(t, a, o, phi, pv, pb, out, partials, partials_cap, s);
                if (vg == 9) return wafer_launch_step_lds_ry<T, C, R, 2, -1, true, true, false, 8, 9, 1>(t, a, o, phi, pv, pb, out, partials, partials_cap, s);
                return hipErrorInvalidValue;
            }
        }
    }
    { // ThreePoint / FivePoint with a, b from V: 8 waves on a 128x16 tile, one workgroup per CU (half the
      // halo rows per tile: 0.56 -> 0.54 ms/step at 512^3); an explicit WAFER_LDS_RY
      // select the 4-wave kernels
        // SevenPoint as well (round 2): 8 waves x 2 rows on the 128x16 tile -- two waves per SIMD with half the
        // registers each -- against 4 waves x 4 rows (256 VGPRs + 54 AGPRs, one wave per SIMD): 0.561 against
        // 0.605 ms/step at 512^3.  (8 waves x 4 rows on 128x32 tiles spill 240 B/lane: 1.69 ms.)
        const bool eight = t.lds_ry == 0;
        if (eight && o.abv != 0) {
            a.target_blocks = (a.target_blocks + 1) / 2; // one workgroup per CU
            if (o.nt != 0) return wafer_launch_step_lds_ry<T, C, R, 2, -1, true, true, false, 8>(t, a, o, phi, pv, pb, out, partials, partials_cap, s);
            return wafer_launch_step_lds_ry<T, C, R, 2, -1, false, true, false, 8>(t, a, o, phi, pv, pb, out, partials, partials_cap, s);
        }
    }
    // with ABV, V takes a's slot and b is not read
#define WAFER_LDS_CASE(RY_, NT_, ABV_)                                                               \
    if (o.ry == RY_ && (o.nt != 0) == NT_ && (o.abv != 0) == ABV_)                                   \
        return wafer_launch_step_lds_ry<T, C, R, RY_, -1, NT_, ABV_>(t, a, o, phi, ABV_ ? pv : pa, pb,   \
                                                                       out, partials, partials_cap, s);
    WAFER_LDS_CASE(2, false, false)
    WAFER_LDS_CASE(2, true, false)
    WAFER_LDS_CASE(2, false, true)
    WAFER_LDS_CASE(2, true, true)
    WAFER_LDS_CASE(4, false, false)
    WAFER_LDS_CASE(4, true, false)
    WAFER_LDS_CASE(4, false, true)
    WAFER_LDS_CASE(4, true, true)
#undef WAFER_LDS_CASE
    return hipErrorInvalidValue;
}
