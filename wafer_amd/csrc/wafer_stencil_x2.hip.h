// TWO excited-state steps per pass over HBM (ThreePoint, fp64): grid.rs:562-686 with wnum = k > 0.
//
// The reference renormalises and runs modified Gram-Schmidt against the k stored states after EVERY step
// (grid.rs:674-681), which needs 1 + k global sums between two steps.  The step operator A (grid.rs:568-592) is linear, so
// the sums of the first step need not be known while the second is computed:
//
//   x0                      the pass's input as the reference holds it (normalised, projected) -- or any positive multiple
//   Y1 = A x0               first raw step            n_b = sqrt(sum Y1^2),  s^b_j from t_j = sum l_j Y1      (:675-680)
//   x1 = Y1 / n_b - sum_j s^b_j l_j                   the reference's state after step 1 (the scale of x0 has dropped out)
//   Y2 = A x1 = A(Y1) / n_b - sum_j s^b_j M_j         second raw step, with M_j = A l_j stored once per stored state
//   x2 = (Y2 - sum_j sigma_j l_j) / n_c               the reference's state after step 2: sigma_j = n_c s^c_j, n_c = |Y2|
//
// The kernel computes Y1 = A x0 and Z = A Y1 in one pass (Y1 never leaves the CU) and stores the RAW Z; the NEXT pass forms
//   x~ = Z / n_b - sum_j s^b_j M_j - sum_j sigma_j l_j      ( = n_c x2 )
// for every cell it loads (own cells, halo rows, halo columns).  It never divides by n_c: the first thing that happens to
// the next pass's Y1 is the division by ITS norm, so a common positive factor of the input changes no later state -- and n_c,
// the one scalar whose regrouped form would subtract nearly equal sums when the state lies close to the span of the stored
// states, is not needed at all (n_c is 1 - O(dt): the magnitudes stay where the reference has them; the last pass's n_c is
// taken directly, as a sum of squares, when phi is materialised).  The scalars come from 1 + 2k sums taken during the pass --
// sum Y1^2, sum l_j Y1 (they give n_b, s^b by the recurrence of wafer_k_gs_apply) and sum l_j Z, from which, with the constant
// matrix <l_j, M_i>,   sum l_j Y2 = sum l_j Z / n_b - sum_i s^b_i <l_j, M_i>   and sigma_j by the same recurrence
// (wafer_k_x2_coeffs, one thread).  Exact in exact arithmetic; in fp64 the state stays within 1e-16 of the reference's
// sequence per cell (tests hold it to the 1e-13 of every excited-state test).
// Streams per TWO updates: Z in, k x l_j, k x M_j, Z out (+ V unless its closed form is evaluated): (2 + 2k) * 8 B against
// 2 * (2 + k) * 8 B for two one-step passes -- 16 / 24 / 32 B per update at k = 1 / 2 / 3 against 24 / 32 / 40.
//
// Structure: the three-step ground-state kernel's (wafer_stencil_fused3.hip.h) with one level less and a transform on load.
// Eight waves on a 128 x (8 RY) tile marched up z; every wave owns RY rows at both levels plus ONE extra slot:
//   wave 0   row y0-1   (x0 queue, Y1)           wave 7   row y0+TY   (x0 queue, Y1)
//   wave 1   row y0-2   (x0 staged to LDS only)  wave 6   row y0+TY+1 (x0 staged to LDS only)
//   waves 2..5   the 4 (TY + 4) x0 halo-column cells, one per lane: Y1 on the inner column
// a, b of a cell are formed once per pass (level 1) and ride to level 2 as a and b dt.
// The sums of a plane are all taken when its Z is produced (Y1 of that plane is still in its z-queue), which is three
// planes after l_j of that plane was loaded for the transform: the stored states' values at the lane's own cells wait in an
// LDS queue private to each lane (no barrier involved).  Registers set the tile height: 128 x 16 tiles (RY = 2) for one stored
// state, 128 x 8 tiles (RY = 1) for two and three (wafer_x2_ry).
#pragma once
#include <hip/hip_runtime.h>
#include "wafer_geom.h"
#include "wafer_stencil.hip.h"
#include "wafer_stencil_lds.hip.h"
#include "wafer_stencil_fused2.hip.h"
#include "wafer_rowwalk.h"
#include "wafer_storage.h"

// the coefficient block the load transform reads (device memory, doubles):  x~ = w * W0 - sum_j m_j SB_j - sum_j l_j SC_j
enum {
    WAFER_X2_W0 = 0,                      // 1 / n_b
    WAFER_X2_SB = 1,                      // s^b_j, j < WAFER_MAX_LOW
    WAFER_X2_SC = 1 + WAFER_MAX_LOW,      // sigma_j
    WAFER_X2_COEF_DOUBLES = 1 + 2 * WAFER_MAX_LOW
};
enum { WAFER_X2_MAX_LOW = 3 };

struct WaferX2Ptrs {   // arrays of the context's storage type (double, or float for fp32 storage: the kernels' first template argument)
    const void *l[WAFER_X2_MAX_LOW] = {nullptr, nullptr, nullptr};   // stored states
    const void *m[WAFER_X2_MAX_LOW] = {nullptr, nullptr, nullptr};   // M_j = A l_j
};

// sums a pass leaves: partials[q * pstride + workgroup], q in this order
//   0: sum Y1^2   1 .. k: sum l_j Y1   k+1 .. 2k: sum l_j Z
static inline int wafer_x2_nsums(int k) { return 1 + 2 * k; }

// where in the plane iteration a wave issues its requests (issue_group in the kernel): 0 = at the top, 1 = behind level 1 of the main
// rows, 2 = behind level 1 of the extra slot, 3 = behind level 2.  A: the main rows' input (and V), L: their stored states, M: the images
// M_j, X: the extra slot's.  Measured per tile (profiles/r04_ab_x2_request_placement.jsonl): on the 128 x 16 tile M and X move back
// (0.665 -> 0.63 ms/step at k = 2); on the 128 x 8 tile only the extra slot's requests do (0.964 -> 0.87 at k = 3).
// (positions as constants of the kernel: wafer_stencil_x2_iter.inc.h.  Falling issue priorities through the iteration, the three-step
//  kernel's -3 %, measured +-0.5 % here and are not used; ring z-queues are: -2 %.  Diagnostic builds: -DWAFER_DIAG=<bits>, 2 = every
//  prefetch asks for the column's first planes again, 4 = nothing is stored -- see wafer_stencil_fused3.hip.h.)
#ifndef WAFER_DIAG
#define WAFER_DIAG 0
#endif
template <int RY_>
struct WaferX2Cfg {
    static constexpr int VEC = 2;
    static constexpr int RY = RY_;
    static constexpr int NW = 8;
    static constexpr int NT_ = NW * 64;
    static constexpr int TX = 64 * VEC, TY = NW * RY;
    static constexpr int HC0 = 2, HC1 = 1;              // halo columns per side of x0 / Y1
    static constexpr int HX0 = 2, HX1 = 2;              // ... as kept in LDS (VEC aligned)
    static constexpr int LP0 = TX + 2 * HX0, LP1 = TX + 2 * HX1;
    static constexpr int ROWS0 = TY + 4, ROWS1 = TY + 2;
    static constexpr int TILE0 = ROWS0 * LP0, TILE1 = ROWS1 * LP1;
    static constexpr int NCOL = 2 * HC0 * ROWS0;        // x0 halo-column cells per plane
    static constexpr int HCW0 = 2, HCWN = 4;            // waves HCW0 .. HCW0 + HCWN - 1 take them, one per lane
    static constexpr int CPW = (NCOL + HCWN - 1) / HCWN;
    static_assert(CPW <= 64, "one halo-column cell per lane");
    static constexpr int QSLOT = TY * TX;               // elements of one array's plane in the LDS queue
};

// x~ = w / n_b - sum_j m_j s^b_j - sum_j l_j sigma_j, evaluated with the reciprocal of n_b and SEPARATE products and differences:
// measured on one box, 512^3, k = 1 (profiles/r04_ab_x2_transform_k1.jsonl): 0.461 ms per step against 0.480 with a true
// division and 0.496 with fused multiply-adds -- the independent products overlap, the fused chain is serial.  Each rounding
// is one ulp of a cell's value; the pass is held to the excited-state bar, 1e-13 per cell.
template <int NL>
struct WaferX2Coef {
    double w0, sb[NL], sc[NL];
};
template <int NL>
__device__ __forceinline__ double wafer_x2_xform(const WaferX2Coef<NL> &k, double w, const double *l, const double *m, double *u_out = nullptr)
{
    double x = w * k.w0;
#pragma unroll
    for (int j = 0; j < NL; ++j) x = x - m[j] * k.sb[j];
    if (u_out) *u_out = x;   // Y2 up to the factor the pass carries: its norm is n_c (wafer_k_x2_apply)
#pragma unroll
    for (int j = 0; j < NL; ++j) x = x - l[j] * k.sc[j];
    return x;
}

// XS (grids made of whole tiles): one full-vector store per main row in EVERY plane iteration (the two iterations that fill the
// pipeline store into the column's first plane, which the first real store overwrites; their sums are not taken), so that the
// wait for the prefetched planes behind the loop's barrier is an exact count that leaves the stores in flight -- with the stores
// inside conditions it was vmcnt(0): every wave sat out the completion of the store it had issued just before the barrier.
// TS: the storage tag of the fused ground-state kernels (wafer_storage.h): double, or wafer_f32_wide = fp32 STORAGE with fp64 arithmetic
// (round 6: dtype f32 / f32fast, whose excited-state steps compute in fp64) -- the raw pass result Z, V, the stored states and their images
// are float in HBM (8 bytes per lane and request); queues, LDS tiles, the lane-private queue of the l_j and every sum are double, the fp64
// kernel's.  What arrives is widened where the pipelines rotate, behind the barrier; Z is rounded to float BEFORE its sums are taken, so the
// coefficients of the next pass's transform belong to exactly the numbers that pass will load.
template <typename TS, int RY, int NL, int VG, bool VIR, bool XS = false>
__global__ __launch_bounds__(512) void wafer_k_xstep2(WaferStepArgs a, int ntx, int nty, int swz, const typename WaferF3Store<TS>::S *__restrict__ phi,
                                                       const typename WaferF3Store<TS>::S *__restrict__ pv, typename WaferF3Store<TS>::S *__restrict__ out,
                                                       double *__restrict__ partials, long long pstride, WaferX2Ptrs st,
                                                       const double *__restrict__ coef)
{
    using Cfg = WaferX2Cfg<RY>;
    typedef double T;
    typedef double C;
    using VT = typename WaferVec<double>::type;
    using ST = typename WaferF3Store<TS>::S;                       // the arrays in HBM
    typedef ST __attribute__((ext_vector_type(2))) SVT;            // a lane's request
    [[maybe_unused]] constexpr bool WIDE = !std::is_same<ST, T>::value;
    auto widen = [](const SVT &x) -> VT { return wafer_f3_widen<SVT, VT, 2>(x); };
    auto as_stored = [](T x) -> T { return (T)(ST)x; };          // what the array will hold (fp32 storage: rounded once, like a store and a load)
#define WAFER_X2_L(j) (static_cast<const ST *>(st.l[j]))
#define WAFER_X2_M(j) (static_cast<const ST *>(st.m[j]))
    constexpr int R = 1;
    constexpr int VEC = Cfg::VEC, TX = Cfg::TX, TY = Cfg::TY;
    constexpr int HX0 = Cfg::HX0, HX1 = Cfg::HX1, LP0 = Cfg::LP0, LP1 = Cfg::LP1;
    constexpr int QS = Cfg::QSLOT;
    static_assert(NL >= 1 && NL <= WAFER_X2_MAX_LOW, "one to three stored states");
    __shared__ __attribute__((aligned(16))) T lds0[2 * Cfg::TILE0];
    __shared__ __attribute__((aligned(16))) T lds1[2 * Cfg::TILE1];
    // The stored states' values at the lane's own cells wait three planes between the transform and the sums: in an LDS queue of
    // three slots per state (written right after the transform, read when that plane's Z is produced; private to each lane, no
    // barrier), or -- 128 x 16 tiles with two stored states, where LDS has room for two slots only -- two slots plus one plane
    // in registers (HOLD).  Only the l_j wait: the M_j are used by the transform alone (see the head of this file).
    // fp32 storage: the queue holds the stored states as they arrived, in float -- half the bytes, so three slots fit beside the tall tile's
    // LDS for two AND three stored states (154 KB at k = 3), and nothing is held in registers
    using QT = ST;
    typedef QT __attribute__((ext_vector_type(2))) QVT;
    constexpr bool HOLD = (RY == 2 && NL == 2 && !WIDE);
    // ring z-queues (the plane loop unrolled by three, no shifts): the kernels with exact store counts, as in the three-step kernel
    constexpr bool RING = XS;
    constexpr int NSLOT = HOLD ? 2 : 3;
    __shared__ __attribute__((aligned(16))) QT ldsq[NSLOT * NL * QS];   // [slot][state][row][x]
    __shared__ double red[Cfg::NW];

    const WaferGeom &g = a.g;
    int bid = blockIdx.x;
    if (swz) { // XCD-contiguous tile order (wafer_stencil_lds.hip.h)
        const int n = gridDim.x, q = n >> 3, r = n & 7, k = bid & 7;
        bid = k * q + min(k, r) + (bid >> 3);
    }
    const int tx_i = bid % ntx, ty_i = (bid / ntx) % nty, tz_i = bid / (ntx * nty);
    const int zs = a.lz_lo + tz_i * a.zchunk;
    const int ze = min(zs + a.zchunk, a.lz_hi);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = tx_i * TX, y0 = ty_i * TY;
    const C dt = (C)a.dt;
    constexpr bool vir = VIR;
    const WaferDen<C> den = wafer_den<C>(a, vir);
    const bool x_row = wave < 2 || wave >= 6;
    const bool x_l1 = wave == 0 || wave == 7;            // the halo row next to the tile: Y1 as well

    WaferX2Coef<NL> kf;
    kf.w0 = coef[WAFER_X2_W0];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        kf.sb[j] = coef[WAFER_X2_SB + j];
        kf.sc[j] = coef[WAFER_X2_SC + j];
    }
    [[maybe_unused]] WaferPotArgs vgen;
    if constexpr (VG != 0) {
        vgen.g = g;
        vgen.type = VG;
        vgen.dn = a.vg_dn; vgen.dt = a.dt; vgen.mass = a.vg_mass; vgen.sig = a.vg_sig;
        vgen.mu_t = vgen.alphas_2pit = vgen.xi_coef = vgen.xi_fac = 0.0;
    }

    VT zero;
    zero[0] = zero[1] = 0.0;
    SVT szero;
    szero[0] = szero[1] = ST(0);
    const int xl = lane * VEC, xi = x0 + xl;
    const unsigned xlu = (unsigned)(lane * VEC);

    // ---- main rows
    int yrow[RY];
    bool rowwk[RY];
    long long rowoff[RY];
    int qoff[RY];
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        const int y = y0 + wave * RY + r;
        yrow[r] = y;
        rowwk[r] = y < g.ny;
        rowoff[r] = (long long)(y + R) * g.pitch + g.xoff + R + x0;
        qoff[r] = (wave * RY + r) * TX + xl;
    }
    // ---- the extra halo row (requests of rows outside the work area are redirected to the wave's own first row and the
    //      value replaced by the zero it stands for, as in the three-step kernel)
    const int xy = wave == 0 ? y0 - 1 : wave == 1 ? y0 - 2 : wave == 6 ? y0 + TY + 1 : y0 + TY;
    const bool xwk = x_row && xy >= 0 && xy < g.ny;
    const bool xy_out = xy < 0 || xy >= g.ny;
    const long long xoff_row = xy_out ? rowoff[0] : (long long)(xy + R) * g.pitch + g.xoff + R + x0;
    // ---- halo-column cell of this lane (waves 2..5): cell c: row c / 4 of the x0 tile, k = c % 4: k < 2: column x0-1-k,
    //      else column x0+TX+(k-2)
    const int cidx = min((wave - Cfg::HCW0) * Cfg::CPW + lane, Cfg::NCOL - 1);
    const int crow = cidx / (2 * Cfg::HC0), ck = cidx % (2 * Cfg::HC0);
    const int ckk = (ck < Cfg::HC0) ? ck : ck - Cfg::HC0;
    const int clc = (ck < Cfg::HC0) ? (-1 - ckk) : (TX + ckk);
    const int cxw = x0 + clc, cy = y0 - 2 + crow;
    const bool c_ok = !x_row && lane < Cfg::CPW && (wave - Cfg::HCW0) * Cfg::CPW + lane < Cfg::NCOL;
    const bool c_wk = cy >= 0 && cy < g.ny && cxw >= 0 && cxw < g.nx;
    const bool c_l1 = c_ok && ckk < Cfg::HC1 && crow >= 2 && crow < Cfg::ROWS0 - 2;   // rows y0 .. y0+TY-1: what level 2 reads
    const bool c_xout = cxw < 0 || cxw >= g.nx || cy < 0 || cy >= g.ny;
    const long long c_off = (long long)((cy < 0 ? y0 : cy >= g.ny ? y0 + TY - 1 : cy) + R) * g.pitch + g.xoff + R +
                            ((cxw < 0 || cxw >= g.nx) ? (ck < Cfg::HC0 ? x0 : x0 + TX - 1) : cxw);
    const int c_lds0 = crow * LP0 + HX0 + clc, c_lds1 = (crow - 1) * LP1 + HX1 + clc;

    auto work_plane = [&](int p) {
        const int kg = g.z_begin + (p - g.G);
        return kg >= 0 && kg < g.nz;
    };
    // level 1: a, b from V (potential.rs:104-110); what rides to level 2 is a and the product b * dt (grid.rs:580-589:
    // w * a + b * dt * S / den, left to right)
    auto update_keep = [&](C w, C vv, C S, C &ca, C &cbdt) -> T {
        C cb;
        wafer_ab_from_v<C>(vv, dt, vir, ca, cb);
        cbdt = cb * dt;
        return (T)(w * ca + wafer_div_invariant<C>(cbdt * S, den));
    };
    auto update_with = [&](C w, C ca, C cbdt, C S) -> T { return (T)(w * ca + wafer_div_invariant<C>(cbdt * S, den)); };
    // V of a cell: streamed (vv) or its closed form at the padded global index
    auto v_at = [&](C vv, int xw, int yw, int lzp) -> C {
        if constexpr (VG != 0) return (C)wafer_vgen_at<VG>(vgen, xw + R, yw + R, g.zp_of(lzp));
        else return vv;
    };

    // raw loads of one plane: the pass's input and the stored states / their images at the same cells
    auto xform_vec = [&](VT w, const VT *l, const VT *m) -> VT {
        VT x;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            double lv[NL], mv[NL];
#pragma unroll
            for (int j = 0; j < NL; ++j) { lv[j] = l[j][v]; mv[j] = m[j][v]; }
            x[v] = wafer_x2_xform<NL>(kf, w[v], lv, mv);
        }
        return x;
    };

    // ---- state
    const int z1 = zs - 1;   // level 1 produces planes z1 .. ze, level 2 planes zs .. ze - 1
    VT q0[3][RY], q1[3][RY], vcur[RY], caq[RY], cbq[RY];
    [[maybe_unused]] VT hold_l[NL][RY];     // HOLD: the stored states at the lane's own cells, plane z + 1 (transformed last iteration)
    // queue slot of plane p (HOLD: two slots by parity, written one iteration late from hold_l)
    auto qslot = [&](int p) -> QT * { return ldsq + ((HOLD ? (p & 1) : (p % 3)) * NL) * QS; };
    auto narrow_q = [](const VT &x) -> QVT { QVT r; r[0] = (QT)x[0]; r[1] = (QT)x[1]; return r; };   // (exact: the value came from a QT)
    VT xq0[3], xv;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
#pragma unroll
        for (int r = 0; r < RY; ++r) q1[m][r] = zero;
    }
#pragma unroll
    for (int r = 0; r < RY; ++r) vcur[r] = caq[r] = cbq[r] = zero;
    xv = zero;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const long long po = (long long)(z1 + (m - 1)) * g.plane;
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            VT l[NL], mm[NL];
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                l[j] = widen(*reinterpret_cast<const SVT *>((WAFER_X2_L(j) + po + rowoff[r]) + xlu));
                mm[j] = widen(*reinterpret_cast<const SVT *>((WAFER_X2_M(j) + po + rowoff[r]) + xlu));
                if (m == 2) {
                    if constexpr (HOLD) hold_l[j][r] = l[j];
                    else *reinterpret_cast<QVT *>(qslot(z1 + 1) + j * QS + qoff[r]) = narrow_q(l[j]);
                }
            }
            q0[m][r] = xform_vec(widen(*reinterpret_cast<const SVT *>((phi + po + rowoff[r]) + xlu)), l, mm);
        }
        if (x_row) {
            VT l[NL], mm[NL];
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                l[j] = widen(*reinterpret_cast<const SVT *>((WAFER_X2_L(j) + po + xoff_row) + xlu));
                mm[j] = widen(*reinterpret_cast<const SVT *>((WAFER_X2_M(j) + po + xoff_row) + xlu));
            }
            xq0[m] = xform_vec(widen(*reinterpret_cast<const SVT *>((phi + po + xoff_row) + xlu)), l, mm);
        } else {
            double l[NL], mm[NL];
#pragma unroll
            for (int j = 0; j < NL; ++j) { l[j] = (double)WAFER_X2_L(j)[po + c_off]; mm[j] = (double)WAFER_X2_M(j)[po + c_off]; }
            xq0[m] = zero;
            xq0[m][0] = wafer_x2_xform<NL>(kf, (double)phi[po + c_off], l, mm);
        }
    }
    if constexpr (VG == 0) {
        const long long po = (long long)z1 * g.plane;
#pragma unroll
        for (int r = 0; r < RY; ++r) vcur[r] = widen(*reinterpret_cast<const SVT *>((pv + po + rowoff[r]) + xlu));
        if (x_row) xv = widen(*reinterpret_cast<const SVT *>((pv + po + xoff_row) + xlu));
        else xv[0] = (T)pv[po + c_off];
    }
    for (int i = tid; i < 2 * Cfg::TILE0; i += Cfg::NT_) lds0[i] = T(0);
    for (int i = tid; i < 2 * Cfg::TILE1; i += Cfg::NT_) lds1[i] = T(0);
    __syncthreads();
    {
        T *t0 = lds0 + (z1 & 1) * Cfg::TILE0;
#pragma unroll
        for (int r = 0; r < RY; ++r) *reinterpret_cast<VT *>(t0 + (yrow[r] - (y0 - 2)) * LP0 + HX0 + xl) = q0[1][r];
        if (x_row) *reinterpret_cast<VT *>(t0 + (xy - (y0 - 2)) * LP0 + HX0 + xl) = xy_out ? zero : xq0[1];
        else if (c_ok) t0[c_lds0] = c_xout ? T(0) : xq0[1][0];
    }
    __syncthreads();

    double acc_y = 0.0, acc_yl[NL], acc_zl[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) acc_yl[j] = acc_zl[j] = 0.0;

    const int niter = (ze - zs) + 2;
#define WAFER_X2_Q0(m) (RING ? ((m) + WAFER_X2_PH) % 3 : (m))
#define WAFER_X2_Q1(m) (RING ? ((m) + WAFER_X2_PH + 1) % 3 : (m))
    if constexpr (RING) {
        for (int it0 = 0; it0 < niter; it0 += 3) {
            {
                const int it = it0;
#define WAFER_X2_PH 0
#include "wafer_stencil_x2_iter.inc.h"
#undef WAFER_X2_PH
            }
            if (it0 + 1 >= niter) break;
            {
                const int it = it0 + 1;
#define WAFER_X2_PH 1
#include "wafer_stencil_x2_iter.inc.h"
#undef WAFER_X2_PH
            }
            if (it0 + 2 >= niter) break;
            {
                const int it = it0 + 2;
#define WAFER_X2_PH 2
#include "wafer_stencil_x2_iter.inc.h"
#undef WAFER_X2_PH
            }
        }
    } else {
        for (int it = 0; it < niter; ++it) {
#define WAFER_X2_PH 0
#include "wafer_stencil_x2_iter.inc.h"
#undef WAFER_X2_PH
        }
    }
#undef WAFER_X2_Q0
#undef WAFER_X2_Q1
#undef WAFER_X2_L
#undef WAFER_X2_M
    // ---- the workgroup's partial sums
    {
        int q = 0;
        auto put = [&](double v) {
            const double s = wafer_block_sum<Cfg::NW>(v, red, tid);
            if (tid == 0) partials[(size_t)q * pstride + blockIdx.x] = s;
            ++q;
        };
        put(acc_y);
#pragma unroll
        for (int j = 0; j < NL; ++j) put(acc_yl[j]);
#pragma unroll
        for (int j = 0; j < NL; ++j) put(acc_zl[j]);
    }
}

// ---- the scalars between two passes (one thread) ------------------------------------------------------------------
// kind 1: `sums` = what a ONE-step kernel left (sum Y^2, t_j = sum l_j Y): the buffer holds Y = A x and the transform is the
//         reference's x = Y / n - sum_j s_j l_j: W0 = 1 / n, SB = 0, SC = s (0 * M_j is exact);
// kind 2: `sums` = the 1 + 2k sums of a two-step pass (order: wafer_x2_nsums).
// gram[j * WAFER_MAX_LOW + i] = <l_j, l_i> (i < j), amat[j * WAFER_MAX_LOW + i] = <l_j, M_i>.
__global__ void wafer_k_x2_coeffs(int kind, int k, const double *__restrict__ sums, const double *__restrict__ gram,
                                  const double *__restrict__ amat, double *__restrict__ coef)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double sb[WAFER_MAX_LOW] = {0, 0, 0, 0}, sc[WAFER_MAX_LOW] = {0, 0, 0, 0};
    // the reference's sequential overlaps from raw ones: s_j = t_j / n - sum_{i<j} s_i G_ji  (wafer_k_gs_apply)
    auto mgs = [&](double n, const double *t, double *s) {
        for (int j = 0; j < k; ++j) {
            double v = t[j] / n;
            for (int i = 0; i < j; ++i) v -= s[i] * gram[j * WAFER_MAX_LOW + i];
            s[j] = v;
        }
    };
    const double n = sqrt(sums[0]);
    if (kind == 1) {
        mgs(n, sums + 1, sc);
    } else {
        const double *t1 = sums + 1, *zl = sums + k + 1;
        mgs(n, t1, sb);
        double t2[WAFER_MAX_LOW];
        for (int j = 0; j < k; ++j) {   // sum l_j Y2 (in the pass's scale)
            double v = zl[j] / n;
            for (int i = 0; i < k; ++i) v -= sb[i] * amat[j * WAFER_MAX_LOW + i];
            t2[j] = v;
        }
        mgs(1.0, t2, sc);               // sigma_j = n_c s^c_j: the same recurrence, not divided by the norm
    }
    coef[WAFER_X2_W0] = 1.0 / n;
    for (int j = 0; j < WAFER_MAX_LOW; ++j) {
        coef[WAFER_X2_SB + j] = sb[j];
        coef[WAFER_X2_SC + j] = sc[j];
    }
}

// ---- phi materialised after the last pass: x~ written in place, sum Y2^2 (= n_c^2 in the pass's scale) to partials[workgroup];
//      the caller divides by its square root (wafer_k_row_op<2>: grid.rs:467) ------------------------------------------------
template <typename TS, int NL>
__global__ __launch_bounds__(256) void wafer_k_x2_apply(WaferRowArgs a, typename WaferF3Store<TS>::S *__restrict__ phi, WaferX2Ptrs st,
                                                        const double *__restrict__ coef, double *__restrict__ partials)
{
    using VT = typename WaferRowVec<double>::type;
    using ST = typename WaferF3Store<TS>::S;
    typedef ST __attribute__((ext_vector_type(2))) SVT;
    auto widen = [](const SVT &x) -> VT { return wafer_f3_widen<SVT, VT, 2>(x); };
    constexpr int VEC = 2;
    __shared__ double red[4];
    const WaferGeom &g = a.g;
    WaferX2Coef<NL> kf;
    kf.w0 = coef[WAFER_X2_W0];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        kf.sb[j] = coef[WAFER_X2_SB + j];
        kf.sc[j] = coef[WAFER_X2_SC + j];
    }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nsegx = (g.nx + 64 * VEC - 1) / (64 * VEC);
    const int wlim = g.pitch - g.xoff - g.R;
    double acc = 0.0;
    WAFER_ROW_WALK_BEGIN(a, g)
    for (int xs = 0; xs < nsegx; ++xs) {
        const int xi = xs * 64 * VEC + lane * VEC;
        if (xi >= wlim || xi >= g.nx) continue;
        const long long p = rowp + xi;
        const VT w = widen(*reinterpret_cast<const SVT *>(phi + p));
        VT l[NL], m[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            l[j] = widen(__builtin_nontemporal_load(reinterpret_cast<const SVT *>(static_cast<const ST *>(st.l[j]) + p)));
            m[j] = widen(__builtin_nontemporal_load(reinterpret_cast<const SVT *>(static_cast<const ST *>(st.m[j]) + p)));
        }
        VT r;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            double lv[NL], mv[NL], u;
#pragma unroll
            for (int j = 0; j < NL; ++j) { lv[j] = l[j][v]; mv[j] = m[j][v]; }
            r[v] = wafer_x2_xform<NL>(kf, w[v], lv, mv, &u);
            if (xi + v < g.nx) acc += u * u;
        }
        SVT rs;
#pragma unroll
        for (int v = 0; v < VEC; ++v) rs[v] = (ST)r[v];
        if (xi + VEC <= g.nx) {
            *reinterpret_cast<SVT *>(phi + p) = rs;
        } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                if (xi + v < g.nx) phi[p + v] = rs[v];
        }
    }
    WAFER_ROW_WALK_END(g)
    const double sum = wafer_block_sum<4>(acc, red, threadIdx.x);
    if (threadIdx.x == 0) partials[blockIdx.x] = sum;
}

// ---- host side -----------------------------------------------------------------------------------------------------
// rows per lane for k stored states: the registers decide.  128 x 16 tiles (RY = 2) for one stored state (230-246 VGPRs, closed-form
// and streamed V alike); for two the tall tile needs 256 VGPRs + 72 B of scratch per lane and loses to the lower one (512^3
// Coulomb, same box: 0.793 against 0.755 ms/step; one step per pass 0.817), so two and three run on 128 x 8 tiles (RY = 1).
// WAFER_X2_RY overrides for k <= 2 (tests, sweeps).
static inline int wafer_x2_ry(const WaferTuning &t, int k, int vg, bool wide = false)
{
    (void)vg;
    // three stored states: the lane-private queue of the tall tile does not fit the LDS in double (176 KB); in float -- fp32 storage -- it does
    // (154 KB), and there the tall tile is the default at every k (fewer halo rows per row: what the wide planes of a slab want)
    if (k > 2 && !wide) return 1;
    if (t.x2_ry == 1 || t.x2_ry == 2) return t.x2_ry;
    return 2;              // 128 x 16 (k = 2: since the requests are spread over the iteration the kernel fits 229 VGPRs; 0.648 against 0.677 ms/step)
}
static inline void wafer_x2_tile(const WaferTuning &t, int k, int vg, int *tx, int *ty, bool wide = false)
{
    *tx = 128;
    *ty = 8 * wafer_x2_ry(t, k, vg, wide);
}
static inline int wafer_x2_zchunk(const WaferTuning &t, const WaferGeom &g, int k, int vg, int nplanes, int target_blocks, bool wide = false)
{
    int tx, ty;
    wafer_x2_tile(t, k, vg, &tx, &ty, wide);
    if (t.zchunk > 0) return t.zchunk;
    const long long per_layer = (long long)((g.nx + tx - 1) / tx) * ((g.ny + ty - 1) / ty);
    const long long target = t.target_blocks > 0 ? t.target_blocks : (target_blocks > 0 ? target_blocks : 256);
    return wafer_pick_zchunk(per_layer, nplanes, target, 5);   // two iterations of pipeline fill + the prologue's three planes
}
static inline long long wafer_x2_blocks(const WaferTuning &t, const WaferGeom &g, int k, int vg, int lz_lo, int lz_hi, int target_blocks, bool wide = false)
{
    int tx, ty;
    wafer_x2_tile(t, k, vg, &tx, &ty, wide);
    const int zc = wafer_x2_zchunk(t, g, k, vg, lz_hi - lz_lo, target_blocks, wide);
    return (long long)((g.nx + tx - 1) / tx) * ((g.ny + ty - 1) / ty) * ((lz_hi - lz_lo + zc - 1) / zc);
}

template <typename TS, int RY, int NL, int VG>
static inline hipError_t wafer_launch_xstep2_one(const WaferTuning &t, WaferStepArgs a, const typename WaferF3Store<TS>::S *phi,
                                                 const typename WaferF3Store<TS>::S *pv, typename WaferF3Store<TS>::S *out,
                                                 double *partials, size_t partials_cap, const WaferX2Ptrs &st, const double *coef,
                                                 hipStream_t s)
{
    using Cfg = WaferX2Cfg<RY>;
    const WaferGeom &g = a.g;
    a.zchunk = wafer_x2_zchunk(t, g, NL, VG, a.lz_hi - a.lz_lo, a.target_blocks, !std::is_same<typename WaferF3Store<TS>::S, double>::value);
    const int ntx = (g.nx + Cfg::TX - 1) / Cfg::TX, nty = (g.ny + Cfg::TY - 1) / Cfg::TY;
    const int ntz = (a.lz_hi - a.lz_lo + a.zchunk - 1) / a.zchunk;
    const long long nblocks = (long long)ntx * nty * ntz;
    if ((size_t)nblocks > partials_cap) return hipErrorInvalidValue;
    if (t.f3_xs != 0 && g.nx % Cfg::TX == 0 && g.ny % Cfg::TY == 0)
        hipLaunchKernelGGL((wafer_k_xstep2<TS, RY, NL, VG, true, true>), dim3((unsigned)nblocks), dim3(Cfg::NT_), 0, s, a, ntx, nty, t.swz, phi, pv,
                           out, partials, (long long)partials_cap, st, coef);
    else
        hipLaunchKernelGGL((wafer_k_xstep2<TS, RY, NL, VG, true, false>), dim3((unsigned)nblocks), dim3(Cfg::NT_), 0, s, a, ntx, nty, t.swz, phi, pv,
                           out, partials, (long long)partials_cap, st, coef);
    return hipGetLastError();
}

// out = A A x with x = the load transform of `phi` (coef); the 2 + 3k sums of the pass go to partials[q * partials_cap + wg].
// Needs a.v_in_range (the short reciprocal); vg: the closed form V was generated from, or 0 (streamed).
template <typename TS>
static inline hipError_t wafer_launch_xstep2(const WaferTuning &t, const WaferStepArgs &a, int k, int vg, const typename WaferF3Store<TS>::S *phi,
                                             const typename WaferF3Store<TS>::S *pv, typename WaferF3Store<TS>::S *out, double *partials,
                                             size_t partials_cap, const WaferX2Ptrs &st, const double *coef, hipStream_t s)
{
    const int ry = wafer_x2_ry(t, k, vg, !std::is_same<typename WaferF3Store<TS>::S, double>::value);
    constexpr bool WIDE = !std::is_same<typename WaferF3Store<TS>::S, double>::value;   // fp32 storage: V is streamed (no closed form there)
#define WAFER_X2_CASE(RY_, NL_, VG_)                                                                                               \
    if constexpr (!WIDE || VG_ == 0)                                                                                               \
        if (ry == RY_ && k == NL_ && vg == VG_)                                                                                    \
            return wafer_launch_xstep2_one<TS, RY_, NL_, VG_>(t, a, phi, pv, out, partials, partials_cap, st, coef, s);
#define WAFER_X2_CASES(RY_, NL_) WAFER_X2_CASE(RY_, NL_, 0) WAFER_X2_CASE(RY_, NL_, 4) WAFER_X2_CASE(RY_, NL_, 7) WAFER_X2_CASE(RY_, NL_, 9)
    WAFER_X2_CASES(2, 1)
    WAFER_X2_CASES(2, 2)
    WAFER_X2_CASES(1, 1)
    WAFER_X2_CASES(1, 2)
    WAFER_X2_CASES(1, 3)
    if constexpr (WIDE) { WAFER_X2_CASE(2, 3, 0) }
#undef WAFER_X2_CASES
#undef WAFER_X2_CASE
    return hipErrorInvalidValue;
}

// x~ in place and the partial sums of Y2^2 (one per workgroup; *nblocks_out of them)
template <typename TS>
static inline hipError_t wafer_launch_x2_apply(const WaferRowArgs &ra, int k, typename WaferF3Store<TS>::S *phi, const WaferX2Ptrs &st, const double *coef,
                                               double *partials, size_t partials_cap, int num_cus, hipStream_t s, int *nblocks_out)
{
    const dim3 grid((unsigned)(num_cus * 8)), block(256);
    if ((size_t)grid.x > partials_cap) return hipErrorInvalidValue;
    *nblocks_out = (int)grid.x;
    switch (k) {
    case 1: hipLaunchKernelGGL((wafer_k_x2_apply<TS, 1>), grid, block, 0, s, ra, phi, st, coef, partials); break;
    case 2: hipLaunchKernelGGL((wafer_k_x2_apply<TS, 2>), grid, block, 0, s, ra, phi, st, coef, partials); break;
    case 3: hipLaunchKernelGGL((wafer_k_x2_apply<TS, 3>), grid, block, 0, s, ra, phi, st, coef, partials); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
