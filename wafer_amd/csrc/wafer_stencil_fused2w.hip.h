// TWO imaginary-time steps per pass for the FivePoint stencil (grid.rs:593-624 twice; ext = 2), in the three-step kernel's
// structure (wafer_stencil_fused3.hip.h) instead of the two-step kernel's dedicated helper waves (wafer_stencil_fused2.hip.h).
//
//   phi0 --step--> phi1 --step--> phi2
//
// The two-step kernel serves FivePoint with 4 main waves on a 128 x 8 tile + 2 halo-row waves + 1 halo-column wave: phi1 on 12 rows
// per 8 stored, phi0 on 16, seven waves of which three idle through the second step -- 0.43 ms/step at 512^3, 4.0 TB/s of traffic,
// half of the device's rate.  Here a workgroup is EIGHT waves on a 128 x 16 tile (fp64; 16 B of x per lane) and every wave owns two
// rows of the tile at both levels plus ONE extra slot:
//   wave 0   phi1 of row y0-1,  stages phi0 of row y0-3        wave 7   phi1 of row y0+16, stages row y0+18
//   wave 1   phi1 of row y0-2,  stages phi0 of row y0-4        wave 6   phi1 of row y0+17, stages row y0+19
//   waves 2..5   48 of the 192 phi0 halo-column cells each (4 columns per side x 24 rows), one per lane; phi1 on the inner two
//                columns of rows y0-2 .. y0+17
// so phi1 is computed on 20 rows per 16 stored and phi0 read on 24, every global access stays 128-byte aligned and the waves'
// work is even.  z-queues in registers (five planes of phi0 and of phi1 per row slot), x / y neighbours through LDS: phi0 in two
// buffers (the plane of level 1 and the one being staged), phi1 in a ring of three planes (level 2 reads the plane written two
// iterations earlier).  One s_barrier per plane.
//
// a and b of a cell (potential.rs:104-110) are formed from V at both levels (V of planes z-2, z-1 waits in a two-plane queue):
// carrying them as a and b * dt needs 32 more registers than the kernel has -- measured with the phi1 z-queue moved into a deeper
// LDS ring to make room: 0.377 against 0.378 ms/step, the kernel is not bound by its instruction count
// (profiles/r05_ab_fivepoint_carry.jsonl).  Requests are spread over the iteration and a wave's issue priority falls as it
// advances (the three-step kernel's round-4 findings).  The first template argument is the storage tag of
// wafer_storage.h: fp64, fp32 storage with fp64 arithmetic (float in HBM, everything in the CU double, each level rounded to
// float), or all-fp32 (256 x 16 tiles).  Per-update arithmetic is the single-step kernel's: bit-identical to two single steps.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "wafer_geom.h"
#include "wafer_stencil.hip.h"
#include "wafer_stencil_lds.hip.h"
#include "wafer_stencil_fused2.hip.h"
#include "wafer_storage.h"

#define WAFER_W2_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
#ifndef WAFER_DIAG
#define WAFER_DIAG 0
#endif

template <typename T>
struct WaferW2Cfg {
    static constexpr int R = 2;
    static constexpr int VEC = WaferVec<T>::N;
    static constexpr int RY = 2, NW = 8, NT_ = NW * 64;
    static constexpr int TX = 64 * VEC, TY = NW * RY;
    static constexpr int HC0 = 2 * R, HC1 = R;            // halo columns per side of phi0 / phi1
    static constexpr int HX0 = ((HC0 + VEC - 1) / VEC) * VEC, HX1 = ((HC1 + VEC - 1) / VEC) * VEC;
    static constexpr int LP0 = TX + 2 * HX0, LP1 = TX + 2 * HX1;
    static constexpr int ROWS0 = TY + 4 * R, ROWS1 = TY + 2 * R;
    static constexpr int TILE0 = ROWS0 * LP0, TILE1 = ROWS1 * LP1;
    static constexpr int NB1 = R + 1;                      // planes in the phi1 ring
    static constexpr int NCOL = 2 * HC0 * ROWS0;           // phi0 halo-column cells per plane: 192
    static constexpr int HCW0 = 2, HCWN = 4, CPW = (NCOL + HCWN - 1) / HCWN;   // waves 2..5, 48 cells each, one per lane
    static_assert(CPW <= 64, "one halo-column cell per lane");
};

template <typename TS, typename C, bool VIR>
__global__ __launch_bounds__((WaferW2Cfg<typename WaferF3Store<TS>::Q>::NT_)) void wafer_k_step2_wide(
    WaferStepArgs a, int ntx, int nty, int swz, const typename WaferF3Store<TS>::S *__restrict__ phi,
    const typename WaferF3Store<TS>::S *__restrict__ pv, typename WaferF3Store<TS>::S *__restrict__ out)
{
    using T = typename WaferF3Store<TS>::Q;    // z-queues and LDS
    using ST = typename WaferF3Store<TS>::S;   // the arrays in HBM
    using Cfg = WaferW2Cfg<T>;
    using VT = typename WaferVec<T>::type;
    constexpr int R = 2, VEC = Cfg::VEC, RY = Cfg::RY, TX = Cfg::TX, TY = Cfg::TY;
    constexpr int HX0 = Cfg::HX0, HX1 = Cfg::HX1, LP0 = Cfg::LP0, LP1 = Cfg::LP1;
    typedef ST __attribute__((ext_vector_type(VEC))) SVT;   // a lane's request: the same cells in the storage type
    typedef T __attribute__((ext_vector_type(2))) T2;       // the two cells beyond either end of a lane's cells, from LDS
    __shared__ __attribute__((aligned(16))) T lds0[2 * Cfg::TILE0];
    __shared__ __attribute__((aligned(16))) T lds1[Cfg::NB1 * Cfg::TILE1];
    auto gload_raw = [](const ST *p) -> SVT { return *reinterpret_cast<const SVT *>(p); };
    auto widen = [](const SVT &x) -> VT { return wafer_f3_widen<SVT, VT, VEC>(x); };
    auto gload = [&](const ST *p) -> VT { return widen(gload_raw(p)); };
    // a level's result as the storage type holds it (fp32 storage: rounded once per step, like a store and a load would)
    auto as_stored = [](C x) -> T { return (T)(ST)x; };

    const WaferGeom &g = a.g;
    int bid = blockIdx.x + a.block0;
    if (swz) {   // XCD-contiguous tile ranges (workgroup b runs on XCD b % 8; a round of a longer schedule starts at a multiple of 8)
        const int n = a.nblocks_all > 0 ? a.nblocks_all : (int)gridDim.x, q = n >> 3, r = n & 7, k = bid & 7;
        bid = k * q + min(k, r) + (bid >> 3);
    }
    const int tz_i = bid / (ntx * nty);
    const int tx_i = bid % ntx, ty_i = (bid / ntx) % nty;
    const int zs = a.lz_lo + tz_i * a.zchunk, ze = min(zs + a.zchunk, a.lz_hi);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = tx_i * TX, y0 = ty_i * TY;
    const C dt = (C)a.dt;
    constexpr bool vir = VIR;
    const WaferDen<C> den = wafer_den<C>(a, vir);
    const bool x_row = wave < 2 || wave >= 6;   // the extra slot is a halo row (else: halo-column cells)

    VT zero;
#pragma unroll
    for (int v = 0; v < VEC; ++v) zero[v] = T(0);
    SVT szero;
#pragma unroll
    for (int v = 0; v < VEC; ++v) szero[v] = ST(0);
    const int xl = lane * VEC, xi = x0 + xl;

    // ---- main rows
    int yrow[RY];
    bool rowwk[RY];
    long long rowoff[RY];
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        const int y = y0 + wave * RY + r;
        yrow[r] = y;
        rowwk[r] = y < g.ny;
        rowoff[r] = (long long)(y + R) * g.pitch + g.xoff + R + xi;
    }
    // ---- the extra halo row (phi1 as well) and the outer row this wave stages (phi0 only)
    const int xy = wave == 0 ? y0 - 1 : wave == 1 ? y0 - 2 : wave == 6 ? y0 + TY + 1 : y0 + TY;
    const int oy = wave == 0 ? y0 - 3 : wave == 1 ? y0 - 4 : wave == 6 ? y0 + TY + 3 : y0 + TY + 2;
    const bool xwk = x_row && xy >= 0 && xy < g.ny;
    // (a row above / below the work area -- frame and guard rows, zeros that no kernel writes -- is not fetched: the wave asks for
    //  its own first row again and takes the zero the row stands for)
    const bool xy_out = xy < 0 || xy >= g.ny, oy_out = oy < 0 || oy >= g.ny;
    const long long xoff_row = xy_out ? rowoff[0] : (long long)(xy + R) * g.pitch + g.xoff + R + xi;
    const long long orow_off = oy_out ? rowoff[0] : (long long)(oy + R) * g.pitch + g.xoff + R + xi;
    const int xrow_lds0 = (xy - (y0 - 2 * R)) * LP0 + HX0 + xl, orow_lds = (oy - (y0 - 2 * R)) * LP0 + HX0 + xl;
    // ---- the halo-column cell of this lane (waves 2..5): cell c: row c / 8 of the phi0 tile, k = c % 8: k < 4: column x0-1-k,
    //      else column x0+TX+(k-4)
    const int cidx = min((wave - Cfg::HCW0) * Cfg::CPW + lane, Cfg::NCOL - 1);
    const int crow = cidx / (2 * Cfg::HC0), ck = cidx % (2 * Cfg::HC0);
    const int ckk = ck < Cfg::HC0 ? ck : ck - Cfg::HC0;
    const int clc = ck < Cfg::HC0 ? -1 - ckk : TX + ckk;
    const int cxw = x0 + clc, cy = y0 - 2 * R + crow;
    const bool c_ok = !x_row && lane < Cfg::CPW && (wave - Cfg::HCW0) * Cfg::CPW + lane < Cfg::NCOL;
    const bool c_wk = cy >= 0 && cy < g.ny && cxw >= 0 && cxw < g.nx;
    const bool c_l1 = c_ok && ckk < Cfg::HC1 && crow >= R && crow < Cfg::ROWS0 - R;
    // (a cell left / right of the work area or above / below it: not fetched -- its 128-byte line holds nothing anybody else
    //  reads -- the lane asks for the tile's own edge cell of that row and takes a zero)
    const bool c_xout = cxw < 0 || cxw >= g.nx || cy < 0 || cy >= g.ny;
    const long long c_off = (long long)((cy < 0 ? y0 : cy >= g.ny ? y0 + TY - 1 : cy) + R) * g.pitch + g.xoff + R +
                            ((cxw < 0 || cxw >= g.nx) ? (ck < Cfg::HC0 ? x0 : x0 + TX - 1) : cxw);
    const int c_lds0 = crow * LP0 + HX0 + clc, c_lds1 = (crow - R) * LP1 + HX1 + clc;
    // the extra slot's requests are the SAME instructions in every wave, the address chosen per lane (a halo row's vector, or
    // the vector that starts at the lane's halo-column cell: component 0 is the cell)
    const long long xslot_off = x_row ? xoff_row : c_off;
    const long long oslot_off = x_row ? orow_off : c_off;

    auto work_plane = [&](int p) {
        const int kg = g.z_begin + (p - g.G);
        return kg >= 0 && kg < g.nz;
    };
    auto update_keep = [&](C w, C vv, C S, C &ca, C &cbdt) -> T {
        C cb;
        wafer_ab_from_v<C>(vv, dt, vir, ca, cb);
        cbdt = cb * dt;
        return as_stored(w * ca + wafer_div_invariant<C>(cbdt * S, den));
    };

    // ---- state.  Main rows: phi0 planes z-2 .. z+2, phi1 planes z-4 .. z-1, V of planes z-2 .. z.
    //      Extra slot (component 0 only for a halo-column cell): phi0 planes z-2 .. z+2, V of plane z, the outer row of plane z+1.
    const int z1 = zs - R;   // the first phi1 plane
    VT q0[2 * R + 1][RY], q1[2 * R][RY], vcur[RY], vq[R][RY];
    VT xq0[2 * R + 1], xv, orow_nxt = zero;
#pragma unroll
    for (int m = 0; m <= 2 * R; ++m) {
        const long long po = (long long)(z1 - R + m) * g.plane;
#pragma unroll
        for (int r = 0; r < RY; ++r) q0[m][r] = gload(phi + po + rowoff[r]);
        if (x_row) xq0[m] = gload(phi + po + xoff_row);
        else {
            xq0[m] = zero;
            xq0[m][0] = (T)phi[po + c_off];
        }
    }
#pragma unroll
    for (int m = 0; m < 2 * R; ++m)
#pragma unroll
        for (int r = 0; r < RY; ++r) q1[m][r] = zero;
#pragma unroll
    for (int m = 0; m < R; ++m)
#pragma unroll
        for (int r = 0; r < RY; ++r) vq[m][r] = zero;
    {
        const long long po = (long long)z1 * g.plane;
#pragma unroll
        for (int r = 0; r < RY; ++r) vcur[r] = gload(pv + po + rowoff[r]);
        if (x_row) xv = gload(pv + po + xoff_row);
        else {
            xv = zero;
            xv[0] = (T)pv[po + c_off];
        }
    }
    for (int i = tid; i < 2 * Cfg::TILE0; i += Cfg::NT_) lds0[i] = T(0);
    for (int i = tid; i < Cfg::NB1 * Cfg::TILE1; i += Cfg::NT_) lds1[i] = T(0);
    __syncthreads();
    {
        T *t0 = lds0 + (z1 & 1) * Cfg::TILE0;
#pragma unroll
        for (int r = 0; r < RY; ++r) *reinterpret_cast<VT *>(t0 + (yrow[r] - (y0 - 2 * R)) * LP0 + HX0 + xl) = q0[R][r];
        if (x_row) {
            *reinterpret_cast<VT *>(t0 + xrow_lds0) = xy_out ? zero : xq0[R];
            *reinterpret_cast<VT *>(t0 + orow_lds) = oy_out ? zero : gload(phi + (long long)z1 * g.plane + orow_off);
            orow_nxt = gload(phi + (long long)(z1 + 1) * g.plane + orow_off);
        } else if (c_ok) t0[c_lds0] = c_xout ? T(0) : xq0[R][0];
    }
    __syncthreads();

    const int zend = ze + R;   // phi1 planes z1 .. zend - 1
    for (int z = z1; z < zend; ++z) {
        const bool more = z + 1 < zend;
#if WAFER_DIAG & 2   // timing experiment (wafer_stencil_fused3.hip.h): every prefetch asks for the column's first planes again (cache hits)
        const long long zo = (long long)(z1 + (z & 1)) * g.plane;
#else
        const long long zo = (long long)z * g.plane;
#endif
        // ---- 1. prefetch: phi0 three planes ahead, V one plane ahead, the outer row two planes ahead -- not issued together:
        //         the main rows' phi0 and V at the top, the extra slot's three behind level 1 of the main rows.  Everything
        //         requested in an iteration is consumed at its END (the rotation behind the barrier), so a request placed late in
        //         the iteration has a fraction of an iteration to come back: with V behind level 1 and the extra slot's behind ITS
        //         level 1 -- the three-step kernel's placement, where those values are consumed an iteration later -- this kernel
        //         waited for memory behind every barrier and reached 0.88 of the copy rate: 0.364 -> 0.3455 ms/step (all seven
        //         at the top: 256 VGPRs and scratch, 0.361; profiles/r05_ab_fivepoint_request_placement.jsonl)
        SVT pre[RY], pre_v[RY], xpre = szero, xpre_v = szero, orow_pre = szero;
#pragma unroll
        for (int r = 0; r < RY; ++r) pre[r] = pre_v[r] = szero;
        WAFER_W2_SETPRIO(3);
#pragma unroll
        for (int r = 0; r < RY; ++r) pre[r] = gload_raw(phi + zo + (long long)(R + 1) * g.plane + rowoff[r]);
        // (all-fp32, 256 x 16 tiles, 16 B of floats per lane: the early placement costs that instantiation 32 B of scratch; it keeps
        //  V behind level 1 and the extra slot's three behind its level 1)
        constexpr bool EARLY = sizeof(C) == 8;
        auto issue_v = [&]() {
#pragma unroll
            for (int r = 0; r < RY; ++r) pre_v[r] = gload_raw(pv + zo + g.plane + rowoff[r]);
        };
        auto issue_x = [&]() {
            xpre = gload_raw(phi + zo + (long long)(R + 1) * g.plane + xslot_off);
            xpre_v = gload_raw(pv + zo + g.plane + xslot_off);
            orow_pre = gload_raw(phi + zo + 2 * g.plane + oslot_off);
        };
        if constexpr (EARLY) issue_v();
        // ---- 2. stage phi0 plane z+1 into the other buffer (what was requested in place of a cell outside the work area becomes
        //         the zero it stands for here)
        if (more) {
            T *nt = lds0 + ((z + 1) & 1) * Cfg::TILE0;
#pragma unroll
            for (int r = 0; r < RY; ++r) *reinterpret_cast<VT *>(nt + (yrow[r] - (y0 - 2 * R)) * LP0 + HX0 + xl) = q0[R + 1][r];
            if (x_row) {
                *reinterpret_cast<VT *>(nt + xrow_lds0) = xy_out ? zero : xq0[R + 1];
                *reinterpret_cast<VT *>(nt + orow_lds) = oy_out ? zero : orow_nxt;
            } else if (c_ok) nt[c_lds0] = c_xout ? T(0) : xq0[R + 1][0];
        }
        const T *c0 = lds0 + (z & 1) * Cfg::TILE0;
        T *w1 = lds1 + (((z % Cfg::NB1) + Cfg::NB1) % Cfg::NB1) * Cfg::TILE1;
        const int zo2 = z - R;   // the phi2 plane of this iteration
        const T *c1 = lds1 + (((zo2 % Cfg::NB1) + Cfg::NB1) % Cfg::NB1) * Cfg::TILE1;
        const bool wplane1 = work_plane(z);
        const bool do2 = zo2 >= zs;   // (zo2 < ze always: z < ze + R)
        // ---- 2b. the x / y neighbours of the main rows: level 1's here, level 2's (phi1 of plane z-2, written two barriers ago)
        //          behind level 1's arithmetic and ahead of the extra slot's: per level the two cells left of the lane's first and
        //          right of its last on each row, the two rows above the first row and the two below the last
        T2 nbl[2][RY], nbr[2][RY];
        VT nbu[2][2], nbd[2][2];
        auto nbload = [&](auto level_tag) {
            constexpr int L = decltype(level_tag)::value;
            const T *cc = L == 0 ? c0 : c1;
            constexpr int lp = L == 0 ? LP0 : LP1, hx = L == 0 ? HX0 : HX1;
            const int yb = y0 - (2 - L) * R;
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                const int o = (yrow[r] - yb) * lp + hx + xl;
                nbl[L][r] = *reinterpret_cast<const T2 *>(cc + o - 2);
                nbr[L][r] = *reinterpret_cast<const T2 *>(cc + o + VEC);
            }
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                nbu[L][d] = *reinterpret_cast<const VT *>(cc + (yrow[0] - yb - 2 + d) * lp + hx + xl);        // rows y-2, y-1 of the first row
                nbd[L][d] = *reinterpret_cast<const VT *>(cc + (yrow[RY - 1] - yb + 1 + d) * lp + hx + xl);   // rows y+1, y+2 of the last row
            }
        };
        nbload(std::integral_constant<int, 0>{});
        // the 2R+1 values along x and y of cell (r, v) at level L from the centre plane `ctr` of its z-queue and the neighbours above
        auto gather = [&](auto level_tag, const VT (&ctr)[RY], int r, int v, C *xs, C *ys) {
            constexpr int L = decltype(level_tag)::value;
#pragma unroll
            for (int d = -R; d <= R; ++d) {
                if (d == 0) continue;
                const int vv = v + d, rr = r + d;
                xs[d + R] = (vv >= 0 && vv < VEC) ? (C)ctr[r][(vv + VEC) % VEC] : vv < 0 ? (C)nbl[L][r][(vv + 2) & 1] : (C)nbr[L][r][(vv - VEC) & 1];
                ys[d + R] = (rr >= 0 && rr < RY) ? (C)ctr[(rr + RY) % RY][v] : rr < 0 ? (C)nbu[L][(rr + 2) & 1][v] : (C)nbd[L][(rr - RY) & 1][v];
            }
        };
        // ---- 3. level 1, main rows.  INTERIOR: the plane, both rows and the tile's columns are work cells: no tests inside, the
        //         RY x VEC updates form one basic block
        VT p1new[RY];
#pragma unroll
        for (int r = 0; r < RY; ++r) p1new[r] = zero;
        bool all_rows = x0 + TX <= g.nx;
#pragma unroll
        for (int r = 0; r < RY; ++r) all_rows = all_rows && rowwk[r];
        auto level1 = [&](auto interior_tag) {
            constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                VT res = zero;
                if (INTERIOR || (wplane1 && rowwk[r])) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const C w = (C)q0[R][r][v];
                        C xs[2 * R + 1], ys[2 * R + 1], zz[2 * R + 1];
#pragma unroll
                        for (int m = 0; m <= 2 * R; ++m) zz[m] = (C)q0[m][r][v];
                        xs[R] = ys[R] = w;
                        gather(std::integral_constant<int, 0>{}, q0[R], r, v, xs, ys);
                        const C S = wafer_stencil_sum<C, R>(xs, ys, zz, w);
                        C ka, kb;
                        const T rs = update_keep(w, (C)vcur[r][v], S, ka, kb);
                        res[v] = (INTERIOR || xi + v < g.nx) ? rs : T(0);
                    }
                }
                p1new[r] = res;
                *reinterpret_cast<VT *>(w1 + (yrow[r] - (y0 - R)) * LP1 + HX1 + xl) = res;
            }
        };
        if (all_rows && wplane1) level1(std::true_type{});
        else level1(std::false_type{});
        WAFER_W2_SETPRIO(2);
        if constexpr (EARLY) issue_x();
        else issue_v();
        if (do2) nbload(std::integral_constant<int, 1>{});   // (behind level 1: both levels' neighbours at once do not fit the registers)
        // ---- 3x. level 1, the extra slot: every neighbour from LDS, the z-column from its own queue
        if (x_row) {
            VT res = zero;
            if (wplane1 && xwk) {
                const int o = xrow_lds0;
                const T2 l2 = *reinterpret_cast<const T2 *>(c0 + o - 2), r2 = *reinterpret_cast<const T2 *>(c0 + o + VEC);
                VT yn[2 * R + 1];
#pragma unroll
                for (int d = -R; d <= R; ++d)
                    if (d != 0) yn[d + R] = *reinterpret_cast<const VT *>(c0 + o + d * LP0);
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const C w = (C)xq0[R][v];
                    C xs[2 * R + 1], ys[2 * R + 1], zz[2 * R + 1];
#pragma unroll
                    for (int m = 0; m <= 2 * R; ++m) zz[m] = (C)xq0[m][v];
                    xs[R] = ys[R] = w;
#pragma unroll
                    for (int d = -R; d <= R; ++d) {
                        if (d == 0) continue;
                        const int vv = v + d;
                        xs[d + R] = (vv >= 0 && vv < VEC) ? (C)xq0[R][(vv + VEC) % VEC] : vv < 0 ? (C)l2[(vv + 2) & 1] : (C)r2[(vv - VEC) & 1];
                        ys[d + R] = (C)yn[d + R][v];
                    }
                    const C S = wafer_stencil_sum<C, R>(xs, ys, zz, w);
                    C ka, kb;
                    const T rs = update_keep(w, (C)xv[v], S, ka, kb);
                    res[v] = (xi + v < g.nx) ? rs : T(0);
                }
            }
            *reinterpret_cast<VT *>(w1 + (xy - (y0 - R)) * LP1 + HX1 + xl) = res;
        } else if (c_l1) {
            T rs = T(0);
            if (wplane1 && c_wk) {
                const C w = (C)xq0[R][0];
                C xs[2 * R + 1], ys[2 * R + 1], zz[2 * R + 1];
#pragma unroll
                for (int m = 0; m <= 2 * R; ++m) zz[m] = (C)xq0[m][0];
#pragma unroll
                for (int d = -R; d <= R; ++d) {
                    xs[d + R] = d == 0 ? w : (C)c0[c_lds0 + d];
                    ys[d + R] = d == 0 ? w : (C)c0[c_lds0 + d * LP0];
                }
                const C S = wafer_stencil_sum<C, R>(xs, ys, zz, w);
                C ka, kb;
                rs = update_keep(w, (C)xv[0], S, ka, kb);
            }
            w1[c_lds1] = rs;
        }
        WAFER_W2_SETPRIO(1);
        if constexpr (!EARLY) issue_x();
        // ---- 4. level 2: phi2 of plane z-2 from the phi1 queue (planes z-4 .. z-1 and the plane just made), a and b from V of
        //         that plane; stored
        if (do2) {
            auto level2 = [&](auto interior_tag) {
                constexpr bool INTERIOR = decltype(interior_tag)::value;
                VT res2[RY];
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    res2[r] = zero;
                    if (INTERIOR || rowwk[r]) {
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            const C w = (C)q1[R][r][v];
                            C xs[2 * R + 1], ys[2 * R + 1], zz[2 * R + 1];
#pragma unroll
                            for (int m = 0; m < 2 * R; ++m) zz[m] = (C)q1[m][r][v];
                            zz[2 * R] = (C)p1new[r][v];
                            xs[R] = ys[R] = w;
                            gather(std::integral_constant<int, 1>{}, q1[R], r, v, xs, ys);
                            const C S = wafer_stencil_sum<C, R>(xs, ys, zz, w);
                            C ka, kb;
                            res2[r][v] = update_keep(w, (C)vq[0][r][v], S, ka, kb);
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    if (INTERIOR || rowwk[r]) {
#if WAFER_DIAG & 4   // timing experiment: nothing is stored, the results stay live
#pragma unroll
                        for (int v = 0; v < VEC; ++v) asm volatile("" ::"v"(res2[r][v]));
                        continue;
#endif
                        ST *dst = out + (long long)zo2 * g.plane + rowoff[r];
                        SVT st;   // (the value is a storage-type number already: as_stored)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) st[v] = (ST)res2[r][v];
                        if (INTERIOR || xi + VEC <= g.nx) wafer_store_result(reinterpret_cast<SVT *>(dst), st);   // (streamed: wafer_stencil_fused3.hip.h, gstore)
                        else {
#pragma unroll
                            for (int v = 0; v < VEC; ++v)
                                if (xi + v < g.nx) dst[v] = st[v];
                        }
                    }
                }
            };
            if (all_rows) level2(std::true_type{});
            else level2(std::false_type{});
        }
        WAFER_W2_SETPRIO(0);
        __syncthreads();
        // ---- 5. rotate.  The prefetched values are pinned HERE, behind the barrier, so that no wave waits for its requests
        //         before it has reached the barrier
        auto pin = [](auto &x) { asm volatile("" : "+v"(x)); };
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            pin(pre[r]);
            pin(pre_v[r]);
        }
        pin(xpre);
        pin(xpre_v);
        pin(orow_pre);
#pragma unroll
        for (int r = 0; r < RY; ++r) {
#pragma unroll
            for (int m = 0; m < 2 * R; ++m) q0[m][r] = q0[m + 1][r];
            q0[2 * R][r] = widen(pre[r]);
#pragma unroll
            for (int m = 0; m + 1 < 2 * R; ++m) q1[m][r] = q1[m + 1][r];
            q1[2 * R - 1][r] = p1new[r];
#pragma unroll
            for (int m = 0; m + 1 < R; ++m) vq[m][r] = vq[m + 1][r];
            vq[R - 1][r] = vcur[r];
            vcur[r] = widen(pre_v[r]);
        }
#pragma unroll
        for (int m = 0; m < 2 * R; ++m) xq0[m] = xq0[m + 1];
        xq0[2 * R] = widen(xpre);
        xv = widen(xpre_v);
        orow_nxt = widen(orow_pre);
    }
}

// planes per workgroup: one workgroup per CU marching a long column (wafer_pick_zchunk)
template <typename T>
static inline int wafer_w2_zchunk(const WaferTuning &t, const WaferGeom &g, int nplanes, int target_blocks)
{
    using Cfg = WaferW2Cfg<T>;
    if (t.zchunk > 0) return t.zchunk;
    if (target_blocks < 0) return -target_blocks < nplanes ? -target_blocks : nplanes;   // the caller fixed the chunk length
    const long long per_layer = (long long)((g.nx + Cfg::TX - 1) / Cfg::TX) * ((g.ny + Cfg::TY - 1) / Cfg::TY);
    const long long target = t.target_blocks > 0 ? t.target_blocks : (target_blocks > 0 ? target_blocks : 256);
    return wafer_pick_zchunk(per_layer, nplanes, target, 2 * 2 + 3);
}

// Advances planes [lz_lo, lz_hi) by TWO FivePoint steps: out = step(step(phi)).
template <typename TS, typename C>
static inline hipError_t wafer_launch_step2_wide(const WaferTuning &t, WaferStepArgs a, const typename WaferF3Store<TS>::S *phi,
                                                 const typename WaferF3Store<TS>::S *pv, typename WaferF3Store<TS>::S *out, hipStream_t s)
{
    using T = typename WaferF3Store<TS>::Q;
    using Cfg = WaferW2Cfg<T>;
    const WaferGeom &g = a.g;
    const int ntx = (g.nx + Cfg::TX - 1) / Cfg::TX, nty = (g.ny + Cfg::TY - 1) / Cfg::TY;
    // (the mixed launch of slab interiors -- long columns and a few short ones -- is the two-step kernel's; here every column
    //  is cut alike: nsub > 1 asks for that many pieces)
    const int nplanes = a.lz_hi - a.lz_lo;
    const long long slots = t.target_blocks > 0 ? t.target_blocks : (a.target_blocks > 0 ? a.target_blocks : 256);
    // (long columns on several whole rounds of CUs: cut to 384 planes at most and launched round by round -- wafer_f3_by_rounds)
    const bool rounds = !(a.nsub > 1) && a.target_blocks >= 0 && slots % 8 == 0 && wafer_f3_by_rounds(t, (long long)ntx * nty, nplanes, slots);
    if (a.nsub > 1 && nplanes >= 8 * a.nsub) a.zchunk = (nplanes + a.nsub - 1) / a.nsub;
    else if (rounds) a.zchunk = wafer_pick_zchunk((long long)ntx * nty, nplanes, slots, 2 * 2 + 3, 384);
    else a.zchunk = wafer_w2_zchunk<T>(t, g, nplanes, a.target_blocks);
    const long long nblocks = (long long)ntx * nty * ((nplanes + a.zchunk - 1) / a.zchunk);
    const long long per_launch = rounds ? slots : nblocks;
    const dim3 block(Cfg::NT_);
    a.nblocks_all = (int)nblocks;
    for (long long first = 0; first < nblocks; first += per_launch) {
        const dim3 grid((unsigned)std::min(per_launch, nblocks - first));
        a.block0 = (int)first;
        if (a.v_in_range != 0) hipLaunchKernelGGL((wafer_k_step2_wide<TS, C, true>), grid, block, 0, s, a, ntx, nty, t.swz, phi, pv, out);
        else hipLaunchKernelGGL((wafer_k_step2_wide<TS, C, false>), grid, block, 0, s, a, ntx, nty, t.swz, phi, pv, out);
    }
    return hipGetLastError();
}
