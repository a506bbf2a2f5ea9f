// The three-step kernel on EIGHT waves with a, b carried from level to level.
//
// wafer_stencil_fused3.hip.h (twelve waves: eight main + four helpers, three per SIMD, 168-VGPR cap) forms a and b from V
// three times per cell and pass -- once per level -- because the registers that would carry them do not exist there: the
// reciprocal sequence is 12 of the 29 fp64 instructions of an update, and the kernel runs at ~70 % of the VALU issue rate.
// Here the workgroup has no helper waves: two waves per SIMD, 256 VGPRs each.  Every wave owns two rows of the 128 x 16 tile
// at all three levels (as before) plus ONE extra slot:
//   wave 0   row y0-1   (phi1 and phi2)          wave 7   row y0+16  (phi1 and phi2)
//   wave 1   row y0-2   (phi1), stages row y0-3  wave 6   row y0+17  (phi1), stages row y0+18
//   waves 2..5   33 of the 132 phi0 halo-column cells each, one per lane: phi1 on the inner two columns, phi2 on the innermost
// and a, b of a cell are formed ONCE, at level 1, and ride in registers to levels 2 and 3 (the same values: a, b of a cell do
// not depend on the level): 29 + 17 + 17 instead of 3 x 29 fp64 instructions per cell and pass.
// Tile, LDS rings, workgroup table (WaferF3Block), flags / counters of the single-launch slab pass, both marching directions
// and every per-update expression are those of wafer_stencil_fused3.hip.h: the results are bit-identical.
#pragma once
#include "wafer_stencil_fused3.hip.h"

template <typename T>
struct WaferF3cCfg {
    static constexpr int NW = 8, NT_ = NW * 64;
    static constexpr int HCW0 = 2, HCWN = 4;                      // halo-column cells: waves 2..5
    static constexpr int CPW = (WaferF3Cfg<T>::NCOL + HCWN - 1) / HCWN; // cells per such wave (33 <= 64 lanes)
    static_assert(CPW <= 64, "one halo-column cell per lane");
};

template <typename T, typename C, bool VIR, bool DOWN>
__device__ __forceinline__ void wafer_step3c_body(const WaferStepArgs &a, const WaferF3Block &blk, int ntx, const WaferF3Sync &sy,
                                                  const T *__restrict__ phi, const T *__restrict__ pv, T *__restrict__ out,
                                                  T *lds0, T *lds1, T *lds2)
{
    using Cfg = WaferF3Cfg<T>;
    using Cc = WaferF3cCfg<T>;
    using VT = typename WaferVec<T>::type;
    constexpr int R = 1;
    constexpr int VEC = Cfg::VEC, RY = Cfg::RY, TX = Cfg::TX, TY = Cfg::TY;
    constexpr int HX0 = Cfg::HX0, HX1 = Cfg::HX1, HX2 = Cfg::HX2, LP0 = Cfg::LP0, LP1 = Cfg::LP1, LP2 = Cfg::LP2;
    constexpr int SD = DOWN ? -1 : 1;
    constexpr int ZLO = DOWN ? 2 : 0, ZHI = DOWN ? 0 : 2;

    const WaferGeom &g = a.g;
    const int tx_i = blk.tile % ntx, ty_i = blk.tile / ntx;
    const int zs = blk.zs, ze = blk.ze;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = tx_i * TX, y0 = ty_i * TY;
    const C dt = (C)a.dt, den = (C)a.den;
    constexpr bool vir = VIR;
    // the extra slot: a halo row (waves 0, 1, 6, 7) or halo-column cells (waves 2..5)
    const bool x_row = wave < 2 || wave >= 6;
    const bool x_l2 = wave == 0 || wave == 7;            // the halo row next to the tile: phi2 as well
    const bool has_orow = wave == 1 || wave == 6;

    VT zero;
#pragma unroll
    for (int v = 0; v < VEC; ++v) zero[v] = T(0);
    const int xl = lane * VEC, xi = x0 + xl;
    const unsigned xlu = (unsigned)(lane * VEC);

    // ---- main rows
    int yrow[RY];
    bool rowwk[RY];
    long long rowoff[RY];
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        const int y = y0 + wave * RY + r;
        yrow[r] = y;
        rowwk[r] = y < g.ny;
        rowoff[r] = (long long)(y + R) * g.pitch + g.xoff + R + x0;
    }
    // ---- the extra halo row
    const int xy = wave == 0 ? y0 - 1 : wave == 1 ? y0 - 2 : wave == 6 ? y0 + TY + 1 : y0 + TY;
    const bool xwk = x_row && xy >= 0 && xy < g.ny;
    const long long xoff_row = (long long)(xy + R) * g.pitch + g.xoff + R + x0;
    // ---- outermost phi0 halo rows y0-3 / y0+18 (plain vector loads staged through LDS)
    const int oy = wave == 1 ? y0 - 3 : y0 + TY + 2;
    const long long orow_off = (long long)(oy + R) * g.pitch + g.xoff + R + x0;
    const int orow_lds = (oy - (y0 - 3)) * LP0 + HX0 + xl;
    // ---- halo-column cell of this lane (waves 2..5): cell c: row c / 6 of the phi0 tile, k = c % 6: k < 3: column x0-1-k,
    //      else column x0+TX+(k-3)
    const int cidx = min((wave - Cc::HCW0) * Cc::CPW + lane, Cfg::NCOL - 1);
    const int crow = cidx / (2 * Cfg::HC0), ck = cidx % (2 * Cfg::HC0);
    const int ckk = (ck < Cfg::HC0) ? ck : ck - Cfg::HC0;
    const int clc = (ck < Cfg::HC0) ? (-1 - ckk) : (TX + ckk);
    const int cxw = x0 + clc, cy = y0 - 3 + crow;
    const bool c_ok = !x_row && lane < Cc::CPW && (wave - Cc::HCW0) * Cc::CPW + lane < Cfg::NCOL;
    const bool c_wk = cy >= 0 && cy < g.ny && cxw >= 0 && cxw < g.nx;
    const bool c_l1 = c_ok && ckk < Cfg::HC1 && crow >= 1 && crow < Cfg::ROWS0 - 1;
    const bool c_l2 = c_ok && ckk < Cfg::HC2 && crow >= 2 && crow < Cfg::ROWS0 - 2;
    const long long c_off = (long long)(cy + R) * g.pitch + g.xoff + R + cxw;
    const int c_lds0 = crow * LP0 + HX0 + clc, c_lds1 = (crow - 1) * LP1 + HX1 + clc, c_lds2 = (crow - 2) * LP2 + HX2 + clc;

    auto work_plane = [&](int p) {
        const int kg = g.z_begin + (p - g.G);
        return kg >= 0 && kg < g.nz;
    };
    // level 1: a, b from V (potential.rs:104-110); what rides to levels 2 and 3 is a and the product b * dt -- b enters the
    // update (grid.rs:580-589: w * a + b * dt * S / den, left to right) only through that product, which is the same number
    // at every level
    auto update_keep = [&](C w, C vv, C S, C &ca, C &cbdt) -> T {
        C cb;
        wafer_ab_from_v<C>(vv, dt, vir, ca, cb);
        cbdt = cb * dt;
        return (T)(w * ca + wafer_div_invariant<C>(cbdt * S, den));
    };
    auto update_with = [&](C w, C ca, C cbdt, C S) -> T { return (T)(w * ca + wafer_div_invariant<C>(cbdt * S, den)); };

    // ---- state.  Main rows: three z-queues, V of the level-1 plane, a / b of the planes of levels 2 and 3.
    //      Extra slot (component 0 only for a halo-column cell): phi0 and phi1 queues, V, a / b of the level-2 plane.
    const int z1 = DOWN ? ze + 1 : zs - 2;
    VT q0[3][RY], q1[3][RY], q2[3][RY], vcur[RY], caq[2][RY], cbq[2][RY];
    VT xq0[3], xq1[3], xv, xca, xcb;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
#pragma unroll
        for (int r = 0; r < RY; ++r) q0[m][r] = q1[m][r] = q2[m][r] = zero;
        xq0[m] = xq1[m] = zero;
    }
#pragma unroll
    for (int r = 0; r < RY; ++r) vcur[r] = caq[0][r] = caq[1][r] = cbq[0][r] = cbq[1][r] = zero;
    xv = xca = xcb = zero;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const long long po = (long long)(z1 + SD * (m - 1)) * g.plane;
#pragma unroll
        for (int r = 0; r < RY; ++r) q0[m][r] = *reinterpret_cast<const VT *>((phi + po + rowoff[r]) + xlu);
        if (x_row) xq0[m] = *reinterpret_cast<const VT *>((phi + po + xoff_row) + xlu);
        else xq0[m][0] = phi[po + c_off];
    }
    {
        const long long po = (long long)z1 * g.plane;
#pragma unroll
        for (int r = 0; r < RY; ++r) vcur[r] = *reinterpret_cast<const VT *>((pv + po + rowoff[r]) + xlu);
        if (x_row) xv = *reinterpret_cast<const VT *>((pv + po + xoff_row) + xlu);
        else xv[0] = pv[po + c_off];
    }
    for (int i = tid; i < 2 * Cfg::TILE0; i += Cc::NT_) lds0[i] = T(0);
    for (int i = tid; i < 2 * Cfg::TILE1; i += Cc::NT_) lds1[i] = T(0);
    for (int i = tid; i < 2 * Cfg::TILE2; i += Cc::NT_) lds2[i] = T(0);
    __syncthreads();
    {
        T *t0 = lds0 + (z1 & 1) * Cfg::TILE0;
#pragma unroll
        for (int r = 0; r < RY; ++r) *reinterpret_cast<VT *>(t0 + (yrow[r] - (y0 - 3)) * LP0 + HX0 + xl) = q0[1][r];
        if (x_row) *reinterpret_cast<VT *>(t0 + (xy - (y0 - 3)) * LP0 + HX0 + xl) = xq0[1];
        else if (c_ok) t0[c_lds0] = xq0[1][0];
        if (has_orow) *reinterpret_cast<VT *>(t0 + orow_lds) = *reinterpret_cast<const VT *>((phi + (long long)z1 * g.plane + orow_off) + xlu);
    }
    VT orow_nxt = zero;
    if (has_orow) orow_nxt = *reinterpret_cast<const VT *>((phi + (long long)(z1 + SD) * g.plane + orow_off) + xlu);
    __syncthreads();

    const int niter = (ze - zs) + 4;
    for (int it = 0; it < niter; ++it) {
        const int z = z1 + SD * it;
        const bool more = it + 1 < niter;
        const long long zo = (long long)z * g.plane;
        if (blk.wait_late >= 0 && it == blk.wait_it) wafer_f3_wait(sy, blk.wait_late, tid);
        // ---- 1. prefetch: phi0 two planes ahead, V one plane ahead
        VT pre[RY], pre_v[RY], xpre = zero, xpre_v = zero, orow_pre = zero;
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            pre[r] = *reinterpret_cast<const VT *>((phi + zo + SD * 2 * g.plane + rowoff[r]) + xlu);
            pre_v[r] = *reinterpret_cast<const VT *>((pv + zo + SD * g.plane + rowoff[r]) + xlu);
        }
        if (x_row) {
            xpre = *reinterpret_cast<const VT *>((phi + zo + SD * 2 * g.plane + xoff_row) + xlu);
            xpre_v = *reinterpret_cast<const VT *>((pv + zo + SD * g.plane + xoff_row) + xlu);
            if (has_orow) orow_pre = *reinterpret_cast<const VT *>((phi + zo + SD * 2 * g.plane + orow_off) + xlu);
        } else {
            xpre[0] = phi[zo + SD * 2 * g.plane + c_off];
            xpre_v[0] = pv[zo + SD * g.plane + c_off];
        }
        // ---- 2. stage the next phi0 plane into the other buffer
        if (more) {
            T *nt = lds0 + ((z + 1) & 1) * Cfg::TILE0;
#pragma unroll
            for (int r = 0; r < RY; ++r) *reinterpret_cast<VT *>(nt + (yrow[r] - (y0 - 3)) * LP0 + HX0 + xl) = q0[2][r];
            if (x_row) {
                *reinterpret_cast<VT *>(nt + (xy - (y0 - 3)) * LP0 + HX0 + xl) = xq0[2];
                if (has_orow) *reinterpret_cast<VT *>(nt + orow_lds) = orow_nxt;
            } else if (c_ok) nt[c_lds0] = xq0[2][0];
        }
        const T *c0 = lds0 + (z & 1) * Cfg::TILE0;
        T *w1 = lds1 + (z & 1) * Cfg::TILE1;
        const T *c1 = lds1 + ((z + 1) & 1) * Cfg::TILE1;
        T *w2 = lds2 + ((z + 1) & 1) * Cfg::TILE2;
        const T *c2 = lds2 + (z & 1) * Cfg::TILE2;
        const bool wplane1 = work_plane(z), wplane2 = work_plane(z - SD);
        const int zp2 = z - SD;
        const bool need2 = zp2 >= zs - 1 && zp2 <= ze;   // phi2 is read on planes zs-1 .. ze only
        VT p1new[RY], p2new[RY], canew[RY], cbnew[RY];
#pragma unroll
        for (int r = 0; r < RY; ++r) p1new[r] = p2new[r] = canew[r] = cbnew[r] = zero;
        VT xp1 = zero, xcanew = zero, xcbnew = zero;

        bool all_rows = x0 + TX <= g.nx;
#pragma unroll
        for (int r = 0; r < RY; ++r) all_rows = all_rows && rowwk[r];
        // ---- 3. level 1, main rows.  INTERIOR: the plane and both rows are work cells, the tile's columns too: no tests
        //         inside, the RY x VEC updates form one basic block
        auto level1 = [&](auto interior_tag) {
            constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                VT res = zero;
                if (INTERIOR || (wplane1 && rowwk[r])) {
                    const int ly = yrow[r] - (y0 - 3);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const C w = (C)q0[1][r][v];
                        C xs[3], ys[3], zz[3];
                        zz[0] = (C)q0[ZLO][r][v]; zz[1] = w; zz[2] = (C)q0[ZHI][r][v];
                        xs[1] = ys[1] = w;
                        xs[0] = (v >= 1) ? (C)q0[1][r][(v + VEC - 1) % VEC] : (C)c0[ly * LP0 + HX0 + xl + v - 1];
                        xs[2] = (v + 1 < VEC) ? (C)q0[1][r][(v + 1) % VEC] : (C)c0[ly * LP0 + HX0 + xl + v + 1];
                        ys[0] = (r >= 1) ? (C)q0[1][r >= 1 ? r - 1 : 0][v] : (C)c0[(ly - 1) * LP0 + HX0 + xl + v];
                        ys[2] = (r + 1 < RY) ? (C)q0[1][r + 1 < RY ? r + 1 : RY - 1][v] : (C)c0[(ly + 1) * LP0 + HX0 + xl + v];
                        const C S = wafer_stencil_sum<C, 1>(xs, ys, zz, w);
                        C ka, kb;
                        const T rs = update_keep(w, (C)vcur[r][v], S, ka, kb);
                        canew[r][v] = (T)ka;
                        cbnew[r][v] = (T)kb;
                        res[v] = (INTERIOR || xi + v < g.nx) ? rs : T(0);
                    }
                }
                p1new[r] = res;
                *reinterpret_cast<VT *>(w1 + (yrow[r] - (y0 - 2)) * LP1 + HX1 + xl) = res;
            }
        };
        if (all_rows && wplane1) level1(std::true_type{});
        else level1(std::false_type{});
        // ---- 3x. level 1, the extra slot
        if (x_row) {
            VT res = zero;
            if (wplane1 && xwk) {
                const int ly = xy - (y0 - 3);
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const C w = (C)xq0[1][v];
                    C xs[3], ys[3], zz[3];
                    zz[0] = (C)xq0[ZLO][v]; zz[1] = w; zz[2] = (C)xq0[ZHI][v];
                    xs[1] = ys[1] = w;
                    xs[0] = (v >= 1) ? (C)xq0[1][(v + VEC - 1) % VEC] : (C)c0[ly * LP0 + HX0 + xl + v - 1];
                    xs[2] = (v + 1 < VEC) ? (C)xq0[1][(v + 1) % VEC] : (C)c0[ly * LP0 + HX0 + xl + v + 1];
                    ys[0] = (C)c0[(ly - 1) * LP0 + HX0 + xl + v];
                    ys[2] = (C)c0[(ly + 1) * LP0 + HX0 + xl + v];
                    const C S = wafer_stencil_sum<C, 1>(xs, ys, zz, w);
                    C ka, kb;
                    const T rs = update_keep(w, (C)xv[v], S, ka, kb);
                    xcanew[v] = (T)ka;
                    xcbnew[v] = (T)kb;
                    res[v] = (xi + v < g.nx) ? rs : T(0);
                }
            }
            xp1 = res;
            *reinterpret_cast<VT *>(w1 + (xy - (y0 - 2)) * LP1 + HX1 + xl) = res;
        } else if (c_l1) {
            T rs = T(0);
            if (wplane1 && c_wk) {
                const C w = (C)xq0[1][0];
                C xs[3], ys[3], zz[3];
                zz[0] = (C)xq0[ZLO][0]; zz[1] = w; zz[2] = (C)xq0[ZHI][0];
                xs[1] = ys[1] = w;
                xs[0] = (C)c0[c_lds0 - 1]; xs[2] = (C)c0[c_lds0 + 1];
                ys[0] = (C)c0[c_lds0 - LP0]; ys[2] = (C)c0[c_lds0 + LP0];
                const C S = wafer_stencil_sum<C, 1>(xs, ys, zz, w);
                C ka, kb;
                rs = update_keep(w, (C)xv[0], S, ka, kb);
                xcanew[0] = (T)ka;
                xcbnew[0] = (T)kb;
            }
            w1[c_lds1] = rs;
            xp1[0] = rs;
        }
        // ---- 4. level 2: phi2 of the plane behind, from the phi1 queues; a, b as level 1 formed them one iteration ago
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            q1[0][r] = q1[1][r];
            q1[1][r] = q1[2][r];
            q1[2][r] = p1new[r];
        }
        xq1[0] = xq1[1];
        xq1[1] = xq1[2];
        xq1[2] = xp1;
        if (need2) {
            auto level2 = [&](auto interior_tag) {
                constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    VT res = zero;
                    if (INTERIOR || (wplane2 && rowwk[r])) {
                        const int ly = yrow[r] - (y0 - 2);
                        const VT m1 = q1[1][r];
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            const C w = (C)m1[v];
                            C xs[3], ys[3], zz[3];
                            zz[0] = (C)q1[ZLO][r][v]; zz[1] = w; zz[2] = (C)q1[ZHI][r][v];
                            xs[1] = ys[1] = w;
                            xs[0] = (v >= 1) ? (C)m1[(v + VEC - 1) % VEC] : (C)c1[ly * LP1 + HX1 + xl + v - 1];
                            xs[2] = (v + 1 < VEC) ? (C)m1[(v + 1) % VEC] : (C)c1[ly * LP1 + HX1 + xl + v + 1];
                            ys[0] = (r >= 1) ? (C)q1[1][r >= 1 ? r - 1 : 0][v] : (C)c1[(ly - 1) * LP1 + HX1 + xl + v];
                            ys[2] = (r + 1 < RY) ? (C)q1[1][r + 1 < RY ? r + 1 : RY - 1][v] : (C)c1[(ly + 1) * LP1 + HX1 + xl + v];
                            const C S = wafer_stencil_sum<C, 1>(xs, ys, zz, w);
                            const T rs = update_with(w, (C)caq[1][r][v], (C)cbq[1][r][v], S);
                            res[v] = (INTERIOR || xi + v < g.nx) ? rs : T(0);
                        }
                    }
                    p2new[r] = res;
                    *reinterpret_cast<VT *>(w2 + (yrow[r] - (y0 - 1)) * LP2 + HX2 + xl) = res;
                }
            };
            if (all_rows && wplane2) level2(std::true_type{});
            else level2(std::false_type{});
            if (x_row) {
                if (x_l2) {
                    VT res = zero;
                    if (wplane2 && xwk) {
                        const int ly = xy - (y0 - 2);
                        const VT m1 = xq1[1];
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            const C w = (C)m1[v];
                            C xs[3], ys[3], zz[3];
                            zz[0] = (C)xq1[ZLO][v]; zz[1] = w; zz[2] = (C)xq1[ZHI][v];
                            xs[1] = ys[1] = w;
                            xs[0] = (v >= 1) ? (C)m1[(v + VEC - 1) % VEC] : (C)c1[ly * LP1 + HX1 + xl + v - 1];
                            xs[2] = (v + 1 < VEC) ? (C)m1[(v + 1) % VEC] : (C)c1[ly * LP1 + HX1 + xl + v + 1];
                            ys[0] = (C)c1[(ly - 1) * LP1 + HX1 + xl + v];
                            ys[2] = (C)c1[(ly + 1) * LP1 + HX1 + xl + v];
                            const C S = wafer_stencil_sum<C, 1>(xs, ys, zz, w);
                            const T rs = update_with(w, (C)xca[v], (C)xcb[v], S);
                            res[v] = (xi + v < g.nx) ? rs : T(0);
                        }
                    }
                    *reinterpret_cast<VT *>(w2 + (xy - (y0 - 1)) * LP2 + HX2 + xl) = res;
                }
            } else if (c_l2) {
                T rs = T(0);
                if (wplane2 && c_wk) {
                    const C w = (C)xq1[1][0];
                    C xs[3], ys[3], zz[3];
                    zz[0] = (C)xq1[ZLO][0]; zz[1] = w; zz[2] = (C)xq1[ZHI][0];
                    xs[1] = ys[1] = w;
                    xs[0] = (C)c1[c_lds1 - 1]; xs[2] = (C)c1[c_lds1 + 1];
                    ys[0] = (C)c1[c_lds1 - LP1]; ys[2] = (C)c1[c_lds1 + LP1];
                    const C S = wafer_stencil_sum<C, 1>(xs, ys, zz, w);
                    rs = update_with(w, (C)xca[0], (C)xcb[0], S);
                }
                w2[c_lds2] = rs;
            }
        }
        // ---- 5. level 3: phi3 two planes behind from the phi2 queue, a, b as formed two iterations ago; stored
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            q2[0][r] = q2[1][r];
            q2[1][r] = q2[2][r];
            q2[2][r] = p2new[r];
        }
        const int zo3 = z - 2 * SD;
        const bool wthrough = blk.bump >= 0 && (DOWN ? zo3 < zs + blk.wt : zo3 >= ze - blk.wt);
        if (zo3 >= zs && zo3 < ze) {
            auto level3 = [&](auto interior_tag) {
                constexpr bool INTERIOR = decltype(interior_tag)::value;
                VT res3[RY];
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    res3[r] = zero;
                    if (INTERIOR || rowwk[r]) {
                        const int ly = yrow[r] - (y0 - 1);
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            const C w = (C)q2[1][r][v];
                            C xs[3], ys[3], zz[3];
                            zz[0] = (C)q2[ZLO][r][v]; zz[1] = w; zz[2] = (C)q2[ZHI][r][v];
                            xs[1] = ys[1] = w;
                            xs[0] = (v >= 1) ? (C)q2[1][r][(v + VEC - 1) % VEC] : (C)c2[ly * LP2 + HX2 + xl + v - 1];
                            xs[2] = (v + 1 < VEC) ? (C)q2[1][r][(v + 1) % VEC] : (C)c2[ly * LP2 + HX2 + xl + v + 1];
                            ys[0] = (r >= 1) ? (C)q2[1][r - 1 < 0 ? 0 : r - 1][v] : (C)c2[(ly - 1) * LP2 + HX2 + xl + v];
                            ys[2] = (r + 1 < RY) ? (C)q2[1][r + 1 < RY ? r + 1 : RY - 1][v] : (C)c2[(ly + 1) * LP2 + HX2 + xl + v];
                            const C S = wafer_stencil_sum<C, 1>(xs, ys, zz, w);
                            res3[r][v] = update_with(w, (C)caq[0][r][v], (C)cbq[0][r][v], S);
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    if (INTERIOR || rowwk[r]) {
                        T *dst = (out + (long long)zo3 * g.plane + rowoff[r]) + xlu;
                        if (wthrough) {
#pragma unroll
                            for (int v = 0; v < VEC; ++v)
                                if (INTERIOR || xi + v < g.nx) __hip_atomic_store(dst + v, res3[r][v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        } else if (INTERIOR || xi + VEC <= g.nx) {
                            *reinterpret_cast<VT *>(dst) = res3[r];
                        } else {
#pragma unroll
                            for (int v = 0; v < VEC; ++v)
                                if (xi + v < g.nx) dst[v] = res3[r][v];
                        }
                    }
                }
            };
            if (all_rows) level3(std::true_type{});
            else level3(std::false_type{});
        }
        __syncthreads();
        // ---- 6. rotate the phi0 / V / a, b pipelines
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            q0[0][r] = q0[1][r];
            q0[1][r] = q0[2][r];
            q0[2][r] = pre[r];
            vcur[r] = pre_v[r];
            caq[0][r] = caq[1][r];
            cbq[0][r] = cbq[1][r];
            caq[1][r] = canew[r];
            cbq[1][r] = cbnew[r];
        }
        xq0[0] = xq0[1];
        xq0[1] = xq0[2];
        xq0[2] = xpre;
        xv = xpre_v;
        xca = xcanew;
        xcb = xcbnew;
        orow_nxt = orow_pre;
    }
    if (blk.bump >= 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(sy.cnt + blk.bump * WAFER_F3_SYNC_STRIDE, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <typename T, typename C, bool VIR>
__global__ __launch_bounds__((WaferF3cCfg<T>::NT_)) void wafer_k_step3_fused_c(WaferStepArgs a, int ntx, const WaferF3Block *__restrict__ table,
                                                                              WaferF3Sync sy, const T *__restrict__ phi,
                                                                              const T *__restrict__ pv, T *__restrict__ out)
{
    using Cfg = WaferF3Cfg<T>;
    __shared__ __attribute__((aligned(16))) T lds0[2 * Cfg::TILE0];
    __shared__ __attribute__((aligned(16))) T lds1[2 * Cfg::TILE1];
    __shared__ __attribute__((aligned(16))) T lds2[2 * Cfg::TILE2];
    const WaferF3Block blk = table[blockIdx.x];
    if (blk.down) wafer_step3c_body<T, C, VIR, true>(a, blk, ntx, sy, phi, pv, out, lds0, lds1, lds2);
    else wafer_step3c_body<T, C, VIR, false>(a, blk, ntx, sy, phi, pv, out, lds0, lds1, lds2);
}

template <typename T, typename C>
static inline hipError_t wafer_launch_step3_fused_c(const WaferTuning &t, const WaferStepArgs &a, const WaferF3Block *table, int nblocks,
                                                    const WaferF3Sync &sy, const T *phi, const T *pv, T *out, hipStream_t s)
{
    using Cfg = WaferF3Cfg<T>;
    const int ntx = (a.g.nx + Cfg::TX - 1) / Cfg::TX;
    const dim3 grid((unsigned)nblocks), block(WaferF3cCfg<T>::NT_);
    if (a.v_in_range != 0)
        hipLaunchKernelGGL((wafer_k_step3_fused_c<T, C, true>), grid, block, (size_t)t.lds_pad, s, a, ntx, table, sy, phi, pv, out);
    else
        hipLaunchKernelGGL((wafer_k_step3_fused_c<T, C, false>), grid, block, (size_t)t.lds_pad, s, a, ntx, table, sy, phi, pv, out);
    return hipGetLastError();
}
