// THREE imaginary-time steps per pass over HBM for the ThreePoint ground-state evolve loop
// (grid.rs:562-686 with wnum == 0: nothing but the stencil happens between steps).
//
//   phi0 --step--> phi1 --step--> phi2 --step--> phi3
//
// The two-step kernel (wafer_stencil_fused2.hip.h) reads phi0 and V once and writes phi2 once per TWO
// updates and runs at the device's traffic ceiling; this one amortises the same three streams over
// THREE updates: 24 B + halo per three updates (8 B + halo per update against 12 B + halo).
//
// A workgroup marches a 128 x 16 tile (fp64; TX = 16 B per lane) along z with THREE pipelines:
// plane z of phi1 is produced from the phi0 register queue, feeds the phi1 queue from which plane
// z-1 of phi2 is produced, which feeds the phi2 queue from which plane z-2 of phi3 is produced and
// stored.  phi1 and phi2 exist only in registers (own z-columns) and in two-slot LDS rings (x / y
// neighbours).  One s_barrier per plane, as in the two-step kernel.
//
// Wave roles (11 waves, RY = 2 rows per lane; OPT bit 7 -- the default -- uses 12: see BAL below):
//   waves 0..7   "main": own rows y0..y0+15 at all three levels;
//   wave  8, 9   "halo-row": rows (y0-2, y0-1) and (y0+16, y0+17): phi1 on both rows, phi2 on the
//                inner one (y0-1 / y0+16);
//   wave  10     "halo-column": each lane keeps up to three phi0 halo-column cells (3 columns per side
//                x 22 rows, z-queues in components of the row-slot registers) and produces phi1 on
//                the inner two columns and phi2 on the innermost one.
// phi0's outermost halo rows (y0-3, y0+18) are plain vector loads staged through LDS.
//
// a and b are formed from V in registers (potential.rs:104-110) at every level -- carrying them from
// level to level as the two-step kernel does would cost the registers the third z-queue needs.
// Per-update arithmetic is the single-step kernel's, so results are bit-identical to three single
// steps (tests/test_gpu_parity.py::test_fused_three_step_kernel_bit_exact).  Cells of phi1 / phi2
// outside the work area (Dirichlet frame, config.rs:597-622) and planes outside the global work range
// are forced to 0 exactly as the reference never updates them.  z-chunks recompute two planes of
// phi1 and one of phi2 on each side; slabs of a sharded grid need 3 valid ghost planes of phi0.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "wafer_geom.h"
#include "wafer_stencil.hip.h"
#include "wafer_stencil_lds.hip.h"
#include "wafer_stencil_fused2.hip.h"

#ifndef WAFER_F3_OPT_DEFAULT
#define WAFER_F3_OPT_DEFAULT 232
#endif

template <typename T, int NWH_ = 2>
struct WaferF3Cfg {
    static constexpr int VEC = WaferVec<T>::N;
    static constexpr int RY = 2;
    static constexpr int NW2 = 8;                       // main waves: tile height 16
    static constexpr int NWH = NWH_;                    // halo-row waves (3: the balanced assignment, OPT bit 7)
    static constexpr int NW = NW2 + NWH + 1;            // + halo-column wave
    static constexpr int NT_ = NW * 64;
    static constexpr int TX = 64 * VEC, TY = NW2 * RY;
    static constexpr int HC0 = 3, HC1 = 2, HC2 = 1;     // halo columns per side of phi0 / phi1 / phi2
    static constexpr int HX0 = ((HC0 + VEC - 1) / VEC) * VEC;
    static constexpr int HX1 = ((HC1 + VEC - 1) / VEC) * VEC;
    static constexpr int HX2 = ((HC2 + VEC - 1) / VEC) * VEC;
    static constexpr int LP0 = TX + 2 * HX0, LP1 = TX + 2 * HX1, LP2 = TX + 2 * HX2;
    static constexpr int ROWS0 = TY + 6, ROWS1 = TY + 4, ROWS2 = TY + 2;
    static constexpr int TILE0 = ROWS0 * LP0, TILE1 = ROWS1 * LP1, TILE2 = ROWS2 * LP2;
    static constexpr int NCOL = 2 * HC0 * ROWS0;        // phi0 halo-column cells per plane
    static constexpr int CPL = (NCOL + 63) / 64;        // cells per lane of the halo-column wave
    static_assert(CPL <= RY * VEC, "halo-column cells per lane must fit the row-slot registers");
};

// OPT (tuning variants kept side by side for A/B runs on one box, WAFER_F3_OPT):
//   bit 0: the outermost phi0 halo rows are fetched by the halo-row waves instead of main waves 0 and 1 (their
//          staging registers leave the main waves' path, where the pressure is);
//   bit 1: b is carried from level 1 to levels 2 and 3 in registers (a = (1 - dt V/2) b formed from it: the same
//          expressions, the same bits) instead of being formed from V three times; the registers come from the
//          phi1 z-queue, whose two older planes are read back from a THREE-slot phi1 LDS ring instead;
//   bit 2: s_setprio: main waves above the halo waves.
//   bit 7 (BAL): the work of the helper waves spread evenly over the four SIMDs.  Waves go to SIMDs round robin, so
//          with 8 main waves (6 row-updates per plane each: 2 rows x 3 levels), two halo-row waves (3 each: phi1 on
//          two rows, phi2 on one) and the halo-column wave (about 1) the SIMDs carry 15 / 15 / 13 / 12 row-updates
//          per plane, and the barrier waits for the fullest.  With THREE halo-row waves -- (y0-1: phi1 + phi2),
//          (y0+16: phi1 + phi2), (y0-2 and y0+17: phi1) -- and the halo-column wave on the fourth SIMD it is
//          14 / 14 / 14 / 13.  Twelve waves are three per SIMD, like eleven: the same 168-VGPR cap.
template <typename T, typename C, bool VIR, int OPT>
__global__ __launch_bounds__((WaferF3Cfg<T, ((OPT & 128) ? 3 : 2)>::NT_)) void wafer_k_step3_fused(WaferStepArgs a, int ntx, int nty, int swz,
                                                                          const T *__restrict__ phi,
                                                                          const T *__restrict__ pv, T *__restrict__ out)
{
    constexpr bool BAL = (OPT & 128) != 0;
    using Cfg = WaferF3Cfg<T, (BAL ? 3 : 2)>;
    using VT = typename WaferVec<T>::type;
    constexpr int R = 1;
    constexpr int VEC = Cfg::VEC, RY = Cfg::RY, TX = Cfg::TX, TY = Cfg::TY;
    constexpr int HX0 = Cfg::HX0, HX1 = Cfg::HX1, HX2 = Cfg::HX2, LP0 = Cfg::LP0, LP1 = Cfg::LP1, LP2 = Cfg::LP2;
    __shared__ __attribute__((aligned(16))) T lds0[2 * Cfg::TILE0];
    constexpr bool OROW_H = (OPT & 1) != 0, CARRY_B = (OPT & 2) != 0, PRIO = (OPT & 4) != 0;
    constexpr bool YREG = (OPT & 32) != 0;   // bit 5: y neighbours inside the lane's own two rows from registers at levels 1 and 2 (as level 3 does)
    constexpr bool NOXMASK = (OPT & 8) != 0; // bit 3: INTERIOR also requires the tile's columns to be work columns: no per-cell x mask
    constexpr int NB1 = CARRY_B ? 3 : 2;                 // phi1 ring slots
    __shared__ __attribute__((aligned(16))) T lds1[NB1 * Cfg::TILE1];
    __shared__ __attribute__((aligned(16))) T lds2[2 * Cfg::TILE2];

    const WaferGeom &g = a.g;
    int bid = blockIdx.x;
    if (swz) {
        const int n = gridDim.x, q = n >> 3, r = n & 7, k = bid & 7;
        bid = k * q + min(k, r) + (bid >> 3);
    }
    int tx_i, ty_i, zs, ze;
    if (a.nsub > 1) { // mixed launch (slab interiors): long workgroups first, the last tiles as short ones (WaferStepArgs)
        int tile, sub = 0;
        if (bid < a.n_long) {
            tile = bid;
        } else {
            tile = a.n_long + (bid - a.n_long) / a.nsub;
            sub = (bid - a.n_long) % a.nsub;
        }
        tx_i = tile % ntx;
        ty_i = tile / ntx;
        zs = bid < a.n_long ? a.lz_lo : a.lz_lo + sub * a.zchunk;
        ze = bid < a.n_long ? a.lz_hi : min(zs + a.zchunk, a.lz_hi);
    } else {
        const int tz_i = bid / (ntx * nty);
        tx_i = bid % ntx;
        ty_i = (bid / ntx) % nty;
        zs = a.lz_lo + tz_i * a.zchunk;
        ze = min(zs + a.zchunk, a.lz_hi);
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // wave-uniform: role tests become scalar branches
    const int x0 = tx_i * TX, y0 = ty_i * TY;
    const C dt = (C)a.dt, den = (C)a.den;
    constexpr bool vir = VIR;
    const bool is_main = wave < Cfg::NW2;
    const bool is_hrow = wave >= Cfg::NW2 && wave < Cfg::NW2 + Cfg::NWH;
    const bool is_hcol = wave == Cfg::NW - 1;
    // bit 2: the main waves (three levels per plane: the critical path to the barrier) issue ahead of the halo waves
    if constexpr (PRIO) {
        if (is_main) __builtin_amdgcn_s_setprio(3);
        else __builtin_amdgcn_s_setprio(0);
    }

    VT zero;
#pragma unroll
    for (int v = 0; v < VEC; ++v) zero[v] = T(0);

    // ---- row slots of the main and halo-row waves (no bounds predicates on loads: whole tiles, three
    //      halo rows / columns and three planes past the slab lie in the zero guard zone, wafer_geom.h)
    const int xl = lane * VEC, xi = x0 + xl;
    int yrow[RY];
    bool rowwk[RY], lvl2[RY], slot_on[RY];
    // bit 6 (SROW): rowoff holds only the WAVE-UNIFORM part of a row's element offset (scalar registers) and the
    // lane adds its 32-bit x offset at the access.  Per-lane 64-bit offsets cost six VGPRs in a kernel that sits at
    // its 168-VGPR cap: the compiler spilled them, and reloading the store addresses from scratch put an
    // s_waitcnt vmcnt(0) -- scratch loads share the counter -- in front of each store of the plane, i.e. a wait
    // for every prefetch in flight and, before the second store, for the first store's write to complete.
    constexpr bool SROW = (OPT & 64) != 0;
    long long rowoff[RY];
    const unsigned xlu = SROW ? (unsigned)(lane * VEC) : 0u;
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        int y;
        bool l2 = true, on = true;
        if (is_hrow && BAL) {
            const int h = wave - Cfg::NW2;   // 0: row y0-1 (phi1, phi2);  1: row y0+16 (phi1, phi2);  2: rows y0-2, y0+17 (phi1)
            if (h == 2) {
                y = (r == 0) ? (y0 - 2) : (y0 + TY + 1);
                l2 = false;
            } else {
                y = (h == 0) ? (y0 - 1) : (y0 + TY);   // the second slot repeats the row (its loads are the same values) and computes nothing
                l2 = on = (r == 0);
            }
        } else if (is_hrow) {
            const int h = wave - Cfg::NW2;                     // 0: rows y0-2, y0-1;  1: rows y0+16, y0+17
            y = (h == 0) ? (y0 - 2 + r) : (y0 + TY + r);
            l2 = (h == 0) ? (r == 1) : (r == 0);               // phi2 on the inner row only
        } else {
            y = y0 + wave * RY + r;                            // main (unused by the halo-column wave)
        }
        yrow[r] = y;
        rowwk[r] = on && (y >= 0) && (y < g.ny);
        lvl2[r] = l2;
        slot_on[r] = on;
        rowoff[r] = (long long)(y + R) * g.pitch + g.xoff + R + (SROW ? x0 : xi);
    }
    // ---- outermost phi0 halo rows y0-3 and y0+18, fetched by main waves 0 and 1
    const bool has_orow = OROW_H ? is_hrow : (is_main && wave < 2);
    const int oy = ((OROW_H ? wave - Cfg::NW2 : wave) == 0) ? (y0 - 3) : (y0 + TY + 2);
    const long long orow_off = (long long)(oy + R) * g.pitch + g.xoff + R + (SROW ? x0 : xi);   // as rowoff
    const int orow_lds = (oy - (y0 - 3)) * LP0 + HX0 + xl;
    // ---- halo-column cells of the last wave: cell c = lane + 64 q: row c / 6 of the phi0 tile, k = c % 6:
    //      k < 3: column x0-1-k, else column x0+TX+(k-3)
    bool c_ok[Cfg::CPL], c_l1[Cfg::CPL], c_l2[Cfg::CPL], c_wk[Cfg::CPL];
    long long c_off[Cfg::CPL];
    int c_lds0[Cfg::CPL], c_lds1[Cfg::CPL], c_lds2[Cfg::CPL];
#pragma unroll
    for (int q = 0; q < Cfg::CPL; ++q) {
        const int cidx = min(lane + q * 64, Cfg::NCOL - 1);    // surplus lanes repeat the last cell
        const int row = cidx / (2 * Cfg::HC0), k = cidx % (2 * Cfg::HC0);
        const int kk = (k < Cfg::HC0) ? k : k - Cfg::HC0;       // distance - 1 from the tile edge
        const int lc = (k < Cfg::HC0) ? (-1 - kk) : (TX + kk);
        const int xw = x0 + lc, y = y0 - 3 + row;
        c_ok[q] = is_hcol && lane + q * 64 < Cfg::NCOL;
        c_wk[q] = (y >= 0) && (y < g.ny) && (xw >= 0) && (xw < g.nx);
        c_l1[q] = c_ok[q] && kk < Cfg::HC1 && row >= 1 && row < Cfg::ROWS0 - 1;   // phi1: inner two columns, rows y0-2 .. y0+17
        c_l2[q] = c_ok[q] && kk < Cfg::HC2 && row >= 2 && row < Cfg::ROWS0 - 2;   // phi2: innermost column, rows y0-1 .. y0+16
        c_off[q] = (long long)(y + R) * g.pitch + g.xoff + R + xw;
        c_lds0[q] = row * LP0 + HX0 + lc;
        c_lds1[q] = (row - 1) * LP1 + HX1 + lc;
        c_lds2[q] = (row - 2) * LP2 + HX2 + lc;
    }

    auto work_plane = [&](int p) {
        const int kg = g.z_begin + (p - g.G);
        return kg >= 0 && kg < g.nz;
    };
    // one update: a, b from V (potential.rs:104-110), then grid.rs:580-589
    auto update = [&](C w, C vv, C S) -> T {
        C ca, cb;
        wafer_ab_from_v<C>(vv, dt, vir, ca, cb);
        return (T)wafer_update<C>(w, ca, cb, dt, S, den);
    };
    // level 1 with CARRY_B: also hands b out; levels 2, 3: b given, a = (1 - dt V / 2) * b (potential.rs:108-110)
    auto update_keep_b = [&](C w, C vv, C S, C &cb) -> T {
        C ca;
        wafer_ab_from_v<C>(vv, dt, vir, ca, cb);
        return (T)wafer_update<C>(w, ca, cb, dt, S, den);
    };
    auto update_with_b = [&](C w, C vv, C cb, C S) -> T {
        const C ca = (C(1) - dt * vv / C(2)) * cb;
        return (T)wafer_update<C>(w, ca, cb, dt, S, den);
    };

    // ---- prologue: first phi1 plane is z1 = zs - 2; the phi0 queue holds planes z1-1 .. z1+1
    const int z1 = zs - 2;
    VT q0[3][RY], q1[3][RY], q2[3][RY];
    VT vq[3][RY];   // V of planes z-2, z-1, z: levels 3, 2 and 1 of one iteration
    VT cbq[CARRY_B ? 2 : 1][RY], cbnew[RY]; // CARRY_B: b of planes z-2, z-1 (levels 3, 2) and of plane z as level 1 forms it
#pragma unroll
    for (int r = 0; r < RY; ++r) cbq[0][r] = cbq[CARRY_B ? 1 : 0][r] = cbnew[r] = zero;
    // (the halo-column wave keeps cell q of its CPL cells in component q % VEC of row slot q / VEC)
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int r = 0; r < RY; ++r) q0[m][r] = q1[m][r] = q2[m][r] = vq[m][r] = zero;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const long long po = (long long)(z1 - 1 + m) * g.plane;
        if (!is_hcol) {
#pragma unroll
            for (int r = 0; r < RY; ++r) q0[m][r] = *reinterpret_cast<const VT *>((phi + po + rowoff[r]) + xlu);
        } else {
#pragma unroll
            for (int q = 0; q < Cfg::CPL; ++q) q0[m][q / VEC][q % VEC] = phi[po + c_off[q]];
        }
    }
    if (!is_hcol) {
#pragma unroll
        for (int r = 0; r < RY; ++r) vq[2][r] = *reinterpret_cast<const VT *>((pv + (long long)z1 * g.plane + rowoff[r]) + xlu);
    } else {
#pragma unroll
        for (int q = 0; q < Cfg::CPL; ++q) vq[2][q / VEC][q % VEC] = pv[(long long)z1 * g.plane + c_off[q]];
    }
    for (int i = tid; i < 2 * Cfg::TILE0; i += Cfg::NT_) lds0[i] = T(0);
    for (int i = tid; i < NB1 * Cfg::TILE1; i += Cfg::NT_) lds1[i] = T(0);
    for (int i = tid; i < 2 * Cfg::TILE2; i += Cfg::NT_) lds2[i] = T(0);
    __syncthreads();
    {
        T *t0 = lds0 + (z1 & 1) * Cfg::TILE0;
        if (!is_hcol) {
#pragma unroll
            for (int r = 0; r < RY; ++r) *reinterpret_cast<VT *>(t0 + (yrow[r] - (y0 - 3)) * LP0 + HX0 + xl) = q0[1][r];
        } else {
#pragma unroll
            for (int q = 0; q < Cfg::CPL; ++q)
                if (c_ok[q]) t0[c_lds0[q]] = q0[1][q / VEC][q % VEC];
        }
        if (has_orow) *reinterpret_cast<VT *>(t0 + orow_lds) = *reinterpret_cast<const VT *>((phi + (long long)z1 * g.plane + orow_off) + xlu);
    }
    VT orow_nxt = zero;
    if (has_orow) orow_nxt = *reinterpret_cast<const VT *>((phi + (long long)(z1 + 1) * g.plane + orow_off) + xlu);
    __syncthreads();

    const int zend = ze + 2; // phi1 planes z1 .. zend-1
    for (int z = z1; z < zend; ++z) {
        const bool more = z + 1 < zend;
        const long long zo = (long long)z * g.plane;
        // ---- 1. prefetch: phi0 plane z+2, V plane z+1, outer halo rows of plane z+2
        VT pre[RY], pre_v[RY], orow_pre = zero;
#pragma unroll
        for (int r = 0; r < RY; ++r) pre[r] = pre_v[r] = zero;
        if (!is_hcol) {
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                pre[r] = *reinterpret_cast<const VT *>((phi + zo + 2 * g.plane + rowoff[r]) + xlu);
                pre_v[r] = *reinterpret_cast<const VT *>((pv + zo + g.plane + rowoff[r]) + xlu);
            }
            if (has_orow) orow_pre = *reinterpret_cast<const VT *>((phi + zo + 2 * g.plane + orow_off) + xlu);
        } else {
#pragma unroll
            for (int q = 0; q < Cfg::CPL; ++q) {
                pre[q / VEC][q % VEC] = phi[zo + 2 * g.plane + c_off[q]];
                pre_v[q / VEC][q % VEC] = pv[zo + g.plane + c_off[q]];
            }
        }
        // ---- 2. stage phi0 plane z+1 into the other buffer
        if (more) {
            T *nt = lds0 + ((z + 1) & 1) * Cfg::TILE0;
            if (!is_hcol) {
#pragma unroll
                for (int r = 0; r < RY; ++r) *reinterpret_cast<VT *>(nt + (yrow[r] - (y0 - 3)) * LP0 + HX0 + xl) = q0[2][r];
            } else {
#pragma unroll
                for (int q = 0; q < Cfg::CPL; ++q)
                    if (c_ok[q]) nt[c_lds0[q]] = q0[2][q / VEC][q % VEC];
            }
            if (has_orow) *reinterpret_cast<VT *>(nt + orow_lds) = orow_nxt;
        }
        const T *c0 = lds0 + (z & 1) * Cfg::TILE0;
        // phi1 ring: plane p lives in slot p mod NB1 (z1 may be negative: + 3 * 2^20 keeps the operand positive)
        const int s1w = CARRY_B ? (z + 3145728) % 3 : (z & 1), s1c = CARRY_B ? (z - 1 + 3145728) % 3 : ((z - 1) & 1);
        T *w1 = lds1 + s1w * Cfg::TILE1;
        const T *c1 = lds1 + s1c * Cfg::TILE1;
        [[maybe_unused]] const T *c1o = lds1 + (CARRY_B ? (z - 2 + 3145728) % 3 : 0) * Cfg::TILE1; // plane z-2 (CARRY_B)
        T *w2 = lds2 + ((z - 1) & 1) * Cfg::TILE2;
        const T *c2 = lds2 + (z & 1) * Cfg::TILE2;          // plane z-2
        const bool wplane1 = work_plane(z), wplane2 = work_plane(z - 1);
        VT p1new[RY], p2new[RY];
#pragma unroll
        for (int r = 0; r < RY; ++r) p1new[r] = p2new[r] = zero;

        if (!is_hcol) {
            bool all_rows = NOXMASK ? (x0 + TX <= g.nx) : true;
#pragma unroll
            for (int r = 0; r < RY; ++r) all_rows = all_rows && rowwk[r];   // (rowwk is false for a slot that is off)
            // ---- 3. level 1: phi1 plane z (main and halo-row waves).  INTERIOR: the plane and every row of this
            //         wave are work cells -- no tests inside, so the RY x VEC updates form one basic block
            // (BAL: the rows of a halo-row wave are not neighbours: their y neighbours come from LDS -- yreg_tag)
            auto level1 = [&](auto interior_tag, auto yreg_tag) {
                constexpr bool INTERIOR = decltype(interior_tag)::value;
                constexpr bool YR = YREG && decltype(yreg_tag)::value;
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    VT res = zero;
                    if (INTERIOR || (wplane1 && rowwk[r])) {
                        const int ly = yrow[r] - (y0 - 3);
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            const C w = (C)q0[1][r][v];
                            C xs[3], ys[3], zz[3];
                            zz[0] = (C)q0[0][r][v]; zz[1] = w; zz[2] = (C)q0[2][r][v];
                            xs[1] = ys[1] = w;
                            xs[0] = (v >= 1) ? (C)q0[1][r][(v + VEC - 1) % VEC] : (C)c0[ly * LP0 + HX0 + xl + v - 1];
                            xs[2] = (v + 1 < VEC) ? (C)q0[1][r][(v + 1) % VEC] : (C)c0[ly * LP0 + HX0 + xl + v + 1];
                            ys[0] = (YR && r >= 1) ? (C)q0[1][r >= 1 ? r - 1 : 0][v] : (C)c0[(ly - 1) * LP0 + HX0 + xl + v];
                            ys[2] = (YR && r + 1 < RY) ? (C)q0[1][r + 1 < RY ? r + 1 : RY - 1][v] : (C)c0[(ly + 1) * LP0 + HX0 + xl + v];
                            const C S = wafer_stencil_sum<C, 1>(xs, ys, zz, w);
                            T rs;
                            if constexpr (CARRY_B) {
                                C cb;
                                rs = update_keep_b(w, (C)vq[2][r][v], S, cb);
                                cbnew[r][v] = (T)cb;
                            } else rs = update(w, (C)vq[2][r][v], S);
                            res[v] = ((NOXMASK && INTERIOR) || xi + v < g.nx) ? rs : T(0);
                        }
                    }
                    p1new[r] = res;
                    if (INTERIOR || !BAL || slot_on[r]) *reinterpret_cast<VT *>(w1 + (yrow[r] - (y0 - 2)) * LP1 + HX1 + xl) = res;
                }
            };
            if (!BAL || is_main) {
                if (all_rows && wplane1) level1(std::true_type{}, std::true_type{});
                else level1(std::false_type{}, std::true_type{});
            } else {
                if (all_rows && wplane1) level1(std::true_type{}, std::false_type{});
                else level1(std::false_type{}, std::false_type{});
            }
            // ---- 4. level 2: phi2 plane z-1 from the phi1 queue; x / y neighbours from the phi1 ring slot written
            //         one iteration ago
            if constexpr (!CARRY_B) {
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    q1[0][r] = q1[1][r];
                    q1[1][r] = q1[2][r];
                    q1[2][r] = p1new[r];
                }
            }
            auto level2 = [&](auto interior_tag, auto yreg_tag) {
                constexpr bool INTERIOR = decltype(interior_tag)::value;
                constexpr bool YR = YREG && decltype(yreg_tag)::value;
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    VT res = zero;
                    if (INTERIOR || (lvl2[r] && wplane2 && rowwk[r])) {
                        const int ly = yrow[r] - (y0 - 2);
                        // the own column of phi1: planes z-2 and z-1 from the register queue, or (CARRY_B) back from the ring
                        VT m0, m1;
                        if constexpr (CARRY_B) {
                            m0 = *reinterpret_cast<const VT *>(c1o + ly * LP1 + HX1 + xl);
                            m1 = *reinterpret_cast<const VT *>(c1 + ly * LP1 + HX1 + xl);
                        } else {
                            m0 = q1[0][r];
                            m1 = q1[1][r];
                        }
                        const VT m2 = CARRY_B ? p1new[r] : q1[2][r];
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            const C w = (C)m1[v];
                            C xs[3], ys[3], zz[3];
                            zz[0] = (C)m0[v]; zz[1] = w; zz[2] = (C)m2[v];
                            xs[1] = ys[1] = w;
                            xs[0] = (v >= 1) ? (C)m1[(v + VEC - 1) % VEC] : (C)c1[ly * LP1 + HX1 + xl + v - 1];
                            xs[2] = (v + 1 < VEC) ? (C)m1[(v + 1) % VEC] : (C)c1[ly * LP1 + HX1 + xl + v + 1];
                            ys[0] = (YR && !CARRY_B && r >= 1) ? (C)q1[1][r >= 1 ? r - 1 : 0][v] : (C)c1[(ly - 1) * LP1 + HX1 + xl + v];
                            ys[2] = (YR && !CARRY_B && r + 1 < RY) ? (C)q1[1][r + 1 < RY ? r + 1 : RY - 1][v] : (C)c1[(ly + 1) * LP1 + HX1 + xl + v];
                            const C S = wafer_stencil_sum<C, 1>(xs, ys, zz, w);
                            T rs;
                            if constexpr (CARRY_B) rs = update_with_b(w, (C)vq[1][r][v], (C)cbq[1][r][v], S);
                            else rs = update(w, (C)vq[1][r][v], S);
                            res[v] = ((NOXMASK && INTERIOR) || xi + v < g.nx) ? rs : T(0);
                        }
                    }
                    p2new[r] = res;
                    if (INTERIOR || lvl2[r]) *reinterpret_cast<VT *>(w2 + (yrow[r] - (y0 - 1)) * LP2 + HX2 + xl) = res;
                }
            };
            if (is_main && all_rows && wplane2) level2(std::true_type{}, std::true_type{});
            else if (!BAL || is_main) level2(std::false_type{}, std::true_type{});
            else level2(std::false_type{}, std::false_type{});
            // ---- 5. level 3 (main waves): phi3 plane z-2 from the phi2 queue, stored
            if (is_main) {
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    q2[0][r] = q2[1][r];
                    q2[1][r] = q2[2][r];
                    q2[2][r] = p2new[r];
                }
                const int zo3 = z - 2;
                if (zo3 >= zs) {
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
                                    zz[0] = (C)q2[0][r][v]; zz[1] = w; zz[2] = (C)q2[2][r][v];
                                    xs[1] = ys[1] = w;
                                    xs[0] = (v >= 1) ? (C)q2[1][r][(v + VEC - 1) % VEC] : (C)c2[ly * LP2 + HX2 + xl + v - 1];
                                    xs[2] = (v + 1 < VEC) ? (C)q2[1][r][(v + 1) % VEC] : (C)c2[ly * LP2 + HX2 + xl + v + 1];
                                    ys[0] = (r >= 1) ? (C)q2[1][r - 1 < 0 ? 0 : r - 1][v] : (C)c2[(ly - 1) * LP2 + HX2 + xl + v];
                                    ys[2] = (r + 1 < RY) ? (C)q2[1][r + 1 < RY ? r + 1 : RY - 1][v] : (C)c2[(ly + 1) * LP2 + HX2 + xl + v];
                                    const C S = wafer_stencil_sum<C, 1>(xs, ys, zz, w);
                                    if constexpr (CARRY_B) res3[r][v] = update_with_b(w, (C)vq[0][r][v], (C)cbq[0][r][v], S);
                                    else res3[r][v] = update(w, (C)vq[0][r][v], S);
                                }
                            }
                        }
#pragma unroll
                        for (int r = 0; r < RY; ++r) {
                            if (INTERIOR || rowwk[r]) {
                                T *dst = (out + (long long)zo3 * g.plane + rowoff[r]) + xlu;
                                if ((NOXMASK && INTERIOR) || xi + VEC <= g.nx) {
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
            }
        } else {
            // ---- halo-column wave: phi1 on the inner two columns, phi2 on the innermost one
#pragma unroll
            for (int q = 0; q < Cfg::CPL; ++q) {
                T rs = T(0);
                if (c_l1[q]) {
                    if (wplane1 && c_wk[q]) {
                        const int o0 = c_lds0[q];
                        const C w = (C)q0[1][q / VEC][q % VEC];
                        C xs[3], ys[3], zz[3];
                        zz[0] = (C)q0[0][q / VEC][q % VEC]; zz[1] = w; zz[2] = (C)q0[2][q / VEC][q % VEC];
                        xs[1] = ys[1] = w;
                        xs[0] = (C)c0[o0 - 1]; xs[2] = (C)c0[o0 + 1];
                        ys[0] = (C)c0[o0 - LP0]; ys[2] = (C)c0[o0 + LP0];
                        const C S = wafer_stencil_sum<C, 1>(xs, ys, zz, w);
                        if constexpr (CARRY_B) {
                            C cb;
                            rs = update_keep_b(w, (C)vq[2][q / VEC][q % VEC], S, cb);
                            cbnew[q / VEC][q % VEC] = (T)cb;
                        } else rs = update(w, (C)vq[2][q / VEC][q % VEC], S);
                    }
                    w1[c_lds1[q]] = rs;
                }
                p1new[q / VEC][q % VEC] = rs;
            }
            if constexpr (!CARRY_B) {
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    q1[0][r] = q1[1][r];
                    q1[1][r] = q1[2][r];
                    q1[2][r] = p1new[r];
                }
            }
#pragma unroll
            for (int q = 0; q < Cfg::CPL; ++q) {
                if (c_l2[q]) {
                    T rs = T(0);
                    if (wplane2 && c_wk[q]) {
                        const int o1 = c_lds1[q];
                        const C w = CARRY_B ? (C)c1[o1] : (C)q1[1][q / VEC][q % VEC];
                        C xs[3], ys[3], zz[3];
                        zz[0] = CARRY_B ? (C)c1o[o1] : (C)q1[0][q / VEC][q % VEC]; zz[1] = w;
                        zz[2] = CARRY_B ? (C)p1new[q / VEC][q % VEC] : (C)q1[2][q / VEC][q % VEC];
                        xs[1] = ys[1] = w;
                        xs[0] = (C)c1[o1 - 1]; xs[2] = (C)c1[o1 + 1];
                        ys[0] = (C)c1[o1 - LP1]; ys[2] = (C)c1[o1 + LP1];
                        const C S = wafer_stencil_sum<C, 1>(xs, ys, zz, w);
                        if constexpr (CARRY_B) rs = update_with_b(w, (C)vq[1][q / VEC][q % VEC], (C)cbq[1][q / VEC][q % VEC], S);
                        else rs = update(w, (C)vq[1][q / VEC][q % VEC], S);
                    }
                    w2[c_lds2[q]] = rs;
                }
            }
        }
        __syncthreads();
        // ---- 6. rotate the phi0 / V pipelines
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            q0[0][r] = q0[1][r];
            q0[1][r] = q0[2][r];
            q0[2][r] = pre[r];
            vq[0][r] = vq[1][r];
            vq[1][r] = vq[2][r];
            vq[2][r] = pre_v[r];
            if constexpr (CARRY_B) {
                cbq[0][r] = cbq[1][r];
                cbq[1][r] = cbnew[r];
            }
        }
        orow_nxt = orow_pre;
    }
}

// Advances planes [lz_lo, lz_hi) by THREE steps: out = step(step(step(phi))).  ThreePoint only.
template <typename T, typename C>
static inline hipError_t wafer_launch_step3_fused(WaferStepArgs a, const T *phi, const T *pv, T *out, hipStream_t s)
{
    using Cfg = WaferF3Cfg<T>;
    const WaferLdsOpts o = wafer_lds_opts();
    const WaferGeom &g = a.g;
    const int ntx = (g.nx + Cfg::TX - 1) / Cfg::TX;
    const int nty = (g.ny + Cfg::TY - 1) / Cfg::TY;
    const int nplanes = a.lz_hi - a.lz_lo;
    int swz = o.swz;
    long long nblocks = 0;
    if (a.nsub > 1 && nplanes < 8 * a.nsub) a.nsub = 0; // too thin to cut
    if (a.nsub > 1) { // the interior launch of a slab: see wafer_launch_step2_fused_nw
        const int ntiles = ntx * nty;
        const int nshort_tiles = ntiles / 16 > 0 ? ntiles / 16 : 1;
        a.n_long = ntiles - nshort_tiles;
        a.zchunk = (nplanes + a.nsub - 1) / a.nsub;
        nblocks = a.n_long + (long long)nshort_tiles * a.nsub;
        swz = 0; // the hardware's dispatch order is the point
    } else { // planes per workgroup: one workgroup per CU marching a long column (as the two-step kernel)
        const char *f = getenv("WAFER_ZCHUNK");
        if (f && atoi(f) > 0) {
            a.zchunk = atoi(f);
        } else if (a.target_blocks < 0) {
            a.zchunk = -a.target_blocks < nplanes ? -a.target_blocks : nplanes;
        } else {
            const long long per_layer = (long long)ntx * nty;
            const char *t = getenv("WAFER_TARGET_BLOCKS");
            const long long target = (t && atoi(t) > 0) ? atoi(t) : (a.target_blocks > 0 ? a.target_blocks : 256);
            long long nch = (target + per_layer / 2) / per_layer;
            if (nch < 1) nch = 1;
            if (nch > nplanes) nch = nplanes;
            a.zchunk = (int)((nplanes + nch - 1) / nch);
        }
        nblocks = (long long)ntx * nty * ((nplanes + a.zchunk - 1) / a.zchunk);
    }
    const dim3 grid((unsigned)nblocks);
    const char *eo = getenv("WAFER_F3_OPT");
    const int opt = (eo && *eo) ? atoi(eo) : WAFER_F3_OPT_DEFAULT;
#define WAFER_F3_CASE(VIR_, OPT_)                                                                                          \
    if ((a.v_in_range != 0) == VIR_ && opt == OPT_) {                                                                      \
        const dim3 block(WaferF3Cfg<T, ((OPT_ & 128) ? 3 : 2)>::NT_);                                                      \
        hipLaunchKernelGGL((wafer_k_step3_fused<T, C, VIR_, OPT_>), grid, block, (size_t)o.pad, s, a, ntx, nty, swz, phi, pv, out); \
        return hipGetLastError();                                                                                          \
    }
    WAFER_F3_CASE(true, 0)
    WAFER_F3_CASE(true, 1)
    WAFER_F3_CASE(true, 3)
    WAFER_F3_CASE(true, 4)
    WAFER_F3_CASE(true, 8)
    WAFER_F3_CASE(true, 40)
    WAFER_F3_CASE(true, 104)
    WAFER_F3_CASE(true, 232)
    WAFER_F3_CASE(false, 0)
    WAFER_F3_CASE(false, 1)
    WAFER_F3_CASE(false, 3)
    WAFER_F3_CASE(false, 4)
    WAFER_F3_CASE(false, 8)
    WAFER_F3_CASE(false, 40)
    WAFER_F3_CASE(false, 104)
    WAFER_F3_CASE(false, 232)
#undef WAFER_F3_CASE
    return hipErrorInvalidValue;
}
