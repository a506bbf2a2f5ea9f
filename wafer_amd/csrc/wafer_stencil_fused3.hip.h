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
// plane z of phi1 is produced from the phi0 register queue, feeds the phi1 queue from which the plane
// before it of phi2 is produced, which feeds the phi2 queue from which the plane before that of phi3 is
// produced and stored.  phi1 and phi2 exist only in registers (own z-columns) and in two-slot LDS rings
// (x / y neighbours).  One s_barrier per plane.
//
// Wave roles (12 waves, RY = 2 row slots per lane, three per SIMD; the helper waves are spread so that the four
// SIMDs carry 14 / 14 / 14 / 13 row-updates per plane):
//   waves 0..7   "main": own rows y0..y0+15 at all three levels;
//   wave  8      "halo-row": row y0-1  (phi1, phi2);
//   wave  9      "halo-row": row y0+16 (phi1, phi2);
//   wave  10     "halo-row": rows y0-2 and y0+17 (phi1);
//   wave  11     "halo-column": each lane keeps up to three phi0 halo-column cells (3 columns per side
//                x 22 rows, z-queues in components of the row-slot registers) and produces phi1 on
//                the inner two columns and phi2 on the innermost one.
// phi0's outermost halo rows (y0-3, y0+18) are plain vector loads by main waves 0 and 1, staged through LDS.
//
// a and b are formed from V in registers (potential.rs:104-110) at every level -- carrying them from
// level to level as the two-step kernel does would cost the registers the third z-queue needs.
// Per-update arithmetic is the single-step kernel's, so results are bit-identical to three single
// steps (tests/test_gpu_parity.py::test_fused_three_step_kernel_bit_exact).  Cells of phi1 / phi2
// outside the work area (Dirichlet frame, config.rs:597-622) and planes outside the global work range
// are forced to 0 exactly as the reference never updates them.  z-chunks recompute two planes of
// phi1 and one of phi2 on each side; slabs of a sharded grid need 3 valid ghost planes of phi0.
//
// WHAT a workgroup does comes from a table (WaferF3Block, built by the host once per launch shape):
// tile, plane range, marching direction, and -- for the single-launch pass of a z-slab (wafer_engine.hip,
// overlap mode 2) -- which ghost-plane flag to wait for before the first load that touches ghost planes and which
// completion counter to bump after its last store.  The schedule (XCD-contiguous tile ranges, long and short
// columns, the two halves of a slab in either order) is therefore host code; the kernel has one code path per
// marching direction.  Marching DOWN mirrors the pipeline in z and keeps every sum's operand order
// (grid.rs:582-588: ... + z[+1] + z[-1] ...), so the bits do not depend on the direction.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "wafer_geom.h"
#include "wafer_stencil.hip.h"
#include "wafer_stencil_lds.hip.h"
#include "wafer_stencil_fused2.hip.h"

template <typename T>
struct WaferF3Cfg {
    static constexpr int VEC = WaferVec<T>::N;
    static constexpr int RY = 2;
    static constexpr int NW2 = 8;                       // main waves: tile height 16
    static constexpr int NWH = 3;                       // halo-row waves
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

// One workgroup's assignment.  32 bytes, read with scalar loads.
struct WaferF3Block {
    int tile;        // ty * ntx + tx
    int zs, ze;      // output planes [zs, ze) (local plane indices)
    int down;        // 1: march from ze-1 down to zs
    int wait_late;   // >= 0: wait for ghost flag [wait_late] at the top of iteration wait_it (the first prefetch of a ghost plane)
    int wait_it;
    int bump;        // >= 0: add 1 to completion counter [bump] after the last store
    int wt;          // with bump: the last wt planes of the march are what the exchange sends: stored write-through
};

// Device words of the single-launch slab pass, all in device memory and accessed at agent scope (relaxed atomics; the
// payload is ordered by one release fence before a count and one acquire fence after a poll -- MI355X_MICROARCH.md,
// "Workgroup dispatch, XCD placement & inter-workgroup visibility").  cnt[i] counts finished workgroups of half i; a
// one-wave gate kernel on the exchange stream polls it.  flag[i] is written by a one-wave kernel on the exchange
// stream after the exchange that fills ghost side i (0: lower ghost planes, 1: upper) has completed; need[i] is the
// value the workgroups of THIS launch wait for.  *err (host memory) is set when a wait gives up (a bounded spin:
// the host reports WAFER_ERR_COMM instead of hanging).
// (Counters in host / signal memory -- what hipStreamWaitValue64 needs -- were measured first: a round of 256
//  workgroups ends within microseconds, their 256 system-scope atomics queue up on the host link and every CU idles
//  until its own has returned: 0.51 against 0.30 ms/step at the bench slab.)
struct WaferF3Sync {
    unsigned long long *cnt = nullptr;          // [0], [8]: one 64-byte line each
    const unsigned long long *flag = nullptr;   // [0], [8]
    unsigned long long need[2] = {0, 0};
    unsigned *err = nullptr;
    int debug = 0;   // WAFER_HV_DEBUG bit 4: no acquire fence (timing experiments)
};
enum { WAFER_F3_SYNC_STRIDE = 8 }; // 64-bit words between the two counters / flags

__device__ __forceinline__ void wafer_f3_wait(const WaferF3Sync &sy, int idx, int tid)
{
    if (tid == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(sy.flag + idx * WAFER_F3_SYNC_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < sy.need[idx]) {
            __builtin_amdgcn_s_sleep(32);
            if (++spins > (1u << 19)) { // ~half a second (a legitimate wait is well under a millisecond): the exchange never arrived
                __hip_atomic_store(sy.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
        // system scope: the ghost planes were written by another kernel, possibly (through the fabric) of another device
        if (!(sy.debug & 4)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

template <typename T, typename C, bool VIR, bool DOWN>
__device__ __forceinline__ void wafer_step3_body(const WaferStepArgs &a, const WaferF3Block &blk, int ntx, const WaferF3Sync &sy,
                                                 const T *__restrict__ phi, const T *__restrict__ pv, T *__restrict__ out,
                                                 T *lds0, T *lds1, T *lds2)
{
    using Cfg = WaferF3Cfg<T>;
    using VT = typename WaferVec<T>::type;
    constexpr int R = 1;
    constexpr int VEC = Cfg::VEC, RY = Cfg::RY, TX = Cfg::TX, TY = Cfg::TY;
    constexpr int HX0 = Cfg::HX0, HX1 = Cfg::HX1, HX2 = Cfg::HX2, LP0 = Cfg::LP0, LP1 = Cfg::LP1, LP2 = Cfg::LP2;
    constexpr int SD = DOWN ? -1 : 1;                    // marching direction along z
    constexpr int ZLO = DOWN ? 2 : 0, ZHI = DOWN ? 0 : 2; // queue slots of the planes below / above the centre plane

    const WaferGeom &g = a.g;
    const int tx_i = blk.tile % ntx, ty_i = blk.tile / ntx;
    const int zs = blk.zs, ze = blk.ze;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // wave-uniform: role tests become scalar branches
    const int x0 = tx_i * TX, y0 = ty_i * TY;
    const C dt = (C)a.dt, den = (C)a.den;
    constexpr bool vir = VIR;
    const bool is_main = wave < Cfg::NW2;
    const bool is_hrow = wave >= Cfg::NW2 && wave < Cfg::NW2 + Cfg::NWH;
    const bool is_hcol = wave == Cfg::NW - 1;

    VT zero;
#pragma unroll
    for (int v = 0; v < VEC; ++v) zero[v] = T(0);

    // ---- row slots of the main and halo-row waves (no bounds predicates on loads: whole tiles, three
    //      halo rows / columns and three planes past the slab lie in the zero guard zone, wafer_geom.h)
    const int xl = lane * VEC, xi = x0 + xl;
    int yrow[RY];
    bool rowwk[RY], lvl2[RY], slot_on[RY];
    // rowoff holds only the WAVE-UNIFORM part of a row's element offset (scalar registers) and the lane adds its
    // 32-bit x offset at the access.  Per-lane 64-bit offsets cost six VGPRs in a kernel that sits at its 168-VGPR
    // cap: the compiler spilled them, and reloading the store addresses from scratch put an s_waitcnt vmcnt(0) --
    // scratch loads share the counter -- in front of each store of the plane.
    long long rowoff[RY];
    const unsigned xlu = (unsigned)(lane * VEC);
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        int y;
        bool l2 = true, on = true;
        if (is_hrow) {
            const int h = wave - Cfg::NW2;   // 0: row y0-1 (phi1, phi2);  1: row y0+16 (phi1, phi2);  2: rows y0-2, y0+17 (phi1)
            if (h == 2) {
                y = (r == 0) ? (y0 - 2) : (y0 + TY + 1);
                l2 = false;
            } else {
                y = (h == 0) ? (y0 - 1) : (y0 + TY);   // the second slot repeats the row (its loads are the same values) and computes nothing
                l2 = on = (r == 0);
            }
        } else {
            y = y0 + wave * RY + r;                    // main (unused by the halo-column wave)
        }
        yrow[r] = y;
        rowwk[r] = on && (y >= 0) && (y < g.ny);
        lvl2[r] = l2;
        slot_on[r] = on;
        rowoff[r] = (long long)(y + R) * g.pitch + g.xoff + R + x0;
    }
    // ---- outermost phi0 halo rows y0-3 and y0+18, fetched by main waves 0 and 1
    const bool has_orow = is_main && wave < 2;
    const int oy = (wave == 0) ? (y0 - 3) : (y0 + TY + 2);
    const long long orow_off = (long long)(oy + R) * g.pitch + g.xoff + R + x0;   // as rowoff
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

    // ---- prologue: the first phi1 plane is z1 (two planes before the first output plane in marching order); the
    //      phi0 queue holds planes z1-SD, z1, z1+SD
    const int z1 = DOWN ? ze + 1 : zs - 2;
    VT q0[3][RY], q1[3][RY], q2[3][RY];
    VT vq[3][RY];   // V of the planes of levels 3, 2 and 1 of one iteration
    // (the halo-column wave keeps cell q of its CPL cells in component q % VEC of row slot q / VEC)
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int r = 0; r < RY; ++r) q0[m][r] = q1[m][r] = q2[m][r] = vq[m][r] = zero;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const long long po = (long long)(z1 + SD * (m - 1)) * g.plane;
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
    for (int i = tid; i < 2 * Cfg::TILE1; i += Cfg::NT_) lds1[i] = T(0);
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
    if (has_orow) orow_nxt = *reinterpret_cast<const VT *>((phi + (long long)(z1 + SD) * g.plane + orow_off) + xlu);
    __syncthreads();

    const int niter = (ze - zs) + 4; // phi1 planes z1, z1+SD, ..., two past the last output plane
    for (int it = 0; it < niter; ++it) {
        const int z = z1 + SD * it;
        const bool more = it + 1 < niter;
        const long long zo = (long long)z * g.plane;
        if (blk.wait_late >= 0 && it == blk.wait_it) wafer_f3_wait(sy, blk.wait_late, tid);
        // ---- 1. prefetch: phi0 two planes ahead, V one plane ahead, outer halo rows two planes ahead
        VT pre[RY], pre_v[RY], orow_pre = zero;
#pragma unroll
        for (int r = 0; r < RY; ++r) pre[r] = pre_v[r] = zero;
        if (!is_hcol) {
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                pre[r] = *reinterpret_cast<const VT *>((phi + zo + SD * 2 * g.plane + rowoff[r]) + xlu);
                pre_v[r] = *reinterpret_cast<const VT *>((pv + zo + SD * g.plane + rowoff[r]) + xlu);
            }
            if (has_orow) orow_pre = *reinterpret_cast<const VT *>((phi + zo + SD * 2 * g.plane + orow_off) + xlu);
        } else {
#pragma unroll
            for (int q = 0; q < Cfg::CPL; ++q) {
                pre[q / VEC][q % VEC] = phi[zo + SD * 2 * g.plane + c_off[q]];
                pre_v[q / VEC][q % VEC] = pv[zo + SD * g.plane + c_off[q]];
            }
        }
        // ---- 2. stage the next phi0 plane into the other buffer
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
        // rings: a plane lives in slot (plane & 1); the plane one step behind in marching order has the other parity
        T *w1 = lds1 + (z & 1) * Cfg::TILE1;
        const T *c1 = lds1 + ((z + 1) & 1) * Cfg::TILE1;     // phi1 plane z - SD
        T *w2 = lds2 + ((z + 1) & 1) * Cfg::TILE2;           // phi2 plane z - SD
        const T *c2 = lds2 + (z & 1) * Cfg::TILE2;           // phi2 plane z - 2 SD
        const bool wplane1 = work_plane(z), wplane2 = work_plane(z - SD);
        VT p1new[RY], p2new[RY];
#pragma unroll
        for (int r = 0; r < RY; ++r) p1new[r] = p2new[r] = zero;

        if (!is_hcol) {
            bool all_rows = x0 + TX <= g.nx;   // INTERIOR also requires the tile's columns to be work columns: no per-cell x mask
#pragma unroll
            for (int r = 0; r < RY; ++r) all_rows = all_rows && rowwk[r];   // (rowwk is false for a slot that is off)
            // ---- 3. level 1: phi1 plane z (main and halo-row waves).  INTERIOR: the plane and every row of this
            //         wave are work cells -- no tests inside, so the RY x VEC updates form one basic block
            // (the rows of a halo-row wave are not neighbours: their y neighbours come from LDS -- yreg_tag)
            auto level1 = [&](auto interior_tag, auto yreg_tag) {
                constexpr bool INTERIOR = decltype(interior_tag)::value;
                constexpr bool YR = decltype(yreg_tag)::value;
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
                            ys[0] = (YR && r >= 1) ? (C)q0[1][r >= 1 ? r - 1 : 0][v] : (C)c0[(ly - 1) * LP0 + HX0 + xl + v];
                            ys[2] = (YR && r + 1 < RY) ? (C)q0[1][r + 1 < RY ? r + 1 : RY - 1][v] : (C)c0[(ly + 1) * LP0 + HX0 + xl + v];
                            const C S = wafer_stencil_sum<C, 1>(xs, ys, zz, w);
                            const T rs = update(w, (C)vq[2][r][v], S);
                            res[v] = (INTERIOR || xi + v < g.nx) ? rs : T(0);
                        }
                    }
                    p1new[r] = res;
                    if (INTERIOR || slot_on[r]) *reinterpret_cast<VT *>(w1 + (yrow[r] - (y0 - 2)) * LP1 + HX1 + xl) = res;
                }
            };
            if (is_main) {
                if (all_rows && wplane1) level1(std::true_type{}, std::true_type{});
                else level1(std::false_type{}, std::true_type{});
            } else {
                if (all_rows && wplane1) level1(std::true_type{}, std::false_type{});
                else level1(std::false_type{}, std::false_type{});
            }
            // ---- 4. level 2: phi2 of the plane behind from the phi1 queue; x / y neighbours from the phi1 ring slot
            //         written one iteration ago
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                q1[0][r] = q1[1][r];
                q1[1][r] = q1[2][r];
                q1[2][r] = p1new[r];
            }
            auto level2 = [&](auto interior_tag, auto yreg_tag) {
                constexpr bool INTERIOR = decltype(interior_tag)::value;
                constexpr bool YR = decltype(yreg_tag)::value;
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    VT res = zero;
                    if (INTERIOR || (lvl2[r] && wplane2 && rowwk[r])) {
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
                            ys[0] = (YR && r >= 1) ? (C)q1[1][r >= 1 ? r - 1 : 0][v] : (C)c1[(ly - 1) * LP1 + HX1 + xl + v];
                            ys[2] = (YR && r + 1 < RY) ? (C)q1[1][r + 1 < RY ? r + 1 : RY - 1][v] : (C)c1[(ly + 1) * LP1 + HX1 + xl + v];
                            const C S = wafer_stencil_sum<C, 1>(xs, ys, zz, w);
                            const T rs = update(w, (C)vq[1][r][v], S);
                            res[v] = (INTERIOR || xi + v < g.nx) ? rs : T(0);
                        }
                    }
                    p2new[r] = res;
                    if (INTERIOR || lvl2[r]) *reinterpret_cast<VT *>(w2 + (yrow[r] - (y0 - 1)) * LP2 + HX2 + xl) = res;
                }
            };
            // (phi2 is needed on planes zs-1 .. ze: the first two iterations of a column produce planes outside that range --
            //  zeros are as good there, level 3 never reads them)
            const int zp2 = z - SD;
            if (zp2 >= zs - 1 && zp2 <= ze) {
                if (is_main && all_rows && wplane2) level2(std::true_type{}, std::true_type{});
                else if (is_main) level2(std::false_type{}, std::true_type{});
                else level2(std::false_type{}, std::false_type{});
            }
            // ---- 5. level 3 (main waves): phi3 two planes behind from the phi2 queue, stored
            if (is_main) {
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    q2[0][r] = q2[1][r];
                    q2[1][r] = q2[2][r];
                    q2[2][r] = p2new[r];
                }
                const int zo3 = z - 2 * SD;
                // the planes the exchange sends once this workgroup has counted itself done: the last `wt` planes of the march
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
                                    res3[r][v] = update(w, (C)vq[0][r][v], S);
                                }
                            }
                        }
#pragma unroll
                        for (int r = 0; r < RY; ++r) {
                            if (INTERIOR || rowwk[r]) {
                                T *dst = (out + (long long)zo3 * g.plane + rowoff[r]) + xlu;
                                if (wthrough) {
                                    // a plane the exchange will send: write-through stores (agent-scope relaxed atomics: sc1), so
                                    // that the count after them needs no cache write-back (MI355X_MICROARCH.md, valid forms: sc1
                                    // payload, every storing wave's vmcnt(0), the workgroup's barrier, then the counter)
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
                        zz[0] = (C)q0[ZLO][q / VEC][q % VEC]; zz[1] = w; zz[2] = (C)q0[ZHI][q / VEC][q % VEC];
                        xs[1] = ys[1] = w;
                        xs[0] = (C)c0[o0 - 1]; xs[2] = (C)c0[o0 + 1];
                        ys[0] = (C)c0[o0 - LP0]; ys[2] = (C)c0[o0 + LP0];
                        const C S = wafer_stencil_sum<C, 1>(xs, ys, zz, w);
                        rs = update(w, (C)vq[2][q / VEC][q % VEC], S);
                    }
                    w1[c_lds1[q]] = rs;
                }
                p1new[q / VEC][q % VEC] = rs;
            }
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                q1[0][r] = q1[1][r];
                q1[1][r] = q1[2][r];
                q1[2][r] = p1new[r];
            }
#pragma unroll
            for (int q = 0; q < Cfg::CPL; ++q) {
                if (c_l2[q]) {
                    T rs = T(0);
                    if (wplane2 && c_wk[q] && (z - SD) >= zs - 1 && (z - SD) <= ze) {
                        const int o1 = c_lds1[q];
                        const C w = (C)q1[1][q / VEC][q % VEC];
                        C xs[3], ys[3], zz[3];
                        zz[0] = (C)q1[ZLO][q / VEC][q % VEC]; zz[1] = w; zz[2] = (C)q1[ZHI][q / VEC][q % VEC];
                        xs[1] = ys[1] = w;
                        xs[0] = (C)c1[o1 - 1]; xs[2] = (C)c1[o1 + 1];
                        ys[0] = (C)c1[o1 - LP1]; ys[2] = (C)c1[o1 + LP1];
                        const C S = wafer_stencil_sum<C, 1>(xs, ys, zz, w);
                        rs = update(w, (C)vq[1][q / VEC][q % VEC], S);
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
        }
        orow_nxt = orow_pre;
    }
    // ---- completion counter of the single-launch slab pass: every storing wave drains its stores, the workgroup
    //      meets, one lane counts
    if (blk.bump >= 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // (the planes the exchange reads were stored write-through; a half thinner than the exchange depth also sends planes
        //  of the other half's workgroups, whose own write-through planes cover them: wt = the whole piece there)
        if (tid == 0) __hip_atomic_fetch_add(sy.cnt + blk.bump * WAFER_F3_SYNC_STRIDE, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <typename T, typename C, bool VIR>
__global__ __launch_bounds__((WaferF3Cfg<T>::NT_)) void wafer_k_step3_fused(WaferStepArgs a, int ntx, const WaferF3Block *__restrict__ table,
                                                                           WaferF3Sync sy, const T *__restrict__ phi,
                                                                           const T *__restrict__ pv, T *__restrict__ out)
{
    using Cfg = WaferF3Cfg<T>;
    __shared__ __attribute__((aligned(16))) T lds0[2 * Cfg::TILE0];
    __shared__ __attribute__((aligned(16))) T lds1[2 * Cfg::TILE1];
    __shared__ __attribute__((aligned(16))) T lds2[2 * Cfg::TILE2];
    const WaferF3Block blk = table[blockIdx.x];   // uniform address: scalar loads
    if (blk.down) wafer_step3_body<T, C, VIR, true>(a, blk, ntx, sy, phi, pv, out, lds0, lds1, lds2);
    else wafer_step3_body<T, C, VIR, false>(a, blk, ntx, sy, phi, pv, out, lds0, lds1, lds2);
}

// ---- schedules (host) ---------------------------------------------------------------------------------------------
#include <vector>

// tiles in the order the XCD-aware map visits them: workgroup b runs on XCD b % 8 (observed, speed only), and each XCD
// should work on one contiguous range of tiles so that neighbouring tiles' halo rows are in its L2.  Returns the tile
// for dispatch slot b of n.
static inline int wafer_f3_xcd_slot(int b, int n)
{
    const int q = n >> 3, r = n & 7, k = b & 7;
    return k * q + (k < r ? k : r) + (b >> 3);
}

// Planes [lz_lo, lz_hi) of every tile, cut into chunks of `zchunk` planes, all marching up: the schedule of an
// undecomposed grid (one workgroup per CU marching a long column) and of every unsplit pass.
static inline void wafer_f3_schedule_plain(std::vector<WaferF3Block> &out, int ntx, int nty, int lz_lo, int lz_hi, int zchunk, bool swz)
{
    const int nplanes = lz_hi - lz_lo, nch = (nplanes + zchunk - 1) / zchunk, n = ntx * nty * nch;
    out.resize((size_t)n);
    for (int b = 0; b < n; ++b) {
        const int id = swz ? wafer_f3_xcd_slot(b, n) : b;     // x fastest, then y, then z-chunk
        WaferF3Block k{};
        k.tile = id % (ntx * nty);
        k.zs = lz_lo + (id / (ntx * nty)) * zchunk;
        k.ze = k.zs + zchunk < lz_hi ? k.zs + zchunk : lz_hi;
        k.down = 0;
        k.wait_late = k.bump = -1;
        k.wait_it = k.wt = 0;
        out[(size_t)b] = k;
    }
}

// The interior launch of a split slab pass: one long workgroup per tile, except the last 1/16 of the tiles, which go as
// `nsub` short workgroups each -- dispatched last, they fill the holes the exchange's kernels leave (wafer_engine.hip).
static inline void wafer_f3_schedule_mixed(std::vector<WaferF3Block> &out, int ntx, int nty, int lz_lo, int lz_hi, int nsub)
{
    const int ntiles = ntx * nty, nplanes = lz_hi - lz_lo;
    const int nshort = ntiles / 16 > 0 ? ntiles / 16 : 1, nlong = ntiles - nshort;
    const int zc = (nplanes + nsub - 1) / nsub;
    out.clear();
    auto push = [&](int tile, int zs, int ze) {
        WaferF3Block k{};
        k.tile = tile; k.zs = zs; k.ze = ze; k.down = 0;
        k.wait_late = k.bump = -1;
        k.wait_it = k.wt = 0;
        out.push_back(k);
    };
    for (int t = 0; t < nlong; ++t) push(t, lz_lo, lz_hi);
    for (int t = nlong; t < ntiles; ++t)
        for (int sub = 0; sub < nsub; ++sub) {
            const int zs = lz_lo + sub * zc, ze = zs + zc < lz_hi ? zs + zc : lz_hi;
            if (zs < ze) push(t, zs, ze);
        }
}

// The single-launch pass of a z-slab (overlap mode 2): the slab is cut at `mid` into half A = [lo, mid), marched DOWN from the
// cut to the lower boundary, and half B = [mid, hi), marched UP to the upper boundary.  Both halves read the pass's input
// across the cut, so the split costs one extra pipeline fill per tile and no redundant planes.  Marching outwards, a
// workgroup reads its ghost planes (filled by the previous pass's exchange) LAST and stores its boundary planes LAST:
// `first` names the half dispatched first; its exchange (released by counter [half] when all its workgroups have
// finished) runs beside the other half, and the other half's exchange beside the next pass, whose first half is the one
// that does not read the ghost planes still in flight (the order alternates from pass to pass).  need_wait[h]: half h
// has a neighbour on its side (its ghost planes come from an exchange).  The last `nshort_tiles` tiles of each half go as
// `nsub` short workgroups: dispatched at the head of the SECOND half they retire soon after the first half's exchange has
// been released and hand it their CUs (the exchange's workgroups cannot share a CU with a stencil workgroup).
static inline void wafer_f3_schedule_halves(std::vector<WaferF3Block> &out, int ntx, int nty, int lo, int hi, int mid, int first,
                                            const bool need_wait[2], int nshort_tiles, int nsub, int depth, bool sync = true, int debug = 0, int layout = 0)
{
    const int ntiles = ntx * nty;
    out.clear();
    auto push = [&](int half, int tile_, int zs, int ze) {
        WaferF3Block k{};
        const int tile = (debug & 16) ? wafer_f3_xcd_slot(tile_, ntiles) : tile_;
        k.tile = tile; k.zs = zs; k.ze = ze;
        k.down = half == 0;
        k.wait_late = k.bump = -1;
        k.wait_it = k.wt = 0;
        if (!sync) { out.push_back(k); return; }
        // only the piece that stores the half's boundary planes counts itself done: that is what the exchange waits for
        const bool at_boundary = half == 0 ? zs == lo : ze == hi;
        // (a half thinner than the exchange depth: its side's planes reach into the other half, so every plane of both is
        //  stored write-through and the host waits for both counters)
        const bool thin = mid - lo < depth || hi - mid < depth;
        if (at_boundary && !(debug & 32)) { k.bump = half; k.wt = (thin || depth > ze - zs) ? ze - zs : depth; }
        // The first load that touches a ghost plane is the phi0 prefetch two planes ahead: half A (z = ze + 1 - it going
        // down) reaches plane lo - 1 at it = ze - lo; half B (z = zs - 2 + it going up) reaches plane hi at it = hi - zs.
        // The prologue stays within three planes of the piece's start, which lies on the side of the cut.
        if (at_boundary && need_wait[half]) { k.wait_late = half; k.wait_it = half == 0 ? ze - lo : hi - zs; }
        out.push_back(k);
    };
    auto column = [&](int half, int tile, int pieces) {
        const int zs0 = half == 0 ? lo : mid, ze0 = half == 0 ? mid : hi, n = ze0 - zs0;
        if (pieces <= 1 || n < 8 * pieces) { push(half, tile, zs0, ze0); return; }
        const int zc = (n + pieces - 1) / pieces;
        // in marching order: the piece at the cut first, the piece at the boundary last
        for (int p = 0; p < pieces; ++p) {
            int zs, ze;
            if (half == 0) { ze = ze0 - p * zc; zs = ze - zc > zs0 ? ze - zc : zs0; }
            else { zs = zs0 + p * zc; ze = zs + zc < ze0 ? zs + zc : ze0; }
            if (zs < ze) push(half, tile, zs, ze);
        }
    };
    const int ns = nshort_tiles < ntiles ? nshort_tiles : 0, nlong = ntiles - ns;
    for (int i = 0; i < 2; ++i) {
        const int half = (first + i) & 1;
        // layout 0: the short columns sit between the halves' long ones (tail of the first half, head of the second);
        // layout 1: the second half's short columns are split between its head (CUs for the exchange) and its tail
        // (the kernel's last round evens out)
        // layout 2: only the second half has short columns, at its head
        const int head = i == 0 ? 0 : (layout == 1 ? ns / 2 : ns);
        const int nl = (layout == 2 && i == 0) ? ntiles : nlong;
        for (int t = nl; t < nl + head && t < ntiles; ++t) column(half, t, nsub);
        for (int t = 0; t < nl; ++t) column(half, t, 1);
        for (int t = nl + head; t < ntiles; ++t) column(half, t, nsub);
    }
}

// planes per workgroup of the plain schedule: one workgroup per CU marching a long column (as the two-step kernel)
static inline int wafer_f3_zchunk(const WaferTuning &t, int ntx, int nty, int nplanes, int target_blocks)
{
    if (t.zchunk > 0) return t.zchunk;
    if (target_blocks < 0) return -target_blocks < nplanes ? -target_blocks : nplanes;
    const long long per_layer = (long long)ntx * nty;
    const long long target = t.target_blocks > 0 ? t.target_blocks : (target_blocks > 0 ? target_blocks : 256);
    long long nch = (target + per_layer / 2) / per_layer;
    if (nch < 1) nch = 1;
    if (nch > nplanes) nch = nplanes;
    return (int)((nplanes + nch - 1) / nch);
}

// Advances the planes of `table` (device copy, nblocks entries) by THREE steps: out = step(step(step(phi))).  ThreePoint only.
template <typename T, typename C>
static inline hipError_t wafer_launch_step3_fused(const WaferTuning &t, const WaferStepArgs &a, const WaferF3Block *table, int nblocks,
                                                  const WaferF3Sync &sy, const T *phi, const T *pv, T *out, hipStream_t s)
{
    using Cfg = WaferF3Cfg<T>;
    const int ntx = (a.g.nx + Cfg::TX - 1) / Cfg::TX;
    const dim3 grid((unsigned)nblocks), block(Cfg::NT_);
    if (a.v_in_range != 0)
        hipLaunchKernelGGL((wafer_k_step3_fused<T, C, true>), grid, block, (size_t)t.lds_pad, s, a, ntx, table, sy, phi, pv, out);
    else
        hipLaunchKernelGGL((wafer_k_step3_fused<T, C, false>), grid, block, (size_t)t.lds_pad, s, a, ntx, table, sy, phi, pv, out);
    return hipGetLastError();
}
