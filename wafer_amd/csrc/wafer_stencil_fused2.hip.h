// Two imaginary-time steps per pass over HBM (temporal blocking) for the
// ground-state evolve loop (grid.rs:562-686 with wnum == 0, where nothing but
// the stencil happens between two steps).
//
//   phi0 --step--> phi1 --step--> phi2
//
// A workgroup marches a (TX x TY) tile along z like wafer_k_step_lds, but keeps
// TWO register pipelines: plane z of phi1 is produced from the phi0 z-queue and
// immediately feeds the phi1 z-queue from which plane z-R of phi2 is produced.
// phi1 never touches HBM.  a and b are formed from V in registers
// (potential.rs:104-110, same expressions => same bits), so the pass reads
// phi0 and V once and writes phi2 once: 24 B of HBM traffic per TWO updates
// (12 B/update against the 32 B/update the roofline figure is priced at).
// Arithmetic per update is unchanged, so results stay bit-identical to two
// single steps.
//
// Tile roles (RY = 2 rows per lane, VEC = 16 B of x per lane):
//   waves 0..3            "main": own rows y0..y0+7 in both steps;
//   waves 4..4+R-1        "halo-row": own the 2R phi1 halo rows (y0-R.., y0+TY..),
//                         step 1 only;
//   last wave             "halo-column": each lane owns a few phi0 halo-column
//                         cells (2R columns each side, z-queue in registers) and
//                         produces the phi1 halo-column cells (R columns each side).
// phi0's outermost 2R halo rows are plain vector loads staged through LDS.
// LDS: double-buffered phi0 centre tile (TY+4R rows) + ring of R+1 phi1 tiles
// (TY+2R rows).  One s_barrier per plane.
//
// Cells of phi1 outside the work area (Dirichlet frame, config.rs:597-622) are
// forced to 0 exactly as the reference never updates them; phi1 planes outside
// the global work range likewise.  z-chunks recompute R planes of phi1 on each
// side; slabs of a sharded grid need 2R valid ghost planes of phi0.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "wafer_geom.h"
#include "wafer_stencil.hip.h"
#include "wafer_stencil_lds.hip.h"

template <typename T, int R, int NW2_ = 4>
struct WaferF2Cfg {
    static constexpr int VEC = WaferVec<T>::N;
    static constexpr int RY = 2;
    static constexpr int NW2 = NW2_;                     // main waves (tile height = 2 * NW2)
    static constexpr int NWH = (2 * R) / RY;             // halo-row waves (R)
    static constexpr int NW = NW2 + NWH + 1;             // + halo-column wave
    static constexpr int NT_ = NW * 64;                  // threads per workgroup
    static constexpr int TX = 64 * VEC, TY = NW2 * RY;
    static constexpr int HC0 = 2 * R, HC1 = R;           // halo columns per side: phi0, phi1
    static constexpr int HX0 = ((HC0 + VEC - 1) / VEC) * VEC;
    static constexpr int HX1 = ((HC1 + VEC - 1) / VEC) * VEC;
    static constexpr int LP0 = TX + 2 * HX0, LP1 = TX + 2 * HX1;
    static constexpr int ROWS0 = TY + 4 * R, ROWS1 = TY + 2 * R;
    static constexpr int TILE0 = ROWS0 * LP0, TILE1 = ROWS1 * LP1;
    static constexpr int NB1 = R + 1;                    // phi1 ring depth
    static constexpr int NCOL = 2 * HC0 * ROWS0;         // phi0 halo-column cells per plane
    static constexpr int CPL = (NCOL + 63) / 64;         // cells per lane of the halo-column wave
    static constexpr int OUTER = 2 * R;                  // outermost phi0 halo rows (vector loads)
    static constexpr int OPW = (OUTER + NW2 - 1) / NW2;  // of which per main wave
};

// a, b from V, then the update: potential.rs:104-110 + grid.rs:580-589
template <typename C>
__device__ __forceinline__ C wafer_update_v(C w, C vv, C dt, C S, const WaferDen<C> &den, bool v_in_range)
{
    const C cb = wafer_recip(C(1) + dt * vv / C(2), v_in_range);
    const C ca = (C(1) - dt * vv / C(2)) * cb;
    return w * ca + wafer_div_invariant<C>(cb * dt * S, den);
}

// a, b of one cell from V (potential.rs:104-110)
template <typename C>
__device__ __forceinline__ void wafer_ab_from_v(C vv, C dt, bool v_in_range, C &ca, C &cb)
{
    cb = wafer_recip(C(1) + dt * vv / C(2), v_in_range);
    ca = (C(1) - dt * vv / C(2)) * cb;
}

// ABV: pv is V and a, b are formed in registers (24 B per two updates);
// !ABV: pv is a, pb is b, streamed (32 B per two updates, ~14 fp64 ops fewer per update).
// VIR: the potential passed check_v_range, so b's reciprocal takes its short form -- a template
// parameter rather than a kernel argument so that the updates of a plane form one basic block and
// their division chains interleave.
// YR: step 1 takes the y neighbours that lie inside the lane's own RY rows from registers instead of LDS (as step 2
// does): fewer LDS reads on the critical path, the same values.
template <typename T, typename C, int R, bool NT, bool ABV, int NW2 = 4, bool VIR = false, bool YR = true>
__global__ __launch_bounds__((WaferF2Cfg<T, R, NW2>::NT_)) void wafer_k_step2_fused(
    WaferStepArgs a, int ntx, int nty, int swz, const T *__restrict__ phi, const T *__restrict__ pv,
    const T *__restrict__ pb, T *__restrict__ out)
{
    using Cfg = WaferF2Cfg<T, R, NW2>;
    using VT = typename WaferVec<T>::type;
    constexpr int VEC = Cfg::VEC, RY = Cfg::RY, TX = Cfg::TX, TY = Cfg::TY;
    constexpr int HX0 = Cfg::HX0, HX1 = Cfg::HX1, LP0 = Cfg::LP0, LP1 = Cfg::LP1;
    __shared__ __attribute__((aligned(16))) T lds0[2 * Cfg::TILE0];
    __shared__ __attribute__((aligned(16))) T lds1[Cfg::NB1 * Cfg::TILE1];

    const WaferGeom &g = a.g;
    int bid = blockIdx.x;
    if (swz) {
        const int n = gridDim.x, q = n >> 3, r = n & 7, k = bid & 7;
        bid = k * q + min(k, r) + (bid >> 3);
    }
    int tx_i, ty_i, zs, ze;
    if (a.nsub > 1) { // mixed launch: long workgroups first (dispatched first), the last tiles as short ones
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
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // provably wave-uniform: role tests become scalar branches
    const int x0 = tx_i * TX, y0 = ty_i * TY;
    const C dt = (C)a.dt;
    constexpr bool vir = VIR;
    const WaferDen<C> den = wafer_den<C>(a, vir);
    const bool is_main = wave < Cfg::NW2;
    const bool is_hrow = wave >= Cfg::NW2 && wave < Cfg::NW2 + Cfg::NWH;
    const bool is_hcol = wave == Cfg::NW - 1;

    VT zero;
#pragma unroll
    for (int v = 0; v < VEC; ++v) zero[v] = T(0);

    // ---- row slots of the main and halo-row waves ---------------------------------
    const int xl = lane * VEC, xi = x0 + xl;
    // Loads carry no bounds predicates: rows y0-2R .. y0+TY+2R, planes zs-2R .. ze+2R and whole
    // tiles of x lie inside the allocation's zero guard zone (wafer_geom.h).
    int yrow[RY];          // work y of slot r
    bool rowwk[RY];        // the row is a work row (wave-uniform)
    long long rowoff[RY];
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        int y;
        if (is_hrow) {
            const int h = (wave - Cfg::NW2) * RY + r;                 // 0..2R-1
            y = (h < R) ? (y0 - R + h) : (y0 + TY + (h - R));
        } else {
            y = y0 + wave * RY + r;                                   // main (unused for hcol)
        }
        yrow[r] = y;
        rowwk[r] = (y >= 0) && (y < g.ny);
        rowoff[r] = (long long)(y + R) * g.pitch + g.xoff + R + xi;
    }
    // ---- outermost phi0 halo rows, fetched by the main waves -------------------------
    long long orow_off[Cfg::OPW];
    int orow_lds[Cfg::OPW];
#pragma unroll
    for (int q = 0; q < Cfg::OPW; ++q) {
        const int o = wave + q * Cfg::NW2;                            // 0..2R-1 valid
        const int y = (o < R) ? (y0 - 2 * R + o) : (y0 + TY + R + (o - R));
        orow_off[q] = (long long)(y + R) * g.pitch + g.xoff + R + xi;
        orow_lds[q] = (y - (y0 - 2 * R)) * LP0 + HX0 + xl;
    }
    // ---- halo-column cells of the last wave ------------------------------------------------
    // (a cell outside the work area -- frame and pad columns, frame rows: zeros that no kernel writes -- is not fetched: its
    //  128-byte line holds nothing anybody else reads, so each such request was an HBM read of its own.  The lane requests the
    //  tile's own edge cell instead, a line the row's owner requests in the same iteration, and the value becomes the zero it
    //  stands for where it is staged into the tile; the cell's queues are read nowhere else: c_p1.)
    bool c_p1[Cfg::CPL], c_out[Cfg::CPL];
    long long c_off[Cfg::CPL];
    int c_lds0[Cfg::CPL], c_lds1[Cfg::CPL];
#pragma unroll
    for (int q = 0; q < Cfg::CPL; ++q) {
        const int cidx = min(lane + q * 64, Cfg::NCOL - 1);         // surplus lanes repeat the last cell
        const int row = cidx / (2 * Cfg::HC0), k = cidx % (2 * Cfg::HC0);
        const int kk = (k < Cfg::HC0) ? k : k - Cfg::HC0;             // distance-1 from the tile edge
        const int lc = (k < Cfg::HC0) ? (-1 - kk) : (TX + kk);        // column relative to x0
        const int xw = x0 + lc, y = y0 - 2 * R + row;
        const bool valid = is_hcol && lane + q * 64 < Cfg::NCOL;
        // phi1 is produced on the inner R columns and rows y0-R .. y0+TY+R-1, work cells only
        c_p1[q] = valid && (kk < R) && (row >= R) && (row < Cfg::ROWS0 - R) && (y >= 0) && (y < g.ny) &&
                  (xw >= 0) && (xw < g.nx);
        c_out[q] = xw < 0 || xw >= g.nx || y < 0 || y >= g.ny;
        c_off[q] = (long long)((y < 0 ? y0 : y >= g.ny ? y0 + TY - 1 : y) + R) * g.pitch + g.xoff + R +
                   ((xw < 0 || xw >= g.nx) ? ((k < Cfg::HC0) ? x0 : x0 + TX - 1) : xw);
        c_lds0[q] = row * LP0 + HX0 + lc;
        c_lds1[q] = (row - R) * LP1 + HX1 + lc;
    }
    const bool c_p1slot_valid = is_hcol;

    auto work_plane = [&](int p) {
        const int kg = g.z_begin + (p - g.G);
        return kg >= 0 && kg < g.nz;
    };

    // ---- prologue ---------------------------------------------------------------------------------
    // first phi1 plane is z1 = zs - R; phi0 z-queue holds planes z1-R .. z1+R
    const int z1 = zs - R;
    VT q0[2 * R + 1][RY];
    VT vq[R + 1][RY];     // V (or a) of planes z-R .. z (oldest first); only vq[R] is used by step 1
    VT bq[R + 1][RY];     // b of the same planes (!ABV only)
    // The halo-column wave keeps ITS state in the same registers: cell q of its CPL cells lives in
    // component q % VEC of row slot q / VEC (q0[m][q / VEC][q % VEC], vq[R][...], ...), so the roles do not add up in the
    // kernel's register budget.
    static_assert(Cfg::CPL <= RY * VEC, "halo-column cells per lane must fit the row-slot registers");
#pragma unroll
    for (int m = 0; m <= 2 * R; ++m) {
        const int p = z1 - R + m;
#pragma unroll
        for (int r = 0; r < RY; ++r) q0[m][r] = zero;
        if (!is_hcol) {
#pragma unroll
            for (int r = 0; r < RY; ++r) q0[m][r] = *reinterpret_cast<const VT *>(phi + (long long)p * g.plane + rowoff[r]);
        } else {
#pragma unroll
            for (int q = 0; q < Cfg::CPL; ++q) q0[m][q / VEC][q % VEC] = phi[(long long)p * g.plane + c_off[q]];
        }
    }
#pragma unroll
    for (int m = 0; m <= R; ++m)
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            vq[m][r] = zero;
            bq[m][r] = zero;
        }
    if (!is_hcol) {
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            vq[R][r] = wafer_ld_stream<NT>(reinterpret_cast<const VT *>(pv + (long long)z1 * g.plane + rowoff[r]));
            if constexpr (!ABV) bq[R][r] = wafer_ld_stream<NT>(reinterpret_cast<const VT *>(pb + (long long)z1 * g.plane + rowoff[r]));
        }
    } else {
#pragma unroll
        for (int q = 0; q < Cfg::CPL; ++q) {
            vq[R][q / VEC][q % VEC] = pv[(long long)z1 * g.plane + c_off[q]];
            if constexpr (!ABV) bq[R][q / VEC][q % VEC] = pb[(long long)z1 * g.plane + c_off[q]];
        }
    }
    // phi1 z-queue (main waves), planes z-2R .. z; starts empty (zeros never reach an output:
    // the first output plane zs is produced at z = zs + R, by when all 2R+1 entries are real)
    VT q1[2 * R + 1][RY];
#pragma unroll
    for (int m = 0; m <= 2 * R; ++m)
#pragma unroll
        for (int r = 0; r < RY; ++r) q1[m][r] = zero;
    // ABV: the a, b formed for step 1 at plane z serve step 2 at the same plane R iterations later -- for
    // ThreePoint / FivePoint.  SevenPoint's two seven-plane z-queues leave no room for an (R+1)-deep a, b
    // queue (64 VGPRs): step 2 forms them again from V (same expressions, same bits).
    // Only where storage and arithmetic types agree: with fp32 storage and fp64 arithmetic a queue of storage type would
    // hand step 2 fp32-rounded a, b where the single-step kernel (and step 1 here) use the fp64 values -- the two kernels'
    // results would differ in the last fp32 bits (found by bench.py's cross-kernel check on the fp32 row, round 3) -- and a
    // queue of arithmetic type spills 176 B/lane there: that combination forms a, b again at step 2.
    constexpr bool CARRY_AB = R < 3 && std::is_same<T, C>::value;
    VT caq[CARRY_AB ? R + 1 : 1][RY], cbq[CARRY_AB ? R + 1 : 1][RY];
#pragma unroll
    for (int m = 0; m <= (CARRY_AB ? R : 0); ++m)
#pragma unroll
        for (int r = 0; r < RY; ++r) caq[m][r] = cbq[m][r] = zero;

    // LDS: zero the phi0 pad columns once (they stay zero), stage the centre plane z1
    for (int i = tid; i < 2 * Cfg::TILE0; i += Cfg::NT_) lds0[i] = T(0);
    for (int i = tid; i < Cfg::NB1 * Cfg::TILE1; i += Cfg::NT_) lds1[i] = T(0);
    __syncthreads();
    {
        T *t0 = lds0 + (z1 & 1) * Cfg::TILE0;
        if (!is_hcol) {
#pragma unroll
            for (int r = 0; r < RY; ++r)
                *reinterpret_cast<VT *>(t0 + (yrow[r] - (y0 - 2 * R)) * LP0 + HX0 + xl) = q0[R][r];
        }
#pragma unroll
        for (int q = 0; q < Cfg::OPW; ++q)
            if (is_main && wave + q * Cfg::NW2 < Cfg::OUTER)
                *reinterpret_cast<VT *>(t0 + orow_lds[q]) = *reinterpret_cast<const VT *>(phi + (long long)z1 * g.plane + orow_off[q]);
#pragma unroll
        for (int q = 0; q < Cfg::CPL; ++q)
            if (is_hcol && lane + q * 64 < Cfg::NCOL) t0[c_lds0[q]] = c_out[q] ? T(0) : q0[R][q / VEC][q % VEC];
    }
    VT orow_nxt[Cfg::OPW];
#pragma unroll
    for (int q = 0; q < Cfg::OPW; ++q) {
        orow_nxt[q] = zero;
        if (is_main && wave + q * Cfg::NW2 < Cfg::OUTER)
            orow_nxt[q] = *reinterpret_cast<const VT *>(phi + (long long)(z1 + 1) * g.plane + orow_off[q]);
    }
    __syncthreads();

    const int zend = ze + R; // phi1 planes z1 .. zend-1
    for (int z = z1; z < zend; ++z) {
        const bool more = z + 1 < zend;
        const long long zo = (long long)z * g.plane;
        // ---- 1. prefetch: phi0 plane z+R+1, V plane z+1, outer halo rows of plane z+2 ---------------
        VT pre[RY], pre_v[RY], pre_b[RY], orow_pre[Cfg::OPW];
#pragma unroll
        for (int r = 0; r < RY; ++r) pre[r] = pre_v[r] = pre_b[r] = zero;
#pragma unroll
        for (int q = 0; q < Cfg::OPW; ++q) orow_pre[q] = zero;
        // (no bounds tests: one plane past the last one needed still lies in the guard zone; the
        //  role tests are wave-uniform)
        if (!is_hcol) {
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                pre[r] = *reinterpret_cast<const VT *>(phi + zo + (long long)(R + 1) * g.plane + rowoff[r]);
                pre_v[r] = wafer_ld_stream<NT>(reinterpret_cast<const VT *>(pv + zo + g.plane + rowoff[r]));
                if constexpr (!ABV) pre_b[r] = wafer_ld_stream<NT>(reinterpret_cast<const VT *>(pb + zo + g.plane + rowoff[r]));
            }
#pragma unroll
            for (int q = 0; q < Cfg::OPW; ++q)
                if (is_main && wave + q * Cfg::NW2 < Cfg::OUTER)
                    orow_pre[q] = *reinterpret_cast<const VT *>(phi + zo + 2 * g.plane + orow_off[q]);
        } else {
#pragma unroll
            for (int q = 0; q < Cfg::CPL; ++q) {
                pre[q / VEC][q % VEC] = phi[zo + (long long)(R + 1) * g.plane + c_off[q]];
                pre_v[q / VEC][q % VEC] = pv[zo + g.plane + c_off[q]];
                if constexpr (!ABV) pre_b[q / VEC][q % VEC] = pb[zo + g.plane + c_off[q]];
            }
        }
        // ---- 2. stage phi0 plane z+1 into the other buffer ------------------------------------------------
        if (more) {
            T *nt = lds0 + ((z + 1) & 1) * Cfg::TILE0;
            if (!is_hcol) {
#pragma unroll
                for (int r = 0; r < RY; ++r)
                    *reinterpret_cast<VT *>(nt + (yrow[r] - (y0 - 2 * R)) * LP0 + HX0 + xl) = q0[R + 1][r];
            }
#pragma unroll
            for (int q = 0; q < Cfg::OPW; ++q)
                if (is_main && wave + q * Cfg::NW2 < Cfg::OUTER) *reinterpret_cast<VT *>(nt + orow_lds[q]) = orow_nxt[q];
#pragma unroll
            for (int q = 0; q < Cfg::CPL; ++q)
                if (is_hcol && lane + q * 64 < Cfg::NCOL) nt[c_lds0[q]] = c_out[q] ? T(0) : q0[R + 1][q / VEC][q % VEC];
        }
        // ---- 3. step 1: phi1 plane z ---------------------------------------------------------------------------
        const T *c0 = lds0 + (z & 1) * Cfg::TILE0;
        T *w1 = lds1 + (((z % Cfg::NB1) + Cfg::NB1) % Cfg::NB1) * Cfg::TILE1;
        const bool wplane = work_plane(z);
        VT p1new[RY];
        if (!is_hcol) {
            // INTERIOR: the plane and every row of this wave are work cells (the common case) -- no
            // wave-uniform tests are left inside, so the RY x VEC updates form ONE basic block and
            // their division chains interleave; the general form tests per row.
            auto step1 = [&](auto interior_tag, auto adjacent_tag) {
                constexpr bool INTERIOR = decltype(interior_tag)::value;
                constexpr bool ADJ = decltype(adjacent_tag)::value; // the wave's RY rows are neighbours (main waves; the halo-row waves' are not)
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    VT res = zero;
                    if (INTERIOR || (wplane && rowwk[r])) {
                        const int ly = yrow[r] - (y0 - 2 * R);
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            C xs[2 * R + 1], ys[2 * R + 1], zz[2 * R + 1];
                            const C w = (C)q0[R][r][v];
#pragma unroll
                            for (int d = -R; d <= R; ++d) {
                                zz[d + R] = (C)q0[R + d][r][v];
                                if (d == 0) {
                                    xs[R] = w;
                                    ys[R] = w;
                                } else {
                                    xs[d + R] = (v + d >= 0 && v + d < VEC) ? (C)q0[R][r][(v + d + VEC) % VEC]
                                                                           : (C)c0[ly * LP0 + HX0 + xl + v + d];
                                    ys[d + R] = (YR && ADJ && r + d >= 0 && r + d < RY) ? (C)q0[R][(r + d + RY) % RY][v]
                                                                                 : (C)c0[(ly + d) * LP0 + HX0 + xl + v];
                                }
                            }
                            const C S = wafer_stencil_sum<C, R>(xs, ys, zz, w);
                            T rs;
                            if constexpr (ABV) {
                                C ca, cb;
                                wafer_ab_from_v<C>((C)vq[R][r][v], dt, vir, ca, cb);
                                if constexpr (CARRY_AB) {
                                    caq[R][r][v] = (T)ca;
                                    cbq[R][r][v] = (T)cb;
                                }
                                rs = (T)wafer_update<C>(w, ca, cb, dt, S, den);
                            } else rs = (T)wafer_update<C>(w, (C)vq[R][r][v], (C)bq[R][r][v], dt, S, den);
                            res[v] = (xi + v < g.nx) ? rs : T(0);
                        }
                    }
                    p1new[r] = res;
                    *reinterpret_cast<VT *>(w1 + (yrow[r] - (y0 - R)) * LP1 + HX1 + xl) = res;
                }
            };
            bool all_rows = wplane;
#pragma unroll
            for (int r = 0; r < RY; ++r) all_rows = all_rows && rowwk[r];
            if (is_main) {
                if (all_rows) step1(std::true_type{}, std::true_type{});
                else step1(std::false_type{}, std::true_type{});
            } else {
                if (all_rows) step1(std::true_type{}, std::false_type{});
                else step1(std::false_type{}, std::false_type{});
            }
        } else {
#pragma unroll
            for (int q = 0; q < Cfg::CPL; ++q) {
                if (lane + q * 64 < Cfg::NCOL) {
                    const int cidx = lane + q * 64;
                    const int row = cidx / (2 * Cfg::HC0), k = cidx % (2 * Cfg::HC0);
                    const int kk = (k < Cfg::HC0) ? k : k - Cfg::HC0;
                    if (kk < R && row >= R && row < Cfg::ROWS0 - R) {
                        T rs = T(0);
                        if (wplane && c_p1[q]) {
                            const int o0 = c_lds0[q];
                            C xs[2 * R + 1], ys[2 * R + 1], zz[2 * R + 1];
                            const C w = (C)q0[R][q / VEC][q % VEC];
#pragma unroll
                            for (int d = -R; d <= R; ++d) {
                                zz[d + R] = (C)q0[R + d][q / VEC][q % VEC];
                                xs[d + R] = (d == 0) ? w : (C)c0[o0 + d];
                                ys[d + R] = (d == 0) ? w : (C)c0[o0 + d * LP0];
                            }
                            const C S = wafer_stencil_sum<C, R>(xs, ys, zz, w);
                            if constexpr (ABV) rs = (T)wafer_update_v<C>(w, (C)vq[R][q / VEC][q % VEC], dt, S, den, vir);
                            else rs = (T)wafer_update<C>(w, (C)vq[R][q / VEC][q % VEC], (C)bq[R][q / VEC][q % VEC], dt, S, den);
                        }
                        w1[c_lds1[q]] = rs;
                    }
                }
            }
        }
        (void)c_p1slot_valid;
        // the phi1 tile of plane z-R (written R iterations ago) is complete: barriers in between
        // ---- 4. step 2: phi2 plane z-R from the phi1 queue (main waves) --------------------------------------
        if (is_main) {
#pragma unroll
            for (int m = 0; m < 2 * R; ++m)
#pragma unroll
                for (int r = 0; r < RY; ++r) q1[m][r] = q1[m + 1][r];
#pragma unroll
            for (int r = 0; r < RY; ++r) q1[2 * R][r] = p1new[r];
            const int zo2 = z - R;
            if (zo2 >= zs) {
                const T *c1 = lds1 + (((zo2 % Cfg::NB1) + Cfg::NB1) % Cfg::NB1) * Cfg::TILE1;
                auto step2 = [&](auto interior_tag) {
                    constexpr bool INTERIOR = decltype(interior_tag)::value;
                    VT res2[RY];
#pragma unroll
                    for (int r = 0; r < RY; ++r) {
                        res2[r] = zero;
                        if (INTERIOR || rowwk[r]) {
                            const int ly = yrow[r] - (y0 - R);
#pragma unroll
                            for (int v = 0; v < VEC; ++v) {
                                C xs[2 * R + 1], ys[2 * R + 1], zz[2 * R + 1];
                                const C w = (C)q1[R][r][v];
#pragma unroll
                                for (int d = -R; d <= R; ++d) {
                                    zz[d + R] = (C)q1[R + d][r][v];
                                    if (d == 0) {
                                        xs[R] = w;
                                        ys[R] = w;
                                    } else {
                                        xs[d + R] = (v + d >= 0 && v + d < VEC) ? (C)q1[R][r][(v + d + VEC) % VEC]
                                                                               : (C)c1[ly * LP1 + HX1 + xl + v + d];
                                        ys[d + R] = (r + d >= 0 && r + d < RY) ? (C)q1[R][(r + d + RY) % RY][v]
                                                                             : (C)c1[(ly + d) * LP1 + HX1 + xl + v];
                                    }
                                }
                                const C S = wafer_stencil_sum<C, R>(xs, ys, zz, w);
                                if constexpr (ABV && CARRY_AB) res2[r][v] = (T)wafer_update<C>(w, (C)caq[0][r][v], (C)cbq[0][r][v], dt, S, den);
                                else if constexpr (ABV) {
                                    C ca, cb;
                                    wafer_ab_from_v<C>((C)vq[0][r][v], dt, vir, ca, cb);
                                    res2[r][v] = (T)wafer_update<C>(w, ca, cb, dt, S, den);
                                } else res2[r][v] = (T)wafer_update<C>(w, (C)vq[0][r][v], (C)bq[0][r][v], dt, S, den);
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < RY; ++r) {
                        if (INTERIOR || rowwk[r]) {
                            T *dst = out + (long long)zo2 * g.plane + rowoff[r];
                            if (xi + VEC <= g.nx) {
                                wafer_st_stream<NT>(reinterpret_cast<VT *>(dst), res2[r]);
                            } else {
#pragma unroll
                                for (int v = 0; v < VEC; ++v)
                                    if (xi + v < g.nx) dst[v] = res2[r][v];
                            }
                        }
                    }
                };
                bool all_rows2 = true;
#pragma unroll
                for (int r = 0; r < RY; ++r) all_rows2 = all_rows2 && rowwk[r];
                if (all_rows2) step2(std::true_type{});
                else step2(std::false_type{});
            }
        }
        __syncthreads();
        // ---- 5. rotate ---------------------------------------------------------------------------------------------
#pragma unroll
        for (int m = 0; m < 2 * R; ++m) {
#pragma unroll
            for (int r = 0; r < RY; ++r) q0[m][r] = q0[m + 1][r];
        }
#pragma unroll
        for (int r = 0; r < RY; ++r) q0[2 * R][r] = pre[r];
#pragma unroll
        for (int m = 0; m < R; ++m)
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                vq[m][r] = vq[m + 1][r];
                if constexpr (!ABV) bq[m][r] = bq[m + 1][r];
            }
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            vq[R][r] = pre_v[r];
            if constexpr (!ABV) bq[R][r] = pre_b[r];
        }
        if constexpr (ABV && CARRY_AB) {
#pragma unroll
            for (int m = 0; m < R; ++m)
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    caq[m][r] = caq[m + 1][r];
                    cbq[m][r] = cbq[m + 1][r];
                }
        }
#pragma unroll
        for (int q = 0; q < Cfg::OPW; ++q) orow_nxt[q] = orow_pre[q];
    }
}

// planes per workgroup / launch size for the fused kernel
template <typename T, int R, int NW2>
static inline int wafer_f2_zchunk(const WaferTuning &t, const WaferGeom &g, int nplanes, int target_blocks)
{
    using Cfg = WaferF2Cfg<T, R, NW2>;
    if (t.zchunk > 0) return t.zchunk;
    if (target_blocks < 0) return -target_blocks < nplanes ? -target_blocks : nplanes; // the caller fixed the chunk length
    const long long per_layer = (long long)((g.nx + Cfg::TX - 1) / Cfg::TX) * ((g.ny + Cfg::TY - 1) / Cfg::TY);
    const long long target = t.target_blocks > 0 ? t.target_blocks : (target_blocks > 0 ? target_blocks : 256);
    return wafer_pick_zchunk(per_layer, nplanes, target, 2 * R + 3);
}

// Advances planes [lz_lo, lz_hi) by TWO steps: out = step(step(phi)).
template <typename T, typename C, int R, int NW2>
static inline hipError_t wafer_launch_step2_fused_nw(const WaferTuning &t, WaferStepArgs a, const WaferLdsOpts &o, const T *phi,
                                                     const T *pa, const T *pb, const T *pv, T *out, hipStream_t s)
{
    using Cfg = WaferF2Cfg<T, R, NW2>;
    const WaferGeom &g = a.g;
    const int ntx = (g.nx + Cfg::TX - 1) / Cfg::TX;
    const int nty = (g.ny + Cfg::TY - 1) / Cfg::TY;
    int swz = o.swz;
    long long nblocks;
    if (a.nsub > 1 && a.lz_hi - a.lz_lo < 8 * a.nsub) a.nsub = 0; // too thin to cut
    if (a.nsub > 1) { // mixed launch (see WaferStepArgs)
        const int nplanes = a.lz_hi - a.lz_lo, ntiles = ntx * nty;
        const int nshort_tiles = ntiles / 16 > 0 ? ntiles / 16 : 1;
        a.n_long = ntiles - nshort_tiles;
        a.zchunk = (nplanes + a.nsub - 1) / a.nsub;
        nblocks = a.n_long + (long long)nshort_tiles * a.nsub;
        swz = 0; // the hardware's dispatch order is the point
    } else {
        a.zchunk = wafer_f2_zchunk<T, R, NW2>(t, g, a.lz_hi - a.lz_lo, a.target_blocks);
        nblocks = (long long)ntx * nty * ((a.lz_hi - a.lz_lo + a.zchunk - 1) / a.zchunk);
    }
    const dim3 grid((unsigned)nblocks), block(Cfg::NT_);
    const bool vir = a.v_in_range != 0;
    // One flavour per (types, ext, tile height): ordinary (cache-retaining) loads, a and b formed from V in registers.  The
    // flavours that streamed a and b or used non-temporal loads were measured slower in round 1 (168+ VGPRs and scratch at
    // ext 1) and had no test of their own; they are gone (96 -> 24 instantiations of this kernel in the library).
    // YR (register y neighbours in step 1): same-box A/B at 512^3: FivePoint fp64 0.4390 against 0.4439 ms/step with it;
    // ThreePoint fp64 no change; fp32 storage 0.2674 against 0.2600 WITHOUT it -- so: on for FivePoint fp64 only.
    constexpr bool YR = R == 2 && std::is_same<T, double>::value;
    if (vir) hipLaunchKernelGGL((wafer_k_step2_fused<T, C, R, false, true, NW2, true, YR>), grid, block, (size_t)o.pad, s, a, ntx, nty, swz, phi, pv, pb, out);
    else hipLaunchKernelGGL((wafer_k_step2_fused<T, C, R, false, true, NW2, false, YR>), grid, block, (size_t)o.pad, s, a, ntx, nty, swz, phi, pv, pb, out);
    return hipGetLastError();
}

template <typename T, typename C, int R>
static inline hipError_t wafer_launch_step2_fused(const WaferTuning &t, WaferStepArgs a, const T *phi, const T *pa, const T *pb,
                                                  const T *pv, T *out, hipStream_t s)
{
    WaferLdsOpts o = wafer_lds_opts(t);
    // Defaults measured at 512^3 fp64 (profiles/r01_sweep_f_512_fused.jsonl):
    //   ext 1: 128x16 tiles (8 main waves + halo-row + halo-column wave = 640 threads), a and b
    //          formed from V -- 0.369 ms/step; the taller tile halves the halo-row overhead
    //          (phi0 20/16, V 18/16 rows per 16) and 147 VGPRs still fit 10 waves per CU.
    //          Streaming a, b instead needs 168+ VGPRs at that size and spills.
    //   ext 2: 128x8 tiles, a and b formed from V as well (0.53 ms/step against 0.66 with a, b
    //          streamed and 0.58 for the single-step kernel, since wafer_recip shortened b's reciprocal).
    // Ordinary (cache-retaining) loads: the halo-row wave re-reads rows its neighbour tile streams.
    const int nw2 = (R == 1 && a.g.ny >= 16) ? 8 : 4;
    if (o.abv < 0) o.abv = 1;
    if (t.nt < 0) o.nt = 0;
    if constexpr (R == 1) {
        if (nw2 == 8) return wafer_launch_step2_fused_nw<T, C, R, 8>(t, a, o, phi, pa, pb, pv, out, s);
    }
    return wafer_launch_step2_fused_nw<T, C, R, 4>(t, a, o, phi, pa, pb, pv, out, s);
}
