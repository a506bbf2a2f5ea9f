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
// Eight waves, two per SIMD, 241 VGPRs, no scratch.  Every wave owns two rows of the tile at all three levels plus ONE extra slot:
//   wave 0   row y0-1   (phi1 and phi2)          wave 7   row y0+16  (phi1 and phi2)
//   wave 1   row y0-2   (phi1), stages row y0-3  wave 6   row y0+17  (phi1), stages row y0+18
//   waves 2..5   33 of the 132 phi0 halo-column cells each, one per lane (z-queues in component 0 of the extra slot's
//                registers): phi1 on the inner two columns, phi2 on the innermost
// so that every global access stays 128-byte aligned and tiles need no overlap.
//
// a and b of a cell (potential.rs:104-110) are formed ONCE per pass, at level 1, and ride in registers to levels 2 and 3 as
// a and b * dt (b enters the update only through that product): 29 + 16 + 16 fp64 instructions per cell and pass instead of
// 3 x 29 -- the reciprocal sequence is 12 of the 29.  (Round 2's kernel had twelve waves -- eight main, four helpers, three
// per SIMD, 168-VGPR cap -- and formed a, b at every level because the registers to carry them did not exist there: 46.4
// VALU instructions per update against 36.6 here; 0.284 against 0.262 ms/step at 512^3 on one box: profiles/NOTES.md.)
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
#include "wafer_storage.h"

template <typename T> struct WaferF3Vec : WaferVec<T> {};   // cells per lane: 16 bytes of x
// Diagnostic builds (never timed as the product, never shipped): -DWAFER_DIAG=<bits>
//   1  in-kernel stamps (cdna_hip_programming.md, "In-kernel stamps"): one workgroup adds up, per wave, the shader-clock cycles
//      between fixed points of the plane iteration and leaves the sums in a buffer of their own (wafer_debug_f3_stamps reads it;
//      tools/f3_stamps.py prints the shares)
//   2  every prefetch asks for the column's first planes again (cache hits)      4  nothing is stored      8  no barrier in the plane loop
// What was built with further switches, measured and rejected in rounds 3 and 4 (x neighbours by DPP wave shifts, one cell per lane with
// two workgroups per CU, ring queues marching down / in the peer instantiation, other request placements, no issue priorities) is in
// git history and profiles/NOTES.md, not here.
#ifndef WAFER_DIAG
#define WAFER_DIAG 0
#endif
#if WAFER_DIAG & 1
enum { WAFER_F3_NSTAMP = 8 };
__device__ unsigned long long wafer_f3_stamp_buf[8 * WAFER_F3_NSTAMP];
#define WAFER_F3_STAMP_AT(k)                                                                        \
    do {                                                                                            \
        unsigned long long t_;                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        stamp_sum[k] += t_ - stamp_last;                                                            \
        stamp_last = t_;                                                                            \
    } while (0)
#else
#define WAFER_F3_STAMP_AT(k) do { } while (0)
#endif
// A wave's issue priority falls as it advances through the plane iteration (s_setprio 3 at the top, 2 behind level 1 of the main
// rows, 1 behind level 1 of the extra slot, 0 behind level 2): of the two waves that share a vector pipe the one that is BEHIND
// gets it.  Left to the hardware's oldest-first arbitration the first wave of each SIMD ran ahead and then waited a quarter of
// the iteration at the barrier while the second finished alone (tools/f3_stamps.py): -3 % (profiles/r04_ab_f3_priority.jsonl;
// flatter schedules gain less, the reverse order loses).
#define WAFER_F3_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
template <typename T>
struct WaferF3Cfg {
    static constexpr int VEC = WaferF3Vec<T>::N;
    static constexpr int RY = 2;
    static constexpr int NW = 8;                        // waves: tile height 16, two rows per wave (+ one extra slot each)
    static constexpr int NT_ = NW * 64;
    static constexpr int TX = 64 * VEC, TY = NW * RY;
    static constexpr int HC0 = 3, HC1 = 2, HC2 = 1;     // halo columns per side of phi0 / phi1 / phi2
    static constexpr int HX0 = ((HC0 + VEC - 1) / VEC) * VEC;
    static constexpr int HX1 = ((HC1 + VEC - 1) / VEC) * VEC;
    static constexpr int HX2 = ((HC2 + VEC - 1) / VEC) * VEC;
    static constexpr int LP0 = TX + 2 * HX0, LP1 = TX + 2 * HX1, LP2 = TX + 2 * HX2;
    static constexpr int ROWS0 = TY + 6, ROWS1 = TY + 4, ROWS2 = TY + 2;
    static constexpr int TILE0 = ROWS0 * LP0, TILE1 = ROWS1 * LP1, TILE2 = ROWS2 * LP2;
    static constexpr int NCOL = 2 * HC0 * ROWS0;        // phi0 halo-column cells per plane
    static constexpr int HCW0 = 2, HCWN = 4;            // waves HCW0 .. HCW0 + HCWN - 1 take the halo-column cells,
    static constexpr int CPW = (NCOL + HCWN - 1) / HCWN; // one per lane: 33 each
    static_assert(CPW <= 64, "one halo-column cell per lane");
};

// One workgroup's assignment.  32 bytes, read with scalar loads.
struct WaferF3Block {
    int tile;        // ty * ntx + tx
    int zs, ze;      // output planes [zs, ze) (local plane indices)
    int down;        // bit 0: march from ze-1 down to zs.  Peer-store whole-column passes (wafer_f3_schedule_whole) also carry
                     // bits 8-9: 1 + the ghost side to wait for BEFORE the prologue (the side the march starts at), bits 16-17: 1 + the
                     // side whose neighbour gets the FIRST wt planes of the march (stored, then counted, early in the column)
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
    unsigned max_spins = 1u << 24;   // bound of a ghost-flag wait, ~1 us per spin (WAFER_HV_WAIT_MS, default 20 s)
    // Peer stores (wafer_set_overlap mode 3): a boundary workgroup of half h stores the planes the exchange would send ALSO into
    // the z-neighbour's ghost planes -- peer_out[h] is the neighbour's output buffer as mapped here (same process: its pointer;
    // another process: through HIP IPC), plane z of this slab is plane z + peer_zshift[h] there -- and then adds 1 to the
    // neighbour's arrival counter peer_flag[h] (system scope).  flag[] then counts ARRIVED WORKGROUPS, need[] = workgroups per
    // pass x passes so far, and both are accessed at system scope.  No gate kernel, no exchange kernel, no second stream.
    // The three per-side values live in DEVICE memory (WaferF3Peer, written by wafer_peer_connect) and are read inside the few
    // iterations that use them: as kernel arguments they stayed live in scalar registers through the whole plane loop, which
    // cost the kernel 8 % (0.2840 against 0.2506 ms/step at the bench slab, the same effect that made the peer paths a
    // separate instantiation).
    const struct WaferF3Peer *peer_dev = nullptr;
    int peer_buf = 0;   // which of the neighbour's two buffers this pass writes (its output buffer: the same index as mine)
    int peer = 0;
};
struct WaferF3Peer {
    void *out[2][2];                 // [side][buffer]: the neighbour's phi buffers (nullptr: no neighbour on that side)
    long long zshift[2];
    unsigned long long *flag[2];
};
enum { WAFER_F3_SYNC_STRIDE = 8 }; // 64-bit words between the two counters / flags

// Returns true when the wait gave up (the whole workgroup sees the same answer).  The bound is generous -- a legitimate
// wait is well under a millisecond inside a run of passes, but the FIRST pass of a wafer_evolve call waits for a neighbour
// that may still be busy with host work between calls (file output, a table upload, RCCL channel set-up) -- and a
// workgroup that gives up poisons everything it still stores (NaN), so that results built from stale ghost planes cannot
// be mistaken for an answer; the host reports WAFER_ERR_COMM at its next synchronisation.
__device__ __forceinline__ bool wafer_f3_wait(const volatile WaferF3Sync *sy, int idx, int tid)
{
    __shared__ unsigned gave_up;
    if (tid == 0) {
        unsigned spins = 0, bad = 0;
        const unsigned long long *fw = const_cast<const unsigned long long *>(sy->flag) + idx * WAFER_F3_SYNC_STRIDE;
        const unsigned long long need = sy->need[idx];
        const unsigned max_spins = sy->max_spins;
        const bool peer = sy->peer != 0;
        // (peer mode: the word is written by another device, or another process on this one: system scope)
        while ((peer ? __hip_atomic_load(fw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                     : __hip_atomic_load(fw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < need) {
            __builtin_amdgcn_s_sleep(32);
            if (++spins > max_spins) { // the exchange never arrived
                __hip_atomic_store(const_cast<unsigned *>(sy->err), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                bad = 1;
                break;
            }
        }
        gave_up = bad;
        // system scope: the ghost planes were written by another kernel, possibly (through the fabric) of another device
        if (!(sy->debug & 4)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(gave_up) != 0;
}

// The kernel's arguments as they lie in the kernarg segment (by-value aggregates follow the C layout rules).  The
// synchronising instantiations read WaferF3Sync THERE, at the two or three places that need it, instead of through the
// parameter: as a parameter its sixteen scalars are live across the plane loop of a kernel that is out of scalar registers
// (106 SGPRs; 42 of them spilled to vector lanes with mode 2's paths, 76 with the peer paths, 2 without either).
struct WaferF3KernArgs {
    WaferStepArgs a;
    int ntx;
    const WaferF3Block *table;
    WaferF3Sync sy;
    const void *phi, *pv;
    void *out;
};
__device__ __forceinline__ const volatile WaferF3Sync *wafer_f3_sync_in_kernarg()
{
    return &((const volatile WaferF3KernArgs *)__builtin_amdgcn_kernarg_segment_ptr())->sy;
}

// PEER: the instantiation that serves overlap mode 3 (peer stores, the early wait and the early count of whole-column passes).
// A separate instantiation because the mere presence of those paths costs the plain kernel 8 % (0.2789 against 0.2578 ms/step at
// 512^3, same box: more live scalars and a longer loop body around the stores).
// MODE: 0 = no synchronisation at all (undecomposed grids, unsplit passes: the benchmark's kernel) -- the table's wait / bump
// fields are ignored and WaferF3Sync is dead, which takes its sixteen scalars out of a kernel that spills scalar registers
// (106 SGPRs, 42 spilled to vector lanes with the sync paths compiled in); 1 = the single-launch pass of overlap mode 2;
// 2 = peer stores (overlap mode 3).
// XS ("exact stores"; plain launches over grids made of whole tiles): every plane iteration issues exactly two stores -- one full
// vector per main row -- on every path, so that the compiler's wait for the prefetched planes, behind the loop's barrier, is an
// exact count that leaves the stores in flight.  With the stores inside conditions (pipeline fill, ragged tiles) the wait-count
// pass has to assume the path without them, the last prefetch is then waited for with vmcnt(0), and in the steady state that
// makes every wave sit out the completion of the two stores it issued a few hundred cycles earlier, once per plane (the
// ablations of profiles/NOTES.md: the kernel without its stores 0.202 ms/step, without its loads 0.215, with both 0.253, without
// either 0.192).  While the pipeline fills, the two stores go to the column's first plane, which the first real store overwrites.
template <typename TS, typename C, bool VIR, bool DOWN, int MODE, bool XS = false, bool RING_T = false>
__device__ __forceinline__ void wafer_step3_body(const WaferStepArgs &a, const WaferF3Block &blk, int ntx, const WaferF3Sync &sy,
                                                  const typename WaferF3Store<TS>::S *__restrict__ phi, const typename WaferF3Store<TS>::S *__restrict__ pv,
                                                  typename WaferF3Store<TS>::S *__restrict__ out,
                                                  typename WaferF3Store<TS>::Q *lds0, typename WaferF3Store<TS>::Q *lds1, typename WaferF3Store<TS>::Q *lds2)
{
    using T = typename WaferF3Store<TS>::Q;   // queues, LDS rings, carried a / b * dt
    using ST = typename WaferF3Store<TS>::S;  // the arrays in HBM
    constexpr bool WIDE = !std::is_same<ST, T>::value;
    using Cfg = WaferF3Cfg<T>;
    using VT = typename WaferF3Vec<T>::type;
    typedef ST __attribute__((ext_vector_type(WaferF3Vec<T>::N))) SVT;   // a lane's request: the same cells, in the storage type
    // global loads of a lane's vector / of one cell, widened to the register type
    auto gload = [](const ST *p) -> VT { return wafer_f3_widen<SVT, VT, WaferF3Vec<T>::N>(*reinterpret_cast<const SVT *>(p)); };
    auto gload_raw = [](const ST *p) -> SVT { return *reinterpret_cast<const SVT *>(p); };
    // the result is streamed: nobody reads it before the next launch (the same box, 512^3 / 1024^3: 0.2150 -> 0.2067 / 1.98 -> 1.87
    // ms per step against plain stores; non-temporal LOADS of V or phi0 lose 4-12 %: the halo requests of the tiles next door want
    // those lines in the L2 -- profiles/r05_ab_f3_nontemporal.jsonl)
    auto gstore = [](ST *p, SVT v) { wafer_store_result(reinterpret_cast<SVT *>(p), v); };
    auto widen = [](const SVT &x) -> VT { return wafer_f3_widen<SVT, VT, WaferF3Vec<T>::N>(x); };
    // a level's result as the storage type holds it (fp32 storage: rounded once per step, like a store and a load would)
    auto as_stored = [](C x) -> T { return (T)(ST)x; };
    constexpr int R = 1;
    constexpr int VEC = Cfg::VEC, RY = Cfg::RY, TX = Cfg::TX, TY = Cfg::TY;
    constexpr int HX0 = Cfg::HX0, HX1 = Cfg::HX1, HX2 = Cfg::HX2, LP0 = Cfg::LP0, LP1 = Cfg::LP1, LP2 = Cfg::LP2;
    constexpr int SD = DOWN ? -1 : 1;
    constexpr int ZLO = DOWN ? 2 : 0, ZHI = DOWN ? 0 : 2;

    const WaferGeom &g = a.g;
    const int tx_i = blk.tile % ntx, ty_i = blk.tile / ntx;
    const int zs = blk.zs, ze = blk.ze;
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr bool PEER = MODE == 2, SYNC = MODE != 0;
    constexpr bool RING = RING_T;   // (a local name: the peer instantiation's middle segment shadows it, with PEER and SYNC)
    // (XS with MODE 1 -- overlap mode 2's single-launch pass -- only through the segmented loop below: its boundary planes are
    //  written through store by store in the tail segment, which runs the generic code)
    (void)sy;   // (never read through the parameter: see WaferF3KernArgs)
    [[maybe_unused]] const volatile WaferF3Sync *const syv = SYNC ? wafer_f3_sync_in_kernarg() : nullptr;
    const int wait_early = PEER ? ((blk.down >> 8) & 3) - 1 : -1, bump_early = PEER ? ((blk.down >> 16) & 3) - 1 : -1;
    bool poisoned = false;   // a ghost-flag wait gave up: everything stored from here on is NaN (wafer_f3_wait)
    // a whole-column pass starts at a ghost side: its planes are loaded by the prologue
    if constexpr (PEER) {
        if (wait_early >= 0) poisoned = wafer_f3_wait(syv, wait_early, tid);
    }
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = tx_i * TX, y0 = ty_i * TY;
    const C dt = (C)a.dt;
    constexpr bool vir = VIR;
    const WaferDen<C> den = wafer_den<C>(a, vir);
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
    // (a halo row above / below the work area -- frame and guard rows, zeros -- is not fetched either: the wave requests its
    //  own first row again and takes zeros)
    const bool xy_out = xy < 0 || xy >= g.ny;
    const long long xoff_row = xy_out ? rowoff[0] : (long long)(xy + R) * g.pitch + g.xoff + R + x0;
    // ---- outermost phi0 halo rows y0-3 / y0+18 (plain vector loads staged through LDS)
    const int oy = wave == 1 ? y0 - 3 : y0 + TY + 2;
    const bool oy_out = oy < 0 || oy >= g.ny;
    const long long orow_off = oy_out ? rowoff[0] : (long long)(oy + R) * g.pitch + g.xoff + R + x0;
    const int orow_lds = (oy - (y0 - 3)) * LP0 + HX0 + xl;
    // ---- halo-column cell of this lane (waves 2..5): cell c: row c / 6 of the phi0 tile, k = c % 6: k < 3: column x0-1-k,
    //      else column x0+TX+(k-3)
    const int cidx = min((wave - Cfg::HCW0) * Cfg::CPW + lane, Cfg::NCOL - 1);
    const int crow = cidx / (2 * Cfg::HC0), ck = cidx % (2 * Cfg::HC0);
    const int ckk = (ck < Cfg::HC0) ? ck : ck - Cfg::HC0;
    const int clc = (ck < Cfg::HC0) ? (-1 - ckk) : (TX + ckk);
    const int cxw = x0 + clc, cy = y0 - 3 + crow;
    const bool c_ok = !x_row && lane < Cfg::CPW && (wave - Cfg::HCW0) * Cfg::CPW + lane < Cfg::NCOL;
    const bool c_wk = cy >= 0 && cy < g.ny && cxw >= 0 && cxw < g.nx;
    const bool c_l1 = c_ok && ckk < Cfg::HC1 && crow >= 1 && crow < Cfg::ROWS0 - 1;
    const bool c_l2 = c_ok && ckk < Cfg::HC2 && crow >= 2 && crow < Cfg::ROWS0 - 2;
    // a cell left or right of the work area (the Dirichlet frame column and the pad cells behind it: zeros that no kernel
    // writes) is not fetched -- its 128-byte line holds nothing anybody else reads, so each such request was an HBM read of its
    // own, 44 + 40 lines (phi0, V) per plane and row of tiles, 6 % of this kernel's reads at 512^3.  The lane requests the tile's
    // own edge cell of that row instead (a line the row's owner requests in the same iteration) and phi0 becomes the zero it
    // stands for; V of such a cell is never used (c_wk).  (Found with the halo-attribution runs of profiles/NOTES.md, round 3.
    //  The same for a cell above / below the work area: the tile's own first / last row.)
    const bool c_xout = cxw < 0 || cxw >= g.nx || cy < 0 || cy >= g.ny;
    const long long c_off = (long long)((cy < 0 ? y0 : cy >= g.ny ? y0 + TY - 1 : cy) + R) * g.pitch + g.xoff + R +
                            ((cxw < 0 || cxw >= g.nx) ? (ck < Cfg::HC0 ? x0 : x0 + TX - 1) : cxw);
    const int c_lds0 = crow * LP0 + HX0 + clc, c_lds1 = (crow - 1) * LP1 + HX1 + clc, c_lds2 = (crow - 2) * LP2 + HX2 + clc;
    // per-lane element offsets of the extra slot's requests inside a plane (see the prefetch at the top of the plane loop)
    const long long xslot_off = x_row ? xoff_row + (long long)xlu : c_off;
    const long long orow_slot_off = has_orow ? orow_off + (long long)xlu : xslot_off;

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
        return as_stored(w * ca + wafer_div_invariant<C>(cbdt * S, den));
    };
    auto update_with = [&](C w, C ca, C cbdt, C S) -> T { return as_stored(w * ca + wafer_div_invariant<C>(cbdt * S, den)); };

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
        for (int r = 0; r < RY; ++r) q0[m][r] = gload((phi + po + rowoff[r]) + xlu);
        if (x_row) xq0[m] = gload((phi + po + xoff_row) + xlu);
        else xq0[m][0] = (T)phi[po + c_off];
    }
    {
        const long long po = (long long)z1 * g.plane;
#pragma unroll
        for (int r = 0; r < RY; ++r) vcur[r] = gload((pv + po + rowoff[r]) + xlu);
        if (x_row) xv = gload((pv + po + xoff_row) + xlu);
        else xv[0] = (T)pv[po + c_off];
    }
    for (int i = tid; i < 2 * Cfg::TILE0; i += Cfg::NT_) lds0[i] = T(0);
    for (int i = tid; i < 2 * Cfg::TILE1; i += Cfg::NT_) lds1[i] = T(0);
    for (int i = tid; i < 2 * Cfg::TILE2; i += Cfg::NT_) lds2[i] = T(0);
    __syncthreads();
    {
        T *t0 = lds0 + (z1 & 1) * Cfg::TILE0;
#pragma unroll
        for (int r = 0; r < RY; ++r) *reinterpret_cast<VT *>(t0 + (yrow[r] - (y0 - 3)) * LP0 + HX0 + xl) = q0[1][r];
        if (x_row) *reinterpret_cast<VT *>(t0 + (xy - (y0 - 3)) * LP0 + HX0 + xl) = xy_out ? zero : xq0[1];
        else if (c_ok) t0[c_lds0] = c_xout ? T(0) : xq0[1][0];
        if (has_orow) *reinterpret_cast<VT *>(t0 + orow_lds) = oy_out ? zero : gload((phi + (long long)z1 * g.plane + orow_off) + xlu);
    }
    VT orow_nxt = zero;
    if (has_orow) orow_nxt = gload((phi + (long long)(z1 + SD) * g.plane + orow_off) + xlu);
    __syncthreads();

    // Peer stores: where the first / last wt planes of the march go in the neighbours' buffers, read ONCE from the device copy
    // of the connection and held in VECTOR registers (the pin): as scalars they stay live across a plane loop that has none to
    // spare (6-8 % in round 4's first version), and read inside the loop the loads of a rarely taken branch make the wait-count
    // pass pessimistic about every prefetch in flight (0.325 against 0.273 ms/step at the bench slab).
    [[maybe_unused]] ST *peer_first = nullptr, *peer_last = nullptr;
    if constexpr (PEER) {
        const volatile WaferF3Peer *pi = syv->peer_dev;
        const int buf = syv->peer_buf;
        if (bump_early >= 0) {
            ST *const base = static_cast<ST *>(pi->out[bump_early][buf]);
            if (base) peer_first = base + pi->zshift[bump_early] * g.plane;
        }
        if (blk.bump >= 0) {
            ST *const base = static_cast<ST *>(pi->out[blk.bump & 1][buf]);
            if (base) peer_last = base + pi->zshift[blk.bump & 1] * g.plane;
        }
        asm volatile("" : "+v"(peer_first), "+v"(peer_last));
    }
    // The early arrival counter's address likewise: read inside the loop (two dependent volatile loads in a branch taken once per
    // column) it made the wait-count pass end every iteration's wait for the prefetched planes in vmcnt(0) -- the stores just
    // issued included.
    [[maybe_unused]] unsigned long long *flag_early = nullptr;
    if constexpr (PEER) {
        if (bump_early >= 0) flag_early = const_cast<const volatile WaferF3Peer *>(syv->peer_dev)->flag[bump_early];
        asm volatile("" : "+v"(flag_early));
    }
    const int niter = (ze - zs) + 4;
#if WAFER_DIAG & 1
    unsigned long long stamp_sum[WAFER_F3_NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
#endif
#define WAFER_F3_Q0(m) (RING ? ((m) + WAFER_F3_PH) % 3 : (m))
#define WAFER_F3_Q1(m) (RING ? ((m) + WAFER_F3_PH + 1) % 3 : (m))
    if constexpr (SYNC && XS) {
        // Peer-store passes, three segments.  The peer stores, the early count and the late wait all lie within the first
        // seven and the last five iterations of a column; everything between runs the body of the PLAIN kernel -- no peer path,
        // no wait, ring queues (a multiple of three iterations, so that the queues enter and leave it in their natural order)
        // -- whose waits the compiler can count exactly.  With the peer paths in every iteration the wait behind the barrier is
        // vmcnt(0) (the rare branches' volatile loads) and the ring does not fit (24 B of scratch, 55 scalar spills).  A
        // workgroup whose early wait gave up poisons every store: it stays on the generic body.
        // (overlap mode 2's pass has no early count and no early wait: no head segment)
        const int head_need = PEER ? 7 : 0;
        const int head_end = niter < head_need ? niter : head_need;
        int mid_len = niter - 5 - head_end;
        // (the middle never reaches the iteration whose requests first touch a ghost plane: the wait belongs to the tail)
        if (blk.wait_late >= 0 && blk.wait_it >= head_end && mid_len > blk.wait_it - head_end) mid_len = blk.wait_it - head_end;
        mid_len = (mid_len > 0 && !poisoned) ? mid_len / 3 * 3 : 0;
        {
            const int IT_BEGIN = 0, IT_END = head_end;
        for (int it = IT_BEGIN; it < IT_END; ++it) {
#define WAFER_F3_PH 0
#include "wafer_stencil_fused3_iter.inc.h"
#undef WAFER_F3_PH
        }
        }
        if (mid_len > 0) {
            // (shadow the function's: the included text reads these names; ring queues marching up only, see the kernel)
            constexpr bool PEER = false, SYNC = false, RING = !DOWN;
            const int IT_BEGIN = head_end, IT_END = head_end + mid_len;
            if constexpr (RING) {
                for (int it0 = IT_BEGIN; it0 < IT_END; it0 += 3) {
                    {
                        const int it = it0;
#define WAFER_F3_PH 0
#include "wafer_stencil_fused3_iter.inc.h"
#undef WAFER_F3_PH
                    }
                    {
                        const int it = it0 + 1;
#define WAFER_F3_PH 1
#include "wafer_stencil_fused3_iter.inc.h"
#undef WAFER_F3_PH
                    }
                    {
                        const int it = it0 + 2;
#define WAFER_F3_PH 2
#include "wafer_stencil_fused3_iter.inc.h"
#undef WAFER_F3_PH
                    }
                }
            } else {
                for (int it = IT_BEGIN; it < IT_END; ++it) {
#define WAFER_F3_PH 0
#include "wafer_stencil_fused3_iter.inc.h"
#undef WAFER_F3_PH
                }
            }
        }
        {
            // The tail waits for its ghost side ONCE, ahead of its loop -- an iteration or so earlier than the first request that
            // touches a ghost plane needs it, which costs nothing when the neighbour's planes arrived long ago, and keeps the
            // wait's volatile loads out of the loop, whose waits the compiler can then count (the wait-count pathology again).
            const int IT_BEGIN = head_end + mid_len, IT_END = niter;
            if (blk.wait_late >= 0 && blk.wait_it >= head_end) poisoned = wafer_f3_wait(syv, blk.wait_late, tid) || poisoned;
#define WAFER_F3_LATE_WAIT_HOISTED
        for (int it = IT_BEGIN; it < IT_END; ++it) {
#define WAFER_F3_PH 0
#include "wafer_stencil_fused3_iter.inc.h"
#undef WAFER_F3_PH
        }
#undef WAFER_F3_LATE_WAIT_HOISTED
        }
    } else if constexpr (RING) {
        const int IT_BEGIN = 0, IT_END = niter;
        for (int it0 = IT_BEGIN; it0 < IT_END; it0 += 3) {
            {
                const int it = it0;
#define WAFER_F3_PH 0
#include "wafer_stencil_fused3_iter.inc.h"
#undef WAFER_F3_PH
            }
            if (it0 + 1 >= IT_END) break;
            {
                const int it = it0 + 1;
#define WAFER_F3_PH 1
#include "wafer_stencil_fused3_iter.inc.h"
#undef WAFER_F3_PH
            }
            if (it0 + 2 >= IT_END) break;
            {
                const int it = it0 + 2;
#define WAFER_F3_PH 2
#include "wafer_stencil_fused3_iter.inc.h"
#undef WAFER_F3_PH
            }
        }
    } else {
        const int IT_BEGIN = 0, IT_END = niter;
        for (int it = IT_BEGIN; it < IT_END; ++it) {
#define WAFER_F3_PH 0
#include "wafer_stencil_fused3_iter.inc.h"
#undef WAFER_F3_PH
        }
    }
#undef WAFER_F3_Q0
#undef WAFER_F3_Q1
#if WAFER_DIAG & 1
    if (blockIdx.x == gridDim.x / 2 + 3 && lane == 0) {
#pragma unroll
        for (int k = 0; k < WAFER_F3_NSTAMP; ++k) wafer_f3_stamp_buf[wave * WAFER_F3_NSTAMP + k] = stamp_sum[k];
    }
#endif
    if (SYNC && blk.bump >= 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if constexpr (PEER) {
                // every wave's system-scope write-through stores have been acknowledged (vmcnt(0) + barrier), so the count that
                // follows in program order cannot overtake them; the consumer's poll is followed by an acquire (wafer_f3_wait).
                // (A release fence here writes back the whole L2 of the XCD once per workgroup: 0.2928 against %s ms/step at the
                //  bench slab -- the lesson of round 3's mode 2 again.)
                unsigned long long *const pf = const_cast<const volatile WaferF3Peer *>(syv->peer_dev)->flag[blk.bump & 1];
                if (pf) __hip_atomic_fetch_add(pf, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
                __hip_atomic_fetch_add(const_cast<unsigned long long *>(syv->cnt) + blk.bump * WAFER_F3_SYNC_STRIDE, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// DIR: 0 = the table holds workgroups of both marching directions, 1 = all up, 2 = all down.  A kernel that carries one copy of the
// plane loop instead of two gets the better register allocation (the peer instantiation: 88 B of scratch with both, none with one).
template <typename T, typename C, bool VIR, int MODE = 0, bool XS = false, int DIR = 0>
__global__ __launch_bounds__((WaferF3Cfg<typename WaferF3Store<T>::Q>::NT_)) void wafer_k_step3_fused(WaferStepArgs a, int ntx, const WaferF3Block *__restrict__ table,
                                                                              WaferF3Sync sy, const typename WaferF3Store<T>::S *__restrict__ phi,
                                                                              const typename WaferF3Store<T>::S *__restrict__ pv, typename WaferF3Store<T>::S *__restrict__ out)
{
    using Q = typename WaferF3Store<T>::Q;
    using Cfg = WaferF3Cfg<Q>;
    __shared__ __attribute__((aligned(16))) Q lds0[2 * Cfg::TILE0];
    __shared__ __attribute__((aligned(16))) Q lds1[2 * Cfg::TILE1];
    __shared__ __attribute__((aligned(16))) Q lds2[2 * Cfg::TILE2];
    const WaferF3Block blk = table[blockIdx.x];
    // (ring queues where the kernel has the registers for the unrolled loop: the plain instantiation with exact store counts)
    constexpr bool RING = XS && MODE == 0;
    // (marching down keeps the shifting queues: in that copy of the loop the ring version needs 36 B of scratch, and every reload
    //  from scratch is a vector-memory operation whose wait, vmcnt(0), also waits for every prefetch in flight -- 0.319 against
    //  0.220 ms/step at 512^3; marching up the ring version has no scratch operation inside the loop.  The synchronising
    //  instantiations take ring queues in the middle segment of their passes only: wafer_step3_body)
    if constexpr (DIR == 1) wafer_step3_body<T, C, VIR, false, MODE, XS, RING>(a, blk, ntx, sy, phi, pv, out, lds0, lds1, lds2);
    else if constexpr (DIR == 2) wafer_step3_body<T, C, VIR, true, MODE, XS, false>(a, blk, ntx, sy, phi, pv, out, lds0, lds1, lds2);
    else if (blk.down & 1) wafer_step3_body<T, C, VIR, true, MODE, XS, false>(a, blk, ntx, sy, phi, pv, out, lds0, lds1, lds2);
    else wafer_step3_body<T, C, VIR, false, MODE, XS, RING>(a, blk, ntx, sy, phi, pv, out, lds0, lds1, lds2);
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
static inline void wafer_f3_schedule_plain(std::vector<WaferF3Block> &out, int ntx, int nty, int lz_lo, int lz_hi, int zchunk, bool swz, bool down = false)
{
    const int nplanes = lz_hi - lz_lo, nch = (nplanes + zchunk - 1) / zchunk, n = ntx * nty * nch;
    out.resize((size_t)n);
    for (int b = 0; b < n; ++b) {
        const int id = swz ? wafer_f3_xcd_slot(b, n) : b;     // x fastest, then y, then z-chunk
        WaferF3Block k{};
        k.tile = id % (ntx * nty);
        k.zs = lz_lo + (id / (ntx * nty)) * zchunk;
        k.ze = k.zs + zchunk < lz_hi ? k.zs + zchunk : lz_hi;
        k.down = down ? 1 : 0;
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

// Peer-store passes without a cut (overlap mode 3 where there is at least a tile per CU): every tile's whole column [lo, hi) in
// ONE workgroup, all marching in the same direction, which alternates from pass to pass.  Marching up, a workgroup reads its lower
// ghost planes first (prologue: it waits for that side's arrivals before anything else) and its upper ghost planes last (the late
// wait of the halves schedule); it stores its lowest `depth` planes first -- into the lower neighbour's upper ghost planes as well,
// counted into that neighbour's arrival counter a few iterations into the column -- and its highest `depth` planes last.  The
// neighbour's next pass marches DOWN: it reads those upper ghost planes first, half a pass (the second round of workgroups) to a
// whole pass after they were stored, and its own lower ghost planes last.  Every wait precedes the stores to the same side's
// neighbour, and an arrival from a neighbour implies that its workgroups have read the ghost planes the next stores overwrite
// (they store boundary planes only after reading the ghost planes behind them): the ping-pong buffers need no other protection.
static inline void wafer_f3_schedule_whole(std::vector<WaferF3Block> &out, int ntx, int nty, int lo, int hi, int down, const bool need_wait[2],
                                           int depth, bool swz)
{
    const int n = ntx * nty;
    out.resize((size_t)n);
    const int first_side = down ? 1 : 0, last_side = down ? 0 : 1;
    for (int b = 0; b < n; ++b) {
        WaferF3Block k{};
        k.tile = swz ? wafer_f3_xcd_slot(b, n) : b;
        k.zs = lo; k.ze = hi;
        k.down = (down ? 1 : 0) | ((need_wait[first_side] ? 1 + first_side : 0) << 8) | ((1 + first_side) << 16);
        k.wait_late = need_wait[last_side] ? last_side : -1;
        k.wait_it = hi - lo;          // the first prefetch that touches the far side's ghost planes (either direction)
        k.bump = last_side;
        k.wt = depth;
        out[(size_t)b] = k;
    }
}

// planes per workgroup of the plain schedule: one workgroup per CU marching a long column (as the two-step kernel)
static inline int wafer_f3_zchunk(const WaferTuning &t, int ntx, int nty, int nplanes, int target_blocks)
{
    if (t.zchunk > 0) return t.zchunk;
    if (target_blocks < 0) return -target_blocks < nplanes ? -target_blocks : nplanes;
    const long long per_layer = (long long)ntx * nty;
    const long long target = t.target_blocks > 0 ? t.target_blocks : (target_blocks > 0 ? target_blocks : 256);
    const int zc = wafer_pick_zchunk(per_layer, nplanes, target, 6);   // four iterations of pipeline fill + the prologue
    // More workgroups than CUs (1024^3 on one GPU: 512 tiles): over columns of a thousand planes the workgroups of a round drift
    // apart, the second round starts staggered, and the halo rows of the tile next door are no longer in the XCD's L2 when a
    // workgroup asks for them (1.25 x the arrays in traffic against 1.08 at 512^3); how far apart depends on the box: 1.78 - 2.05
    // ms per step over the pool.  Columns of at most 384 planes, every round of CUs a launch of its own (launch_step3): 1.69 - 1.83
    // on the same boxes (profiles/r05_sweep_rounds_1024.jsonl).
    // Only where a layer of tiles is whole rounds of CUs (a launch per round ends on its slowest workgroup; rounds that do not fill
    // the chip are better served by one launch that hands a free CU the next column) and the columns are long (1024 x 1024 x 256:
    // 2.6 % slower by rounds).
    if (wafer_f3_by_rounds(t, per_layer, nplanes, target)) return wafer_pick_zchunk(per_layer, nplanes, target, 6, 384);
    return zc;
}

// which instantiation the last launch of this thread took (bench.py prints it and matches the committed counter figures by it)
struct WaferF3Instance {
    int tsize = 0, csize = 0;   // sizeof storage / arithmetic type (0: nothing launched yet)
    bool vir = false, xs = false;
    int mode = 0, dir = 0;
};
inline WaferF3Instance &wafer_f3_last_instance()
{
    static thread_local WaferF3Instance inst;
    return inst;
}

// Advances the planes of `table` (device copy, nblocks entries) by THREE steps: out = step(step(step(phi))).  ThreePoint only.
template <typename T, typename C>
static inline hipError_t wafer_launch_step3_fused(const WaferTuning &t, const WaferStepArgs &a, const WaferF3Block *table, int nblocks,
                                                  const WaferF3Sync &sy, const typename WaferF3Store<T>::S *phi, const typename WaferF3Store<T>::S *pv,
                                                  typename WaferF3Store<T>::S *out, hipStream_t s, int dir = 0)
{
    using Cfg = WaferF3Cfg<typename WaferF3Store<T>::Q>;
    const int ntx = (a.g.nx + Cfg::TX - 1) / Cfg::TX;
    const dim3 grid((unsigned)nblocks), block(Cfg::NT_);
    // the synchronisation a launch needs picks the instantiation (see wafer_step3_body): none, mode 2's flags and counters, peer stores
    const int mode = sy.peer ? 2 : (sy.flag != nullptr ? 1 : 0);
    // exact store counts (XS): plain launches over grids made of whole tiles (every store a full vector of work cells)
    const bool xs = t.f3_xs != 0 && a.g.nx % Cfg::TX == 0 && a.g.ny % Cfg::TY == 0;
#define WAFER_F3_LAUNCH3(VIR_, MODE_, XS_, DIR_)                                                                                                  \
    do {                                                                                                                                          \
        hipLaunchKernelGGL((wafer_k_step3_fused<T, C, VIR_, MODE_, XS_, DIR_>), grid, block, 0, s, a, ntx, table, sy, phi, pv, out); \
        WaferF3Instance &li_ = wafer_f3_last_instance();                                                                                          \
        li_.tsize = std::is_same<T, wafer_f32_wide>::value ? -4 : (int)sizeof(T); li_.csize = (int)sizeof(C); li_.vir = (VIR_); li_.mode = (MODE_); li_.xs = (XS_); li_.dir = (DIR_);           \
    } while (0)
    // (the single-direction kernels exist for the instantiations that need them: plain XS launches, peer-store passes)
#define WAFER_F3_LAUNCH(VIR_, MODE_, XS_)                                              \
    do {                                                                               \
        if ((XS_) && (MODE_) != 1 && dir == 1) WAFER_F3_LAUNCH3(VIR_, MODE_, XS_, ((XS_) && (MODE_) != 1) ? 1 : 0); \
        else if ((XS_) && (MODE_) != 1 && dir == 2) WAFER_F3_LAUNCH3(VIR_, MODE_, XS_, ((XS_) && (MODE_) != 1) ? 2 : 0); \
        else WAFER_F3_LAUNCH3(VIR_, MODE_, XS_, 0);                                    \
    } while (0)
    if (a.v_in_range != 0) {
        if (mode == 2 && xs) WAFER_F3_LAUNCH(true, 2, true);
        else if (mode == 2) WAFER_F3_LAUNCH(true, 2, false);
        else if (mode == 1 && xs) WAFER_F3_LAUNCH(true, 1, true);
        else if (mode == 1) WAFER_F3_LAUNCH(true, 1, false);
        else if (xs) WAFER_F3_LAUNCH(true, 0, true);
        else WAFER_F3_LAUNCH(true, 0, false);
    } else {
        if (mode == 2 && xs) WAFER_F3_LAUNCH(false, 2, true);
        else if (mode == 2) WAFER_F3_LAUNCH(false, 2, false);
        else if (mode == 1 && xs) WAFER_F3_LAUNCH(false, 1, true);
        else if (mode == 1) WAFER_F3_LAUNCH(false, 1, false);
        else if (xs) WAFER_F3_LAUNCH(false, 0, true);
        else WAFER_F3_LAUNCH(false, 0, false);
    }
#undef WAFER_F3_LAUNCH
#undef WAFER_F3_LAUNCH3
    return hipGetLastError();
}
