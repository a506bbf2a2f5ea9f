// Tuning knobs of the launch layer, read ONCE per context.
//
// wafer_ctx_create fills a WaferTuning from the WAFER_* environment variables (tools/stencil_sweep.py
// and the tests drive A/B runs through them) and every launcher takes it by reference: nothing on a
// launch path -- wafer_evolve, wafer_observables, the per-pass loop -- calls getenv.  A value of 0 / -1
// means "the kernel's own default" unless stated otherwise.
//
// Retired in round 6 (nothing set them any more; what they measured is in profiles/NOTES.md): WAFER_LDS_PAD, WAFER_LDS_NW,
// WAFER_SEVEN_VG, WAFER_F2_NW2, WAFER_HV_SHORT_TILES, WAFER_MAILBOX_WAIT_SPINS.
#pragma once
#include <cstdlib>

struct WaferTuning {
    // launch geometry of the column-marching kernels
    int zchunk = 0;         // WAFER_ZCHUNK: planes per workgroup (0: from the CU count)
    int target_blocks = 0;  // WAFER_TARGET_BLOCKS: workgroups per launch (0: from the CU count)
    int swz = 1;            // WAFER_XCD_SWIZZLE: XCD-aware workgroup -> tile map
    // single-step LDS kernel
    int lds_ry = 0;         // WAFER_LDS_RY: rows per lane, 2 or 4 (0: default per stencil order)
    int nt = -1;            // WAFER_NT: non-temporal streams (-1: default per kernel)
    int abv = -1;           // WAFER_ABV: 0 = stream the stored a, b arrays instead of forming them from V
    // excited-state step kernels
    int xf_nw = 0;          // WAFER_XF_NW: excited-state step kernels forced onto 8 waves / 128 x 16 tiles or 4 waves / 128 x 8 (0: by stencil, storage type and number of stored states, wafer_excited_nw)
    int xf_deep = 1;        // WAFER_XF_DEEP: the raw staging pipeline
    int one_pass = 1;       // WAFER_ONE_PASS: transform-on-load (0: the two-pass scheme)
    int vgen = 1;           // WAFER_VGEN: evaluate Coulomb / SimpleCornell / Harmonic per cell instead of streaming V
    int x2 = 1;             // WAFER_X2: two excited-state steps per pass (ThreePoint fp64, 1..3 stored states); 0: one step per pass
    int x2_max_k = 0;       // WAFER_X2_MAX_K: most stored states the two-step kernel takes (1 .. 3); 0 = by plane size: 3 up to 300 000 cells per plane (fp32 storage: always), else 2 (wafer_engine_schedules.hip, x2_applies)
                            // (k = 3 runs on 128 x 8 tiles: -4 % per step at 512 x 512 planes, +3 ... +5 % at 1024 x 1024, profiles/r04_x2_shapes.log)
    int x2_ry = 0;          // WAFER_X2_RY: rows per lane of that kernel (1: 128 x 8 tiles, 2: 128 x 16, k = 1 and 2; 0: default = 2 where it exists)
    // fused kernels
    int f2_wide = 1;        // WAFER_F2_WIDE: 0 keeps FivePoint on the two-step kernel with dedicated helper waves (128 x 8 tiles)
    int fuse3 = 1;          // WAFER_FUSE3: 0 keeps ThreePoint fp64 on the two-step kernel
    int fuse3_min_ny = -1;  // WAFER_FUSE3_MIN_NY (tests; lifts the cell threshold too)
    long long fuse3_min_cells = 1500000; // WAFER_FUSE3_MIN_CELLS (round 5: 6 000 000 until the planned division and the streamed stores moved the crossover)
    // z-slabs
    int overlap = -1;       // WAFER_OVERLAP: initial wafer_set_overlap mode (-1: default)
    int hv_debug = 0;       // WAFER_HV_DEBUG: experiments on the single-launch pass (bits: 4 no acquire fence
                            // (timing only), 8 no short pieces, 16 XCD-contiguous tile order inside each half, 32 no counters / gates (timing only),
                            // 64 exchange stream at normal priority)
    int hv_whole_max = 384; // WAFER_HV_WHOLE_MAX: peer-store passes march whole columns on slabs of up to this many planes, the two halves on thicker ones
    int hv_layout = 0;      // WAFER_HV_LAYOUT: where the short columns go (wafer_f3_schedule_halves); peer-store passes (mode 3): 3 = always
                            // the two halves, 4 = always whole columns (default: whole columns where there is a tile per CU)
    int hv_wait_ms = 20000; // WAFER_HV_WAIT_MS: how long a workgroup of the single-launch pass waits for its ghost planes before it gives up
                            // (WAFER_ERR_COMM; the gate kernels wait four times as long)
    int copy_sched = 2;     // WAFER_COPY_SCHED: the schedule under overlap mode 4 (copies instead of the halo hook): 2 = the single launch on two
                            // halves (default), 1 = boundary planes first (three launches per pass), 0 = the exchange after the pass -- in 1 and 0
                            // every kernel that reads ghost planes starts after the copy that filled them has completed
    int peer_same_device = 0; // WAFER_PEER_SAME_DEVICE: 1 = wafer_peer_connect accepts a neighbour that is another context on this device (tests: ranks folded onto one GPU)
    int f3_xs = 1;          // WAFER_F3_XS: the three-step kernel with an exact store count per plane iteration where it applies (plain launches, grids of whole tiles); 0 = never
    int f3_plain_down = 0;  // WAFER_F3_PLAIN_DOWN: 1 = the plain schedule's workgroups march their columns downwards (the same bits; the two directions are separate copies of the loop, and the compiler's register allocation differs between them)
    int f3_rounds = -1;     // WAFER_F3_ROUNDS: a plain launch of the three-step kernel with more workgroups than CUs goes as one launch per round of CUs, its
                            // columns cut to at most 384 planes (-1: yes; 0: one launch, columns as long as the makespan rule says -- rounds 1-4)
    int f3_sched = 0;       // WAFER_F3_SCHED: 1 = undecomposed launches use the two-halves schedule as well (timing experiments)
};

static inline int wafer_env_int(const char *name, int dflt)
{
    const char *s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

static inline WaferTuning wafer_tuning_from_env()
{
    WaferTuning t;
    t.zchunk = wafer_env_int("WAFER_ZCHUNK", t.zchunk);
    t.target_blocks = wafer_env_int("WAFER_TARGET_BLOCKS", t.target_blocks);
    t.swz = wafer_env_int("WAFER_XCD_SWIZZLE", t.swz);
    t.lds_ry = wafer_env_int("WAFER_LDS_RY", t.lds_ry);
    if (t.lds_ry != 2 && t.lds_ry != 4) t.lds_ry = 0;
    t.nt = wafer_env_int("WAFER_NT", t.nt);
    t.abv = wafer_env_int("WAFER_ABV", t.abv);
    t.xf_nw = wafer_env_int("WAFER_XF_NW", t.xf_nw);
    t.xf_deep = wafer_env_int("WAFER_XF_DEEP", t.xf_deep);
    t.one_pass = wafer_env_int("WAFER_ONE_PASS", t.one_pass);
    t.vgen = wafer_env_int("WAFER_VGEN", t.vgen);
    t.x2 = wafer_env_int("WAFER_X2", t.x2);
    t.x2_ry = wafer_env_int("WAFER_X2_RY", t.x2_ry);
    t.x2_max_k = wafer_env_int("WAFER_X2_MAX_K", t.x2_max_k);
    t.f2_wide = wafer_env_int("WAFER_F2_WIDE", t.f2_wide);
    t.fuse3 = wafer_env_int("WAFER_FUSE3", t.fuse3);
    t.fuse3_min_ny = wafer_env_int("WAFER_FUSE3_MIN_NY", t.fuse3_min_ny);
    t.fuse3_min_cells = wafer_env_int("WAFER_FUSE3_MIN_CELLS", (int)t.fuse3_min_cells);
    t.overlap = wafer_env_int("WAFER_OVERLAP", t.overlap);
    t.hv_debug = wafer_env_int("WAFER_HV_DEBUG", t.hv_debug);
    t.f3_sched = wafer_env_int("WAFER_F3_SCHED", t.f3_sched);
    t.f3_plain_down = wafer_env_int("WAFER_F3_PLAIN_DOWN", t.f3_plain_down);
    t.f3_rounds = wafer_env_int("WAFER_F3_ROUNDS", t.f3_rounds);
    t.f3_xs = wafer_env_int("WAFER_F3_XS", t.f3_xs);
    t.hv_layout = wafer_env_int("WAFER_HV_LAYOUT", t.hv_layout);
    t.hv_whole_max = wafer_env_int("WAFER_HV_WHOLE_MAX", t.hv_whole_max);
    t.hv_wait_ms = wafer_env_int("WAFER_HV_WAIT_MS", t.hv_wait_ms);
    if (t.hv_wait_ms < 1) t.hv_wait_ms = 1;
    t.copy_sched = wafer_env_int("WAFER_COPY_SCHED", t.copy_sched);
    if (t.copy_sched < 0 || t.copy_sched > 2) t.copy_sched = 2;
    t.peer_same_device = wafer_env_int("WAFER_PEER_SAME_DEVICE", t.peer_same_device);
    return t;
}

// Chunks per tile column of a column-marching kernel: the count that minimises the launch's makespan,
//   rounds x (planes per chunk + fill),   rounds = ceil(workgroups / slots),
// where `slots` workgroups are resident at once (CUs x workgroups per CU) and `fill` is what a chunk pays beyond its
// own planes (pipeline fill and prologue, in plane iterations).  Rounds 1 and 2 rounded `slots / tiles per layer` to the
// nearest integer instead, which is the same at 256^3, 512^3 and 1024^3 but not in between: 384^3 has 72 tiles per layer
// of the three-step kernel, 3.55 rounded to 4 chunks = 288 workgroups = a full round and a second round of 32 -- two rounds
// of 100 iterations where 7 chunks give two full rounds of 59 (325 against 449 G updates/s, profiles/NOTES.md).
static inline int wafer_pick_zchunk(long long per_layer, int nplanes, long long slots, int fill, int max_planes = 0)
{
    if (per_layer < 1) per_layer = 1;
    if (slots < 1) slots = 1;
    long long best_cost = -1;
    int best_zc = nplanes;
    const int nch_max = nplanes < 64 ? nplanes : 64;   // (at most 64 chunks per column: the engine sizes its partial-sum rows for that)
    for (int nch = 1; nch <= nch_max; ++nch) {
        const int zc = (nplanes + nch - 1) / nch;
        const int real = (nplanes + zc - 1) / zc;
        const long long rounds = (per_layer * real + slots - 1) / slots;
        const long long cost = rounds * (zc + fill);
        if (max_planes > 0 && zc > max_planes && nch < nch_max) continue;   // (columns no longer than that: see wafer_f3_zchunk)
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_zc = zc; }
    }
    return best_zc;
}

// A plain launch of a multi-step kernel goes as one launch per round of CUs, its columns cut to at most 384 planes
// (wafer_f3_zchunk in wafer_stencil_fused3.hip.h says when and why; the FivePoint two-step kernel follows the same rule)
static inline bool wafer_f3_by_rounds(const WaferTuning &t, long long tiles_per_layer, int nplanes, long long slots)
{
    return t.f3_rounds != 0 && t.zchunk <= 0 && tiles_per_layer > slots && tiles_per_layer % slots == 0 && nplanes > 384;
}
