// wafer-hip-slabs -- a native (no Python, no torch) multi-GPU host above the C ABI: one process
// per GPU, the grid z-slab decomposed over the ranks, the engine's two communication hooks served by
// RCCL called directly (ncclSend / ncclRecv in a group, ncclAllReduce) on the stream the engine
// passes.  This is the shape of the Rust host INTEGRATION.md section 3 describes; the hooks are the
// ones libwafer_rccl.so ships (wafer_rccl_hooks.h), which wafer_amd/slab.py + wafer_amd/run.py and
// bench.py install by default.
//
//   RANK=r WORLD_SIZE=n LOCAL_RANK=r WAFER_NCCL_ID_FILE=/tmp/id  wafer-hip-slabs NX NY NZ STEPS [potential]
//       every rank runs this; rank 0 writes the ncclUniqueId to the file, the others wait for it.
//       Ground-state evolve of a Boolean start; rank 0 prints one JSON line.
//   wafer-hip-slabs --self NX NY NZ STEPS
//       ONE process, one GPU: the slab is a middle slab of an 8-rank world whose neighbours are this
//       same rank, so halo planes really travel through ncclSend / ncclRecv (and scalars through
//       ncclAllReduce); the run is repeated with plain device copies as hooks, and once more with peer stores (overlap
//       mode 3: connected and self-checked exactly as a multi-rank run does), and all three must agree bit for bit.
//       Prints SELF-OK.  (tests/test_gpu_multiprocess.py)
//   Multi-rank runs use peer stores (the z-neighbours' buffers mapped through HIP IPC, records exchanged with ncclAllGather) once
//   15 steps under them have left every rank's planes with the bits of an exchange through the halo hook; WAFER_PEER_STORES=0: never.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <sys/stat.h>
#include <unistd.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "../../include/wafer_hip.h"
#include "wafer_rccl_hooks.h"

#define HIPCHECK(x)                                                                         \
    do {                                                                                    \
        hipError_t e_ = (x);                                                                \
        if (e_ != hipSuccess) {                                                             \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));                  \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)
#define NCCLCHECK(x)                                                                        \
    do {                                                                                    \
        ncclResult_t r_ = (x);                                                              \
        if (r_ != ncclSuccess) {                                                            \
            fprintf(stderr, "%s failed: %s\n", #x, ncclGetErrorString(r_));                 \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)
#define WCHECK(x)                                                                           \
    do {                                                                                    \
        if ((x) != WAFER_OK) {                                                              \
            fprintf(stderr, "%s failed: %s\n", #x, wafer_last_error());                     \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)

using Fabric = WaferRcclFabric;
static auto &rccl_halo = wafer_rccl_halo;
static auto &rccl_allreduce = wafer_rccl_allreduce;

// the same exchange with itself as both neighbours, by device copies: what the self test expects RCCL to deliver -- a ring
// of one slab, the lower ghost planes taking the slab's own upper boundary planes and the other way round (wafer_rccl_halo
// posts its sends in that order when one rank is both neighbours).  The single-launch pass (wafer_set_overlap mode 2) calls
// the hook with ONE direction -- send_lo and recv_hi, or send_hi and recv_lo, the other two NULL: the same ring.
static int copy_halo(void *, void *send_lo, void *send_hi, void *recv_lo, void *recv_hi, size_t bytes, void *stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((recv_lo && !send_hi) || (recv_hi && !send_lo)) return 1; // a receive without its send: not a self-neighbour exchange
    if (recv_lo && hipMemcpyAsync(recv_lo, send_hi, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return 1;
    if (recv_hi && hipMemcpyAsync(recv_hi, send_lo, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return 1;
    return 0;
}
static int copy_allreduce(void *, void *, size_t, void *) { return 0; }

static int env_int(const char *name, int dflt)
{
    const char *s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

struct RunResult {
    std::vector<double> phi;
    wafer_observables_t obs;
    float ms = 0.f;
    uint64_t steps = 0;
    int overlap_mode = 2;        // what the timed steps ran under
    int peer_check = -1;         // -1: peer stores / copies not tried, 0: tried and dropped (connection or bits), 3 / 4 / 5: that overlap mode checked and in use
};

// all ranks: is `mine` true everywhere?  (world 1: yes if mine)
static int all_agree(Fabric *fab, int world, bool mine, bool *out)
{
    *out = mine;
    if (world <= 1 || !fab) return 0;
    int *d = nullptr, h = mine ? 1 : 0;
    HIPCHECK(hipMalloc((void **)&d, sizeof(int)));
    HIPCHECK(hipMemcpy(d, &h, sizeof h, hipMemcpyHostToDevice));
    NCCLCHECK(ncclAllReduce(d, d, 1, ncclInt, ncclMin, fab->comm, nullptr));
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    *out = h == 1;
    return 0;
}

// Peer stores (wafer_set_overlap mode 3) for this host: every rank's wafer_peer_info travels through ncclAllGather, each rank maps
// its z-neighbours', and -- nothing in this repository has crossed a link -- the mode has to reproduce the bits of an exchange
// through the halo hook on EVERY rank (15 steps from the Boolean start, the checksum of the rank's own planes) before it is used;
// any failure anywhere leaves every rank on mode 2.  Collective.  self_loop: the one rank is its own neighbour on both sides.
static int try_peer_stores(wafer_ctx *ctx, const wafer_params &p, Fabric *fab, int rank, int world, bool self_loop, uint32_t thinnest, int *result)
{
    *result = 0;
    wafer_peer_info mine;
    bool ok = wafer_peer_export(ctx, &mine) == WAFER_OK, all_ok = false;
    std::vector<wafer_peer_info> all((size_t)world, mine);
    if (world > 1) {
        char *d = nullptr;
        HIPCHECK(hipMalloc((void **)&d, sizeof mine * (size_t)(world + 1)));
        HIPCHECK(hipMemcpy(d, &mine, sizeof mine, hipMemcpyHostToDevice));
        NCCLCHECK(ncclAllGather(d, d + sizeof mine, sizeof mine, ncclChar, fab->comm, nullptr));
        HIPCHECK(hipDeviceSynchronize());
        HIPCHECK(hipMemcpy(all.data(), d + sizeof mine, sizeof mine * (size_t)world, hipMemcpyDeviceToHost));
        (void)hipFree(d);
    }
    if (all_agree(fab, world, ok, &all_ok)) return 1;
    if (!all_ok) return 0;
    const wafer_peer_info *lower = self_loop ? &mine : (rank > 0 ? &all[(size_t)rank - 1] : nullptr);
    const wafer_peer_info *upper = self_loop ? &mine : (rank + 1 < world ? &all[(size_t)rank + 1] : nullptr);
    ok = wafer_peer_connect(ctx, lower, upper) == WAFER_OK;
    if (!ok) fprintf(stderr, "rank %d: wafer_peer_connect: %s\n", rank, wafer_last_error());
    if (all_agree(fab, world, ok, &all_ok)) return 1;
    if (!all_ok) { (void)wafer_peer_disconnect(ctx); return 0; }
    // mode 3 needs six owned planes on EVERY rank (`thinnest`: a quantity all ranks share); mode 4 -- peer copies: every exchange a
    // device copy into the neighbour's ghost planes -- serves any slab and is tried where mode 3 does not apply or fails its check
    // (WAFER_PEER_MODE=4: peer copies only -- the self test times both)
    // then 5: the copies under the boundary-first launches, where every reader of ghost planes starts after the copy
    const bool try3 = thinnest >= 6 && env_int("WAFER_PEER_MODE", 3) != 4;
    const int candidates[3] = {try3 ? 3 : 4, try3 ? 4 : 5, try3 ? 5 : 0};
    for (int ci = 0; ci < 3; ++ci) {
        const int mode = candidates[ci];
        if (mode == 0) break;
        uint64_t sum[2] = {0, 0};
        ok = true;
        for (int i = 0; i < 2; ++i) {
            // (self_loop: the generated start carries the GLOBAL grid's values in its ghost planes, the wrap-around exchange delivers this
            //  slab's own; dividing by sqrt(1) changes no bit and marks the ghost planes stale, so both schedules start from an exchange)
            const bool mine_ok = ok && wafer_set_overlap(ctx, i == 0 ? mode : 2) == WAFER_OK && wafer_set_initial_condition(ctx, WAFER_IC_BOOLEAN, 0) == WAFER_OK &&
                                 (!self_loop || wafer_normalise(ctx, 1.0) == WAFER_OK) &&
                                 wafer_evolve(ctx, 0, 15) == WAFER_OK && wafer_synchronize(ctx) == WAFER_OK &&
                                 wafer_diag_checksum(ctx, p.z_begin, p.z_count, &sum[i]) == WAFER_OK;
            bool everyone = false;   // the same collectives on every rank whatever happened here
            if (all_agree(fab, world, mine_ok, &everyone)) return 1;
            ok = everyone;
        }
        ok = ok && sum[0] == sum[1];
        if (all_agree(fab, world, ok, &all_ok)) return 1;
        if (!all_ok && rank == 0)
            fprintf(stderr, "%s (overlap mode %d) do not reproduce the exchange's bits on this fabric\n", mode == 3 ? "peer stores" : "peer copies", mode);
        if (all_ok) {
            WCHECK(wafer_set_overlap(ctx, mode));
            *result = mode;
            return 0;
        }
    }
    WCHECK(wafer_set_overlap(ctx, 2));
    *result = 0;
    return 0;
}

// peers: try overlap mode 3 (fab: the communicator the records travel through; rank / world / self_loop as in try_peer_stores)
static int run_slab(const wafer_params &p, int potential, uint64_t steps, wafer_halo_fn halo, wafer_allreduce_fn allreduce,
                    void *user, bool download, RunResult &out, bool peers = false, Fabric *fab = nullptr, int rank = 0, int world = 1,
                    bool self_loop = false, bool wrap = false)
{
    wafer_ctx *ctx = nullptr;
    WCHECK(wafer_ctx_create(&p, &ctx));
    WCHECK(wafer_set_comm_hooks(ctx, halo, allreduce, user));
    WCHECK(wafer_set_potential_builtin(ctx, potential));
    // try_peer_stores is COLLECTIVE (an all-gather and several all-reduces): whether it is entered must depend on a quantity every
    // rank shares -- the thinnest slab of the partition (nz / world: nz = 47 over 8 ranks hands out 6, 6, 6, 6, 6, 6, 6, 5), never
    // this rank's own thickness; wafer_set_overlap(3) needs 6 owned planes on every rank (thinner partitions go to mode 4, peer copies)
    const uint32_t thinnest = (p.z_count && world > 1) ? p.nz / (uint32_t)world : (p.z_count ? p.z_count : p.nz);
    if (peers) {
        if (try_peer_stores(ctx, p, fab, rank, world, self_loop, thinnest, &out.peer_check)) return 1;
        out.overlap_mode = out.peer_check > 0 ? out.peer_check : 2;
    }
    WCHECK(wafer_set_initial_condition(ctx, WAFER_IC_BOOLEAN, 0));
    if (wrap) WCHECK(wafer_normalise(ctx, 1.0));   // --self: every schedule takes its first ghost planes from the (wrap-around) exchange
    WCHECK(wafer_evolve(ctx, 0, steps < 20 ? steps : 20)); // warm-up (and channel set-up) outside the timing
    WCHECK(wafer_evolve(ctx, 0, steps));
    WCHECK(wafer_last_evolve_ms(ctx, &out.ms, &out.steps));
    WCHECK(wafer_observables(ctx, &out.obs));
    if (download) {
        const uint32_t e = (uint32_t)p.central_difference;
        out.phi.assign((size_t)(p.nx + 2 * e) * (p.ny + 2 * e) * (p.nz + 2 * e), 0.0);
        WCHECK(wafer_download_phi(ctx, out.phi.data()));
    }
    WCHECK(wafer_ctx_destroy(ctx));
    return 0;
}

int main(int argc, char **argv)
{
    const auto started = std::chrono::system_clock::now(); // before anything slow (HIP initialisation of eight ranks at once)
    bool self = false;
    int a0 = 1;
    if (argc > 1 && std::string(argv[1]) == "--self") { self = true; a0 = 2; }
    if (argc < a0 + 4) {
        fprintf(stderr, "usage: wafer-hip-slabs [--self] NX NY NZ STEPS [potential index, default SimpleCornell]\n");
        return 2;
    }
    const uint32_t nx = (uint32_t)atoi(argv[a0]), ny = (uint32_t)atoi(argv[a0 + 1]), nz = (uint32_t)atoi(argv[a0 + 2]);
    const uint64_t steps = (uint64_t)atoll(argv[a0 + 3]);
    const int potential = argc > a0 + 4 ? atoi(argv[a0 + 4]) : (int)WAFER_POT_SIMPLECORNELL;
    const int rank = self ? 0 : env_int("RANK", 0), world = self ? 1 : env_int("WORLD_SIZE", 1);
    // this rank's two engine streams run beside RCCL's: more hardware queues than the runtime's default of four, so that a kernel
    // waiting for a flag and the kernel (or copy) that sets it never share one; read by the HIP runtime when it initialises (below)
    if (world > 1) setenv("GPU_MAX_HW_QUEUES", "8", 0);
    int device = env_int("LOCAL_RANK", rank), visible = 0;
    HIPCHECK(hipGetDeviceCount(&visible));
    if (visible > 0 && device >= visible) device %= visible; // a launcher that shows each rank only its own GPU
    HIPCHECK(hipSetDevice(device));

    // ---- communicator: rank 0 publishes the unique id through a file ------------------------------
    Fabric fab;
    ncclUniqueId id;
    if (world > 1) {
        const char *path = getenv("WAFER_NCCL_ID_FILE");
        if (!path) { fprintf(stderr, "WAFER_NCCL_ID_FILE must name a file all ranks can reach\n"); return 2; }
        // The file carries a run nonce in front of the id so that a file left over from an earlier run is
        // never mistaken for this one's: WAFER_RUN_ID, else the launcher's TORCHELASTIC_RUN_ID, else MASTER_PORT (any
        // value all ranks of ONE run share); without any of them the file must not be older than this process by
        // more than the slack below (rank 0 removes any old file before it generates the id, writes to .tmp and
        // renames; it removes the file again after ncclCommInitRank).
        // MASTER_PORT is usually the same value run after run, so a file that a crashed run left behind carries the nonce of
        // the next one: a nonce taken from MASTER_PORT narrows the match but does NOT lift the age check.
        const char *strong_nonce = getenv("WAFER_RUN_ID") ? getenv("WAFER_RUN_ID") : getenv("TORCHELASTIC_RUN_ID");
        const char *nonce_env = strong_nonce ? strong_nonce : getenv("MASTER_PORT");
        char nonce[64];
        memset(nonce, 0, sizeof nonce);
        if (nonce_env) strncpy(nonce, nonce_env, sizeof nonce - 1);
        if (rank == 0) {
            remove(path);
            NCCLCHECK(ncclGetUniqueId(&id));
            const std::string tmp = std::string(path) + ".tmp";
            FILE *f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(nonce, sizeof nonce, 1, f) != 1 || fwrite(&id, sizeof id, 1, f) != 1) { fprintf(stderr, "cannot write %s\n", tmp.c_str()); return 1; }
            fclose(f);
            rename(tmp.c_str(), path);
        } else {
            bool got = false;
            const char *why = "no such file";
            for (int tries = 0; tries < 6000 && !got; ++tries) {
                FILE *f = fopen(path, "rb");
                if (f) {
                    char seen[64];
                    struct stat st;
                    if (fread(seen, sizeof seen, 1, f) == 1 && fread(&id, sizeof id, 1, f) == 1) {
                        got = true;
                        if (nonce_env && memcmp(seen, nonce, sizeof nonce) != 0) {
                            got = false;
                            why = "its run id is not this run's (a file left over from another run?)";
                        }
                        if (got && !strong_nonce) {
                            // no run id of its own: accept only a file written after this process started, with slack for
                            // ranks started by hand one after the other and the 1 s granularity of st_mtime
                            got = stat(path, &st) == 0 &&
                                  std::chrono::system_clock::from_time_t(st.st_mtime) + std::chrono::seconds(30) >= started;
                            why = "it is older than this run (a file left over from another run? set WAFER_RUN_ID)";
                        }
                    } else {
                        why = "it is shorter than a nonce + ncclUniqueId";
                    }
                    fclose(f);
                }
                if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(10));
            }
            if (!got) { fprintf(stderr, "rank %d: no usable ncclUniqueId in %s after 60 s: %s\n", rank, path, why); return 1; }
        }
    } else {
        NCCLCHECK(ncclGetUniqueId(&id));
    }
    wafer_rccl_default_env();
    {
        // ncclCommInitRank blocks for ever when a peer never arrives (or arrived with another id): a watchdog ends the
        // process instead.  WAFER_COMM_INIT_TIMEOUT_S (default 300, at least 1: a limit of 0 would end the process at once) bounds the call.
        std::atomic<bool> init_done{false};
        const int limit_s = std::max(1, env_int("WAFER_COMM_INIT_TIMEOUT_S", 300));
        std::thread watchdog([&init_done, limit_s, rank]() {
            for (int i = 0; i < limit_s * 10 && !init_done.load(); ++i) std::this_thread::sleep_for(std::chrono::milliseconds(100));
            if (!init_done.load()) {
                fprintf(stderr, "rank %d: ncclCommInitRank did not return within %d s (a rank missing, or a stale id file?)\n", rank, limit_s);
                fflush(stderr);
                _exit(3);
            }
        });
        const ncclResult_t ir = ncclCommInitRank(&fab.comm, world, id, rank);
        init_done.store(true);
        watchdog.join();
        NCCLCHECK(ir);
    }
    if (world > 1 && rank == 0) remove(getenv("WAFER_NCCL_ID_FILE")); // every rank has read it: ncclCommInitRank is collective

    // ---- slab of this rank ----------------------------------------------------------------------------
    wafer_params p;
    memset(&p, 0, sizeof p);
    p.struct_size = sizeof p;
    p.nx = nx; p.ny = ny; p.nz = nz;
    p.central_difference = WAFER_CD_THREEPOINT;
    p.dtype = WAFER_F64;
    p.dn = 0.02; p.dt = 8e-5; p.mass = 2.35; p.sig = 0.223; // BASELINE config #4
    p.max_states = 1;
    p.device = device;
    p.halo_depth = 3; // ThreePoint fp64: three fused steps per exchange (wafer_k_step3_fused)
    int rc = 0;
    if (self) {
        const uint32_t parts = 8, per = nz / parts;
        if (per < 4) { fprintf(stderr, "--self needs NZ >= 32\n"); return 2; }
        p.z_begin = 4 * per; p.z_count = per; // a middle slab: neighbours on both sides
        fab.lower = fab.upper = 0;
        RunResult via_rccl, via_copies, via_peers, via_peer_copies;
        if ((rc = run_slab(p, potential, steps, rccl_halo, rccl_allreduce, &fab, true, via_rccl, false, nullptr, 0, 1, false, true))) return rc;
        if ((rc = run_slab(p, potential, steps, copy_halo, copy_allreduce, nullptr, true, via_copies, false, nullptr, 0, 1, false, true))) return rc;
        // ... and with the boundary workgroups storing into the neighbour's (= this slab's own) ghost planes themselves, after the
        // self-check every rank of a real run makes (try_peer_stores)
        if ((rc = run_slab(p, potential, steps, rccl_halo, rccl_allreduce, &fab, true, via_peers, true, &fab, 0, 1, true, true))) return rc;
        // ... and with every exchange a device copy into the neighbour's (= this slab's own) ghost planes behind the credit / arrival
        // rendezvous (overlap mode 4), after the same self-check
        setenv("WAFER_PEER_MODE", "4", 1);
        if ((rc = run_slab(p, potential, steps, rccl_halo, rccl_allreduce, &fab, true, via_peer_copies, true, &fab, 0, 1, true, true))) return rc;
        unsetenv("WAFER_PEER_MODE");
        const bool same_peer_copies = via_peer_copies.peer_check == 4 && via_peer_copies.overlap_mode == 4 && via_peer_copies.phi.size() == via_rccl.phi.size() &&
                                      memcmp(via_peer_copies.phi.data(), via_rccl.phi.data(), via_rccl.phi.size() * sizeof(double)) == 0 &&
                                      memcmp(&via_peer_copies.obs, &via_rccl.obs, sizeof(wafer_observables_t)) == 0;
        const bool same_copies = via_rccl.phi.size() == via_copies.phi.size() &&
                                 memcmp(via_rccl.phi.data(), via_copies.phi.data(), via_rccl.phi.size() * sizeof(double)) == 0 &&
                                 memcmp(&via_rccl.obs, &via_copies.obs, sizeof(wafer_observables_t)) == 0;
        const bool same_peers = via_peers.peer_check == 3 && via_peers.overlap_mode == 3 && via_peers.phi.size() == via_rccl.phi.size() &&
                                memcmp(via_peers.phi.data(), via_rccl.phi.data(), via_rccl.phi.size() * sizeof(double)) == 0 &&
                                memcmp(&via_peers.obs, &via_rccl.obs, sizeof(wafer_observables_t)) == 0;
        const bool same = same_copies && same_peers && same_peer_copies;
        if (!same) {
            fprintf(stderr, "rccl vs copies: %d; peer stores vs rccl: %d (peer check %d, mode %d); peer copies vs rccl: %d (peer check %d, mode %d)\n", (int)same_copies,
                    (int)same_peers, via_peers.peer_check, via_peers.overlap_mode, (int)same_peer_copies, via_peer_copies.peer_check, via_peer_copies.overlap_mode);
            // which z planes (padded index) differ: ghost planes of the slab, or owned ones?
            const size_t pz = nz + 2;
            std::vector<long> per_plane(pz, 0);
            for (size_t i = 0; i < via_rccl.phi.size() && i < via_peers.phi.size(); ++i)
                if (memcmp(&via_rccl.phi[i], &via_peers.phi[i], sizeof(double)) != 0) ++per_plane[i % pz];
            for (size_t k = 0; k < pz; ++k)
                if (per_plane[k]) fprintf(stderr, "  padded z %zu: %ld cells differ (slab owns padded z %u .. %u)\n", k, per_plane[k], p.z_begin + 1, p.z_begin + p.z_count);
        }
        double sum = 0.0;
        for (double v : via_rccl.phi) sum += v * v;
        if (!same || !(sum > 0.0) || fab.halo_calls == 0 || fab.reduce_calls == 0) {
            fprintf(stderr, "SELF-FAIL same=%d sum=%g halo_calls=%ld reduce_calls=%ld\n", (int)same, sum, fab.halo_calls, fab.reduce_calls);
            return 1;
        }
        printf("SELF-OK halo_calls=%ld reduce_calls=%ld ms_per_step_rccl=%.4f ms_per_step_copies=%.4f ms_per_step_peer_stores=%.4f ms_per_step_peer_copies=%.4f\n", fab.halo_calls,
               fab.reduce_calls, via_rccl.ms / (double)via_rccl.steps, via_copies.ms / (double)via_copies.steps,
               via_peers.ms / (double)via_peers.steps, via_peer_copies.ms / (double)via_peer_copies.steps);
    } else {
        // contiguous balanced z-ranges, the rule of wafer_amd.slab.partition
        const uint32_t base = nz / (uint32_t)world, extra = nz % (uint32_t)world;
        p.z_count = world > 1 ? base + ((uint32_t)rank < extra ? 1u : 0u) : 0;
        p.z_begin = world > 1 ? (uint32_t)rank * base + ((uint32_t)rank < extra ? (uint32_t)rank : extra) : 0;
        if (world == 1) p.halo_depth = 0;
        fab.lower = rank > 0 ? rank - 1 : -1;
        fab.upper = rank + 1 < world ? rank + 1 : -1;
        RunResult r;
        // peer stores unless WAFER_PEER_STORES=0: connected, checked against an exchange's bits on every rank, else mode 2
        const bool peers = world > 1 && env_int("WAFER_PEER_STORES", 1) != 0;
        if ((rc = run_slab(p, potential, steps, rccl_halo, rccl_allreduce, &fab, false, r, peers, &fab, rank, world, false))) return rc;
        // slowest rank's kernel time decides
        float *dms = nullptr;
        HIPCHECK(hipMalloc((void **)&dms, sizeof(float)));
        HIPCHECK(hipMemcpy(dms, &r.ms, sizeof(float), hipMemcpyHostToDevice));
        NCCLCHECK(ncclAllReduce(dms, dms, 1, ncclFloat, ncclMax, fab.comm, nullptr));
        HIPCHECK(hipDeviceSynchronize());
        HIPCHECK(hipMemcpy(&r.ms, dms, sizeof(float), hipMemcpyDeviceToHost));
        (void)hipFree(dms);
        if (rank == 0)
            printf("{\"n_gpus\": %d, \"grid\": [%u, %u, %u], \"steps\": %llu, \"ms_per_step\": %.5f, \"updates_per_s\": %.6e, "
                   "\"energy_per_norm2\": %.12e, \"norm2\": %.12e, \"halo_calls\": %ld, \"halo_overlap_mode\": %d, \"peer_store_check\": %d}\n",
                   world, nx, ny, nz, (unsigned long long)r.steps, r.ms / (double)r.steps,
                   (double)nx * ny * nz * (double)r.steps / (r.ms * 1e-3), r.obs.energy / r.obs.norm2, r.obs.norm2, fab.halo_calls,
                   r.overlap_mode, r.peer_check);
    }
    ncclCommDestroy(fab.comm);
    return 0;
}
