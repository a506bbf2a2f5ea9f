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
//       ncclAllReduce); the run is repeated with plain device copies as hooks and must agree bit for
//       bit.  Prints SELF-OK.  (tests/test_gpu_multiprocess.py)
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

// the same exchange with itself as both neighbours, by device copies: what the self test expects RCCL to deliver.  With one
// rank as both neighbours RCCL pairs receives and sends in posting order (wafer_rccl_halo posts recv_lo, recv_hi, then
// send_lo, send_hi).  The single-launch pass (wafer_set_overlap mode 2) calls the hook with ONE direction -- send_lo and
// recv_hi, or send_hi and recv_lo, the other two NULL: the pairing by posting order covers that as well.
static int copy_halo(void *, void *send_lo, void *send_hi, void *recv_lo, void *recv_hi, size_t bytes, void *stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    void *recvs[2] = {recv_lo, recv_hi}, *sends[2] = {send_lo, send_hi};
    int is = 0;
    for (int ir = 0; ir < 2; ++ir) {
        if (!recvs[ir]) continue;
        while (is < 2 && !sends[is]) ++is;
        if (is == 2) return 1; // a receive without a send: not a self-neighbour exchange
        if (hipMemcpyAsync(recvs[ir], sends[is++], bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return 1;
    }
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
};

static int run_slab(const wafer_params &p, int potential, uint64_t steps, wafer_halo_fn halo, wafer_allreduce_fn allreduce,
                    void *user, bool download, RunResult &out)
{
    wafer_ctx *ctx = nullptr;
    WCHECK(wafer_ctx_create(&p, &ctx));
    WCHECK(wafer_set_comm_hooks(ctx, halo, allreduce, user));
    WCHECK(wafer_set_potential_builtin(ctx, potential));
    WCHECK(wafer_set_initial_condition(ctx, WAFER_IC_BOOLEAN, 0));
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
        RunResult via_rccl, via_copies;
        if ((rc = run_slab(p, potential, steps, rccl_halo, rccl_allreduce, &fab, true, via_rccl))) return rc;
        if ((rc = run_slab(p, potential, steps, copy_halo, copy_allreduce, nullptr, true, via_copies))) return rc;
        const bool same = via_rccl.phi.size() == via_copies.phi.size() &&
                          memcmp(via_rccl.phi.data(), via_copies.phi.data(), via_rccl.phi.size() * sizeof(double)) == 0 &&
                          memcmp(&via_rccl.obs, &via_copies.obs, sizeof(wafer_observables_t)) == 0;
        double sum = 0.0;
        for (double v : via_rccl.phi) sum += v * v;
        if (!same || !(sum > 0.0) || fab.halo_calls == 0 || fab.reduce_calls == 0) {
            fprintf(stderr, "SELF-FAIL same=%d sum=%g halo_calls=%ld reduce_calls=%ld\n", (int)same, sum, fab.halo_calls, fab.reduce_calls);
            return 1;
        }
        printf("SELF-OK halo_calls=%ld reduce_calls=%ld ms_per_step_rccl=%.4f ms_per_step_copies=%.4f\n", fab.halo_calls,
               fab.reduce_calls, via_rccl.ms / (double)via_rccl.steps, via_copies.ms / (double)via_copies.steps);
    } else {
        // contiguous balanced z-ranges, the rule of wafer_amd.slab.partition
        const uint32_t base = nz / (uint32_t)world, extra = nz % (uint32_t)world;
        p.z_count = world > 1 ? base + ((uint32_t)rank < extra ? 1u : 0u) : 0;
        p.z_begin = world > 1 ? (uint32_t)rank * base + ((uint32_t)rank < extra ? (uint32_t)rank : extra) : 0;
        if (world == 1) p.halo_depth = 0;
        fab.lower = rank > 0 ? rank - 1 : -1;
        fab.upper = rank + 1 < world ? rank + 1 : -1;
        RunResult r;
        if ((rc = run_slab(p, potential, steps, rccl_halo, rccl_allreduce, &fab, false, r))) return rc;
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
                   "\"energy_per_norm2\": %.12e, \"norm2\": %.12e, \"halo_calls\": %ld}\n",
                   world, nx, ny, nz, (unsigned long long)r.steps, r.ms / (double)r.steps,
                   (double)nx * ny * nz * (double)r.steps / (r.ms * 1e-3), r.obs.energy / r.obs.norm2, r.obs.norm2, fab.halo_calls);
    }
    ncclCommDestroy(fab.comm);
    return 0;
}
