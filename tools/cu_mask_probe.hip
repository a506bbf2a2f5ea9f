// Which CUs does a hipExtStreamCreateWithCUMask mask select on this device?  For a few masks: launch many
// short workgroups on a masked stream, let each record its XCC and CU (s_getreg HW_ID / XCC_ID), and
// count the distinct CUs per XCC.  (Groundwork for reserving one CU per XCD for the halo exchange:
// DESIGN.md section 9.)    hipcc --offload-arch=gfx950 tools/cu_mask_probe.hip -o cu_mask_probe && ./cu_mask_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdint>
#include <cstdio>
#include <map>
#include <set>
#include <vector>

__global__ void probe(uint32_t *out, int spin)
{
    if (threadIdx.x == 0) {
        const uint32_t hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, 32 bits
        const uint32_t xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));  // HW_REG_XCC_ID, bits 0..3
        out[blockIdx.x] = (xcc << 16) | (hw & 0xffff);
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
}

static void run(const char *name, const std::vector<uint32_t> &mask)
{
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: stream creation failed\n", name); return; }
    const int n = 8192;
    uint32_t *d = nullptr;
    hipMalloc((void **)&d, n * sizeof(uint32_t));
    hipLaunchKernelGGL(probe, dim3(n), dim3(64), 0, s, d, 2000);
    hipStreamSynchronize(s);
    std::vector<uint32_t> h(n);
    hipMemcpy(h.data(), d, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
    std::map<uint32_t, std::set<uint32_t>> per_xcc;
    for (uint32_t v : h) per_xcc[v >> 16].insert((v >> 8) & 0xff); // CU_ID 8..11, SH_ID 12, SE_ID 13..15
    int total = 0;
    printf("%-34s", name);
    for (auto &kv : per_xcc) { printf(" xcc%u:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
    printf("  total %d\n", total);
    hipFree(d);
    hipStreamDestroy(s);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int words = (p.multiProcessorCount + 31) / 32;
    printf("%s: %d CUs, mask words %d\n", p.name, p.multiProcessorCount, words);
    std::vector<uint32_t> full(words, 0xffffffffu);
    run("full mask", full);
    { auto m = full; m[0] &= ~1u; run("without bit 0", m); }
    { auto m = full; m[0] &= ~0xffu; run("without bits 0..7", m); }
    { auto m = full; for (int w = 0; w < words; ++w) m[w] &= ~1u; run("without bit 0 of every word", m); }
    { auto m = full; m[0] &= ~0x01010101u; run("without bits 0, 8, 16, 24", m); }
    { std::vector<uint32_t> m(words, 0u); m[0] = 0xffu; run("only bits 0..7", m); }
    { std::vector<uint32_t> m(words, 0u); m[0] = 0x1u; run("only bit 0", m); }
    { std::vector<uint32_t> m(words, 0u); for (int w = 0; w < words; ++w) m[w] = 0x1u; run("only bit 0 of every word", m); }
    return 0;
}
