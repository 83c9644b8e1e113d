// Which CUs does a CU-masked HIP stream run on (hipExtStreamCreateWithCUMask on MI355X: 8 XCDs x 32 CUs)?  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/probes/cu_mask_probe.hip -o /tmp/cu_mask_probe && /tmp/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <vector>

__global__ void k_where(unsigned *out, int spin) {
    // HW_REG_HW_ID (id 4): wave/simd/cu/sh/se ids; HW_REG_XCC_ID (id 20) on gfx94x/gfx950
    unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
    unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

static void run(const char *label, const std::vector<unsigned> &mask, int blocks) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: create failed %d\n", label, (int)e); return; }
    unsigned *d;
    hipMalloc(&d, sizeof(unsigned) * 2 * blocks);
    hipMemset(d, 0, sizeof(unsigned) * 2 * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    k_where<<<blocks, 256, 0, s>>>(d, 20000);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned> h(2 * blocks);
    hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * blocks, hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> per_xcc;
    for (int b = 0; b < blocks; ++b) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
        per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
    }
    int total = 0;
    printf("%s: %.3f ms for %d blocks;", label, ms, blocks);
    for (auto &kv : per_xcc) { printf(" xcc%u:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
    printf("  => %d distinct CUs\n", total);
    hipFree(d);
    hipStreamDestroy(s);
}

int main() {
    const int W = 8;   // 256 bits
    std::vector<unsigned> all(W, 0xffffffffu), first32(W, 0), first64(W, 0), every8(W, 0), low4(W, 0), word7(W, 0);
    first32[0] = 0xffffffffu;
    first64[0] = first64[1] = 0xffffffffu;
    for (int i = 0; i < 256; i += 8) every8[i / 32] |= 1u << (i % 32);
    for (int i = 0; i < 256; ++i) if ((i % 32) < 4) low4[i / 32] |= 1u << (i % 32);
    word7[7] = 0xffffffffu;
    run("all 256 bits", all, 2048);
    run("bits 0..31", first32, 2048);
    run("bits 0..63", first64, 2048);
    run("every 8th bit (32 bits)", every8, 2048);
    run("bits 0..3 of every word (32 bits)", low4, 2048);
    run("bits 224..255", word7, 2048);
    std::vector<unsigned> w1(1, 0x000000ffu);
    run("1 word, bits 0..7", w1, 2048);
    return 0;
}
