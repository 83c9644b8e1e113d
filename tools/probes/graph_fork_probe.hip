// Does a captured hipGraph run a forked branch beside the main chain on MI355X / ROCm 7.2, and what does a fork + join cost?
// The Bottleneck backward has leaves (the weight-gradient products and their slab reductions: ~1 ms per step) that nothing on the
// dependent chain waits for; at levels 3-5 the chain's kernels occupy a fraction of the chip.  If a graph branch overlaps them, the
// leaves can leave the critical path.
//
// Model: a chain of NCHAIN short kernels (CHAIN_BLOCKS workgroups spinning CHAIN_US each); every PERIOD-th kernel is followed by a leaf
// (LEAF_BLOCKS workgroups spinning LEAF_US) that is either (a) on the chain's stream, (b) forked to a second stream and joined PERIOD
// kernels later, (c) forked, all joins at the end.  Each form is captured once and replayed; also run eagerly.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/graph_fork_probe.hip -o /tmp/graph_fork_probe && /tmp/graph_fork_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d (%s) at %s:%d\n", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_spin(float *out, long long ticks) {   // clock64 counts at 100 MHz on gfx9 (s_memtime): ticks = us * 100
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0) out[blockIdx.x] = (float)ticks;
}

struct Cfg { int nchain, period, chain_blocks, leaf_blocks; float chain_us, leaf_us; };

static void issue(const Cfg &c, int mode, hipStream_t s0, hipStream_t s1, float *buf, std::vector<hipEvent_t> &ev) {
    // mode 0: leaves on the chain's stream; 1: forked, joined `period` kernels later; 2: forked, joined at the end
    const long long ct = (long long)(c.chain_us * 100), lt = (long long)(c.leaf_us * 100);
    int pending = -1, e = 0;
    std::vector<int> joins;
    for (int i = 0; i < c.nchain; ++i) {
        k_spin<<<c.chain_blocks, 256, 0, s0>>>(buf, ct);
        if ((i + 1) % c.period == 0) {
            if (mode == 0) {
                k_spin<<<c.leaf_blocks, 256, 0, s0>>>(buf + 4096, lt);
            } else {
                if (mode == 1 && pending >= 0) { CK(hipStreamWaitEvent(s0, ev[pending], 0)); pending = -1; }
                CK(hipEventRecord(ev[e], s0));
                CK(hipStreamWaitEvent(s1, ev[e], 0));
                k_spin<<<c.leaf_blocks, 256, 0, s1>>>(buf + 4096, lt);
                CK(hipEventRecord(ev[e + 1], s1));
                if (mode == 1) pending = e + 1; else joins.push_back(e + 1);
                e += 2;
            }
        }
    }
    if (mode == 1 && pending >= 0) CK(hipStreamWaitEvent(s0, ev[pending], 0));
    if (mode == 2 && !joins.empty()) CK(hipStreamWaitEvent(s0, ev[joins.back()], 0));   // (the side stream is in order: its last event covers all)
}

int main() {
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    float *buf;
    CK(hipMalloc(&buf, sizeof(float) * 8192));
    std::vector<hipEvent_t> ev(1024);
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    const Cfg cfgs[] = {
        {200, 10, 64, 128, 5.f, 30.f},    // short latency-bound chain kernels, a 30-us leaf every 10
        {200, 10, 64, 128, 10.f, 30.f},
        {200, 5, 64, 128, 10.f, 30.f},
        {200, 10, 256, 128, 10.f, 30.f},  // the chain fills every CU (one workgroup each): leaves get the second slot
        {200, 10, 64, 128, 5.f, 0.f},     // empty leaves: pure fork + join cost
    };
    const char *names[3] = {"leaves on the chain's stream", "forked, joined one period later", "forked, joined at the end"};
    for (const Cfg &c : cfgs) {
        const int nleaf = c.nchain / c.period;
        printf("chain %d x %.0f us on %d workgroups, %d leaves of %.0f us on %d workgroups (sum of spins: chain %.2f ms, leaves %.2f ms)\n", c.nchain, c.chain_us,
               c.chain_blocks, nleaf, c.leaf_us, c.leaf_blocks, c.nchain * c.chain_us * 1e-3, nleaf * c.leaf_us * 1e-3);
        for (int mode = 0; mode < 3; ++mode) {
            // eager
            float eager = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(t0, s0));
                issue(c, mode, s0, s1, buf, ev);
                CK(hipEventRecord(t1, s0));
                CK(hipEventSynchronize(t1));
                CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, t0, t1));
                if (ms < eager) eager = ms;
            }
            // captured
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
            issue(c, mode, s0, s1, buf, ev);
            CK(hipStreamEndCapture(s0, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
            CK(hipGraphLaunch(ge, s0)); CK(hipStreamSynchronize(s0));
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(t0, s0));
                CK(hipGraphLaunch(ge, s0));
                CK(hipEventRecord(t1, s0));
                CK(hipEventSynchronize(t1));
                float ms; CK(hipEventElapsedTime(&ms, t0, t1));
                if (ms < best) best = ms;
            }
            printf("  %-34s eager %.3f ms   graph replay %.3f ms (%zu nodes)\n", names[mode], eager, best, nn);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
    }
    return 0;
}
