// What does a barrier among the workgroups of ONE XCD cost on MI355X, and what can those <= 32 CUs compute?  (round-4 verdict, item 1:
// price a single-XCD barrier before ruling a persistent small-level Bottleneck in or out.)
//
//   (a) barrier among W = 8 / 16 / 32 workgroups that all sit on XCD 0 (blocks b with b % 8 == 0 of an 8 W grid; the XCC id of every
//       participant is read back and reported), per-XCD counter only, two forms:
//         light : s_waitcnt vmcnt(0) -> relaxed agent add -> relaxed poll -> buffer_inv sc1 (L1 invalidate; the XCD's L2 is shared, no
//                 write-back).  Valid ONLY while all participants share one L2, i.e. it leans on placement HIP does not promise.
//         agent : agent-scope release fence -> add -> poll -> agent-scope acquire fence (placement-independent).
//       each with and without a 1 KB publish per workgroup that two other workgroups read and check after the barrier.
//   (b) for calibration on the same box: the flat counter and the XCD-hierarchical barrier over all 256 workgroups.
//   (c) fp32 matrix-core rate of 32 workgroups (one XCD) against 256: what a level-5 Bottleneck (2.9 GFLOP forward) would be bound by.
//
// Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/probes/xcd_barrier_probe.hip -o /tmp/xcd_barrier_probe && /tmp/xcd_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned ld_relaxed(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 0xf; }

// ---- (a) one counter, W participants
template <bool LIGHT>
__device__ __forceinline__ void bar_one(unsigned *ctr, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (LIGHT) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (ld_relaxed(ctr) < target) __builtin_amdgcn_s_sleep(1);
        if (LIGHT) asm volatile("buffer_inv sc1" ::: "memory");
        else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// state: [0] counter, [16 + rank] xcc id of participant `rank`, [64] error count
template <bool LIGHT, bool PAYLOAD>
__global__ __launch_bounds__(256) void k_one_xcd(unsigned *state, float *slots, int W, int rounds, int stride8) {
    if (stride8 && (blockIdx.x & 7) != 0) return;
    const int rank = stride8 ? blockIdx.x >> 3 : blockIdx.x;
    if (threadIdx.x == 0) state[16 + rank] = xcc_id();
    unsigned errs = 0;
    for (int r = 0; r < rounds; ++r) {
        if (PAYLOAD) slots[((size_t)(r & 1) * W + rank) * 256 + threadIdx.x] = (float)(r * 1024 + rank * 4 + (threadIdx.x & 3));
        bar_one<LIGHT>(state, (unsigned)(r + 1) * W);
        if (PAYLOAD) {
            const int a = (rank + 1) % W, b = (rank + W / 2) % W;
            const float va = slots[((size_t)(r & 1) * W + a) * 256 + threadIdx.x], vb = slots[((size_t)(r & 1) * W + b) * 256 + threadIdx.x];
            errs += va != (float)(r * 1024 + a * 4 + (threadIdx.x & 3));
            errs += vb != (float)(r * 1024 + b * 4 + (threadIdx.x & 3));
        }
    }
    if (PAYLOAD && errs) atomicAdd(state + 64, errs);
}

// ---- (b) all 256 workgroups: flat counter / XCD-hierarchical
// state: [0] top counter, [32 + 16 x] per-XCC counter, [192 + 16 x] per-XCC generation, [320 + x] members of XCC x (census), [64] errors
__device__ __forceinline__ void bar_hier(unsigned *state, unsigned xcc, unsigned members, unsigned epoch) {
    // every wave drains its OWN stores into the XCD's L2 before the workgroup arrives: only the XCD's last arriver writes the L2 back
    // (first run of this probe, lane 0 alone waiting: 32 stale words per round -- stores of the other three waves still in flight)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(state + 32 + 16 * xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == epoch * members + members - 1) {   // last arriver of this XCD: publish the XCD's L2, meet the other leaders
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(state, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (ld_relaxed(state) < (epoch + 1) * 8) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(state + 192 + 16 * xcc, epoch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (ld_relaxed(state + 192 + 16 * xcc) < epoch + 1) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    __syncthreads();
}
__global__ __launch_bounds__(256) void k_census(unsigned *state) {
    if (threadIdx.x == 0) atomicAdd(state + 320 + xcc_id(), 1u);
}
template <bool HIER, bool PAYLOAD>
__global__ __launch_bounds__(256) void k_all(unsigned *state, float *slots, int rounds) {
    const unsigned xcc = xcc_id(), members = state[320 + xcc];
    const int W = gridDim.x, rank = blockIdx.x;
    unsigned errs = 0;
    for (int r = 0; r < rounds; ++r) {
        if (PAYLOAD) slots[((size_t)(r & 1) * W + rank) * 256 + threadIdx.x] = (float)(r * 1024 + rank * 4 + (threadIdx.x & 3));
        if (HIER) bar_hier(state, xcc, members, (unsigned)r);
        else bar_one<false>(state, (unsigned)(r + 1) * W);
        if (PAYLOAD) {
            const int a = (rank + 1) % W, b = (rank + W / 2) % W;   // a: the next XCD, b: same XCD (W / 2 is a multiple of 8)
            const float va = slots[((size_t)(r & 1) * W + a) * 256 + threadIdx.x], vb = slots[((size_t)(r & 1) * W + b) * 256 + threadIdx.x];
            errs += va != (float)(r * 1024 + a * 4 + (threadIdx.x & 3));
            errs += vb != (float)(r * 1024 + b * 4 + (threadIdx.x & 3));
        }
    }
    if (PAYLOAD && errs) atomicAdd(state + 64, errs);
}

// ---- (c) fp32 matrix-core issue rate: 4 waves per workgroup, 8 independent accumulators per wave
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_mfma(float *out, int iters, int stride8) {
    if (stride8 && (blockIdx.x & 7) != 0) return;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = (float)threadIdx.x * 1e-3f, b = 1.0f + (float)blockIdx.x * 1e-6f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
static float timed(F &&launch, int reps = 5) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    unsigned *state;
    float *slots, *out;
    CK(hipMalloc(&state, 4096));
    CK(hipMalloc(&slots, sizeof(float) * 2 * 256 * 256));
    CK(hipMalloc(&out, sizeof(float) * 2048 * 256));
    const int R = 2000;
    auto reset = [&] { CK(hipMemset(state, 0, 1280)); };   // (keeps the census at state[320..])
    printf("barrier among W workgroups of ONE XCD (256-thread workgroups, %d rounds per launch; us per round = launch time / rounds; a launch\n"
           "with rounds = 0 costs what the column 'empty' shows)\n", R);
    for (int stride8 = 1; stride8 >= 0; --stride8) {
        printf("%s\n", stride8 ? "participants = blocks b %% 8 == 0 of a grid of 8 W (one XCD by the observed placement rule)"
                                : "participants = a plain grid of W blocks (round-robin over the XCDs): the same W spread over all 8 XCDs");
        for (int W : {8, 16, 32}) {
            const int grid = stride8 ? 8 * W : W;
            float t[5];
            reset(); t[4] = timed([&] { reset(); k_one_xcd<true, false><<<grid, 256>>>(state, slots, W, 0, stride8); });
            reset(); t[0] = timed([&] { reset(); k_one_xcd<true, false><<<grid, 256>>>(state, slots, W, R, stride8); });
            reset(); t[1] = timed([&] { reset(); k_one_xcd<true, true><<<grid, 256>>>(state, slots, W, R, stride8); });
            std::vector<unsigned> h(1024);
            CK(hipMemcpy(h.data(), state, 4096, hipMemcpyDeviceToHost));
            const unsigned err_light = h[64];
            unsigned mask = 0;
            for (int r = 0; r < W; ++r) mask |= 1u << h[16 + r];
            reset(); t[2] = timed([&] { reset(); k_one_xcd<false, false><<<grid, 256>>>(state, slots, W, R, stride8); });
            reset(); t[3] = timed([&] { reset(); k_one_xcd<false, true><<<grid, 256>>>(state, slots, W, R, stride8); });
            CK(hipMemcpy(h.data(), state, 4096, hipMemcpyDeviceToHost));
            printf("  W = %2d (XCC ids seen: mask 0x%02x): light %.2f us, light + 1 KB publish/check %.2f us (stale words: %u), agent fences %.2f us, "
                   "agent + publish %.2f us (stale: %u); empty launch %.1f us\n",
                   W, mask, (t[0] - t[4]) * 1e3 / R, (t[1] - t[4]) * 1e3 / R, err_light, (t[2] - t[4]) * 1e3 / R, (t[3] - t[4]) * 1e3 / R, h[64], t[4] * 1e3);
        }
    }
    // all 256 workgroups
    CK(hipMemset(state, 0, 4096));
    k_census<<<256, 256>>>(state);
    CK(hipDeviceSynchronize());
    {
        std::vector<unsigned> h(1024);
        CK(hipMemcpy(h.data(), state, 4096, hipMemcpyDeviceToHost));
        printf("census of a 256-block grid by XCC id:");
        for (int x = 0; x < 8; ++x) printf(" %u", h[320 + x]);
        printf("\n");
        bool even = true;
        for (int x = 0; x < 8; ++x) even = even && h[320 + x] == 32;
        float e = timed([&] { reset(); k_all<false, false><<<256, 256>>>(state, slots, 0); });
        float a = timed([&] { reset(); k_all<false, false><<<256, 256>>>(state, slots, R); });
        float b = timed([&] { reset(); k_all<false, true><<<256, 256>>>(state, slots, R); });
        CK(hipMemcpy(h.data(), state, 4096, hipMemcpyDeviceToHost));
        printf("all 256 workgroups, flat counter + agent fences: %.2f us per round, with 1 KB publish/check %.2f us (stale: %u)\n",
               (a - e) * 1e3 / R, (b - e) * 1e3 / R, h[64]);
        if (even) {
            float c = timed([&] { reset(); k_all<true, false><<<256, 256>>>(state, slots, R); });
            float d = timed([&] { reset(); k_all<true, true><<<256, 256>>>(state, slots, R); });
            CK(hipMemcpy(h.data(), state, 4096, hipMemcpyDeviceToHost));
            printf("all 256 workgroups, XCD-hierarchical: %.2f us per round, with 1 KB publish/check %.2f us (stale: %u)\n",
                   (c - e) * 1e3 / R, (d - e) * 1e3 / R, h[64]);
        } else {
            printf("census uneven: hierarchical form skipped\n");
        }
    }
    // matrix-core rate
    {
        const int iters = 20000;
        const double flop_wave = (double)iters * 8 * 2.0 * 16 * 16 * 4;
        float t32 = timed([&] { k_mfma<<<256, 256>>>(out, iters, 1); });
        float t256 = timed([&] { k_mfma<<<256, 256>>>(out, iters, 0); });
        float t512 = timed([&] { k_mfma<<<512, 256>>>(out, iters, 0); });
        printf("fp32 16x16x4 matrix-core issue, 4 waves per workgroup: 32 workgroups on one XCD %.1f TFLOP/s, 256 workgroups %.1f TFLOP/s, 512 workgroups %.1f TFLOP/s\n",
               32 * 4 * flop_wave / (t32 * 1e-3) / 1e12, 256 * 4 * flop_wave / (t256 * 1e-3) / 1e12, 512 * 4 * flop_wave / (t512 * 1e-3) / 1e12);
        printf("a level-4 / level-5 Bottleneck forward is 2.9 GFLOP (five c x c products of 0.41 GFLOP + the layer's 0.8): at the one-XCD rate that is %.0f us of\n"
               "matrix-core issue alone; the launches of round 4 take ~180 us for the whole block on the whole chip\n", 2.9e9 / (32 * 4 * flop_wave / (t32 * 1e-3)) * 1e6);
    }
    return 0;
}
