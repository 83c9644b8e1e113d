// Shared helpers for the libpdfops HIP translation units (gfx950 / wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../include/pdfops.h"

#define PDF_WAVE 64
#define PDF_MAX_DEVICES_LDS 64

// Launch-status helper: report a launch-configuration error as the entry point's return value.
static inline int pdf_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PDF_OK : (int)e;
}

// Raised dynamic-LDS limit of one kernel, remembered PER DEVICE (the attribute belongs to the device's code object: a process that drives
// several GPUs must set it on each; rounds 1-5 set it once per process).  `state` = the call site's own table (zero-initialised static):
// 0 = not asked yet on this device, 1 = accepted, 2 = refused.  Racing first calls on one device both set the same value: harmless.
struct PdfLdsLimit { std::atomic<int> state[PDF_MAX_DEVICES_LDS]; };
static inline bool pdf_lds_limit_raised(PdfLdsLimit &site, const void *kernel, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PDF_MAX_DEVICES_LDS) return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
    int st = site.state[dev].load(std::memory_order_acquire);
    if (st == 0) {
        st = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? 1 : 2;
        if (st == 2) (void)hipGetLastError();
        site.state[dev].store(st, std::memory_order_release);
    }
    return st == 1;
}

static inline int pdf_divup(long a, long b) { return (int)((a + b - 1) / b); }

// Cap for grid-stride launches of the HBM-bound kernels: 256 CUs x 8 resident blocks.
#define PDF_MAX_BLOCKS 2048

__device__ __forceinline__ int pdf_lane() { return (int)(threadIdx.x & 63); }

// wave64 butterfly reductions (all lanes receive the result)
__device__ __forceinline__ int pdf_wave_min_i32(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int pdf_wave_max_i32(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ unsigned long long pdf_wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        unsigned long long w = __shfl_xor(v, o, 64);
        v = w > v ? w : v;
    }
    return v;
}
__device__ __forceinline__ float pdf_wave_sum_f32(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Squared distance of the geometry kernels (kNN, ball query, FPS) from the three coordinate differences.
//   PDF_DIST_FMA undefined / 0 : the reference's expression as written in IEEE fp32 -- dx*dx + dy*dy + dz*dz, every product and sum rounded
//                                (knn_query_cuda_kernel.cu:92, sampling_cuda_kernel.cu:54, ball_query_cuda_kernel.cu:95); the TUs that use it
//                                are compiled with -ffp-contract=off.  The default: the one variant every toolchain reproduces.
//   PDF_DIST_FMA == 1          : fmaf(dz, dz, fmaf(dy, dy, dx*dx))  -- left-to-right contraction of the same expression.
//   PDF_DIST_FMA == 2          : fmaf(dz, dz, fmaf(dx, dx, dy*dy))  -- what LLVM's and GCC's contraction emit for it (first product of
//                                `a*a + b*b` fused, second kept: clang 22 / gcc 11 -ffp-contract=fast here); nvcc -O2 (fmad on,
//                                libs/pointops/setup.py:29) is LLVM-based and most likely emits this form.
// The variants exist so that a user validating against an NVIDIA build of libs/pointops has a bit-matching mode (libpdfops_fma{1,2}.so,
// selected with PDFOPS_DIST_FMA at import).  Every variant is monotone in |dx|, |dy|, |dz| (each rounding is), which
// is all the pruning arguments of the grid kNN and the bucketed FPS need.
#ifndef PDF_DIST_FMA
#define PDF_DIST_FMA 0
#endif
__host__ __device__ __forceinline__ float pdf_sqdist3(float dx, float dy, float dz) {
#if PDF_DIST_FMA == 1
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
#elif PDF_DIST_FMA == 2
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dx, dx, dy * dy));
#else
    return dx * dx + dy * dy + dz * dz;
#endif
}

// Monotone map float -> uint32 (total order incl. negatives), used to pack (value, key) pairs.
__device__ __forceinline__ unsigned pdf_f32_ordered(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Hardware fp32 atomic add at agent scope (global_atomic_add_f32 with -munsafe-fp-atomics).
__device__ __forceinline__ void pdf_atomic_add(float *p, float v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Workgroup -> work-chunk remap that gives every XCD ONE contiguous stretch of the chunk sequence: the dispatcher places workgroup b
// on XCD b mod 8 (observed, /opt/skills/guides/MI355X_MICROARCH.md), so chunk ids are dealt XCD-major.  Bijective for any grid size g;
// a pure speed choice (locality in the per-XCD L2), never a correctness assumption.
#define PDF_XCDS 8
__device__ __forceinline__ unsigned pdf_xcd_chunked_block(unsigned b, unsigned g) {
    const unsigned q = g / PDF_XCDS, r = g % PDF_XCDS, x = b % PDF_XCDS;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / PDF_XCDS;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// In-kernel BatchNorm finalize in the CONSUMER ("consumer-side reduction", round 4).  A train-mode BatchNorm on this path is
//     producer kernel (emits per-workgroup partial rows [sum | sum of squares])  ->  reducer launch (k_bn_finalize, ~5 us + a launch
//     boundary, ~240 of them per step)  ->  consumer kernel (reads scale | shift).
// Round 3 folded the reducer into the PRODUCER (last-arriver tails): slower -- every producing workgroup had to release its row across
// the XCDs (an L2 write-back while the kernel still streams its output).  Here the rows cross a kernel boundary as before (free), and the
// CONSUMER starts with the reduction: the first workgroups to arrive (role ticket) each reduce 16 channels of the rows and publish
// scale | shift as 8-byte {tag, value} granules (one write-through store each: the data is the flag -- /opt/skills/guides/
// cdna_hip_programming.md, Guideline 16, form R2); every workgroup then polls the granules of the channels it needs (relaxed agent-scope
// loads: L2-served, no fence) while its own prologue work (weight staging, index loads) is in flight.  The granules and the ticket are
// zeroed by workgroup 0 of the PRODUCER kernel (complete and visible at the boundary), so the scratch needs no host-side
// initialisation and the tag is the constant 1.  Order of every sum: fixed by the row indices (bit-reproducible).
struct PdfRowsBn {
    const float *rows; int nrows; int c; double count;           // partial rows [nrows][2 c]
    const float *gamma, *beta; float *running_mean, *running_var; float eps, momentum;
    float *coef;                                                  // (4 c) scale | shift | mean | rstd: written for the later passes / the backward
    unsigned long long *gran;                                     // (2 c) granules of scale | shift
    unsigned *sync;                                               // [0] role ticket
};
#define PDF_HO_WORDS(c) (2 * (size_t)(c) * 2 + 4)                 /* 32-bit words of handoff scratch for a c-channel norm: granules + ticket */
#define PDF_HO_FLOATS PDF_HO_WORDS(1024)                          /* what every `partial` scratch reserves in FRONT of its rows (c <= 1024) */
int pdf_rowlin_forward_stats_ho(long n, int k, int o, const float *x, long ldx, const float *w, const float *bias, const float *scale,
                                const float *shift, int relu, float *y, long ldy, float *rows_out, void *handoff, int *rows, int mma_input,
                                void *stream);   // rowlin.hip
int pdf_bn_apply_rows(long n, int c, const float *x, const float *res, const float *rows, int nrows, const float *gamma, const float *beta,
                      float *running_mean, float *running_var, float eps, float momentum, float *coef, void *handoff, int relu, float *y,
                      void *stream);             // pointwise.hip

// producer side: any ONE workgroup of the kernel that writes the rows (before or after its own work; the kernel boundary orders it)
__device__ __forceinline__ void pdf_handoff_zero(unsigned long long *gran, int c, unsigned *sync) {
    for (int e = threadIdx.x; e < 2 * c; e += blockDim.x) gran[e] = 0ull;
    if (threadIdx.x == 0) sync[0] = 0u;
}

// consumer side, called by ALL threads of EVERY workgroup (blockDim.x == NT, a multiple of 64) before the first use of the coefficients.
// lds_coef: 2 c floats (scale | shift) filled for the whole block; lds_red: 2 * (NT / 16) * 17 doubles of scratch.
// Roles are static -- workgroup i (linear id) reduces channel slice i -- and NO atomic is involved: a ticket word taken by all 2,048
// workgroups of a bandwidth-bound consumer serialises at ~12 ns per arrival (measured: +5 us per launch instead of -5).  HIP promises no
// dispatch order, so a poller that sees no progress for a long time reduces the missing slices itself (same rows, same order, same
// values: idempotent); on this hardware workgroups start in id order and the path is never taken.
template <int NT>
__device__ __forceinline__ void pdf_bn_reduce_slice(const PdfRowsBn &b, int sl, double *lds_red) {
    constexpr int RL = NT / 16;
    const int c = b.c, cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int ch = sl * 16 + cl;
    double s = 0.0, ss = 0.0;
    if (ch < c) {
        const float *p0 = b.rows + ch, *p1 = b.rows + c + ch;
        const size_t stride = 2 * (size_t)c;
        for (int r = rl; r < b.nrows; r += 8 * RL) {   // eight rows in flight per lane; clamped address + select (no load under a branch)
            float v0[8], v1[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int rr = r + t * RL < b.nrows ? r + t * RL : b.nrows - 1;
                v0[t] = p0[(size_t)rr * stride]; v1[t] = p1[(size_t)rr * stride];
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const bool ok = r + t * RL < b.nrows;
                s += ok ? (double)v0[t] : 0.0; ss += ok ? (double)v1[t] : 0.0;
            }
        }
    }
    __syncthreads();   // (lds_red free again)
    lds_red[(0 * RL + rl) * 17 + cl] = s;
    lds_red[(1 * RL + rl) * 17 + cl] = ss;
    __syncthreads();
    if (rl == 0 && ch < c) {
        double a = 0.0, q = 0.0;
        for (int k = 0; k < RL; ++k) { a += lds_red[(0 * RL + k) * 17 + cl]; q += lds_red[(1 * RL + k) * 17 + cl]; }
        const double mean = a / b.count;
        double var = q / b.count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)b.eps));
        const float sc = b.gamma[ch] * rstd, sh = b.beta[ch] - (float)mean * sc;
        b.coef[ch] = sc; b.coef[c + ch] = sh; b.coef[2 * c + ch] = (float)mean; b.coef[3 * c + ch] = rstd;
        if (b.running_mean && b.sync != nullptr) {   // (running statistics: by the slice's OWNER only -- sync == nullptr marks a take-over)
            const double unbiased = b.count > 1.0 ? var * b.count / (b.count - 1.0) : var;
            b.running_mean[ch] = (1.f - b.momentum) * b.running_mean[ch] + b.momentum * (float)mean;
            b.running_var[ch] = (1.f - b.momentum) * b.running_var[ch] + b.momentum * (float)unbiased;
        }
        __hip_atomic_store(b.gran + ch, (1ull << 32) | (unsigned long long)__float_as_uint(sc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(b.gran + c + ch, (1ull << 32) | (unsigned long long)__float_as_uint(sh), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int NT>
__device__ __forceinline__ void pdf_bn_coef_inkernel(const PdfRowsBn &b, float *lds_coef, double *lds_red, unsigned *lds_word) {
    (void)lds_word;
    const int c = b.c, nsl = (c + 15) / 16;
    const unsigned me = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nblk = gridDim.x * gridDim.y * gridDim.z;
    for (unsigned sl = me; sl < (unsigned)nsl; sl += nblk) pdf_bn_reduce_slice<NT>(b, (int)sl, lds_red);   // (block-uniform trip count)
    // every workgroup: sweep the granules until every tag is set
    for (unsigned spins = 0;; ++spins) {
        bool ok = true;
        for (int e = threadIdx.x; e < 2 * c; e += NT) {
            const unsigned long long g = __hip_atomic_load(b.gran + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = ok && (g >> 32) == 1ull;
            lds_coef[e] = __uint_as_float((unsigned)g);
        }
        if (__syncthreads_and(ok)) break;
        if (spins == (1u << 14)) {   // ~50 ms without the owners: take the reduction over (never observed: workgroups start in id order)
            PdfRowsBn mine = b;
            mine.sync = nullptr;
            for (int sl = 0; sl < nsl; ++sl) pdf_bn_reduce_slice<NT>(mine, sl, lds_red);
        }
        __builtin_amdgcn_s_sleep(8);
    }
}
