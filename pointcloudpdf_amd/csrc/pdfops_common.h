// Shared helpers for the libpdfops HIP translation units (gfx950 / wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pdfops.h"

#define PDF_WAVE 64

// Launch-status helper: report a launch-configuration error as the entry point's return value.
static inline int pdf_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PDF_OK : (int)e;
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
