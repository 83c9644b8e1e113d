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
// In-launch reduction of per-workgroup partial rows ("tail"): the tiny reducer launches of rounds 1-2 (k_colsum / k_bn_finalize, ~5 us
// each + a launch boundary) folded into the kernel that produces the rows.  /opt/skills/guides/cdna_hip_programming.md section 6,
// Guideline 16, counter form: a workgroup stores its row (plain stores), ALL its waves drain their stores (s_waitcnt vmcnt(0)), one lane
// releases at agent scope and takes a ticket; the LAST arriver of a group of PDF_TAIL_G workgroups acquires, adds the group's rows in row
// order into a group row (doubles), publishes it the same way and takes the launch's second-level ticket; the last group finisher adds
// the group rows in group order.  The order of every sum is fixed by the row indices, never by the arrival order: bit-reproducible.
// Tickets are zero when a launch starts and are reset by their last arriver; they live in a caller-owned, zero-initialised word array
// bound to the stream (pdf_tickets_bind): launches of one stream do not overlap, launches of different streams use different arrays.
#define PDF_TAIL_G 32
#define PDF_TICKET_WORDS 4096
unsigned *pdf_tickets_for(hipStream_t s, long n);
int pdf_mma_input_mode();   // api.hip: 0 fp32 / 1 fp16 / 2 bfloat16 operands of the streaming Linear products (pdf_set_mma_input)   // nullptr: none bound for this stream (callers then keep their separate reducer launch)

__device__ __forceinline__ bool pdf_arrive_last(unsigned *ticket, unsigned expected, volatile unsigned *lds_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // every storing wave drains its stores
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");            // buffer_wbl2 sc1: the rows leave this XCD's L2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (the compiler may drop the wait behind the write-back: restated)
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned last = (t + 1u == expected) ? 1u : 0u;
        if (last) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero again for the stream's next launch
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");        // buffer_inv sc1: no stale line of another workgroup's rows
        }
        *lds_flag = last;
    }
    __syncthreads();
    return *lds_flag != 0u;
}

// Called by ALL threads of every workgroup of a 1-D range of `nb` workgroups (index `bid`) after the workgroup stored the `ncols` floats
// rows[bid * row_stride + colmap(j)], j < ncols.  Returns true in exactly ONE workgroup -- after emit(j, sum_j) ran there for every j
// (one thread each; sum in double, rows 0 .. nb-1 in order within groups of PDF_TAIL_G, groups in order).  tickets: 1 + ceil(nb / G) words;
// grows: ceil(nb / G) * ncols doubles of scratch.  lds_flag: one LDS word that is free for the duration of the call.
template <typename ColMap, typename Emit>
__device__ __forceinline__ bool pdf_tail_sum(unsigned *tickets, double *grows, const float *rows, size_t row_stride, int ncols, ColMap colmap,
                                             unsigned nb, unsigned bid, volatile unsigned *lds_flag, Emit emit) {
    const unsigned ng = (nb + PDF_TAIL_G - 1) / PDF_TAIL_G, g = bid / PDF_TAIL_G, r0 = g * PDF_TAIL_G;
    const unsigned gsz = nb - r0 < PDF_TAIL_G ? nb - r0 : PDF_TAIL_G;
    if (!pdf_arrive_last(tickets + 1 + g, gsz, lds_flag)) return false;
    for (int j = threadIdx.x; j < ncols; j += blockDim.x) {
        const float *src = rows + (size_t)r0 * row_stride + colmap(j);
        double s = 0.0;
        unsigned r = 0;
        for (; r + 8 <= gsz; r += 8) {   // eight loads in flight, added in row order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(r + u) * row_stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)v[u];
        }
        for (; r < gsz; ++r) s += (double)src[(size_t)r * row_stride];
        if (ng == 1) emit(j, s); else grows[(size_t)g * ncols + j] = s;
    }
    if (ng == 1) return true;
    if (!pdf_arrive_last(tickets, ng, lds_flag)) return false;
    for (int j = threadIdx.x; j < ncols; j += blockDim.x) {
        double s = 0.0;
        for (unsigned gg = 0; gg < ng; ++gg) s += grows[(size_t)gg * ncols + j];
        emit(j, s);
    }
    return true;
}
