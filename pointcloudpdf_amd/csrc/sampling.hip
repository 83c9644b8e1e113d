// Farthest point sampling for gfx950 -- replaces libs/pointops/src/sampling/sampling_cuda_kernel.cu:14-171.
//
// Result contract (bit-exact with the reference): sample 0 of a scene is its first point; sample j is
// the arg-max over the scene of tmp[k] = min_{i<j} d(k, sample_i), d as written in IEEE fp32
// (-ffp-contract=off).  Among equal maxima the reference's strided scan + shared-memory tree
// (sampling_cuda_kernel.cu:5-10, :49-123) picks the point minimising
//     ( bitreverse_{log2 bs}( (k - start) mod bs ),  k ),   bs = opt_n_threads(n)  (cuda_utils.h:11-14)
// which is reproduced here through a packed 64-bit key, independent of OUR block shape:
//     [ ordered(tmp) : 32 | ~bitrev(slot) : 10 | ~(k - start) : 22 ]   -> one u64 max-reduction.
//
// This file holds the plain ("v1") kernel: one 1024-lane workgroup per scene, points strided over
// lanes, wave64 butterfly + one LDS hop per iteration (1 barrier/iteration, double-buffered slots).
// The bucketed kernel (sampling_bucketed.hip) prunes whole spatial buckets and is the fast path.
#include "pdfops_common.h"
#include <math.h>

namespace {

constexpr int FPS_BS = 1024;
constexpr int FPS_NW = FPS_BS / 64;
constexpr int FPS_REL_BITS = 22;
constexpr unsigned FPS_REL_MASK = (1u << FPS_REL_BITS) - 1u;

__device__ __forceinline__ unsigned fps_keybits(int rel, int bs_ref_mask, int bs_ref_log2) {
    const unsigned slot = (unsigned)rel & (unsigned)bs_ref_mask;
    const unsigned rev = bs_ref_log2 ? (__brev(slot) >> (32 - bs_ref_log2)) : 0u;
    return ((~rev & 0x3ffu) << FPS_REL_BITS) | (~(unsigned)rel & FPS_REL_MASK);
}

__global__ __launch_bounds__(FPS_BS) void fps_plain_kernel(const float *__restrict__ xyz,
                                                           const int *__restrict__ offset,
                                                           const int *__restrict__ new_offset,
                                                           float *__restrict__ tmp, int *__restrict__ idx,
                                                           int bs_ref_log2) {
    __shared__ unsigned long long slots[2][FPS_NW];
    const int bid = blockIdx.x;
    const int tid = threadIdx.x;
    const int start_n = bid == 0 ? 0 : offset[bid - 1];
    const int end_n = offset[bid];
    const int start_m = bid == 0 ? 0 : new_offset[bid - 1];
    const int end_m = new_offset[bid];
    const int bs_ref_mask = (1 << bs_ref_log2) - 1;
    if (end_m <= start_m) return;
    if (tid == 0) idx[start_m] = start_n;
    int old = start_n;
    // "empty" result of the reference (best = -1, besti = start_n): decodes to rel = 0
    const unsigned long long empty_key = ((unsigned long long)pdf_f32_ordered(-1.0f) << 32) | FPS_REL_MASK;
    for (int j = start_m + 1; j < end_m; ++j) {
        const float x1 = xyz[3 * (size_t)old + 0];
        const float y1 = xyz[3 * (size_t)old + 1];
        const float z1 = xyz[3 * (size_t)old + 2];
        const bool first = (j == start_m + 1);
        unsigned long long best = empty_key;
        for (int k = start_n + tid; k < end_n; k += FPS_BS) {
            const float x2 = xyz[3 * (size_t)k + 0];
            const float y2 = xyz[3 * (size_t)k + 1];
            const float z2 = xyz[3 * (size_t)k + 2];
            const float d = pdf_sqdist3(x2 - x1, y2 - y1, z2 - z1);
            const float t = first ? 1e10f : tmp[k];  // reference pre-fills tmp with 1e10 (sampling.py:19)
            const float d2 = fminf(d, t);
            tmp[k] = d2;
            const unsigned long long key =
                ((unsigned long long)pdf_f32_ordered(d2) << 32) | fps_keybits(k - start_n, bs_ref_mask, bs_ref_log2);
            best = key > best ? key : best;
        }
        best = pdf_wave_max_u64(best);
        const int par = j & 1;
        if ((tid & 63) == 0) slots[par][tid >> 6] = best;
        __syncthreads();
        unsigned long long w = slots[par][tid & (FPS_NW - 1)];
#pragma unroll
        for (int o = FPS_NW / 2; o >= 1; o >>= 1) {
            unsigned long long v = __shfl_xor(w, o, 64);
            w = v > w ? v : w;
        }
        old = start_n + (int)(~(unsigned)w & FPS_REL_MASK);
        if (tid == 0) idx[j] = old;
    }
}

}  // namespace

// libs/pointops/src/cuda_utils.h:11-14 -- the reference block size for a largest-scene size n
// (double log ratio truncated; exact powers of two are stable with glibc for n <= 2^20)
extern "C" int pdf_fps_reference_block_log2(int n) {
    if (n < 1) return 0;
    const int pow_2 = (int)(log((double)n) / log(2.0));
    int l = pow_2 < 0 ? 0 : pow_2;
    if (l > 10) l = 10;
    return l;
}

extern "C" int pdf_farthest_point_sampling(int b, int n, const float *xyz, const int *offset, const int *new_offset,
                                           float *tmp, int *idx, void *stream) {
    if (b < 1 || n < 1 || !xyz || !offset || !new_offset || !tmp || !idx) return PDF_ERR_BAD_ARG;
    if (n > (1 << FPS_REL_BITS)) return PDF_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    fps_plain_kernel<<<b, FPS_BS, 0, s>>>(xyz, offset, new_offset, tmp, idx, pdf_fps_reference_block_log2(n));
    return pdf_launch_status();
}
