// Per-scene row sums / row broadcasts of the TransitionUp head (point_transformer_seg.py:148-161: every point of a scene gets the scene's
// mean feature appended -- ``x_b.sum(0, True) / cnt`` through linear2, ``repeat(cnt, 1)``).  Rounds 1-4 left the two reductions (the
// forward's mean, the backward's per-scene sum of the incoming rows) to torch: from 512 rows per scene on, torch reduces dim 0 across
// several workgroups with a semaphore it clears by hipMemsetAsync -- a memset NODE in a captured step, and on this stack (ROCm 7.2) the
// first replay after other work ran on the device reads garbage from such a reduction (docs/NOTEBOOK.md, round 5: every encoder gradient
// of a 2 x 131,200-point step was wrong in ~half of the replays; level 5 has exactly 2 x 512 rows there).  Here: one launch each, fixed
// summation order, nothing to clear.  Bound: latency (level 5: ~1,000 x 512 floats).
#include "pdfops_common.h"

namespace {

constexpr int SB = 256;   // 64 columns x 4 row phases

// out[s, col] = scale_s * sum over the rows of scene s of x[row, col]   (scale_s = 1 / rows when mean != 0).  grid = (ceil(c / 64), scenes)
__global__ __launch_bounds__(SB) void k_scene_sum(const int *__restrict__ offset, int c, const float *__restrict__ x, long ldx, int mean,
                                                  float *__restrict__ out) {
    __shared__ float part[4][64];
    const int s = blockIdx.y, col = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
    const long r0 = s ? offset[s - 1] : 0, r1 = offset[s];
    float a = 0.f;
    if (col < c)
        for (long r = r0 + ph; r < r1; r += 4) a += x[r * ldx + col];
    part[ph][threadIdx.x & 63] = a;
    __syncthreads();
    if (ph == 0 && col < c) {
        const float t = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
        out[(long)s * c + col] = mean ? t / (float)(r1 - r0) : t;
    }
}

// out[row, col] = scale_s * rows[s, col] for the rows of scene s (scale_s = 1 / rows of the scene when mean != 0: the backward of the mean).
// One thread per float4 of the output; the scene of a row by a walk over the (few) scene ends.
__global__ __launch_bounds__(SB) void k_scene_repeat(int scenes, const int *__restrict__ offset, long n, int c4, const float4 *__restrict__ rows,
                                                     int mean, float4 *__restrict__ out) {
    const long total = n * c4;
    for (long e = (long)blockIdx.x * SB + threadIdx.x; e < total; e += (long)gridDim.x * SB) {
        const long r = e / c4;
        int s = 0;
        while (s < scenes - 1 && r >= offset[s]) ++s;
        float4 v = rows[(long)s * c4 + (e - r * c4)];
        if (mean) {
            const float k = 1.f / (float)(offset[s] - (s ? offset[s - 1] : 0));
            v.x *= k; v.y *= k; v.z *= k; v.w *= k;
        }
        out[e] = v;
    }
}

}  // namespace

// out (scenes, c) = per-scene sums (mean != 0: means) of the rows of x (n, c; row stride ldx); offset (scenes) int32 = the scenes' end rows.
extern "C" int pdf_scene_sum_rows(int scenes, const int *offset, int c, const float *x, long ldx, int mean, float *out, void *stream) {
    if (scenes < 0 || c < 1 || ldx < c) return PDF_ERR_BAD_ARG;
    if (scenes == 0) return PDF_OK;
    if (!offset || !x || !out) return PDF_ERR_BAD_ARG;
    k_scene_sum<<<dim3((unsigned)pdf_divup(c, 64), (unsigned)scenes), SB, 0, static_cast<hipStream_t>(stream)>>>(offset, c, x, ldx, mean, out);
    return pdf_launch_status();
}

// out (n, c) = rows (scenes, c) repeated over the rows of their scene (mean != 0: divided by the scene's row count).  c % 4 == 0.
extern "C" int pdf_scene_repeat_rows(int scenes, const int *offset, long n, int c, const float *rows, int mean, float *out, void *stream) {
    if (scenes < 1 || n < 0 || c < 4 || (c & 3)) return PDF_ERR_BAD_ARG;
    if (n == 0) return PDF_OK;
    if (!offset || !rows || !out) return PDF_ERR_BAD_ARG;
    long g = pdf_divup(n * (c / 4), SB);
    if (g > PDF_MAX_BLOCKS) g = PDF_MAX_BLOCKS;
    k_scene_repeat<<<(unsigned)g, SB, 0, static_cast<hipStream_t>(stream)>>>(scenes, offset, n, c / 4, reinterpret_cast<const float4 *>(rows), mean,
                                                                             reinterpret_cast<float4 *>(out));
    return pdf_launch_status();
}
