// ball_query / random_ball_query for gfx950 -- replaces libs/pointops/src/ball_query/ball_query_cuda_kernel.cu:58-136 and
// libs/pointops/src/random_ball_query/random_ball_query_cuda_kernel.cu:58-123.  Neither has an in-tree caller on the
// PointTransformer-V1 / PDF path; they complete the `pointops` API.
//
// Mapping: ONE WAVE PER QUERY.  The scene is read 64 points per trip (one point per lane, coalesced 768-B segments), the
// as-written fp32 distance (no FMA contraction: this TU is built with -ffp-contract=off) is tested against the shell
// [min_radius^2, max_radius^2) -- or d2 <= 1e-5 -- and a ballot + popcount prefix appends the accepted points to an LDS
// list IN INDEX ORDER, which is the order the reference's serial loop produces.
//   * ball_query then applies the reference's `heap_sort` to that list.  Upstream never heapifies the array first, so the
//     result is not sorted: it is the fixed permutation that "swap root with slot i, sift the root down" produces from an
//     index-ordered list, and later picks every (count / nsample)-th entry.  Lane 0 replays exactly those swaps on LDS.
//     The reference keeps at most 2048 candidates on the thread's stack and overruns it beyond that (undefined); here the
//     list stops at 2048 accepted points (the first 2048 in index order).
//   * when more than nsample candidates exist the reference stores the candidate INDEX, converted to float, as dist2
//     (`dist2[i] = candi_idx[index]`, ball_query_cuda_kernel.cu:120); reproduced as written.
//   * random_ball_query walks the caller's permutation `order` and keeps the first nsample accepted points.
#include "pdfops_common.h"

namespace {

constexpr int BQ_WAVES = 2;          // waves (= queries) per workgroup
constexpr int BQ_CAP = 2048;         // candi_dist[2048], ball_query_cuda_kernel.cu:85-86

__device__ __forceinline__ int scene_of(int q, const int *__restrict__ new_offset, int b) {
    int bt = 0;  // get_bt_idx, ball_query_cuda_kernel.cu:44-55 (bounded by b)
    while (bt < b - 1 && q >= new_offset[bt]) ++bt;
    return bt;
}

__device__ __forceinline__ bool in_shell(float d2, float min_r2, float max_r2) {
    // `d2 <= 1e-5` compares in double upstream; the largest float not above the double 1e-5 is 1e-5f itself
    return d2 <= 1e-5f || (d2 >= min_r2 && d2 < max_r2);
}

// ball_query_utils::reheap, ball_query_cuda_kernel.cu:15-30
__device__ __forceinline__ void bq_reheap(float *dist, int *idx, int k) {
    int root = 0, child = 1;
    while (child < k) {
        if (child + 1 < k && dist[child + 1] > dist[child]) child++;
        if (dist[root] > dist[child]) return;
        const float td = dist[root]; dist[root] = dist[child]; dist[child] = td;
        const int ti = idx[root]; idx[root] = idx[child]; idx[child] = ti;
        root = child;
        child = root * 2 + 1;
    }
}

__global__ __launch_bounds__(BQ_WAVES * 64) void ball_query_kernel(int m, int nsample, float min_radius, float max_radius,
                                                                   const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                                   const int *__restrict__ offset, const int *__restrict__ new_offset,
                                                                   int b, int *__restrict__ idx, float *__restrict__ dist2) {
    __shared__ float s_dist[BQ_WAVES][BQ_CAP];
    __shared__ int s_idx[BQ_WAVES][BQ_CAP];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = blockIdx.x * BQ_WAVES + wave;
    if (q >= m) return;  // wave-uniform; no block-wide barrier below
    float *cd = s_dist[wave];
    int *ci = s_idx[wave];

    const int bt = scene_of(q, new_offset, b);
    const int start = bt == 0 ? 0 : offset[bt - 1], end = offset[bt];
    const float max_r2 = max_radius * max_radius, min_r2 = min_radius * min_radius;
    const float qx = new_xyz[3 * (size_t)q + 0], qy = new_xyz[3 * (size_t)q + 1], qz = new_xyz[3 * (size_t)q + 2];

    int num = 0;  // wave-uniform
    for (int base = start; base < end && num < BQ_CAP; base += 64) {
        const int i = base + lane;
        bool ok = false;
        float d2 = 0.f;
        if (i < end) {
            const float x = xyz[3 * (size_t)i + 0], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
            d2 = pdf_sqdist3(qx - x, qy - y, qz - z);
            ok = in_shell(d2, min_r2, max_r2);
        }
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(ok);
        const int pos = num + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
        if (ok && pos < BQ_CAP) { cd[pos] = d2; ci[pos] = i; }
        num += __builtin_popcountll(mask);
    }
    if (num > BQ_CAP) num = BQ_CAP;
    // LDS writes of all lanes visible to lane 0: same wave, the LDS queue is in order; the fence stops the compiler
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();

    if (lane == 0) {  // ball_query_utils::heap_sort, ball_query_cuda_kernel.cu:33-42 (applied to an un-heapified list, as upstream)
        for (int i = num - 1; i > 0; --i) {
            const float td = cd[0]; cd[0] = cd[i]; cd[i] = td;
            const int ti = ci[0]; ci[0] = ci[i]; ci[i] = ti;
            bq_reheap(cd, ci, i);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();

    int *oi = idx + (size_t)q * nsample;
    float *od = dist2 + (size_t)q * nsample;
    if (num <= nsample) {  // ball_query_cuda_kernel.cu:103-111
        for (int s = lane; s < nsample; s += 64) {
            oi[s] = s < num ? ci[s] : -1;
            od[s] = s < num ? cd[s] : 1e10f;
        }
    } else {               // ball_query_cuda_kernel.cu:113-121
        const float sep = static_cast<float>(num) / nsample;
        for (int s = lane; s < nsample; s += 64) {
            int index = static_cast<int>(sep * s);
            if (index > num - 1) index = num - 1;  // (never taken for exact arithmetic; keeps a rounded-up product inside the list)
            oi[s] = ci[index];
            od[s] = static_cast<float>(ci[index]);  // sic: the reference writes the index here
        }
    }
}

__global__ __launch_bounds__(BQ_WAVES * 64) void random_ball_query_kernel(int m, int nsample, float min_radius, float max_radius,
                                                                          const int *__restrict__ order,
                                                                          const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                                          const int *__restrict__ offset, const int *__restrict__ new_offset,
                                                                          int b, int *__restrict__ idx, float *__restrict__ dist2) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = blockIdx.x * BQ_WAVES + wave;
    if (q >= m) return;
    const int bt = scene_of(q, new_offset, b);
    const int start = bt == 0 ? 0 : offset[bt - 1], end = offset[bt];
    const float max_r2 = max_radius * max_radius, min_r2 = min_radius * min_radius;
    const float qx = new_xyz[3 * (size_t)q + 0], qy = new_xyz[3 * (size_t)q + 1], qz = new_xyz[3 * (size_t)q + 2];
    int *oi = idx + (size_t)q * nsample;
    float *od = dist2 + (size_t)q * nsample;

    int cnt = 0;  // wave-uniform; random_ball_query_cuda_kernel.cu:88-103
    for (int base = start; base < end && cnt < nsample; base += 64) {
        const int i = base + lane;
        bool ok = false;
        float d2 = 0.f;
        int j = -1;
        if (i < end) {
            j = order[i];
            const float x = xyz[3 * (size_t)j + 0], y = xyz[3 * (size_t)j + 1], z = xyz[3 * (size_t)j + 2];
            d2 = pdf_sqdist3(qx - x, qy - y, qz - z);
            ok = in_shell(d2, min_r2, max_r2);
        }
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(ok);
        const int pos = cnt + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
        if (ok && pos < nsample) { oi[pos] = j; od[pos] = d2; }
        cnt += __builtin_popcountll(mask);
    }
    if (cnt > nsample) cnt = nsample;
    for (int s = cnt + lane; s < nsample; s += 64) { oi[s] = -1; od[s] = 1e10f; }  // :105-110
}

}  // namespace

extern "C" int pdf_ball_query(int m, int nsample, float min_radius, float max_radius, const float *xyz, const float *new_xyz,
                              const int *offset, const int *new_offset, int b, int *idx, float *dist2, void *stream) {
    if (m == 0) return PDF_OK;   // (0-size tensors carry null pointers: not an argument error)
    if (m < 0 || b < 1 || !xyz || !new_xyz || !offset || !new_offset || !idx || !dist2) return PDF_ERR_BAD_ARG;
    if (nsample < 1 || nsample > BQ_CAP) return PDF_ERR_NSAMPLE;
    if (!(min_radius < max_radius)) return PDF_ERR_BAD_ARG;  // query.py:46 asserts the same
    if (m == 0) return PDF_OK;
    ball_query_kernel<<<pdf_divup(m, BQ_WAVES), BQ_WAVES * 64, 0, static_cast<hipStream_t>(stream)>>>(
        m, nsample, min_radius, max_radius, xyz, new_xyz, offset, new_offset, b, idx, dist2);
    return pdf_launch_status();
}

extern "C" int pdf_random_ball_query(int m, int nsample, float min_radius, float max_radius, const int *order, const float *xyz,
                                     const float *new_xyz, const int *offset, const int *new_offset, int b, int *idx, float *dist2,
                                     void *stream) {
    if (m == 0) return PDF_OK;
    if (m < 0 || b < 1 || !order || !xyz || !new_xyz || !offset || !new_offset || !idx || !dist2) return PDF_ERR_BAD_ARG;
    if (nsample < 1) return PDF_ERR_NSAMPLE;
    if (!(min_radius < max_radius)) return PDF_ERR_BAD_ARG;
    if (m == 0) return PDF_OK;
    random_ball_query_kernel<<<pdf_divup(m, BQ_WAVES), BQ_WAVES * 64, 0, static_cast<hipStream_t>(stream)>>>(
        m, nsample, min_radius, max_radius, order, xyz, new_xyz, offset, new_offset, b, idx, dist2);
    return pdf_launch_status();
}
