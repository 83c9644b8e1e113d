// kNN query for gfx950 -- replaces libs/pointops/src/knn_query/knn_query_cuda_kernel.cu:60-112.
//
// Semantics are the reference's, bit for bit: every query scans the points of its own scene in
// index order, keeps an nsample-slot max-heap, replaces the root iff d2 < root (strict), and ends
// with the reference's heap sort -- so ties resolve exactly as upstream (heap-history dependent).
// d2 is evaluated as written in IEEE fp32 (this TU is compiled with -ffp-contract=off).
//
// MI355X mapping (not the reference's one-thread-per-query-over-global-memory layout):
//   * one lane = one query, 128-lane workgroups; the candidate index is wave-uniform, so candidate
//     coordinates arrive through the scalar cache (s_load) and sit in SGPRs: zero LDS/VMEM
//     traffic in the inner loop, 8 VALU ops per pair;
//   * candidates are consumed 8 at a time; a v_min3 tree + one wave-wide compare skips the
//     whole chunk when no lane improves its heap root (the common case after warm-up);
//   * the heap lives in LDS, column per lane ([slot][lane] -> conflict-free), not in scratch.
// Roofline: FP32 VALU (8 flop/pair as written); algorithmic HBM bytes are 12N+12M+8Mk.
#include "pdfops_common.h"
#include <cstdlib>

namespace {

template <int BLOCK>
__device__ __forceinline__ void heap_sift_root(float *hd, int *hi, int tid, int size, float v, int vi) {
    // == "dist[0]=v; reheap(dist, idx, size)" of knn_query_cuda_kernel.cu:15-30
    int pos = 0;
    while (true) {
        int child = 2 * pos + 1;
        if (child >= size) break;
        float cd = hd[child * BLOCK + tid];
        if (child + 1 < size) {
            float cd2 = hd[(child + 1) * BLOCK + tid];
            if (cd2 > cd) { cd = cd2; ++child; }
        }
        if (v > cd) break;
        hd[pos * BLOCK + tid] = cd;
        hi[pos * BLOCK + tid] = hi[child * BLOCK + tid];
        pos = child;
    }
    hd[pos * BLOCK + tid] = v;
    hi[pos * BLOCK + tid] = vi;
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void knn_scan_kernel(int m, int k, const float *__restrict__ xyz,
                                                         const float *__restrict__ new_xyz,
                                                         const int *__restrict__ offset,
                                                         const int *__restrict__ new_offset, int b,
                                                         int *__restrict__ idx, float *__restrict__ dist2,
                                                         const int *__restrict__ qlist, const int *__restrict__ qcount, int min_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char knn_smem[];
    float *hd = reinterpret_cast<float *>(knn_smem);  // [k][BLOCK]
    int *hi = reinterpret_cast<int *>(hd + k * BLOCK);  // [k][BLOCK]
    const int tid = threadIdx.x;
    // optional indirection: only the queries listed in qlist[0 .. *qcount) (the grid path's exact-scan "redo" list)
    int q = blockIdx.x * BLOCK + tid;
    bool active = q < m;
    if (qlist) {
        const int cnt = *qcount;
        if (cnt <= min_count || blockIdx.x * BLOCK >= cnt) return;
        active = q < cnt;
        q = active ? qlist[q] : 0;
    }

    int bt = 0;  // get_bt_idx, knn_query_cuda_kernel.cu:45-56 (bounded by b)
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (active) {
        while (bt < b - 1 && q >= new_offset[bt]) ++bt;
        qx = new_xyz[3 * (size_t)q + 0];
        qy = new_xyz[3 * (size_t)q + 1];
        qz = new_xyz[3 * (size_t)q + 2];
    }
    for (int s = 0; s < k; ++s) {
        hd[s * BLOCK + tid] = 1e10f;
        hi[s * BLOCK + tid] = -1;
    }
    float root = 1e10f;

    const int bt_lo = __builtin_amdgcn_readfirstlane(pdf_wave_min_i32(active ? bt : 0x7fffffff));
    const int bt_hi = __builtin_amdgcn_readfirstlane(pdf_wave_max_i32(active ? bt : -1));
    for (int sb = bt_lo; sb <= bt_hi; ++sb) {  // scenes touched by this wave (normally one)
        const int start = sb == 0 ? 0 : offset[sb - 1];
        const int end = offset[sb];
        const bool mine = active && bt == sb;
        float thr = mine ? root : -1.0f;  // d2 >= 0 never beats -1: foreign lanes stay idle
        int i = start;
        for (; i + 8 <= end; i += 8) {
            const float *__restrict__ p = xyz + 3 * (size_t)i;  // wave-uniform -> s_load
            float d[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float x = p[3 * u + 0], y = p[3 * u + 1], z = p[3 * u + 2];
                d[u] = pdf_sqdist3(qx - x, qy - y, qz - z);
            }
            const float m0 = fminf(fminf(d[0], d[1]), d[2]);
            const float m1 = fminf(fminf(d[3], d[4]), d[5]);
            const float m2 = fminf(fminf(d[6], d[7]), m0);
            const float dmin = fminf(m1, m2);
            if (__builtin_amdgcn_ballot_w64(dmin < thr) != 0ull) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (d[u] < thr) {
                        heap_sift_root<BLOCK>(hd, hi, tid, k, d[u], i + u);
                        thr = hd[tid];
                    }
                }
            }
        }
        for (; i < end; ++i) {
            const float *__restrict__ p = xyz + 3 * (size_t)i;
            const float x = p[0], y = p[1], z = p[2];
            const float d = pdf_sqdist3(qx - x, qy - y, qz - z);
            if (d < thr) {
                heap_sift_root<BLOCK>(hd, hi, tid, k, d, i);
                thr = hd[tid];
            }
        }
        if (mine) root = thr;
    }

    // heap_sort, knn_query_cuda_kernel.cu:33-42
    for (int i = k - 1; i > 0; --i) {
        const float v = hd[i * BLOCK + tid];
        const int vi = hi[i * BLOCK + tid];
        hd[i * BLOCK + tid] = hd[tid];
        hi[i * BLOCK + tid] = hi[tid];
        heap_sift_root<BLOCK>(hd, hi, tid, i, v, vi);
    }
    if (active) {
        int *oi = idx + (size_t)q * k;
        float *od = dist2 + (size_t)q * k;
        for (int s = 0; s < k; ++s) {
            oi[s] = hi[s * BLOCK + tid];
            od[s] = hd[s * BLOCK + tid];
        }
    }
}

constexpr int REDO_WAVE_MAX = 8192, REDO_UNR = 8;

// The same exact process for a SHORT list of queries (the grid path's redo list: typically 0-10 of 10^5..10^6 queries),
// one WAVE per query: 64 candidates are evaluated at a time in index order, a ballot marks the ones below the current heap
// root and they are sifted in ascending index order with a re-test against the updated root -- exactly the insertions the
// serial scan performs, in the same order (heap history and hence the tie order are unchanged).  One lane scanning a
// 25k-point scene alone took 1.5 ms (the whole kNN call: 2.4 ms); this takes ~20 us.
__global__ __launch_bounds__(64) void knn_redo_wave_kernel(int m, int k, const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                           const int *__restrict__ offset, const int *__restrict__ new_offset, int b,
                                                           int *__restrict__ idx, float *__restrict__ dist2,
                                                           const int *__restrict__ qlist, const int *__restrict__ qcount) {
    __shared__ float hd[128];
    __shared__ int hi[128];
    const int lane = threadIdx.x;
    const int cnt = *qcount;
    if (cnt > REDO_WAVE_MAX) return;   // long lists: the lane-per-query kernel (launched next) takes them
    for (int e = blockIdx.x; e < cnt; e += gridDim.x) {
        const int q = qlist[e];
        int bt = 0;
        while (bt < b - 1 && q >= new_offset[bt]) ++bt;
        const float qx = new_xyz[3 * (size_t)q + 0], qy = new_xyz[3 * (size_t)q + 1], qz = new_xyz[3 * (size_t)q + 2];
        for (int s = lane; s < k; s += 64) { hd[s] = 1e10f; hi[s] = -1; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_barrier();
        float thr = 1e10f;
        const int start = bt == 0 ? 0 : offset[bt - 1], end = offset[bt];
        // REDO_UNR chunks of 64 candidates are loaded together (round 4: one chunk per trip was one L2 round trip per 64 points -- ~100 us
        // per table at 100k-point scenes for a handful of queries, 1.5 ms per 12-batch pre-pass); they are then consumed in index order
        for (int c0 = start; c0 < end; c0 += 64 * REDO_UNR) {
            float dch[REDO_UNR];
#pragma unroll
            for (int u = 0; u < REDO_UNR; ++u) {
                const int i = min(c0 + 64 * u + lane, end - 1);   // (clamped address; masked below)
                const float x = xyz[3 * (size_t)i + 0], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
                dch[u] = pdf_sqdist3(qx - x, qy - y, qz - z);
            }
#pragma unroll
            for (int u = 0; u < REDO_UNR; ++u) {
                const int i0 = c0 + 64 * u;
                const float d = dch[u];
                unsigned long long mask = __builtin_amdgcn_ballot_w64(i0 + lane < end && d < thr);
                while (mask) {
                    const int j = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    const float dj = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(d), j));
                    if (dj < thr) {                       // wave-uniform: every lane performs the same sift on the shared heap
                        heap_sift_root<1>(hd, hi, 0, k, dj, i0 + j);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        __builtin_amdgcn_s_barrier();
                        thr = hd[0];
                    }
                }
            }
        }
        // heap_sort, knn_query_cuda_kernel.cu:33-42
        for (int i = k - 1; i > 0; --i) {
            const float v = hd[i];
            const int vi = hi[i];
            __builtin_amdgcn_s_barrier();
            hd[i] = hd[0];
            hi[i] = hi[0];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_barrier();
            heap_sift_root<1>(hd, hi, 0, i, v, vi);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_barrier();
        }
        for (int s = lane; s < k; s += 64) { idx[(size_t)q * k + s] = hi[s]; dist2[(size_t)q * k + s] = hd[s]; }
        __builtin_amdgcn_s_barrier();
    }
}

}  // namespace

// Exact scan restricted to the queries qlist[0 .. *qcount) (both on the device); qlist == nullptr: all m queries.
extern "C" int pdf_knn_query_list(int m, int nsample, const float *xyz, const float *new_xyz, const int *offset,
                                  const int *new_offset, int b, int *idx, float *dist2, const int *qlist,
                                  const int *qcount, void *stream) {
    if (m == 0) return PDF_OK;   // (0-size tensors carry null pointers: not an argument error)
    if (m < 0 || b < 1 || !xyz || !new_xyz || !offset || !new_offset || !idx || !dist2) return PDF_ERR_BAD_ARG;
    if (nsample < 1 || nsample > 128) return PDF_ERR_NSAMPLE;
    if (m == 0) return PDF_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int min_count = -1;
    if (qlist && getenv("PDFOPS_KNN_SERIAL_REDO") == nullptr) {
        // short redo lists (the normal case: 0-10 queries): one wave per listed query; lists longer than REDO_WAVE_MAX
        // (grid-snapped scenes where every query ties) fall through to the lane-per-query kernel below
        knn_redo_wave_kernel<<<1024, 64, 0, s>>>(m, nsample, xyz, new_xyz, offset, new_offset, b, idx, dist2, qlist, qcount);
        min_count = REDO_WAVE_MAX;
    }
    if (nsample <= 32) {
        constexpr int BLOCK = 128;
        const size_t lds = (size_t)nsample * BLOCK * 8;
        knn_scan_kernel<BLOCK><<<pdf_divup(m, BLOCK), BLOCK, lds, s>>>(m, nsample, xyz, new_xyz, offset, new_offset, b, idx, dist2, qlist, qcount, min_count);
    } else {
        constexpr int BLOCK = 64;
        const size_t lds = (size_t)nsample * BLOCK * 8;
        knn_scan_kernel<BLOCK><<<pdf_divup(m, BLOCK), BLOCK, lds, s>>>(m, nsample, xyz, new_xyz, offset, new_offset, b, idx, dist2, qlist, qcount, min_count);
    }
    return pdf_launch_status();
}

extern "C" int pdf_knn_query(int m, int nsample, const float *xyz, const float *new_xyz, const int *offset,
                             const int *new_offset, int b, int *idx, float *dist2, void *stream) {
    return pdf_knn_query_list(m, nsample, xyz, new_xyz, offset, new_offset, b, idx, dist2, nullptr, nullptr, stream);
}

// which squared-distance arithmetic the geometry TUs of this library were compiled with (pdfops_common.h: pdf_sqdist3)
extern "C" int pdf_dist_fma_mode(void) { return PDF_DIST_FMA; }
