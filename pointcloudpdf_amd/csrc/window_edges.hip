// Edge tables of the StratifiedTransformer's window partitions (BASELINE config 5), built on the device.
//
// The reference (pointcept/models/stratified_transformer/stratified_transformer_v1m1_origin.py:45-127, 468-536) forms, per Swin block,
//   * a voxel -> member table of the FINE window partition padded to the fullest window, expands ALL (member, member) pairs of every
//     window through a dense (windows x kmax x kmax) boolean mask,
//   * the same for the COARSE (2 x window) partition with the predicates "key is an FPS-downsampled point" and "key lies in another fine
//     window",
//   * concatenates both lists and sorts the ~10^7 edges by query.
// What that produces is fully determined per QUERY: its row of the CSR-by-query table is
//     [ the points of its fine window, ascending ]  ++  [ the downsampled points of its coarse window that lie in another fine window,
//     ascending ]
// (the stable sort keeps window order, and inside a window the mask expansion walks members in ascending point id).  So the builder
// needs no pair expansion and no edge-sized sort: with the points ordered by fine-window key and the downsampled points ordered by
// coarse-window key (two point-sized stable sorts, done by the caller), one launch counts every query's row, and -- after the caller's
// scan of the counts -- one launch writes the rows: one WAVE per query, 64 candidates per trip, ballot + prefix popcount for the
// in-order compaction of the coarse candidates, coalesced stores of index_0 / index_1 and of the edge's quantised relative position
// (WindowAttention.relative_position_index, :282-292) in the same pass.
// Bit-identical to the reference's tables (tests/test_gpu_pointops2.py against oracle/window_tables.py, the restatement of :45-127).
// Compiled with -ffp-contract=off: the quantisation below must round exactly like the torch expression it replaces.
#include "pdfops_common.h"

namespace we {

typedef long long i64;

__device__ __forceinline__ int lower_bound(const i64 *a, int n, i64 key) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ int upper_bound(const i64 *a, int n, i64 key) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] <= key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// one thread per query: its segment of the fine order, its segment of the downsampled coarse order, the length of its row
__global__ void k_count(int n, const i64 *__restrict__ kf_sorted, const i64 *__restrict__ kf, int m, const i64 *__restrict__ kcd_sorted,
                        const i64 *__restrict__ kc, const i64 *__restrict__ wk, const i64 *__restrict__ wkd, int *__restrict__ count,
                        int4 *__restrict__ seg) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const i64 key = kf[q], key2 = kc[q], w = wk[q];
    const int fs = lower_bound(kf_sorted, n, key), fe = upper_bound(kf_sorted, n, key);
    const int cs = lower_bound(kcd_sorted, m, key2), ce = upper_bound(kcd_sorted, m, key2);
    int c = fe - fs;
    for (int t = cs; t < ce; ++t) c += wkd[t] != w ? 1 : 0;
    count[q] = c;
    seg[q] = make_int4(fs, fe, cs, ce);
}

struct Quant { float c2w, qs; int vmax; };

// one wave per query: the row of the CSR table + the quantised relative positions of its edges
__global__ __launch_bounds__(256) void k_fill(int n, const int *__restrict__ offsets, const int4 *__restrict__ seg,
                                              const int *__restrict__ order_f, const int *__restrict__ order_cd,
                                              const i64 *__restrict__ wk, const i64 *__restrict__ wkd, const float *__restrict__ xyz, Quant Q,
                                              i64 *__restrict__ index0, int *__restrict__ index1, int *__restrict__ rel, int *__restrict__ flag) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= n) return;
    const int4 s = seg[q];
    const i64 w = wk[q];
    const float qx = xyz[3 * (size_t)q], qy = xyz[3 * (size_t)q + 1], qz = xyz[3 * (size_t)q + 2];
    bool bad = false;
    auto emit = [&](size_t e, int k) {
        index0[e] = q;
        index1[e] = k;
        if (rel) {
            const float d[3] = {qx - xyz[3 * (size_t)k], qy - xyz[3 * (size_t)k + 1], qz - xyz[3 * (size_t)k + 2]};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float r = rintf(d[a] * 100000.f) / 100000.f;          // torch.round(rel * 100000) / 100000 (half to even, IEEE division)
                const float t = (r + Q.c2w) - 1e-4f;                        // rel + 2 * window_size - 1e-4, one rounding per operation
                const int v = (int)truncf(t / Q.qs);                        // torch.div(., quant_size, rounding_mode="trunc")
                bad |= v < 0 || v > Q.vmax;
                rel[3 * e + a] = v;
            }
        }
    };
    size_t base = (size_t)offsets[q];
    const int nf = s.y - s.x;
    for (int j = lane; j < nf; j += 64) emit(base + j, order_f[s.x + j]);
    base += nf;
    for (int t0 = s.z; t0 < s.w; t0 += 64) {
        const int t = t0 + lane;
        const bool ok = t < s.w && wkd[min(t, s.w - 1)] != w;
        const unsigned long long mask = __ballot(ok);
        if (ok) emit(base + __popcll(mask & ((1ull << lane) - 1ull)), order_cd[t]);
        base += __popcll(mask);
    }
    if (bad) atomicOr(flag, 1);
}

// ---- window keys per point (the elementwise part of :468-499: torch_geometric's voxel_grid on the plain / half-window-shifted coordinates)
// Floor division as torch.div(a, b, rounding_mode="floor") defines it for floats: the quotient of (a - fmod(a, b)) by b, moved down by one
// when the remainder and the divisor differ in sign, rounded to the nearest integer below -- NOT floorf(a / b) (they differ when a / b
// rounds up to an integer).  IEEE division (hipcc's default), no contraction (-ffp-contract=off).
__device__ __forceinline__ float floor_div(float a, float b) {
    if (b == 0.f) return a / b;
    const float mod = fmodf(a, b);
    float div = (a - mod) / b;
    if (mod != 0.f && ((b < 0.f) != (mod < 0.f))) div -= 1.f;
    if (div == 0.f) return copysignf(0.f, a / b);
    float fl = floorf(div);
    if (div - fl > 0.5f) fl += 1.f;
    return fl;
}

struct KeyArgs { float ws; int parity, scenes; };

// one thread per point.  lo / hi (3): per-axis minimum / maximum of xyz over the whole batch; ends (scenes): scene ends.
// kf / kc: cluster id of the fine (ws) / coarse (2 ws) grid = sum_d cell_d * prod_{e < d} num_e with the scene index as 4th coordinate of
// cell size 1; even parity: grid origin = the minimum, odd parity: coordinates shifted by half a cell, origin = the minimum of the
// UNSHIFTED coordinates.  wk: the fine-window cell of :91-94 (trunc((xyz - min + shift) / ws) per axis) packed 21 bits per axis.
__global__ void k_keys(int n, const float *__restrict__ xyz, const int *__restrict__ ends, const float *__restrict__ lo,
                       const float *__restrict__ hi, KeyArgs A, i64 *__restrict__ kf, i64 *__restrict__ kc, i64 *__restrict__ wk) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int b = 0;
    while (b < A.scenes - 1 && i >= ends[b]) ++b;
    i64 key[2], wc[3];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const float size = g == 0 ? A.ws : 2.f * A.ws;
        const float shift = A.parity ? 0.5f * size : 0.f;
        i64 k = 0, stride = 1;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float p = A.parity ? xyz[3 * (size_t)i + d] + shift : xyz[3 * (size_t)i + d];
            const float e = A.parity ? hi[d] + shift : hi[d];
            const i64 cell = (i64)floor_div(p - lo[d], size);
            i64 num = (i64)floor_div(e - lo[d], size) + 1;
            if (num < 1) num = 1;
            k += cell * stride;
            stride *= num;
        }
        key[g] = k + stride * (i64)b;
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float r = xyz[3 * (size_t)i + d] - lo[d];
        const float t = A.parity ? r + 0.5f * A.ws : r + 0.f;
        wc[d] = (i64)truncf(t / A.ws);
    }
    kf[i] = key[0];
    kc[i] = key[1];
    wk[i] = (wc[0] << 42) | (wc[1] << 21) | wc[2];
}

}  // namespace we

// Window keys of one Swin block's partition (parity 0: plain, 1: shifted by half a window).  xyz (n, 3); ends (scenes) int32 scene ends;
// lo / hi (3) float32: per-axis minimum / maximum of xyz (device).  -> kf, kc, wk (n) int64 (see we::k_keys).
extern "C" int pdf_window_keys(int n, const float *xyz, const int *ends, int scenes, const float *lo, const float *hi, float window_size,
                               int parity, long long *kf, long long *kc, long long *wk, void *stream) {
    if (n < 0 || scenes < 0 || !(window_size > 0.f)) return PDF_ERR_BAD_ARG;
    if (n == 0) return PDF_OK;
    if (!xyz || !ends || scenes < 1 || !lo || !hi || !kf || !kc || !wk) return PDF_ERR_BAD_ARG;
    we::KeyArgs A{window_size, parity & 1, scenes};
    we::k_keys<<<pdf_divup(n, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(n, xyz, ends, lo, hi, A, kf, kc, wk);
    return pdf_launch_status();
}

// Row lengths of the CSR-by-query edge table.  kf / kc / wk (n) int64: fine-window key, coarse-window key and packed fine-window cell of
// every point; kf_sorted (n): kf in ascending order; kcd_sorted (m), wkd (m): kc / wk of the m downsampled points in ascending kc order
// (ties: ascending point id).  -> count (n) int32, seg (n, 4) int32 = [fine begin, fine end, coarse begin, coarse end] per query.
extern "C" int pdf_window_edges_count(int n, const long long *kf_sorted, const long long *kf, int m, const long long *kcd_sorted,
                                      const long long *kc, const long long *wk, const long long *wkd, int *count, int *seg, void *stream) {
    if (n < 0 || m < 0) return PDF_ERR_BAD_ARG;
    if (n == 0) return PDF_OK;
    if (!kf_sorted || !kf || !kc || !wk || !count || !seg || (m > 0 && (!kcd_sorted || !wkd))) return PDF_ERR_BAD_ARG;
    if (reinterpret_cast<uintptr_t>(seg) & 15) return PDF_ERR_BAD_ARG;
    we::k_count<<<pdf_divup(n, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(n, kf_sorted, kf, m, kcd_sorted, kc, wk, wkd, count,
                                                                               reinterpret_cast<int4 *>(seg));
    return pdf_launch_status();
}

// The rows.  offsets (n + 1) int32: exclusive scan of count; order_f (n) int32: point ids in ascending (kf, id) order; order_cd (m) int32:
// downsampled point ids in ascending (kc, id) order; xyz (n, 3).  -> index0 (E) int64 (the query of every edge: ascending), index1 (E)
// int32, rel (E, 3) int32 or null = trunc((round(1e5 (xyz[q] - xyz[k])) / 1e5 + c2w - 1e-4) / qs) per axis, *flag |= 1 when a value
// leaves [0, vmax] (flag: one int32 the caller zeroed).
extern "C" int pdf_window_edges_fill(int n, const int *offsets, const int *seg, const int *order_f, const int *order_cd, const long long *wk,
                                     const long long *wkd, const float *xyz, float c2w, float qs, int vmax, long long *index0, int *index1,
                                     int *rel, int *flag, void *stream) {
    if (n < 0) return PDF_ERR_BAD_ARG;
    if (n == 0) return PDF_OK;
    if (!offsets || !seg || !order_f || !wk || !xyz || !index0 || !index1 || (rel && !flag)) return PDF_ERR_BAD_ARG;
    if (reinterpret_cast<uintptr_t>(seg) & 15) return PDF_ERR_BAD_ARG;
    we::Quant Q{c2w, qs, vmax};
    we::k_fill<<<pdf_divup(n, 4), 256, 0, static_cast<hipStream_t>(stream)>>>(n, offsets, reinterpret_cast<const int4 *>(seg), order_f, order_cd, wk,
                                                                            wkd, xyz, Q, index0, index1, rel, flag);
    return pdf_launch_status();
}
