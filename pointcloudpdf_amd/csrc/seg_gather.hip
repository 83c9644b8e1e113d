// Scatter-adds as segmented gathers over an INVERSE neighbour table.
//
// Every backward of the gather family (grouping / subtraction / interpolation / aggregation; the fused PointTransformerLayer's
// g_xk and g_xv) is  out[v, :] += f(e) x src[row(e), :]  for all entries e = (i, j) of a neighbour table idx (m, nsample) with
// idx[e] == v.  The reference issues one atomicAdd per (entry, channel) (grouping_cuda_kernel.cu:20-25, interpolation_cuda_kernel.cu:
// 27-33, subtraction_cuda_kernel.cu:24-30, aggregation_cuda_kernel.cu:30-39); on gfx950 the memory-side atomic units bound those
// kernels at ~0.31 T atomics/s (18-22 % of the HBM roofline, round 1).  The geometry pre-pass knows the table before any feature
// exists, so it sorts the entry ids by destination once (stable: ascending entry id inside a destination, hence a FIXED summation
// order and bit-reproducible gradients):
//
//     inv_off   (n + 1) int32   positions into inv_entry, ascending; destination v owns [inv_off[v], inv_off[v + 1])
//     inv_entry (..)    int32   entry ids e + entry_base, grouped by destination (entries with idx < 0 are outside every segment)
//
// (entry_base: a batch cut out of a grouped pre-pass keeps the group's arrays and subtracts its first entry id -- Geometry.split).
// The kernels below read each source row exactly once per entry in whole 16-byte pieces (a row = contiguous c floats: full lines),
// keep the running sum in registers and store every destination row once: HBM traffic = the algorithmic bytes, no atomics.
//
// Mapping: lane = (destination row, 16-byte piece of the row); the c/4 lanes of a row walk the row's segment together (the
// entry id is one broadcast load), four entries in flight per trip.  Bound: HBM (gathers of whole rows).
#include "pdfops_common.h"

namespace sg {

constexpr int TB = 256;
constexpr int SB = 8;   // entries of a segment in flight per trip

struct FastDiv { unsigned d, m, s; };   // x / d for x < 2^31 by multiply-high (d >= 1)
static inline FastDiv mk_fastdiv(unsigned d) {
    FastDiv f; f.d = d;
    if (d == 1) { f.m = 0; f.s = 0; return f; }
    unsigned s = 0;
    while ((1u << s) < d) ++s;
    f.s = s;
    f.m = (unsigned)(((1ull << (32 + s)) + d - 1) / d - (1ull << 32));
    return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned x, FastDiv f) {
    if (f.d == 1) return x;
    const unsigned t = __umulhi(x, f.m);
    return (t + ((x - t) >> 1)) >> (f.s - 1);
}

template <int V> struct Vec;
template <> struct Vec<4> { using T = float4; };
template <> struct Vec<1> { using T = float; };
__device__ __forceinline__ float4 vld(const float4 *p) { return *p; }
__device__ __forceinline__ float bf2f(unsigned h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ float4 vld_bf(const unsigned short *p) {   // four consecutive bfloat16 -> fp32
    const uint2 v = *reinterpret_cast<const uint2 *>(p);
    return make_float4(bf2f(v.x & 0xffffu), bf2f(v.x >> 16), bf2f(v.y & 0xffffu), bf2f(v.y >> 16));
}
__device__ __forceinline__ float vld(const float *p) { return *p; }
__device__ __forceinline__ void fma_acc(float4 &a, float4 x, float4 w) { a.x += x.x * w.x; a.y += x.y * w.y; a.z += x.z * w.z; a.w += x.w * w.w; }
__device__ __forceinline__ void fma_acc(float &a, float x, float w) { a += x * w; }
__device__ __forceinline__ void add_acc(float4 &a, float4 x) { a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w; }
__device__ __forceinline__ void add_acc(float &a, float x) { a += x; }
__device__ __forceinline__ float4 scaled(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float scaled(float a, float s) { return a * s; }
template <int V> __device__ __forceinline__ typename Vec<V>::T zero();
template <> __device__ __forceinline__ float4 zero<4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <> __device__ __forceinline__ float zero<1>() { return 0.f; }

// out[v, :] = scale * sum over the segment of src[e, :]          (src rows are indexed by the entry id itself)
// (sv = source row stride in units of T: rows wider than the c channels that are summed, e.g. the (3 + c)-float rows of grouping(with_xyz))
template <int V>
__global__ __launch_bounds__(TB) void k_seg_rows(unsigned total, FastDiv cvd, unsigned sv, const typename Vec<V>::T *__restrict__ src,
                                                 const int *__restrict__ inv_off, const int *__restrict__ inv_entry, int entry_base,
                                                 float scale, typename Vec<V>::T *__restrict__ out) {
    using T = typename Vec<V>::T;
    const unsigned gid = blockIdx.x * TB + threadIdx.x;
    if (gid >= total) return;
    const unsigned v = fdiv(gid, cvd), p = gid - v * cvd.d, cv = cvd.d;
    int t = inv_off[v];
    const int end = inv_off[v + 1];
    T a0 = zero<V>(), a1 = zero<V>();
    // SB entries per trip, all entry ids in flight together, then all rows (a segment of <= SB entries costs two memory round trips;
    // the tail is masked, not a serial loop: its ids are clamped to the segment's last entry -- a line the trip reads anyway)
    for (; t < end; t += SB) {
        int e[SB];
        T x[SB];
#pragma unroll
        for (int k = 0; k < SB; ++k) e[k] = inv_entry[min(t + k, end - 1)] - entry_base;
#pragma unroll
        for (int k = 0; k < SB; ++k) x[k] = vld(src + (size_t)e[k] * sv + p);
#pragma unroll
        for (int k = 0; k < SB; ++k) add_acc((k & 1) ? a1 : a0, t + k < end ? x[k] : zero<V>());
    }
    add_acc(a0, a1);
    out[(size_t)v * cv + p] = scaled(a0, scale);
}

// Both sums of a SELF table's backward in one walk (round 4; subtraction: d input1 = own-row sums, d input2 = - inverse-segment sums):
//     out_own[v, :] = sum_j src[v * nsample + j, :]        out[v, :] = scale * sum over v's segment of src[e, :]
// The two passes of rounds 2-3 each ran at the copy rate but read the (m * nsample, c) gradient twice.  With the destinations visited in
// Morton order (`order`) on XCD-chunked blocks, the rows a destination gathers are the own rows of points a few positions away in the
// same order -- read by a neighbouring workgroup of the same XCD at about the same time: the second read is an L2 hit.
__global__ __launch_bounds__(TB) void k_seg_rows_own(unsigned total, FastDiv cvd, int nsample, const float4 *__restrict__ src,
                                                     const int *__restrict__ inv_off, const int *__restrict__ inv_entry, int entry_base,
                                                     float scale, const int *__restrict__ order, float4 *__restrict__ out, float4 *__restrict__ out_own) {
    const unsigned gid = pdf_xcd_chunked_block(blockIdx.x, gridDim.x) * TB + threadIdx.x;
    if (gid >= total) return;
    const unsigned sv = fdiv(gid, cvd), p = gid - sv * cvd.d, cv = cvd.d, v = order ? (unsigned)order[sv] : sv;
    int t = inv_off[v];
    const int end = inv_off[v + 1];
    float4 o0 = zero<4>(), o1 = zero<4>();
    const float4 *own = src + (size_t)v * nsample * cv + p;
    for (int j = 0; j < nsample; j += SB) {   // own rows: contiguous, SB in flight
        float4 x[SB];
#pragma unroll
        for (int k = 0; k < SB; ++k) x[k] = vld(own + (size_t)min(j + k, nsample - 1) * cv);
#pragma unroll
        for (int k = 0; k < SB; ++k) add_acc((k & 1) ? o1 : o0, j + k < nsample ? x[k] : zero<4>());
    }
    add_acc(o0, o1);
    out_own[(size_t)v * cv + p] = o0;
    float4 a0 = zero<4>(), a1 = zero<4>();
    for (; t < end; t += SB) {
        int e[SB];
        float4 x[SB];
#pragma unroll
        for (int k = 0; k < SB; ++k) e[k] = inv_entry[min(t + k, end - 1)] - entry_base;
#pragma unroll
        for (int k = 0; k < SB; ++k) x[k] = vld(src + (size_t)e[k] * cv + p);
#pragma unroll
        for (int k = 0; k < SB; ++k) add_acc((k & 1) ? a1 : a0, t + k < end ? x[k] : zero<4>());
    }
    add_acc(a0, a1);
    out[(size_t)v * cv + p] = scaled(a0, scale);
}

// out[v, ch] = sum over the segment of src[e / nsample, ch] * w[e, ch mod w_c]
//   aggregation grad_input (w = attention weights), fused layer g_xv (src = g_out, w = softmax weights), interpolation (w_c = 1)
// `order` (nullable): visiting order of the DESTINATION rows (their Morton order): neighbouring destinations are gathered by the same
// source rows, which then hit the L2 of the XCD that owns this stretch of the order (pdf_xcd_chunked_block).
template <int V, bool WVEC>
__global__ __launch_bounds__(TB) void k_seg_weighted(unsigned total, FastDiv cvd, FastDiv nsd, int w_c, const typename Vec<V>::T *__restrict__ src,
                                                     const float *__restrict__ w, const int *__restrict__ inv_off,
                                                     const int *__restrict__ inv_entry, int entry_base, const int *__restrict__ order,
                                                     typename Vec<V>::T *__restrict__ out) {
    using T = typename Vec<V>::T;
    const unsigned gid = pdf_xcd_chunked_block(blockIdx.x, gridDim.x) * TB + threadIdx.x;
    if (gid >= total) return;
    const unsigned sv = fdiv(gid, cvd), p = gid - sv * cvd.d, cv = cvd.d, v = order ? (unsigned)order[sv] : sv;
    const unsigned wo = (p * V) % (unsigned)w_c;   // first weight column of this piece (V consecutive channels; WVEC: w_c % 4 == 0)
    int t = inv_off[v];
    const int end = inv_off[v + 1];
    T a0 = zero<V>(), a1 = zero<V>();
    auto wload = [&](unsigned e) -> T {
        if constexpr (V == 4) {
            if constexpr (WVEC) return *reinterpret_cast<const float4 *>(w + (size_t)e * w_c + wo);
            else {
                const float *r = w + (size_t)e * w_c;
                const unsigned c0 = p * 4;
                return make_float4(r[c0 % w_c], r[(c0 + 1) % w_c], r[(c0 + 2) % w_c], r[(c0 + 3) % w_c]);
            }
        } else {
            return w[(size_t)e * w_c + wo];
        }
    };
    for (; t < end; t += SB) {   // (as k_seg_rows: ids, then rows + weights, masked tail)
        unsigned e[SB];
        T x[SB], ww[SB];
#pragma unroll
        for (int k = 0; k < SB; ++k) e[k] = (unsigned)(inv_entry[min(t + k, end - 1)] - entry_base);
#pragma unroll
        for (int k = 0; k < SB; ++k) { x[k] = vld(src + (size_t)fdiv(e[k], nsd) * cv + p); ww[k] = wload(e[k]); }
#pragma unroll
        for (int k = 0; k < SB; ++k) fma_acc((k & 1) ? a1 : a0, x[k], t + k < end ? ww[k] : zero<V>());
    }
    add_acc(a0, a1);
    out[(size_t)v * cv + p] = a0;
}

// The fused PointTransformerLayer's reduced-precision variant keeps its g_r rows / softmax weights as bfloat16 (fp32 sums here).
__global__ __launch_bounds__(TB) void k_seg_rows_bf(unsigned total, FastDiv cvd, unsigned src_stride, const unsigned short *__restrict__ src,
                                                    const int *__restrict__ inv_off, const int *__restrict__ inv_entry, int entry_base,
                                                    float scale, float4 *__restrict__ out) {
    const unsigned gid = blockIdx.x * TB + threadIdx.x;
    if (gid >= total) return;
    const unsigned v = fdiv(gid, cvd), p = gid - v * cvd.d, cv = cvd.d;
    int t = inv_off[v];
    const int end = inv_off[v + 1];
    float4 a0 = zero<4>(), a1 = zero<4>();
    for (; t + 4 <= end; t += 4) {
        const int e0 = inv_entry[t] - entry_base, e1 = inv_entry[t + 1] - entry_base, e2 = inv_entry[t + 2] - entry_base, e3 = inv_entry[t + 3] - entry_base;
        const float4 x0 = vld_bf(src + (size_t)e0 * src_stride + 4 * p), x1 = vld_bf(src + (size_t)e1 * src_stride + 4 * p);
        const float4 x2 = vld_bf(src + (size_t)e2 * src_stride + 4 * p), x3 = vld_bf(src + (size_t)e3 * src_stride + 4 * p);
        add_acc(a0, x0); add_acc(a1, x1); add_acc(a0, x2); add_acc(a1, x3);
    }
    for (; t < end; ++t) add_acc(a0, vld_bf(src + (size_t)(inv_entry[t] - entry_base) * src_stride + 4 * p));
    add_acc(a0, a1);
    out[(size_t)v * cv + p] = scaled(a0, scale);
}

__global__ __launch_bounds__(TB) void k_seg_weighted_bfw(unsigned total, FastDiv cvd, FastDiv nsd, int w_c, const float4 *__restrict__ src,
                                                         const unsigned short *__restrict__ w, const int *__restrict__ inv_off,
                                                         const int *__restrict__ inv_entry, int entry_base, float4 *__restrict__ out) {
    const unsigned gid = blockIdx.x * TB + threadIdx.x;
    if (gid >= total) return;
    const unsigned v = fdiv(gid, cvd), p = gid - v * cvd.d, cv = cvd.d;
    const unsigned wo = (p * 4) % (unsigned)w_c;
    int t = inv_off[v];
    const int end = inv_off[v + 1];
    float4 a0 = zero<4>(), a1 = zero<4>();
    for (; t + 2 <= end; t += 2) {
        const unsigned e0 = (unsigned)(inv_entry[t] - entry_base), e1 = (unsigned)(inv_entry[t + 1] - entry_base);
        const float4 x0 = vld(src + (size_t)fdiv(e0, nsd) * cv + p), x1 = vld(src + (size_t)fdiv(e1, nsd) * cv + p);
        fma_acc(a0, x0, vld_bf(w + (size_t)e0 * w_c + wo)); fma_acc(a1, x1, vld_bf(w + (size_t)e1 * w_c + wo));
    }
    if (t < end) {
        const unsigned e0 = (unsigned)(inv_entry[t] - entry_base);
        fma_acc(a0, vld(src + (size_t)fdiv(e0, nsd) * cv + p), vld_bf(w + (size_t)e0 * w_c + wo));
    }
    add_acc(a0, a1);
    out[(size_t)v * cv + p] = a0;
}

}  // namespace sg

// out (n, c) = scale * segmented sum of the rows src[e * src_stride + 0 .. c) (src_stride >= c floats between consecutive entries;
// src may point at a column offset inside wider rows) -- grouping / subtraction backward, grouping(with_xyz) backward, g_xk.
extern "C" int pdf_seg_sum_rows_strided(long n, int c, const float *src, long src_stride, const int *inv_off, const int *inv_entry,
                                        int entry_base, float scale, float *out, void *stream) {
    if (n == 0) return PDF_OK;
    if (n < 0 || c < 1 || src_stride < c || !src || !inv_off || !inv_entry || !out) return PDF_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool v4 = c % 4 == 0 && src_stride % 4 == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)out % 16 == 0);
    const int cv = v4 ? c / 4 : c;
    const long total = n * cv;
    if (total >= (1L << 31) || src_stride >= (1L << 31)) return PDF_ERR_UNSUPPORTED;
    const unsigned g = (unsigned)((total + sg::TB - 1) / sg::TB);
    if (v4)
        sg::k_seg_rows<4><<<g, sg::TB, 0, s>>>((unsigned)total, sg::mk_fastdiv(cv), (unsigned)(src_stride / 4), reinterpret_cast<const float4 *>(src),
                                               inv_off, inv_entry, entry_base, scale, reinterpret_cast<float4 *>(out));
    else
        sg::k_seg_rows<1><<<g, sg::TB, 0, s>>>((unsigned)total, sg::mk_fastdiv(cv), (unsigned)src_stride, src, inv_off, inv_entry, entry_base, scale, out);
    return pdf_launch_status();
}

extern "C" int pdf_seg_sum_rows(long n, int c, const float *src, const int *inv_off, const int *inv_entry, int entry_base, float scale,
                                float *out, void *stream) {
    return pdf_seg_sum_rows_strided(n, c, src, c, inv_off, inv_entry, entry_base, scale, out, stream);
}

// SELF tables (n queries = n destinations): out (n, c) = scale * inverse-segment sums of src (n * nsample, c), out_own (n, c) = sums of every
// point's own nsample rows, in ONE walk; `order` (nullable): visiting order of the points.  c % 4 == 0, 16-byte aligned pointers.
extern "C" int pdf_seg_sum_rows_own(long n, int c, int nsample, const float *src, const int *inv_off, const int *inv_entry, int entry_base,
                                    float scale, const int *order, float *out, float *out_own, void *stream) {
    if (n == 0) return PDF_OK;
    if (n < 0 || c < 1 || nsample < 1 || !src || !inv_off || !inv_entry || !out || !out_own) return PDF_ERR_BAD_ARG;
    if (c % 4 != 0 || ((uintptr_t)src % 16) || ((uintptr_t)out % 16) || ((uintptr_t)out_own % 16) || n * (c / 4) >= (1L << 31)) return PDF_ERR_UNSUPPORTED;
    const long total = n * (c / 4);
    sg::k_seg_rows_own<<<(unsigned)((total + sg::TB - 1) / sg::TB), sg::TB, 0, static_cast<hipStream_t>(stream)>>>(
        (unsigned)total, sg::mk_fastdiv(c / 4), nsample, reinterpret_cast<const float4 *>(src), inv_off, inv_entry, entry_base, scale, order,
        reinterpret_cast<float4 *>(out), reinterpret_cast<float4 *>(out_own));
    return pdf_launch_status();
}

// out (n, c)[v, ch] = segmented sum of src[e / nsample, ch] * w[e, ch mod w_c]; src (m, c), w (m * nsample, w_c).
extern "C" int pdf_seg_sum_weighted_ordered(long n, int c, int nsample, int w_c, const float *src, const float *w, const int *inv_off,
                                            const int *inv_entry, int entry_base, const int *order, float *out, void *stream) {
    if (n == 0) return PDF_OK;
    if (n < 0 || c < 1 || nsample < 1 || w_c < 1 || c % w_c != 0 || !src || !w || !inv_off || !inv_entry || !out) return PDF_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool v4 = c % 4 == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)out % 16 == 0);
    const int cv = v4 ? c / 4 : c;
    const long total = n * cv;
    if (total >= (1L << 31)) return PDF_ERR_UNSUPPORTED;
    const unsigned g = (unsigned)((total + sg::TB - 1) / sg::TB);
    const sg::FastDiv cvd = sg::mk_fastdiv(cv), nsd = sg::mk_fastdiv(nsample);
    if (v4 && w_c % 4 == 0 && ((uintptr_t)w % 16 == 0))
        sg::k_seg_weighted<4, true><<<g, sg::TB, 0, s>>>((unsigned)total, cvd, nsd, w_c, reinterpret_cast<const float4 *>(src), w, inv_off,
                                                         inv_entry, entry_base, order, reinterpret_cast<float4 *>(out));
    else if (v4)
        sg::k_seg_weighted<4, false><<<g, sg::TB, 0, s>>>((unsigned)total, cvd, nsd, w_c, reinterpret_cast<const float4 *>(src), w, inv_off,
                                                          inv_entry, entry_base, order, reinterpret_cast<float4 *>(out));
    else
        sg::k_seg_weighted<1, false><<<g, sg::TB, 0, s>>>((unsigned)total, cvd, nsd, w_c, src, w, inv_off, inv_entry, entry_base, order, out);
    return pdf_launch_status();
}

extern "C" int pdf_seg_sum_weighted(long n, int c, int nsample, int w_c, const float *src, const float *w, const int *inv_off,
                                    const int *inv_entry, int entry_base, float *out, void *stream) {
    return pdf_seg_sum_weighted_ordered(n, c, nsample, w_c, src, w, inv_off, inv_entry, entry_base, nullptr, out, stream);
}

// Internal twins used by the fused layer: source rows / weights either fp32 or bfloat16 (`*_bf16` flag; needs c % 4 == 0, w_c % 4 == 0).
extern "C" int pdf_seg_sum_rows_x(long n, int c, const float *src, long src_stride, int src_bf16, const int *inv_off, const int *inv_entry,
                                  int entry_base, float scale, float *out, void *stream) {
    if (!src_bf16) return pdf_seg_sum_rows_strided(n, c, src, src_stride, inv_off, inv_entry, entry_base, scale, out, stream);
    if (n == 0) return PDF_OK;
    if (n < 0 || c < 4 || c % 4 || src_stride % 4 || !src || !inv_off || !inv_entry || !out) return PDF_ERR_BAD_ARG;
    const long total = n * (c / 4);
    if (total >= (1L << 31)) return PDF_ERR_UNSUPPORTED;
    sg::k_seg_rows_bf<<<(unsigned)((total + sg::TB - 1) / sg::TB), sg::TB, 0, static_cast<hipStream_t>(stream)>>>(
        (unsigned)total, sg::mk_fastdiv(c / 4), (unsigned)src_stride, reinterpret_cast<const unsigned short *>(src), inv_off, inv_entry, entry_base,
        scale, reinterpret_cast<float4 *>(out));
    return pdf_launch_status();
}

extern "C" int pdf_seg_sum_weighted_x(long n, int c, int nsample, int w_c, const float *src, const float *w, int w_bf16, const int *inv_off,
                                      const int *inv_entry, int entry_base, const int *order, float *out, void *stream) {
    if (!w_bf16) return pdf_seg_sum_weighted_ordered(n, c, nsample, w_c, src, w, inv_off, inv_entry, entry_base, order, out, stream);
    if (n == 0) return PDF_OK;
    if (n < 0 || c % 4 || w_c % 4 || c % w_c || nsample < 1 || !src || !w || !inv_off || !inv_entry || !out) return PDF_ERR_BAD_ARG;
    const long total = n * (c / 4);
    if (total >= (1L << 31)) return PDF_ERR_UNSUPPORTED;
    sg::k_seg_weighted_bfw<<<(unsigned)((total + sg::TB - 1) / sg::TB), sg::TB, 0, static_cast<hipStream_t>(stream)>>>(
        (unsigned)total, sg::mk_fastdiv(c / 4), sg::mk_fastdiv(nsample), w_c, reinterpret_cast<const float4 *>(src),
        reinterpret_cast<const unsigned short *>(w), inv_off, inv_entry, entry_base, reinterpret_cast<float4 *>(out));
    return pdf_launch_status();
}
