// Gather family for gfx950: grouping2 / interpolation2 / subtraction / aggregation (fwd + bwd),
// the fused twin of pointops.grouping() and the inverse-distance weights of pointops.interpolation().
// Replaces libs/pointops/src/{grouping,interpolation,subtraction,aggregation}/*_cuda_kernel.cu.
//
// All of these are HBM-bound row gathers/scatters: one lane moves 16 B (float4) of a channel row, a
// neighbour's row is read by consecutive lanes (coalesced 64..2048 B segments), launches are
// grid-stride over <= 2048 workgroups.  Backward scatters use the hardware fp32 atomic
// (global_atomic_add_f32) like the reference's atomicAdd; summation order is therefore unordered,
// exactly as upstream (tolerance-based parity for gradients).
// Algorithmic bytes per call: SURVEY.md 8(d).
#include "pdfops_common.h"

namespace {

constexpr int GB = 256;  // workgroup size

template <int V> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<2> { using type = float2; };
template <> struct VecT<4> { using type = float4; };

template <int V> __device__ __forceinline__ typename VecT<V>::type vzero();
template <> __device__ __forceinline__ float vzero<1>() { return 0.f; }
template <> __device__ __forceinline__ float2 vzero<2>() { return make_float2(0.f, 0.f); }
template <> __device__ __forceinline__ float4 vzero<4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }

__device__ __forceinline__ float vget(const float &v, int) { return v; }
__device__ __forceinline__ float vget(const float2 &v, int i) { return i == 0 ? v.x : v.y; }
__device__ __forceinline__ float vget(const float4 &v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
__device__ __forceinline__ void vset(float &v, int, float x) { v = x; }
__device__ __forceinline__ void vset(float2 &v, int i, float x) { if (i == 0) v.x = x; else v.y = x; }
__device__ __forceinline__ void vset(float4 &v, int i, float x) {
    if (i == 0) v.x = x; else if (i == 1) v.y = x; else if (i == 2) v.z = x; else v.w = x;
}

static inline int pick_vec(int c) { return (c % 4 == 0) ? 4 : (c % 2 == 0) ? 2 : 1; }
static inline int grid_for(long total) {
    long g = (total + GB - 1) / GB;
    if (g > PDF_MAX_BLOCKS) g = PDF_MAX_BLOCKS;
    if (g < 1) g = 1;
    return (int)g;
}

// ---------------------------------------------------------------- grouping2
// grouping_cuda_kernel.cu:5-14
template <int V>
__global__ __launch_bounds__(GB) void grouping_fwd_kernel(long rows, int cv, const float *__restrict__ input,
                                                          const int *__restrict__ idx, float *__restrict__ output) {
    using T = typename VecT<V>::type;
    const long total = rows * cv;
    for (long e = (long)blockIdx.x * GB + threadIdx.x; e < total; e += (long)gridDim.x * GB) {
        const long r = e / cv;
        const int col = (int)(e - r * cv);
        const int j = idx[r];
        T v = vzero<V>();
        if (j >= 0) v = reinterpret_cast<const T *>(input)[(long)j * cv + col];
        reinterpret_cast<T *>(output)[e] = v;
    }
}

// grouping_cuda_kernel.cu:16-25
template <int V>
__global__ __launch_bounds__(GB) void grouping_bwd_kernel(long rows, int cv, const float *__restrict__ grad_output,
                                                          const int *__restrict__ idx, float *__restrict__ grad_input) {
    using T = typename VecT<V>::type;
    const long total = rows * cv;
    for (long e = (long)blockIdx.x * GB + threadIdx.x; e < total; e += (long)gridDim.x * GB) {
        const long r = e / cv;
        const int col = (int)(e - r * cv);
        const int j = idx[r];
        if (j < 0) continue;
        const T g = reinterpret_cast<const T *>(grad_output)[e];
        float *dst = grad_input + ((long)j * cv + col) * V;
#pragma unroll
        for (int u = 0; u < V; ++u) pdf_atomic_add(dst + u, vget(g, u));
    }
}

// ---------------------------------------------------------------- pointops.grouping() twin
// libs/pointops/functions/grouping.py:36-60: out[m,s,:] = [ mask*(xyz[idx]-new_xyz[m]) | feat[idx] ], idx<0 -> 0
__global__ __launch_bounds__(GB) void group_fwd_kernel(long rows, int nsample, int c, int with_xyz,
                                                       const float *__restrict__ feat, const float *__restrict__ xyz,
                                                       const float *__restrict__ new_xyz, const int *__restrict__ idx,
                                                       float *__restrict__ output) {
    const int oc = c + (with_xyz ? 3 : 0);
    const long total = rows * oc;
    for (long e = (long)blockIdx.x * GB + threadIdx.x; e < total; e += (long)gridDim.x * GB) {
        const long r = e / oc;
        int col = (int)(e - r * oc);
        const int j = idx[r];
        float v = 0.f;
        if (with_xyz) {
            if (col < 3) {
                if (j >= 0) v = xyz[(long)j * 3 + col] - new_xyz[(r / nsample) * 3 + col];
                output[e] = v;
                continue;
            }
            col -= 3;
        }
        if (j >= 0) v = feat[(long)j * c + col];
        output[e] = v;
    }
}

__global__ __launch_bounds__(GB) void group_bwd_kernel(long rows, int c, int with_xyz,
                                                       const float *__restrict__ grad_output,
                                                       const int *__restrict__ idx, float *__restrict__ grad_feat) {
    const int oc = c + (with_xyz ? 3 : 0);
    const int sh = with_xyz ? 3 : 0;
    const long total = rows * c;
    for (long e = (long)blockIdx.x * GB + threadIdx.x; e < total; e += (long)gridDim.x * GB) {
        const long r = e / c;
        const int col = (int)(e - r * c);
        const int j = idx[r];
        if (j < 0) continue;
        pdf_atomic_add(grad_feat + (long)j * c + col, grad_output[r * oc + sh + col]);
    }
}

// ---------------------------------------------------------------- interpolation2
// interpolation_cuda_kernel.cu:5-18 (sum over k done in registers; output fully overwritten)
template <int V>
__global__ __launch_bounds__(GB) void interp_fwd_kernel(long n, int cv, int k, const float *__restrict__ input,
                                                        const int *__restrict__ idx, const float *__restrict__ weight,
                                                        float *__restrict__ output) {
    using T = typename VecT<V>::type;
    const long total = n * cv;
    for (long e = (long)blockIdx.x * GB + threadIdx.x; e < total; e += (long)gridDim.x * GB) {
        const long r = e / cv;
        const int col = (int)(e - r * cv);
        T acc = vzero<V>();
        for (int i = 0; i < k; ++i) {
            const int j = idx[r * k + i];
            const float w = weight[r * k + i];
            if (j < 0) continue;
            const T v = reinterpret_cast<const T *>(input)[(long)j * cv + col];
#pragma unroll
            for (int u = 0; u < V; ++u) vset(acc, u, vget(acc, u) + vget(v, u) * w);
        }
        reinterpret_cast<T *>(output)[e] = acc;
    }
}

// interpolation_cuda_kernel.cu:20-33
template <int V>
__global__ __launch_bounds__(GB) void interp_bwd_kernel(long n, int cv, int k, const float *__restrict__ grad_output,
                                                        const int *__restrict__ idx, const float *__restrict__ weight,
                                                        float *__restrict__ grad_input) {
    using T = typename VecT<V>::type;
    const long total = n * cv;
    for (long e = (long)blockIdx.x * GB + threadIdx.x; e < total; e += (long)gridDim.x * GB) {
        const long r = e / cv;
        const int col = (int)(e - r * cv);
        const T g = reinterpret_cast<const T *>(grad_output)[e];
        for (int i = 0; i < k; ++i) {
            const int j = idx[r * k + i];
            if (j < 0) continue;
            const float w = weight[r * k + i];
            float *dst = grad_input + ((long)j * cv + col) * V;
#pragma unroll
            for (int u = 0; u < V; ++u) pdf_atomic_add(dst + u, vget(g, u) * w);
        }
    }
}

// libs/pointops/functions/interpolation.py:14-17
__global__ __launch_bounds__(GB) void interp_weights_kernel(long n, int k, const float *__restrict__ dist2,
                                                            float *__restrict__ weight) {
    for (long r = (long)blockIdx.x * GB + threadIdx.x; r < n; r += (long)gridDim.x * GB) {
        float norm = 0.f;
        for (int i = 0; i < k; ++i) norm += 1.0f / (sqrtf(dist2[r * k + i]) + 1e-8f);
        for (int i = 0; i < k; ++i) weight[r * k + i] = (1.0f / (sqrtf(dist2[r * k + i]) + 1e-8f)) / norm;
    }
}

// ---------------------------------------------------------------- subtraction
// subtraction_cuda_kernel.cu:5-16
template <int V>
__global__ __launch_bounds__(GB) void sub_fwd_kernel(long rows, int nsample, int cv, const float *__restrict__ input1,
                                                     const float *__restrict__ input2, const int *__restrict__ idx,
                                                     float *__restrict__ output) {
    using T = typename VecT<V>::type;
    const long total = rows * cv;
    for (long e = (long)blockIdx.x * GB + threadIdx.x; e < total; e += (long)gridDim.x * GB) {
        const long r = e / cv;
        const int col = (int)(e - r * cv);
        const int j = idx[r];
        const T a = reinterpret_cast<const T *>(input1)[(r / nsample) * cv + col];
        T b = vzero<V>();
        if (j >= 0) b = reinterpret_cast<const T *>(input2)[(long)j * cv + col];
        T o;
#pragma unroll
        for (int u = 0; u < V; ++u) vset(o, u, vget(a, u) - vget(b, u));
        reinterpret_cast<T *>(output)[e] = o;
    }
}

// subtraction_cuda_kernel.cu:18-30.  grad_input1 row sums are reduced over nsample in registers
// (one lane owns (n, col)), grad_input2 is an atomic scatter.
template <int V>
__global__ __launch_bounds__(GB) void sub_bwd_kernel(long n, int nsample, int cv, const int *__restrict__ idx,
                                                     const float *__restrict__ grad_output,
                                                     float *__restrict__ grad_input1, float *__restrict__ grad_input2) {
    using T = typename VecT<V>::type;
    const long total = n * cv;
    for (long e = (long)blockIdx.x * GB + threadIdx.x; e < total; e += (long)gridDim.x * GB) {
        const long r = e / cv;
        const int col = (int)(e - r * cv);
        T acc = vzero<V>();
        for (int s = 0; s < nsample; ++s) {
            const T g = reinterpret_cast<const T *>(grad_output)[(r * nsample + s) * cv + col];
            const int j = idx[r * nsample + s];
#pragma unroll
            for (int u = 0; u < V; ++u) vset(acc, u, vget(acc, u) + vget(g, u));
            if (j < 0 || !grad_input2) continue;   // (null: the caller forms grad_input2 by the segmented gather, seg_gather.hip)
            float *dst = grad_input2 + ((long)j * cv + col) * V;
#pragma unroll
            for (int u = 0; u < V; ++u) pdf_atomic_add(dst + u, -vget(g, u));
        }
        float *d1 = grad_input1 + e * V;  // pre-zeroed accumulate target (reference contract)
#pragma unroll
        for (int u = 0; u < V; ++u) d1[u] += vget(acc, u);
    }
}

// ---------------------------------------------------------------- aggregation
// aggregation_cuda_kernel.cu:5-20; one lane owns (n, c_idx), loops over nsample. Output overwritten.
__global__ __launch_bounds__(GB) void agg_fwd_kernel(long n, int nsample, int c, int w_c,
                                                     const float *__restrict__ input, const float *__restrict__ position,
                                                     const float *__restrict__ weight, const int *__restrict__ idx,
                                                     float *__restrict__ output) {
    const long total = n * c;
    for (long e = (long)blockIdx.x * GB + threadIdx.x; e < total; e += (long)gridDim.x * GB) {
        const long r = e / c;
        const int col = (int)(e - r * c);
        const int wcol = col % w_c;
        float acc = 0.f;
        for (int s = 0; s < nsample; ++s) {
            const int j = idx[r * nsample + s];
            const float in = j >= 0 ? input[(long)j * c + col] : 0.f;
            acc += (in + position[(r * nsample + s) * c + col]) * weight[(r * nsample + s) * w_c + wcol];
        }
        output[e] = acc;
    }
}

// aggregation_cuda_kernel.cu:22-39
__global__ __launch_bounds__(GB) void agg_bwd_kernel(long n, int nsample, int c, int w_c,
                                                     const float *__restrict__ input, const float *__restrict__ position,
                                                     const float *__restrict__ weight, const int *__restrict__ idx,
                                                     const float *__restrict__ grad_output, float *__restrict__ grad_input,
                                                     float *__restrict__ grad_position, float *__restrict__ grad_weight) {
    const long total = n * c;
    for (long e = (long)blockIdx.x * GB + threadIdx.x; e < total; e += (long)gridDim.x * GB) {
        const long r = e / c;
        const int col = (int)(e - r * c);
        const int wcol = col % w_c;
        const float go = grad_output[e];
        for (int s = 0; s < nsample; ++s) {
            const int j = idx[r * nsample + s];
            const long pi = (r * nsample + s) * c + col;
            const long wi = (r * nsample + s) * w_c + wcol;
            const float w = weight[wi];
            const float in = j >= 0 ? input[(long)j * c + col] : 0.f;
            if (j >= 0 && grad_input) pdf_atomic_add(grad_input + (long)j * c + col, go * w);
            grad_position[pi] = go * w;
            pdf_atomic_add(grad_weight + wi, go * (in + position[pi]));
        }
    }
}


// ================================================================ 32-bit fast paths (total elements < 2^31)
// The generic kernels above divide 64-bit element ids by a runtime row length per element (an emulated 64-bit division:
// ~100 instructions) and keep one 16-B access in flight per lane.  These twins use a 32-bit multiply-high division,
// issue U independent gathers before the first store, and store the streamed (m, nsample, c) output non-temporally so
// that it does not evict the gathered table from the XCD's L2.
struct FastDiv { unsigned d, m, s; };   // q = n / d for n < 2^31:  (mulhi(n, m) + n) >> s,  m = ceil(2^(32+s) / d) - 2^32
static inline FastDiv mk_fastdiv(unsigned d) {
    FastDiv f; f.d = d; unsigned s = 0;
    while ((1ull << s) < d) ++s;
    f.s = s;
    f.m = (unsigned)(((((unsigned __int128)1) << (32 + s)) + d - 1) / d - (((unsigned __int128)1) << 32));
    return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv f) { return (__umulhi(n, f.m) + n) >> f.s; }

typedef float v4f __attribute__((ext_vector_type(4)));
#ifndef PDF_GATHER_PLAIN_STORES   // measured: streamed stores 57.1 us vs plain 59.2 us (grouping2 fwd, 200k x 8 x 32)
__device__ __forceinline__ void st_stream(v4f *p, v4f v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void st_stream(float *p, float v) { __builtin_nontemporal_store(v, p); }
#else
__device__ __forceinline__ void st_stream(v4f *p, v4f v) { *p = v; }
__device__ __forceinline__ void st_stream(float *p, float v) { *p = v; }
#endif

// grouping_cuda_kernel.cu:5-14, float4 per lane
template <int U>
__global__ __launch_bounds__(GB) void grouping_fwd32(unsigned total, FastDiv cvd, const v4f *__restrict__ input,
                                                     const int *__restrict__ idx, v4f *__restrict__ output) {
    const unsigned stride = gridDim.x * GB, cv = cvd.d;
    unsigned e = blockIdx.x * GB + threadIdx.x;
    for (; e < total && total - e > (U - 1) * stride; e += U * stride) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned ee = e + u * stride, r = fdiv(ee, cvd), col = ee - r * cv;
            const int j = idx[r];
            v[u] = j >= 0 ? input[(unsigned long)j * cv + col] : (v4f)(0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) st_stream(output + e + u * stride, v[u]);
    }
    for (; e < total; e += stride) {
        const unsigned r = fdiv(e, cvd), col = e - r * cv;
        const int j = idx[r];
        st_stream(output + e, j >= 0 ? input[(unsigned long)j * cv + col] : (v4f)(0.f));
    }
}

// grouping forward with the QUERIES visited in a caller-supplied order (a spatially coherent one: the geometry pre-pass sorts every
// level's points by Morton cell) and the workgroups of one XCD walking ONE contiguous stretch of that order.  A gathered row is
// a whole 128-byte line whatever the storage order, so what matters is WHEN rows are touched: consecutive sorted queries share
// most of their neighbours, and with the dispatcher's round-robin placement (workgroup b -> XCD b mod 8) undone by the remap below
// those repeats hit the 4 MB L2 of the XCD instead of crossing the fabric to the Infinity Cache.  Each query's nsample x c output
// block is contiguous, so the permuted visiting order still stores whole lines.
__device__ __forceinline__ unsigned xcd_chunked_block(unsigned b, unsigned g) { return pdf_xcd_chunked_block(b, g); }   // pdfops_common.h

template <int U>
__global__ __launch_bounds__(GB) void grouping_fwd_ord(unsigned total, FastDiv cvd, FastDiv rowd /* nsample * cv */, int nsample,
                                                       const int *__restrict__ order, const v4f *__restrict__ input,
                                                       const int *__restrict__ idx, v4f *__restrict__ output) {
    const unsigned cv = cvd.d, blk = xcd_chunked_block(blockIdx.x, gridDim.x);
    const unsigned w0 = blk * (GB * U) + threadIdx.x;
    // three phases, each with its U loads in flight together (visiting order -> neighbour index -> row piece): loads use clamped
    // addresses and the masks are applied afterwards -- a load under a condition gets a branch and a wait of its own
    v4f v[U];
    unsigned long dst[U], row[U];
    unsigned col[U], sq[U], jn[U];
    int src[U];
    bool live[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const unsigned w = w0 + u * GB;
        live[u] = w < total;
        sq[u] = fdiv(live[u] ? w : 0u, rowd);
        const unsigned rem = (live[u] ? w : 0u) - sq[u] * rowd.d;
        jn[u] = fdiv(rem, cvd); col[u] = rem - jn[u] * cv;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) row[u] = (unsigned long)(order ? (unsigned)order[sq[u]] : sq[u]) * nsample + jn[u];
#pragma unroll
    for (int u = 0; u < U; ++u) { src[u] = idx[row[u]]; dst[u] = row[u] * cv + col[u]; }
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = input[(unsigned long)max(src[u], 0) * cv + col[u]];
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (live[u]) st_stream(output + dst[u], src[u] >= 0 ? v[u] : (v4f)(0.f));
}

// the same visiting scheme for the other forward gathers (order == nullptr: queries in storage order, still one contiguous
// stretch per XCD)
template <int U>
__global__ __launch_bounds__(GB) void sub_fwd_ord(unsigned total, FastDiv cvd, FastDiv rowd, int nsample, const int *__restrict__ order,
                                                  const v4f *__restrict__ input1, const v4f *__restrict__ input2,
                                                  const int *__restrict__ idx, v4f *__restrict__ output) {
    const unsigned cv = cvd.d, blk = xcd_chunked_block(blockIdx.x, gridDim.x);
    const unsigned w0 = blk * (GB * U) + threadIdx.x;
    v4f a[U], b[U];
    unsigned long dst[U], row[U];
    unsigned col[U], sq[U], jn[U], q[U];
    int src[U];
    bool live[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const unsigned w = w0 + u * GB;
        live[u] = w < total;
        sq[u] = fdiv(live[u] ? w : 0u, rowd);
        const unsigned rem = (live[u] ? w : 0u) - sq[u] * rowd.d;
        jn[u] = fdiv(rem, cvd); col[u] = rem - jn[u] * cv;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { q[u] = order ? (unsigned)order[sq[u]] : sq[u]; row[u] = (unsigned long)q[u] * nsample + jn[u]; }
#pragma unroll
    for (int u = 0; u < U; ++u) { src[u] = idx[row[u]]; dst[u] = row[u] * cv + col[u]; a[u] = input1[(unsigned long)q[u] * cv + col[u]]; }
#pragma unroll
    for (int u = 0; u < U; ++u) b[u] = input2[(unsigned long)max(src[u], 0) * cv + col[u]];
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (live[u]) st_stream(output + dst[u], a[u] - (src[u] >= 0 ? b[u] : (v4f)(0.f)));
}

// interpolation / aggregation: one lane = (query, 16-byte piece), the neighbours are a loop
template <int K>
__global__ __launch_bounds__(GB) void interp_fwd_ord(unsigned total, FastDiv cvd, const int *__restrict__ order, const v4f *__restrict__ input,
                                                     const int *__restrict__ idx, const float *__restrict__ weight, v4f *__restrict__ output) {
    const unsigned cv = cvd.d, blk = xcd_chunked_block(blockIdx.x, gridDim.x);
    const unsigned w = blk * GB + threadIdx.x;
    if (w >= total) return;
    const unsigned sq = fdiv(w, cvd), col = w - sq * cv, r = order ? (unsigned)order[sq] : sq;
    int j[K]; float wt[K]; v4f v[K];
#pragma unroll
    for (int i = 0; i < K; ++i) { j[i] = idx[(unsigned long)r * K + i]; wt[i] = weight[(unsigned long)r * K + i]; }
#pragma unroll
    for (int i = 0; i < K; ++i) v[i] = input[(unsigned long)max(j[i], 0) * cv + col];
    v4f acc = (v4f)(0.f);
#pragma unroll
    for (int i = 0; i < K; ++i) acc += (j[i] >= 0 ? v[i] : (v4f)(0.f)) * wt[i];   // same summation order as the reference (i ascending)
    st_stream(output + (unsigned long)r * cv + col, acc);
}

__global__ __launch_bounds__(GB) void agg_fwd_ord(unsigned total, FastDiv cvd, int nsample, unsigned wv, const int *__restrict__ order,
                                                  const v4f *__restrict__ input, const v4f *__restrict__ position,
                                                  const v4f *__restrict__ weight, const int *__restrict__ idx, v4f *__restrict__ output) {
    const unsigned cv = cvd.d, blk = xcd_chunked_block(blockIdx.x, gridDim.x);
    const unsigned w = blk * GB + threadIdx.x;
    if (w >= total) return;
    const unsigned sq = fdiv(w, cvd), col = w - sq * cv, wcol = col % wv, r = order ? (unsigned)order[sq] : sq;
    v4f acc = (v4f)(0.f);
    const unsigned long base = (unsigned long)r * nsample;
    int s = 0;
    for (; s + 4 <= nsample; s += 4) {   // four neighbours at a time: indices, then rows / positions / weights, then the sum (s ascending)
        int j[4]; v4f in[4], pos[4], wt[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) j[t] = idx[base + s + t];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            in[t] = input[(unsigned long)max(j[t], 0) * cv + col];
            pos[t] = __builtin_nontemporal_load(position + (base + s + t) * cv + col);
            wt[t] = weight[(base + s + t) * wv + wcol];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) acc += ((j[t] >= 0 ? in[t] : (v4f)(0.f)) + pos[t]) * wt[t];
    }
    for (; s < nsample; ++s) {
        const int j = idx[base + s];
        const v4f in = input[(unsigned long)max(j, 0) * cv + col];
        acc += ((j >= 0 ? in : (v4f)(0.f)) + __builtin_nontemporal_load(position + (base + s) * cv + col)) * weight[(base + s) * wv + wcol];
    }
    output[(unsigned long)r * cv + col] = acc;
}

// pointops.grouping(with_xyz) rows of 3 + c floats (see group_fwd_rows): lane = (row, piece), U items per lane
template <int U>
__global__ __launch_bounds__(GB) void group_fwd_rows_ord(unsigned total, FastDiv pd, FastDiv nsd, FastDiv qd /* nsample * pieces */, int c,
                                                         int with_xyz, const int *__restrict__ order, const float *__restrict__ feat,
                                                         const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                         const int *__restrict__ idx, float *__restrict__ output) {
    typedef float u4f __attribute__((ext_vector_type(4), aligned(4)));
    const unsigned pieces = pd.d, cv = c >> 2, oc = c + (with_xyz ? 3 : 0), sh = with_xyz ? 3u : 0u;
    const unsigned blk = xcd_chunked_block(blockIdx.x, gridDim.x);
    const unsigned w0 = blk * (GB * U) + threadIdx.x;
    v4f v[U];
    float *o[U];
    unsigned q[U], m[U], sq[U], jn[U];
    unsigned long r[U];
    int j[U];
    float dx[U][3];
    bool live[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const unsigned w = w0 + u * GB;
        live[u] = w < total;
        sq[u] = fdiv(live[u] ? w : 0u, qd);
        const unsigned rem = (live[u] ? w : 0u) - sq[u] * qd.d;
        jn[u] = fdiv(rem, pd);
        q[u] = rem - jn[u] * pieces;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { m[u] = order ? (unsigned)order[sq[u]] : sq[u]; r[u] = (unsigned long)m[u] * nsd.d + jn[u]; }
#pragma unroll
    for (int u = 0; u < U; ++u) { j[u] = idx[r[u]]; o[u] = output + r[u] * oc; }
    // (every lane issues the feature piece AND, clamped, nothing else under a branch: the coordinate lanes read xyz / new_xyz)
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const unsigned long jc = (unsigned long)max(j[u], 0);
        v[u] = reinterpret_cast<const v4f *>(feat)[jc * cv + min(q[u], cv - 1)];
        if (with_xyz && q[u] >= cv) {   // (round 4: only the row's coordinate lane -- six scalar gathers in EVERY lane cost more than the branch)
#pragma unroll
            for (int a = 0; a < 3; ++a) dx[u][a] = xyz[jc * 3 + a] - new_xyz[(unsigned long)m[u] * 3 + a];
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (!live[u]) continue;
        if (q[u] < cv) {
            *reinterpret_cast<u4f *>(o[u] + sh + 4 * q[u]) = (u4f)(j[u] >= 0 ? v[u] : (v4f)(0.f));
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a) o[u][a] = j[u] >= 0 ? dx[u][a] : 0.f;
        }
    }
}

// pointops.grouping(with_xyz) rows through LDS (round 4).  The rows of a query point are nsample * (3 + c) floats in a row, a multiple of 16
// bytes when nsample % 4 == 0 -- but a single row (3 + c floats) is not: group_fwd_rows_ord stores its 16-byte feature pieces at
// dword-aligned addresses (split by the hardware) and the three coordinates as scalars with a 4 * (3 + c)-byte stride, and ran at 0.39 of
// the HBM peak where the same table without coordinates runs at 0.70.  Here a block assembles the rows of `ppb` query points in LDS
// (same gathers) and writes every point's chunk out as whole aligned float4 lines.
template <int TPR, int U>   // TPR lanes per row, U float4 feature pieces per lane and trip
__global__ __launch_bounds__(GB) void group_fwd_lds(unsigned npts, unsigned ppb, FastDiv nsd, FastDiv c4d /* nsample * oc / 4 */, int c,
                                                    const int *__restrict__ order, const float *__restrict__ feat,
                                                    const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                    const int *__restrict__ idx, float *__restrict__ output) {
    extern __shared__ __attribute__((aligned(16))) float g_stage[];   // [ppb * nsample * oc] | point ids [ppb]
    const unsigned nsample = nsd.d, cv = c >> 2, oc = c + 3;
    const unsigned chunk = nsample * oc;                      // floats per query point (% 4 == 0)
    unsigned *pid = reinterpret_cast<unsigned *>(g_stage + ppb * chunk);
    const unsigned blk = xcd_chunked_block(blockIdx.x, gridDim.x);
    const unsigned s0 = blk * ppb, np = min(ppb, npts - s0), rows = np * nsample;   // rows <= GB / TPR
    // TPR lanes per row: visiting order -> neighbour id -> (the row's feature pieces q = t, t + TPR, ... | coordinates t, t + TPR, ..): one
    // dependent chain of three loads for everything the row needs
    const unsigned rl = min(threadIdx.x / TPR, rows - 1), t = threadIdx.x % TPR;
    const bool live = threadIdx.x / TPR < rows;
    const unsigned pl = fdiv(rl, nsd), sq = s0 + pl;
    const unsigned m = order ? (unsigned)order[sq] : sq;
    const int j = idx[(unsigned long)m * nsample + (rl - pl * nsample)];
    const unsigned long jc = (unsigned long)max(j, 0);
    float dx[(3 + TPR - 1) / TPR];
#pragma unroll
    for (int i = 0; i < (3 + TPR - 1) / TPR; ++i) {
        const unsigned a = min(t + i * TPR, 2u);
        dx[i] = xyz[jc * 3 + a] - new_xyz[(unsigned long)m * 3 + a];
    }
    float *o = g_stage + rl * oc;
    for (unsigned q0 = t; q0 < cv; q0 += TPR * U) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = reinterpret_cast<const v4f *>(feat)[jc * cv + min(q0 + TPR * u, cv - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned q = q0 + TPR * u;
            if (live && q < cv) {
                const v4f z = j >= 0 ? v[u] : (v4f)(0.f);
                o[3 + 4 * q + 0] = z.x; o[3 + 4 * q + 1] = z.y; o[3 + 4 * q + 2] = z.z; o[3 + 4 * q + 3] = z.w;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < (3 + TPR - 1) / TPR; ++i)
        if (live && t + i * TPR < 3) o[t + i * TPR] = j >= 0 ? dx[i] : 0.f;
    if (live && t == 0 && rl == pl * nsample) pid[pl] = m;
    __syncthreads();
    const unsigned c4 = chunk >> 2, tot4 = np * c4;
    for (unsigned e = threadIdx.x; e < tot4; e += GB) {
        const unsigned p2 = fdiv(e, c4d), w = e - p2 * c4;
        st_stream(reinterpret_cast<v4f *>(output) + (unsigned long)pid[p2] * c4 + w, reinterpret_cast<const v4f *>(g_stage)[e]);
    }
}

// subtraction_cuda_kernel.cu:5-16, float4 per lane (rowd = nsample * cv: element -> query point)
template <int U>
__global__ __launch_bounds__(GB) void sub_fwd32(unsigned total, FastDiv cvd, FastDiv rowd, const v4f *__restrict__ input1,
                                                const v4f *__restrict__ input2, const int *__restrict__ idx,
                                                v4f *__restrict__ output) {
    const unsigned stride = gridDim.x * GB, cv = cvd.d;
    unsigned e = blockIdx.x * GB + threadIdx.x;
    for (; e < total && total - e > (U - 1) * stride; e += U * stride) {
        v4f a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned ee = e + u * stride, r = fdiv(ee, cvd), col = ee - r * cv, n = fdiv(ee, rowd);
            const int j = idx[r];
            a[u] = input1[(unsigned long)n * cv + col];
            b[u] = j >= 0 ? input2[(unsigned long)j * cv + col] : (v4f)(0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) st_stream(output + e + u * stride, a[u] - b[u]);
    }
    for (; e < total; e += stride) {
        const unsigned r = fdiv(e, cvd), col = e - r * cv, n = fdiv(e, rowd);
        const int j = idx[r];
        const v4f b = j >= 0 ? input2[(unsigned long)j * cv + col] : (v4f)(0.f);
        st_stream(output + e, input1[(unsigned long)n * cv + col] - b);
    }
}

// interpolation_cuda_kernel.cu:5-18, float4 per lane, K neighbours gathered before the first multiply-add
template <int K>
__global__ __launch_bounds__(GB) void interp_fwd32(unsigned total, FastDiv cvd, int k_rt, const v4f *__restrict__ input,
                                                   const int *__restrict__ idx, const float *__restrict__ weight,
                                                   v4f *__restrict__ output) {
    const unsigned stride = gridDim.x * GB, cv = cvd.d;
    for (unsigned e = blockIdx.x * GB + threadIdx.x; e < total; e += stride) {
        const unsigned r = fdiv(e, cvd), col = e - r * cv;
        v4f acc = (v4f)(0.f);
        if (K > 0) {
            int j[K > 0 ? K : 1]; float w[K > 0 ? K : 1]; v4f v[K > 0 ? K : 1];
#pragma unroll
            for (int i = 0; i < K; ++i) { j[i] = idx[(unsigned long)r * K + i]; w[i] = weight[(unsigned long)r * K + i]; }
#pragma unroll
            for (int i = 0; i < K; ++i) v[i] = j[i] >= 0 ? input[(unsigned long)j[i] * cv + col] : (v4f)(0.f);
#pragma unroll
            for (int i = 0; i < K; ++i) acc += v[i] * w[i];   // same summation order as the reference (i ascending)
        } else {
            for (int i = 0; i < k_rt; ++i) {
                const int j = idx[(unsigned long)r * k_rt + i];
                const float w = weight[(unsigned long)r * k_rt + i];
                if (j >= 0) acc += input[(unsigned long)j * cv + col] * w;
            }
        }
        st_stream(output + e, acc);
    }
}

// aggregation_cuda_kernel.cu:5-20 with 4 consecutive channels per lane (needs c % 4 == 0 and w_c % 4 == 0: the four
// channels then use four consecutive weights)
__global__ __launch_bounds__(GB) void agg_fwd32(unsigned total, FastDiv cvd, int nsample, unsigned wv,
                                                const v4f *__restrict__ input, const v4f *__restrict__ position,
                                                const v4f *__restrict__ weight, const int *__restrict__ idx,
                                                v4f *__restrict__ output) {
    const unsigned stride = gridDim.x * GB, cv = cvd.d;
    for (unsigned e = blockIdx.x * GB + threadIdx.x; e < total; e += stride) {
        const unsigned r = fdiv(e, cvd), col = e - r * cv, wcol = col % wv;
        v4f acc = (v4f)(0.f);
        const unsigned long base = (unsigned long)r * nsample;
#pragma unroll 4
        for (int s = 0; s < nsample; ++s) {
            const int j = idx[base + s];
            const v4f in = j >= 0 ? input[(unsigned long)j * cv + col] : (v4f)(0.f);
            acc += (in + __builtin_nontemporal_load(position + (base + s) * cv + col)) * weight[(base + s) * wv + wcol];
        }
        output[e] = acc;
    }
}

// pointops.grouping() twin, with_xyz or not: each lane builds 4 consecutive floats of the flat (rows, oc) output (they may
// straddle the xyz | feat boundary or two rows) and stores them as one 16-B line piece
__global__ __launch_bounds__(GB) void group_fwd32(unsigned total4, unsigned total, FastDiv ocd, FastDiv nsd, int c, int with_xyz,
                                                  const float *__restrict__ feat, const float *__restrict__ xyz,
                                                  const float *__restrict__ new_xyz, const int *__restrict__ idx,
                                                  float *__restrict__ output) {
    const unsigned stride = gridDim.x * GB, oc = ocd.d, sh = with_xyz ? 3u : 0u;
    for (unsigned q = blockIdx.x * GB + threadIdx.x; q < total4; q += stride) {
        const unsigned e0 = q * 4;
        unsigned r = fdiv(e0, ocd), col = e0 - r * oc;
        int j = idx[r];
        float o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v = 0.f;
            if (e0 + u < total) {
                if (col < sh) { if (j >= 0) v = xyz[(unsigned long)j * 3 + col] - new_xyz[(unsigned long)fdiv(r, nsd) * 3 + col]; }
                else if (j >= 0) v = feat[(unsigned long)j * c + (col - sh)];
            }
            o[u] = v;
            if (++col == oc) { col = 0; ++r; j = (e0 + u + 1 < total) ? idx[r] : -1; }
        }
        if (e0 + 3 < total) st_stream(reinterpret_cast<v4f *>(output) + q, (v4f){o[0], o[1], o[2], o[3]});
        else for (int u = 0; u < 4 && e0 + u < total; ++u) output[e0 + u] = o[u];
    }
}



// pointops.grouping() twin for c % 4 == 0: lane = (row, piece) with c/4 float4 pieces of the feature row + one xyz piece per row.
// Feature pieces are 16-B loads (aligned in the table) and 16-B stores at dword-aligned addresses (rows of 3 + c floats are not
// 16-B aligned; gfx950 splits such stores in hardware), the xyz piece is 3 scalars.
__global__ __launch_bounds__(GB) void group_fwd_rows(unsigned total, FastDiv pd, FastDiv nsd, int c, int with_xyz,
                                                     const float *__restrict__ feat, const float *__restrict__ xyz,
                                                     const float *__restrict__ new_xyz, const int *__restrict__ idx,
                                                     float *__restrict__ output) {
    typedef float u4f __attribute__((ext_vector_type(4), aligned(4)));   // 16 bytes, dword alignment
    const unsigned stride = gridDim.x * GB, pieces = pd.d, cv = c >> 2, oc = c + (with_xyz ? 3 : 0), sh = with_xyz ? 3u : 0u;
    for (unsigned e = blockIdx.x * GB + threadIdx.x; e < total; e += stride) {
        const unsigned r = fdiv(e, pd), q = e - r * pieces;
        const int j = idx[r];
        float *o = output + (size_t)r * oc;
        if (q < cv) {
            const v4f v = j >= 0 ? reinterpret_cast<const v4f *>(feat)[(unsigned long)j * cv + q] : (v4f)(0.f);
            *reinterpret_cast<u4f *>(o + sh + 4 * q) = (u4f)v;
        } else {   // the xyz piece (only when with_xyz)
            const unsigned m = fdiv(r, nsd);
#pragma unroll
            for (int a = 0; a < 3; ++a) o[a] = j >= 0 ? xyz[(unsigned long)j * 3 + a] - new_xyz[(unsigned long)m * 3 + a] : 0.f;
        }
    }
}

// aggregation_cuda_kernel.cu:22-39 with the grad_weight sums of one row reduced across the wave before they leave it: the
// c / w_c channels that share a weight sit w_c lanes apart in the SAME wave (needs 64 % w_c == 0 and c | 64 or 64 | c), so
// xor-shuffles over w_c, 2 w_c, ... replace c / w_c atomics per element by a plain store (c <= 64) or c / 64 atomics.
__global__ __launch_bounds__(GB) void agg_bwd32(unsigned total, FastDiv cd, int nsample, int w_c, int span,
                                                const float *__restrict__ input, const float *__restrict__ position,
                                                const float *__restrict__ weight, const int *__restrict__ idx,
                                                const float *__restrict__ grad_output, float *__restrict__ grad_input,
                                                float *__restrict__ grad_position, float *__restrict__ grad_weight) {
    const unsigned stride = gridDim.x * GB, c = cd.d;
    const unsigned padded = (total + 63u) & ~63u;   // whole waves iterate together (shuffles need every lane)
    for (unsigned e = blockIdx.x * GB + threadIdx.x; e < padded; e += stride) {
        const bool live = e < total;
        const unsigned r = live ? fdiv(e, cd) : 0, col = live ? e - r * c : 0, wcol = col % w_c;
        const float go = live ? grad_output[e] : 0.f;
        const unsigned long base = (unsigned long)r * nsample;
        for (int s = 0; s < nsample; ++s) {
            float gw = 0.f;
            if (live) {
                const int j = idx[base + s];
                const unsigned long pi = (base + s) * c + col;
                const float w = weight[(base + s) * w_c + wcol];
                const float in = j >= 0 ? input[(unsigned long)j * c + col] : 0.f;
                if (j >= 0 && grad_input) pdf_atomic_add(grad_input + (unsigned long)j * c + col, go * w);
                st_stream(grad_position + pi, go * w);
                gw = go * (in + __builtin_nontemporal_load(position + pi));
            }
            for (int o = w_c; o < span; o <<= 1) gw += __shfl_xor(gw, o, 64);
            if (live && (col & (span - 1)) < (unsigned)w_c) {
                float *dst = grad_weight + (base + s) * w_c + wcol;
                if (span == (int)c) *dst += gw;   // the whole row was in this wave (pre-zeroed accumulate target)
                else pdf_atomic_add(dst, gw);
            }
        }
    }
}

// aggregation_cuda_kernel.cu:22-39 WITHOUT grad_input (the caller forms it with pdf_seg_sum_weighted), 4 consecutive channels per
// lane: grad_position = g_out * w streamed out as 16-byte pieces; grad_weight[n, s, wc] = sum over the c / w_c channels that share
// the weight of g_out * (input[idx] + position): those channels sit wv = w_c / 4 lanes apart in the SAME wave (needs cv = c / 4 <= 64,
// cv and wv powers of two), so xor-shuffles over wv, 2 wv, ... replace the atomics and the first wv lanes of a row store the result.
__global__ __launch_bounds__(GB) void agg_bwd_vec(unsigned total, FastDiv cvd, int nsample, unsigned wv,
                                                  const v4f *__restrict__ input, const v4f *__restrict__ position,
                                                  const v4f *__restrict__ weight, const int *__restrict__ idx,
                                                  const v4f *__restrict__ grad_output, v4f *__restrict__ grad_position,
                                                  v4f *__restrict__ grad_weight) {
    const unsigned stride = gridDim.x * GB, cv = cvd.d;
    const unsigned padded = (total + 63u) & ~63u;   // whole waves iterate together (the shuffles need every lane)
    for (unsigned e = blockIdx.x * GB + threadIdx.x; e < padded; e += stride) {
        const bool live = e < total;
        const unsigned r = live ? fdiv(e, cvd) : 0, col = live ? e - r * cv : 0, wcol = col % wv;
        const v4f go = live ? grad_output[e] : (v4f)(0.f);
        const unsigned long base = (unsigned long)r * nsample;
#pragma unroll 2
        for (int s = 0; s < nsample; ++s) {
            v4f gw = (v4f)(0.f);
            if (live) {
                const int j = idx[base + s];
                const v4f in = j >= 0 ? input[(unsigned long)j * cv + col] : (v4f)(0.f);
                const v4f w = weight[(base + s) * wv + wcol];
                st_stream(grad_position + (base + s) * cv + col, go * w);
                gw = go * (in + __builtin_nontemporal_load(position + (base + s) * cv + col));
            }
            for (unsigned o = wv; o < cv; o <<= 1) {
                gw.x += __shfl_xor(gw.x, o, 64); gw.y += __shfl_xor(gw.y, o, 64); gw.z += __shfl_xor(gw.z, o, 64); gw.w += __shfl_xor(gw.w, o, 64);
            }
            if (live && col < wv) grad_weight[(base + s) * wv + col] = gw;
        }
    }
}

#define DISPATCH_VEC(V_, ...)                 \
    switch (V_) {                             \
        case 4: { constexpr int V = 4; __VA_ARGS__; } break; \
        case 2: { constexpr int V = 2; __VA_ARGS__; } break; \
        default: { constexpr int V = 1; __VA_ARGS__; } break; \
    }

}  // namespace

// The forward gathers with an optional visiting order of the queries (a permutation of 0 .. m-1, nullptr = storage order): same
// output either way; see grouping_fwd_ord.  The reference-ABI entry points below are these with order = nullptr.
extern "C" int pdf_grouping_forward_ordered(int m, int nsample, int c, const float *input, const int *idx, const int *order, float *output,
                                            void *stream) {
    if (m == 0) return PDF_OK;
    if (m < 0 || nsample < 1 || c < 1 || !input || !idx || !output) return PDF_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int v = pick_vec(c);
    const long rows = (long)m * nsample;
    if (v == 4 && rows * (c / 4) < (1L << 31)) {
        constexpr int U = 4;
        const unsigned total = (unsigned)(rows * (c / 4));
        grouping_fwd_ord<U><<<(total + GB * U - 1) / (GB * U), GB, 0, s>>>(total, mk_fastdiv(c / 4), mk_fastdiv(nsample * (c / 4)), nsample, order,
                                                                          (const v4f *)input, idx, (v4f *)output);
        return pdf_launch_status();
    }
    DISPATCH_VEC(v, (grouping_fwd_kernel<V><<<grid_for(rows * (c / V)), GB, 0, s>>>(rows, c / V, input, idx, output)));
    return pdf_launch_status();
}

extern "C" int pdf_grouping_forward(int m, int nsample, int c, const float *input, const int *idx, float *output, void *stream) {
    return pdf_grouping_forward_ordered(m, nsample, c, input, idx, nullptr, output, stream);
}

extern "C" int pdf_grouping_backward(int m, int nsample, int c, const float *grad_output, const int *idx, float *grad_input, void *stream) {
    if (m == 0) return PDF_OK;
    if (m < 0 || nsample < 1 || c < 1 || !grad_output || !idx || !grad_input) return PDF_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // scatter-adds run one float per lane (64 consecutive floats = whole 128-byte lines per instruction): the atomic units
    // bill per request, and the 16-byte-per-lane shape splits every line into four requests (see fused_layer.hip)
    const int v = 1;
    const long rows = (long)m * nsample;
    DISPATCH_VEC(v, (grouping_bwd_kernel<V><<<grid_for(rows * (c / V)), GB, 0, s>>>(rows, c / V, grad_output, idx, grad_input)));
    return pdf_launch_status();
}

extern "C" int pdf_group_forward_ordered(int m, int nsample, int c, int with_xyz, const float *feat, const float *xyz, const float *new_xyz,
                                         const int *idx, const int *order, float *output, void *stream) {
    if (m == 0) return PDF_OK;
    if (m < 0 || nsample < 1 || c < 1 || !feat || !idx || !output) return PDF_ERR_BAD_ARG;
    if (with_xyz && (!xyz || !new_xyz)) return PDF_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long rows = (long)m * nsample;
    const int oc = c + (with_xyz ? 3 : 0);
    // rows through LDS: narrow rows only (measured in one process, L1 c = 32 nsample 8: 65.5 us against 74.5 for the direct stores; L2
    // c = 64 nsample 16: 45.7 against 44.5 -- there the 16-byte pieces already make whole lines)
    if (c % 4 == 0 && c <= 32 && with_xyz && nsample % 4 == 0 && nsample <= 64 && rows * (c / 4 + 1) < (1L << 31) &&
        getenv("PDFOPS_GROUP_LDS_OFF") == nullptr) {
        // narrow rows: 2 lanes per row, 128 rows per block (a block's fixed latency -- three dependent loads, one barrier -- over 18 KB of
        // output instead of 9); wide rows: 4 lanes per row, 64 rows
        const bool narrow = c <= 32 && nsample <= 128;
        const unsigned ppb = (unsigned)std::max(1, (narrow ? 128 : 64) / nsample);
        const size_t lds = ((size_t)ppb * nsample * oc + ppb) * sizeof(float);
        if (narrow) group_fwd_lds<2, 4><<<(m + ppb - 1) / ppb, GB, lds, s>>>((unsigned)m, ppb, mk_fastdiv((unsigned)nsample), mk_fastdiv((unsigned)(nsample * oc / 4)),
                                                                            c, order, feat, xyz, new_xyz, idx, output);
        else group_fwd_lds<4, 4><<<(m + ppb - 1) / ppb, GB, lds, s>>>((unsigned)m, ppb, mk_fastdiv((unsigned)nsample), mk_fastdiv((unsigned)(nsample * oc / 4)), c,
                                                                    order, feat, xyz, new_xyz, idx, output);
        return pdf_launch_status();
    }
    if (c % 4 == 0 && rows * (c / 4 + 1) < (1L << 31)) {
        const unsigned pieces = (unsigned)(c / 4 + (with_xyz ? 1 : 0)), total = (unsigned)(rows * pieces);
        constexpr int U = 4;
        group_fwd_rows_ord<U><<<(total + GB * U - 1) / (GB * U), GB, 0, s>>>(total, mk_fastdiv(pieces), mk_fastdiv(nsample), mk_fastdiv(nsample * pieces),
                                                                           c, with_xyz, order, feat, xyz, new_xyz, idx, output);
        return pdf_launch_status();
    }
    if (rows * oc < (1L << 31)) {
        const unsigned total = (unsigned)(rows * oc), total4 = (total + 3) / 4;
        group_fwd32<<<grid_for(total4), GB, 0, s>>>(total4, total, mk_fastdiv(oc), mk_fastdiv(nsample), c, with_xyz, feat, xyz, new_xyz, idx, output);
        return pdf_launch_status();
    }
    group_fwd_kernel<<<grid_for(rows * oc), GB, 0, s>>>(rows, nsample, c, with_xyz, feat, xyz, new_xyz, idx, output);
    return pdf_launch_status();
}

extern "C" int pdf_group_forward(int m, int nsample, int c, int with_xyz, const float *feat, const float *xyz,
                                 const float *new_xyz, const int *idx, float *output, void *stream) {
    return pdf_group_forward_ordered(m, nsample, c, with_xyz, feat, xyz, new_xyz, idx, nullptr, output, stream);
}

extern "C" int pdf_group_backward(int m, int nsample, int c, int with_xyz, const float *grad_output, const int *idx,
                                  float *grad_feat, void *stream) {
    if (m == 0) return PDF_OK;
    if (m < 0 || nsample < 1 || c < 1 || !grad_output || !idx || !grad_feat) return PDF_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long rows = (long)m * nsample;
    group_bwd_kernel<<<grid_for(rows * c), GB, 0, s>>>(rows, c, with_xyz, grad_output, idx, grad_feat);
    return pdf_launch_status();
}

extern "C" int pdf_interpolation_forward_ordered(int n, int c, int k, const float *input, const int *idx, const float *weight, const int *order,
                                                 float *output, void *stream) {
    if (n == 0) return PDF_OK;
    if (n < 0 || c < 1 || k < 1 || !input || !idx || !weight || !output) return PDF_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int v = pick_vec(c);
    if (v == 4 && (long)n * (c / 4) < (1L << 31) && (long)n * k < (1L << 31)) {
        const unsigned total = (unsigned)((long)n * (c / 4));
        const FastDiv cvd = mk_fastdiv(c / 4);
        if (k == 3) interp_fwd_ord<3><<<(total + GB - 1) / GB, GB, 0, s>>>(total, cvd, order, (const v4f *)input, idx, weight, (v4f *)output);
        else interp_fwd32<0><<<grid_for(total), GB, 0, s>>>(total, cvd, k, (const v4f *)input, idx, weight, (v4f *)output);
        return pdf_launch_status();
    }
    DISPATCH_VEC(v, (interp_fwd_kernel<V><<<grid_for((long)n * (c / V)), GB, 0, s>>>(n, c / V, k, input, idx, weight, output)));
    return pdf_launch_status();
}

extern "C" int pdf_interpolation_forward(int n, int c, int k, const float *input, const int *idx, const float *weight, float *output, void *stream) {
    return pdf_interpolation_forward_ordered(n, c, k, input, idx, weight, nullptr, output, stream);
}

extern "C" int pdf_interpolation_backward(int n, int c, int k, const float *grad_output, const int *idx, const float *weight, float *grad_input, void *stream) {
    if (n == 0) return PDF_OK;
    if (n < 0 || c < 1 || k < 1 || !grad_output || !idx || !weight || !grad_input) return PDF_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // scatter-adds run one float per lane (64 consecutive floats = whole 128-byte lines per instruction): the atomic units
    // bill per request, and the 16-byte-per-lane shape splits every line into four requests (see fused_layer.hip)
    const int v = 1;
    DISPATCH_VEC(v, (interp_bwd_kernel<V><<<grid_for((long)n * (c / V)), GB, 0, s>>>(n, c / V, k, grad_output, idx, weight, grad_input)));
    return pdf_launch_status();
}

extern "C" int pdf_interpolation_weights(int n, int k, const float *dist2, float *weight, void *stream) {
    if (n == 0) return PDF_OK;
    if (n < 0 || k < 1 || !dist2 || !weight) return PDF_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    interp_weights_kernel<<<grid_for(n), GB, 0, s>>>(n, k, dist2, weight);
    return pdf_launch_status();
}

extern "C" int pdf_subtraction_forward_ordered(int n, int nsample, int c, const float *input1, const float *input2, const int *idx,
                                               const int *order, float *output, void *stream) {
    if (n == 0) return PDF_OK;
    if (n < 0 || nsample < 1 || c < 1 || !input1 || !input2 || !idx || !output) return PDF_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int v = pick_vec(c);
    const long rows = (long)n * nsample;
    if (v == 4 && rows * (c / 4) < (1L << 31)) {
        constexpr int U = 4;
        const unsigned total = (unsigned)(rows * (c / 4));
        sub_fwd_ord<U><<<(total + GB * U - 1) / (GB * U), GB, 0, s>>>(total, mk_fastdiv(c / 4), mk_fastdiv((unsigned)nsample * (c / 4)), nsample, order,
                                                                     (const v4f *)input1, (const v4f *)input2, idx, (v4f *)output);
        return pdf_launch_status();
    }
    DISPATCH_VEC(v, (sub_fwd_kernel<V><<<grid_for(rows * (c / V)), GB, 0, s>>>(rows, nsample, c / V, input1, input2, idx, output)));
    return pdf_launch_status();
}

extern "C" int pdf_subtraction_forward(int n, int nsample, int c, const float *input1, const float *input2, const int *idx, float *output, void *stream) {
    return pdf_subtraction_forward_ordered(n, nsample, c, input1, input2, idx, nullptr, output, stream);
}

extern "C" int pdf_subtraction_backward(int n, int nsample, int c, const int *idx, const float *grad_output, float *grad_input1, float *grad_input2, void *stream) {
    if (n == 0) return PDF_OK;
    if (n < 0 || nsample < 1 || c < 1 || !idx || !grad_output || !grad_input1) return PDF_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // scatter-adds run one float per lane (64 consecutive floats = whole 128-byte lines per instruction): the atomic units
    // bill per request, and the 16-byte-per-lane shape splits every line into four requests (see fused_layer.hip)
    const int v = grad_input2 ? 1 : pick_vec(c);   // (no scatter target: row sums only, 16 bytes per lane)
    DISPATCH_VEC(v, (sub_bwd_kernel<V><<<grid_for((long)n * (c / V)), GB, 0, s>>>(n, nsample, c / V, idx, grad_output, grad_input1, grad_input2)));
    return pdf_launch_status();
}

extern "C" int pdf_aggregation_forward_ordered(int n, int nsample, int c, int w_c, const float *input, const float *position,
                                               const float *weight, const int *idx, const int *order, float *output, void *stream) {
    if (n == 0) return PDF_OK;
    if (n < 0 || nsample < 1 || c < 1 || w_c < 1 || !input || !position || !weight || !idx || !output) return PDF_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (c % 4 == 0 && w_c % 4 == 0 && (long)n * nsample * (c / 4) < (1L << 31)) {
        const unsigned total = (unsigned)((long)n * (c / 4));
        agg_fwd_ord<<<(total + GB - 1) / GB, GB, 0, s>>>(total, mk_fastdiv(c / 4), nsample, (unsigned)(w_c / 4), order, (const v4f *)input,
                                                       (const v4f *)position, (const v4f *)weight, idx, (v4f *)output);
        return pdf_launch_status();
    }
    agg_fwd_kernel<<<grid_for((long)n * c), GB, 0, s>>>(n, nsample, c, w_c, input, position, weight, idx, output);
    return pdf_launch_status();
}

extern "C" int pdf_aggregation_forward(int n, int nsample, int c, int w_c, const float *input, const float *position,
                                       const float *weight, const int *idx, float *output, void *stream) {
    return pdf_aggregation_forward_ordered(n, nsample, c, w_c, input, position, weight, idx, nullptr, output, stream);
}

extern "C" int pdf_aggregation_backward(int n, int nsample, int c, int w_c, const float *input, const float *position,
                                        const float *weight, const int *idx, const float *grad_output,
                                        float *grad_input, float *grad_position, float *grad_weight, void *stream) {
    if (n == 0) return PDF_OK;
    if (n < 0 || nsample < 1 || c < 1 || w_c < 1 || !input || !position || !weight || !idx || !grad_output || !grad_position || !grad_weight)
        return PDF_ERR_BAD_ARG;   // (grad_input may be null: the caller forms it by the segmented gather, pdf_seg_sum_weighted)
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int cv = c / 4, wv = w_c / 4;
    if (!grad_input && c % 4 == 0 && w_c % 4 == 0 && c % w_c == 0 && cv <= 64 && (cv & (cv - 1)) == 0 && (wv & (wv - 1)) == 0 &&
        (long)n * nsample * cv < (1L << 31)) {
        const unsigned total = (unsigned)((long)n * cv);
        agg_bwd_vec<<<grid_for(total), GB, 0, s>>>(total, mk_fastdiv(cv), nsample, (unsigned)wv, (const v4f *)input, (const v4f *)position,
                                                   (const v4f *)weight, idx, (const v4f *)grad_output, (v4f *)grad_position, (v4f *)grad_weight);
        return pdf_launch_status();
    }
    if ((long)n * nsample * c < (1L << 31) && 64 % w_c == 0 && c % w_c == 0 && (c % 64 == 0 || 64 % c == 0)) {
        const unsigned total = (unsigned)((long)n * c);
        agg_bwd32<<<grid_for(total), GB, 0, s>>>(total, mk_fastdiv(c), nsample, w_c, c < 64 ? c : 64, input, position, weight, idx,
                                                 grad_output, grad_input, grad_position, grad_weight);
        return pdf_launch_status();
    }
    agg_bwd_kernel<<<grid_for((long)n * c), GB, 0, s>>>(n, nsample, c, w_c, input, position, weight, idx, grad_output,
                                                        grad_input, grad_position, grad_weight);
    return pdf_launch_status();
}
