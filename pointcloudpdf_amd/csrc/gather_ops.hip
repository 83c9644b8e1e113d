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
            if (j < 0) continue;
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
            if (j >= 0) pdf_atomic_add(grad_input + (long)j * c + col, go * w);
            grad_position[pi] = go * w;
            pdf_atomic_add(grad_weight + wi, go * (in + position[pi]));
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

extern "C" int pdf_grouping_forward(int m, int nsample, int c, const float *input, const int *idx, float *output, void *stream) {
    if (m < 0 || nsample < 1 || c < 1 || !input || !idx || !output) return PDF_ERR_BAD_ARG;
    if (m == 0) return PDF_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int v = pick_vec(c);
    const long rows = (long)m * nsample;
    DISPATCH_VEC(v, (grouping_fwd_kernel<V><<<grid_for(rows * (c / V)), GB, 0, s>>>(rows, c / V, input, idx, output)));
    return pdf_launch_status();
}

extern "C" int pdf_grouping_backward(int m, int nsample, int c, const float *grad_output, const int *idx, float *grad_input, void *stream) {
    if (m < 0 || nsample < 1 || c < 1 || !grad_output || !idx || !grad_input) return PDF_ERR_BAD_ARG;
    if (m == 0) return PDF_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // scatter-adds run one float per lane (64 consecutive floats = whole 128-byte lines per instruction): the atomic units
    // bill per request, and the 16-byte-per-lane shape splits every line into four requests (see fused_layer.hip)
    const int v = 1;
    const long rows = (long)m * nsample;
    DISPATCH_VEC(v, (grouping_bwd_kernel<V><<<grid_for(rows * (c / V)), GB, 0, s>>>(rows, c / V, grad_output, idx, grad_input)));
    return pdf_launch_status();
}

extern "C" int pdf_group_forward(int m, int nsample, int c, int with_xyz, const float *feat, const float *xyz,
                                 const float *new_xyz, const int *idx, float *output, void *stream) {
    if (m < 0 || nsample < 1 || c < 1 || !feat || !idx || !output) return PDF_ERR_BAD_ARG;
    if (with_xyz && (!xyz || !new_xyz)) return PDF_ERR_BAD_ARG;
    if (m == 0) return PDF_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long rows = (long)m * nsample;
    group_fwd_kernel<<<grid_for(rows * (c + (with_xyz ? 3 : 0))), GB, 0, s>>>(rows, nsample, c, with_xyz, feat, xyz, new_xyz, idx, output);
    return pdf_launch_status();
}

extern "C" int pdf_group_backward(int m, int nsample, int c, int with_xyz, const float *grad_output, const int *idx,
                                  float *grad_feat, void *stream) {
    if (m < 0 || nsample < 1 || c < 1 || !grad_output || !idx || !grad_feat) return PDF_ERR_BAD_ARG;
    if (m == 0) return PDF_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long rows = (long)m * nsample;
    group_bwd_kernel<<<grid_for(rows * c), GB, 0, s>>>(rows, c, with_xyz, grad_output, idx, grad_feat);
    return pdf_launch_status();
}

extern "C" int pdf_interpolation_forward(int n, int c, int k, const float *input, const int *idx, const float *weight, float *output, void *stream) {
    if (n < 0 || c < 1 || k < 1 || !input || !idx || !weight || !output) return PDF_ERR_BAD_ARG;
    if (n == 0) return PDF_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int v = pick_vec(c);
    DISPATCH_VEC(v, (interp_fwd_kernel<V><<<grid_for((long)n * (c / V)), GB, 0, s>>>(n, c / V, k, input, idx, weight, output)));
    return pdf_launch_status();
}

extern "C" int pdf_interpolation_backward(int n, int c, int k, const float *grad_output, const int *idx, const float *weight, float *grad_input, void *stream) {
    if (n < 0 || c < 1 || k < 1 || !grad_output || !idx || !weight || !grad_input) return PDF_ERR_BAD_ARG;
    if (n == 0) return PDF_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // scatter-adds run one float per lane (64 consecutive floats = whole 128-byte lines per instruction): the atomic units
    // bill per request, and the 16-byte-per-lane shape splits every line into four requests (see fused_layer.hip)
    const int v = 1;
    DISPATCH_VEC(v, (interp_bwd_kernel<V><<<grid_for((long)n * (c / V)), GB, 0, s>>>(n, c / V, k, grad_output, idx, weight, grad_input)));
    return pdf_launch_status();
}

extern "C" int pdf_interpolation_weights(int n, int k, const float *dist2, float *weight, void *stream) {
    if (n < 0 || k < 1 || !dist2 || !weight) return PDF_ERR_BAD_ARG;
    if (n == 0) return PDF_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    interp_weights_kernel<<<grid_for(n), GB, 0, s>>>(n, k, dist2, weight);
    return pdf_launch_status();
}

extern "C" int pdf_subtraction_forward(int n, int nsample, int c, const float *input1, const float *input2, const int *idx, float *output, void *stream) {
    if (n < 0 || nsample < 1 || c < 1 || !input1 || !input2 || !idx || !output) return PDF_ERR_BAD_ARG;
    if (n == 0) return PDF_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int v = pick_vec(c);
    const long rows = (long)n * nsample;
    DISPATCH_VEC(v, (sub_fwd_kernel<V><<<grid_for(rows * (c / V)), GB, 0, s>>>(rows, nsample, c / V, input1, input2, idx, output)));
    return pdf_launch_status();
}

extern "C" int pdf_subtraction_backward(int n, int nsample, int c, const int *idx, const float *grad_output, float *grad_input1, float *grad_input2, void *stream) {
    if (n < 0 || nsample < 1 || c < 1 || !idx || !grad_output || !grad_input1 || !grad_input2) return PDF_ERR_BAD_ARG;
    if (n == 0) return PDF_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // scatter-adds run one float per lane (64 consecutive floats = whole 128-byte lines per instruction): the atomic units
    // bill per request, and the 16-byte-per-lane shape splits every line into four requests (see fused_layer.hip)
    const int v = 1;
    DISPATCH_VEC(v, (sub_bwd_kernel<V><<<grid_for((long)n * (c / V)), GB, 0, s>>>(n, nsample, c / V, idx, grad_output, grad_input1, grad_input2)));
    return pdf_launch_status();
}

extern "C" int pdf_aggregation_forward(int n, int nsample, int c, int w_c, const float *input, const float *position,
                                       const float *weight, const int *idx, float *output, void *stream) {
    if (n < 0 || nsample < 1 || c < 1 || w_c < 1 || !input || !position || !weight || !idx || !output) return PDF_ERR_BAD_ARG;
    if (n == 0) return PDF_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    agg_fwd_kernel<<<grid_for((long)n * c), GB, 0, s>>>(n, nsample, c, w_c, input, position, weight, idx, output);
    return pdf_launch_status();
}

extern "C" int pdf_aggregation_backward(int n, int nsample, int c, int w_c, const float *input, const float *position,
                                        const float *weight, const int *idx, const float *grad_output,
                                        float *grad_input, float *grad_position, float *grad_weight, void *stream) {
    if (n < 0 || nsample < 1 || c < 1 || w_c < 1 || !input || !position || !weight || !idx || !grad_output ||
        !grad_input || !grad_position || !grad_weight)
        return PDF_ERR_BAD_ARG;
    if (n == 0) return PDF_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    agg_bwd_kernel<<<grid_for((long)n * c), GB, 0, s>>>(n, nsample, c, w_c, input, position, weight, idx, grad_output,
                                                        grad_input, grad_position, grad_weight);
    return pdf_launch_status();
}
