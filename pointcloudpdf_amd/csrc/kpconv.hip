// Rigid KPConv (the StratifiedTransformer stem: KPConvSimpleBlock / KPConvResBlock, stratified_transformer_v1m1_origin.py:582-662, over
// torch_points3d's KPConvLayer / KPConv_ops) split into the part that is NOT a matrix product and two plain products:
//
//   weighted[n, k, c] = sum_m  max(0, 1 - |s[nb[n, m]] - q[n] - K[k]| / extent) * x[nb[n, m], c]        (pdf_kpconv_gather, this file)
//   out[n, o]         = sum_{k, c} weighted[n, k, c] W[k, c, o]                                        (pdf_rowlin_forward on (N, K*C))
//   backward:  g_weighted = g_out W^T,  g_W = weighted^T g_out  (pdf_rowlin_*),  g_x[nb[n, m], c] += sum_k w g_weighted[n, k, c]
//                                                                                                      (pdf_kpconv_scatter, this file)
//
// The torch composition it replaces materialises (N, M, K) distance / influence tensors (326 MB at N = 160k, M = 34, K = 15) and runs a
// batched product of 160k tiny (15 x 34)(34 x C) matrices: ~8 ms of the ST-v1m1 step.  Here 16 lanes own a query (lane = kernel
// point k, one idle), walk its M neighbours together (the neighbour's coordinates and feature row are one broadcast load per group) and
// keep the C_in accumulators of their kernel point in registers: no intermediate tensor, HBM traffic = neighbour table + gathered rows.
// -1 = no neighbour (upstream: a far "shadow" point with zero features, i.e. influence 0).  C_in <= 16, K <= 16.
#include "pdfops_common.h"

namespace {
constexpr int KB = 256;     // 16 queries per workgroup

template <int CV>   // CV = ceil(C_in / 4)
__global__ __launch_bounds__(KB) void k_kp_gather(int N, int M, int KP, int cin, const float *__restrict__ query, const float *__restrict__ support,
                                                  const int *__restrict__ nb, const float *__restrict__ x, const float *__restrict__ kpts,
                                                  float inv_extent, float *__restrict__ weighted) {
    const int k = threadIdx.x & 15;
    const long q = (long)blockIdx.x * (KB / 16) + (threadIdx.x >> 4);
    if (q >= N) return;
    const bool on = k < KP;
    const float cx = query[3 * q] + (on ? kpts[3 * k] : 0.f), cy = query[3 * q + 1] + (on ? kpts[3 * k + 1] : 0.f), cz = query[3 * q + 2] + (on ? kpts[3 * k + 2] : 0.f);
    float acc[CV * 4];
#pragma unroll
    for (int c = 0; c < CV * 4; ++c) acc[c] = 0.f;
    for (int m = 0; m < M; ++m) {
        const int j = nb[q * M + m];
        if (j < 0) continue;                                   // (uniform over the query's 16 lanes)
        const float dx = support[3 * (long)j] - cx, dy = support[3 * (long)j + 1] - cy, dz = support[3 * (long)j + 2] - cz;
        const float w = fmaxf(1.f - sqrtf(dx * dx + dy * dy + dz * dz) * inv_extent, 0.f);
        const float *xr = x + (long)j * cin;
#pragma unroll
        for (int c = 0; c < CV * 4; ++c) acc[c] += w * (c < cin ? xr[c] : 0.f);
    }
    if (on) {
        float *dst = weighted + (q * KP + k) * cin;
#pragma unroll
        for (int c = 0; c < CV * 4; ++c) if (c < cin) dst[c] = acc[c];
    }
}

// adjoint of the gather: g_x[nb[n, m], c] += sum_k w(n, m, k) * g_weighted[n, k, c]   (g_x zeroed by the caller; float atomics: rows are
// shared between queries -- as everywhere in the pointops2 backward, the order of these sums is not fixed)
template <int CV>
__global__ __launch_bounds__(KB) void k_kp_scatter(int N, int M, int KP, int cin, const float *__restrict__ query, const float *__restrict__ support,
                                                   const int *__restrict__ nb, const float *__restrict__ gw, const float *__restrict__ kpts,
                                                   float inv_extent, float *__restrict__ gx) {
    const int k = threadIdx.x & 15;
    const long q = (long)blockIdx.x * (KB / 16) + (threadIdx.x >> 4);
    const bool live = q < N, on = live && k < KP;
    const long qq = live ? q : 0;
    const float cx = query[3 * qq] + (on ? kpts[3 * k] : 0.f), cy = query[3 * qq + 1] + (on ? kpts[3 * k + 1] : 0.f), cz = query[3 * qq + 2] + (on ? kpts[3 * k + 2] : 0.f);
    float g[CV * 4];
#pragma unroll
    for (int c = 0; c < CV * 4; ++c) g[c] = (on && c < cin) ? gw[(qq * KP + k) * cin + c] : 0.f;
    for (int m = 0; m < M; ++m) {
        const int j = live ? nb[qq * M + m] : -1;
        const long jc = j < 0 ? 0 : j;
        const float dx = support[3 * jc] - cx, dy = support[3 * jc + 1] - cy, dz = support[3 * jc + 2] - cz;
        const float w = (on && j >= 0) ? fmaxf(1.f - sqrtf(dx * dx + dy * dy + dz * dz) * inv_extent, 0.f) : 0.f;
#pragma unroll
        for (int c = 0; c < CV * 4; ++c) {
            float v = w * g[c];                                // sum over the 16 lanes (kernel points) of the query
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            if (k == 0 && j >= 0 && c < cin && v != 0.f) pdf_atomic_add(gx + jc * cin + c, v);
        }
    }
}
}  // namespace

extern "C" int pdf_kpconv_supported(int kp, int cin) { return kp >= 1 && kp <= 16 && cin >= 1 && cin <= 16; }

// weighted (N, KP, C_in) is overwritten.  Bytes: 4NM (table) + 12(N + gathered rows) + 4 C_in (gathered rows) + 4 N KP C_in.
extern "C" int pdf_kpconv_gather(int n, int m, int kp, int cin, const float *query, const float *support, const int *neighbors, const float *x,
                                 const float *k_points, float extent, float *weighted, void *stream) {
    if (n < 0 || m < 1 || !query || !support || !neighbors || !x || !k_points || !weighted || !(extent > 0.f)) return PDF_ERR_BAD_ARG;
    if (!pdf_kpconv_supported(kp, cin)) return PDF_ERR_UNSUPPORTED;
    if (n == 0) return PDF_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int grid = (n + KB / 16 - 1) / (KB / 16);
    const float inv = 1.f / extent;
    switch ((cin + 3) / 4) {
    case 1: k_kp_gather<1><<<grid, KB, 0, s>>>(n, m, kp, cin, query, support, neighbors, x, k_points, inv, weighted); break;
    case 2: k_kp_gather<2><<<grid, KB, 0, s>>>(n, m, kp, cin, query, support, neighbors, x, k_points, inv, weighted); break;
    case 3: k_kp_gather<3><<<grid, KB, 0, s>>>(n, m, kp, cin, query, support, neighbors, x, k_points, inv, weighted); break;
    default: k_kp_gather<4><<<grid, KB, 0, s>>>(n, m, kp, cin, query, support, neighbors, x, k_points, inv, weighted); break;
    }
    return pdf_launch_status();
}

// grad_x (rows of x, C_in) must be zeroed by the caller; grad_weighted (N, KP, C_in).
extern "C" int pdf_kpconv_scatter(int n, int m, int kp, int cin, const float *query, const float *support, const int *neighbors,
                                  const float *grad_weighted, const float *k_points, float extent, float *grad_x, void *stream) {
    if (n < 0 || m < 1 || !query || !support || !neighbors || !grad_weighted || !k_points || !grad_x || !(extent > 0.f)) return PDF_ERR_BAD_ARG;
    if (!pdf_kpconv_supported(kp, cin)) return PDF_ERR_UNSUPPORTED;
    if (n == 0) return PDF_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int grid = (n + KB / 16 - 1) / (KB / 16);
    const float inv = 1.f / extent;
    switch ((cin + 3) / 4) {
    case 1: k_kp_scatter<1><<<grid, KB, 0, s>>>(n, m, kp, cin, query, support, neighbors, grad_weighted, k_points, inv, grad_x); break;
    case 2: k_kp_scatter<2><<<grid, KB, 0, s>>>(n, m, kp, cin, query, support, neighbors, grad_weighted, k_points, inv, grad_x); break;
    case 3: k_kp_scatter<3><<<grid, KB, 0, s>>>(n, m, kp, cin, query, support, neighbors, grad_weighted, k_points, inv, grad_x); break;
    default: k_kp_scatter<4><<<grid, KB, 0, s>>>(n, m, kp, cin, query, support, neighbors, grad_weighted, k_points, inv, grad_x); break;
    }
    return pdf_launch_status();
}
