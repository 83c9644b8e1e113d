// LayerNorm over the channel dim of (n, c) rows for gfx950 -- the norm1 / norm2 / Upsample / TransitionDown norms of StratifiedTransformer
// (pointcept/models/stratified_transformer/stratified_transformer_v1m1_origin.py:123-139, 366-368, 566-569: nn.LayerNorm on 48 .. 384
// channels over 10^3 .. 10^6 rows).  torch runs them as vectorized_layer_norm_kernel + cuComputeGradInput + cuComputePartGradGammaBeta /
// cuComputeGradGammaBeta: 6.4 ms of the 50 ms ST-v1m1 step (profiles/r05_i_stratified_kernel_trace_stats.txt), at 15-25 % of the bytes'
// time.  Here: one pass per direction.
//
//   forward :  y = (x - mean) * rstd * gamma + beta,  mean / rstd per row saved (fp32, two-pass variance in registers)
//   backward:  xhat = (x - mean) * rstd, g = gy * gamma:  gx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat));
//              d gamma = sum_rows gy * xhat, d beta = sum_rows gy -- per-lane register sums over the rows a lane walks, combined per
//              workgroup in LDS in wave order, one partial row per workgroup, summed by a second kernel in a fixed order
//              (bit-reproducible; torch's two-stage reduction is too, its atomics-free path: same property, fewer passes here).
//
// Mapping: a row is c / 4 float4 pieces; G = the next power of two >= c / 4 lanes (<= 64) own a row, V = pieces per lane (1, or 2 for
// c in (256, 512]); 64 / G rows per wave and trip; row sums are xor-butterflies inside the lane group.  Bound: HBM (x read once, y
// written once; backward: gy + x read, gx written).  c % 4 == 0, c <= 512, 16-byte aligned rows.
#include "pdfops_common.h"

namespace ln {

constexpr int TB = 256;
// Workgroups in flight.  A wave's trip is one 16-byte load per lane and a butterfly: the walk is bound by the load's round trip, so the
// forward spreads the rows over as many waves as the device holds (640k x 48 rows: 91 us with 512 workgroups, 46 us with 2,048); the
// backward keeps 512 (= its partial rows: the scratch stays under the caching allocator's 1-2 MB classes for every width; 2,048 measured
// -0.2 ms per config-5 step on one box and +2 ms on another).  PDFOPS_LN_BLOCKS: both, for A/B runs.
static inline int max_blocks(bool backward) {
    static const int v = [] { const char *e = getenv("PDFOPS_LN_BLOCKS"); return e ? atoi(e) : 0; }();
    return v > 0 ? v : (backward ? 512 : 2048);
}

__device__ __forceinline__ float group_sum(float v, int G) {
    for (int m = 1; m < G; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}

struct Geo { int G, V, rpw; };   // lanes per row, float4 pieces per lane, rows per wave
static inline Geo geo_of(int c) {
    const int q = c / 4;
    Geo g;
    g.V = q > 64 ? 2 : 1;
    const int need = (q + g.V - 1) / g.V;
    g.G = 1;
    while (g.G < need) g.G <<= 1;
    g.rpw = 64 / g.G;
    return g;
}

template <int V>
__global__ __launch_bounds__(TB) void k_fwd(long n, int c, int G, const float *__restrict__ x, const float *__restrict__ gamma,
                                            const float *__restrict__ beta, float eps, float *__restrict__ y, float *__restrict__ mean,
                                            float *__restrict__ rstd) {
    const int lane = threadIdx.x & 63, rpw = 64 / G, sub = lane / G, l = lane - sub * G, q = c / 4;
    const long wave = ((long)blockIdx.x * TB + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * TB) >> 6;
    float4 gm[V], bt[V];
    bool on[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int p = l + v * G;
        on[v] = p < q;
        gm[v] = on[v] ? reinterpret_cast<const float4 *>(gamma)[p] : make_float4(0.f, 0.f, 0.f, 0.f);
        bt[v] = on[v] ? reinterpret_cast<const float4 *>(beta)[p] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float inv_c = 1.f / (float)c;
    for (long r0 = wave * rpw; r0 < n; r0 += nwaves * rpw) {
        const long r = r0 + sub;
        const bool live = r < n;
        float4 xv[V];
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            xv[v] = (live && on[v]) ? reinterpret_cast<const float4 *>(x + (size_t)r * c)[l + v * G] : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (xv[v].x + xv[v].y) + (xv[v].z + xv[v].w);
        }
        const float mu = group_sum(s, G) * inv_c;
        float ss = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            if (on[v]) {
                const float a = xv[v].x - mu, b = xv[v].y - mu, cc = xv[v].z - mu, d = xv[v].w - mu;
                ss += (a * a + b * b) + (cc * cc + d * d);
            }
        }
        const float rs = rsqrtf(group_sum(ss, G) * inv_c + eps);
        if (live) {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                if (on[v]) {
                    float4 o;
                    o.x = (xv[v].x - mu) * rs * gm[v].x + bt[v].x; o.y = (xv[v].y - mu) * rs * gm[v].y + bt[v].y;
                    o.z = (xv[v].z - mu) * rs * gm[v].z + bt[v].z; o.w = (xv[v].w - mu) * rs * gm[v].w + bt[v].w;
                    reinterpret_cast<float4 *>(y + (size_t)r * c)[l + v * G] = o;
                }
            }
            if (l == 0) { mean[r] = mu; rstd[r] = rs; }
        }
    }
}

// partial row per workgroup: [d gamma (c) | d beta (c)]
template <int V>
__global__ __launch_bounds__(TB) void k_bwd(long n, int c, int G, const float *__restrict__ gy, const float *__restrict__ x,
                                            const float *__restrict__ mean, const float *__restrict__ rstd, const float *__restrict__ gamma,
                                            float *__restrict__ gx, float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [4 waves][2 c]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, rpw = 64 / G, sub = lane / G, l = lane - sub * G, q = c / 4;
    const long wave = ((long)blockIdx.x * TB + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * TB) >> 6;
    float4 gm[V], dg[V], db[V];
    bool on[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int p = l + v * G;
        on[v] = p < q;
        gm[v] = on[v] ? reinterpret_cast<const float4 *>(gamma)[p] : make_float4(0.f, 0.f, 0.f, 0.f);
        dg[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        db[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float inv_c = 1.f / (float)c;
    for (long r0 = wave * rpw; r0 < n; r0 += nwaves * rpw) {
        const long r = r0 + sub;
        const bool live = r < n;
        const float mu = live ? mean[r] : 0.f, rs = live ? rstd[r] : 0.f;
        float4 xh[V], g[V];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const bool ok = live && on[v];
            const float4 xv = ok ? reinterpret_cast<const float4 *>(x + (size_t)r * c)[l + v * G] : make_float4(mu, mu, mu, mu);
            const float4 gv = ok ? reinterpret_cast<const float4 *>(gy + (size_t)r * c)[l + v * G] : make_float4(0.f, 0.f, 0.f, 0.f);
            xh[v] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
            g[v] = make_float4(gv.x * gm[v].x, gv.y * gm[v].y, gv.z * gm[v].z, gv.w * gm[v].w);
            s1 += (g[v].x + g[v].y) + (g[v].z + g[v].w);
            s2 += (g[v].x * xh[v].x + g[v].y * xh[v].y) + (g[v].z * xh[v].z + g[v].w * xh[v].w);
            dg[v].x += gv.x * xh[v].x; dg[v].y += gv.y * xh[v].y; dg[v].z += gv.z * xh[v].z; dg[v].w += gv.w * xh[v].w;
            db[v].x += gv.x; db[v].y += gv.y; db[v].z += gv.z; db[v].w += gv.w;
        }
        const float m1 = group_sum(s1, G) * inv_c, m2 = group_sum(s2, G) * inv_c;
        if (live) {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                if (on[v]) {
                    float4 o;
                    o.x = rs * (g[v].x - m1 - xh[v].x * m2); o.y = rs * (g[v].y - m1 - xh[v].y * m2);
                    o.z = rs * (g[v].z - m1 - xh[v].z * m2); o.w = rs * (g[v].w - m1 - xh[v].w * m2);
                    reinterpret_cast<float4 *>(gx + (size_t)r * c)[l + v * G] = o;
                }
            }
        }
    }
    // the wave's rpw row groups hold sums of different rows of the SAME channels: add them across the groups (xor over the group index),
    // then the four waves through LDS in wave order
#pragma unroll
    for (int v = 0; v < V; ++v) {
        for (int m = G; m < 64; m <<= 1) {
            dg[v].x += __shfl_xor(dg[v].x, m, 64); dg[v].y += __shfl_xor(dg[v].y, m, 64); dg[v].z += __shfl_xor(dg[v].z, m, 64); dg[v].w += __shfl_xor(dg[v].w, m, 64);
            db[v].x += __shfl_xor(db[v].x, m, 64); db[v].y += __shfl_xor(db[v].y, m, 64); db[v].z += __shfl_xor(db[v].z, m, 64); db[v].w += __shfl_xor(db[v].w, m, 64);
        }
        if (sub == 0 && on[v]) {
            reinterpret_cast<float4 *>(red + (size_t)wv * 2 * c)[l + v * G] = dg[v];
            reinterpret_cast<float4 *>(red + (size_t)wv * 2 * c + c)[l + v * G] = db[v];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * c; e += TB)
        partial[(size_t)blockIdx.x * 2 * c + e] = (red[e] + red[2 * c + e]) + (red[4 * c + e] + red[6 * c + e]);
}

// out[e] = sum over the rows of partial (rows x width) in a fixed order: a workgroup owns 16 columns, 16 row-lanes per column walk the
// rows 16 apart (up to 64 dependent loads each instead of 1,024), the 16 partial sums are added in lane order through LDS
__global__ __launch_bounds__(TB) void k_colsum(const float *__restrict__ partial, int rows, int width, float *__restrict__ out_a, float *__restrict__ out_b, int split) {
    __shared__ float red[16][17];
    const int col = threadIdx.x & 15, rl = threadIdx.x >> 4, e = blockIdx.x * 16 + col;
    float s0 = 0.f, s1 = 0.f;
    if (e < width) {
        int r = rl;
        for (; r + 16 < rows; r += 32) { s0 += partial[(size_t)r * width + e]; s1 += partial[(size_t)(r + 16) * width + e]; }
        if (r < rows) s0 += partial[(size_t)r * width + e];
    }
    red[rl][col] = s0 + s1;
    __syncthreads();
    if (rl == 0 && e < width) {
        float s = red[0][col];
#pragma unroll
        for (int k = 1; k < 16; ++k) s += red[k][col];
        if (e < split) out_a[e] = s; else out_b[e - split] = s;
    }
}

static inline int grid_for(long n, const Geo &g, bool backward) {
    const long waves = (n + g.rpw - 1) / g.rpw, blocks = (waves + 3) / 4;
    return (int)(blocks < 1 ? 1 : (blocks > max_blocks(backward) ? max_blocks(backward) : blocks));
}

}  // namespace ln

extern "C" int pdf_layernorm_supported(int c) { return c >= 4 && c % 4 == 0 && c <= 512; }

extern "C" long pdf_layernorm_partial_floats(long n, int c) {
    if (n < 1 || !pdf_layernorm_supported(c)) return 0;
    return (long)ln::grid_for(n, ln::geo_of(c), true) * 2 * c;
}

// torch.nn.functional.layer_norm(x, (c,), gamma, beta, eps) on (n, c) rows; mean / rstd (n) are saved for the backward.
extern "C" int pdf_layernorm_forward(long n, int c, const float *x, const float *gamma, const float *beta, float eps, float *y, float *mean,
                                     float *rstd, void *stream) {
    if (n < 0 || !x || !gamma || !beta || !y || !mean || !rstd) return PDF_ERR_BAD_ARG;
    if (n == 0) return PDF_OK;
    if (!pdf_layernorm_supported(c) || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gamma) |
                                         reinterpret_cast<uintptr_t>(beta)) & 15)) return PDF_ERR_UNSUPPORTED;
    const ln::Geo g = ln::geo_of(c);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (g.V == 1) ln::k_fwd<1><<<ln::grid_for(n, g, false), ln::TB, 0, s>>>(n, c, g.G, x, gamma, beta, eps, y, mean, rstd);
    else ln::k_fwd<2><<<ln::grid_for(n, g, false), ln::TB, 0, s>>>(n, c, g.G, x, gamma, beta, eps, y, mean, rstd);
    return pdf_launch_status();
}

// gx (n, c), dgamma (c), dbeta (c): all written.  partial: pdf_layernorm_partial_floats(n, c) floats of scratch.
extern "C" int pdf_layernorm_backward(long n, int c, const float *gy, const float *x, const float *mean, const float *rstd, const float *gamma,
                                      float *gx, float *partial, float *dgamma, float *dbeta, void *stream) {
    if (n < 0 || !gy || !x || !mean || !rstd || !gamma || !gx || !partial || !dgamma || !dbeta) return PDF_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!pdf_layernorm_supported(c) || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(gx) |
                                         reinterpret_cast<uintptr_t>(gamma)) & 15)) return PDF_ERR_UNSUPPORTED;
    if (n == 0) {
        if (hipMemsetAsync(dgamma, 0, sizeof(float) * c, s) != hipSuccess || hipMemsetAsync(dbeta, 0, sizeof(float) * c, s) != hipSuccess) return PDF_ERR_BAD_ARG;
        return PDF_OK;
    }
    const ln::Geo g = ln::geo_of(c);
    const int grid = ln::grid_for(n, g, true);
    const size_t lds = sizeof(float) * 8 * (size_t)c;
    if (g.V == 1) ln::k_bwd<1><<<grid, ln::TB, lds, s>>>(n, c, g.G, gy, x, mean, rstd, gamma, gx, partial);
    else ln::k_bwd<2><<<grid, ln::TB, lds, s>>>(n, c, g.G, gy, x, mean, rstd, gamma, gx, partial);
    ln::k_colsum<<<(2 * c + 15) / 16, ln::TB, 0, s>>>(partial, grid, 2 * c, dgamma, dbeta, c);
    return pdf_launch_status();
}
