// Fused PointTransformerLayer passes on the matrix cores, for nsample == 16 and C >= 128 (levels 3 and 4: 12.5k / 3.1k
// points, 128 / 256 channels).  Same math and buffers as fused_layer.hip (which documents the pass structure and cites
// point_transformer_seg.py:45-78); only the mapping differs.
//
// The row-per-lane kernels keep one (point, neighbour) row per lane and stream the weights through SGPRs; at C >= 128 that
// is C * C/8 = 2048..8192 serial FMAs per lane and pass behind scalar loads, and the level-4 layers (1.5 % of the points)
// took 17 % of the step.  Here one wave works on ONE point = 16 neighbour rows at a time and the C x C/8 products run as
// v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate):
//
//   lane l: row = l & 15 (neighbour), kq = l >> 4;  channels of the lane = {16 j + 4 kq + e : e < 4} for j < C/16
//   -> every per-channel quantity (x_k[idx] row gather, x_q row, p_r, BatchNorm coefficients) is a float4 per (lane, j):
//      64 contiguous bytes per row and instruction, per-channel constants from LDS (block-wide copy, broadcast reads)
//   -> h^T = Ww1 . relu(BN1(r))^T : A = Ww1 (16 hidden units x 4 channels), B = activations^T; the D fragment
//      (hidden unit 4 (l >> 4) + reg, row l & 15) is again "4 consecutive values per lane": float4 stores of h.
//   Column statistics (sums over rows) are per-lane accumulators reduced over the 16 row-lanes once per kernel.
#include "fused_layer.h"
#include <cstdlib>

namespace flm {

using fl::cfloat_p;
using fl::LayerArgs;
using fl::WPB;
typedef float f32x4 __attribute__((ext_vector_type(4)));

bool supported(int nsample, int c) { return nsample == 16 && (c == 128 || c == 256) && getenv("PDFOPS_PT_NO_MFMA") == nullptr; }

__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ f32x4 relu4(f32x4 v) { return f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)}; }

// Block-wide LDS copy of the per-channel constants: [Wp2 (3C: channel-major, 3 per channel) | bp2 (C) | s1 (C) | t1 (C)]
template <int C>
__device__ __forceinline__ void stage_consts(float *cst, const LayerArgs &A) {
    const float *wp2 = (const float *)(uintptr_t)A.Wp2, *bp2 = (const float *)(uintptr_t)A.bp2;
    const float *s1 = (const float *)(uintptr_t)A.s1, *t1 = (const float *)(uintptr_t)A.t1;
    for (int e = threadIdx.x; e < 3 * C; e += 64 * WPB) cst[e] = wp2[e];
    for (int e = threadIdx.x; e < C; e += 64 * WPB) { cst[3 * C + e] = bp2[e]; cst[4 * C + e] = s1[e]; cst[5 * C + e] = t1[e]; }
    __syncthreads();
}

// One point's 16 neighbour rows: geometry branch of the lane's row (3-channel, cheap, replicated over the 4 kq lanes)
struct PRow {
    int nb;          // neighbour index (-1: zero row)
    float t1[3];     // Linear(3,3) output (pre-BN)
    float t1n[3];    // relu(BNp(t1))
};
__device__ __forceinline__ PRow load_prow(const LayerArgs &A, long i, int row) {
    PRow R;
    R.nb = A.idx[i * 16 + row];
    float rel[3] = {0.f, 0.f, 0.f};
    if (R.nb >= 0) {
#pragma unroll
        for (int b = 0; b < 3; ++b) rel[b] = A.p[(size_t)R.nb * 3 + b] - A.p[(size_t)i * 3 + b];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        R.t1[a] = rel[0] * A.Wp1[a * 3 + 0] + rel[1] * A.Wp1[a * 3 + 1] + rel[2] * A.Wp1[a * 3 + 2] + A.bp1[a];
        R.t1n[a] = fmaxf(R.t1[a] * A.sp[a] + A.tp[a], 0.f);
    }
    return R;
}

// p_r for the lane's four channels of group g = 4 j + kq (channels 4 g .. 4 g + 3)
__device__ __forceinline__ f32x4 pos4(const float *cst, int C, int g, const float *t1n) {
    const f32x4 w0 = ld4(cst + 12 * g), w1 = ld4(cst + 12 * g + 4), w2 = ld4(cst + 12 * g + 8), b = ld4(cst + 3 * C + 4 * g);
    const float w[12] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3], w2[0], w2[1], w2[2], w2[3]};
    f32x4 pr;
#pragma unroll
    for (int e = 0; e < 4; ++e) pr[e] = t1n[0] * w[3 * e] + t1n[1] * w[3 * e + 1] + t1n[2] * w[3 * e + 2] + b[e];
    return pr;
}

// r = x_k[nb] - x_q[i] + p_r for channel group g of the lane's row
__device__ __forceinline__ f32x4 r4_of(const LayerArgs &A, const float *cst, int C, long i, const PRow &R, int g) {
    const f32x4 xk = R.nb >= 0 ? ld4(A.xk + (size_t)R.nb * C + 4 * g) : zero4();
    const f32x4 xq = ld4(A.xq + (size_t)i * C + 4 * g);
    return (xk - xq) + pos4(cst, C, g, R.t1n);
}

// ------------------------------------------------------------------------------------------------ P3: h (+ stats of h)
template <int C, bool STATS>
__global__ __launch_bounds__(64 * WPB) void k_p3(LayerArgs A) {
    constexpr int CS = C / 8, NJ = C / 16, NOB = CS / 16, WS = C + 4;   // WS: padded row stride of the Ww1 copy
    __shared__ __attribute__((aligned(16))) float cst[6 * C];
    __shared__ __attribute__((aligned(16))) float wl[CS * WS];
    {
        const float *ww1 = (const float *)(uintptr_t)A.Ww1;
        for (int e = threadIdx.x; e < CS * C; e += 64 * WPB) wl[(e / C) * WS + e % C] = ww1[e];
    }
    stage_consts<C>(cst, A);
    const int lane = threadIdx.x & 63, row = lane & 15, kq = lane >> 4;
    const long wave_g = (long)blockIdx.x * WPB + (threadIdx.x >> 6), nwaves = (long)gridDim.x * WPB;
    const float *bw1 = (const float *)(uintptr_t)A.bw1;
    const float *wa = wl + row * WS + 4 * kq;   // A operand (hidden unit ob*16 + (l & 15), k = kq): wa[ob * 16 * WS + 16 j ..+4]
    f32x4 b4[NOB];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) b4[ob] = ld4(bw1 + ob * 16 + 4 * kq);
    f32x4 s4[NOB], ss4[NOB];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) { s4[ob] = zero4(); ss4[ob] = zero4(); }
    for (long i = wave_g; i < A.N; i += nwaves) {
        const PRow R = load_prow(A, i, row);
        f32x4 acc[NOB];
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) acc[ob] = zero4();
#pragma unroll 2
        for (int j = 0; j < NJ; ++j) {
            const int g = 4 * j + kq;
            const f32x4 r = r4_of(A, cst, C, i, R, g);
            const f32x4 y = relu4(r * ld4(cst + 4 * C + 4 * g) + ld4(cst + 5 * C + 4 * g));
            f32x4 w[NOB];
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) w[ob] = ld4(wa + ob * 16 * WS + 16 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ob][e], y[e], acc[ob], 0, 0, 0);
        }
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            const f32x4 h = acc[ob] + b4[ob];   // h[row][ob*16 + 4 kq + reg]
            *reinterpret_cast<f32x4 *>(A.H + ((size_t)i * 16 + row) * CS + ob * 16 + 4 * kq) = h;
            if (STATS) { s4[ob] += h; ss4[ob] += h * h; }
        }
    }
    if (STATS) {
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = s4[ob][r], ss = ss4[ob][r];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) { s += __shfl_xor(s, m, 64); ss += __shfl_xor(ss, m, 64); }
                if (row == 0) {
                    A.partial[wave_g * 2 * CS + ob * 16 + 4 * kq + r] = s;
                    A.partial[wave_g * 2 * CS + CS + ob * 16 + 4 * kq + r] = ss;
                }
            }
    }
}

void launch_p3(const LayerArgs &A, int c, bool stats, int grid, hipStream_t s) {
    if (c == 128) { if (stats) k_p3<128, true><<<grid, 64 * WPB, 0, s>>>(A); else k_p3<128, false><<<grid, 64 * WPB, 0, s>>>(A); }
    else          { if (stats) k_p3<256, true><<<grid, 64 * WPB, 0, s>>>(A); else k_p3<256, false><<<grid, 64 * WPB, 0, s>>>(A); }
}

}  // namespace flm
