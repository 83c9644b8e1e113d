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

bool supported(int nsample, int c) {
    return nsample == 16 && (c == 64 || c == 128 || c == 256 || c == 512) && getenv("PDFOPS_PT_NO_MFMA") == nullptr;
}

// C/8 hidden units, padded to whole 16-row MFMA blocks (C = 64: 8 units, the upper half of the block is zero padding).
// Lane (row, kq) owns hidden units {16 ob + 4 kq + e}; `hv` = those exist.  With 8 units the channel -> unit map
// (c mod 8) sends the channels of lanes kq = 2, 3 to the units of lanes kq - 2: one xor-32 exchange where that matters.
__host__ __device__ constexpr int nob_of(int c) { return c / 8 >= 16 ? c / 128 : 1; }
__host__ __device__ constexpr int csp_of(int c) { return 16 * nob_of(c); }

__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ void st4(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ f32x4 relu4(f32x4 v) { return f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)}; }
__device__ __forceinline__ const float *gp(cfloat_p p) { return (const float *)(uintptr_t)p; }

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xf, 0xf, false));
}
// all-reduce over the 16 row-lanes of a DPP row (the 16 neighbours of the point)
__device__ __forceinline__ float max16(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v)); v = fmaxf(v, dpp_f<0x4E>(v)); v = fmaxf(v, dpp_f<0x141>(v)); v = fmaxf(v, dpp_f<0x140>(v));
    return v;
}
__device__ __forceinline__ float sum16(float v) {
    v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v); v += dpp_f<0x140>(v);
    return v;
}
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- dynamic LDS layout (floats).  cst = per-channel constants [Wp2 (3C, channel-major) | bp2 (C) | s1 (C) | t1 (C)];
// W2 = padded copy of Ww2 (row stride CS + 4); TS = row stride of the 16 x 64 transposition tiles.
constexpr int TS = 68;
__host__ __device__ constexpr int w2_floats(int c) { return csp_of(c) * (csp_of(c) + 4); }

template <int C>
__device__ __forceinline__ void stage_consts(float *cst, const LayerArgs &A, bool with_bn1) {
    for (int e = threadIdx.x; e < 3 * C; e += 64 * WPB) cst[e] = gp(A.Wp2)[e];
    for (int e = threadIdx.x; e < C; e += 64 * WPB) {
        cst[3 * C + e] = gp(A.bp2)[e];
        if (with_bn1) { cst[4 * C + e] = gp(A.s1)[e]; cst[5 * C + e] = gp(A.t1)[e]; }
    }
}
template <int C>
__device__ __forceinline__ void stage_w2(float *w2, const LayerArgs &A) {
    constexpr int CS = C / 8, CSP = csp_of(C);
    for (int e = threadIdx.x; e < CSP * (CSP + 4); e += 64 * WPB) {
        const int o = e / (CSP + 4), u = e % (CSP + 4);
        w2[e] = (o < CS && u < CS) ? gp(A.Ww2)[o * CS + u] : 0.f;
    }
}
// guarded float4 of a CS-long per-unit array (zero where the lane's units do not exist)
__device__ __forceinline__ f32x4 ldu(const float *p, int o, bool hv) { return hv ? ld4(p + o) : zero4(); }
__device__ __forceinline__ f32x4 xchg32(f32x4 v) {
    return f32x4{__shfl_xor(v[0], 32, 64), __shfl_xor(v[1], 32, 64), __shfl_xor(v[2], 32, 64), __shfl_xor(v[3], 32, 64)};
}

// One point's 16 neighbour rows: geometry branch of the lane's row (3-channel, cheap, replicated over the 4 kq lanes)
struct PRow {
    int nb;          // neighbour index (-1: zero row)
    float t1[3];     // Linear(3,3) output (pre-BN)
    float t1n[3];    // relu(BNp(t1))
};
__device__ __forceinline__ PRow load_prow(const LayerArgs &A, long i, int nb) {
    PRow R;
    R.nb = nb;
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

// Attention branch of one point (16 rows): u = relu(BN2(h)), w = softmax over the rows of (u Ww2^T + bw2).
// Lane (row, kq) holds hidden units {16 ob + 4 kq + e}; the MFMA D fragment of z^T = Ww2 u^T has the same index set.
template <int C>
__device__ __forceinline__ void attn_weights(const LayerArgs &A, const float *w2, int row, int kq, const f32x4 *h, f32x4 *u, f32x4 *w) {
    constexpr int CS = C / 8, NOB = nob_of(C), WS2 = csp_of(C) + 4;
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
        const bool hv = 16 * ob + 4 * kq < CS;
        u[ob] = relu4(h[ob] * ldu(gp(A.s2), 16 * ob + 4 * kq, hv) + ldu(gp(A.t2), 16 * ob + 4 * kq, hv));   // padding lanes: 0
    }
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
        f32x4 z = zero4();
#pragma unroll
        for (int jo = 0; jo < NOB; ++jo) {
            const f32x4 a = ld4(w2 + (ob * 16 + row) * WS2 + 16 * jo + 4 * kq);   // A operand: Ww2[16 ob + (l & 15)][16 jo + 4 kq + e]
#pragma unroll
            for (int e = 0; e < 4; ++e) z = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], u[jo][e], z, 0, 0, 0);
        }
        z += ldu(gp(A.bw2), 16 * ob + 4 * kq, 16 * ob + 4 * kq < CS);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float m = max16(z[r]);
            const float ex = __expf(z[r] - m);
            w[ob][r] = ex / sum16(ex);
        }
    }
    if (CS < 16) {   // channels of lanes kq = 2, 3 use the units of lanes kq - 2
        const f32x4 x = xchg32(w[0]);
        if (kq >= 2) w[0] = x;
    }
}

// h (or G2) row fragment of the lane: units {16 ob + 4 kq + e}
template <int C>
__device__ __forceinline__ f32x4 ld_units(const float *base, long rowidx, int ob, int kq, int bf16) {
    constexpr int CS = C / 8;
    if (!(16 * ob + 4 * kq < CS)) return zero4();
    const float4 v = fl::ld_u4(base, (size_t)rowidx * CS + 16 * ob + 4 * kq, bf16);
    return f32x4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void st_units4(float *base, size_t idx, f32x4 v, int bf16) { fl::st_u4(base, idx, v[0], v[1], v[2], v[3], bf16); }

// ------------------------------------------------------------------------------------------------ P2: stats of r (slabs)
// partial row per wave-row: [sum r (C) | sum r^2 (C)]; slab y writes channels [64 y, 64 y + 64)
template <int C>
__global__ __launch_bounds__(64 * WPB) void k_p2(LayerArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *cst = lds;
    stage_consts<C>(cst, A, false);
    __syncthreads();
    const int lane = threadIdx.x & 63, row = lane & 15, kq = lane >> 4;
    const long wave_g = (long)blockIdx.x * WPB + (threadIdx.x >> 6), nwaves = (long)gridDim.x * WPB;
    const int c0 = 64 * blockIdx.y;
    f32x4 s[4], ss[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) { s[jj] = zero4(); ss[jj] = zero4(); }
    fl::PointWalk pw(A, threadIdx.x >> 6);
    int nb_next = pw.valid() ? A.idx[pw.point() * 16 + row] : -1;
    for (; pw.valid(); pw.step()) {
        const long i = pw.point();
        const int nb_cur = nb_next;
        nb_next = pw.has_next() ? A.idx[pw.next_point() * 16 + row] : -1;   // next trip's index: in flight during this trip
        const PRow R = load_prow(A, i, nb_cur);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const f32x4 r = r4_of(A, cst, C, i, R, 4 * (4 * (int)blockIdx.y + jj) + kq);
            s[jj] += r;
            ss[jj] += r * r;
        }
    }
    float *o = A.partial + wave_g * 2 * C;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = s[jj][e], b = ss[jj][e];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }
            if (row == 0) { o[c0 + 16 * jj + 4 * kq + e] = a; o[C + c0 + 16 * jj + 4 * kq + e] = b; }
        }
}

// ------------------------------------------------------------------------------------------------ P3: h (+ stats of h)
template <int C, bool STATS>
__global__ __launch_bounds__(64 * WPB) void k_p3(LayerArgs A) {
    constexpr int CS = C / 8, NJ = C / 16, NOB = nob_of(C), CSP = csp_of(C), WS = C + 4;   // WS: padded row stride of the Ww1 copy
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *cst = lds, *wl = lds + 6 * C;
    for (int e = threadIdx.x; e < CSP * C; e += 64 * WPB) wl[(e / C) * WS + e % C] = e / C < CS ? gp(A.Ww1)[e] : 0.f;
    stage_consts<C>(cst, A, true);
    __syncthreads();
    const int lane = threadIdx.x & 63, row = lane & 15, kq = lane >> 4;
    const long wave_g = (long)blockIdx.x * WPB + (threadIdx.x >> 6), nwaves = (long)gridDim.x * WPB;
    const float *wa = wl + row * WS + 4 * kq;   // A operand (hidden unit ob*16 + (l & 15), k = kq): wa[ob * 16 * WS + 16 j ..+4]
    f32x4 b4[NOB], s4[NOB], ss4[NOB];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) { b4[ob] = ldu(gp(A.bw1), ob * 16 + 4 * kq, ob * 16 + 4 * kq < CS); s4[ob] = zero4(); ss4[ob] = zero4(); }
    fl::PointWalk pw(A, threadIdx.x >> 6);
    int nb_next = pw.valid() ? A.idx[pw.point() * 16 + row] : -1;
    for (; pw.valid(); pw.step()) {
        const long i = pw.point();
        const int nb_cur = nb_next;
        nb_next = pw.has_next() ? A.idx[pw.next_point() * 16 + row] : -1;   // next trip's index: in flight during this trip
        const PRow R = load_prow(A, i, nb_cur);
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
            if (ob * 16 + 4 * kq < CS) st_units4(A.H, ((size_t)i * 16 + row) * CS + ob * 16 + 4 * kq, h, A.bf16);
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
                if (row == 0 && ob * 16 + 4 * kq < CS) {
                    A.partial[wave_g * 2 * CS + ob * 16 + 4 * kq + r] = s;
                    A.partial[wave_g * 2 * CS + CS + ob * 16 + 4 * kq + r] = ss;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------ P4: aggregation
// out[i][c] = sum_rows (x_v[nb][c] + p_r[c]) * w[row][c mod CS]
template <int C>
__global__ __launch_bounds__(64 * WPB) void k_p4(LayerArgs A) {
    constexpr int NJ = C / 16, NOB = nob_of(C);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *cst = lds, *w2 = lds + 4 * C;
    stage_consts<C>(cst, A, false);
    stage_w2<C>(w2, A);
    __syncthreads();
    const int lane = threadIdx.x & 63, row = lane & 15, kq = lane >> 4;
    const long wave_g = (long)blockIdx.x * WPB + (threadIdx.x >> 6), nwaves = (long)gridDim.x * WPB;
    fl::PointWalk pw(A, threadIdx.x >> 6);
    int nb_next = pw.valid() ? A.idx[pw.point() * 16 + row] : -1;
    for (; pw.valid(); pw.step()) {
        const long i = pw.point();
        const int nb_cur = nb_next;
        nb_next = pw.has_next() ? A.idx[pw.next_point() * 16 + row] : -1;   // next trip's index: in flight during this trip
        const PRow R = load_prow(A, i, nb_cur);
        f32x4 h[NOB], u[NOB], w[NOB];
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) h[ob] = ld_units<C>(A.H, i * 16 + row, ob, kq, A.bf16);
        attn_weights<C>(A, w2, row, kq, h, u, w);
#pragma unroll 1
        for (int j0 = 0; j0 < NJ; j0 += NOB) {
#pragma unroll
            for (int jo = 0; jo < NOB; ++jo) {   // channel 16 j + 4 kq + e -> hidden unit block j % NOB == jo
                const int g = 4 * (j0 + jo) + kq;
                const f32x4 xv = R.nb >= 0 ? ld4(A.xv + (size_t)R.nb * C + 4 * g) : zero4();
                const f32x4 v = (xv + pos4(cst, C, g, R.t1n)) * w[jo];
                f32x4 t;
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = sum16(v[e]);
                if (row == 0) st4(A.out + (size_t)i * C + 4 * g, t);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ B1
// partial row per wave: [sum g_y2 (CS) | sum g_y2*hhat (CS) | g_bw2 (CS) | g_Ww2 (CS*CS)]   (as fl::k_b1)
template <int C>
__global__ __launch_bounds__(64 * WPB) void k_b1(LayerArgs A) {
    constexpr int CS = C / 8, NOB = nob_of(C), CSP = csp_of(C), NCHK = C / 64, GS = CSP + 4, WS2 = CSP + 4, W = 3 * CS + CS * CS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *cst = lds, *w2 = cst + 4 * C;                                  // Wp2 (3C) | bp2 (C); padded Ww2
    const int wv = threadIdx.x >> 6;
    float *tile = w2 + w2_floats(C) + wv * 16 * TS;
    float *gz_t = w2 + w2_floats(C) + WPB * 16 * TS + wv * 16 * GS, *u_t = gz_t + WPB * 16 * GS;
    int *rowid = reinterpret_cast<int *>(w2 + w2_floats(C) + WPB * 16 * TS + 2 * WPB * 16 * GS) + wv * 16;
    stage_consts<C>(cst, A, false);
    stage_w2<C>(w2, A);
    for (int e = threadIdx.x; e < 2 * WPB * 16 * GS; e += 64 * WPB) (w2 + w2_floats(C) + WPB * 16 * TS)[e] = 0.f;   // unit tiles incl. padding columns
    __syncthreads();
    const int lane = threadIdx.x & 63, row = lane & 15, kq = lane >> 4;
    const long wave_g = (long)blockIdx.x * WPB + wv, nwaves = (long)gridDim.x * WPB;
    f32x4 sg[NOB], sgh[NOB], sgz[NOB], accw[NOB][NOB];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
        sg[ob] = zero4(); sgh[ob] = zero4(); sgz[ob] = zero4();
#pragma unroll
        for (int ub = 0; ub < NOB; ++ub) accw[ob][ub] = zero4();
    }
    fl::PointWalk pw(A, threadIdx.x >> 6);
    int nb_next = pw.valid() ? A.idx[pw.point() * 16 + row] : -1;
    for (; pw.valid(); pw.step()) {
        const long i = pw.point();
        const int nb_cur = nb_next;
        nb_next = pw.has_next() ? A.idx[pw.next_point() * 16 + row] : -1;   // next trip's index: in flight during this trip
        const PRow R = load_prow(A, i, nb_cur);
        f32x4 h[NOB], u[NOB], w[NOB], gw[NOB];
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) { h[ob] = ld_units<C>(A.H, i * 16 + row, ob, kq, A.bf16); gw[ob] = zero4(); }
        attn_weights<C>(A, w2, row, kq, h, u, w);
#pragma unroll 1
        for (int q = 0; q < NCHK; ++q) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int g = 4 * (4 * q + jj) + kq;   // channel 16 j + 4 kq + e  ->  hidden unit (c mod CS): block jj % NOB, same (kq, e)
                const f32x4 go = ld4(A.gout + (size_t)i * C + 4 * g);
                const f32x4 xv = R.nb >= 0 ? ld4(A.xv + (size_t)R.nb * C + 4 * g) : zero4();
                const f32x4 pr = pos4(cst, C, g, R.t1n);
                gw[jj % NOB] += go * (xv + pr);
            }
        }
        // softmax weights of the 16 rows: g_xv[nb] = sum over the inverse kNN table of g_out[i] * w  (pdf_seg_sum_weighted)
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
            if (16 * ob + 4 * kq < CS) st_units4(A.Wsm, ((size_t)i * 16 + row) * CS + 16 * ob + 4 * kq, w[ob], A.bf16);
        // softmax backward over the 16 rows, Linear(CS, CS) backward, ReLU / BN2 bookkeeping
        f32x4 gz[NOB];
        if (CS < 16) gw[0] += xchg32(gw[0]);   // 8 units: lanes kq and kq ^ 2 hold partial sums of the same units
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            const bool hv = 16 * ob + 4 * kq < CS;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dot = sum16(w[ob][r] * gw[ob][r]);
                gz[ob][r] = hv ? w[ob][r] * (gw[ob][r] - dot) : 0.f;
            }
            sgz[ob] += gz[ob];
            if (hv) {
                st4(gz_t + row * GS + 16 * ob + 4 * kq, gz[ob]);
                st4(u_t + row * GS + 16 * ob + 4 * kq, u[ob]);
            }
        }
#pragma unroll
        for (int ub = 0; ub < NOB; ++ub) {
            f32x4 gu = zero4();   // g_u^T = Ww2^T g_z^T: A operand Ww2[16 jo + 4 kq + e][16 ub + (l & 15)]
#pragma unroll
            for (int jo = 0; jo < NOB; ++jo)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    gu = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[(16 * jo + 4 * kq + e) * WS2 + 16 * ub + row], gz[jo][e], gu, 0, 0, 0);
            f32x4 gy2;
#pragma unroll
            for (int r = 0; r < 4; ++r) gy2[r] = u[ub][r] > 0.f ? gu[r] : 0.f;
            const bool hv = 16 * ub + 4 * kq < CS;
            if (hv) st_units4(A.G2, ((size_t)i * 16 + row) * CS + 16 * ub + 4 * kq, gy2, A.bf16);
            sg[ub] += gy2;
            sgh[ub] += gy2 * ((h[ub] - ldu(gp(A.mean) + 3 + C, 16 * ub + 4 * kq, hv)) * ldu(gp(A.rstd) + 3 + C, 16 * ub + 4 * kq, hv));
        }
        wave_sync();
        // g_Ww2[o][u'] += sum_rows g_z[row][o] u[row][u']  (reduction index = rows: operands re-read lanes-along-units)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float a_[NOB], b_[NOB];
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) { a_[ob] = gz_t[(4 * t + kq) * GS + 16 * ob + row]; b_[ob] = u_t[(4 * t + kq) * GS + 16 * ob + row]; }
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
                for (int ub = 0; ub < NOB; ++ub) accw[ob][ub] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[ob], b_[ub], accw[ob][ub], 0, 0, 0);
        }
        wave_sync();
    }
    float *o = A.partial + wave_g * W;
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a = sg[ob][r], b = sgh[ob][r], d = sgz[ob][r];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); d += __shfl_xor(d, m, 64); }
            if (row == 0 && 16 * ob + 4 * kq < CS) { o[16 * ob + 4 * kq + r] = a; o[CS + 16 * ob + 4 * kq + r] = b; o[2 * CS + 16 * ob + 4 * kq + r] = d; }
        }
#pragma unroll
        for (int ub = 0; ub < NOB; ++ub)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (16 * ob + 4 * kq < CS && 16 * ub + row < CS) o[3 * CS + (16 * ob + 4 * kq + r) * CS + 16 * ub + row] = accw[ob][ub][r];
    }
}

// g_h of the lane's hidden units from the stored G2 / H rows and the BN2-backward sums (`sums` = [sum g_y2 | sum g_y2*hhat])
template <int C>
__device__ __forceinline__ void hidden_grad(const LayerArgs &A, const float *sums, long i, int row, int kq, f32x4 *gh, f32x4 *h_out = nullptr) {
    constexpr int CS = C / 8, NOB = nob_of(C);
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
        const int o = 16 * ob + 4 * kq;
        const bool hv = o < CS;
        const f32x4 h = ld_units<C>(A.H, i * 16 + row, ob, kq, A.bf16), g2 = ld_units<C>(A.G2, i * 16 + row, ob, kq, A.bf16);
        const f32x4 hhat = (h - ldu(gp(A.mean) + 3 + C, o, hv)) * ldu(gp(A.rstd) + 3 + C, o, hv);
        gh[ob] = ldu(gp(A.s2), o, hv) * (g2 - ldu(sums, o, hv) * A.inv_rows - hhat * (ldu(sums + CS, o, hv) * A.inv_rows));   // padding lanes: 0
        if (h_out) h_out[ob] = h;
    }
}

// ------------------------------------------------------------------------------------------------ B2 (64-channel slabs)
// partial row per wave-row (all slabs of one blockIdx.x write disjoint columns of the same rows):
//   [sum g_y1 (C) | sum g_y1*rhat (C) | g_bw1 (CS) | g_Ww1 (CS*C)]   (as fl::k_b2)
template <int C>
__global__ __launch_bounds__(64 * WPB) void k_b2(LayerArgs A) {
    constexpr int CS = C / 8, NOB = nob_of(C), CSP = csp_of(C), GS = CSP + 4, WS = 68, W = 2 * C + CS + CS * C;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wv = threadIdx.x >> 6;
    float *cst = lds, *wl = cst + 6 * C;                                   // Ww1[:, slab], row stride 68
    float *gh_t = wl + CSP * WS + wv * 16 * GS, *v1_t = wl + CSP * WS + WPB * 16 * GS + wv * 16 * TS;
    const int slab = blockIdx.y, c0 = 64 * slab;
    for (int e = threadIdx.x; e < CSP * 64; e += 64 * WPB) wl[(e / 64) * WS + e % 64] = e / 64 < CS ? gp(A.Ww1)[(size_t)(e / 64) * C + c0 + e % 64] : 0.f;
    for (int e = threadIdx.x; e < WPB * 16 * GS; e += 64 * WPB) (wl + CSP * WS)[e] = 0.f;   // g_h tiles incl. padding columns
    stage_consts<C>(cst, A, true);
    __syncthreads();
    const int lane = threadIdx.x & 63, row = lane & 15, kq = lane >> 4;
    const long wave_g = (long)blockIdx.x * WPB + wv, nwaves = (long)gridDim.x * WPB;
    f32x4 sg[4], sgr[4], sgh[NOB], accw[NOB][4], m1[4], r1[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        sg[jj] = zero4(); sgr[jj] = zero4();
        m1[jj] = ld4(gp(A.mean) + 3 + c0 + 16 * jj + 4 * kq);
        r1[jj] = ld4(gp(A.rstd) + 3 + c0 + 16 * jj + 4 * kq);
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) accw[ob][jj] = zero4();
    }
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) sgh[ob] = zero4();
    fl::PointWalk pw(A, threadIdx.x >> 6);
    int nb_next = pw.valid() ? A.idx[pw.point() * 16 + row] : -1;
    for (; pw.valid(); pw.step()) {
        const long i = pw.point();
        const int nb_cur = nb_next;
        nb_next = pw.has_next() ? A.idx[pw.next_point() * 16 + row] : -1;   // next trip's index: in flight during this trip
        const PRow R = load_prow(A, i, nb_cur);
        f32x4 gh[NOB];
        hidden_grad<C>(A, gp(A.sums), i, row, kq, gh);
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            sgh[ob] += gh[ob];
            if (16 * ob + 4 * kq < CS) st4(gh_t + row * GS + 16 * ob + 4 * kq, gh[ob]);
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int g = 4 * (4 * slab + jj) + kq;
            f32x4 acc = zero4();   // (Ww1^T g_h)[channel 16 j + 4 kq + reg][row]
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[(16 * ob + 4 * kq + e) * WS + 16 * jj + row], gh[ob][e], acc, 0, 0, 0);
            const f32x4 r = r4_of(A, cst, C, i, R, g);
            const f32x4 y1 = r * ld4(cst + 4 * C + 4 * g) + ld4(cst + 5 * C + 4 * g);
            f32x4 gy1;
#pragma unroll
            for (int e = 0; e < 4; ++e) gy1[e] = y1[e] > 0.f ? acc[e] : 0.f;
            sg[jj] += gy1;
            sgr[jj] += gy1 * ((r - m1[jj]) * r1[jj]);
            st4(v1_t + row * TS + 16 * jj + 4 * kq, relu4(y1));
        }
        wave_sync();
        // g_Ww1[o][c] += sum_rows g_h[row][o] v1[row][c]
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float a_[NOB], b_[4];
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) a_[ob] = gh_t[(4 * t + kq) * GS + 16 * ob + row];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) b_[jj] = v1_t[(4 * t + kq) * TS + 16 * jj + row];
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) accw[ob][jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[ob], b_[jj], accw[ob][jj], 0, 0, 0);
        }
        wave_sync();
    }
    float *o = A.partial + wave_g * W;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a = sg[jj][r], b = sgr[jj][r];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }
            if (row == 0) { o[c0 + 16 * jj + 4 * kq + r] = a; o[C + c0 + 16 * jj + 4 * kq + r] = b; }
        }
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
        if (slab == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = sgh[ob][r];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) a += __shfl_xor(a, m, 64);
                if (row == 0 && 16 * ob + 4 * kq < CS) o[2 * C + 16 * ob + 4 * kq + r] = a;
            }
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (16 * ob + 4 * kq < CS) o[2 * C + CS + (size_t)(16 * ob + 4 * kq + r) * C + c0 + 16 * jj + row] = accw[ob][jj][r];
    }
}

// ------------------------------------------------------------------------------------------------ B3
// partial row per wave: [sum g_yp (3) | sum g_yp*that (3) | pad 2 | g_bp2 (C) | g_Wp2 (C*3)]   (as fl::k_b3)
#ifndef PDF_B3_ROLLED
#define PDF_B3_ROLLED 1   // rolled is 5-15 % faster at C <= 256 (measured), the unrolled form only wins at C = 512
#endif
template <int C>
__global__ __launch_bounds__(64 * WPB) void k_b3(LayerArgs A) {
    constexpr int CS = C / 8, NOB = nob_of(C), CSP = csp_of(C), NCHK = C / 64, WS = 68, W = 8 + 4 * C;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wv = threadIdx.x >> 6;
    float *cst = lds, *w2 = cst + 6 * C, *wl = w2 + w2_floats(C);        // wl: Ww1[:, 64 q ..+64] of the current chunk
    float *tile = wl + CSP * WS + wv * 32 * TS, *tile2 = tile + 16 * TS;   // g_r tile, g_pr tile
    float *t1nt = wl + CSP * WS + WPB * 32 * TS + wv * 64;
    int *rowid = reinterpret_cast<int *>(wl + CSP * WS + WPB * 32 * TS + WPB * 64) + wv * 16;
    stage_consts<C>(cst, A, true);
    stage_w2<C>(w2, A);
    const int lane = threadIdx.x & 63, row = lane & 15, kq = lane >> 4;
    const long wave_g = (long)blockIdx.x * WPB + wv, nwaves = (long)gridDim.x * WPB;
    const float *S2 = gp(A.sums), *S1 = gp(A.sums2);   // [sum g_y1 (C) | sum g_y1*rhat (C)], [sum g_y2 | sum g_y2*hhat]
    float sgp[3] = {0.f, 0.f, 0.f}, sgpt[3] = {0.f, 0.f, 0.f};
    float *o = A.partial + wave_g * W;
    // The 64-channel chunks are the OUTER loop (every chunk re-derives the cheap per-point quantities): the per-channel
    // accumulators then are four scalars, the Ww1 copy in LDS is one 64-column slab, and g_t1n (a sum over all channels)
    // is accumulated in G3 by the owning lane.
#pragma unroll 1
    for (int q = 0; q < NCHK; ++q) {
        __syncthreads();   // previous slab fully consumed (and, first trip, constants staged)
        for (int e = threadIdx.x; e < CSP * 64; e += 64 * WPB) wl[(e / 64) * WS + e % 64] = e / 64 < CS ? gp(A.Ww1)[(size_t)(e / 64) * C + 64 * q + e % 64] : 0.f;
        __syncthreads();
        float sbp2 = 0.f, awp2[3] = {0.f, 0.f, 0.f};
        for (fl::PointWalk pw(A, wv); pw.valid(); pw.step()) {
            const long i = pw.point();
            const PRow R = load_prow(A, i, A.idx[i * 16 + row]);   // (index prefetch one trip ahead: slower here at C = 256, measured)
            if (kq == 0) { t1nt[row * 4 + 0] = R.t1n[0]; t1nt[row * 4 + 1] = R.t1n[1]; t1nt[row * 4 + 2] = R.t1n[2]; }
            f32x4 gh[NOB], h[NOB], u[NOB], w[NOB];
            hidden_grad<C>(A, S1, i, row, kq, gh, h);
            attn_weights<C>(A, w2, row, kq, h, u, w);
            float gt1n[3] = {0.f, 0.f, 0.f};
#if PDF_B3_ROLLED
#pragma unroll 1
#else
#pragma unroll
#endif
            for (int jj = 0; jj < 4; ++jj) {
                const int g = 4 * (4 * q + jj) + kq;
                f32x4 acc = zero4();
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[(16 * ob + 4 * kq + e) * WS + 16 * jj + row], gh[ob][e], acc, 0, 0, 0);
                const f32x4 r = r4_of(A, cst, C, i, R, g);
                const f32x4 s1 = ld4(cst + 4 * C + 4 * g);
                const f32x4 y1 = r * s1 + ld4(cst + 5 * C + 4 * g);
                f32x4 gy1;
#pragma unroll
                for (int e = 0; e < 4; ++e) gy1[e] = y1[e] > 0.f ? acc[e] : 0.f;
                // BN1 backward: g_r = s1 * (g_y1 - mean(g_y1) - rhat * mean(g_y1 * rhat))
                const f32x4 rhat = (r - ld4(gp(A.mean) + 3 + 4 * g)) * ld4(gp(A.rstd) + 3 + 4 * g);
                const f32x4 gr = s1 * (gy1 - ld4(S2 + 4 * g) * A.inv_rows - rhat * (ld4(S2 + C + 4 * g) * A.inv_rows));
                st4(tile + row * TS + 16 * jj + 4 * kq, gr);
                f32x4 wsel = w[0];   // w[jj % NOB] with a rolled jj
#pragma unroll
                for (int t = 1; t < NOB; ++t) if (jj % NOB == t) wsel = w[t];
                const f32x4 gpr = gr + ld4(A.gout + (size_t)i * C + 4 * g) * wsel;   // + the aggregation's share of p_r
                st4(tile2 + row * TS + 16 * jj + 4 * kq, gpr);
                const f32x4 w0 = ld4(cst + 12 * g), w1 = ld4(cst + 12 * g + 4), w2v = ld4(cst + 12 * g + 8);
                const float wp[12] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3], w2v[0], w2v[1], w2v[2], w2v[3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gt1n[0] += gpr[e] * wp[3 * e]; gt1n[1] += gpr[e] * wp[3 * e + 1]; gt1n[2] += gpr[e] * wp[3 * e + 2];
                }
            }
            wave_sync();
            {   // lanes along channels: g_r rows out (256 B per row and chunk), g_xq[i] = - sum_rows g_r
                float acc = 0.f;
#pragma unroll 4
                for (int rr = 0; rr < 16; ++rr) {
                    const float v = tile[rr * TS + lane];
                    acc += v;
                    fl::st_u1_stream(A.GR, ((size_t)i * 16 + rr) * C + 64 * q + lane, v, A.bf16);   // g_xk = segmented sum of these rows
                }
                A.gxq[(size_t)i * C + 64 * q + lane] = -acc;
            }
#pragma unroll 4
            for (int rr = 0; rr < 16; ++rr) {   // g_bp2 / g_Wp2 of channel 64 q + lane
                const float v = tile2[rr * TS + lane];
                sbp2 += v;
                awp2[0] += v * t1nt[rr * 4 + 0]; awp2[1] += v * t1nt[rr * 4 + 1]; awp2[2] += v * t1nt[rr * 4 + 2];
            }
            wave_sync();
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float v = gt1n[a];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (kq == 0) {
                    float *g3 = A.G3 + ((size_t)i * 16 + row) * 3 + a;
                    if (q > 0) v += *g3;
                    if (q == NCHK - 1) {   // all channels seen: ReLU mask of BNp, BNp-backward sums
                        v = R.t1n[a] > 0.f ? v : 0.f;
                        sgp[a] += v;
                        sgpt[a] += v * ((R.t1[a] - A.mean[a]) * A.rstd[a]);
                    }
                    *g3 = v;
                }
            }
        }
        o[8 + 64 * q + lane] = sbp2;
#pragma unroll
        for (int a = 0; a < 3; ++a) o[8 + C + (size_t)(64 * q + lane) * 3 + a] = awp2[a];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float x = pdf_wave_sum_f32(sgp[a]), y = pdf_wave_sum_f32(sgpt[a]);
        if (lane == 0) { o[a] = x; o[3 + a] = y; }
    }
    if (lane == 0) { o[6] = 0.f; o[7] = 0.f; }
}

// ------------------------------------------------------------------------------------------------ launchers
template <typename KernelT>
static void launch(KernelT kernel, dim3 grid, size_t lds_floats, const LayerArgs &A, hipStream_t s) {
    const size_t lds = lds_floats * sizeof(float);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    kernel<<<grid, 64 * WPB, lds, s>>>(A);
}
#define PDF_FLM(KERNEL, GRID, LDS)                                                   \
    do {                                                                              \
        if (c == 64) launch(KERNEL<64>, GRID, LDS, A, s);                             \
        else if (c == 128) launch(KERNEL<128>, GRID, LDS, A, s);                      \
        else if (c == 256) launch(KERNEL<256>, GRID, LDS, A, s);                      \
        else launch(KERNEL<512>, GRID, LDS, A, s);                                    \
    } while (0)

void launch_p2(const LayerArgs &A, int c, int grid, hipStream_t s) { PDF_FLM(k_p2, dim3(grid, c / 64), (size_t)4 * c); }
void launch_p3(const LayerArgs &A, int c, bool stats, int grid, hipStream_t s) {
    const size_t lds = (size_t)6 * c + (size_t)csp_of(c) * (c + 4);
    if (stats) {
        if (c == 64) launch(k_p3<64, true>, dim3(grid), lds, A, s); else if (c == 128) launch(k_p3<128, true>, dim3(grid), lds, A, s); else if (c == 256) launch(k_p3<256, true>, dim3(grid), lds, A, s); else launch(k_p3<512, true>, dim3(grid), lds, A, s);
    } else {
        if (c == 64) launch(k_p3<64, false>, dim3(grid), lds, A, s); else if (c == 128) launch(k_p3<128, false>, dim3(grid), lds, A, s); else if (c == 256) launch(k_p3<256, false>, dim3(grid), lds, A, s); else launch(k_p3<512, false>, dim3(grid), lds, A, s);
    }
}
void launch_p4(const LayerArgs &A, int c, int grid, hipStream_t s) {
    // no per-wave partial rows in this pass: size the grid for occupancy (one point per wave and trip, latency-bound)
    long g = ((long)A.N + WPB - 1) / WPB;
    g = g > 2048 ? 2048 : (g < grid ? grid : g);
    PDF_FLM(k_p4, dim3((unsigned)g), (size_t)4 * c + w2_floats(c));
}
void launch_b1(const LayerArgs &A, int c, int grid, hipStream_t s) {
    PDF_FLM(k_b1, dim3(grid), (size_t)4 * c + w2_floats(c) + WPB * 16 * TS + 2 * WPB * 16 * (csp_of(c) + 4) + WPB * 16);
}
void launch_b2(const LayerArgs &A, int c, int grid, hipStream_t s) {
    PDF_FLM(k_b2, dim3(grid, c / 64), (size_t)6 * c + (size_t)csp_of(c) * 68 + WPB * 16 * (csp_of(c) + 4) + WPB * 16 * TS);
}
void launch_b3(const LayerArgs &A, int c, int grid, hipStream_t s) {
    PDF_FLM(k_b3, dim3(grid), (size_t)6 * c + w2_floats(c) + (size_t)csp_of(c) * 68 + WPB * 32 * TS + WPB * 64 + WPB * 16);
}
#undef PDF_FLM

}  // namespace flm
