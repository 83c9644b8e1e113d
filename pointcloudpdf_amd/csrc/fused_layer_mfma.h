// Device helpers shared by the matrix-core passes of the fused PointTransformerLayer (fused_layer_mfma.hip: one wave = one point;
// fused_layer_slab.hip: one wave = one 64-channel slab of one point).  Lane layout, LDS staging and the per-point building blocks
// (geometry branch, attention weights, BN2 backward) are documented at their definitions.
#pragma once
#include "fused_layer.h"
#include <algorithm>
#include <cstdlib>

namespace flm {

using fl::cfloat_p;
using fl::LayerArgs;
using fl::WPB;
using fl::RowAcc;
using fl::block_row;
using fl::store_row;
typedef float f32x4 __attribute__((ext_vector_type(4)));

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

// ---- staging (global -> LDS).  Every helper issues ALL of a thread's global loads before its first LDS store: written as the
// obvious `for (e ...) lds[e] = global[e]` loop the compiler keeps load -> store pairs in order and a block pays one L2 round trip
// per iteration -- 16 per 64-channel Ww1 slab, ~10 us per chunk at C >= 256, which was most of k_b3 / k_b2 at levels 4 and 5
// (3,124 / 780 points: one to three points per wave, the staging is not amortised).
constexpr int NT = 64 * WPB;
// (loads use a clamped index and the select happens afterwards: a load under a condition gets its own branch and wait)
// dst[r * WS + c] = r < rows_valid ? src[r * src_stride + c] : 0 for r < ROWS, c < COLS (float4 pieces; COLS, WS, src_stride % 4 == 0,
// 16-byte aligned src -- checked at the C entry points)
template <int ROWS, int COLS, int WS>
__device__ __forceinline__ void stage_rows(float *dst, const float *src, int src_stride, int rows_valid) {
    constexpr int Q = COLS / 4, N4 = ROWS * Q, PER = (N4 + NT - 1) / NT, B = PER > 8 ? 8 : PER;
#pragma unroll 1
    for (int t0 = 0; t0 < PER; t0 += B) {
        f32x4 v[B];
#pragma unroll
        for (int t = 0; t < B; ++t) {
            const int e = threadIdx.x + NT * (t0 + t), r = min(e / Q, rows_valid - 1), c4 = e % Q;
            v[t] = ld4(src + (size_t)r * src_stride + 4 * c4);
        }
#pragma unroll
        for (int t = 0; t < B; ++t) {
            const int e = threadIdx.x + NT * (t0 + t), r = e / Q, c4 = e % Q;
            if (e < N4) st4(dst + r * WS + 4 * c4, r < rows_valid ? v[t] : zero4());
        }
    }
}
template <int C>
__device__ __forceinline__ void stage_consts(float *cst, const LayerArgs &A, bool with_bn1) {
    constexpr int P3 = (3 * C + NT - 1) / NT, P1 = (C + NT - 1) / NT;
    float a[P3], b[P1], c[P1], d[P1];
#pragma unroll
    for (int t = 0; t < P3; ++t) a[t] = gp(A.Wp2)[min((int)threadIdx.x + NT * t, 3 * C - 1)];
#pragma unroll
    for (int t = 0; t < P1; ++t) {
        const int e = min((int)threadIdx.x + NT * t, C - 1);
        b[t] = gp(A.bp2)[e];
        if (with_bn1) { c[t] = gp(A.s1)[e]; d[t] = gp(A.t1)[e]; }   // (block-uniform condition)
    }
#pragma unroll
    for (int t = 0; t < P3; ++t) { const int e = threadIdx.x + NT * t; if (e < 3 * C) cst[e] = a[t]; }
#pragma unroll
    for (int t = 0; t < P1; ++t) {
        const int e = threadIdx.x + NT * t;
        if (e < C) { cst[3 * C + e] = b[t]; if (with_bn1) { cst[4 * C + e] = c[t]; cst[5 * C + e] = d[t]; } }
    }
}
template <int C>
__device__ __forceinline__ void stage_w2(float *w2, const LayerArgs &A) {   // zero-padded copy of Ww2, row stride CSP + 4
    constexpr int CS = C / 8, CSP = csp_of(C), N = CSP * (CSP + 4), PER = (N + NT - 1) / NT;
    float v[PER];
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int e = threadIdx.x + NT * t, o = min(e / (CSP + 4), CS - 1), u = min(e % (CSP + 4), CS - 1);
        v[t] = gp(A.Ww2)[o * CS + u];
    }
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int e = threadIdx.x + NT * t, o = e / (CSP + 4), u = e % (CSP + 4);
        if (e < N) w2[e] = (o < CS && u < CS) ? v[t] : 0.f;
    }
}
// guarded float4 of a CS-long per-unit array (zero where the lane's units do not exist)
__device__ __forceinline__ f32x4 ldu(const float *p, int o, bool hv) { return hv ? ld4(p + o) : zero4(); }
__device__ __forceinline__ f32x4 xchg32(f32x4 v) {
    return f32x4{__shfl_xor(v[0], 32, 64), __shfl_xor(v[1], 32, 64), __shfl_xor(v[2], 32, 64), __shfl_xor(v[3], 32, 64)};
}

// ================================================================================================ trip structure
// Memory-level parallelism.  One wave works on one point (16 neighbour rows) per trip.  Written naively every global load sits next
// to its use and the compiler waits for each in turn (`global_load; s_waitcnt vmcnt(0)`: ~25 serialised L2 round trips per trip --
// that, not bandwidth or arithmetic, was what these passes cost).  So every kernel below
//   * issues ALL global loads of a trip (neighbour index of the NEXT trip, coordinates, H / G2 / Wsm rows, the first group of channel
//     loads) into registers behind a scheduling barrier, then computes; long channel loops run in groups of four 16-channel blocks with
//     the next group's loads in flight during the current group's arithmetic (double-buffered registers);
//   * keeps per-channel / per-unit constants (BatchNorm coefficients, statistics, backward sums) in LDS, staged once per block;
//   * never loads under a condition (a load inside a branch is a wait): clamped address + select;
//   * takes the storage type of the row arrays as a template parameter (a runtime `if (bf16)` around a load splits the basic block).
template <bool BF>
__device__ __forceinline__ f32x4 ld_row4(const float *base, size_t idx) {   // 4 consecutive elements of a row array (fp32 / bfloat16 storage)
    if constexpr (BF) {
        const uint2 v = *reinterpret_cast<const uint2 *>(reinterpret_cast<const unsigned short *>(base) + idx);
        return f32x4{fl::bf2f(v.x & 0xffffu), fl::bf2f(v.x >> 16), fl::bf2f(v.y & 0xffffu), fl::bf2f(v.y >> 16)};
    } else {
        return ld4(base + idx);
    }
}
template <bool BF>
__device__ __forceinline__ void st_row4(float *base, size_t idx, f32x4 v) {
    if constexpr (BF) {
        uint2 o; o.x = fl::f2bf(v[0]) | (fl::f2bf(v[1]) << 16); o.y = fl::f2bf(v[2]) | (fl::f2bf(v[3]) << 16);
        *reinterpret_cast<uint2 *>(reinterpret_cast<unsigned short *>(base) + idx) = o;
    } else {
        st4(base + idx, v);
    }
}
__device__ __forceinline__ f32x4 sel4(bool c, f32x4 v) { return c ? v : zero4(); }

// geometry branch of the lane's row (3 channels, replicated over the 4 kq lanes): weights in SGPRs, coordinates preloaded
struct GeoW { float wp1[9], bp1[3], sp[3], tp[3]; };
struct Geo { float t1[3], t1n[3], rel[3]; };   // Linear(3,3) output (pre-BN), relu(BNp(t1)), the masked relative coordinates
__device__ __forceinline__ GeoW geo_weights(const LayerArgs &A) {
    GeoW G;
#pragma unroll
    for (int e = 0; e < 9; ++e) G.wp1[e] = A.Wp1[e];
#pragma unroll
    for (int a = 0; a < 3; ++a) { G.bp1[a] = A.bp1[a]; G.sp[a] = A.sp[a]; G.tp[a] = A.tp[a]; }
    return G;
}
__device__ __forceinline__ Geo geo_of(const GeoW &G, int nb, const float *pn, const float *pi) {
    Geo R;
    float *rel = R.rel;
#pragma unroll
    for (int b = 0; b < 3; ++b) rel[b] = nb >= 0 ? pn[b] - pi[b] : 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        R.t1[a] = rel[0] * G.wp1[a * 3 + 0] + rel[1] * G.wp1[a * 3 + 1] + rel[2] * G.wp1[a * 3 + 2] + G.bp1[a];
        R.t1n[a] = fmaxf(R.t1[a] * G.sp[a] + G.tp[a], 0.f);
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

// per-unit constants in LDS, CSP floats each (padding units = 0):
//   [0] s2 | [1] t2 | [2] bw2 | backward only (S != nullptr): [3] mean2 | [4] rstd2 | [5] S[0..CS) / rows | [6] S[CS..2CS) / rows
constexpr int U_S2 = 0, U_T2 = 1, U_BW2 = 2, U_MEAN = 3, U_RSTD = 4, U_SA = 5, U_SB = 6;
template <int C, bool BWD>   // (two instantiations on purpose: one body called with S == nullptr here and S != nullptr there crashes clang 22's CGSCC pipeline)
__device__ __forceinline__ void stage_units(float *ucst, const LayerArgs &A, const float *S) {
    constexpr int CS = C / 8, CSP = csp_of(C), N = (BWD ? 7 : 3) * CSP;
    for (int e = threadIdx.x; e < N; e += NT) {
        const int arr = e / CSP, u = min(e % CSP, CS - 1);
        float v;
        if constexpr (BWD) {
            const float *src = arr == U_S2 ? gp(A.s2) : arr == U_T2 ? gp(A.t2) : arr == U_BW2 ? gp(A.bw2)
                             : arr == U_MEAN ? gp(A.mean) + 3 + C : arr == U_RSTD ? gp(A.rstd) + 3 + C : arr == U_SA ? S : S + CS;
            v = src[u] * (arr >= U_SA ? A.inv_rows : 1.f);
        } else {
            const float *src = arr == U_S2 ? gp(A.s2) : arr == U_T2 ? gp(A.t2) : gp(A.bw2);
            v = src[u];
        }
        ucst[e] = e % CSP < CS ? v : 0.f;
    }
}
template <int C> __device__ __forceinline__ f32x4 unit4(const float *ucst, int arr, int ob, int kq) { return ld4(ucst + arr * csp_of(C) + 16 * ob + 4 * kq); }
// element offset of the lane's units of block ob inside a CS-long row, clamped into the row (invalid lanes re-read valid units, masked later)
template <int C> __device__ __forceinline__ int unit_off(int ob, int kq) { return min(16 * ob + 4 * kq, C / 8 - 4); }

// Attention branch of one point (16 rows): u = relu(BN2(h)), w = softmax over the rows of (u Ww2^T + bw2).
// Lane (row, kq) holds hidden units {16 ob + 4 kq + e}; the MFMA D fragment of z^T = Ww2 u^T has the same index set.
template <int C>
__device__ __forceinline__ void attn_weights(const float *ucst, const float *w2, int row, int kq, const f32x4 *h, f32x4 *u, f32x4 *w) {
    constexpr int CS = C / 8, NOB = nob_of(C), WS2 = csp_of(C) + 4;
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
        u[ob] = sel4(16 * ob + 4 * kq < CS, relu4(h[ob] * unit4<C>(ucst, U_S2, ob, kq) + unit4<C>(ucst, U_T2, ob, kq)));   // padding lanes: 0
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
        f32x4 z = zero4();
#pragma unroll
        for (int jo = 0; jo < NOB; ++jo) {
            const f32x4 a = ld4(w2 + (ob * 16 + row) * WS2 + 16 * jo + 4 * kq);   // A operand: Ww2[16 ob + (l & 15)][16 jo + 4 kq + e]
#pragma unroll
            for (int e = 0; e < 4; ++e) z = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], u[jo][e], z, 0, 0, 0);
        }
        z += unit4<C>(ucst, U_BW2, ob, kq);
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
// g_h of the lane's hidden units from the H / G2 fragments and the BN2-backward sums (BN2 backward)
template <int C>
__device__ __forceinline__ void hidden_grad(const float *ucst, int kq, const f32x4 *h, const f32x4 *g2, f32x4 *gh) {
    constexpr int CS = C / 8, NOB = nob_of(C);
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
        const f32x4 hhat = (h[ob] - unit4<C>(ucst, U_MEAN, ob, kq)) * unit4<C>(ucst, U_RSTD, ob, kq);
        gh[ob] = sel4(16 * ob + 4 * kq < CS,
                      unit4<C>(ucst, U_S2, ob, kq) * (g2[ob] - unit4<C>(ucst, U_SA, ob, kq) - hhat * unit4<C>(ucst, U_SB, ob, kq)));   // padding lanes: 0
    }
}

}  // namespace flm

