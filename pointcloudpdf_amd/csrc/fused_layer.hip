// Fused PointTransformerLayer (vector attention with shared planes) for gfx950 -- forward passes.
// Reference semantics: pointcept/models/point_transformer/point_transformer_seg.py:45-78 with
// LayerNorm1d == BatchNorm1d over all n*nsample rows (utils.py:7-14), in train mode (batch statistics) or eval mode.
//
//   rel   = mask * (p[idx] - p[i])                         (grouping(), libs/pointops/functions/grouping.py:49-57)
//   t1    = rel Wp1^T + bp1 ; t1n = relu(BNp(t1))          (linear_p[0..2])
//   p_r   = t1n Wp2^T + bp2                                (linear_p[3])
//   r     = x_k[idx] - x_q[i] + p_r                        (:63-69)
//   h     = relu(BN1(r)) Ww1^T + bw1                       (linear_w[0..2])
//   z     = relu(BN2(h)) Ww2^T + bw2 ; w = softmax_j z     (linear_w[3..5], softmax over the neighbour dim)
//   out_i = sum_j (x_v[idx] + p_r) * w[.., ch mod C/8]     (:72-77)
//
// The reference materialises ~12 (n, ns, c) tensors per layer (205 MB each at levels 1/2).  Here nothing c-wide ever
// reaches HBM: every pass re-gathers the neighbour rows (they sit in L2 / Infinity Cache) and recomputes the cheap
// 3-channel branch; only h (n, ns, c/8) is stored.  Train-mode BatchNorm needs the statistics of t1, r and h over all
// rows before they can be normalised, hence four passes:
//   P1 stats(t1) -> P2 stats(r) -> P3 h + stats(h) -> P4 softmax + aggregation,   each followed by a tiny finalize.
// Mapping: one lane = one (point, neighbour) row, 64 rows per wave tile; neighbour rows are gathered with coalesced
// 16-byte loads (8 lanes per 128-byte row chunk) into a padded LDS tile and read back row-per-lane (conflict-free,
// stride 33); weights are wave-uniform and arrive through the scalar cache; per-channel statistics are reduced through
// the same LDS tile read column-wise.  HBM-bound by design (algorithmic bytes: q,k,v rows once + p + idx + out).
#include "fused_layer.h"
#include <algorithm>
#include <cstdlib>

namespace fl {

// LDS tile row strides (floats) are odd (33 / 65 / 17): row-per-lane and column reads are both bank-conflict-free.
__host__ __device__ constexpr int tile_stride(int c) { return c / 8 > 32 ? 65 : 33; }      // holds 32-channel chunks and CS-wide rows
__host__ __device__ constexpr int aux_stride(int c) { return c / 8 <= 16 ? 17 : c / 8 + 1; }
constexpr int MAX_BLOCKS = 512;   // default persistent grid of the forward passes (PDFOPS_PT_BLOCKS_FWD overrides)
static inline int env_blocks(const char *name, int dflt) {
    const char *v = getenv(name);
    const int x = v ? atoi(v) : 0;
    return x > 0 ? x : dflt;
}

struct WaveLds {
    float *tile;   // [64][TS]
    float *qtile;  // [8][TS]   rows of the tile's centre points
    int *rowid;    // [64]      gathered row index per tile row (-1 = zero row)
    float *aux;    // [64][as] second operand of the weight-gradient products (backward kernels only)
    int ts, as;    // row strides of tile / qtile and of aux
};

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Gather 32 channels [c0, c0+32) of the 64 rows listed in L.rowid into L.tile (coalesced: 8 lanes x 16 B per row).
__device__ __forceinline__ void stage_rows(const WaveLds &L, const float *__restrict__ table, int C, int c0, int lane) {
    const int sub = lane >> 3, col = (lane & 7) * 4;
    // all eight gathers in flight at once: clamped address + select (a load under `if (src >= 0)` gets its own branch and wait,
    // which serialised the eight round trips of every staged chunk)
    int src[8];
    float4 v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) src[t] = L.rowid[t * 8 + sub];
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = *reinterpret_cast<const float4 *>(table + (size_t)max(src[t], 0) * C + c0 + col);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        float *d = L.tile + (t * 8 + sub) * L.ts + col;
        const bool ok = src[t] >= 0;
        d[0] = ok ? v[t].x : 0.f; d[1] = ok ? v[t].y : 0.f; d[2] = ok ? v[t].z : 0.f; d[3] = ok ? v[t].w : 0.f;
    }
}

// Rows of the tile's PPT centre points (first point i0) into L.qtile.
template <int PPT>
__device__ __forceinline__ void stage_points(const WaveLds &L, const float *__restrict__ table, int C, int c0, int i0, int N, int lane) {
    if (lane < PPT * 8) {
        const int pt = lane >> 3, col = (lane & 7) * 4;
        const bool ok = i0 + pt < N;
        const float4 v = *reinterpret_cast<const float4 *>(table + (size_t)(ok ? i0 + pt : N - 1) * C + c0 + col);
        float *d = L.qtile + pt * L.ts + col;
        d[0] = ok ? v.x : 0.f; d[1] = ok ? v.y : 0.f; d[2] = ok ? v.z : 0.f; d[3] = ok ? v.w : 0.f;
    }
}

struct Row {
    int row, i, nb;
    bool valid;
    float rel[3];  // masked relative coordinates
    float t1[3];   // Linear(3,3) output (pre-BN)
};

template <int K>
__device__ __forceinline__ Row load_row(const LayerArgs &A, long tile, int lane) {
    Row R;
    const long row = tile * 64 + lane;
    R.valid = row < (long)A.N * K;
    R.row = (int)row;
    R.i = (int)(row / K);
    const long rowc = R.valid ? row : (long)A.N * K - 1;   // (clamped addresses + selects: no loads under a branch)
    const int nb = A.idx[rowc];
    R.nb = R.valid ? nb : -1;
    const size_t nbc = (size_t)max(R.nb, 0), ic = (size_t)(rowc / K);
    float pn[3], pi[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) { pn[b] = A.p[nbc * 3 + b]; pi[b] = A.p[ic * 3 + b]; }
#pragma unroll
    for (int b = 0; b < 3; ++b) R.rel[b] = R.nb >= 0 ? pn[b] - pi[b] : 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a)
        R.t1[a] = R.rel[0] * A.Wp1[a * 3 + 0] + R.rel[1] * A.Wp1[a * 3 + 1] + R.rel[2] * A.Wp1[a * 3 + 2] + A.bp1[a];
    return R;
}

__device__ __forceinline__ void bn_relu3(const LayerArgs &A, const float *t1, float *t1n) {
#pragma unroll
    for (int a = 0; a < 3; ++a) t1n[a] = fmaxf(t1[a] * A.sp[a] + A.tp[a], 0.f);
}

// p_r chunk: 32 channels [c0, c0+32) of Linear(3, C)
__device__ __forceinline__ void pos_chunk(const LayerArgs &A, const float *t1n, int c0, float *pr) {
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        cfloat_p w = A.Wp2 + (size_t)(c0 + c) * 3;
        pr[c] = t1n[0] * w[0] + t1n[1] * w[1] + t1n[2] * w[2] + A.bp2[c0 + c];
    }
}

// Column sums of the 64x32 tile: lane (ch = lane & 31, half = lane >> 5) adds its 32 rows into s / ss.
__device__ __forceinline__ void column_stats(const WaveLds &L, int lane, float &s, float &ss, int col0 = 0) {
    const int ch = col0 + (lane & 31), r0 = (lane >> 5) * 32;
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        const float v = L.tile[(r0 + r) * L.ts + ch];
        a += v;
        b += v * v;
    }
    s += a;
    ss += b;
}

__host__ __device__ constexpr int lds_floats_per_wave(int c, bool bwd) {
    return 64 * tile_stride(c) + 8 * tile_stride(c) + 64 + (bwd ? 64 * aux_stride(c) : 0);
}
template <bool BWD, int C>
__device__ __forceinline__ WaveLds carve_lds(float *base, int wave) {
    WaveLds L;
    L.ts = tile_stride(C);
    L.as = aux_stride(C);
    float *w = base + wave * lds_floats_per_wave(C, BWD);
    L.tile = w;
    L.qtile = w + 64 * L.ts;
    L.rowid = reinterpret_cast<int *>(w + 64 * L.ts + 8 * L.ts);
    L.aux = w + lds_floats_per_wave(C, false);
    return L;
}

// ------------------------------------------------------------------------------------------------ P1: stats of t1
template <int K>
__global__ __launch_bounds__(64 * WPB) void k_p1(LayerArgs A) {
    const int lane = threadIdx.x & 63;
    const long wave_g = (long)blockIdx.x * WPB + (threadIdx.x >> 6), nwaves = (long)gridDim.x * WPB;
    const long ntiles = ((long)A.N * K + 63) / 64;
    float s[3] = {0.f, 0.f, 0.f}, ss[3] = {0.f, 0.f, 0.f};
    for (long tile = wave_g; tile < ntiles; tile += nwaves) {
        const Row R = load_row<K>(A, tile, lane);
        if (R.valid) {
#pragma unroll
            for (int a = 0; a < 3; ++a) { s[a] += R.t1[a]; ss[a] += R.t1[a] * R.t1[a]; }
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { s[a] = pdf_wave_sum_f32(s[a]); ss[a] = pdf_wave_sum_f32(ss[a]); }
    __shared__ float stage[WPB * 8];
    block_row(stage, 8, [&](RowAcc o) {
        if (lane == 0) { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = ss[0]; o[4] = ss[1]; o[5] = ss[2]; }
    });
    store_row(stage, 6, A.partial + (size_t)blockIdx.x * 6);
}

// r chunk of one row: needs the staged xk chunk (tile) and xq chunk (qtile)
template <int K>
__device__ __forceinline__ void rqk_chunk(const WaveLds &L, int lane, const float *pr, float *r) {
    const float *xk = L.tile + lane * L.ts;
    const float *xq = L.qtile + (lane / K) * L.ts;
#pragma unroll
    for (int c = 0; c < 32; ++c) r[c] = (xk[c] - xq[c]) + pr[c];
}

// ------------------------------------------------------------------------------------------------ P2: stats of r
template <int C, int K>
__global__ __launch_bounds__(64 * WPB) void k_p2(LayerArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NCH = C / 32, PPT = 64 / K;
    const int lane = threadIdx.x & 63;
    const WaveLds L = carve_lds<false, C>(lds, threadIdx.x >> 6);
    const long wave_g = (long)blockIdx.x * WPB + (threadIdx.x >> 6), nwaves = (long)gridDim.x * WPB;
    const long ntiles = ((long)A.N * K + 63) / 64;
    float s[NCH], ss[NCH];
#pragma unroll
    for (int q = 0; q < NCH; ++q) { s[q] = 0.f; ss[q] = 0.f; }
    const BnP B = bnp_of(A, (long)A.N * K, blockIdx.x == 0 && threadIdx.x == 0);   // (from the geometry moments when the caller has them)
    const LayerArgs A0 = A;
    for (long tile = wave_g; tile < ntiles; tile += nwaves) {
        const LayerArgs A = fresh_consts(A0);   // (constants re-read at their uses: fused_layer.h, fresh_consts)
        const Row R = load_row<K>(A, tile, lane);
        float t1n[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) t1n[a] = fmaxf(R.t1[a] * B.sp[a] + B.tp[a], 0.f);
        L.rowid[lane] = R.nb;
        wave_sync();
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            stage_rows(L, A.xk, C, q * 32, lane);
            stage_points<PPT>(L, A.xq, C, q * 32, (int)(tile * PPT), A.N, lane);
            wave_sync();
            float pr[32], r[32];
            pos_chunk(A, t1n, q * 32, pr);
            rqk_chunk<K>(L, lane, pr, r);
            wave_sync();
#pragma unroll
            for (int c = 0; c < 32; ++c) L.tile[lane * L.ts + c] = R.valid ? r[c] : 0.f;
            wave_sync();
            column_stats(L, lane, s[q], ss[q]);
            wave_sync();
        }
    }
    // partial row layout: [sum(C) | sumsq(C)]
    block_row(lds, 2 * C, [&](RowAcc o) {
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const float a = s[q] + __shfl_xor(s[q], 32, 64), b = ss[q] + __shfl_xor(ss[q], 32, 64);
            if (lane < 32) { o[q * 32 + lane] = a; o[C + q * 32 + lane] = b; }
        }
    });
    store_row(lds, 2 * C, A.partial + (size_t)blockIdx.x * 2 * C);
}

// ------------------------------------------------------------------------------------------------ P3: h (+ stats)
template <int C, int K, bool STATS>
__global__ __launch_bounds__(64 * WPB) void k_p3(LayerArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NCH = C / 32, PPT = 64 / K, CS = C / 8;
    const int lane = threadIdx.x & 63;
    const WaveLds L = carve_lds<false, C>(lds, threadIdx.x >> 6);
    const long wave_g = (long)blockIdx.x * WPB + (threadIdx.x >> 6), nwaves = (long)gridDim.x * WPB;
    const long ntiles = ((long)A.N * K + 63) / 64;
    constexpr int NH = (CS + 31) / 32;
    float s[NH], ss[NH];      // lane (ch = lane & 31, half) accumulates channel hh*32 + ch of h
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) { s[hh] = 0.f; ss[hh] = 0.f; }
    const LayerArgs A0 = A;
    for (long tile = wave_g; tile < ntiles; tile += nwaves) {
        const LayerArgs A = fresh_consts(A0);
        const Row R = load_row<K>(A, tile, lane);
        float t1n[3];
        bn_relu3(A, R.t1, t1n);
        L.rowid[lane] = R.nb;
        wave_sync();
        float h[CS];
#pragma unroll
        for (int o = 0; o < CS; ++o) h[o] = A.bw1[o];
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            stage_rows(L, A.xk, C, q * 32, lane);
            stage_points<PPT>(L, A.xq, C, q * 32, (int)(tile * PPT), A.N, lane);
            wave_sync();
            float pr[32], r[32];
            pos_chunk(A, t1n, q * 32, pr);
            rqk_chunk<K>(L, lane, pr, r);
            wave_sync();
#pragma unroll
            for (int c = 0; c < 32; ++c) r[c] = fmaxf(r[c] * A.s1[q * 32 + c] + A.t1[q * 32 + c], 0.f);
#pragma unroll
            for (int o = 0; o < CS; ++o) {
                cfloat_p wrow = A.Ww1 + (size_t)o * C + q * 32;  // wave-uniform, contiguous -> batched s_load
#pragma unroll
                for (int c = 0; c < 32; ++c) h[o] += r[c] * wrow[c];
            }
        }
        if (R.valid) {
#pragma unroll
            for (int o = 0; o < CS; o += 4) st_u4(A.H, (size_t)R.row * CS + o, h[o], h[o + 1], h[o + 2], h[o + 3], A.bf16);
        }
        if (STATS) {
#pragma unroll
            for (int o = 0; o < CS; ++o) L.tile[lane * L.ts + o] = R.valid ? h[o] : 0.f;
            wave_sync();
#pragma unroll
            for (int hh = 0; hh < NH; ++hh)
                if (hh * 32 + (lane & 31) < CS) column_stats(L, lane, s[hh], ss[hh], hh * 32);
            wave_sync();
        }
    }
    if (STATS) {
        block_row(lds, 2 * CS, [&](RowAcc o) {
#pragma unroll
            for (int hh = 0; hh < NH; ++hh) {
                const float a = s[hh] + __shfl_xor(s[hh], 32, 64), b = ss[hh] + __shfl_xor(ss[hh], 32, 64);
                const int ch = hh * 32 + lane;
                if (lane < 32 && ch < CS) { o[ch] = a; o[CS + ch] = b; }
            }
        });
        store_row(lds, 2 * CS, A.partial + (size_t)blockIdx.x * 2 * CS);
    }
}

// butterfly reductions inside groups of K (8 or 16) consecutive lanes, every lane gets the result (DPP only)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xf, 0xf, false));
}
template <int K>
__device__ __forceinline__ float group_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));   // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_f<0x4E>(v));   // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_f<0x141>(v));  // row_half_mirror
    if (K == 16) v = fmaxf(v, dpp_f<0x140>(v));  // row_mirror
    return v;
}
template <int K>
__device__ __forceinline__ float group_sum(float v) {
    v += dpp_f<0xB1>(v);
    v += dpp_f<0x4E>(v);
    v += dpp_f<0x141>(v);
    if (K == 16) v += dpp_f<0x140>(v);
    return v;
}

// attention weights of one row from its stored h: w = softmax_j( relu(BN2(h)) Ww2^T + bw2 )
template <int CS>
__device__ __forceinline__ void load_hidden(const LayerArgs &A, const Row &R, float *h) {
    const size_t src = (size_t)(R.valid ? R.row : 0) * CS;
#pragma unroll
    for (int o = 0; o < CS; o += 4) {
        const float4 v = ld_u4(A.H, src + o, A.bf16);
        h[o] = v.x; h[o + 1] = v.y; h[o + 2] = v.z; h[o + 3] = v.w;
    }
}

template <int CS, int K>
__device__ __forceinline__ void attn_weights(const LayerArgs &A, const Row &R, float *w, float *h_out = nullptr, float *u_out = nullptr) {
    float u[CS];
    load_hidden<CS>(A, R, u);
#pragma unroll
    for (int o = 0; o < CS; ++o) {
        if (h_out) h_out[o] = u[o];
        u[o] = fmaxf(u[o] * A.s2[o] + A.t2[o], 0.f);
        if (u_out) u_out[o] = u[o];
    }
#pragma unroll
    for (int o = 0; o < CS; ++o) {
        float z = A.bw2[o];
#pragma unroll
        for (int c = 0; c < CS; ++c) z += u[c] * A.Ww2[o * CS + c];
        const float m = group_max<K>(z);
        const float e = __expf(z - m);
        w[o] = e / group_sum<K>(e);
    }
}

// ------------------------------------------------------------------------------------------------ P4: aggregation
// STATS: the block also leaves one partial row [sum out (C) | sum out^2 (C)] in A.partial -- the statistics of the Bottleneck's bn2, which
// reads this pass's output (point_transformer_seg.py:187): its own statistics pass over `out` (one launch per block) is not needed then.
template <int C, int K, bool STATS>
__global__ __launch_bounds__(64 * WPB) void k_p4(LayerArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NCH = C / 32, PPT = 64 / K, CS = C / 8;
    const int lane = threadIdx.x & 63;
    const WaveLds L = carve_lds<false, C>(lds, threadIdx.x >> 6);
    const long wave_g = (long)blockIdx.x * WPB + (threadIdx.x >> 6), nwaves = (long)gridDim.x * WPB;
    const long ntiles = ((long)A.N * K + 63) / 64;
    float so[STATS ? NCH : 1], sso[STATS ? NCH : 1];   // lane (half, ch): channel q * 32 + ch of the points this lane finishes
    if (STATS) {
#pragma unroll
        for (int q = 0; q < NCH; ++q) { so[q] = 0.f; sso[q] = 0.f; }
    }
    const LayerArgs A0 = A;
    for (long tile = wave_g; tile < ntiles; tile += nwaves) {
        const LayerArgs A = fresh_consts(A0);
        const Row R = load_row<K>(A, tile, lane);
        float t1n[3];
        bn_relu3(A, R.t1, t1n);
        float w[CS];
        attn_weights<CS, K>(A, R, w);
        L.rowid[lane] = R.nb;
        wave_sync();
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            stage_rows(L, A.xv, C, q * 32, lane);
            wave_sync();
            float pr[32], val[32];
            pos_chunk(A, t1n, q * 32, pr);
            const float *xv = L.tile + lane * L.ts;
#pragma unroll
            for (int c = 0; c < 32; ++c) val[c] = (xv[c] + pr[c]) * w[(q * 32 + c) % CS];
            wave_sync();
#pragma unroll
            for (int c = 0; c < 32; ++c) L.tile[lane * L.ts + c] = val[c];
            wave_sync();
            // out[(i0+pt), q*32+ch] = sum_j tile[pt*K + j][ch]
#pragma unroll
            for (int m = 0; m < PPT / 2; ++m) {
                const int pt = m * 2 + (lane >> 5), ch = lane & 31;
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < K; ++j) acc += L.tile[(pt * K + j) * L.ts + ch];
                const long i = tile * PPT + pt;
                if (i < A.N) {
                    A.out[(size_t)i * C + q * 32 + ch] = acc;
                    if (STATS) { so[STATS ? q : 0] += acc; sso[STATS ? q : 0] += acc * acc; }
                }
            }
            wave_sync();
        }
    }
    if (STATS) {
        block_row(lds, 2 * C, [&](RowAcc o) {
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                const float a = so[STATS ? q : 0] + __shfl_xor(so[STATS ? q : 0], 32, 64), b = sso[STATS ? q : 0] + __shfl_xor(sso[STATS ? q : 0], 32, 64);
                if (lane < 32) { o[q * 32 + lane] = a; o[C + q * 32 + lane] = b; }
            }
        });
        store_row(lds, 2 * C, A.partial + (size_t)blockIdx.x * 2 * C);
    }
}

// ================================================================================================ backward passes
// Given g_out (N,C) the chain is differentiated in four passes, mirroring the forward ones in reverse (each train-mode
// BatchNorm needs sum(dy) and sum(dy * xhat) over ALL rows before dx can be formed):
//   B1: d softmax / Ww2 / BN2-sums, scatter g_xv                  -> G2 = dL/d(BN2 out, masked by its ReLU)
//   B2: BN2 backward -> g_h, d Ww1, BN1-sums
//   B3: BN1 backward -> g_r: scatter g_xk, g_xq, g_pr -> d Wp2, G3 = dL/d(BNp out, masked), BNp-sums
//   B4: BNp backward -> d Wp1
// Parameter gradients and BatchNorm sums are accumulated per wave in registers (lanes <-> output elements, operands
// exchanged through LDS tiles), combined per workgroup in LDS (fl::block_row: fixed order), written as one partial row per
// workgroup and column-summed by k_colsum (deterministic).
constexpr int MAX_BLOCKS_BWD = 256;

// Store the 64x32 tile as channels [c0, c0+32) of rows row0 .. row0+63 of `table` (E x C): 2 rows x 32 consecutive channels per
// instruction = whole 128-byte lines, streamed (read once by the segmented gather that follows).
__device__ __forceinline__ void store_rows(const WaveLds &L, float *__restrict__ table, int C, int c0, long row0, long nrows, int lane, int bf16) {
    const int half = lane >> 5, col = lane & 31;
#pragma unroll 8
    for (int t = 0; t < 32; ++t) {
        const int row = 2 * t + half;
        if (row0 + row < nrows) st_u1_stream(table, (size_t)(row0 + row) * C + c0 + col, L.tile[row * L.ts + col], bf16);
    }
}

__device__ __forceinline__ float column_sum(const WaveLds &L, int lane) {  // lane (ch = lane & 31, half): 32 rows
    const int ch = lane & 31, r0 = (lane >> 5) * 32;
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) a += L.tile[(r0 + r) * L.ts + ch];
    return a;
}

// sum over the 64 tile rows of A[r][a] * B[r][b]; A in L.aux (stride AUX), B in L.tile (stride TS)
__device__ __forceinline__ float dot_aux_tile(const WaveLds &L, int a, int b) {
    float acc = 0.f;
#pragma unroll 8
    for (int r = 0; r < 64; ++r) acc += L.aux[r * L.as + a] * L.tile[r * L.ts + b];
    return acc;
}

// ---- B1 partial layout (floats): [sum g_y2 (CS) | sum g_y2*hhat (CS) | g_bw2 (CS) | g_Ww2 (CS*CS)]
template <int C> constexpr int b1_width() { return 3 * (C / 8) + (C / 8) * (C / 8); }
// ---- B2: [sum g_y1 (C) | sum g_y1*rhat (C) | g_bw1 (CS) | g_Ww1 (CS*C)]
template <int C> constexpr int b2_width() { return 2 * C + C / 8 + (C / 8) * C; }
// ---- B3: [sum g_yp (3) | sum g_yp*that (3) | pad 2 | g_bp2 (C) | g_Wp2 (C*3)]; the PARTIAL rows carry 16 more columns
//      [sum g_yp[a] * rel[b] (9) | pad 7] for the closed-form BNp backward (k_colsum_b3)
template <int C> constexpr int b3_width() { return 8 + C + 3 * C; }
template <int C> constexpr int b3_pwidth() { return b3_width<C>() + 16; }
// ---- B4: [g_bp1 (3) | g_Wp1 (9) | pad 4]
constexpr int b4_width() { return 16; }

template <int C, int K>
__global__ __launch_bounds__(64 * WPB) void k_b1(LayerArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NCH = C / 32, PPT = 64 / K, CS = C / 8, NW2 = (CS * CS + 63) / 64, W = b1_width<C>();
    const int lane = threadIdx.x & 63;
    const WaveLds L = carve_lds<true, C>(lds, threadIdx.x >> 6);
    const long wave_g = (long)blockIdx.x * WPB + (threadIdx.x >> 6), nwaves = (long)gridDim.x * WPB;
    const long ntiles = ((long)A.N * K + 63) / 64;
    float sg[CS], sgh[CS], sgz[CS], aw2[NW2];
#pragma unroll
    for (int o = 0; o < CS; ++o) { sg[o] = 0.f; sgh[o] = 0.f; sgz[o] = 0.f; }
#pragma unroll
    for (int m = 0; m < NW2; ++m) aw2[m] = 0.f;
    const LayerArgs A0 = A;
    for (long tile = wave_g; tile < ntiles; tile += nwaves) {
        const LayerArgs A = fresh_consts(A0);
        cfloat_p m2 = A.mean + 3 + C, r2 = A.rstd + 3 + C;
        const Row R = load_row<K>(A, tile, lane);
        float t1n[3];
        bn_relu3(A, R.t1, t1n);
        float w[CS], h[CS], u[CS], gw[CS];
        attn_weights<CS, K>(A, R, w, h, u);
#pragma unroll
        for (int o = 0; o < CS; ++o) gw[o] = 0.f;
        L.rowid[lane] = R.nb;
        wave_sync();
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            stage_rows(L, A.xv, C, q * 32, lane);
            stage_points<PPT>(L, A.gout, C, q * 32, (int)(tile * PPT), A.N, lane);
            wave_sync();
            float pr[32];
            pos_chunk(A, t1n, q * 32, pr);
            const float *xv = L.tile + lane * L.ts;
            const float *go = L.qtile + (lane / K) * L.ts;
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                const float g = R.valid ? go[c] : 0.f;
                gw[(q * 32 + c) % CS] += g * (xv[c] + pr[c]);
            }
            wave_sync();
        }
        if (R.valid) {   // softmax weights of the row: g_xv[nb] = sum over the inverse table of g_out[i] * w  (pdf_seg_sum_weighted)
#pragma unroll
            for (int o = 0; o < CS; o += 4) st_u4(A.Wsm, (size_t)R.row * CS + o, w[o], w[o + 1], w[o + 2], w[o + 3], A.bf16);
        }
        // softmax backward over the K neighbours, then Linear(CS,CS) and the ReLU of BN2
        float gz[CS], gy2[CS];
#pragma unroll
        for (int o = 0; o < CS; ++o) {
            const float dot = group_sum<K>(w[o] * gw[o]);
            gz[o] = R.valid ? w[o] * (gw[o] - dot) : 0.f;
        }
#pragma unroll
        for (int c = 0; c < CS; ++c) {
            float gu = 0.f;
#pragma unroll
            for (int o = 0; o < CS; ++o) gu += gz[o] * A.Ww2[o * CS + c];
            gy2[c] = u[c] > 0.f ? gu : 0.f;
            const float hhat = (h[c] - m2[c]) * r2[c];
            sg[c] += gy2[c];
            sgh[c] += gy2[c] * hhat;
            sgz[c] += gz[c];
        }
        if (R.valid) {
#pragma unroll
            for (int o = 0; o < CS; o += 4) st_u4(A.G2, (size_t)R.row * CS + o, gy2[o], gy2[o + 1], gy2[o + 2], gy2[o + 3], A.bf16);
        }
        // g_Ww2[o][c] += sum_rows gz[o] * u[c]
#pragma unroll
        for (int o = 0; o < CS; ++o) { L.aux[lane * L.as + o] = gz[o]; L.tile[lane * L.ts + o] = R.valid ? u[o] : 0.f; }
        wave_sync();
#pragma unroll
        for (int m = 0; m < NW2; ++m) {
            const int e = m * 64 + lane;
            if (e < CS * CS) aw2[m] += dot_aux_tile(L, e / CS, e % CS);
        }
        wave_sync();
    }
    auto emit = [&](RowAcc o) {
#pragma unroll
        for (int c = 0; c < CS; ++c) {
            const float a = pdf_wave_sum_f32(sg[c]), b = pdf_wave_sum_f32(sgh[c]), d = pdf_wave_sum_f32(sgz[c]);
            if (lane == 0) { o[c] = a; o[CS + c] = b; o[2 * CS + c] = d; }
        }
#pragma unroll
        for (int m = 0; m < NW2; ++m) {
            const int e = m * 64 + lane;
            if (e < CS * CS) o[3 * CS + e] = aw2[m];
        }
    };
    if constexpr (W <= lds_floats_per_wave(C, true)) block_row(lds, W, emit); else block_row_seq(lds, emit);   // (WPB rows fit the tiles' LDS?)
    store_row(lds, W, A.partial + (size_t)blockIdx.x * W);
}

// g_h of one row from the stored G2 / H rows and the BN2-backward sums
template <int C>
__device__ __forceinline__ void hidden_grad(const LayerArgs &A, const Row &R, float *gh) {
    constexpr int CS = C / 8;
    cfloat_p m2 = A.mean + 3 + C, r2 = A.rstd + 3 + C;
    float h[CS];
    load_hidden<CS>(A, R, h);
    const size_t src = (size_t)(R.valid ? R.row : 0) * CS;
#pragma unroll
    for (int o = 0; o < CS; o += 4) {
        const float4 v = ld_u4(A.G2, src + o, A.bf16);
        gh[o] = v.x; gh[o + 1] = v.y; gh[o + 2] = v.z; gh[o + 3] = v.w;
    }
#pragma unroll
    for (int o = 0; o < CS; ++o) {
        const float hhat = (h[o] - m2[o]) * r2[o];
        const float g = A.s2[o] * (gh[o] - A.sums[o] * A.inv_rows - hhat * A.sums[CS + o] * A.inv_rows);
        gh[o] = R.valid ? g : 0.f;
    }
}

// r, y1 = BN1(r) and g_y1 = (Ww1^T g_h) masked by the ReLU, for one 32-channel chunk (xk / xq staged)
template <int C, int K>
__device__ __forceinline__ void chunk_r_gy1(const LayerArgs &A, const WaveLds &L, int lane, int q, const float *t1n,
                                            const float *gh, float *r, float *gy1) {
    constexpr int CS = C / 8;
    float pr[32];
    pos_chunk(A, t1n, q * 32, pr);
    rqk_chunk<K>(L, lane, pr, r);
#pragma unroll
    for (int c = 0; c < 32; ++c) gy1[c] = 0.f;
#pragma unroll
    for (int o = 0; o < CS; ++o) {
        cfloat_p wrow = A.Ww1 + (size_t)o * C + q * 32;
#pragma unroll
        for (int c = 0; c < 32; ++c) gy1[c] += gh[o] * wrow[c];
    }
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        const float y1 = r[c] * A.s1[q * 32 + c] + A.t1[q * 32 + c];
        if (!(y1 > 0.f)) gy1[c] = 0.f;
    }
}

template <int C, int K>
__global__ __launch_bounds__(64 * WPB) void k_b2(LayerArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NCH = C / 32, PPT = 64 / K, CS = C / 8, W = b2_width<C>();
    const int lane = threadIdx.x & 63;
    const WaveLds L = carve_lds<true, C>(lds, threadIdx.x >> 6);
    const long wave_g = (long)blockIdx.x * WPB + (threadIdx.x >> 6), nwaves = (long)gridDim.x * WPB;
    const long ntiles = ((long)A.N * K + 63) / 64;
    constexpr int NH = (CS + 31) / 32;
    float sg[NCH], sgr[NCH], sgh[NH], aw1[NCH][CS / 2];
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) sgh[hh] = 0.f;
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
        sg[q] = 0.f; sgr[q] = 0.f;
#pragma unroll
        for (int m = 0; m < CS / 2; ++m) aw1[q][m] = 0.f;
    }
    cfloat_p m1 = A.mean + 3, r1 = A.rstd + 3;   // (this pass keeps its hoisted constants: with fresh_consts it needs 138 VGPRs -- 3 waves per SIMD instead
    for (long tile = wave_g; tile < ntiles; tile += nwaves) {   // of 4, 133 -> 136 us; held to 128 it spills to scratch: 166 us)
        const Row R = load_row<K>(A, tile, lane);
        float t1n[3];
        bn_relu3(A, R.t1, t1n);
        float gh[CS];
        hidden_grad<C>(A, R, gh);
        L.rowid[lane] = R.nb;
#pragma unroll
        for (int o = 0; o < CS; ++o) L.aux[lane * L.as + o] = gh[o];
        wave_sync();
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) {  // g_bw1: column sums of the g_h tile
            const int ch = hh * 32 + (lane & 31), r0 = (lane >> 5) * 32;
            if (ch < CS) {
                float a = 0.f;
#pragma unroll
                for (int r = 0; r < 32; ++r) a += L.aux[(r0 + r) * L.as + ch];
                sgh[hh] += a;
            }
        }
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            stage_rows(L, A.xk, C, q * 32, lane);
            stage_points<PPT>(L, A.xq, C, q * 32, (int)(tile * PPT), A.N, lane);
            wave_sync();
            float r[32], gy1[32];
            chunk_r_gy1<C, K>(A, L, lane, q, t1n, gh, r, gy1);
            wave_sync();
            // v1 = relu(BN1(r)) -> tile, g_Ww1[o][q*32+c] += sum_rows g_h[o] * v1[c]
#pragma unroll
            for (int c = 0; c < 32; ++c) L.tile[lane * L.ts + c] = R.valid ? fmaxf(r[c] * A.s1[q * 32 + c] + A.t1[q * 32 + c], 0.f) : 0.f;
            wave_sync();
#pragma unroll
            for (int m = 0; m < CS / 2; ++m) aw1[q][m] += dot_aux_tile(L, 2 * m + (lane >> 5), lane & 31);
            wave_sync();
#pragma unroll
            for (int c = 0; c < 32; ++c) L.tile[lane * L.ts + c] = gy1[c];
            wave_sync();
            sg[q] += column_sum(L, lane);
            wave_sync();
#pragma unroll
            for (int c = 0; c < 32; ++c) L.tile[lane * L.ts + c] = gy1[c] * ((r[c] - m1[q * 32 + c]) * r1[q * 32 + c]);
            wave_sync();
            sgr[q] += column_sum(L, lane);
            wave_sync();
        }
    }
    auto emit = [&](RowAcc o) {
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const float a = sg[q] + __shfl_xor(sg[q], 32, 64), b = sgr[q] + __shfl_xor(sgr[q], 32, 64);
            if (lane < 32) { o[q * 32 + lane] = a; o[C + q * 32 + lane] = b; }
#pragma unroll
            for (int m = 0; m < CS / 2; ++m) o[2 * C + CS + (size_t)(2 * m + (lane >> 5)) * C + q * 32 + (lane & 31)] = aw1[q][m];
        }
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) {
            const float hsum = sgh[hh] + __shfl_xor(sgh[hh], 32, 64);
            if (lane < 32 && hh * 32 + lane < CS) o[2 * C + hh * 32 + lane] = hsum;
        }
    };
    if constexpr (W <= lds_floats_per_wave(C, true)) block_row(lds, W, emit); else block_row_seq(lds, emit);   // (WPB rows fit the tiles' LDS?)
    store_row(lds, W, A.partial + (size_t)blockIdx.x * W);
}

template <int C, int K>
__global__ __launch_bounds__(64 * WPB) void k_b3(LayerArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NCH = C / 32, PPT = 64 / K, CS = C / 8, W = b3_pwidth<C>();
    const int lane = threadIdx.x & 63;
    const WaveLds L = carve_lds<true, C>(lds, threadIdx.x >> 6);
    const long wave_g = (long)blockIdx.x * WPB + (threadIdx.x >> 6), nwaves = (long)gridDim.x * WPB;
    const long ntiles = ((long)A.N * K + 63) / 64;
    float sgp[3] = {0.f, 0.f, 0.f}, sgpt[3] = {0.f, 0.f, 0.f}, sbp2[NCH], awp2[NCH][2], sgr[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) sgr[e] = 0.f;
#pragma unroll
    for (int q = 0; q < NCH; ++q) { sbp2[q] = 0.f; awp2[q][0] = 0.f; awp2[q][1] = 0.f; }
    const LayerArgs A0 = A;
    for (long tile = wave_g; tile < ntiles; tile += nwaves) {
        const LayerArgs A = fresh_consts(A0);   // (constants re-read at their uses: see fresh_consts)
        cfloat_p m1 = A.mean + 3, r1 = A.rstd + 3, mp = A.mean, rp = A.rstd;
        cfloat_p sum_gy1 = A.sums, sum_gy1r = A.sums + C;  // column sums of B2 (A.sums points at B2's result here)
        const Row R = load_row<K>(A, tile, lane);
        float t1n[3];
        bn_relu3(A, R.t1, t1n);
        float gh[CS], w[CS];
        {   // hidden_grad needs B1's sums (A.sums2); A.sums holds B2's in this pass
            LayerArgs A2 = A;
            A2.sums = A.sums2;
            hidden_grad<C>(A2, R, gh);
        }
        {   // softmax weights of the row: stored by B1 (Wsm) -- read back instead of redoing BN2, Linear(CS, CS) and the softmax
            const size_t src = (size_t)(R.valid ? R.row : 0) * CS;
#pragma unroll
            for (int o = 0; o < CS; o += 4) {
                const float4 v = ld_u4(A.Wsm, src + o, A.bf16);
                w[o] = v.x; w[o + 1] = v.y; w[o + 2] = v.z; w[o + 3] = v.w;
            }
        }
        L.rowid[lane] = R.nb;
        // t1n tile for the Wp2 gradient
        L.aux[lane * L.as + 0] = R.valid ? t1n[0] : 0.f;
        L.aux[lane * L.as + 1] = R.valid ? t1n[1] : 0.f;
        L.aux[lane * L.as + 2] = R.valid ? t1n[2] : 0.f;
        float gt1n[3] = {0.f, 0.f, 0.f};
        wave_sync();
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            stage_rows(L, A.xk, C, q * 32, lane);
            stage_points<PPT>(L, A.xq, C, q * 32, (int)(tile * PPT), A.N, lane);
            wave_sync();
            float r[32], g[32];
            chunk_r_gy1<C, K>(A, L, lane, q, t1n, gh, r, g);
            // BN1 backward: g_r = s1 * (g_y1 - mean(g_y1) - rhat * mean(g_y1 * rhat))
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                const int ch = q * 32 + c;
                const float rhat = (r[c] - m1[ch]) * r1[ch];
                const float gr = A.s1[ch] * (g[c] - sum_gy1[ch] * A.inv_rows - rhat * sum_gy1r[ch] * A.inv_rows);
                g[c] = R.valid ? gr : 0.f;
            }
            wave_sync();
            stage_points<PPT>(L, A.gout, C, q * 32, (int)(tile * PPT), A.N, lane);  // qtile: xq no longer needed
#pragma unroll
            for (int c = 0; c < 32; ++c) L.tile[lane * L.ts + c] = g[c];
            wave_sync();
            store_rows(L, A.GR, C, q * 32, tile * 64, (long)A.N * K, lane, A.bf16);   // g_xk[nb] = segmented sum of these rows (pdf_seg_sum_rows)
            // g_xq[i] = - sum_j g_r
#pragma unroll
            for (int m = 0; m < PPT / 2; ++m) {
                const int pt = m * 2 + (lane >> 5), ch = lane & 31;
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < K; ++j) acc += L.tile[(pt * K + j) * L.ts + ch];
                const long i = tile * PPT + pt;
                if (i < A.N) A.gxq[(size_t)i * C + q * 32 + ch] = -acc;
            }
            // g_pr = g_r + g_out * w   (the aggregation's share of p_r)
            const float *go = L.qtile + (lane / K) * L.ts;
#pragma unroll
            for (int c = 0; c < 32; ++c) g[c] += R.valid ? go[c] * w[(q * 32 + c) % CS] : 0.f;
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                cfloat_p wp = A.Wp2 + (size_t)(q * 32 + c) * 3;
                gt1n[0] += g[c] * wp[0]; gt1n[1] += g[c] * wp[1]; gt1n[2] += g[c] * wp[2];
            }
            wave_sync();
#pragma unroll
            for (int c = 0; c < 32; ++c) L.tile[lane * L.ts + c] = g[c];
            wave_sync();
            sbp2[q] += column_sum(L, lane);
            // g_Wp2[q*32+c][a] += sum_rows g_pr[c] * t1n[a]    (96 outputs per chunk: e = c*3 + a)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int e = m * 64 + lane;
                if (e < 96) {
                    const int c = e / 3, a = e % 3;
                    float acc = 0.f;
#pragma unroll 8
                    for (int rr = 0; rr < 64; ++rr) acc += L.tile[rr * L.ts + c] * L.aux[rr * L.as + a];
                    awp2[q][m] += acc;
                }
            }
            wave_sync();
        }
        float gyp[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            gyp[a] = (R.valid && t1n[a] > 0.f) ? gt1n[a] : 0.f;
            sgp[a] += gyp[a];
            sgpt[a] += gyp[a] * ((R.t1[a] - mp[a]) * rp[a]);
            sgr[3 * a + 0] += gyp[a] * R.rel[0]; sgr[3 * a + 1] += gyp[a] * R.rel[1]; sgr[3 * a + 2] += gyp[a] * R.rel[2];
        }
        if (R.valid) {
            float *dst = A.G3 + (size_t)R.row * 3;
            dst[0] = gyp[0]; dst[1] = gyp[1]; dst[2] = gyp[2];
        }
    }
    auto emit = [&](RowAcc o) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float x = pdf_wave_sum_f32(sgp[a]), y = pdf_wave_sum_f32(sgpt[a]);
            if (lane == 0) { o[a] = x; o[3 + a] = y; }
        }
        if (lane == 0) { o[6] = 0.f; o[7] = 0.f; }
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const float a = sbp2[q] + __shfl_xor(sbp2[q], 32, 64);
            if (lane < 32) o[8 + q * 32 + lane] = a;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int e = m * 64 + lane;
                if (e < 96) o[8 + C + (size_t)q * 96 + e] = awp2[q][m];  // == [(q*32 + c)*3 + a]
            }
        }
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            const float x = pdf_wave_sum_f32(sgr[e]);
            if (lane == 0) o[b3_width<C>() + e] = x;
        }
        if (lane == 0) { for (int e = 9; e < 16; ++e) o[b3_width<C>() + e] = 0.f; }
    };
    if constexpr (W <= lds_floats_per_wave(C, true)) block_row(lds, W, emit); else block_row_seq(lds, emit);   // (WPB rows fit the tiles' LDS?)
    store_row(lds, W, A.partial + (size_t)blockIdx.x * W);
}

template <int K>
__global__ __launch_bounds__(64 * WPB) void k_b4(LayerArgs A) {
    const int lane = threadIdx.x & 63;
    const long wave_g = (long)blockIdx.x * WPB + (threadIdx.x >> 6), nwaves = (long)gridDim.x * WPB;
    const long ntiles = ((long)A.N * K + 63) / 64;
    float acc[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) acc[e] = 0.f;
    for (long tile = wave_g; tile < ntiles; tile += nwaves) {
        const Row R = load_row<K>(A, tile, lane);
        if (!R.valid) continue;
        const float *g3 = A.G3 + (size_t)R.row * 3;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float that = (R.t1[a] - A.mean[a]) * A.rstd[a];
            const float gt1 = A.sp[a] * (g3[a] - A.sums[a] * A.inv_rows - that * A.sums[3 + a] * A.inv_rows);
            acc[a] += gt1;
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[3 + a * 3 + b] += gt1 * R.rel[b];
        }
    }
    __shared__ float stage[WPB * 16];
    block_row(stage, 16, [&](RowAcc o) {
#pragma unroll
        for (int e = 0; e < 12; ++e) {
            const float v = pdf_wave_sum_f32(acc[e]);
            if (lane == 0) o[e] = v;
        }
        if (lane == 0) { o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 0.f; }
    });
    store_row(stage, b4_width(), A.partial + (size_t)blockIdx.x * b4_width());
}

// out[col] = sum_rows partial[row][col]   (double accumulation, deterministic)
// block = 16 columns x RED_RL row-lanes; every row-lane sums a strided subset of the rows.  These kernels are pure latency (a few
// hundred KB read by 4 .. 64 workgroups, ~280 launches per step): every lane keeps RED_B independent loads in flight, so ~2,000 partial
// rows cost two memory round trips instead of eight.
constexpr int RED_RL = 16, RED_THREADS = 16 * RED_RL, RED_B = 16;
__device__ __forceinline__ double strided_sum(const float *__restrict__ src, size_t stride, int rl, int rows) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int r = rl; r < rows; r += RED_B * RED_RL) {
        float v[RED_B];
        // (clamped address + select afterwards: a load under a condition gets its own branch and wait, which serialises the batch)
#pragma unroll
        for (int t = 0; t < RED_B; ++t) v[t] = src[(size_t)min(r + t * RED_RL, rows - 1) * stride];
#pragma unroll
        for (int t = 0; t < RED_B; ++t) v[t] = (r + t * RED_RL < rows) ? v[t] : 0.f;
#pragma unroll
        for (int t = 0; t < RED_B; t += 4) { s0 += (double)v[t]; s1 += (double)v[t + 1]; s2 += (double)v[t + 2]; s3 += (double)v[t + 3]; }
    }
    return (s0 + s1) + (s2 + s3);
}

// sum over the 4 row-lanes of a wave (lanes l, l ^ 16, l ^ 32, l ^ 48 share a column)
__device__ __forceinline__ double wave_rows_sum(double v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
constexpr int RED_WAVES = RED_THREADS / 64;

// Closed-form backward of the geometry branch's BatchNorm + Linear(3, 3) (point_transformer_seg.py:27-29 differentiated): the extra block
// of k_colsum's launch over B3's rows.  With the row sums  s_a = sum g_yp[a],  u_a = sum g_yp[a] that[a],  G_ab = sum g_yp[a] rel[b]  (B3),
// the coordinate sums S, M of the kNN table (geom_moments.hip; R = all (point, neighbour) rows) and the saved BNp statistics:
//     g_t1[a] = sp_a (g_yp[a] - s_a / R - that[a] u_a / R)                      (BatchNorm backward per row)
//     d Wp1[a][b] = sum g_t1[a] rel[b] = sp_a (G_ab - s_a S_b / R - u_a T_ab / R),   T_ab = sum that[a] rel[b] = rstd_a (sum_c Wp1[a][c] M_cb + (bp1_a - mean_a) S_b)
//     d bp1[a]    = sum g_t1[a]        = sp_a (- u_a / R) sum that[a],               sum that[a] = rstd_a (sum_c Wp1[a][c] S_c + R (bp1_a - mean_a))
// -- what fl::k_b4 + its column sum computed with one more pass over all rows (2 launches per layer; rounds 1-3).  out4: [d bp1 (3) | d Wp1 (9) | 0 x 4].
struct BnpClosed {
    const double *mom;            // [S (3) | Mxx Mxy Mxz Myy Myz Mzz]; nullptr: no closed form (the caller runs B4)
    cfloat_p Wp1, bp1, sp, mean, rstd;
    double rows;
    int col_g;                    // first column of G in the partial rows
    float *out4;
};
__global__ __launch_bounds__(RED_THREADS) void k_colsum(const float *__restrict__ partial, int rows, int width, int stride, float *__restrict__ out,
                                                        BnpClosed bc) {
    __shared__ double red[RED_WAVES][17];
    if (blockIdx.x * 16 >= (unsigned)width) {   // the closed-form block (one past the column blocks; only launched with bc.mom)
        __shared__ double col[16];
        // 15 columns: s (0..2), u (3..5), G (col_g .. col_g + 8): lane = (column, row-lane) as below, one pass
        const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
        const int c = cl < 6 ? cl : bc.col_g + (cl - 6);
        const double v = wave_rows_sum(cl < 15 ? strided_sum(partial + c, (size_t)stride, rl, rows) : 0.0);
        if ((threadIdx.x & 63) < 16) red[threadIdx.x >> 6][cl] = v;
        __syncthreads();
        if (threadIdx.x < 16) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < RED_WAVES; ++k) s += red[k][threadIdx.x];
            col[threadIdx.x] = s;
        }
        __syncthreads();
        if (threadIdx.x < 12) {
            const double S[3] = {bc.mom[0], bc.mom[1], bc.mom[2]};
            const double M[3][3] = {{bc.mom[3], bc.mom[4], bc.mom[5]}, {bc.mom[4], bc.mom[6], bc.mom[7]}, {bc.mom[5], bc.mom[7], bc.mom[8]}};
            const int a = threadIdx.x < 3 ? (int)threadIdx.x : ((int)threadIdx.x - 3) / 3, b = threadIdx.x < 3 ? 0 : ((int)threadIdx.x - 3) % 3;
            const double w[3] = {(double)bc.Wp1[a * 3], (double)bc.Wp1[a * 3 + 1], (double)bc.Wp1[a * 3 + 2]};
            const double d = (double)bc.bp1[a] - (double)bc.mean[a], rs = (double)bc.rstd[a], sp = (double)bc.sp[a], R = bc.rows;
            const double s_a = col[a], u_a = col[3 + a];
            double r;
            if (threadIdx.x < 3) {
                const double sum_that = rs * (w[0] * S[0] + w[1] * S[1] + w[2] * S[2] + R * d);
                r = sp * (-(u_a / R) * sum_that);
            } else {
                const double T = rs * (w[0] * M[0][b] + w[1] * M[1][b] + w[2] * M[2][b] + d * S[b]);
                r = sp * (col[6 + 3 * a + b] - s_a * S[b] / R - u_a * T / R);
            }
            bc.out4[threadIdx.x] = (float)r;
        } else if (threadIdx.x < 16) {
            bc.out4[threadIdx.x] = 0.f;
        }
        return;
    }
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + cl;
    const double v = wave_rows_sum(col < width ? strided_sum(partial + col, (size_t)stride, rl, rows) : 0.0);
    if ((threadIdx.x & 63) < 16) red[threadIdx.x >> 6][cl] = v;
    __syncthreads();
    if (rl != 0 || col >= width) return;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < RED_WAVES; ++k) s += red[k][cl];
    out[col] = (float)s;
}

// ------------------------------------------------------------------------------------------------ BN finalize
// partial: [rows][2*nch] (sum | sumsq).  Train: batch mean/var -> scale/shift (+ running-stat update, saved mean/rstd).
// block = 16 channels x RED_RL row-lanes: every row-lane sums a strided subset of the partial rows in double, LDS combine.
__global__ __launch_bounds__(RED_THREADS) void k_bn_finalize(const float *__restrict__ partial, int rows, int nch, double count,
                              const float *__restrict__ gamma, const float *__restrict__ beta, float eps, float momentum,
                              float *__restrict__ running_mean, float *__restrict__ running_var,
                              float *__restrict__ scale, float *__restrict__ shift, float *__restrict__ mean_out,
                              float *__restrict__ rstd_out) {
    __shared__ double red[2][RED_WAVES][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int ch = blockIdx.x * 16 + cl;
    const double a = wave_rows_sum(ch < nch ? strided_sum(partial + ch, 2 * (size_t)nch, rl, rows) : 0.0);
    const double b = wave_rows_sum(ch < nch ? strided_sum(partial + nch + ch, 2 * (size_t)nch, rl, rows) : 0.0);
    if ((threadIdx.x & 63) < 16) { red[0][threadIdx.x >> 6][cl] = a; red[1][threadIdx.x >> 6][cl] = b; }
    __syncthreads();
    if (rl != 0 || ch >= nch) return;
    double s = 0.0, ss = 0.0;
#pragma unroll
    for (int k = 0; k < RED_WAVES; ++k) { s += red[0][k][cl]; ss += red[1][k][cl]; }
    const double mean = s / count;
    double var = ss / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[ch] * rstd;
    scale[ch] = sc;
    shift[ch] = beta[ch] - (float)mean * sc;
    mean_out[ch] = (float)mean;
    rstd_out[ch] = rstd;
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * (float)mean;
        running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)unbiased;
    }
}

// Eval: scale/shift from running statistics.
__global__ void k_bn_eval(int nch, const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                          const float *__restrict__ running_mean, const float *__restrict__ running_var,
                          float *__restrict__ scale, float *__restrict__ shift, float *__restrict__ mean_out,
                          float *__restrict__ rstd_out) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= nch) return;
    const float rstd = 1.0f / sqrtf(running_var[ch] + eps);
    const float sc = gamma[ch] * rstd;
    scale[ch] = sc;
    shift[ch] = beta[ch] - running_mean[ch] * sc;
    if (mean_out) { mean_out[ch] = running_mean[ch]; rstd_out[ch] = rstd; }
}

// host-side launchers shared with pointwise.hip (kernels stay private to this translation unit)
void launch_bn_finalize(const float *partial, int rows, int nch, double count, const float *gamma, const float *beta, float eps,
                        float momentum, float *running_mean, float *running_var, float *scale, float *shift, float *mean_out,
                        float *rstd_out, hipStream_t s) {
    k_bn_finalize<<<pdf_divup(nch, 16), RED_THREADS, 0, s>>>(partial, rows, nch, count, gamma, beta, eps, momentum, running_mean, running_var,
                                                     scale, shift, mean_out, rstd_out);
}
void launch_bn_eval(int nch, const float *gamma, const float *beta, float eps, const float *running_mean, const float *running_var,
                    float *scale, float *shift, float *mean_out, float *rstd_out, hipStream_t s) {
    k_bn_eval<<<pdf_divup(nch, 64), 64, 0, s>>>(nch, gamma, beta, eps, running_mean, running_var, scale, shift, mean_out, rstd_out);
}
void launch_colsum(const float *partial, int rows, int width, float *out, hipStream_t s) {
    BnpClosed none;
    none.mom = nullptr; none.out4 = nullptr; none.rows = 0.0; none.col_g = 0;
    none.Wp1 = none.bp1 = none.sp = none.mean = none.rstd = as_const(nullptr);
    k_colsum<<<pdf_divup(width, 16), RED_THREADS, 0, s>>>(partial, rows, width, width, out, none);
}

template <typename KernelT>
static inline void allow_lds(KernelT kernel, size_t lds) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

// matrix-core passes work on one point (16 rows) per wave and trip: size their grids by points, not by 64-row tiles
// (level 5 has 780 points: 49 workgroups by tiles, 195 by points)
static inline int grid_for_points(long n, int cap) {
    static const int env_cap = env_blocks("PDFOPS_PT_BLOCKS_MFMA", 0);
    if (env_cap > 0) cap = env_cap;
    long g = (n + WPB - 1) / WPB;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// Per-pass grids of the matrix-core passes (one point per wave and trip).  The passes differ in what a resident block costs: P3 / B3 / B1
// write short partial rows (2 C/8, 8 + 4 C, 3 C/8 + (C/8)^2 floats per workgroup) and gain from more waves per SIMD; B2's partial row holds a
// (C/8) x C weight-gradient block, so its grid stays small.  PDFOPS_PT_CAP_<pass> overrides the default cap.
// P2 / B2 run one workgroup per (point block, 64-channel slab): at C >= 256 the slab dimension already fills the chip and every extra
// point block pays the slab staging and the partial-row epilogue again (B2 at 780 points x 512 channels: 90 us with 195 point blocks,
// 45 us with 32).
enum Pass { P3 = 0, B1 = 1, B2 = 2, B3 = 3, P2 = 4 };
static inline int pass_grid(Pass pass, long n, int dflt_cap) {
    static const int env_cap[5] = {env_blocks("PDFOPS_PT_CAP_P3", 0), env_blocks("PDFOPS_PT_CAP_B1", 0), env_blocks("PDFOPS_PT_CAP_B2", 0),
                                   env_blocks("PDFOPS_PT_CAP_B3", 0), env_blocks("PDFOPS_PT_CAP_P2", 0)};   // (read once per process, like the other PDFOPS_PT_* caps)
    return grid_for_points(n, env_cap[pass] > 0 ? env_cap[pass] : dflt_cap);
}

static inline int grid_for_tiles(long ntiles) {
    static const int cap = env_blocks("PDFOPS_PT_BLOCKS_FWD", MAX_BLOCKS);
    long g = (ntiles + WPB - 1) / WPB;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace fl

// ================================================================================================ C ABI
// Workspace contract (caller-owned floats): `bn` = 2*(3 + C + C/8) scale/shift values laid out
// [sp(3) tp(3) s1(C) t1(C) s2(CS) t2(CS)], `saved` = [mean: p(3) 1(C) 2(CS)][rstd: same] (train mode, for backward),
// `partial` = pdf_pt_layer_partial_floats(N, K, C) floats of scratch.
extern "C" long pdf_pt_layer_partial_floats(int n, int nsample, int c) {
    const long ntiles = ((long)n * nsample + 63) / 64;
    const int g = flm::supported(nsample, c) ? fl::grid_for_points(n, fl::MAX_BLOCKS) : fl::grid_for_tiles(ntiles);
    return (long)g * fl::WPB * 2 * (c > 3 ? c : 3);
}

extern "C" int pdf_pt_layer_supported(int nsample, int c) {
    return ((nsample == 8 || nsample == 16) && (c == 32 || c == 64 || c == 128)) || (nsample == 16 && c == 256) ||
           (c == 512 && flm::supported(nsample, c));   // C = 512: matrix-core passes only (fused_layer_mfma.hip)
}

namespace fl {

template <int C, int K>
int forward_impl(LayerArgs A, int training, float eps, float momentum, const float *const *bn_params, float *const *bn_buffers,
                 float *bn, float *saved, int *out_stat_rows, hipStream_t s) {
    constexpr int CS = C / 8;
    const long rows = (long)A.N * K;
    const long ntiles = (rows + 63) / 64;
    const int grid = flm::supported(K, C) ? grid_for_points(A.N, MAX_BLOCKS) : grid_for_tiles(ntiles);
    const size_t lds = (size_t)WPB * lds_floats_per_wave(C, false) * sizeof(float);
    float *sp = bn, *tp = bn + 3, *s1 = bn + 6, *t1 = bn + 6 + C, *s2 = bn + 6 + 2 * C, *t2 = bn + 6 + 2 * C + CS;
    A.sp = as_const(sp); A.tp = as_const(tp); A.s1 = as_const(s1); A.t1 = as_const(t1); A.s2 = as_const(s2); A.t2 = as_const(t2);
    // bn_params: gamma_p, beta_p, gamma_1, beta_1, gamma_2, beta_2 ; bn_buffers: rm_p, rv_p, rm_1, rv_1, rm_2, rv_2
    if (!training) {
        k_bn_eval<<<1, 64, 0, s>>>(3, bn_params[0], bn_params[1], eps, bn_buffers[0], bn_buffers[1], sp, tp, nullptr, nullptr);
        k_bn_eval<<<pdf_divup(C, 64), 64, 0, s>>>(C, bn_params[2], bn_params[3], eps, bn_buffers[2], bn_buffers[3], s1, t1, nullptr, nullptr);
        k_bn_eval<<<1, 64, 0, s>>>(CS, bn_params[4], bn_params[5], eps, bn_buffers[4], bn_buffers[5], s2, t2, nullptr, nullptr);
        allow_lds(k_p3<C, K, false>, lds);
        allow_lds(k_p4<C, K, false>, lds);
        if (flm::supported(K, C)) flm::launch_p3(A, C, false, grid, s);
        else k_p3<C, K, false><<<grid, 64 * WPB, lds, s>>>(A);
        if (flm::supported(K, C)) flm::launch_p4(A, C, grid, false, s); else k_p4<C, K, false><<<grid, 64 * WPB, lds, s>>>(A);
        if (out_stat_rows) *out_stat_rows = 0;   // (eval: bn2 uses its running statistics)
        return pdf_launch_status();
    }
    constexpr int T = 3 + C + CS;   // saved = [mean: p(3) | 1(C) | 2(CS)] [rstd: same order]  (the layout the backward kernels index)
    float *mp = saved, *m1 = saved + 3, *m2 = saved + 3 + C, *rp = saved + T, *r1 = saved + T + 3, *r2 = saved + T + 3 + C;
    allow_lds(k_p2<C, K>, lds);
    allow_lds(k_p3<C, K, true>, lds);
    allow_lds(k_p4<C, K, false>, lds);
    allow_lds(k_p4<C, K, true>, lds);
    if (A.mom) {   // BNp from the geometry moments: P2 computes the coefficients in its prologue, its block 0 stores them
        A.gam_p = as_const(bn_params[0]); A.bet_p = as_const(bn_params[1]);
        A.bnp_coef = sp; A.bnp_saved = mp; A.bnp_T = T; A.bnp_rm = bn_buffers[0]; A.bnp_rv = bn_buffers[1];
        A.bnp_eps = eps; A.bnp_momentum = momentum;
    } else {
        k_p1<K><<<grid, 64 * WPB, 0, s>>>(A);
        k_bn_finalize<<<1, RED_THREADS, 0, s>>>(A.partial, grid, 3, (double)rows, bn_params[0], bn_params[1], eps, momentum, bn_buffers[0], bn_buffers[1], sp, tp, mp, rp);
    }
    int g2 = grid;
    if (flm::supported(K, C)) { g2 = pass_grid(P2, A.N, C == 512 ? 64 : (C == 256 ? 128 : grid)); flm::launch_p2(A, C, g2, s); }
    else k_p2<C, K><<<grid, 64 * WPB, lds, s>>>(A);
    A.mom = nullptr;   // (P3 / P4 read the coefficients block 0 of P2 stored)
    k_bn_finalize<<<pdf_divup(C, 16), RED_THREADS, 0, s>>>(A.partial, g2, C, (double)rows, bn_params[2], bn_params[3], eps, momentum, bn_buffers[2], bn_buffers[3], s1, t1, m1, r1);
    int nw3 = grid;   // (P3's partial rows are 2 C/8 floats: a larger grid than P2's fits the same scratch)
    if (flm::supported(K, C)) { const int g3 = pass_grid(P3, A.N, C <= 256 ? 2 * MAX_BLOCKS : MAX_BLOCKS); nw3 = g3; flm::launch_p3(A, C, true, g3, s); }
    else k_p3<C, K, true><<<grid, 64 * WPB, lds, s>>>(A);
    k_bn_finalize<<<pdf_divup(CS, 16), RED_THREADS, 0, s>>>(A.partial, nw3, CS, (double)rows, bn_params[4], bn_params[5], eps, momentum, bn_buffers[4], bn_buffers[5], s2, t2, m2, r2);
    // P4 (+ the statistics of its output for the Bottleneck's bn2 when the caller wants them: *out_stat_rows partial rows of 2 C floats in
    // A.partial -- P3's rows there are consumed, the finalizer above ran in stream order)
    const bool ostats = out_stat_rows != nullptr;
    if (flm::supported(K, C)) {
        flm::launch_p4(A, C, grid, ostats, s);
        if (ostats) *out_stat_rows = flm::p4_grid(A.N, grid);
    } else {
        if (ostats) k_p4<C, K, true><<<grid, 64 * WPB, lds, s>>>(A); else k_p4<C, K, false><<<grid, 64 * WPB, lds, s>>>(A);
        if (ostats) *out_stat_rows = grid;
    }
    return pdf_launch_status();
}

}  // namespace fl

namespace fl {

// two blocks per CU pay off where the passes are latency-bound on many small tiles (C <= 64: 1.0 -> 0.78 ms at level 1);
// at C >= 128 the partial rows (weight-gradient blocks) cost more than the extra waves bring
static inline int grid_for_tiles_bwd(long ntiles, int c) {
    static const int env_cap = env_blocks("PDFOPS_PT_BLOCKS_BWD", 0);
    const int cap = env_cap > 0 ? env_cap : (c <= 64 ? 2 * MAX_BLOCKS_BWD : MAX_BLOCKS_BWD);
    long g = (ntiles + WPB - 1) / WPB;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

template <int C, int K>
int backward_impl(LayerArgs A, float *sums, const int *inv_off, const int *inv_entry, int entry_base, const int *gather_order, hipStream_t s) {
    constexpr int CS = C / 8;
    const long rows = (long)A.N * K;
    const long ntiles = (rows + 63) / 64;
    const int grid = flm::supported(K, C) ? grid_for_points(A.N, C <= 128 ? 2 * MAX_BLOCKS_BWD : MAX_BLOCKS_BWD) : grid_for_tiles_bwd(ntiles, C);
    const size_t lds = (size_t)WPB * lds_floats_per_wave(C, true) * sizeof(float);
    A.inv_rows = (float)(1.0 / (double)rows);
    // sums layout (floats): [S1: b1_width | S2: b2_width | S3: b3_width | S4: 16 | X: 2C + 2CS scratch for B3]
    float *S1 = sums, *S2 = S1 + b1_width<C>(), *S3 = S2 + b2_width<C>(), *S4 = S3 + b3_width<C>();
    allow_lds(k_b1<C, K>, lds);
    allow_lds(k_b2<C, K>, lds);
    allow_lds(k_b3<C, K>, lds);
    const bool mfma = flm::supported(K, C);
    // (the scratch is sized for B2's rows at `grid`; B1 / B3 rows are shorter, so more waves fit)
    int g1 = mfma ? pass_grid(B1, A.N, C == 64 ? 768 : grid) : grid, g3 = mfma ? pass_grid(B3, A.N, C == 256 ? 2 * grid : (C == 64 ? 4 * grid : grid)) : grid;
    g1 = (int)std::min<long>(g1, (long)grid * b2_width<C>() / b1_width<C>());
    g3 = (int)std::min<long>(g3, (long)grid * b2_width<C>() / b3_pwidth<C>());
    BnpClosed none;
    none.mom = nullptr; none.out4 = nullptr; none.rows = 0.0; none.col_g = 0;
    none.Wp1 = none.bp1 = none.sp = none.mean = none.rstd = as_const(nullptr);
    if (mfma) flm::launch_b1(A, C, g1, s); else k_b1<C, K><<<grid, 64 * WPB, lds, s>>>(A);
    k_colsum<<<pdf_divup(b1_width<C>(), 16), RED_THREADS, 0, s>>>(A.partial, g1, b1_width<C>(), b1_width<C>(), S1, none);
    // g_xv[nb, c] = sum over the entries (i, j) with idx[i, j] == nb of g_out[i, c] * w[i, j, c mod C/8]   (no atomics, fixed order)
    int rc = pdf_seg_sum_weighted_x(A.N, C, K, CS, A.gout, A.Wsm, A.bf16, inv_off, inv_entry, entry_base, gather_order, A.gxv, s);   // (destinations in Morton order: the g_out rows they share hit L2)
    if (rc != PDF_OK) return rc;
    A.sums = as_const(S1);
    const int g2 = mfma ? std::min(C == 64 ? 2 * grid : grid, pass_grid(B2, A.N, C == 512 ? 32 : (C == 256 ? 128 : (C == 64 ? 768 : grid)))) : grid;   // (C = 64: one slab, three workgroups per CU since round 6 -- the scratch holds grid * WPB rows)
    // level 1 on the matrix-core form: 112 / 154 registers -> 4 / 3 waves per SIMD, so 4 / 3 workgroups per CU instead of the row-per-lane
    // passes' 2 (the scratch holds grid * WPB rows of B2's width: room for them)
    static const int l1_mul = [] { const char *v = getenv("PDFOPS_L1_GRID_MUL"); return v ? atoi(v) : 1; }();   // (measured 1 .. 4: 15.32 - 15.39 ms per step, no trend)
    const long pair_blocks = ((long)(A.N + 1) / 2 + WPB - 1) / WPB;
    const int g2l = (int)std::min<long>((long)l1_mul * grid, pair_blocks), g3l = (int)std::min<long>((long)(l1_mul * 3 / 2 > 0 ? l1_mul * 3 / 2 : 1) * grid, pair_blocks);
    const bool l1 = !mfma && flm::supported_l1(K, C);
    if (mfma) flm::launch_b2(A, C, g2, s);
    else if (l1) flm::launch_b2_l1(A, g2l, s);   // (the partial rows of fl::k_b2<32, 8>)
    else k_b2<C, K><<<grid, 64 * WPB, lds, s>>>(A);
    k_colsum<<<pdf_divup(b2_width<C>(), 16), RED_THREADS, 0, s>>>(A.partial, l1 ? g2l : g2, b2_width<C>(), b2_width<C>(), S2, none);
    A.sums = as_const(S2);   // B3: BN1-backward terms from B2, BN2-backward terms from B1
    A.sums2 = as_const(S1);
    const int gb3 = mfma ? g3 : (int)std::min<long>(grid, (long)grid * b2_width<C>() / b3_pwidth<C>());   // (fl::k_b3's rows are 16 wider than S3)
    static const bool l1_b3 = [] { const char *v = getenv("PDFOPS_PT_L1_B3"); return !(v && v[0] == '0'); }();   // (A/B: 0 = fl::k_b3 at level 1)
    const bool l1b3 = l1 && l1_b3;
    static const bool closed_on = [] { const char *v = getenv("PDFOPS_BNP_CLOSED"); return !(v && v[0] == '0'); }();
    const bool closed = closed_on && A.mom != nullptr;
    // slab form (fused_layer_slab.hip: wave = 64-channel slab): needs the closed-form geometry backward (it does not write G3)
    const bool slab3 = mfma && closed && fls::enabled(C);
    if (slab3) { g3 = fls::b3_grid(A.N, C, (int)std::min<long>(1 << 20, (long)grid * b2_width<C>() / b3_pwidth<C>())); fls::launch_b3(A, C, g3, s); }
    else if (mfma) flm::launch_b3(A, C, g3, s);
    else if (l1b3) flm::launch_b3_l1(A, g3l, s);   // (the partial rows of fl::k_b3<32, 8>)
    else k_b3<C, K><<<gb3, 64 * WPB, lds, s>>>(A);
    // column sums of B3's rows -> S3; with the kNN table's coordinate sums at hand one more block of the SAME launch finishes the geometry
    // branch's BatchNorm + Linear(3, 3) backward in closed form (S4): no B4 pass, no second reducer (PDFOPS_BNP_CLOSED=0: B4, as before)
    BnpClosed bc = none;
    if (closed) {
        bc.mom = A.mom; bc.Wp1 = A.Wp1; bc.bp1 = A.bp1; bc.sp = A.sp; bc.mean = A.mean; bc.rstd = A.rstd; bc.rows = (double)rows;
        bc.col_g = b3_width<C>(); bc.out4 = S4;
    }
    k_colsum<<<pdf_divup(b3_width<C>(), 16) + (closed ? 1 : 0), RED_THREADS, 0, s>>>(A.partial, mfma ? g3 : (l1b3 ? g3l : gb3), b3_width<C>(), b3_pwidth<C>(), S3, bc);
    rc = pdf_seg_sum_rows_x(A.N, C, A.GR, C, A.bf16, inv_off, inv_entry, entry_base, 1.0f, A.gxk, s);   // g_xk[nb] = sum of the g_r rows that gathered nb
    if (rc != PDF_OK) return rc;
    if (!closed) {
        A.sums = as_const(S3);
        k_b4<K><<<grid, 64 * WPB, 0, s>>>(A);
        k_colsum<<<1, RED_THREADS, 0, s>>>(A.partial, grid, b4_width(), b4_width(), S4, none);
    }
    return pdf_launch_status();
}

}  // namespace fl

extern "C" long pdf_pt_layer_bwd_partial_floats(int n, int nsample, int c) {
    const long ntiles = ((long)n * nsample + 63) / 64;
    const long w = 2L * c + c / 8 + (long)(c / 8) * c;  // widest pass (B2)
    const int g = flm::supported(nsample, c) ? fl::grid_for_points(n, c <= 128 ? 2 * fl::MAX_BLOCKS_BWD : fl::MAX_BLOCKS_BWD) : fl::grid_for_tiles_bwd(ntiles, c);
    return (long)g * fl::WPB * w;
}

// floats in the `sums` result buffer and the offsets of its four sections [S1 | S2 | S3 | S4 | scratch]
extern "C" long pdf_pt_layer_bwd_sums_floats(int c) {
    const int cs = c / 8;
    return (long)(3 * cs + cs * cs) + (2L * c + cs + (long)cs * c) + (8 + 4L * c) + 16 + (2L * c + 2 * cs);
}

// Backward of one PointTransformerLayer (training mode).  Inputs as the forward plus g_out, the forward's bn / saved /
// H buffers and the INVERSE of the kNN table (inv_off (N+1), inv_entry, entry_base: csrc/seg_gather.hip); outputs: gxq, gxk, gxv
// (N,C) written (no pre-zeroing: the scatters run as segmented gathers), G2 (N*K*C/8), G3 (N*K*3), Wsm (N*K*C/8) and GR (N*K*C)
// scratch, partial scratch (pdf_pt_layer_bwd_partial_floats), sums (pdf_pt_layer_bwd_sums_floats) which receives
//   S1 = [sum g_y2 | sum g_y2*hhat | g_bw2 | g_Ww2]          -> d beta2, d gamma2, d bw2, d Ww2
//   S2 = [sum g_y1 | sum g_y1*rhat | g_bw1 | g_Ww1]          -> d beta1, d gamma1, d bw1, d Ww1
//   S3 = [sum g_yp(3) | sum g_yp*that(3) | 0 0 | g_bp2 | g_Wp2] -> d betap, d gammap, d bp2, d Wp2
//   S4 = [g_bp1(3) | g_Wp1(9) | 0 x4]
extern "C" int pdf_pt_layer_backward(int n, int nsample, int c, const float *xq, const float *xk, const float *xv,
                                     const float *p, const int *idx, const float *const *weights, const float *bn,
                                     const float *saved, const float *H, const float *gout, float *gxq, float *gxk,
                                     float *gxv, float *G2, float *G3, float *Wsm, float *GR, const int *inv_off, const int *inv_entry,
                                     int entry_base, float *partial, float *sums, int storage_bf16, const int *order, const double *moments,
                                     void *stream) {
    if (n < 1 || !xq || !xk || !xv || !p || !idx || !weights || !bn || !saved || !H || !gout || !gxq || !gxk || !gxv ||
        !G2 || !G3 || !Wsm || !GR || !inv_off || !inv_entry || !partial || !sums)
        return PDF_ERR_BAD_ARG;
    if (!pdf_pt_layer_supported(nsample, c)) return PDF_ERR_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(weights[4]) & 15) return PDF_ERR_BAD_ARG;   // Ww1 is staged in 16-byte pieces
    const int cs = c / 8;
    fl::LayerArgs A;
    A.N = n; A.xq = xq; A.xk = xk; A.xv = xv; A.p = p; A.idx = idx;
    using fl::as_const;
    A.Wp1 = as_const(weights[0]); A.bp1 = as_const(weights[1]); A.Wp2 = as_const(weights[2]); A.bp2 = as_const(weights[3]);
    A.Ww1 = as_const(weights[4]); A.bw1 = as_const(weights[5]); A.Ww2 = as_const(weights[6]); A.bw2 = as_const(weights[7]);
    A.sp = as_const(bn); A.tp = as_const(bn + 3); A.s1 = as_const(bn + 6); A.t1 = as_const(bn + 6 + c);
    A.s2 = as_const(bn + 6 + 2 * c); A.t2 = as_const(bn + 6 + 2 * c + cs);
    A.H = const_cast<float *>(H); A.out = nullptr; A.partial = partial;
    A.mom = moments;   // the kNN table's coordinate sums (or null): closed-form backward of the geometry branch's first layers
    A.gout = gout; A.G2 = G2; A.G3 = G3; A.gxq = gxq; A.gxk = gxk; A.gxv = gxv; A.Wsm = Wsm; A.GR = GR; A.bf16 = storage_bf16 & 1; A.chunked = (storage_bf16 >> 1) & 1; A.order = (storage_bf16 & 4) ? order : nullptr; A.sums = as_const(nullptr); A.sums2 = as_const(nullptr);
    hipStream_t s = static_cast<hipStream_t>(stream);
    A.mean = as_const(saved); A.rstd = as_const(saved + (3 + c + cs));   // forward's layout: [mean p|1|2][rstd p|1|2]
#define PDF_BWD(C_, K_) return fl::backward_impl<C_, K_>(A, sums, inv_off, inv_entry, entry_base, order, s)
    if (nsample == 8) {
        if (c == 32) PDF_BWD(32, 8);
        if (c == 64) PDF_BWD(64, 8);
        PDF_BWD(128, 8);
    } else {
        if (c == 32) PDF_BWD(32, 16);
        if (c == 64) PDF_BWD(64, 16);
        if (c == 128) PDF_BWD(128, 16);
        if (c == 256) PDF_BWD(256, 16);
        PDF_BWD(512, 16);
    }
#undef PDF_BWD
}

// Forward of one PointTransformerLayer.  weights: Wp1,bp1,Wp2,bp2,Ww1,bw1,Ww2,bw2 ; bn_params: gamma/beta of the three
// norms ; bn_buffers: running mean/var of the three norms (updated in train mode; may hold nulls to skip the update).
extern "C" int pdf_pt_layer_forward_m(int n, int nsample, int c, const float *xq, const float *xk, const float *xv,
                                      const float *p, const int *idx, const float *const *weights,
                                      const float *const *bn_params, float *const *bn_buffers, int training, float eps,
                                      float momentum, float *bn, float *saved, float *H, float *partial, float *out,
                                      int storage_bf16, const int *order, const double *moments, int *out_stat_rows, void *stream);
extern "C" int pdf_pt_layer_forward(int n, int nsample, int c, const float *xq, const float *xk, const float *xv,
                                    const float *p, const int *idx, const float *const *weights,
                                    const float *const *bn_params, float *const *bn_buffers, int training, float eps,
                                    float momentum, float *bn, float *saved, float *H, float *partial, float *out,
                                    int storage_bf16, const int *order, void *stream) {
    return pdf_pt_layer_forward_m(n, nsample, c, xq, xk, xv, p, idx, weights, bn_params, bn_buffers, training, eps, momentum, bn, saved, H,
                                  partial, out, storage_bf16, order, nullptr, nullptr, stream);
}
extern "C" int pdf_pt_layer_forward_m(int n, int nsample, int c, const float *xq, const float *xk, const float *xv,
                                      const float *p, const int *idx, const float *const *weights,
                                      const float *const *bn_params, float *const *bn_buffers, int training, float eps,
                                      float momentum, float *bn, float *saved, float *H, float *partial, float *out,
                                      int storage_bf16, const int *order, const double *moments, int *out_stat_rows, void *stream) {
    if (n < 1 || !xq || !xk || !xv || !p || !idx || !weights || !bn_params || !bn_buffers || !bn || !H || !partial || !out)
        return PDF_ERR_BAD_ARG;
    if (!pdf_pt_layer_supported(nsample, c)) return PDF_ERR_UNSUPPORTED;
    if (training && !saved) return PDF_ERR_BAD_ARG;
    if (reinterpret_cast<uintptr_t>(weights[4]) & 15) return PDF_ERR_BAD_ARG;   // Ww1 is staged in 16-byte pieces
    fl::LayerArgs A;
    A.N = n; A.xq = xq; A.xk = xk; A.xv = xv; A.p = p; A.idx = idx;
    using fl::as_const;
    A.Wp1 = as_const(weights[0]); A.bp1 = as_const(weights[1]); A.Wp2 = as_const(weights[2]); A.bp2 = as_const(weights[3]);
    A.Ww1 = as_const(weights[4]); A.bw1 = as_const(weights[5]); A.Ww2 = as_const(weights[6]); A.bw2 = as_const(weights[7]);
    A.H = H; A.out = out; A.partial = partial; A.bf16 = storage_bf16 & 1; A.chunked = (storage_bf16 >> 1) & 1; A.order = (storage_bf16 & 4) ? order : nullptr;
    A.mom = training ? moments : nullptr;
    hipStream_t s = static_cast<hipStream_t>(stream);
#define PDF_FWD(C_, K_) return fl::forward_impl<C_, K_>(A, training, eps, momentum, bn_params, bn_buffers, bn, saved, out_stat_rows, s)
    if (nsample == 8) {
        if (c == 32) PDF_FWD(32, 8);
        if (c == 64) PDF_FWD(64, 8);
        PDF_FWD(128, 8);
    } else {
        if (c == 32) PDF_FWD(32, 16);
        if (c == 64) PDF_FWD(64, 16);
        if (c == 128) PDF_FWD(128, 16);
        if (c == 256) PDF_FWD(256, 16);
        PDF_FWD(512, 16);
    }
#undef PDF_FWD
}
