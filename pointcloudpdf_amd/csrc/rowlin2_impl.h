// Streaming per-point Linear layers for channel widths 32..512 (second-generation kernels behind pdf_rowlin_*).
//
// The Bottleneck's Linear layers (point_transformer_seg.py:184-192, 45-78: linear1, linear_q/k/v, linear3) are skinny GEMMs:
// N = 10^5..10^3 rows, 32..512 channels, i.e. 16..256 FLOP per byte of activations -- HBM / latency bound, never
// MFMA bound.  The kernels therefore stream the activation rows exactly once and keep the whole weight slab in registers:
//
//   forward / input gradient (k_fwd):  Y_out[n, o] = sum_in sum_k f(X_in[n, k]) Wt_in/out(k, o) + bias[o]
//     one wave = 16 rows per trip; v_mfma_f32_16x16x4_f32 with A = W (rows = output channels) and B = X^T (cols = rows n):
//     lane (n = l & 15, kq = l >> 4) loads float4 X[n][16 j + 4 kq ..+4] (64 contiguous bytes per row and instruction) and
//     ends up with float4 Y[n][o0 + 4 (l >> 4) ..+4]: both sides are 16-byte accesses, no LDS, no barrier.
//     The reduction index is permuted (k = 16 j + 4 kq + c at MFMA step (j, c)), which a dot product does not mind.
//     f = identity or relu(x * scale[k] + shift[k]) (the BatchNorm + ReLU in front of the layer), optional epilogue =
//     per-column sum / sum of squares of Y for the BatchNorm behind it.  Up to three inputs (dX = sum G_i W_i) or three
//     outputs (q, k, v from one read of the activations).
//   weight gradient (k_wg):  dW[o, k] += sum_n G[n, o] f(X[n, k]),  db[o] += sum_n G[n, o]
//     reduction index = rows: lane (i = l & 15, nq = l >> 4) loads VW consecutive channels of row n0 + nq of G and of X;
//     MFMA (c, c') accumulates the 16x16 sub-block {o = VW i + c} x {k = VW j + c'}: VW^2 MFMAs per 4 rows cover a
//     (16 VW)^2 block of dW.  Waves reduce through LDS, every workgroup stores its block into its own SLAB of a caller-owned workspace,
//     and k_slab_reduce sums the slabs of a block in a fixed order (round 3: no float atomics -- bit-reproducible gradients; dW is
//     written, not accumulated).
//
// fp32 in / fp32 accumulate (bit-equal to an fmaf chain): the reference computes these layers in fp32.
//
// This header is the implementation; it is compiled three times, once per input precision of the products (template parameter MP, see
// Mma below): rowlin2.hip (MP = 0, fp32 operands -- plus the precision-independent helpers, RL2_MAIN_TU), rowlin2_f16.hip (MP = 1),
// rowlin2_bf16.hip (MP = 2).  Each translation unit instantiates try_forward_mp<MP> / try_wgrad_mp<MP> explicitly.
#pragma once
#include "pdfops_common.h"
#include <algorithm>
#include <cstdlib>

namespace rl2 {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Input precision of the products (template parameter MP of the kernels; `mma_input` of the C entry points, include/pdfops.h):
//   0  fp32 operands, v_mfma_f32_16x16x4_f32 (bit-equal to an fmaf chain) -- the default and the parity path;
//   1  operands rounded to fp16 in registers, v_mfma_f32_16x16x16_f16;   2  the same with bfloat16, v_mfma_f32_16x16x16_bf16.
// Activations, weights and gradients stay fp32 in HBM and the accumulators are fp32: what torch.autocast does to an nn.Linear
// (engines/train.py:340-363, enable_amp) minus the half-precision rounding of the OUTPUT.  The lane layouts coincide: lane (i = l & 15,
// kq = l >> 4) holds reduction indices 4 kq .. 4 kq + 3 of a 16-wide step in both forms, so one 16x16x16 product replaces the four
// 16x16x4 products of a step.
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 b16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int MP> struct Mma;
template <> struct Mma<0> {
    typedef f32x4 frag;
    static __device__ __forceinline__ frag cvt(f32x4 v) { return v; }
};
template <> struct Mma<1> {
    typedef h16x4 frag;
    static __device__ __forceinline__ frag cvt(f32x4 v) { return __builtin_convertvector(v, h16x4); }
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 acc) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, acc, 0, 0, 0); }
};
template <> struct Mma<2> {
    typedef s16x4 frag;
    static __device__ __forceinline__ frag cvt(f32x4 v) { return __builtin_bit_cast(s16x4, __builtin_convertvector(v, b16x4)); }
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 acc) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc, 0, 0, 0); }
};

long stats_rows_floats(long n, int o);

struct FwdArgs {
    long N;
    int O;                   // width of ONE output tensor
    const float *X[3]; long ldx;
    const float *W[3]; long wso, wsk;   // Wt(k, o) = W[o * wso + k * wsk]; indexed by input (NIN > 1) or by output
    const float *bias[3];    // per output (nullable)
    const float *scale, *shift; int relu;
    float *Y[3]; long ldy;
    int accumulate;
    float *partial;          // [gridDim.x][2 * O] (ST != 0; single output)
    // ST == 2: the output is the gradient of a BatchNorm(+ReLU) OUTPUT; the partial rows then carry that BatchNorm's backward sums
    // [sum g' | sum g' xhat] (g' = output masked by the ReLU of bx * scale + shift, xhat = (bx - mean) * rstd) instead of [sum | sum of squares]
    const float *bx; long ldb; const float *bcoef; int brelu;   // bx (N, O; row stride ldb), bcoef = [scale | shift | mean | rstd] (4 O)
    const float *roww; long rws;   // optional per-row factor of the product (y = roww[n] * (f(x) Wt) + bias), element stride rws
    // ST == 1, optional: handoff scratch of the CONSUMER's in-kernel BatchNorm finalize (pdfops_common.h: PdfRowsBn) -- zeroed here
    unsigned long long *ho_gran; unsigned *ho_sync;
};

constexpr int FWD_CAP = 1024;   // row-blocks (4 waves each) of the persistent grid
static inline int fwd_row_blocks(long n) {
    const long tiles = (n + 15) / 16, b = (tiles + 3) / 4;
    return (int)(b < 1 ? 1 : (b > FWD_CAP ? FWD_CAP : b));
}

template <int K, int NOB, int NIN, bool PRE, int ST, int MP>   // ST: 0 = no statistics, 1 = [sum | sum of squares] of the output, 2 = BatchNorm-backward sums (FwdArgs); MP: Mma
__global__ __launch_bounds__(256) void k_fwd(FwdArgs a) {
    typedef typename Mma<MP>::frag wfrag;
    constexpr bool STATS = ST != 0;
    constexpr int NJ = K / 16;
    constexpr bool COEF_REGS = K <= 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, kq = lane >> 4;
    if (ST == 1 && a.ho_gran && blockIdx.x == 0 && blockIdx.y == 0) pdf_handoff_zero(a.ho_gran, a.O, a.ho_sync);
    const int gcol0 = blockIdx.y * NOB * 16;   // column over the concatenated outputs; a 16-column block never straddles two
    int outi[NOB], colb[NOB];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) { outi[ob] = (gcol0 + ob * 16) / a.O; colb[ob] = gcol0 + ob * 16 - outi[ob] * a.O; }
    wfrag Wr[NIN][NOB][NJ];
#pragma unroll
    for (int in = 0; in < NIN; ++in) {
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            const float *W = a.W[NIN > 1 ? in : outi[ob]];
            const long o = colb[ob] + li;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                f32x4 w4;
                if (a.wsk == 1) {   // (out, in) row-major: four consecutive reduction indices in one 16-byte load
                    w4 = *reinterpret_cast<const f32x4 *>(W + o * a.wso + 16 * j + 4 * kq);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) w4[c] = W[o * a.wso + (long)(16 * j + 4 * kq + c) * a.wsk];
                }
                Wr[in][ob][j] = Mma<MP>::cvt(w4);
            }
        }
    }
    f32x4 bias4[NOB];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
        const float *b = a.bias[NIN > 1 ? 0 : outi[ob]];
#pragma unroll
        for (int r = 0; r < 4; ++r) bias4[ob][r] = b ? b[colb[ob] + 4 * kq + r] : 0.f;
    }
    f32x4 sc4[COEF_REGS && PRE ? NJ : 1], sh4[COEF_REGS && PRE ? NJ : 1];
    if (PRE && COEF_REGS) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            sc4[j] = *reinterpret_cast<const f32x4 *>(a.scale + 16 * j + 4 * kq);
            sh4[j] = *reinterpret_cast<const f32x4 *>(a.shift + 16 * j + 4 * kq);
        }
    }
    f32x4 s4[STATS ? NOB : 1], ss4[STATS ? NOB : 1];
    if (STATS) {
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) { s4[ob] = f32x4{0.f, 0.f, 0.f, 0.f}; ss4[ob] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
    // ST == 2: the BatchNorm coefficients of this workgroup's columns live in LDS (registers are full of weight fragments: 16 more per
    // column block would halve the occupancy of the three-input K = 256 kernel); rstd is applied once, to the finished sums.
    __shared__ float bco[ST == 2 ? 3 : 1][NOB * 16];   // scale | shift | mean
    if (ST == 2) {
        if (threadIdx.x < 3 * NOB * 16) {
            const int v = threadIdx.x / (NOB * 16), t = threadIdx.x % (NOB * 16);
            bco[ST == 2 ? v : 0][t] = a.bcoef[(long)v * a.O + gcol0 + t];
        }
        __syncthreads();
    }
    // Prologue coefficients of the wide shapes (K > 64) in LDS: read next to the row loads without a trip to L2.
    __shared__ float pco[PRE && !COEF_REGS ? 2 : 1][PRE && !COEF_REGS ? K : 1];
    if (PRE && !COEF_REGS) {
        for (int t = threadIdx.x; t < K; t += 256) { pco[0][t] = a.scale[t]; pco[PRE && !COEF_REGS ? 1 : 0][t] = a.shift[t]; }
        __syncthreads();
    }
    const float relu_lo = a.relu ? 0.f : -INFINITY;   // max(x, lo): no branch between the loads of a trip
    // Row loads of a tile are issued XB at a time ahead of the products that consume them.  Left to the compiler the K = 128 .. 512 shapes
    // compile to load -> s_waitcnt -> 8 products per 16 reduction indices (the weight fragments fill the register file): 16-48 dependent
    // trips to L2 per tile, which is what a level-4 / level-5 launch (one tile per wave) consists of.  XB is what the registers allow
    // next to the weights: the three-input K = 256 kernel holds 192 of them (VGPRs + AGPRs <= 256 keeps two waves per SIMD).
    // (with statistics and several column blocks per wave the accumulators of the sums take the room of half a batch)
    constexpr int XB0 = NJ < 16 ? NJ : 16;
    constexpr int XB = (K == 256 && NIN == 3) ? 4 : ((ST != 0 && NOB > 1 && NIN * NOB * NJ * 4 >= 128) ? ((PRE && K == 128) ? 2 : XB0 / 2) : XB0);
    const long ntiles = (a.N + 15) / 16;
    for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
        const long n = tile * 16 + li;
        const bool valid = n < a.N;
        f32x4 acc[NOB];
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) acc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int in = 0; in < NIN; ++in) {
            const float *xr = a.X[in] + (valid ? n : 0) * a.ldx + 4 * kq;   // (rows past the end read row 0: a column of the product depends on its own row only)
#pragma unroll
            for (int j0 = 0; j0 < NJ; j0 += XB) {
                f32x4 xb[XB];
#pragma unroll
                for (int t = 0; t < XB; ++t) xb[t] = *reinterpret_cast<const f32x4 *>(xr + 16 * (j0 + t));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < XB; ++t) {
                    const int j = j0 + t;
                    f32x4 x4 = xb[t];
                    if (PRE) {
                        const f32x4 sc = COEF_REGS ? sc4[COEF_REGS ? j : 0] : *reinterpret_cast<const f32x4 *>(&pco[0][16 * j + 4 * kq]);
                        const f32x4 sh = COEF_REGS ? sh4[COEF_REGS ? j : 0] : *reinterpret_cast<const f32x4 *>(&pco[PRE && !COEF_REGS ? 1 : 0][16 * j + 4 * kq]);
                        x4 = x4 * sc + sh;
#pragma unroll
                        for (int c = 0; c < 4; ++c) x4[c] = fmaxf(x4[c], relu_lo);
                    }
                    if constexpr (MP == 0) {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
#pragma unroll
                            for (int ob = 0; ob < NOB; ++ob)
                                acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wr[in][ob][j][c], x4[c], acc[ob], 0, 0, 0);
                    } else {
                        const wfrag xh = Mma<MP>::cvt(x4);
#pragma unroll
                        for (int ob = 0; ob < NOB; ++ob) acc[ob] = Mma<MP>::mma(Wr[in][ob][j], xh, acc[ob]);
                    }
                }
            }
        }
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            f32x4 v = acc[ob];
            if (a.roww) v *= a.roww[(valid ? n : 0) * a.rws];
            v += bias4[ob];
            float *dst = a.Y[outi[ob]] + (valid ? n : 0) * a.ldy + colb[ob] + 4 * kq;
            if (a.accumulate && valid) v += *reinterpret_cast<const f32x4 *>(dst);
            if (valid) *reinterpret_cast<f32x4 *>(dst) = v;
            if (ST == 1 && valid) { s4[ob] += v; ss4[ob] += v * v; }
            if (ST == 2) {
                const f32x4 x4 = *reinterpret_cast<const f32x4 *>(a.bx + (valid ? n : 0) * a.ldb + colb[ob] + 4 * kq);
                const f32x4 sc = *reinterpret_cast<const f32x4 *>(&bco[0][ob * 16 + 4 * kq]);
                const f32x4 sh = *reinterpret_cast<const f32x4 *>(&bco[ST == 2 ? 1 : 0][ob * 16 + 4 * kq]);
                const f32x4 mu = *reinterpret_cast<const f32x4 *>(&bco[ST == 2 ? 2 : 0][ob * 16 + 4 * kq]);
                const f32x4 pre = x4 * sc + sh;
                f32x4 gm;
#pragma unroll
                for (int c = 0; c < 4; ++c) gm[c] = (valid && (!a.brelu || pre[c] > 0.f)) ? v[c] : 0.f;
                s4[ob] += gm;
                ss4[ob] += gm * (x4 - mu);
            }
        }
    }
    if (STATS) {
        __shared__ float red[4][2][NOB * 16];
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = s4[ob][r], ss = ss4[ob][r];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) { s += __shfl_xor(s, m, 64); ss += __shfl_xor(ss, m, 64); }
                if (li == 0) { red[wave][0][ob * 16 + 4 * kq + r] = s; red[wave][1][ob * 16 + 4 * kq + r] = ss; }
            }
        __syncthreads();
        const int t = threadIdx.x;
        if (t < NOB * 16) {
            float *row = a.partial + (size_t)blockIdx.x * 2 * a.O;
            row[gcol0 + t] = red[0][0][t] + red[1][0][t] + red[2][0][t] + red[3][0][t];      // STATS: single output, gcol0 == column
            const float ss = red[0][1][t] + red[1][1][t] + red[2][1][t] + red[3][1][t];
            row[a.O + gcol0 + t] = ST == 2 ? ss * a.bcoef[3 * (long)a.O + gcol0 + t] : ss;
        }
    }
}

constexpr int WG_MAXG = 5;   // weight gradients of one launch (blockIdx.z): q / k / v of one input, or the five c x c products of a Bottleneck
struct WArgs {
    long N;
    int K, O;
    const float *G[WG_MAXG]; long ldg;
    const float *X; long ldx;
    const float *scale, *shift; int relu;
    // per-matrix inputs (per_z != 0; kernels instantiated with PRE = true): its own X and, where scalez[z] is non-null, its own folded
    // BatchNorm + ReLU prologue (a null scalez[z] runs the prologue with scale 1, shift 0, no ReLU: x * 1 + 0 is x)
    int per_z; const float *Xz[WG_MAXG]; const float *scalez[WG_MAXG], *shiftz[WG_MAXG]; int reluz[WG_MAXG];
    float *dW[WG_MAXG], *db[WG_MAXG];
    long rows_per_block;     // multiple of 64
    const float *roww; long rws;   // optional per-row weight of G (dW = sum_n roww[n] G[n]^T f(X[n]))
    float *slab;             // [gridDim.z][gridDim.y][gridDim.x][(16 VW)^2]: one block of dW per workgroup
    float *bslab;            // [gridDim.z][O / (16 VW)][gridDim.x][16 VW]: the workgroup's column sums of G (bias gradient), k-block 0 only
};

template <int VW> struct Vec;
template <> struct Vec<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct Vec<4> { typedef float type __attribute__((ext_vector_type(4))); };

template <int VW, bool PRE, bool RW, int MP>   // RW: per-row weight of G (a.roww); MP: Mma
__global__ __launch_bounds__(256) void k_wg(WArgs a) {
    typedef typename Vec<VW>::type vec;
    constexpr int B = 16 * VW;   // block edge of dW
    __shared__ float red[4][B * B];   // one (16 VW)^2 block per wave: plain stores, then a 4-way sum (LDS atomics cost 20 us here)
    __shared__ float redb[4][B];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, nq = lane >> 4;
    const int nkb = a.K / B;
    const int ob = (blockIdx.y / nkb) * B, kb = (blockIdx.y % nkb) * B;
    const float *G = a.G[blockIdx.z];
    const float *X = (PRE && a.per_z) ? a.Xz[blockIdx.z] : a.X;
    vec sc, sh;
    bool relu = a.relu != 0;
    if (PRE) {
        const float *scp = a.per_z ? a.scalez[blockIdx.z] : a.scale, *shp = a.per_z ? a.shiftz[blockIdx.z] : a.shift;
        if (a.per_z) relu = scp != nullptr && a.reluz[blockIdx.z] != 0;
        if (scp) {   // (uniform per workgroup)
            sc = *reinterpret_cast<const vec *>(scp + kb + VW * li);
            sh = *reinterpret_cast<const vec *>(shp + kb + VW * li);
        } else {
#pragma unroll
            for (int c = 0; c < VW; ++c) { sc[c] = 1.f; sh[c] = 0.f; }
        }
    }
    const float relu_lo = relu ? 0.f : -INFINITY;
    f32x4 acc[VW][VW];
    float gsum[VW];
#pragma unroll
    for (int c = 0; c < VW; ++c) {
        gsum[c] = 0.f;
#pragma unroll
        for (int c2 = 0; c2 < VW; ++c2) acc[c][c2] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const long rb = (long)blockIdx.x * a.rows_per_block;
    const long re = rb + a.rows_per_block < a.N ? rb + a.rows_per_block : a.N;
    for (long r0 = rb + 16 * wave; r0 < re; r0 += 64) {
        // the eight (twelve) loads of a trip first, nothing conditional between them: with the row weight behind `if (a.roww)` and the
        // ReLU behind `if (a.relu)` the trip compiled to four dependent (g, x) round trips (k_wg<4, true>: 22.7 -> 17.7 us, <4, false>:
        // 14.9 -> 12.2 us per launch over a step).  (Issuing the next trip's loads ahead of this trip's products: the compiler merges the
        // two trips into one of 16 loads, 208 + 88 registers, one wave per SIMD.)
        vec gv[4], xv[4];
        float rw[RW ? 4 : 1];
        bool ok[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const long n = r0 + 4 * t + nq;
            ok[t] = n < re;
            const long nn = ok[t] ? n : rb;
            gv[t] = *reinterpret_cast<const vec *>(G + nn * a.ldg + ob + VW * li);
            xv[t] = *reinterpret_cast<const vec *>(X + nn * a.ldx + kb + VW * li);
            if (RW) rw[RW ? t : 0] = a.roww[nn * a.rws];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (RW) gv[t] *= rw[RW ? t : 0];
            if (PRE) {
                xv[t] = xv[t] * sc + sh;
#pragma unroll
                for (int c = 0; c < VW; ++c) xv[t][c] = fmaxf(xv[t][c], relu_lo);
            }
#pragma unroll
            for (int c = 0; c < VW; ++c) { gv[t][c] = ok[t] ? gv[t][c] : 0.f; xv[t][c] = ok[t] ? xv[t][c] : 0.f; }
        }
        if constexpr (MP == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int c = 0; c < VW; ++c) {
                    gsum[c] += gv[t][c];
#pragma unroll
                    for (int c2 = 0; c2 < VW; ++c2)
                        acc[c][c2] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv[t][c], xv[t][c2], acc[c][c2], 0, 0, 0);
                }
        } else {   // reduction index of the 16x16x16 product = 4 nq + t  <->  row r0 + 4 t + nq (the same bijection on both operands)
            typename Mma<MP>::frag ga[VW], xa[VW];
#pragma unroll
            for (int c = 0; c < VW; ++c) {
                gsum[c] += (gv[0][c] + gv[1][c]) + (gv[2][c] + gv[3][c]);
                ga[c] = Mma<MP>::cvt(f32x4{gv[0][c], gv[1][c], gv[2][c], gv[3][c]});
                xa[c] = Mma<MP>::cvt(f32x4{xv[0][c], xv[1][c], xv[2][c], xv[3][c]});
            }
#pragma unroll
            for (int c = 0; c < VW; ++c)
#pragma unroll
                for (int c2 = 0; c2 < VW; ++c2) acc[c][c2] = Mma<MP>::mma(ga[c], xa[c2], acc[c][c2]);
        }
    }
    // D layout: acc[c][c2][r] = dW[ob + VW (4 nq + r) + c][kb + VW li + c2]
#pragma unroll
    for (int c = 0; c < VW; ++c)
#pragma unroll
        for (int c2 = 0; c2 < VW; ++c2)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][(VW * (4 * nq + r) + c) * B + VW * li + c2] = acc[c][c2][r];
    float *db = a.db[blockIdx.z];
    const bool want_db = db && kb == 0;
    if (want_db) {
#pragma unroll
        for (int c = 0; c < VW; ++c) {
            float g = gsum[c];
            g += __shfl_xor(g, 16, 64);
            g += __shfl_xor(g, 32, 64);
            if (nq == 0) redb[wave][VW * li + c] = g;
        }
    }
    __syncthreads();
    float *slab = a.slab + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (B * B);
    for (int e = threadIdx.x; e < B * B; e += 256) slab[e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    if (want_db && threadIdx.x < B)
        a.bslab[(((size_t)blockIdx.z * (a.O / B) + ob / B) * gridDim.x + blockIdx.x) * B + threadIdx.x] =
            (redb[0][threadIdx.x] + redb[1][threadIdx.x]) + (redb[2][threadIdx.x] + redb[3][threadIdx.x]);
}

// Sum of the `split` slabs of every dW block (and bias block) in a fixed order: lane l of an element adds slabs l, l + L, l + 2L, ...,
// the L partial sums are combined in lane order.  grid = (ceil(B^2 / 64), tiles + bias tiles, ng), 64 L threads.  Works for the tiled
// kernel of rowlin.hip as well (B = 32, edge blocks masked by O / K).
struct RArgs {
    const float *slab, *bslab;
    float *dW[WG_MAXG], *db[WG_MAXG];
    int B, tiles_k, tiles, otiles, split, K, O;
};
void launch_slab_reduce(const RArgs &a, int ng, bool any_bias, hipStream_t s);
template <int L>
__global__ __launch_bounds__(64 * L) void k_slab_reduce(RArgs a) {
    __shared__ float red[L][64];
    const int e = threadIdx.x & 63, l = threadIdx.x >> 6, z = blockIdx.z;
    const int ei = blockIdx.x * 64 + e;
    const bool bias = (int)blockIdx.y >= a.tiles;
    const int t = bias ? (int)blockIdx.y - a.tiles : (int)blockIdx.y;
    const int len = bias ? a.B : a.B * a.B;
    if (bias && (int)blockIdx.x * 64 >= len) return;   // (uniform per block)
    const float *src = bias ? a.bslab + ((size_t)z * a.otiles + t) * a.split * a.B : a.slab + ((size_t)z * a.tiles + t) * a.split * (size_t)(a.B * a.B);
    float s = 0.f;
    if (ei < len) {
        int k = l;
        for (; k + 3 * L < a.split; k += 4 * L) {   // four loads in flight per lane
            const float v0 = src[(size_t)k * len + ei], v1 = src[(size_t)(k + L) * len + ei];
            const float v2 = src[(size_t)(k + 2 * L) * len + ei], v3 = src[(size_t)(k + 3 * L) * len + ei];
            s = (((s + v0) + v1) + v2) + v3;
        }
        for (; k < a.split; k += L) s += src[(size_t)k * len + ei];
    }
    red[l][e] = s;
    __syncthreads();
    if (l != 0 || ei >= len) return;
#pragma unroll
    for (int k = 1; k < L; ++k) s += red[k][e];
    if (bias) {
        const int o = t * a.B + ei;
        if (a.db[z] && o < a.O) a.db[z][o] = s;
    } else {
        const int o = (t / a.tiles_k) * a.B + ei / a.B, kk = (t % a.tiles_k) * a.B + ei % a.B;
        if (o < a.O && kk < a.K) a.dW[z][(size_t)o * a.K + kk] = s;
    }
}

#ifdef RL2_MAIN_TU
void launch_slab_reduce(const RArgs &a, int ng, bool any_bias, hipStream_t s) {
    const dim3 grid((unsigned)((a.B * a.B + 63) / 64), (unsigned)(a.tiles + (any_bias ? a.otiles : 0)), (unsigned)ng);
    if (a.split > 32) k_slab_reduce<16><<<grid, 64 * 16, 0, s>>>(a);
    else if (a.split > 4) k_slab_reduce<4><<<grid, 64 * 4, 0, s>>>(a);
    else k_slab_reduce<1><<<grid, 64, 0, s>>>(a);
}
#endif

// split of the row range for (n, k, o, ng): workgroups in flight vs rows per workgroup (shared by the launch and the workspace size)
struct WgPlan { int vw, b, nblk; long split, rows_per_block; };
static WgPlan wg_plan(long n, int k, int o, int ng) {
    WgPlan p;
    p.vw = (k % 64 == 0 && o % 64 == 0) ? 4 : ((k % 32 == 0 && o % 32 == 0) ? 2 : 0);
    p.b = 16 * p.vw;
    if (!p.vw) { p.nblk = 0; p.split = 0; p.rows_per_block = 0; return p; }
    p.nblk = (o / p.b) * (k / p.b) * ng;
    static const int target = [] { const char *v = getenv("PDFOPS_WG_BLOCKS"); const int x = v ? atoi(v) : 0; return x > 0 ? x : 512; }();
    long split = (target + p.nblk - 1) / p.nblk;       // workgroups in flight
    const long max_split = (n + 255) / 256;            // at least 256 rows (4 trips per wave) per workgroup
    if (split > max_split) split = max_split;
    if (split < 1) split = 1;
    p.rows_per_block = ((n + split - 1) / split + 63) / 64 * 64;
    p.split = (n + p.rows_per_block - 1) / p.rows_per_block;
    return p;
}
#ifdef RL2_MAIN_TU
long wgrad_ws_floats(long n, int k, int o, int ng) {
    const WgPlan p = wg_plan(n, k, o, ng);
    if (!p.vw) return 0;
    return (long)p.nblk * p.split * p.b * p.b + (long)ng * (o / p.b) * p.split * p.b;
}
#endif

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int K, int NOB, int NIN, int MP>
static int launch_fwd(const FwdArgs &a, int nslabs, hipStream_t s) {   // returns the number of row blocks (= partial rows written)
    const bool pre = a.scale != nullptr, stats = a.partial != nullptr, bst = stats && a.bx != nullptr;
    // Every workgroup first loads its waves' weight fragments (16 NOB columns x K: 8-32 KB per wave).  With many column slabs and few
    // rows (levels 4-5: 3,124 / 780 rows, up to 96 slabs for the three-output q / k / v product) one row block per 64 rows means a wave
    // loads its fragment for a single 16-row tile; fewer row blocks amortise it over several tiles.  (Not with statistics: the partial
    // rows are indexed by row block and sized by pdf_rowlin_partial_rows.)
    // Measured (PDFOPS_RL_BLOCKS = total workgroups aimed at): 512 -> q / k / v product at 3,124 x 256: 41 -> 31 us, three-input dgrad 36 -> 30 us,
    // 780 x 512: 54 -> 46 us, single-slab shapes unchanged; 384 and below lose on the large levels, 1024 and above change nothing.
    static const int total = [] { const char *v = getenv("PDFOPS_RL_BLOCKS"); const int x = v ? atoi(v) : 0; return x > 0 ? x : 512; }();
    int gx = fwd_row_blocks(a.N);
    if (!stats || bst) gx = std::max(1, std::min(gx, std::max(8, total / std::max(nslabs, 1))));   // (bst: the caller is told the row count)
    const dim3 grid((unsigned)gx, (unsigned)nslabs);
    if (bst) {   // BatchNorm-backward sums of the output (dgrad kernels: no input prologue)
        k_fwd<K, NOB, NIN, false, 2, MP><<<grid, 256, 0, s>>>(a);
    } else if (NIN == 1 && stats) {
        if (pre) k_fwd<K, NOB, 1, true, 1, MP><<<grid, 256, 0, s>>>(a);
        else k_fwd<K, NOB, 1, false, 1, MP><<<grid, 256, 0, s>>>(a);
    } else {
        if (pre) k_fwd<K, NOB, NIN, true, 0, MP><<<grid, 256, 0, s>>>(a);
        else k_fwd<K, NOB, NIN, false, 0, MP><<<grid, 256, 0, s>>>(a);
    }
    return gx;
}

// rows of the statistics epilogue [row blocks][2 O] floats
#ifdef RL2_MAIN_TU
long stats_rows_floats(long n, int o) { return (long)fwd_row_blocks(n) * 2 * o; }
#endif

// returns 1 when a streaming kernel took the job, 0 when the shape is not covered (caller falls back to the tiled kernel)
template <int MP>
int try_forward_mp(long n, int k, int o, int nin, int nout, const float *const *x, long ldx, const float *const *w, int transpose_w,
                   const float *const *bias, const float *scale, const float *shift, int relu, float *const *y, long ldy,
                   int accumulate, float *partial, hipStream_t s, const float *roww, long rws, const float *bx, long ldb, const float *bcoef,
                   int brelu, int *partial_rows, long ldw, void *handoff) {
    if (nin < 1 || nout < 1 || (nin > 1 && nout > 1) || nin > 3 || nout > 3) return 0;
    if (nin == 2 && (scale || bx)) return 0;   // (the two-window form: plain product only)
    if (partial && !bx && (nin != 1 || nout != 1)) return 0;
    if (bx && (!partial || nout != 1 || scale || !bcoef || (ldb & 3) || !aligned16(bx) || !aligned16(bcoef))) return 0;
    if ((ldx & 3) || (ldy & 3) || (o & 15)) return 0;
    for (int i = 0; i < (nin > 1 ? nin : nout); ++i) if (!aligned16(w[i])) return 0;
    if (scale && (!aligned16(scale) || !aligned16(shift))) return 0;
    for (int i = 0; i < nin; ++i) if (!aligned16(x[i])) return 0;
    for (int i = 0; i < nout; ++i) if (!aligned16(y[i])) return 0;
    FwdArgs a;
    a.N = n; a.O = o; a.ldx = ldx; a.ldy = ldy; a.scale = scale; a.shift = shift; a.relu = relu; a.accumulate = accumulate;
    a.partial = partial; a.roww = roww; a.rws = rws;
    a.bx = bx; a.ldb = ldb; a.bcoef = bcoef; a.brelu = brelu;
    a.ho_gran = static_cast<unsigned long long *>(handoff);
    a.ho_sync = handoff ? reinterpret_cast<unsigned *>(a.ho_gran + 2 * (size_t)o) : nullptr;
    a.wso = transpose_w ? 1 : (ldw ? ldw : k); a.wsk = transpose_w ? o : 1;   // ldw: row stride of an (o, k) window of a wider weight matrix
    for (int i = 0; i < 3; ++i) {
        a.X[i] = i < nin ? x[i] : nullptr;
        a.W[i] = i < (nin > 1 ? nin : nout) ? w[i] : nullptr;
        a.bias[i] = (bias && i < nout) ? bias[i] : nullptr;
        a.Y[i] = i < nout ? y[i] : nullptr;
    }
    const int cols = o * nout;
#define PDF_RL2(K_, NOB_, NIN_) do { if (cols % (NOB_ * 16) == 0) { \
        const int r_ = launch_fwd<K_, NOB_, NIN_, MP>(a, cols / (NOB_ * 16), s); \
        if (partial_rows) *partial_rows = r_; \
        return 1; } } while (0)
    if (nin == 1) {
        switch (k) {
        case 32: if (!partial) PDF_RL2(32, 6, 1); PDF_RL2(32, 2, 1); PDF_RL2(32, 1, 1); break;   // (statistics: at most 2 column blocks per wave)
        case 64: PDF_RL2(64, 4, 1); PDF_RL2(64, 1, 1); break;
        case 128: if (!bx) PDF_RL2(128, 4, 1); else PDF_RL2(128, 2, 1); PDF_RL2(128, 1, 1); break;   // (ST == 2 with 4 column blocks: 305 registers, one wave per SIMD)
        case 256: PDF_RL2(256, 2, 1); PDF_RL2(256, 1, 1); break;
        case 512: PDF_RL2(512, 1, 1); break;
        default: break;
        }
    } else if (nin == 2) {   // two 512-wide column windows of one 1024-wide input (ldw): ONE accumulation chain over both, bias at the end
        if (k == 512) PDF_RL2(512, 1, 2);
    } else if (nin == 3) {
        switch (k) {
        case 32: PDF_RL2(32, 2, 3); break;
        case 64: PDF_RL2(64, 2, 3); break;
        case 128: PDF_RL2(128, 1, 3); break;
        case 256: PDF_RL2(256, 1, 3); break;
        default: break;
        }
    }
#undef PDF_RL2
    return 0;
}

template <int MP>
int try_wgrad_mp(long n, int k, int o, int ng, const float *const *g, long ldg, const float *x, long ldx, const float *scale,
                 const float *shift, int relu, float *const *dw, float *const *db, float *ws, hipStream_t s, const float *roww, long rws) {
    if (ng < 1 || ng > 3 || !ws) return 0;
    const WgPlan p = wg_plan(n, k, o, ng);
    const int vw = p.vw;
    if (!vw) return 0;
    if ((ldg % vw) || (ldx % vw) || !aligned16(x)) return 0;
    if (scale && (!aligned16(scale) || !aligned16(shift))) return 0;
    for (int i = 0; i < ng; ++i) if (!aligned16(g[i])) return 0;
    WArgs a;
    a.N = n; a.K = k; a.O = o; a.ldg = ldg; a.X = x; a.ldx = ldx; a.scale = scale; a.shift = shift; a.relu = relu; a.roww = roww; a.rws = rws;
    a.per_z = 0;
    bool any_bias = false;
    for (int i = 0; i < WG_MAXG; ++i) {
        a.G[i] = i < ng ? g[i] : nullptr; a.dW[i] = i < ng ? dw[i] : nullptr; a.db[i] = (db && i < ng) ? db[i] : nullptr;
        a.Xz[i] = nullptr; a.scalez[i] = a.shiftz[i] = nullptr; a.reluz[i] = 0;
        any_bias = any_bias || a.db[i] != nullptr;
    }
    const int b = p.b;
    a.rows_per_block = p.rows_per_block;
    a.slab = ws;
    a.bslab = ws + (size_t)p.nblk * p.split * b * b;
    const dim3 grid((unsigned)p.split, (unsigned)((o / b) * (k / b)), (unsigned)ng);
#define PDF_WG(VW_) do { \
        if (roww) { if (scale) k_wg<VW_, true, true, MP><<<grid, 256, 0, s>>>(a); else k_wg<VW_, false, true, MP><<<grid, 256, 0, s>>>(a); } \
        else      { if (scale) k_wg<VW_, true, false, MP><<<grid, 256, 0, s>>>(a); else k_wg<VW_, false, false, MP><<<grid, 256, 0, s>>>(a); } } while (0)
    if (vw == 4) PDF_WG(4); else PDF_WG(2);
#undef PDF_WG
    RArgs r;
    r.slab = a.slab; r.bslab = a.bslab; r.B = b; r.tiles_k = k / b; r.tiles = (o / b) * (k / b); r.otiles = o / b; r.split = (int)p.split; r.K = k; r.O = o;
    for (int i = 0; i < WG_MAXG; ++i) { r.dW[i] = a.dW[i]; r.db[i] = a.db[i]; }
    launch_slab_reduce(r, ng, any_bias, s);
    return 1;
}

// Up to WG_MAXG weight gradients of ONE shape (n, k, o) with their OWN inputs in one launch + one slab reduction: dW_i = G_i^T f_i(X_i),
// f_i = relu?(x * scale_i + shift_i) where scale_i is non-null, the identity otherwise.  A Bottleneck's backward has five such products
// (linear3, q / k / v, linear1: all c x c over the block's n rows); issued one by one at levels 3-5 each of them is a launch that fills a
// fraction of the chip followed by its own reduction launch.  ws: wgrad_ws_floats(n, k, o, ng) floats.
template <int MP>
int try_wgrad_group_mp(long n, int k, int o, int ng, const float *const *g, long ldg, const float *const *x, long ldx,
                       const float *const *scale, const float *const *shift, const int *relu, float *const *dw, float *const *db,
                       float *ws, hipStream_t s) {
    if (ng < 1 || ng > WG_MAXG || !ws) return 0;
    const WgPlan p = wg_plan(n, k, o, ng);
    const int vw = p.vw;
    if (!vw) return 0;
    if ((ldg % vw) || (ldx % vw)) return 0;
    for (int i = 0; i < ng; ++i) {
        if (!aligned16(g[i]) || !aligned16(x[i])) return 0;
        if (scale[i] && (!aligned16(scale[i]) || !aligned16(shift[i]))) return 0;
    }
    WArgs a;
    a.N = n; a.K = k; a.O = o; a.ldg = ldg; a.X = nullptr; a.ldx = ldx; a.scale = a.shift = nullptr; a.relu = 0; a.roww = nullptr; a.rws = 0;
    a.per_z = 1;
    bool any_bias = false;
    for (int i = 0; i < WG_MAXG; ++i) {
        const bool on = i < ng;
        a.G[i] = on ? g[i] : nullptr; a.Xz[i] = on ? x[i] : nullptr;
        a.scalez[i] = on ? scale[i] : nullptr; a.shiftz[i] = on ? shift[i] : nullptr; a.reluz[i] = on ? relu[i] : 0;
        a.dW[i] = on ? dw[i] : nullptr; a.db[i] = (on && db) ? db[i] : nullptr;
        any_bias = any_bias || a.db[i] != nullptr;
    }
    const int b = p.b;
    a.rows_per_block = p.rows_per_block;
    a.slab = ws;
    a.bslab = ws + (size_t)p.nblk * p.split * b * b;
    const dim3 grid((unsigned)p.split, (unsigned)((o / b) * (k / b)), (unsigned)ng);
    if (vw == 4) k_wg<4, true, false, MP><<<grid, 256, 0, s>>>(a); else k_wg<2, true, false, MP><<<grid, 256, 0, s>>>(a);
    RArgs r;
    r.slab = a.slab; r.bslab = a.bslab; r.B = b; r.tiles_k = k / b; r.tiles = (o / b) * (k / b); r.otiles = o / b; r.split = (int)p.split; r.K = k; r.O = o;
    for (int i = 0; i < WG_MAXG; ++i) { r.dW[i] = a.dW[i]; r.db[i] = a.db[i]; }
    launch_slab_reduce(r, ng, any_bias, s);
    return 1;
}

#ifdef RL2_MAIN_TU
int stats_rows(long n) { return fwd_row_blocks(n); }
#endif

}  // namespace rl2
