// Fused TransitionDown (stride > 1) for gfx950 -- point_transformer_seg.py:96-119:
//
//   in[m,k]  = [ mask * (p[idx[m,k]] - p_new[m]) (3) | x[idx[m,k]] (C_in) ]          (knn_query_and_group, with_xyz)
//   z        = in W^T                         (Linear(3 + C_in, C_out), no bias)
//   out[m,c] = max_k relu(BN(z)[m,k,c])       (train-mode BatchNorm over all m*k rows, MaxPool1d(k)),  k = 16
//
// The reference materialises in (m,k,3+C_in), z, BN(z) and their gradients: ~16 passes over 205 MB tensors at level 1->2.
// Here only `out` (m, C_out) and the arg-max neighbour (m, C_out, uint8) exist.  Two observations remove most of the work:
//
//  * Every quantity that is LINEAR in the rows of `in` needs only its moments.  Rows that gather the same source point j share
//    x[j], so with the geometry-only tables cnt[j] = #rows gathering j and R[j] = sum of their relative coordinates
//    (Z = [R | cnt | 0..] (N,32), memoised per Geometry) the Gram matrix G = sum_rows in^T in is
//    [[sum rel^T rel, R^T x], [x^T R, x^T diag(cnt) x]] -- two small weight-gradient-shaped products over the N SOURCE points,
//    no gather.  BatchNorm statistics: sum z_c = W_c s_in, sum z_c^2 = W_c G W_c^T (fp64).
//  * The backward of BN(z) is dz = s dy - B z + A per channel (A, B from the two BN sums), and dy is SPARSE (one neighbour per
//    (m, c): the arg-max).  Its dense part reaches the parameters and the inputs only through moments again:
//    dW = [sparse] - diag(B) W G + A^T s_in,   gx[j] = [sparse scatter] - R[j] Q[0:3,3:] - cnt[j] x[j] Q[3:,3:] + cnt[j] v3[3:],
//    Q = W^T diag(B) W, v3 = A W: two row-linear launches over the N source points.  BN sums: S1 = sum g', S2 = sum g' (out-beta)/gamma
//    over (m, C_out) (the normalised value at the arg-max is recoverable from `out`).
//
// Dense k-fold work that remains: the forward pass (k_td_fwd) and the two sparse backward terms (k_td_din, k_td_wg), all on the
// matrix cores (v_mfma_f32_16x16x4_f32, one wave = one new point = 16 rows, as fused_layer_mfma.hip).
#include "pdfops_common.h"

namespace fl {
void launch_colsum(const float *partial, int rows, int width, float *out, hipStream_t s);
void launch_bn_eval(int nch, const float *gamma, const float *beta, float eps, const float *running_mean, const float *running_var,
                    float *scale, float *shift, float *mean_out, float *rstd_out, hipStream_t s);
}

namespace td {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int WPB = 4;
__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ float max16(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v)); v = fmaxf(v, dpp_f<0x4E>(v)); v = fmaxf(v, dpp_f<0x141>(v)); v = fmaxf(v, dpp_f<0x140>(v));
    return v;
}
__device__ __forceinline__ int min16(int v) {
    v = min(v, dpp_i<0xB1>(v)); v = min(v, dpp_i<0x4E>(v)); v = min(v, dpp_i<0x141>(v)); v = min(v, dpp_i<0x140>(v));
    return v;
}

// element (a, b) of the Gram matrix in W's column order (rel first): consts = [S_rr (9) | s_r (3)], gz = (32, cin): rows 0-2 = R^T x,
// row 3 = cnt^T x; gxx = x^T diag(cnt) x
__device__ __forceinline__ double gram(int a, int b, int cin, const float *consts, const float *gz, const float *gxx) {
    if (a < 3 && b < 3) return (double)consts[a * 3 + b];
    if (a < 3) return (double)gz[(size_t)a * cin + (b - 3)];
    if (b < 3) return (double)gz[(size_t)b * cin + (a - 3)];
    return (double)gxx[(size_t)(a - 3) * cin + (b - 3)];
}
__device__ __forceinline__ double s_in(int a, int cin, const float *consts, const float *gz) {
    return a < 3 ? (double)consts[9 + a] : (double)gz[(size_t)3 * cin + (a - 3)];
}

// t_i = sum_j G[i][j] w[j] (G symmetric, W's column order: rel first).  The feature part is a plain strided-by-row walk with
// independent loads: unrolled by 8, otherwise these short kernels are a chain of ~260 L2 latencies.
__device__ __forceinline__ double gram_row_dot(int i, int cin, const float *consts, const float *gz, const float *gxx, const float *w) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) t += gram(i, j, cin, consts, gz, gxx) * (double)w[j];
    if (i < 3) {
        const float *g = gz + (size_t)i * cin;
#pragma unroll 8
        for (int j = 0; j < cin; ++j) t += (double)g[j] * (double)w[3 + j];
    } else {   // column i - 3 of gxx (symmetric: coalesced across the threads of a block)
        const float *g = gxx + (i - 3);
#pragma unroll 8
        for (int j = 0; j < cin; ++j) t += (double)g[(size_t)j * cin] * (double)w[3 + j];
    }
    return t;
}

__device__ __forceinline__ double block_sum(double v, double *red) {   // 256 threads
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const double r = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return r;
}

// ---------------------------------------------------------------------------------------------------- BN coefficients (forward)
// one block per output channel: sum z = W_c . s_in, sum z^2 = W_c G W_c^T in fp64 -> scale | shift | mean | rstd (+ running stats)
__global__ __launch_bounds__(256) void k_td_coef(int cin, int cout, double rows, const float *__restrict__ W, const float *__restrict__ consts,
                                                 const float *__restrict__ gz, const float *__restrict__ gxx, const float *__restrict__ gamma,
                                                 const float *__restrict__ beta, float eps, float momentum, float *__restrict__ running_mean,
                                                 float *__restrict__ running_var, float *__restrict__ coef) {
    __shared__ double red[4];
    const int c = blockIdx.x, d = 3 + cin;
    const float *w = W + (size_t)c * d;
    double sz = 0.0, sz2 = 0.0;
    for (int i = threadIdx.x; i < d; i += 256) {
        const double t = gram_row_dot(i, cin, consts, gz, gxx, w);
        sz2 += (double)w[i] * t;
        sz += (double)w[i] * s_in(i, cin, consts, gz);
    }
    sz = block_sum(sz, red);
    sz2 = block_sum(sz2, red);
    if (threadIdx.x == 0) {
        const double mean = sz / rows;
        double var = sz2 / rows - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * rstd;
        coef[c] = sc; coef[cout + c] = beta[c] - (float)mean * sc; coef[2 * cout + c] = (float)mean; coef[3 * cout + c] = rstd;
        if (running_mean) {
            const double unbiased = rows > 1.0 ? var * rows / (rows - 1.0) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
}

// ---------------------------------------------------------------------------------------------------- forward (dense, MFMA)
// one wave = one new point m (16 rows); block = 64 output channels (blockIdx.y); lane (row = l & 15, kq = l >> 4).
// LDS: W slab [64][CIN + 4] with the feature columns first and the three relative-coordinate columns at [CIN .. CIN + 2].
template <int CIN>
__global__ __launch_bounds__(64 * WPB) void k_td_fwd(int M, int cout, const float *__restrict__ x, const int *__restrict__ idx,
                                                     const float *__restrict__ rel4, const float *__restrict__ W, const float *__restrict__ coef,
                                                     float *__restrict__ out, unsigned char *__restrict__ arg) {
    constexpr int NJ = CIN / 16, WS = CIN + 4, D = 3 + CIN;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *wl = lds, *sc = wl + 64 * WS, *sh = sc + 64;
    const int c0 = 64 * blockIdx.y;
    for (int e = threadIdx.x; e < 64 * D; e += 64 * WPB) {
        const int c = e / D, j = e % D;
        wl[c * WS + (j < 3 ? CIN + j : j - 3)] = W[(size_t)(c0 + c) * D + j];
    }
    if (threadIdx.x < 64) { sc[threadIdx.x] = coef[c0 + threadIdx.x]; sh[threadIdx.x] = coef[cout + c0 + threadIdx.x]; }
    __syncthreads();
    const int lane = threadIdx.x & 63, row = lane & 15, kq = lane >> 4;
    const long wave_g = (long)blockIdx.x * WPB + (threadIdx.x >> 6), nwaves = (long)gridDim.x * WPB;
    for (long m = wave_g; m < M; m += nwaves) {
        const int nb = idx[m * 16 + row];
        const f32x4 rl = ld4(rel4 + (m * 16 + row) * 4);
        f32x4 acc[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[cb] = zero4();
#pragma unroll 2
        for (int j = 0; j < NJ; ++j) {
            const f32x4 xg = nb >= 0 ? ld4(x + (size_t)nb * CIN + 16 * j + 4 * kq) : zero4();
            f32x4 w[4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) w[cb] = ld4(wl + (16 * cb + row) * WS + 16 * j + 4 * kq);   // A: (channel 16 cb + (l & 15), k = kq)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[cb][e], xg[e], acc[cb], 0, 0, 0);
        }
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            f32x4 best;
            unsigned packed = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {   // D fragment: channel 16 cb + 4 kq + r of row (l & 15)
                const int cl = 16 * cb + 4 * kq + r;
                const float *wr = wl + cl * WS + CIN;
                const float z = acc[cb][r] + rl[0] * wr[0] + rl[1] * wr[1] + rl[2] * wr[2];
                const float y = fmaxf(z * sc[cl] + sh[cl], 0.f);
                const float mx = max16(y);
                const int am = min16(y == mx ? row : 16);   // first neighbour attaining the maximum (MaxPool1d's index)
                best[r] = mx;
                packed |= (unsigned)am << (8 * r);
            }
            if (row == 0) {
                *reinterpret_cast<f32x4 *>(out + (size_t)m * cout + c0 + 16 * cb + 4 * kq) = best;
                *reinterpret_cast<unsigned *>(arg + (size_t)m * cout + c0 + 16 * cb + 4 * kq) = packed;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------- backward: small algebra
// fake BatchNorm coefficients that make k_bn_bwd_reduce return [S1 | S2]: pre = out (mask out > 0), xhat = (out - beta) / gamma
__global__ void k_td_fake(int cout, const float *__restrict__ gamma, const float *__restrict__ beta, float *__restrict__ fake) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cout) return;
    fake[c] = 1.f; fake[cout + c] = 0.f; fake[2 * cout + c] = beta[c]; fake[3 * cout + c] = 1.f / gamma[c];
}

// per channel c: A_c, B_c and the dense part of dW: dW[c][col] = -B_c (W G)[c][col] + A_c s_in[col]   (plain stores; k_td_wg adds the sparse part)
__global__ __launch_bounds__(256) void k_td_small1(int cin, int cout, double rows, const float *__restrict__ W, const float *__restrict__ consts,
                                                   const float *__restrict__ gz, const float *__restrict__ gxx, const float *__restrict__ coef,
                                                   const float *__restrict__ sums, float *__restrict__ AB, float *__restrict__ dW) {
    const int c = blockIdx.x, d = 3 + cin;
    const float *w = W + (size_t)c * d;
    const double s = coef[c], mean = coef[2 * cout + c], rstd = coef[3 * cout + c], S1 = sums[c], S2 = sums[cout + c];
    const double Bc = s * rstd * S2 / rows, Ac = -s * S1 / rows + s * rstd * mean * S2 / rows;
    if (threadIdx.x == 0) { AB[c] = (float)Ac; AB[cout + c] = (float)Bc; }
    for (int col = threadIdx.x; col < d; col += 256) {
        const double t = gram_row_dot(col, cin, consts, gz, gxx, w);   // (W G)[c][col] = (G w)[col], G symmetric
        dW[(size_t)c * d + col] = (float)(-Bc * t + Ac * s_in(col, cin, consts, gz));
    }
}

// Q = W^T diag(B) W and v3 = A W, written as the two Linear weights of the dense input-gradient launches:
//   Qp  (cin, cin)  = -Q[3:, 3:]                     gx  = cnt .* (x Qp^T)           (Q symmetric)
//   QqT (cin, 32)   : column q < 3 = -Q[q, 3:], column 3 = v3[3:], rest 0            gx += Z QqT^T
__global__ __launch_bounds__(256) void k_td_small2(int cin, int cout, const float *__restrict__ W, const float *__restrict__ AB,
                                                   float *__restrict__ Qp, float *__restrict__ QqT) {
    const int i = blockIdx.x, d = 3 + cin;   // i < d: row i of Q; i == d: v3
    for (int j = 3 + threadIdx.x; j < d; j += 256) {
        double t = 0.0;
        if (i < d) {
#pragma unroll 8
            for (int c = 0; c < cout; ++c) t += (double)W[(size_t)c * d + i] * (double)AB[cout + c] * (double)W[(size_t)c * d + j];
        } else {
#pragma unroll 8
            for (int c = 0; c < cout; ++c) t += (double)AB[c] * (double)W[(size_t)c * d + j];
        }
        if (i >= 3 && i < d) Qp[(size_t)(i - 3) * cin + (j - 3)] = (float)(-t);
        else if (i < 3) QqT[(size_t)(j - 3) * 32 + i] = (float)(-t);
        else {
            float *rowp = QqT + (size_t)(j - 3) * 32;
            rowp[3] = (float)t;
            for (int q = 4; q < 32; ++q) rowp[q] = 0.f;
        }
    }
}

// ---------------------------------------------------------------------------------------------------- backward: sparse input gradient
// d_in[(m,k)][i] = sum_c [k == arg[m,c]] s_c g'[m,c] W[c][3+i]  -> scatter-add into gx[idx[m,k]].  One wave = one point, one block =
// 64 output channels (their shares add up in gx).  A = selected gradients (rows = neighbours, contraction = channels), B = W slab.
template <int CIN>
__global__ __launch_bounds__(64 * WPB) void k_td_din(int M, int cout, const float *__restrict__ gout, const float *__restrict__ out,
                                                     const unsigned char *__restrict__ arg, const int *__restrict__ idx,
                                                     const float *__restrict__ W, const float *__restrict__ coef, float *__restrict__ gx) {
    constexpr int NI = CIN / 16, WS = CIN + 4, D = 3 + CIN;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *wl = lds;
    int *rowids = reinterpret_cast<int *>(wl + 64 * WS);
    const int c0 = 64 * blockIdx.y;
    for (int e = threadIdx.x; e < 64 * CIN; e += 64 * WPB) wl[(e / CIN) * WS + e % CIN] = W[(size_t)(c0 + e / CIN) * D + 3 + e % CIN];
    __syncthreads();
    const int lane = threadIdx.x & 63, row = lane & 15, kq = lane >> 4, wv = threadIdx.x >> 6;
    int *rowid = rowids + wv * 16;
    const long wave_g = (long)blockIdx.x * WPB + wv, nwaves = (long)gridDim.x * WPB;
    for (long m = wave_g; m < M; m += nwaves) {
        if (kq == 0) rowid[row] = idx[m * 16 + row];
        f32x4 sel[4];   // A operand: row = neighbour l & 15, contraction index = channel 16 cj + 4 kq + e
#pragma unroll
        for (int cj = 0; cj < 4; ++cj) {
            const int c = c0 + 16 * cj + 4 * kq;
            const f32x4 g = ld4(gout + (size_t)m * cout + c), o = ld4(out + (size_t)m * cout + c), s = ld4(coef + c);
            const unsigned a4 = *reinterpret_cast<const unsigned *>(arg + (size_t)m * cout + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) sel[cj][e] = (o[e] > 0.f && (int)((a4 >> (8 * e)) & 0xffu) == row) ? s[e] * g[e] : 0.f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 2
        for (int ib = 0; ib < NI; ++ib) {
            f32x4 acc = zero4();
#pragma unroll
            for (int cj = 0; cj < 4; ++cj)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sel[cj][e], wl[(16 * cj + 4 * kq + e) * WS + 16 * ib + row], acc, 0, 0, 0);
            // D fragment: neighbour 4 kq + r, feature 16 ib + (l & 15)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int dst = rowid[4 * kq + r];
                if (dst >= 0 && acc[r] != 0.f) pdf_atomic_add(gx + (size_t)dst * CIN + 16 * ib + row, acc[r]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------- backward: sparse input gradient, destination order
// The same sum as k_td_din without atomics (round 3: bit-reproducible): gx[v, :] += sum over the entries (m, j) of the grouping table that
// gather source point v -- ascending entry id, from the INVERSE table (csrc/seg_gather.hip layout) -- of
//     sum_{c : arg[m, c] == j and out[m, c] > 0} s_c g[m, c] W[c, 3 + :].
// The product is sparse (one arg-max neighbour per (m, c): ~cout / 16 channels hit per entry), so the dense MFMA form would spend
// 15/16 of its work on zeros: one wave = one source point, lanes = the 64 channels of a chunk for the hit test (ballot), then lanes =
// features for the few hit rows of W.  Every element of gx is owned by one lane: plain read-modify-write, fixed order.
// WLDS (cin <= 64: the feature part of W is <= 32 KB): the hit loop reads its W rows from an LDS copy -- every hit is a DEPENDENT load
// (ballot -> row address), ~16 of them per source point at level 1; from L2 that chain was most of the kernel (SQ counters, round 4:
// 83 % of the wave cycles parked in s_waitcnt, 201 us at 200k source points).
template <int CIN, bool WLDS>
__global__ __launch_bounds__(64 * WPB) void k_td_din_dst(int n, int cout, const float *__restrict__ gout, const float *__restrict__ out,
                                                         const unsigned char *__restrict__ arg, const float *__restrict__ W,
                                                         const float *__restrict__ coef, const int *__restrict__ inv_off,
                                                         const int *__restrict__ inv_entry, int entry_base, float *__restrict__ gx) {
    constexpr int NQ = (CIN + 63) / 64, D = 3 + CIN;
    extern __shared__ __attribute__((aligned(16))) float wlds[];   // [cout][CIN] (WLDS)
    if (WLDS) {
        for (int e = threadIdx.x; e < cout * CIN; e += 64 * WPB) wlds[e] = W[(size_t)(e / CIN) * D + 3 + e % CIN];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long wave_g = (long)blockIdx.x * WPB + wv, nwaves = (long)gridDim.x * WPB;
    for (long v = wave_g; v < n; v += nwaves) {
        const int beg = inv_off[v], end = inv_off[v + 1];
        float acc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = 0.f;
        // entries in batches of EB: the entry ids, then the EB x 3 row loads of a channel chunk go out together (the chain entry id ->
        // rows -> ballot -> W rows is pure latency: one entry at a time it was 262 us at level 1, 4 entries per source on average)
        constexpr int EB = 4;
        for (int p0 = beg; p0 < end; p0 += EB) {
            long m[EB];
            int j[EB];
#pragma unroll
            for (int u = 0; u < EB; ++u) {
                const int pp = p0 + u < end ? p0 + u : end - 1;
                const int e = inv_entry[pp] - entry_base;
                m[u] = e >> 4;
                j[u] = p0 + u < end ? (e & 15) : -1;   // (-1: matches no arg byte)
            }
            for (int cb = 0; cb < cout; cb += 64) {
                const int c = cb + lane;
                const float sc = coef[c];
                int a[EB];
                float o[EB], g[EB];
#pragma unroll
                for (int u = 0; u < EB; ++u) {
                    const size_t at = (size_t)m[u] * cout + c;
                    a[u] = arg[at]; o[u] = out[at]; g[u] = gout[at];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < EB; ++u) {   // ascending entry id: the summation order is fixed
                    const bool hit = a[u] == j[u] && o[u] > 0.f;
                    const float val = sc * g[u];
                    unsigned long long mask = __ballot(hit);
                    while (mask) {
                        const int b = __ffsll((long long)mask) - 1;
                        mask &= mask - 1;
                        const float sv = __shfl(val, b, 64);
                        const float *wrow = WLDS ? wlds + (cb + b) * CIN : W + (size_t)(cb + b) * D + 3;
#pragma unroll
                        for (int q = 0; q < NQ; ++q)
                            if (lane + 64 * q < CIN) acc[q] += sv * wrow[lane + 64 * q];
                    }
                }
            }
        }
        if (end > beg) {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (lane + 64 * q < CIN) gx[(size_t)v * CIN + lane + 64 * q] += acc[q];
        }
    }
}

// ---------------------------------------------------------------------------------------------------- backward: sparse weight gradient
// dW[c][col] += sum_m sum_k [k == arg[m,c]] s_c g'[m,c] in[(m,k)][col]:  a weight-gradient product whose reduction index runs over the
// m*16 rows without materialising them.  One block = a 64 x 64 block of dW (columns in feature-first order: [x (cin) | rel (3) | pad]).
__global__ __launch_bounds__(256) void k_td_wg(int M, int cin, int cout, const float *__restrict__ gout, const float *__restrict__ out,
                                               const unsigned char *__restrict__ arg, const int *__restrict__ idx, const float *__restrict__ rel4,
                                               const float *__restrict__ x, const float *__restrict__ coef, float *__restrict__ slab,
                                               long points_per_block) {
    __shared__ float red[4][64 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, nq = lane >> 4;
    const int nkb = (cin + 4 + 63) / 64;
    const int ob = (blockIdx.y / nkb) * 64, kb = (blockIdx.y % nkb) * 64;
    const int d = 3 + cin;
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = zero4();
    const long pb = (long)blockIdx.x * points_per_block;
    const long pe = pb + points_per_block < M ? pb + points_per_block : M;
    const f32x4 s4 = ld4(coef + ob + 4 * li);
    const int col = kb + 4 * li;   // first of this lane's four dW columns (feature-first order)
    for (long m0 = pb + 4 * wave; m0 < pe; m0 += 16) {   // 4 points per wave and trip (one per nq)
        const long m = m0 + nq;
        const bool ok = m < pe;
        f32x4 sg = zero4();
        unsigned a4 = 0xffffffffu;
        if (ok) {
            const f32x4 g = ld4(gout + (size_t)m * cout + ob + 4 * li), o = ld4(out + (size_t)m * cout + ob + 4 * li);
            a4 = *reinterpret_cast<const unsigned *>(arg + (size_t)m * cout + ob + 4 * li);
#pragma unroll
            for (int e = 0; e < 4; ++e) sg[e] = o[e] > 0.f ? s4[e] * g[e] : 0.f;
        }
        int nbs[16];   // the point's 16 neighbour indices up front: the gathers below then are independent loads
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int4 v = ok ? *reinterpret_cast<const int4 *>(idx + m * 16 + 4 * q) : make_int4(-1, -1, -1, -1);
            nbs[4 * q] = v.x; nbs[4 * q + 1] = v.y; nbs[4 * q + 2] = v.z; nbs[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int k0 = 0; k0 < 16; k0 += 4) {
            f32x4 xv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xv[u] = zero4();
                if (ok) {
                    if (col < cin) { if (nbs[k0 + u] >= 0) xv[u] = ld4(x + (size_t)nbs[k0 + u] * cin + col); }
                    else if (col == cin) xv[u] = ld4(rel4 + (m * 16 + k0 + u) * 4);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                f32x4 gv;
#pragma unroll
                for (int e = 0; e < 4; ++e) gv[e] = (int)((a4 >> (8 * e)) & 0xffu) == k0 + u ? sg[e] : 0.f;
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv[a], xv[u][b], acc[a][b], 0, 0, 0);
            }
        }
    }
    // D layout: acc[a][b][r] = dW[ob + 4 (4 nq + r) + a][kb + 4 li + b]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][(4 * (4 * nq + r) + a) * 64 + 4 * li + b] = acc[a][b][r];
    __syncthreads();
    // the workgroup's 64 x 64 block goes to its own slab [block of dW][row-range][64 * 64]; k_td_wg_reduce adds the slabs in order
    float *dst = slab + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (64 * 64);
    for (int e = threadIdx.x; e < 64 * 64; e += 256) dst[e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    (void)d;
}

// dW[o][wc] += sum over the row-range slabs (fixed order) of the 64 x 64 blocks; feature-first column cc -> W's column wc (rel first).
// dW already holds the dense part (k_td_small1).  grid = (64, blocks of dW), 256 threads = 64 elements x 4 slab-lanes.
__global__ __launch_bounds__(256) void k_td_wg_reduce(int cin, int cout, const float *__restrict__ slab, int split, float *__restrict__ dW) {
    __shared__ float red[4][64];
    const int e64 = threadIdx.x & 63, l = threadIdx.x >> 6;
    const int nkb = (cin + 4 + 63) / 64;
    const int ob = (blockIdx.y / nkb) * 64, kb = (blockIdx.y % nkb) * 64;
    const int e = blockIdx.x * 64 + e64;
    const float *src = slab + (size_t)blockIdx.y * split * (64 * 64) + e;
    float s = 0.f;
    for (int k = l; k < split; k += 4) s += src[(size_t)k * (64 * 64)];
    red[l][e64] = s;
    __syncthreads();
    if (l != 0) return;
    s = ((red[0][e64] + red[1][e64]) + red[2][e64]) + red[3][e64];
    const int o = ob + e / 64, cc = kb + e % 64;
    if (cc >= cin + 3) return;
    const int wc = cc < cin ? 3 + cc : cc - cin;
    dW[(size_t)o * (3 + cin) + wc] += s;
}

template <typename KernelT>
static void set_lds(KernelT kernel, size_t bytes) {
    if (bytes > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

static inline int grid_points(long m, int cap) {
    long g = (m + WPB - 1) / WPB;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace td

extern "C" int pdf_bn_bwd_sums(long n, int c, const float *gy, const float *x, const float *coef, int relu, float *partial, float *sums,
                               void *stream);

extern "C" int pdf_td_supported(int nsample, int cin, int cout) {
    return nsample == 16 && (cin == 32 || cin == 64 || cin == 128 || cin == 256) && cout % 64 == 0 && cout <= 1024;
}
// scratch floats of the forward (Gram blocks, kept for the backward): gxx (cin*cin) | gz (32*cin)
extern "C" long pdf_td_gram_floats(int cin) { return (long)cin * cin + 32L * cin; }
// split of the sparse weight-gradient kernel's point range (k_td_wg): workgroups in flight vs points per workgroup
static inline void td_wg_plan(long m, int cin, int cout, int *nblk, long *split, long *ppb) {
    *nblk = (cout / 64) * ((cin + 4 + 63) / 64);
    long sp = (512 + *nblk - 1) / *nblk;
    const long max_split = (m + 63) / 64;
    if (sp > max_split) sp = max_split;
    if (sp < 1) sp = 1;
    *ppb = ((m + sp - 1) / sp + 15) / 16 * 16;
    *split = (m + *ppb - 1) / *ppb;
}
// scratch floats of the forward: slabs of the two Gram products (pdf_rowlin_wgrad_ws_floats)
extern "C" long pdf_td_fwd_scratch_floats(long n, int cin) {
    const long a = pdf_rowlin_wgrad_ws_floats(n, cin, cin, 1), b = pdf_rowlin_wgrad_ws_floats(n, cin, 32, 1);
    return a > b ? a : b;
}
// scratch floats of the backward: fake coef (4 cout) | AB (2 cout) | Qp (cin*cin) | QqT (cin*32) | partial (pdf_bn_partial_floats(m, cout))
// | slabs of the sparse weight gradient (blocks x row ranges x 64 x 64)
extern "C" long pdf_td_bwd_scratch_floats(long m, int cin, int cout) {
    int nblk; long split, ppb;
    td_wg_plan(m, cin, cout, &nblk, &split, &ppb);
    return 6L * cout + (long)cin * cin + 32L * cin + pdf_bn_partial_floats(m, cout) + (long)nblk * split * 64 * 64;
}

// Forward.  p[]: 0 x (n,cin) | 1 idx (m,16) | 2 rel4 (m,16,4) | 3 Z (n,32) = [R | cnt | 0] | 4 consts (16) = [S_rr | s_r] | 5 W (cout, 3+cin)
//   6 gamma 7 beta 8 running_mean 9 running_var | outputs: 10 coef (4 cout) 11 out (m,cout) 12 arg (m,cout) u8 13 gram (pdf_td_gram_floats)
//   14 scratch (pdf_td_fwd_scratch_floats(n, cin); training only)
extern "C" int pdf_td_forward(long n, long m, int cin, int cout, void *const *p, int training, float eps, float momentum, int mma_input, void *stream) {
    if (n < 1 || m < 1 || !p) return PDF_ERR_BAD_ARG;
    if (!pdf_td_supported(16, cin, cout)) return PDF_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float *x = (const float *)p[0], *rel4 = (const float *)p[2], *Z = (const float *)p[3], *consts = (const float *)p[4];
    const float *W = (const float *)p[5], *gamma = (const float *)p[6], *beta = (const float *)p[7];
    const int *idx = (const int *)p[1];
    float *coef = (float *)p[10], *out = (float *)p[11], *gxx = (float *)p[13], *gz = gxx + (size_t)cin * cin;
    int rc = 0;
    if (training) {
        if (!p[14]) return PDF_ERR_BAD_ARG;
        rc = pdf_rowlin_wgrad_roww(n, cin, cin, x, cin, x, cin, gxx, Z + 3, 32, (float *)p[14], mma_input, stream);       // x^T diag(cnt) x (written)
        if (rc) return rc;
        rc = pdf_rowlin_wgrad_roww(n, cin, 32, Z, 32, x, cin, gz, nullptr, 0, (float *)p[14], mma_input, stream);        // [R | cnt]^T x
        if (rc) return rc;
        td::k_td_coef<<<cout, 256, 0, s>>>(cin, cout, (double)m * 16.0, W, consts, gz, gxx, gamma, beta, eps, momentum, (float *)p[8],
                                           (float *)p[9], coef);
    } else {
        fl::launch_bn_eval(cout, gamma, beta, eps, (const float *)p[8], (const float *)p[9], coef, coef + cout, coef + 2 * cout, coef + 3 * cout, s);
    }
    const dim3 grid(td::grid_points(m, 1024), cout / 64);
#define PDF_TD_FWD(CIN_) do { const size_t lds = sizeof(float) * (64 * (CIN_ + 4) + 128); td::set_lds(td::k_td_fwd<CIN_>, lds); \
        td::k_td_fwd<CIN_><<<grid, 64 * td::WPB, lds, s>>>((int)m, cout, x, idx, rel4, W, coef, out, (unsigned char *)p[12]); } while (0)
    if (cin == 32) PDF_TD_FWD(32); else if (cin == 64) PDF_TD_FWD(64); else if (cin == 128) PDF_TD_FWD(128); else PDF_TD_FWD(256);
#undef PDF_TD_FWD
    return pdf_launch_status();
}

// Backward (training mode).  p[]: 0 gout (m,cout) | 1 out | 2 arg | 3 x | 4 idx | 5 rel4 | 6 Z | 7 consts | 8 W | 9 gamma | 10 beta | 11 coef
//   12 gram (from the forward) | outputs: 13 gx (n,cin) 14 dW (cout, 3+cin) 15 dgb (2 cout) = [d beta | d gamma] | 16 scratch
//   17 inv_off (n + 1), 18 inv_entry: the INVERSE of the grouping table idx (entries i * 16 + j grouped by source point, ascending inside a
//   point; csrc/seg_gather.hip), entry ids offset by entry_base: the sparse input gradient then runs in destination order without
//   atomics (bit-reproducible); both null: the scatter kernel with float atomics (not reproducible run to run)
extern "C" int pdf_td_backward(long n, long m, int cin, int cout, void *const *p, int entry_base, int mma_input, void *stream) {
    if (n < 1 || m < 1 || !p) return PDF_ERR_BAD_ARG;
    if (!pdf_td_supported(16, cin, cout)) return PDF_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float *gout = (const float *)p[0], *out = (const float *)p[1], *x = (const float *)p[3], *rel4 = (const float *)p[5];
    const float *Z = (const float *)p[6], *consts = (const float *)p[7], *W = (const float *)p[8], *coef = (const float *)p[11];
    const unsigned char *arg = (const unsigned char *)p[2];
    const int *idx = (const int *)p[4];
    const float *gxx = (const float *)p[12], *gz = gxx + (size_t)cin * cin;
    float *gx = (float *)p[13], *dW = (float *)p[14], *dgb = (float *)p[15];
    float *fake = (float *)p[16], *AB = fake + 4 * cout, *Qp = AB + 2 * cout, *QqT = Qp + (size_t)cin * cin, *partial = QqT + 32 * (size_t)cin;
    const double rows = (double)m * 16.0;
    td::k_td_fake<<<(cout + 255) / 256, 256, 0, s>>>(cout, (const float *)p[9], (const float *)p[10], fake);
    int rc = pdf_bn_bwd_sums(m, cout, gout, out, fake, 1, partial, dgb, stream);                  // [S1 | S2] = [d beta | d gamma]
    if (rc) return rc;
    td::k_td_small1<<<cout, 256, 0, s>>>(cin, cout, rows, W, consts, gz, gxx, coef, dgb, AB, dW);
    td::k_td_small2<<<3 + cin + 1, 256, 0, s>>>(cin, cout, W, AB, Qp, QqT);
    rc = pdf_rowlin_forward_roww(n, cin, cin, x, cin, Qp, 0, gx, cin, 0, Z + 3, 32, mma_input, stream);     // gx = cnt .* (x Qp^T)
    if (rc) return rc;
    rc = pdf_rowlin_forward_roww(n, 32, cin, Z, 32, QqT, 0, gx, cin, 1, nullptr, 0, mma_input, stream);     // gx += Z QqT^T
    if (rc) return rc;
    if (p[17] && p[18]) {
        const int *inv_off = (const int *)p[17], *inv_entry = (const int *)p[18];
        const int g = td::grid_points(n, 2048);
#define PDF_TD_DST(CIN_, WLDS_) td::k_td_din_dst<CIN_, WLDS_><<<g, 64 * td::WPB, WLDS_ ? sizeof(float) * (size_t)cout * CIN_ : 0, s>>>( \
            (int)n, cout, gout, out, arg, W, coef, inv_off, inv_entry, entry_base, gx)
        const bool wlds = (size_t)cout * cin * sizeof(float) <= 48 * 1024;   // (the model's cout = 2 cin: 8 KB / 32 KB at cin = 32 / 64)
        if (cin == 32) { if (wlds) PDF_TD_DST(32, true); else PDF_TD_DST(32, false); }
        else if (cin == 64) { if (wlds) PDF_TD_DST(64, true); else PDF_TD_DST(64, false); }
        else if (cin == 128) PDF_TD_DST(128, false); else PDF_TD_DST(256, false);
#undef PDF_TD_DST
    } else {
        const dim3 grid(td::grid_points(m, 1024), cout / 64);
#define PDF_TD_DIN(CIN_) do { const size_t lds = sizeof(float) * (64 * (CIN_ + 4)) + sizeof(int) * 16 * td::WPB; td::set_lds(td::k_td_din<CIN_>, lds); \
        td::k_td_din<CIN_><<<grid, 64 * td::WPB, lds, s>>>((int)m, cout, gout, out, arg, idx, W, coef, gx); } while (0)
        if (cin == 32) PDF_TD_DIN(32); else if (cin == 64) PDF_TD_DIN(64); else if (cin == 128) PDF_TD_DIN(128); else PDF_TD_DIN(256);
#undef PDF_TD_DIN
    }
    int nblk; long split, ppb;
    td_wg_plan(m, cin, cout, &nblk, &split, &ppb);
    float *slab = partial + pdf_bn_partial_floats(m, cout);
    td::k_td_wg<<<dim3((unsigned)split, nblk), 256, 0, s>>>((int)m, cin, cout, gout, out, arg, idx, rel4, x, coef, slab, ppb);
    td::k_td_wg_reduce<<<dim3(64, nblk), 256, 0, s>>>(cin, cout, slab, (int)split, dW);
    return pdf_launch_status();
}

// ---------------------------------------------------------------------------------------------------- geometry-only tables
// One lane per (new point, neighbour) row: rel4 = masked relative coordinates; Z[nb][0..3] += [rel | 1] (Z zeroed by the caller).
// (The twelve per-scene moments of rel are formed by the host side from rel4 with a prefix sum: atomics onto a dozen
// addresses from 75k waves cost 20 ms.)
namespace td {
// Z in destination order (no atomics): one lane per source point walks its entries of the inverse table in ascending entry id.
__global__ __launch_bounds__(256) void k_td_z_dst(long n, const float *__restrict__ p_src, const float *__restrict__ p_new,
                                                  const int *__restrict__ inv_off, const int *__restrict__ inv_entry, int entry_base,
                                                  float *__restrict__ Z) {
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    if (v >= n) return;
    const int beg = inv_off[v], end = inv_off[v + 1];
    const float sx = p_src[v * 3], sy = p_src[v * 3 + 1], sz = p_src[v * 3 + 2];
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (int p = beg; p < end; ++p) {
        const long mq = (inv_entry[p] - entry_base) >> 4;
        ax += sx - p_new[mq * 3]; ay += sy - p_new[mq * 3 + 1]; az += sz - p_new[mq * 3 + 2];
    }
    *reinterpret_cast<f32x4 *>(Z + v * 32) = f32x4{ax, ay, az, (float)(end - beg)};
}

template <bool ZATOMIC>
__global__ __launch_bounds__(256) void k_td_tables(long m, int b, const float *__restrict__ p_src, const float *__restrict__ p_new,
                                                   const int *__restrict__ idx, const int *__restrict__ new_offset, float *__restrict__ rel4,
                                                   float *__restrict__ Z, float *__restrict__ scene_sums) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    const bool ok = r < m * 16;
    const long pt = ok ? r >> 4 : m - 1;
    float rel[3] = {0.f, 0.f, 0.f};
    if (ok) {
        const int nb = idx[r];
        if (nb >= 0) {
#pragma unroll
            for (int a = 0; a < 3; ++a) rel[a] = p_src[(size_t)nb * 3 + a] - p_new[(size_t)pt * 3 + a];
            if (ZATOMIC) {
                float *z = Z + (size_t)nb * 32;
                pdf_atomic_add(z + 0, rel[0]); pdf_atomic_add(z + 1, rel[1]); pdf_atomic_add(z + 2, rel[2]); pdf_atomic_add(z + 3, 1.f);
            }
        }
        *reinterpret_cast<f32x4 *>(rel4 + r * 4) = f32x4{rel[0], rel[1], rel[2], 0.f};
    }
}
}  // namespace td

// rel4 (m,16,4), Z (n,32) and scene_sums (b,16) -- the last two zeroed by the caller.  With the inverse of idx (inv_off (n + 1), inv_entry,
// entry_base; csrc/seg_gather.hip) Z is summed in destination order (fixed order, bit-reproducible); without it (both null) by float atomics.
extern "C" int pdf_td_tables(long m, int b, const float *p_src, const float *p_new, const int *idx, const int *new_offset, float *rel4, float *Z,
                             float *scene_sums, long n, const int *inv_off, const int *inv_entry, int entry_base, void *stream) {
    if (m < 1 || b < 1 || !p_src || !p_new || !idx || !new_offset || !rel4 || !Z || !scene_sums) return PDF_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (inv_off && inv_entry) {
        if (n < 1) return PDF_ERR_BAD_ARG;
        td::k_td_tables<false><<<(unsigned)((m * 16 + 255) / 256), 256, 0, s>>>(m, b, p_src, p_new, idx, new_offset, rel4, Z, scene_sums);
        td::k_td_z_dst<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(n, p_src, p_new, inv_off, inv_entry, entry_base, Z);
    } else {
        td::k_td_tables<true><<<(unsigned)((m * 16 + 255) / 256), 256, 0, s>>>(m, b, p_src, p_new, idx, new_offset, rel4, Z, scene_sums);
    }
    return pdf_launch_status();
}
