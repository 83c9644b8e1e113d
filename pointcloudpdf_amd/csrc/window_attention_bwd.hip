// Atomic-free backward of the libs/pointops2 window-attention ops (BASELINE config 5) for gfx950.
//
// Reference: libs/pointops2/src/attention_v2/attention_cuda_kernel_v2.cu:50-93 (attention_step1 backward) and
// libs/pointops2/src/rpe_v2/relative_pos_encoding_cuda_kernel_v2.cu:287-340, 441-484 (dot_prod_with_idx_v3 / attention_step2_with_rel_pos_value_v2
// backward): one atomicAdd per (edge, channel) into grad_k / grad_v (rows shared between queries) and three per (edge, channel) into every
// relative-position table (L ~ 64 rows receive millions of edges).  Rounds 1-4 kept fp32 atomics for exactly those sums
// (csrc/window_attention.hip): 31 of the 67 ms of the ST-v1m1 step, and a step that is not bit-reproducible.
//
// Every one of those sums is a SEGMENTED sum once the edge list is also grouped by KEY.  The edges arrive in CSR order by query
// (index0_offsets); the coordinate pre-pass adds the transposed list (stable sort of index1: ascending edge id inside a key, hence a fixed
// summation order) with the per-edge integers it needs permuted alongside:
//     key_off (N + 1)   key_edge (M): edge ids grouped by key   key_q (M): query of that edge   key_rel (M, 3): rel_idx of that edge
// and the backward becomes two kernel shapes, used from either side:
//
//   rows  (pdf_wa_segment_rows):   out[n, c] = sum_{e in seg(n)} w[eid(e), head(c)] * ( X[other(e), c] + T(rel(e))[c] )
//         query side (CSR, eid(e) = e):        grad_q = sum g * (k[index1] + T_q)          (step1 + dot_prod)
//         key side   (CSC, eid = key_edge):    grad_k = sum g * (q[key_q] + T_k),   grad_v = sum attn * grad_out[key_q]
//         lane = (segment owner, 16-byte piece of a 3-head group of its row); the table slabs of the head group sit in LDS, transposed
//         to [axis][row][channel]; 8 entries of a segment in flight (ids, then rows + weights); one store per owner row.  Bound: the row
//         gathers (M x C x 4 bytes per pass, served by L2 / Infinity Cache: the row table is N x C x 4 <= 31 MB).
//   table (pdf_wa_table_grad):     G[r, c, a] = sum_n x[n, c] * S_n[a][r][head(c)],   S_n[a][r][hh] = sum_{e in seg(n), rel(e)[a] = r} w[eid(e), hh]
//         i.e. the table gradient FACTORS through a per-owner histogram of the edge scalars whenever the row that multiplies the one-hot
//         selector is the segment owner's: q[query] for T_q and grad_out[query] for T_v over the CSR list, k[key] for T_k over the CSC list.
//         One lane per (owner, head, axis) walks the owner's segment and adds into its PRIVATE histogram row in LDS (program order: no
//         atomics, no races); 192 lanes then own one (axis, row) pair each and accumulate the 48 channels of the head group in
//         registers across all owner chunks of the (persistent) workgroup; workgroups leave partial slabs that a second kernel sums in
//         a fixed order.  Bit-reproducible, and no pre-zeroed outputs.
//
// Needs head dim d = 16 and h % 3 == 0 or h == 1, 2 (head groups of 3 / 1); table length L <= 64.  Other shapes keep the kernels of
// window_attention.hip (PDF_ERR_UNSUPPORTED from here; the caller falls back).
#include "pdfops_common.h"

namespace wb {

constexpr int TB = 256;
constexpr int SB = 8;      // entries of a segment in flight per trip
constexpr int D = 16;      // head dim
constexpr int LMAX = 64;

__device__ __forceinline__ float4 f4(float x) { return make_float4(x, x, x, x); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 fma4(float4 a, float s, float4 acc) { return make_float4(acc.x + a.x * s, acc.y + a.y * s, acc.z + a.z * s, acc.w + a.w * s); }

// table (L, h, 16, 3) -> LDS [axis][row][HG * 16] for the heads [h0, h0 + HG)
template <int HG>
__device__ __forceinline__ void stage_table(float *dst, const float *__restrict__ table, int L, int h, int h0) {
    constexpr int CG = HG * D;
    for (int e = threadIdx.x; e < L * CG * 3; e += TB) {
        const int r = e / (CG * 3), x = e - r * (CG * 3), c = x / 3, a = x - 3 * c;
        dst[(a * L + r) * CG + c] = table[((size_t)r * h * D + (size_t)h0 * D + c) * 3 + a];
    }
}

// ---------------------------------------------------------------------------------------------------------------- rows
// grid = (ceil(N / RPB), h / HG); RPB = TB / (HG * 4) owner rows per workgroup; lane = (row slot, 16-byte piece p of the head group's
// HG * 16 channels).  w: (M, h) edge scalars; X: (N, h, 16) rows or null; table: (L, h, 16, 3) or null.
template <int HG, bool CSC, bool ROWS, bool TABLE>
__global__ __launch_bounds__(TB) void k_rows(int N, int h, int L, const int *__restrict__ seg_off, const int *__restrict__ seg_edge,
                                             const int *__restrict__ other, const int *__restrict__ rel, const float *__restrict__ w,
                                             const float *__restrict__ X, long ldx, float xscale, const float *__restrict__ table,
                                             float *__restrict__ out, long ldo, float oscale) {
    constexpr int CG = HG * D, PQ = CG / 4, RPB = TB / PQ;
    extern __shared__ __attribute__((aligned(16))) float tl[];   // [3][L][CG]
    const int h0 = blockIdx.y * HG;
    if (TABLE) {
        stage_table<HG>(tl, table, L, h, h0);
        __syncthreads();
    }
    const int slot = threadIdx.x / PQ, p = threadIdx.x - slot * PQ;
    const long n = (long)blockIdx.x * RPB + slot;
    if (slot >= RPB || n >= N) return;
    const int hh = h0 + p / 4;                       // this lane's head
    const size_t col = (size_t)h0 * D + 4 * p;
    int t = seg_off[n];
    const int end = seg_off[n + 1];
    float4 a0 = f4(0.f), a1 = f4(0.f);
    for (; t < end; t += SB) {
        int eid[SB], oth[SB], r[SB][3];
        float ww[SB];
        float4 x[SB];
#pragma unroll
        for (int k = 0; k < SB; ++k) {
            const int e = min(t + k, end - 1);
            eid[k] = CSC ? seg_edge[e] : e;
            if (ROWS) oth[k] = other[e];
            if (TABLE) { r[k][0] = rel[(size_t)e * 3]; r[k][1] = rel[(size_t)e * 3 + 1]; r[k][2] = rel[(size_t)e * 3 + 2]; }
        }
#pragma unroll
        for (int k = 0; k < SB; ++k) {
            ww[k] = w[(size_t)eid[k] * h + hh];
            if (ROWS) x[k] = *reinterpret_cast<const float4 *>(X + (size_t)oth[k] * ldx + col);
        }
#pragma unroll
        for (int k = 0; k < SB; ++k) {
            float4 v = ROWS ? make_float4(x[k].x * xscale, x[k].y * xscale, x[k].z * xscale, x[k].w * xscale) : f4(0.f);
            if (TABLE) {
                const float4 t0 = *reinterpret_cast<const float4 *>(tl + (0 * L + r[k][0]) * CG + 4 * p);
                const float4 t1 = *reinterpret_cast<const float4 *>(tl + (1 * L + r[k][1]) * CG + 4 * p);
                const float4 t2 = *reinterpret_cast<const float4 *>(tl + (2 * L + r[k][2]) * CG + 4 * p);
                v = add4(v, add4(add4(t0, t1), t2));            // (t[r1,.,0] + t[r2,.,1]) + t[r3,.,2], as upstream writes it
            }
            const float s = t + k < end ? ww[k] : 0.f;
            if (k & 1) a1 = fma4(v, s, a1); else a0 = fma4(v, s, a0);
        }
    }
    const float4 r4 = add4(a0, a1);
    *reinterpret_cast<float4 *>(out + (size_t)n * ldo + col) = make_float4(r4.x * oscale, r4.y * oscale, r4.z * oscale, r4.w * oscale);
}

// ---------------------------------------------------------------------------------------------------------------- table gradients
constexpr int OC = 24;              // owners per chunk (histogram rows in LDS: OC x 9 x (L + 1) floats)
constexpr int HS = LMAX + 1;        // histogram row stride (odd: lanes of different owners spread over the banks)

// grid = (G, h / HG): workgroup (g, hg) walks the owner chunks g, g + G, ...; partial[(g * HGN + hg)][a][r][c] (3 x L x CG floats)
template <int HG, bool CSC>
__global__ __launch_bounds__(TB) void k_table(int N, int h, int L, const int *__restrict__ seg_off, const int *__restrict__ seg_edge,
                                              const int *__restrict__ rel, const float *__restrict__ w, const float *__restrict__ x,
                                              long ldx, float xscale, float *__restrict__ partial) {
    constexpr int CG = HG * D, NP = HG * 3;            // NP histogram rows per owner
    __shared__ __attribute__((aligned(16))) float S[OC * NP * HS];
    __shared__ __attribute__((aligned(16))) float xs[OC * CG];
    const int h0 = blockIdx.y * HG;
    // phase-2 role: thread ar < 3 L owns (axis, row) = (ar / L, ar % L) and the CG channels of the head group
    const int ar = threadIdx.x, pa = ar / L, pr = ar - pa * L;
    const bool owner2 = ar < 3 * L;
    float acc[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) acc[c] = 0.f;
    // phase-1 role: lane (slot, hh, axis) walks the segment of owner chunk0 + slot
    const int slot = threadIdx.x / NP, j = threadIdx.x - slot * NP, hh = j / 3, ax = j - 3 * hh;
    const bool owner1 = slot < OC;
    const int nchunks = (N + OC - 1) / OC;
    for (int ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        const int n0 = ch * OC, cnt = min(OC, N - n0);
        for (int e = threadIdx.x; e < OC * NP * HS; e += TB) S[e] = 0.f;
        for (int e = threadIdx.x; e < OC * CG; e += TB) {
            const int s = e / CG, c = e - s * CG;
            xs[e] = s < cnt ? x[(size_t)(n0 + s) * ldx + (size_t)h0 * D + c] * xscale : 0.f;
        }
        __syncthreads();
        if (owner1 && slot < cnt) {
            float *row = S + (slot * NP + j) * HS;
            int t = seg_off[n0 + slot];
            const int end = seg_off[n0 + slot + 1];
            for (; t < end; t += SB) {
                int eid[SB], rr[SB];
                float ww[SB];
#pragma unroll
                for (int k = 0; k < SB; ++k) {
                    const int e = min(t + k, end - 1);
                    eid[k] = CSC ? seg_edge[e] : e;
                    rr[k] = rel[(size_t)e * 3 + ax];
                }
#pragma unroll
                for (int k = 0; k < SB; ++k) ww[k] = w[(size_t)eid[k] * h + h0 + hh];
#pragma unroll
                for (int k = 0; k < SB; ++k)
                    if (t + k < end) row[rr[k]] += ww[k];       // private row, program order: a fixed summation order without atomics
            }
        }
        __syncthreads();
        if (owner2) {
            for (int s = 0; s < cnt; ++s) {
                float sv[HG];
#pragma unroll
                for (int g = 0; g < HG; ++g) sv[g] = S[(s * NP + g * 3 + pa) * HS + pr];
                const float4 *xr = reinterpret_cast<const float4 *>(xs + s * CG);   // (same address in every lane: broadcast reads)
#pragma unroll
                for (int q = 0; q < CG / 4; ++q) {
                    const float4 v = xr[q];
                    const float f = sv[q / 4];
                    acc[4 * q] += v.x * f; acc[4 * q + 1] += v.y * f; acc[4 * q + 2] += v.z * f; acc[4 * q + 3] += v.w * f;
                }
            }
        }
        __syncthreads();
    }
    if (owner2) {
        float4 *dst = reinterpret_cast<float4 *>(partial + (((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 3 * L + ar) * CG);
#pragma unroll
        for (int q = 0; q < CG / 4; ++q) dst[q] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    }
}

// grad_table[r, h0 + c / 16, c % 16, a] = sum_g partial[g][hg][a][r][c]   (fixed order; element = one thread)
__global__ __launch_bounds__(TB) void k_table_reduce(int G, int HGN, int HGsz, int L, int h, const float *__restrict__ partial, float *__restrict__ grad_table) {
    const int CG = HGsz * D;
    const long total = (long)HGN * 3 * L * CG;
    const long e = (long)blockIdx.x * TB + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % CG);
    const long rest = e / CG;
    const int r = (int)(rest % L), a = (int)((rest / L) % 3), hg = (int)(rest / (3L * L));
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int g = 0;
    for (; g + 3 < G; g += 4) {
        s0 += partial[(size_t)g * total + e]; s1 += partial[(size_t)(g + 1) * total + e];
        s2 += partial[(size_t)(g + 2) * total + e]; s3 += partial[(size_t)(g + 3) * total + e];
    }
    for (; g < G; ++g) s0 += partial[(size_t)g * total + e];
    grad_table[((size_t)r * h * D + (size_t)hg * CG + c) * 3 + a] = (s0 + s1) + (s2 + s3);
}

// ---------------------------------------------------------------------------------------------------------------- grad_attn
// attention_step2_with_rel_pos_value_v2 backward, the edge-indexed result (relative_pos_encoding_cuda_kernel_v2.cu:441-470):
//     grad_attn[m, hh] = < grad_out[q(m), hh, :], v[index1[m], hh, :] + T(m, hh, :) >
// The table term factors through the projection P_q[a][r] = < table[r, hh, :, a], grad_out[q, hh, :] > (3 L dot products per query instead
// of 3 x 16 multiply-adds per edge).  grid = (ceil(N / QF), h); lane = edge (value row: 4 x 16 bytes contiguous).
constexpr int QF = 32;
__global__ __launch_bounds__(TB) void k_grad_attn(int N, int h, int L, const float *__restrict__ go, long ldg, const int *__restrict__ offsets,
                                                  const int *__restrict__ index1, const float *__restrict__ v, long ldv, const float *__restrict__ table,
                                                  const int *__restrict__ rel, float *__restrict__ grad_attn) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *tb = sm, *gr = tb + 3 * L * D, *P = gr + QF * D;     // tb [3][L][16] | gr [QF][16] | P [QF][3 L]
    int *offs = reinterpret_cast<int *>(P + QF * 3 * L);
    const int hh = blockIdx.y, q0 = blockIdx.x * QF, nq = min(QF, N - q0), W = 3 * L;
    for (int jj = threadIdx.x; jj <= nq; jj += TB) offs[jj] = offsets[q0 + jj];
    stage_table<1>(tb, table, L, h, hh);
    for (int e = threadIdx.x; e < QF * D; e += TB) gr[e] = e < nq * D ? go[(size_t)(q0 + e / D) * ldg + (size_t)hh * D + e % D] : 0.f;
    __syncthreads();
    for (int e = threadIdx.x; e < nq * W; e += TB) {
        const int ql = e / W, x = e - ql * W;   // x = a * L + r
        const float4 *t4 = reinterpret_cast<const float4 *>(tb + x * D), *g4 = reinterpret_cast<const float4 *>(gr + ql * D);
        float acc = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < D / 4; ++c4) { const float4 a = t4[c4], b = g4[c4]; acc += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
        P[e] = acc;
    }
    __syncthreads();
    const int e0 = offs[0], e1 = offs[nq];
    int ql = 0;
    for (int m = e0 + threadIdx.x; m < e1; m += TB) {           // (a lane's edges ascend: the query is found by stepping on)
        while (offs[ql + 1] <= m) ++ql;
        const int r1 = rel[(size_t)m * 3], r2 = rel[(size_t)m * 3 + 1], r3 = rel[(size_t)m * 3 + 2];
        const float4 *vr = reinterpret_cast<const float4 *>(v + (size_t)index1[m] * ldv + (size_t)hh * D);
        const float4 *g4 = reinterpret_cast<const float4 *>(gr + ql * D);
        float dot = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < D / 4; ++c4) { const float4 a = vr[c4], b = g4[c4]; dot += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
        grad_attn[(size_t)m * h + hh] = dot + ((P[ql * W + r1] + P[ql * W + L + r2]) + P[ql * W + 2 * L + r3]);
    }
}

// ---------------------------------------------------------------------------------------------------------------- fused logits
// attention_step1_v2 + dot_prod_with_idx_v3 in one pass (WindowAttention.forward adds the two, stratified_transformer_v1m1_origin.py:
// 300-321): logit[m, hh] = sum_i q[q(m), hh, i] * (k[j, hh, i] + T_q(m, hh, i)) + k[j, hh, i] * T_k(m, hh, i),  j = index1[m].
// One gather of the key row instead of two and no (M, h) addition.  grid = (ceil(N / QL), h); lane = edge; the head's two table slabs
// and the chunk's query rows sit in LDS.
constexpr int QL = 64;
__global__ __launch_bounds__(TB) void k_logits_fwd(int N, int h, int L, const float *__restrict__ q, const float *__restrict__ k, long ld, float qscale,
                                                   const int *__restrict__ offsets, const int *__restrict__ index1,
                                                   const float *__restrict__ table_q, const float *__restrict__ table_k,
                                                   const int *__restrict__ rel, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *tq = sm, *tk = tq + 3 * L * D, *qs = tk + 3 * L * D;     // [3][L][16] x 2 | qs [QL][16]
    int *offs = reinterpret_cast<int *>(qs + QL * D);
    const int hh = blockIdx.y, q0 = blockIdx.x * QL, nq = min(QL, N - q0);
    for (int jj = threadIdx.x; jj <= nq; jj += TB) offs[jj] = offsets[q0 + jj];
    stage_table<1>(tq, table_q, L, h, hh);
    stage_table<1>(tk, table_k, L, h, hh);
    for (int e = threadIdx.x; e < QL * D; e += TB) qs[e] = e < nq * D ? q[(size_t)(q0 + e / D) * ld + (size_t)hh * D + e % D] * qscale : 0.f;
    __syncthreads();
    const int e0 = offs[0], e1 = offs[nq];
    int ql = 0;
    for (int m = e0 + threadIdx.x; m < e1; m += TB) {
        while (offs[ql + 1] <= m) ++ql;
        const int r1 = rel[(size_t)m * 3], r2 = rel[(size_t)m * 3 + 1], r3 = rel[(size_t)m * 3 + 2];
        const float4 *kr = reinterpret_cast<const float4 *>(k + (size_t)index1[m] * ld + (size_t)hh * D);
        const float4 *q4 = reinterpret_cast<const float4 *>(qs + ql * D);
        const float4 *a0 = reinterpret_cast<const float4 *>(tq + (0 * L + r1) * D), *a1 = reinterpret_cast<const float4 *>(tq + (1 * L + r2) * D);
        const float4 *a2 = reinterpret_cast<const float4 *>(tq + (2 * L + r3) * D);
        const float4 *b0 = reinterpret_cast<const float4 *>(tk + (0 * L + r1) * D), *b1 = reinterpret_cast<const float4 *>(tk + (1 * L + r2) * D);
        const float4 *b2 = reinterpret_cast<const float4 *>(tk + (2 * L + r3) * D);
        float sum = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < D / 4; ++c4) {
            const float4 kv = kr[c4], qv = q4[c4];
            const float4 tqv = add4(add4(a0[c4], a1[c4]), a2[c4]), tkv = add4(add4(b0[c4], b1[c4]), b2[c4]);
            sum += qv.x * (kv.x + tqv.x) + kv.x * tkv.x;
            sum += qv.y * (kv.y + tqv.y) + kv.y * tkv.y;
            sum += qv.z * (kv.z + tqv.z) + kv.z * tkv.z;
            sum += qv.w * (kv.w + tqv.w) + kv.w * tkv.w;
        }
        out[(size_t)m * h + hh] = sum;
    }
}

static inline int head_group(int h) { return h % 3 == 0 ? 3 : 1; }
static inline int table_grid(int N, int h) {
    const int nchunks = (N + OC - 1) / OC, hgn = h / head_group(h);
    int g = 512 / (hgn < 1 ? 1 : hgn);          // ~512 workgroups in flight over all head groups
    if (g < 8) g = 8;
    if (g > nchunks) g = nchunks;
    return g < 1 ? 1 : g;
}

}  // namespace wb

// out (N, h, 16) = segmented sums over the entries [seg_off[n], seg_off[n + 1]) of owner n:
//     out[n, c] = oscale * sum_e w[eid(e), c / 16] * ( (X ? xscale * X[other[e], c] : 0) + (table ? T(rel[e])[c] : 0) ),   eid(e) = seg_edge ? seg_edge[e] : e
// ldx / ldo: row strides (floats) of X and out -- rows may be slices of wider rows (q / k / v inside the (N, 3 C) output of the qkv Linear).
// other / rel are given IN SEGMENT ORDER (the CSR arrays themselves, or the permuted copies of the CSC list).  Bytes: 4 M C per gathered
// row table + (8 + 12 + 4 h) M of integers / scalars + 4 N C written.
extern "C" int pdf_wa_segment_rows(int N, int h, int d, int L, const int *seg_off, const int *seg_edge, const int *other, const int *rel,
                                   const float *w, const float *X, long ldx, float xscale, const float *table, float *out, long ldo, float oscale,
                                   void *stream) {
    if (N < 0 || h < 1 || d < 1 || !seg_off || !w || !out || (X && !other) || (table && (!rel || L < 1))) return PDF_ERR_BAD_ARG;
    if (N == 0) return PDF_OK;
    if (d != wb::D || (table && L > wb::LMAX) || (!X && !table)) return PDF_ERR_UNSUPPORTED;
    if ((X && ((reinterpret_cast<uintptr_t>(X) & 15) || (ldx & 3) || ldx < (long)h * d)) || (reinterpret_cast<uintptr_t>(out) & 15) || (ldo & 3) || ldo < (long)h * d)
        return PDF_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int hg = wb::head_group(h);
#define PDF_WB_ROWS(HG_, CSC_, ROWS_, TABLE_) do { \
        constexpr int rpb = wb::TB / (HG_ * 4); \
        const size_t lds = TABLE_ ? sizeof(float) * 3 * (size_t)L * HG_ * wb::D : 0; \
        wb::k_rows<HG_, CSC_, ROWS_, TABLE_><<<dim3((unsigned)((N + rpb - 1) / rpb), (unsigned)(h / HG_)), wb::TB, lds, s>>>( \
            N, h, L, seg_off, seg_edge, other, rel, w, X, ldx, xscale, table, out, ldo, oscale); } while (0)
#define PDF_WB_ROWS2(HG_, CSC_) do { \
        if (X && table) PDF_WB_ROWS(HG_, CSC_, true, true); else if (X) PDF_WB_ROWS(HG_, CSC_, true, false); else PDF_WB_ROWS(HG_, CSC_, false, true); } while (0)
    if (hg == 3) { if (seg_edge) PDF_WB_ROWS2(3, true); else PDF_WB_ROWS2(3, false); }
    else { if (seg_edge) PDF_WB_ROWS2(1, true); else PDF_WB_ROWS2(1, false); }
#undef PDF_WB_ROWS2
#undef PDF_WB_ROWS
    return pdf_launch_status();
}

// grad_attn (M, h), every element written:  < grad_out[q(m), hh, :], v[index1[m], hh, :] + T(m, hh, :) >
extern "C" int pdf_wa_grad_attn(int N, int M, int h, int d, int L, const float *grad_out, long ldg, const int *offsets, const int *index1,
                                const float *v, long ldv, const float *table, const int *rel, float *grad_attn, void *stream) {
    if (N < 0 || M < 0 || h < 1 || d < 1 || L < 1 || !grad_out || !offsets || !index1 || !v || !table || !rel || !grad_attn) return PDF_ERR_BAD_ARG;
    if (N == 0 || M == 0) return PDF_OK;
    if (d != wb::D || L > wb::LMAX || (reinterpret_cast<uintptr_t>(v) & 15) || (ldv & 3) || ldv < (long)h * d || ldg < (long)h * d) return PDF_ERR_UNSUPPORTED;
    const size_t lds = sizeof(float) * (size_t)(3 * L * wb::D + wb::QF * wb::D + wb::QF * 3 * L) + sizeof(int) * (wb::QF + 1);
    wb::k_grad_attn<<<dim3((unsigned)((N + wb::QF - 1) / wb::QF), (unsigned)h), wb::TB, lds, static_cast<hipStream_t>(stream)>>>(
        N, h, L, grad_out, ldg, offsets, index1, v, ldv, table, rel, grad_attn);
    return pdf_launch_status();
}

// logits (M, h) = attention_step1_v2(q, k) + dot_prod_with_idx_v3(q, k, table_q, table_k), every element written
extern "C" int pdf_wa_logits_forward(int N, int M, int h, int d, int L, const float *q, const float *k, long ld, float qscale, const int *offsets,
                                     const int *index1, const float *table_q, const float *table_k, const int *rel, float *out, void *stream) {
    if (N < 0 || M < 0 || h < 1 || d < 1 || L < 1 || !q || !k || !offsets || !index1 || !table_q || !table_k || !rel || !out) return PDF_ERR_BAD_ARG;
    if (N == 0 || M == 0) return PDF_OK;
    if (d != wb::D || L > wb::LMAX || (reinterpret_cast<uintptr_t>(k) & 15) || (ld & 3) || ld < (long)h * d) return PDF_ERR_UNSUPPORTED;
    const size_t lds = sizeof(float) * (size_t)(6 * L * wb::D + wb::QL * wb::D) + sizeof(int) * (wb::QL + 1);
    wb::k_logits_fwd<<<dim3((unsigned)((N + wb::QL - 1) / wb::QL), (unsigned)h), wb::TB, lds, static_cast<hipStream_t>(stream)>>>(
        N, h, L, q, k, ld, qscale, offsets, index1, table_q, table_k, rel, out);
    return pdf_launch_status();
}

extern "C" long pdf_wa_table_grad_ws_floats(int N, int h, int L) {
    if (N < 1 || h < 1 || L < 1) return 0;
    return (long)wb::table_grid(N, h) * h * wb::D * 3 * L;
}

// grad_table (L, h, 16, 3), WRITTEN:  G[r, c, a] = sum_n x[n, c] * sum_{e in seg(n), rel[e][a] == r} w[eid(e), c / 16]
// x (N, h, 16): the row of the segment OWNER (q / grad_out over the CSR list, k over the CSC list); ws: pdf_wa_table_grad_ws_floats floats.
extern "C" int pdf_wa_table_grad(int N, int h, int d, int L, const int *seg_off, const int *seg_edge, const int *rel, const float *w,
                                 const float *x, long ldx, float xscale, float *ws, float *grad_table, void *stream) {
    if (N < 0 || h < 1 || d < 1 || L < 1 || !seg_off || !rel || !w || !x || !ws || !grad_table) return PDF_ERR_BAD_ARG;
    if (d != wb::D || L > wb::LMAX || ldx < (long)h * d) return PDF_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (N == 0) return hipMemsetAsync(grad_table, 0, sizeof(float) * (size_t)L * h * d * 3, s) == hipSuccess ? PDF_OK : PDF_ERR_BAD_ARG;
    const int hg = wb::head_group(h), hgn = h / hg, G = wb::table_grid(N, h);
    const dim3 grid((unsigned)G, (unsigned)hgn);
    if (hg == 3) {
        if (seg_edge) wb::k_table<3, true><<<grid, wb::TB, 0, s>>>(N, h, L, seg_off, seg_edge, rel, w, x, ldx, xscale, ws);
        else wb::k_table<3, false><<<grid, wb::TB, 0, s>>>(N, h, L, seg_off, seg_edge, rel, w, x, ldx, xscale, ws);
    } else {
        if (seg_edge) wb::k_table<1, true><<<grid, wb::TB, 0, s>>>(N, h, L, seg_off, seg_edge, rel, w, x, ldx, xscale, ws);
        else wb::k_table<1, false><<<grid, wb::TB, 0, s>>>(N, h, L, seg_off, seg_edge, rel, w, x, ldx, xscale, ws);
    }
    const long total = (long)hgn * 3 * L * hg * wb::D;
    wb::k_table_reduce<<<(unsigned)((total + wb::TB - 1) / wb::TB), wb::TB, 0, s>>>(G, hgn, hg, L, h, ws, grad_table);
    return pdf_launch_status();
}
