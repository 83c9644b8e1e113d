// Window-attention ops of libs/pointops2 for gfx950 (the CSR-by-query "v2 / v3" kernels that
// StratifiedTransformer's WindowAttention calls, stratified_transformer_v1m1_origin.py:277-341):
//   attention_step1_v2                        attn[m,h]   = <q[q(m),h,:], k[index1[m],h,:]>
//   dot_prod_with_idx_v3                      out[m,h]    = <q[q(m),h,:], Tq(m,h,:)> + <k[index_k[m],h,:], Tk(m,h,:)>
//   attention_step2_with_rel_pos_value_v2     out[q,h,:]  = sum_{m in q} attn[m,h] * (v[index1[m],h,:] + T(m,h,:))
// with T(m,h,i) = table[r1,h,i,0] + table[r2,h,i,1] + table[r3,h,i,2], (r1,r2,r3) = rel_idx[m,:], edges of query q =
// [offsets[q], offsets[q+1]).  Replaces libs/pointops2/src/attention_v2/attention_cuda_kernel_v2.cu:7-93 and
// libs/pointops2/src/rpe_v2/relative_pos_encoding_cuda_kernel_v2.cu:247-527.
//
// The reference launches a (query, head) block of n_max threads, one thread per edge, each walking its d = 16 / 32 channels
// with stride-C gathers, and reduces through shared-memory atomics.  Here one 256-lane workgroup owns a QUERY (all heads):
//   * edge-indexed results (attn, out[m,h], grad_attn): lane = (edge, head) with the head fastest, so a wave reads whole
//     contiguous key rows (h*d = C floats per edge) and writes consecutive outputs;
//   * query-indexed results (out[q,:], grad_q): lane = channel, loop over the query's edges -- coalesced row reads, no
//     atomics, fixed summation order (the reference's shared atomics are unordered);
//   * true scatters (grad_k, grad_v: rows shared between queries) are fp32 atomics, one float per lane over whole rows;
//   * table gradients collide massively (L ~ 50 rows receive millions of edges): see the per-head kernels further down (table
//     slabs in LDS, gradients as one-hot MFMA products); the query-owned kernels here are the fallback for other shapes.
// d is any positive size (the reference throws unless d is 16 or 32); n_max is accepted for signature parity and unused.
// All kernels: HBM / atomic-rate bound; algorithmic bytes are listed at the entry points.
#include "pdfops_common.h"
#include <cstdlib>

namespace {

constexpr int WB = 256;

__device__ __forceinline__ float table_sum(const float *__restrict__ t, int r1, int r2, int r3, int C, int c) {
    // table layout (L, h, d, 3); summation order as written upstream: (t[r1,.,0] + t[r2,.,1]) + t[r3,.,2]
    return t[((size_t)r1 * C + c) * 3] + t[((size_t)r2 * C + c) * 3 + 1] + t[((size_t)r3 * C + c) * 3 + 2];
}

// ---------------------------------------------------------------- attention_step1_v2
// attention_cuda_kernel_v2.cu:7-48
__global__ __launch_bounds__(WB) void k_step1_fwd(int h, int d, const float *__restrict__ q, const float *__restrict__ k,
                                                  const int *__restrict__ offsets, const int *__restrict__ index1,
                                                  float *__restrict__ attn) {
    extern __shared__ float qv[];
    const int qi = blockIdx.x, C = h * d;
    const int start = offsets[qi], end = offsets[qi + 1];
    if (end <= start) return;
    for (int c = threadIdx.x; c < C; c += WB) qv[c] = q[(size_t)qi * C + c];
    __syncthreads();
    const int total = (end - start) * h;
    for (int e = threadIdx.x; e < total; e += WB) {
        const int m = start + e / h, hh = e % h;
        const float *kr = k + (size_t)index1[m] * C + hh * d;
        const float *qr = qv + hh * d;
        float sum = 0.f;
        for (int i = 0; i < d; ++i) sum += qr[i] * kr[i];
        attn[(size_t)m * h + hh] = sum;
    }
}

// attention_cuda_kernel_v2.cu:50-93.  grad_q[q,:] = sum_m go[m,h] k[index1[m],:] (written, not accumulated);
// grad_k[index1[m],:] += go[m,h] q[q,:] (pre-zeroed scatter target).
// grad_q: S = WB / C edge slots walk the query's edges side by side (slot s takes edges start + s, start + s + S, ...: S dependent gather
// chains in flight instead of one with C <= 128 of the 256 lanes idle -- the one-lane-per-channel walk took 466 us per launch of the ST-v1m1
// step), the slots' partial rows are added in slot order by the first C lanes: a fixed order.  C > WB: channel chunks, one slot.
__global__ __launch_bounds__(WB) void k_step1_bwd(int h, int d, const float *__restrict__ go, const int *__restrict__ offsets,
                                                  const int *__restrict__ index1, const float *__restrict__ q,
                                                  const float *__restrict__ k, float *__restrict__ grad_q,
                                                  float *__restrict__ grad_k) {
    extern __shared__ float qv[];          // [C] q row | [S * C] partial rows of grad_q
    const int qi = blockIdx.x, C = h * d;
    const int start = offsets[qi], end = offsets[qi + 1];
    float *part = qv + C;
    const int S = C <= WB ? WB / C : 1;
    for (int c = threadIdx.x; c < C; c += WB) qv[c] = q[(size_t)qi * C + c];
    if (C <= WB) {
        const int s = threadIdx.x / C, c = threadIdx.x - s * C;
        if (s < S) {
            const int hh = c / d;
            float acc = 0.f;
            for (int m = start + s; m < end; m += S) acc += go[(size_t)m * h + hh] * k[(size_t)index1[m] * C + c];
            part[s * C + c] = acc;
        }
        __syncthreads();
        if (threadIdx.x < C) {
            float acc = part[threadIdx.x];
            for (int t = 1; t < S; ++t) acc += part[t * C + threadIdx.x];
            grad_q[(size_t)qi * C + threadIdx.x] = acc;
        }
    } else {
        for (int c = threadIdx.x; c < C; c += WB) {
            const int hh = c / d;
            float acc = 0.f;
            for (int m = start; m < end; ++m) acc += go[(size_t)m * h + hh] * k[(size_t)index1[m] * C + c];
            grad_q[(size_t)qi * C + c] = acc;
        }
        __syncthreads();
    }
    const long total = (long)(end - start) * C;
    for (long e = threadIdx.x; e < total; e += WB) {
        const int m = start + (int)(e / C), c = (int)(e % C);
        pdf_atomic_add(grad_k + (size_t)index1[m] * C + c, go[(size_t)m * h + c / d] * qv[c]);
    }
}

// ---------------------------------------------------------------- dot_prod_with_idx_v3
// relative_pos_encoding_cuda_kernel_v2.cu:247-285
__global__ __launch_bounds__(WB) void k_dot3_fwd(int h, int d, const float *__restrict__ q, const int *__restrict__ offsets,
                                                 const float *__restrict__ k, const int *__restrict__ index_k,
                                                 const float *__restrict__ table_q, const float *__restrict__ table_k,
                                                 const int *__restrict__ rel_idx, float *__restrict__ output) {
    extern __shared__ float qv[];
    const int qi = blockIdx.x, C = h * d;
    const int start = offsets[qi], end = offsets[qi + 1];
    if (end <= start) return;
    for (int c = threadIdx.x; c < C; c += WB) qv[c] = q[(size_t)qi * C + c];
    __syncthreads();
    const int total = (end - start) * h;
    for (int e = threadIdx.x; e < total; e += WB) {
        const int m = start + e / h, hh = e % h;
        const int r1 = rel_idx[(size_t)m * 3], r2 = rel_idx[(size_t)m * 3 + 1], r3 = rel_idx[(size_t)m * 3 + 2];
        const float *kr = k + (size_t)index_k[m] * C;
        float sum = 0.f;
        for (int i = 0; i < d; ++i) {
            const int c = hh * d + i;
            sum += qv[c] * table_sum(table_q, r1, r2, r3, C, c);
            sum += kr[c] * table_sum(table_k, r1, r2, r3, C, c);
        }
        output[(size_t)m * h + hh] = sum;
    }
}

// relative_pos_encoding_cuda_kernel_v2.cu:287-340.  grad_q written; grad_k, grad_table_q, grad_table_k pre-zeroed.
__global__ __launch_bounds__(WB) void k_dot3_bwd(int h, int d, const float *__restrict__ go, const float *__restrict__ q,
                                                 const int *__restrict__ offsets, const float *__restrict__ k,
                                                 const int *__restrict__ index_k, const float *__restrict__ table_q,
                                                 const float *__restrict__ table_k, const int *__restrict__ rel_idx,
                                                 float *__restrict__ grad_q, float *__restrict__ grad_k,
                                                 float *__restrict__ grad_table_q, float *__restrict__ grad_table_k) {
    extern __shared__ float qv[];
    const int qi = blockIdx.x, C = h * d;
    const int start = offsets[qi], end = offsets[qi + 1];
    for (int c = threadIdx.x; c < C; c += WB) {
        qv[c] = q[(size_t)qi * C + c];
        const int hh = c / d;
        float acc = 0.f;
        for (int m = start; m < end; ++m) {
            const int r1 = rel_idx[(size_t)m * 3], r2 = rel_idx[(size_t)m * 3 + 1], r3 = rel_idx[(size_t)m * 3 + 2];
            acc += table_sum(table_q, r1, r2, r3, C, c) * go[(size_t)m * h + hh];
        }
        grad_q[(size_t)qi * C + c] = acc;
    }
    __syncthreads();
    const long total = (long)(end - start) * C;
    for (long e = threadIdx.x; e < total; e += WB) {
        const int m = start + (int)(e / C), c = (int)(e % C);
        const int r1 = rel_idx[(size_t)m * 3], r2 = rel_idx[(size_t)m * 3 + 1], r3 = rel_idx[(size_t)m * 3 + 2];
        const float g = go[(size_t)m * h + c / d];
        const size_t kc = (size_t)index_k[m] * C + c;
        pdf_atomic_add(grad_k + kc, table_sum(table_k, r1, r2, r3, C, c) * g);
        const float gq = qv[c] * g, gk = k[kc] * g;
        pdf_atomic_add(grad_table_q + ((size_t)r1 * C + c) * 3, gq);
        pdf_atomic_add(grad_table_q + ((size_t)r2 * C + c) * 3 + 1, gq);
        pdf_atomic_add(grad_table_q + ((size_t)r3 * C + c) * 3 + 2, gq);
        pdf_atomic_add(grad_table_k + ((size_t)r1 * C + c) * 3, gk);
        pdf_atomic_add(grad_table_k + ((size_t)r2 * C + c) * 3 + 1, gk);
        pdf_atomic_add(grad_table_k + ((size_t)r3 * C + c) * 3 + 2, gk);
    }
}

// ---------------------------------------------------------------- attention_step2_with_rel_pos_value_v2
// relative_pos_encoding_cuda_kernel_v2.cu:397-439 (output written, one lane per channel; the reference reduces through
// shared atomics in arbitrary order)
__global__ __launch_bounds__(WB) void k_step2rv_fwd(int h, int d, const float *__restrict__ attn, const float *__restrict__ v,
                                                    const int *__restrict__ offsets, const int *__restrict__ index1,
                                                    const float *__restrict__ table, const int *__restrict__ rel_idx,
                                                    float *__restrict__ output) {
    const int qi = blockIdx.x, C = h * d;
    const int start = offsets[qi], end = offsets[qi + 1];
    for (int c = threadIdx.x; c < C; c += WB) {
        const int hh = c / d;
        float acc = 0.f;
        for (int m = start; m < end; ++m) {
            const int r1 = rel_idx[(size_t)m * 3], r2 = rel_idx[(size_t)m * 3 + 1], r3 = rel_idx[(size_t)m * 3 + 2];
            acc += (table_sum(table, r1, r2, r3, C, c) + v[(size_t)index1[m] * C + c]) * attn[(size_t)m * h + hh];
        }
        output[(size_t)qi * C + c] = acc;
    }
}

// relative_pos_encoding_cuda_kernel_v2.cu:441-484.  grad_attn written; grad_v, grad_table pre-zeroed.
__global__ __launch_bounds__(WB) void k_step2rv_bwd(int h, int d, const float *__restrict__ go, const int *__restrict__ offsets,
                                                    const int *__restrict__ index1, const float *__restrict__ attn,
                                                    const float *__restrict__ v, const float *__restrict__ table,
                                                    const int *__restrict__ rel_idx, float *__restrict__ grad_attn,
                                                    float *__restrict__ grad_v, float *__restrict__ grad_table) {
    extern __shared__ float gv[];
    const int qi = blockIdx.x, C = h * d;
    const int start = offsets[qi], end = offsets[qi + 1];
    if (end <= start) return;
    for (int c = threadIdx.x; c < C; c += WB) gv[c] = go[(size_t)qi * C + c];
    __syncthreads();
    const int total = (end - start) * h;
    for (int e = threadIdx.x; e < total; e += WB) {
        const int m = start + e / h, hh = e % h;
        const int r1 = rel_idx[(size_t)m * 3], r2 = rel_idx[(size_t)m * 3 + 1], r3 = rel_idx[(size_t)m * 3 + 2];
        const float *vr = v + (size_t)index1[m] * C;
        float sum = 0.f;
        for (int i = 0; i < d; ++i) {
            const int c = hh * d + i;
            sum += (table_sum(table, r1, r2, r3, C, c) + vr[c]) * gv[c];
        }
        grad_attn[(size_t)m * h + hh] = sum;
    }
    const long totc = (long)(end - start) * C;
    for (long e = threadIdx.x; e < totc; e += WB) {
        const int m = start + (int)(e / C), c = (int)(e % C);
        const int r1 = rel_idx[(size_t)m * 3], r2 = rel_idx[(size_t)m * 3 + 1], r3 = rel_idx[(size_t)m * 3 + 2];
        const float g = attn[(size_t)m * h + c / d] * gv[c];
        pdf_atomic_add(grad_v + (size_t)index1[m] * C + c, g);
        pdf_atomic_add(grad_table + ((size_t)r1 * C + c) * 3, g);
        pdf_atomic_add(grad_table + ((size_t)r2 * C + c) * 3 + 1, g);
        pdf_atomic_add(grad_table + ((size_t)r3 * C + c) * 3 + 2, g);
    }
}


// ================================================================ per-head kernels with the tables in LDS
// The kernels above read the relative-position tables from global memory: 3 strided dwords per (edge, channel) and table,
// 4.6e8 table loads for the reference's own test shape (M = 800k, C = 96), and every table-gradient atomic lands on one of
// ~L*3 hot rows.  A head's slab of a table is small (L * d * 3 floats: 9 KB at L = 48, d = 16), so these twins give a
// workgroup ONE head and a chunk of QCH consecutive queries (a contiguous edge range in CSR order): the head's table slabs
// are staged in LDS once per workgroup, table gradients accumulate in LDS (ds_add_f32) and leave as one global atomic per
// table element and workgroup, grad_q of the chunk accumulates in LDS and is stored plainly.  lane = (edge, channel) with the
// channel fastest: key / value rows are read and scattered 4 * d contiguous bytes per edge.  Needs d a power of two <= 64 and
// the slabs to fit 64 KB; anything else takes the generic kernels.
constexpr int QCH = 64;   // queries per workgroup

__device__ __forceinline__ int find_query(const int *__restrict__ offs, int nq, int m) {
    // largest j in [0, nq) with offs[j] <= m  (offs has nq + 1 entries, offs[nq] > m)
    int lo = 0, hi = nq;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offs[mid] <= m) lo = mid; else hi = mid;
    }
    return lo;
}

// LDS slab of one head: row r at r * (3 d + 1) floats.  The odd stride spreads the rows over the banks: with the natural
// stride 48 (d = 16) the lanes of a lane-per-edge loop -- same channel, different r -- would hit two banks (32-way conflict).
__device__ __forceinline__ int slab_stride(int d) { return 3 * d + 1; }
__device__ __forceinline__ int slab_floats(int L, int d) { return L * (3 * d + 1); }
__device__ __forceinline__ void stage_table(float *__restrict__ dst, const float *__restrict__ table, int L, int C, int d, int hh) {
    const int per = d * 3;   // floats of one (r, head) slab row
    for (int e = threadIdx.x; e < L * per; e += WB) {
        const int r = e / per, x = e - r * per;
        dst[r * (per + 1) + x] = table[((size_t)r * C + (size_t)hh * d) * 3 + x];
    }
}
__device__ __forceinline__ float lds_table_sum(const float *t, int r1, int r2, int r3, int d, int i) {
    const int st = 3 * d + 1;
    return t[r1 * st + 3 * i] + t[r2 * st + 3 * i + 1] + t[r3 * st + 3 * i + 2];
}

// dot_prod_with_idx_v3 forward, lane = edge (the d-loop keeps the reference's summation order)
__global__ __launch_bounds__(WB) void k_dot3_fwd_h(int N, int h, int d, int L, const float *__restrict__ q, const int *__restrict__ offsets,
                                                   const float *__restrict__ k, const int *__restrict__ index_k,
                                                   const float *__restrict__ table_q, const float *__restrict__ table_k,
                                                   const int *__restrict__ rel_idx, float *__restrict__ output) {
    extern __shared__ float sm[];
    float *tq = sm, *tk = tq + slab_floats(L, d);
    int *offs = reinterpret_cast<int *>(tk + slab_floats(L, d));
    const int hh = blockIdx.y, C = h * d;
    const int q0 = blockIdx.x * QCH, nq = min(QCH, N - q0);
    for (int j = threadIdx.x; j <= nq; j += WB) offs[j] = offsets[q0 + j];
    stage_table(tq, table_q, L, C, d, hh);
    stage_table(tk, table_k, L, C, d, hh);
    __syncthreads();
    const int e0 = offs[0], e1 = offs[nq];
    for (int m = e0 + threadIdx.x; m < e1; m += WB) {
        const int qi = q0 + find_query(offs, nq, m);
        const int r1 = rel_idx[(size_t)m * 3], r2 = rel_idx[(size_t)m * 3 + 1], r3 = rel_idx[(size_t)m * 3 + 2];
        const float *qr = q + (size_t)qi * C + hh * d, *kr = k + (size_t)index_k[m] * C + hh * d;
        float sum = 0.f;
        for (int i = 0; i < d; ++i) {
            sum += qr[i] * lds_table_sum(tq, r1, r2, r3, d, i);
            sum += kr[i] * lds_table_sum(tk, r1, r2, r3, d, i);
        }
        output[(size_t)m * h + hh] = sum;
    }
}

// dot_prod_with_idx_v3 backward, lane = (edge, channel)
__global__ __launch_bounds__(WB) void k_dot3_bwd_h(int N, int h, int d, int L, const float *__restrict__ go, const float *__restrict__ q,
                                                   const int *__restrict__ offsets, const float *__restrict__ k,
                                                   const int *__restrict__ index_k, const float *__restrict__ table_q,
                                                   const float *__restrict__ table_k, const int *__restrict__ rel_idx,
                                                   float *__restrict__ grad_q, float *__restrict__ grad_k,
                                                   float *__restrict__ grad_table_q, float *__restrict__ grad_table_k) {
    extern __shared__ float sm[];
    const int T = slab_floats(L, d);
    float *tq = sm, *tk = tq + T, *gtq = tk + T, *gtk = gtq + T, *gqs = gtk + T;   // gqs: QCH * d
    int *offs = reinterpret_cast<int *>(gqs + QCH * d);
    const int hh = blockIdx.y, C = h * d;
    const int q0 = blockIdx.x * QCH, nq = min(QCH, N - q0);
    for (int j = threadIdx.x; j <= nq; j += WB) offs[j] = offsets[q0 + j];
    stage_table(tq, table_q, L, C, d, hh);
    stage_table(tk, table_k, L, C, d, hh);
    for (int e = threadIdx.x; e < 2 * T + QCH * d; e += WB) gtq[e] = 0.f;   // gtq | gtk | gqs are contiguous
    __syncthreads();
    const int e0 = offs[0], e1 = offs[nq];
    const int lg = 31 - __clz(d);
    const long total = (long)(e1 - e0) << lg;
    for (long e = threadIdx.x; e < total; e += WB) {
        const int m = e0 + (int)(e >> lg), i = (int)e & (d - 1);
        const int ql = find_query(offs, nq, m);
        const int r1 = rel_idx[(size_t)m * 3], r2 = rel_idx[(size_t)m * 3 + 1], r3 = rel_idx[(size_t)m * 3 + 2];
        const float g = go[(size_t)m * h + hh];
        const size_t kc = (size_t)index_k[m] * C + hh * d + i;
        const float qv = q[(size_t)(q0 + ql) * C + hh * d + i], kv = k[kc];
        atomicAdd(&gqs[ql * d + i], lds_table_sum(tq, r1, r2, r3, d, i) * g);
        pdf_atomic_add(grad_k + kc, lds_table_sum(tk, r1, r2, r3, d, i) * g);
        const float gq = qv * g, gk = kv * g;
        { const int st = slab_stride(d); atomicAdd(&gtq[r1 * st + 3 * i], gq); atomicAdd(&gtq[r2 * st + 3 * i + 1], gq); atomicAdd(&gtq[r3 * st + 3 * i + 2], gq);
        atomicAdd(&gtk[r1 * st + 3 * i], gk); atomicAdd(&gtk[r2 * st + 3 * i + 1], gk); atomicAdd(&gtk[r3 * st + 3 * i + 2], gk); }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nq * d; e += WB) grad_q[(size_t)(q0 + e / d) * C + hh * d + e % d] = gqs[e];
    const int per = d * 3;
    for (int e = threadIdx.x; e < L * per; e += WB) {
        const int r = e / per, x = e - r * per;
        const size_t dst = ((size_t)r * C + (size_t)hh * d) * 3 + x;
        const float a = gtq[r * (per + 1) + x], b = gtk[r * (per + 1) + x];
        if (a != 0.f) pdf_atomic_add(grad_table_q + dst, a);
        if (b != 0.f) pdf_atomic_add(grad_table_k + dst, b);
    }
}

// attention_step2_with_rel_pos_value_v2 backward, lane = (edge, channel); grad_attn = 16/32-lane shuffle reduction
__global__ __launch_bounds__(WB) void k_step2rv_bwd_h(int N, int h, int d, int L, const float *__restrict__ go, const int *__restrict__ offsets,
                                                      const int *__restrict__ index1, const float *__restrict__ attn,
                                                      const float *__restrict__ v, const float *__restrict__ table,
                                                      const int *__restrict__ rel_idx, float *__restrict__ grad_attn,
                                                      float *__restrict__ grad_v, float *__restrict__ grad_table) {
    extern __shared__ float sm[];
    const int T = slab_floats(L, d);
    float *tb = sm, *gt = tb + T;
    int *offs = reinterpret_cast<int *>(gt + T);
    const int hh = blockIdx.y, C = h * d;
    const int q0 = blockIdx.x * QCH, nq = min(QCH, N - q0);
    for (int j = threadIdx.x; j <= nq; j += WB) offs[j] = offsets[q0 + j];
    stage_table(tb, table, L, C, d, hh);
    for (int e = threadIdx.x; e < T; e += WB) gt[e] = 0.f;
    __syncthreads();
    const int e0 = offs[0], e1 = offs[nq];
    const int lg = 31 - __clz(d);
    const long total = (long)(e1 - e0) << lg;
    const long padded = (total + 63) & ~63L;   // whole waves iterate together (shuffles)
    for (long e = threadIdx.x; e < padded; e += WB) {
        const bool live = e < total;
        float part = 0.f;
        int m = 0;
        const int i = (int)e & (d - 1);
        if (live) {
            m = e0 + (int)(e >> lg);
            const int ql = find_query(offs, nq, m);
            const int r1 = rel_idx[(size_t)m * 3], r2 = rel_idx[(size_t)m * 3 + 1], r3 = rel_idx[(size_t)m * 3 + 2];
            const size_t vc = (size_t)index1[m] * C + hh * d + i;
            const float gout = go[(size_t)(q0 + ql) * C + hh * d + i];
            part = (lds_table_sum(tb, r1, r2, r3, d, i) + v[vc]) * gout;
            const float g = attn[(size_t)m * h + hh] * gout;
            pdf_atomic_add(grad_v + vc, g);
            { const int st = slab_stride(d); atomicAdd(&gt[r1 * st + 3 * i], g); atomicAdd(&gt[r2 * st + 3 * i + 1], g); atomicAdd(&gt[r3 * st + 3 * i + 2], g); }
        }
        for (int o = d >> 1; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        if (live && i == 0) grad_attn[(size_t)m * h + hh] = part;
    }
    __syncthreads();
    const int per = d * 3;
    for (int e = threadIdx.x; e < L * per; e += WB) {
        const int r = e / per, x = e - r * per;
        const float a = gt[r * (per + 1) + x];
        if (a != 0.f) pdf_atomic_add(grad_table + ((size_t)r * C + (size_t)hh * d) * 3 + x, a);
    }
}


// ---------------------------------------------------------------- table gradients on the matrix cores (d = 16)
// grad_table[r, h, :, axis] = sum_m [rel_idx[m, axis] == r] * x[m, :] is a product of a one-hot (L x edges) matrix with the
// (edges x 16) matrix of per-edge channel rows.  The k_*_h kernels pay three ds_add_f32 per (edge, channel) and table for it
// (measured: they ARE the kernel time).  In the lane = (edge, channel) mapping a wave already holds 4 edges x 16 channels --
// exactly the B operand of v_mfma_f32_16x16x4_f32 (B[k = lane/16][n = lane%16]); the A operand A[m = lane%16][k = lane/16] is
// the one-hot test of the lane's OWN edge against row 16*rb + lane%16.  So every wave-trip issues 3 axes x ceil(L/16) MFMAs per
// table and keeps G in accumulators (D[m = 4*(lane/16) + j][n = lane%16]); LDS atomics are needed once per wave at the end.
typedef float f32x4 __attribute__((ext_vector_type(4)));
// Elements per lane and trip of the two kernels below (tuning knob, `PDFOPS_WA_UE=<n> python -m pointcloudpdf_amd.build`).  Measured on
// the 2 x 80k-point ST-v1m1 step (A/B of round 3, profiles/r03_wa_ue_ab.txt): 2 -> 1.81 / 1.01 ms per launch (dot_prod backward /
// step2 backward), 4 -> 2.34 / 1.10, 8 -> 2.24 / 1.08: more loads in flight per trip do not pay, the kernels are not waiting for their
// gathers.
#ifndef PDF_WA_UE
#define PDF_WA_UE 2
#endif
constexpr int UE = PDF_WA_UE;

template <int RB>
__global__ __launch_bounds__(WB) void k_dot3_bwd_m16(int N, int h, int L, const float *__restrict__ go, const float *__restrict__ q,
                                                     const int *__restrict__ offsets, const float *__restrict__ k,
                                                     const int *__restrict__ index_k, const float *__restrict__ table_q,
                                                     const float *__restrict__ table_k, const int *__restrict__ rel_idx,
                                                     float *__restrict__ grad_q, float *__restrict__ grad_k,
                                                     float *__restrict__ grad_table_q, float *__restrict__ grad_table_k) {
    constexpr int d = 16;
    extern __shared__ float sm[];
    const int T = slab_floats(L, d);
    float *tq = sm, *tk = tq + T, *gtq = tk + T, *gtk = gtq + T, *gqs = gtk + T;   // gqs: QCH * d
    int *offs = reinterpret_cast<int *>(gqs + QCH * d);
    const int hh = blockIdx.y, C = h * d;
    const int q0 = blockIdx.x * QCH, nq = min(QCH, N - q0);
    for (int j = threadIdx.x; j <= nq; j += WB) offs[j] = offsets[q0 + j];
    stage_table(tq, table_q, L, C, d, hh);
    stage_table(tk, table_k, L, C, d, hh);
    for (int e = threadIdx.x; e < 2 * T + QCH * d; e += WB) gtq[e] = 0.f;
    __syncthreads();
    const int e0 = offs[0], e1 = offs[nq];
    const long total = (long)(e1 - e0) * d;
    const long padded = (total + 63) & ~63L;
    const int lane = threadIdx.x & 63, i = lane & 15;
    f32x4 aq[3][RB], ak[3][RB];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) { aq[a][rb] = (f32x4)(0.f); ak[a][rb] = (f32x4)(0.f); }
    // Two elements per lane and trip, every load of a trip issued before the first use (index / scalar loads, then the rows that depend
    // on them), no load under `if (live)` -- clamped edge id, effects masked (a load under a branch is a wait of its own); the
    // query of an edge is found by stepping on from the lane's previous query (edges are in CSR order: 0-1 steps instead of a 6-step
    // binary search through LDS).
    int ql_hint = 0;
    const int m_last = e1 > e0 ? e1 - 1 : e0;
    for (long e = threadIdx.x; e < padded; e += UE * WB) {
        bool live[UE]; int m[UE], ql[UE], r[UE][3], ik[UE]; float g[UE], xq[UE], xk[UE];
#pragma unroll
        for (int u = 0; u < UE; ++u) {
            const long eu = e + u * WB;
            live[u] = eu < total;
            m[u] = min(e0 + (int)(eu >> 4), m_last);
        }
#pragma unroll
        for (int u = 0; u < UE; ++u) {
            r[u][0] = rel_idx[(size_t)m[u] * 3]; r[u][1] = rel_idx[(size_t)m[u] * 3 + 1]; r[u][2] = rel_idx[(size_t)m[u] * 3 + 2];
            g[u] = go[(size_t)m[u] * h + hh];
            ik[u] = index_k[m[u]];
        }
#pragma unroll
        for (int u = 0; u < UE; ++u) {
            while (ql_hint + 1 < nq && offs[ql_hint + 1] <= m[u]) ++ql_hint;
            ql[u] = ql_hint;
        }
#pragma unroll
        for (int u = 0; u < UE; ++u) {
            xq[u] = q[(size_t)(q0 + ql[u]) * C + hh * d + i];
            xk[u] = k[(size_t)ik[u] * C + hh * d + i];
        }
#pragma unroll
        for (int u = 0; u < UE; ++u) {
            const float gm = live[u] ? g[u] : 0.f;
            // grad_q: the four edges a wave holds for channel i (lanes i, i + 16, i + 32, i + 48) are consecutive edges of the CSR list and
            // mostly belong to ONE query: their contributions are added across the lanes first and one lane issues the LDS atomic (four
            // same-address ds_add_f32 serialise); waves that straddle a query boundary take the per-lane path.
            float cq = live[u] ? lds_table_sum(tq, r[u][0], r[u][1], r[u][2], d, i) * gm : 0.f;
            const int ql0 = __shfl(ql[u], i, 64);
            if (__ballot(ql[u] != ql0) == 0ull) {
                cq += __shfl_xor(cq, 16, 64);
                cq += __shfl_xor(cq, 32, 64);
                if (lane < 16 && cq != 0.f) atomicAdd(&gqs[ql0 * d + i], cq);
            } else if (live[u]) {
                atomicAdd(&gqs[ql[u] * d + i], cq);
            }
            if (live[u]) pdf_atomic_add(grad_k + (size_t)ik[u] * C + hh * d + i, lds_table_sum(tk, r[u][0], r[u][1], r[u][2], d, i) * gm);
            const float vq = xq[u] * gm, vk = xk[u] * gm;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    const float hot = (live[u] && r[u][a] == 16 * rb + i) ? 1.f : 0.f;
                    aq[a][rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(hot, vq, aq[a][rb], 0, 0, 0);
                    ak[a][rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(hot, vk, ak[a][rb], 0, 0, 0);
                }
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = 16 * rb + 4 * (lane >> 4) + j;
                if (row < L) {
                    atomicAdd(&gtq[row * slab_stride(d) + 3 * i + a], aq[a][rb][j]);
                    atomicAdd(&gtk[row * slab_stride(d) + 3 * i + a], ak[a][rb][j]);
                }
            }
    __syncthreads();
    for (int e = threadIdx.x; e < nq * d; e += WB) grad_q[(size_t)(q0 + e / d) * C + hh * d + e % d] = gqs[e];
    const int per = d * 3;
    for (int e = threadIdx.x; e < L * per; e += WB) {
        const int rr = e / per, x = e - rr * per;
        const size_t dst = ((size_t)rr * C + (size_t)hh * d) * 3 + x;
        const float a = gtq[rr * (per + 1) + x], b = gtk[rr * (per + 1) + x];
        if (a != 0.f) pdf_atomic_add(grad_table_q + dst, a);
        if (b != 0.f) pdf_atomic_add(grad_table_k + dst, b);
    }
}

template <int RB>
__global__ __launch_bounds__(WB) void k_step2rv_bwd_m16(int N, int h, int L, const float *__restrict__ go, const int *__restrict__ offsets,
                                                        const int *__restrict__ index1, const float *__restrict__ attn,
                                                        const float *__restrict__ v, const float *__restrict__ table,
                                                        const int *__restrict__ rel_idx, float *__restrict__ grad_attn,
                                                        float *__restrict__ grad_v, float *__restrict__ grad_table) {
    constexpr int d = 16;
    extern __shared__ float sm[];
    const int T = slab_floats(L, d);
    float *tb = sm, *gt = tb + T;
    int *offs = reinterpret_cast<int *>(gt + T);
    const int hh = blockIdx.y, C = h * d;
    const int q0 = blockIdx.x * QCH, nq = min(QCH, N - q0);
    for (int j = threadIdx.x; j <= nq; j += WB) offs[j] = offsets[q0 + j];
    stage_table(tb, table, L, C, d, hh);
    for (int e = threadIdx.x; e < T; e += WB) gt[e] = 0.f;
    __syncthreads();
    const int e0 = offs[0], e1 = offs[nq];
    const long total = (long)(e1 - e0) * d;
    const long padded = (total + 63) & ~63L;
    const int lane = threadIdx.x & 63, i = lane & 15;
    f32x4 acc[3][RB];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[a][rb] = (f32x4)(0.f);
    int ql_hint = 0;   // (trip structure as k_dot3_bwd_m16)
    const int m_last = e1 > e0 ? e1 - 1 : e0;
    for (long e = threadIdx.x; e < padded; e += UE * WB) {
        bool live[UE]; int m[UE], ql[UE], r[UE][3], i1[UE]; float at[UE], gout[UE], vv[UE];
#pragma unroll
        for (int u = 0; u < UE; ++u) {
            const long eu = e + u * WB;
            live[u] = eu < total;
            m[u] = min(e0 + (int)(eu >> 4), m_last);
        }
#pragma unroll
        for (int u = 0; u < UE; ++u) {
            r[u][0] = rel_idx[(size_t)m[u] * 3]; r[u][1] = rel_idx[(size_t)m[u] * 3 + 1]; r[u][2] = rel_idx[(size_t)m[u] * 3 + 2];
            at[u] = attn[(size_t)m[u] * h + hh];
            i1[u] = index1[m[u]];
        }
#pragma unroll
        for (int u = 0; u < UE; ++u) {
            while (ql_hint + 1 < nq && offs[ql_hint + 1] <= m[u]) ++ql_hint;
            ql[u] = ql_hint;
        }
#pragma unroll
        for (int u = 0; u < UE; ++u) {
            gout[u] = go[(size_t)(q0 + ql[u]) * C + hh * d + i];
            vv[u] = v[(size_t)i1[u] * C + hh * d + i];
        }
#pragma unroll
        for (int u = 0; u < UE; ++u) {
            const float x = live[u] ? at[u] * gout[u] : 0.f;
            float part = (lds_table_sum(tb, r[u][0], r[u][1], r[u][2], d, i) + vv[u]) * gout[u];
            if (live[u]) pdf_atomic_add(grad_v + (size_t)i1[u] * C + hh * d + i, x);
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
                    acc[a][rb] = __builtin_amdgcn_mfma_f32_16x16x4f32((live[u] && r[u][a] == 16 * rb + i) ? 1.f : 0.f, x, acc[a][rb], 0, 0, 0);
            for (int o = 8; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
            if (live[u] && i == 0) grad_attn[(size_t)m[u] * h + hh] = part;
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = 16 * rb + 4 * (lane >> 4) + j;
                if (row < L) atomicAdd(&gt[row * slab_stride(d) + 3 * i + a], acc[a][rb][j]);
            }
    __syncthreads();
    const int per = d * 3;
    for (int e = threadIdx.x; e < L * per; e += WB) {
        const int rr = e / per, x = e - rr * per;
        const float a = gt[rr * (per + 1) + x];
        if (a != 0.f) pdf_atomic_add(grad_table + ((size_t)rr * C + (size_t)hh * d) * 3 + x, a);
    }
}


// ---------------------------------------------------------------- table gradients factored by query (d = 16, L <= 64)
// Wherever the value that multiplies the one-hot row selector is constant over the edges of a query -- q[query(m)] for table_q of
// dot_prod_with_idx_v3, grad_out[query(m)] for the table of attention_step2_with_rel_pos_value_v2 -- the table gradient factors:
//     G[r, :, a] = sum_m [rel[m, a] = r] s_m x[query(m), :]  =  sum_q x[q, :] (x) S_q[a, r],      S_q[a, r] = sum_{m in q, rel[m, a] = r} s_m
// i.e. a per-query HISTOGRAM of the edge scalars (three LDS adds per edge and head) followed by one small dense product per workgroup
// (3 L x QF queries x 16 channels) instead of 3 ceil(L / 16) one-hot matrix-core products per FOUR edges (15/16 of whose rows multiply
// zeros).  The same histogram gives the query-indexed result: grad_q[q, :] = sum_{a, r} S_q[a, r] table_q[r, :, a]; and for the value
// table the per-edge term <grad_out[q, :], T(m, :)> becomes three lookups in the projection P_q[a, r] = <table[r, :, a], grad_out[q, :]>.
// Only table_k of dot_prod_with_idx_v3 (value = the gathered key row, different per edge) keeps the one-hot products (k_dot3_bwd_k).
constexpr int QF = 32;   // queries per workgroup of the factored kernels (histogram: QF x 3 L floats of LDS)

// dot_prod_with_idx_v3 backward, query side: grad_q (written) and grad_table_q.  lane = edge for the histogram.
__global__ __launch_bounds__(WB) void k_dot3_bwd_fq(int N, int h, int L, const float *__restrict__ go, const float *__restrict__ q,
                                                    const int *__restrict__ offsets, const float *__restrict__ table_q,
                                                    const int *__restrict__ rel_idx, float *__restrict__ grad_q, float *__restrict__ grad_table_q) {
    constexpr int d = 16;
    extern __shared__ float sm[];
    const int T = slab_floats(L, d), W = 3 * L;
    float *tq = sm, *qr = tq + T, *S = qr + QF * d;
    int *offs = reinterpret_cast<int *>(S + QF * W);
    const int hh = blockIdx.y, C = h * d;
    const int q0 = blockIdx.x * QF, nq = min(QF, N - q0);
    for (int j = threadIdx.x; j <= nq; j += WB) offs[j] = offsets[q0 + j];
    stage_table(tq, table_q, L, C, d, hh);
    for (int e = threadIdx.x; e < QF * d; e += WB) qr[e] = e < nq * d ? q[(size_t)(q0 + e / d) * C + hh * d + e % d] : 0.f;
    for (int e = threadIdx.x; e < QF * W; e += WB) S[e] = 0.f;
    __syncthreads();
    const int e0 = offs[0], e1 = offs[nq];
    for (int m = e0 + threadIdx.x; m < e1; m += WB) {
        const int ql = find_query(offs, nq, m);
        const float g = go[(size_t)m * h + hh];
        const int r1 = rel_idx[(size_t)m * 3], r2 = rel_idx[(size_t)m * 3 + 1], r3 = rel_idx[(size_t)m * 3 + 2];
        atomicAdd(&S[ql * W + r1], g);
        atomicAdd(&S[ql * W + L + r2], g);
        atomicAdd(&S[ql * W + 2 * L + r3], g);
    }
    __syncthreads();
    const int st = slab_stride(d);
    for (int e = threadIdx.x; e < nq * d; e += WB) {          // grad_q[q, i] = sum_{a, r} S_q[a, r] table_q[r, i, a]
        const int ql = e / d, i = e % d;
        const float *Sq = S + ql * W;
        float acc = 0.f;
        for (int r = 0; r < L; ++r)
            acc += Sq[r] * tq[r * st + 3 * i] + Sq[L + r] * tq[r * st + 3 * i + 1] + Sq[2 * L + r] * tq[r * st + 3 * i + 2];
        grad_q[(size_t)(q0 + ql) * C + hh * d + i] = acc;
    }
    for (int e = threadIdx.x; e < W * d; e += WB) {            // grad_table_q[r, i, a] += sum_q S_q[a, r] q[q, i]
        const int x = e / d, i = e % d, a = x / L, r = x - a * L;
        float acc = 0.f;
        for (int ql = 0; ql < nq; ++ql) acc += S[ql * W + x] * qr[ql * d + i];
        if (acc != 0.f) pdf_atomic_add(grad_table_q + ((size_t)r * C + (size_t)hh * d + i) * 3 + a, acc);
    }
}

// dot_prod_with_idx_v3 backward, key side: grad_k (scatter) and grad_table_k -- the value row differs per edge, so the table gradient
// stays a one-hot matrix-core product (see k_dot3_bwd_m16); one table per workgroup: 12 RB accumulator registers, one LDS slab that is
// the table while the edges stream and the gradient slab afterwards.
template <int RB>
__global__ __launch_bounds__(WB) void k_dot3_bwd_k(int N, int h, int L, const float *__restrict__ go, const int *__restrict__ offsets,
                                                   const float *__restrict__ k, const int *__restrict__ index_k,
                                                   const float *__restrict__ table_k, const int *__restrict__ rel_idx,
                                                   float *__restrict__ grad_k, float *__restrict__ grad_table_k) {
    constexpr int d = 16;
    extern __shared__ float sm[];
    const int T = slab_floats(L, d);
    float *tb = sm;
    const int hh = blockIdx.y, C = h * d;
    const int q0 = blockIdx.x * QCH, nq = min(QCH, N - q0);
    stage_table(tb, table_k, L, C, d, hh);
    __syncthreads();
    const int e0 = offsets[q0], e1 = offsets[q0 + nq];
    const long total = (long)(e1 - e0) * d;
    const long padded = (total + 63) & ~63L;
    const int lane = threadIdx.x & 63, i = lane & 15;
    f32x4 acc[3][RB];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[a][rb] = (f32x4)(0.f);
    const int m_last = e1 > e0 ? e1 - 1 : e0;
    for (long e = threadIdx.x; e < padded; e += UE * WB) {
        bool live[UE]; int m[UE], r[UE][3], ik[UE]; float g[UE], x[UE];
#pragma unroll
        for (int u = 0; u < UE; ++u) {
            const long eu = e + u * WB;
            live[u] = eu < total;
            m[u] = min(e0 + (int)(eu >> 4), m_last);
        }
#pragma unroll
        for (int u = 0; u < UE; ++u) {
            r[u][0] = rel_idx[(size_t)m[u] * 3]; r[u][1] = rel_idx[(size_t)m[u] * 3 + 1]; r[u][2] = rel_idx[(size_t)m[u] * 3 + 2];
            g[u] = go[(size_t)m[u] * h + hh];
            ik[u] = index_k[m[u]];
        }
#pragma unroll
        for (int u = 0; u < UE; ++u) x[u] = k[(size_t)ik[u] * C + hh * d + i];
#pragma unroll
        for (int u = 0; u < UE; ++u) {
            const float gm = live[u] ? g[u] : 0.f;
            if (live[u]) pdf_atomic_add(grad_k + (size_t)ik[u] * C + hh * d + i, lds_table_sum(tb, r[u][0], r[u][1], r[u][2], d, i) * gm);
            const float v = x[u] * gm;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
                    acc[a][rb] = __builtin_amdgcn_mfma_f32_16x16x4f32((live[u] && r[u][a] == 16 * rb + i) ? 1.f : 0.f, v, acc[a][rb], 0, 0, 0);
        }
    }
    __syncthreads();                                    // every wave is done reading the table: the slab becomes the gradient slab
    for (int e = threadIdx.x; e < T; e += WB) tb[e] = 0.f;
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = 16 * rb + 4 * (lane >> 4) + j;
                if (row < L) atomicAdd(&tb[row * slab_stride(d) + 3 * i + a], acc[a][rb][j]);
            }
    __syncthreads();
    const int per = d * 3;
    for (int e = threadIdx.x; e < L * per; e += WB) {
        const int rr = e / per, xx = e - rr * per;
        const float a = tb[rr * (per + 1) + xx];
        if (a != 0.f) pdf_atomic_add(grad_table_k + ((size_t)rr * C + (size_t)hh * d) * 3 + xx, a);
    }
}

// attention_step2_with_rel_pos_value_v2 backward, factored: grad_attn (written), grad_v (scatter), grad_table.
__global__ __launch_bounds__(WB) void k_step2rv_bwd_f(int N, int h, int L, const float *__restrict__ go, const int *__restrict__ offsets,
                                                      const int *__restrict__ index1, const float *__restrict__ attn,
                                                      const float *__restrict__ v, const float *__restrict__ table,
                                                      const int *__restrict__ rel_idx, float *__restrict__ grad_attn,
                                                      float *__restrict__ grad_v, float *__restrict__ grad_table) {
    constexpr int d = 16;
    extern __shared__ float sm[];
    const int T = slab_floats(L, d), W = 3 * L;
    float *tb = sm, *gr = tb + T, *P = gr + QF * d, *S = P + QF * W;
    int *offs = reinterpret_cast<int *>(S + QF * W);
    const int hh = blockIdx.y, C = h * d;
    const int q0 = blockIdx.x * QF, nq = min(QF, N - q0);
    for (int j = threadIdx.x; j <= nq; j += WB) offs[j] = offsets[q0 + j];
    stage_table(tb, table, L, C, d, hh);
    for (int e = threadIdx.x; e < QF * d; e += WB) gr[e] = e < nq * d ? go[(size_t)(q0 + e / d) * C + hh * d + e % d] : 0.f;
    for (int e = threadIdx.x; e < QF * W; e += WB) S[e] = 0.f;
    __syncthreads();
    const int st = slab_stride(d);
    for (int e = threadIdx.x; e < nq * W; e += WB) {           // P_q[a, r] = <table[r, :, a], grad_out[q, :]>
        const int ql = e / W, x = e - ql * W, a = x / L, r = x - a * L;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < d; ++i) acc += tb[r * st + 3 * i + a] * gr[ql * d + i];
        P[e] = acc;
    }
    __syncthreads();
    const int e0 = offs[0], e1 = offs[nq];
    for (int m = e0 + threadIdx.x; m < e1; m += WB) {          // lane = edge: grad_attn and the histogram of attn
        const int ql = find_query(offs, nq, m);
        const int r1 = rel_idx[(size_t)m * 3], r2 = rel_idx[(size_t)m * 3 + 1], r3 = rel_idx[(size_t)m * 3 + 2];
        const float at = attn[(size_t)m * h + hh];
        const float4 *vr = reinterpret_cast<const float4 *>(v + (size_t)index1[m] * C + hh * d);
        const float *g = gr + ql * d;
        float dot = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < d / 4; ++c4) {
            const float4 x = vr[c4];
            dot += x.x * g[4 * c4] + x.y * g[4 * c4 + 1] + x.z * g[4 * c4 + 2] + x.w * g[4 * c4 + 3];
        }
        grad_attn[(size_t)m * h + hh] = dot + ((P[ql * W + r1] + P[ql * W + L + r2]) + P[ql * W + 2 * L + r3]);
        atomicAdd(&S[ql * W + r1], at);
        atomicAdd(&S[ql * W + L + r2], at);
        atomicAdd(&S[ql * W + 2 * L + r3], at);
    }
    {                                                           // grad_v[index1[m], :] += attn[m] grad_out[q, :]: lane = (edge, channel)
        const int i = threadIdx.x & 15;
        int ql = 0;                                             // (a lane's edges ascend: the query is found by stepping on)
        for (int m = e0 + (threadIdx.x >> 4); m < e1; m += WB / 16) {
            while (offs[ql + 1] <= m) ++ql;
            pdf_atomic_add(grad_v + (size_t)index1[m] * C + hh * d + i, attn[(size_t)m * h + hh] * gr[ql * d + i]);
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < W * d; e += WB) {            // grad_table[r, i, a] += sum_q S_q[a, r] grad_out[q, i]
        const int x = e / d, i = e % d, a = x / L, r = x - a * L;
        float acc = 0.f;
        for (int ql = 0; ql < nq; ++ql) acc += S[ql * W + x] * gr[ql * d + i];
        if (acc != 0.f) pdf_atomic_add(grad_table + ((size_t)r * C + (size_t)hh * d + i) * 3 + a, acc);
    }
}

// ---------------------------------------------------------------- segment softmax over the edges of a query
// StratifiedTransformer normalises the attention logits per query and head with torch_scatter.scatter_softmax(src, index_0, dim=0)
// (stratified_transformer_v1m1_origin.py:322-324; torch_scatter is an unvendored dependency, absent here): y[m,h] =
// exp(x[m,h] - max_q) / sum_q over the edges m of query q.  With the CSR offsets the op is one pass per query: lane = (query slot,
// head) with HP = next power of two >= h heads per slot, so a wave carries 64 / HP queries and walks their edge lists in lockstep;
// max, sum and the write are three sweeps over <= n_max edges.  The rows of a wave's queries are ONE contiguous range of x: when it
// fits the wave's slice of LDS (SM_CAP floats) it is brought in with coalesced loads, swept there and written back coalesced (the
// lockstep walk reads 64 / HP separate rows per instruction, h consecutive floats each: 171 -> see profiles/r06_wa_softmax_ab.txt);
// longer ranges take the sweeps over global memory.  Same operations in the same order on both paths: identical results.
constexpr int SM_CAP = 2048;
__global__ __launch_bounds__(WB) void k_seg_softmax_fwd(int N, int h, int HP, const int *__restrict__ offsets, const float *__restrict__ x,
                                                        float *__restrict__ y) {
    __shared__ float buf[(WB / 64) * SM_CAP];
    const int per_wave = 64 / HP;
    const int lane = threadIdx.x & 63, slot = lane / HP, hh = lane - slot * HP;
    const long wave = ((long)blockIdx.x * WB + threadIdx.x) >> 6;
    const long q0 = wave * per_wave, q = q0 + slot;
    if (q0 >= N) return;
    const int s = offsets[q0], e = offsets[min(q0 + per_wave, (long)N)];
    const bool mine = q < N && hh < h;
    const int start = mine ? offsets[q] : 0, end = mine ? offsets[q + 1] : 0;
    const long cnt = (long)(e - s) * h;
    if (cnt <= SM_CAP) {
        float *b = buf + (threadIdx.x >> 6) * SM_CAP;
        const float *src = x + (size_t)s * h;
        for (int i = lane; i < cnt; i += 64) b[i] = src[i];
        float *row = mine ? b + (size_t)(start - s) * h + hh : b;
        const int len = end - start;
        float mx = -3.0e38f;
        for (int m = 0; m < len; ++m) mx = fmaxf(mx, row[m * h]);
        float sum = 0.f;
        for (int m = 0; m < len; ++m) sum += __expf(row[m * h] - mx);
        const float inv = 1.f / sum;
        for (int m = 0; m < len; ++m) row[m * h] = __expf(row[m * h] - mx) * inv;
        float *dst = y + (size_t)s * h;
        for (int i = lane; i < cnt; i += 64) dst[i] = b[i];
        return;
    }
    if (!mine) return;
    float mx = -3.0e38f;
    for (int m = start; m < end; ++m) mx = fmaxf(mx, x[(size_t)m * h + hh]);
    float sum = 0.f;
    for (int m = start; m < end; ++m) sum += __expf(x[(size_t)m * h + hh] - mx);
    const float inv = 1.f / sum;
    for (int m = start; m < end; ++m) y[(size_t)m * h + hh] = __expf(x[(size_t)m * h + hh] - mx) * inv;
}

// gx = y * (gy - sum_q y * gy); staged through LDS like the forward (y and gy: 2 x SM_CAP floats per wave)
__global__ __launch_bounds__(WB) void k_seg_softmax_bwd(int N, int h, int HP, const int *__restrict__ offsets, const float *__restrict__ y,
                                                        const float *__restrict__ gy, float *__restrict__ gx) {
    __shared__ float buf[(WB / 64) * 2 * SM_CAP];
    const int per_wave = 64 / HP;
    const int lane = threadIdx.x & 63, slot = lane / HP, hh = lane - slot * HP;
    const long wave = ((long)blockIdx.x * WB + threadIdx.x) >> 6;
    const long q0 = wave * per_wave, q = q0 + slot;
    if (q0 >= N) return;
    const int s = offsets[q0], e = offsets[min(q0 + per_wave, (long)N)];
    const bool mine = q < N && hh < h;
    const int start = mine ? offsets[q] : 0, end = mine ? offsets[q + 1] : 0;
    const long cnt = (long)(e - s) * h;
    if (cnt <= SM_CAP) {
        float *by = buf + (threadIdx.x >> 6) * 2 * SM_CAP, *bg = by + SM_CAP;
        const float *sy = y + (size_t)s * h, *sg = gy + (size_t)s * h;
        for (int i = lane; i < cnt; i += 64) { by[i] = sy[i]; bg[i] = sg[i]; }
        float *ry = mine ? by + (size_t)(start - s) * h + hh : by, *rg = mine ? bg + (size_t)(start - s) * h + hh : bg;
        const int len = end - start;
        float dot = 0.f;
        for (int m = 0; m < len; ++m) dot += ry[m * h] * rg[m * h];
        for (int m = 0; m < len; ++m) rg[m * h] = ry[m * h] * (rg[m * h] - dot);
        float *dst = gx + (size_t)s * h;
        for (int i = lane; i < cnt; i += 64) dst[i] = bg[i];
        return;
    }
    if (!mine) return;
    float dot = 0.f;
    for (int m = start; m < end; ++m) dot += y[(size_t)m * h + hh] * gy[(size_t)m * h + hh];
    for (int m = start; m < end; ++m) gx[(size_t)m * h + hh] = y[(size_t)m * h + hh] * (gy[(size_t)m * h + hh] - dot);
}

static inline int heads_pow2(int h) { int p = 1; while (p < h) p <<= 1; return p; }

static inline bool pow2_le64(int d) { return d >= 1 && d <= 64 && (d & (d - 1)) == 0; }

static inline int bad_shape(int N, int M, int h, int C) { return N < 0 || M < 0 || h < 1 || C < 1 || C % h != 0; }

}  // namespace

// replaces attention_step1_forward_cuda_launcher_v2, libs/pointops2/src/attention_v2/attention_cuda_kernel_v2.h
// N = number of queries (rows of q, entries of index0_offsets minus one), M = edges.  Bytes: 4NC (q) + 4MC (gathered k rows)
// + 4M + 4N (indices) + 4Mh (attn).
extern "C" int pdf_attention_step1_forward_v2(int N, int M, int h, int C, unsigned n_max, const float *q, const float *k,
                                              const int *index0_offsets, const int *index1, float *attn, void *stream) {
    (void)n_max;
    if (bad_shape(N, M, h, C)) return PDF_ERR_BAD_ARG;
    if (N == 0 || M == 0) return PDF_OK;   // (0-size tensors carry null pointers)
    if (!q || !k || !index0_offsets || !index1 || !attn) return PDF_ERR_BAD_ARG;
    k_step1_fwd<<<N, WB, sizeof(float) * C, static_cast<hipStream_t>(stream)>>>(h, C / h, q, k, index0_offsets, index1, attn);
    return pdf_launch_status();
}

// replaces attention_step1_backward_cuda_launcher_v2.  grad_q (N,h,d) is overwritten, grad_k must be zeroed by the caller.
extern "C" int pdf_attention_step1_backward_v2(int N, int M, int h, int C, unsigned n_max, const float *grad_out,
                                               const int *index0_offsets, const int *index1, const float *q, const float *k,
                                               float *grad_q, float *grad_k, void *stream) {
    (void)n_max;
    if (bad_shape(N, M, h, C) || !grad_out || !q || !k || !index0_offsets || !index1 || !grad_q || !grad_k) return PDF_ERR_BAD_ARG;
    if (N == 0) return PDF_OK;
    const size_t lds_b = sizeof(float) * ((size_t)C + (C <= WB ? (size_t)(WB / C) * C : 0));
    k_step1_bwd<<<N, WB, lds_b, static_cast<hipStream_t>(stream)>>>(h, C / h, grad_out, index0_offsets, index1, q, k, grad_q, grad_k);
    return pdf_launch_status();
}

// replaces dot_prod_with_idx_forward_cuda_launcher_v3, libs/pointops2/src/rpe_v2/relative_pos_encoding_cuda_kernel_v2.h
extern "C" int pdf_dot_prod_with_idx_forward_v3(int N, int M, int h, int hdim, unsigned n_max, const float *q,
                                                const int *index_q_offsets, const float *k, const int *index_k,
                                                const float *table_q, const float *table_k, const int *rel_idx, float *output,
                                                void *stream) {
    (void)n_max;
    if (N < 0 || M < 0 || h < 1 || hdim < 1 || !q || !index_q_offsets || !k || !index_k || !table_q || !table_k || !rel_idx || !output)
        return PDF_ERR_BAD_ARG;
    if (N == 0 || M == 0) return PDF_OK;
    k_dot3_fwd<<<N, WB, sizeof(float) * h * hdim, static_cast<hipStream_t>(stream)>>>(h, hdim, q, index_q_offsets, k, index_k, table_q, table_k,
                                                                                  rel_idx, output);
    return pdf_launch_status();
}

// The same op with the table length L (rows of table_q / table_k) known: per-head kernels with the tables in LDS.
extern "C" int pdf_dot_prod_with_idx_forward_v3_l(int N, int M, int h, int hdim, int L, const float *q, const int *index_q_offsets,
                                                  const float *k, const int *index_k, const float *table_q, const float *table_k,
                                                  const int *rel_idx, float *output, void *stream) {
    if (N < 0 || M < 0 || h < 1 || hdim < 1 || L < 1) return PDF_ERR_BAD_ARG;
    if (N == 0 || M == 0) return PDF_OK;
    if (!q || !index_q_offsets || !k || !index_k || !table_q || !table_k || !rel_idx || !output) return PDF_ERR_BAD_ARG;
    const size_t lds = sizeof(float) * (size_t)(2 * L * (hdim * 3 + 1)) + sizeof(int) * (QCH + 1);
    if (lds > 64 * 1024) return pdf_dot_prod_with_idx_forward_v3(N, M, h, hdim, 0, q, index_q_offsets, k, index_k, table_q, table_k, rel_idx, output, stream);
    k_dot3_fwd_h<<<dim3(pdf_divup(N, QCH), h), WB, lds, static_cast<hipStream_t>(stream)>>>(N, h, hdim, L, q, index_q_offsets, k, index_k, table_q,
                                                                                       table_k, rel_idx, output);
    return pdf_launch_status();
}

// replaces dot_prod_with_idx_backward_cuda_launcher_v3.  grad_q overwritten; grad_k, grad_table_q, grad_table_k zeroed by the caller.
extern "C" int pdf_dot_prod_with_idx_backward_v3(int N, int M, int h, int hdim, unsigned n_max, const float *grad_out, const float *q,
                                                 const int *index_q_offsets, const float *k, const int *index_k, const float *table_q,
                                                 const float *table_k, const int *rel_idx, float *grad_q, float *grad_k,
                                                 float *grad_table_q, float *grad_table_k, void *stream) {
    (void)n_max;
    if (N < 0 || M < 0 || h < 1 || hdim < 1 || !grad_out || !q || !index_q_offsets || !k || !index_k || !table_q || !table_k || !rel_idx ||
        !grad_q || !grad_k || !grad_table_q || !grad_table_k)
        return PDF_ERR_BAD_ARG;
    if (N == 0) return PDF_OK;
    k_dot3_bwd<<<N, WB, sizeof(float) * h * hdim, static_cast<hipStream_t>(stream)>>>(h, hdim, grad_out, q, index_q_offsets, k, index_k, table_q,
                                                                                  table_k, rel_idx, grad_q, grad_k, grad_table_q, grad_table_k);
    return pdf_launch_status();
}

extern "C" int pdf_dot_prod_with_idx_backward_v3_l(int N, int M, int h, int hdim, int L, const float *grad_out, const float *q,
                                                   const int *index_q_offsets, const float *k, const int *index_k, const float *table_q,
                                                   const float *table_k, const int *rel_idx, float *grad_q, float *grad_k,
                                                   float *grad_table_q, float *grad_table_k, void *stream) {
    if (N < 0 || M < 0 || h < 1 || hdim < 1 || L < 1) return PDF_ERR_BAD_ARG;
    if (N == 0) return PDF_OK;
    if (!q || !index_q_offsets || !k || !table_q || !table_k || !grad_q || !grad_k || !grad_table_q || !grad_table_k) return PDF_ERR_BAD_ARG;
    const size_t lds = sizeof(float) * (size_t)(4 * L * (hdim * 3 + 1) + QCH * hdim) + sizeof(int) * (QCH + 1);
    if (M == 0 || !pow2_le64(hdim) || lds > 64 * 1024)
        return pdf_dot_prod_with_idx_backward_v3(N, M, h, hdim, 0, grad_out, q, index_q_offsets, k, index_k, table_q, table_k, rel_idx, grad_q,
                                                 grad_k, grad_table_q, grad_table_k, stream);
    const dim3 grid(pdf_divup(N, QCH), h);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hdim == 16 && L <= 64) {   // table gradients as one-hot MFMA products
        // PDFOPS_WA_FACTORED=0: both tables as one-hot matrix-core products in one kernel (k_dot3_bwd_m16, rounds 1-3); default: the query
        // side factored by query (k_dot3_bwd_fq), the key side as one-hot products of one table (k_dot3_bwd_k)
        static const bool factored = [] { const char *v = getenv("PDFOPS_WA_FACTORED"); return !(v && v[0] == '0'); }();
        const size_t lds_fq = sizeof(float) * (size_t)(L * (hdim * 3 + 1) + QF * hdim + QF * 3 * L) + sizeof(int) * (QF + 1);
        const size_t lds_k = sizeof(float) * (size_t)(L * (hdim * 3 + 1));
        if (factored)
            k_dot3_bwd_fq<<<dim3(pdf_divup(N, QF), h), WB, lds_fq, st>>>(N, h, L, grad_out, q, index_q_offsets, table_q, rel_idx, grad_q, grad_table_q);
#define PDF_DOT3_M16(RB_) do { if (factored) k_dot3_bwd_k<RB_><<<grid, WB, lds_k, st>>>(N, h, L, grad_out, index_q_offsets, k, index_k, table_k, rel_idx, \
                                                                                      grad_k, grad_table_k); \
        else k_dot3_bwd_m16<RB_><<<grid, WB, lds, st>>>(N, h, L, grad_out, q, index_q_offsets, k, index_k, table_q, table_k, rel_idx, \
                                                        grad_q, grad_k, grad_table_q, grad_table_k); } while (0)
        switch ((L + 15) / 16) {
            case 1: PDF_DOT3_M16(1); break;
            case 2: PDF_DOT3_M16(2); break;
            case 3: PDF_DOT3_M16(3); break;
            default: PDF_DOT3_M16(4); break;
        }
#undef PDF_DOT3_M16
        return pdf_launch_status();
    }
    k_dot3_bwd_h<<<grid, WB, lds, st>>>(N, h, hdim, L, grad_out, q, index_q_offsets, k, index_k, table_q, table_k, rel_idx, grad_q, grad_k,
                                        grad_table_q, grad_table_k);
    return pdf_launch_status();
}

// replaces attention_step2_with_rel_pos_value_forward_cuda_launcher_v2.  output (N,h,d) is overwritten.
extern "C" int pdf_attention_step2_with_rel_pos_value_forward_v2(int N, int M, int h, int hdim, unsigned n_max, const float *attn,
                                                                 const float *v, const int *index0_offsets, const int *index1,
                                                                 const float *table, const int *rel_idx, float *output, void *stream) {
    (void)n_max;
    if (N < 0 || M < 0 || h < 1 || hdim < 1 || !attn || !v || !index0_offsets || !index1 || !table || !rel_idx || !output) return PDF_ERR_BAD_ARG;
    if (N == 0) return PDF_OK;
    k_step2rv_fwd<<<N, WB, 0, static_cast<hipStream_t>(stream)>>>(h, hdim, attn, v, index0_offsets, index1, table, rel_idx, output);
    return pdf_launch_status();
}

// replaces attention_step2_with_rel_pos_value_backward_cuda_launcher_v2.  grad_attn overwritten; grad_v, grad_table zeroed by the caller.
extern "C" int pdf_attention_step2_with_rel_pos_value_backward_v2(int N, int M, int h, int hdim, unsigned n_max, const float *grad_out,
                                                                  const int *index0_offsets, const int *index1, const float *attn,
                                                                  const float *v, const float *table, const int *rel_idx,
                                                                  float *grad_attn, float *grad_v, float *grad_table, void *stream) {
    (void)n_max;
    if (N < 0 || M < 0 || h < 1 || hdim < 1 || !grad_out || !index0_offsets || !index1 || !attn || !v || !table || !rel_idx || !grad_attn ||
        !grad_v || !grad_table)
        return PDF_ERR_BAD_ARG;
    if (N == 0 || M == 0) return PDF_OK;
    k_step2rv_bwd<<<N, WB, sizeof(float) * h * hdim, static_cast<hipStream_t>(stream)>>>(h, hdim, grad_out, index0_offsets, index1, attn, v, table,
                                                                                     rel_idx, grad_attn, grad_v, grad_table);
    return pdf_launch_status();
}

extern "C" int pdf_attention_step2_with_rel_pos_value_backward_v2_l(int N, int M, int h, int hdim, int L, const float *grad_out,
                                                                    const int *index0_offsets, const int *index1, const float *attn,
                                                                    const float *v, const float *table, const int *rel_idx,
                                                                    float *grad_attn, float *grad_v, float *grad_table, void *stream) {
    if (N < 0 || M < 0 || h < 1 || hdim < 1 || L < 1) return PDF_ERR_BAD_ARG;
    if (N == 0 || M == 0) return PDF_OK;
    if (!grad_out || !index0_offsets || !index1 || !attn || !v || !table || !rel_idx || !grad_attn || !grad_v || !grad_table) return PDF_ERR_BAD_ARG;
    const size_t lds = sizeof(float) * (size_t)(2 * L * (hdim * 3 + 1)) + sizeof(int) * (QCH + 1);
    if (!pow2_le64(hdim) || lds > 64 * 1024)
        return pdf_attention_step2_with_rel_pos_value_backward_v2(N, M, h, hdim, 0, grad_out, index0_offsets, index1, attn, v, table, rel_idx,
                                                                  grad_attn, grad_v, grad_table, stream);
    const dim3 grid(pdf_divup(N, QCH), h);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hdim == 16 && L <= 64) {
        static const bool factored = [] { const char *e = getenv("PDFOPS_WA_FACTORED"); return !(e && e[0] == '0'); }();
        if (factored && !(((size_t)h * hdim) & 3) && !(reinterpret_cast<uintptr_t>(v) & 15)) {   // (float4 reads of the value rows)
            const size_t lds_f = sizeof(float) * (size_t)(L * (hdim * 3 + 1) + QF * hdim + 2 * QF * 3 * L) + sizeof(int) * (QF + 1);
            k_step2rv_bwd_f<<<dim3(pdf_divup(N, QF), h), WB, lds_f, st>>>(N, h, L, grad_out, index0_offsets, index1, attn, v, table, rel_idx,
                                                                         grad_attn, grad_v, grad_table);
            return pdf_launch_status();
        }
#define PDF_S2_M16(RB_) k_step2rv_bwd_m16<RB_><<<grid, WB, lds, st>>>(N, h, L, grad_out, index0_offsets, index1, attn, v, table, rel_idx, grad_attn, \
                                                                     grad_v, grad_table)
        switch ((L + 15) / 16) {
            case 1: PDF_S2_M16(1); break;
            case 2: PDF_S2_M16(2); break;
            case 3: PDF_S2_M16(3); break;
            default: PDF_S2_M16(4); break;
        }
#undef PDF_S2_M16
        return pdf_launch_status();
    }
    k_step2rv_bwd_h<<<grid, WB, lds, st>>>(N, h, hdim, L, grad_out, index0_offsets, index1, attn, v, table, rel_idx, grad_attn, grad_v, grad_table);
    return pdf_launch_status();
}


// Softmax over the edges of every query, per head: x, y (M, h); index0_offsets (N + 1).  Replaces torch_scatter.scatter_softmax
// (src, index_0, dim=0) at stratified_transformer_v1m1_origin.py:322-324 for a CSR-ordered edge list.  h <= 64.
extern "C" int pdf_segment_softmax_forward(int N, int M, int h, const int *index0_offsets, const float *x, float *y, void *stream) {
    if (N == 0 || M == 0) return PDF_OK;
    if (N < 0 || M < 0 || h < 1 || h > 64 || !index0_offsets || !x || !y) return PDF_ERR_BAD_ARG;
    const int hp = heads_pow2(h), per_wave = 64 / hp;
    const long waves = ((long)N + per_wave - 1) / per_wave;
    k_seg_softmax_fwd<<<pdf_divup(waves * 64, WB), WB, 0, static_cast<hipStream_t>(stream)>>>(N, h, hp, index0_offsets, x, y);
    return pdf_launch_status();
}

extern "C" int pdf_segment_softmax_backward(int N, int M, int h, const int *index0_offsets, const float *y, const float *grad_y, float *grad_x,
                                            void *stream) {
    if (N == 0 || M == 0) return PDF_OK;
    if (N < 0 || M < 0 || h < 1 || h > 64 || !index0_offsets || !y || !grad_y || !grad_x) return PDF_ERR_BAD_ARG;
    const int hp = heads_pow2(h), per_wave = 64 / hp;
    const long waves = ((long)N + per_wave - 1) / per_wave;
    k_seg_softmax_bwd<<<pdf_divup(waves * 64, WB), WB, 0, static_cast<hipStream_t>(stream)>>>(N, h, hp, index0_offsets, y, grad_y, grad_x);
    return pdf_launch_status();
}
