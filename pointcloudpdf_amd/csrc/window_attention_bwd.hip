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
                                             float *__restrict__ out, long ldo, float oscale, const int *__restrict__ order) {
    constexpr int CG = HG * D, PQ = CG / 4, RPB = TB / PQ;
    extern __shared__ __attribute__((aligned(16))) float tl[];   // [3][L][CG]
    const int h0 = blockIdx.y * HG;
    if (TABLE) {
        stage_table<HG>(tl, table, L, h, h0);
        __syncthreads();
    }
    const int slot = threadIdx.x / PQ, p = threadIdx.x - slot * PQ;
    const long pos = (long)blockIdx.x * RPB + slot;
    if (slot >= RPB || pos >= N) return;
    const long n = order ? order[pos] : pos;     // visiting order of the owners (window by window: their gathered rows coincide)
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
constexpr int HS = LMAX + 1;        // histogram row stride (odd: the MFMA A-operand reads of 4 owners x 16 rows spread over the banks)
constexpr int PR = 3;               // rounds of 64 entries a wave loads ahead per owner quad (= one LDS window of entries)
constexpr int WIN = 64 * PR;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// grid = (G, h / HG): workgroup (g, hg); partial[(g * HGN + hg)][a][r][c] (3 x L x CG floats).  Every WAVE works alone on a sequence of
// owner QUADS (4 consecutive owners: their entries are one contiguous range of the segment-ordered arrays); no block barrier in the loop.
//   loads    lane = ENTRY: coalesced loads of rel (3 ints) and w (HG floats) of the NEXT quad, issued at the top of a quad together with
//            that quad's owner rows and the bounds of the one after it, so no global round trip is waited for inside a quad; the entries
//            go through a window of the wave's LDS slice (WIN entries; longer quads take further windows, loaded in place).
//   phase 1  lane = (owner of the quad, head, axis) adds its owner's entries into its PRIVATE histogram row, in entry order, four at a
//            time: the four bins are read together and written back in order, an entry's value carrying the earlier ones of the same
//            bin in its group -- one LDS round trip per four entries instead of one per entry, no atomics (ds_add_f32 turned out ~15x
//            slower than a read-modify-write here: profiles/r06_wa_table_ab.txt), a fixed summation order.
//            (Round 5: the same private rows fed by two dependent global loads per 8 entries: 258 us per call at 160k owners / 5 M entries.)
//   phase 2  G[(a, r), c] += sum_s S[s][hh(c)][a][r] * x[s][c] on the matrix cores: per (head, axis) a 64 x 16 block = four 16x16x4 tiles
//            with K = the quad's four owners; the HG x 3 x 4 accumulator tiles stay in registers across all quads of the wave; the four
//            waves of a workgroup add theirs through LDS in wave order at the end.
template <int HG, bool CSC>
__global__ __launch_bounds__(TB) __attribute__((amdgpu_waves_per_eu(2, 8))) void k_table(
    int N, int h, int L, const int *__restrict__ seg_off, const int *__restrict__ seg_edge, const int *__restrict__ rel,
    const float *__restrict__ w, const float *__restrict__ x, long ldx, float xscale, float *__restrict__ partial) {
    constexpr int CG = HG * D, NP = HG * 3, NT = NP * 4;   // NP histogram rows per owner; NT 16 x 16 output tiles (head, axis, row block)
    constexpr int QS = 4 * NP * HS;                        // floats of one wave's histogram rows
    constexpr int XL = 4 * CG / 64;                        // owner-row values per lane (4 owners x CG channels over 64 lanes)
    constexpr int WS = QS + WIN * (3 + HG) + 4 * CG;       // a wave's LDS slice: histograms | entry window (rel, w) | owner rows
    static_assert(NT * 256 <= 4 * WS, "the accumulators of one wave must fit the LDS area for the final sum");
    static_assert(QS % 4 == 0 && WS % 4 == 0, "16-byte stores clear the histograms");
    __shared__ __attribute__((aligned(16))) float lds[4 * WS];
    const int h0 = blockIdx.y * HG;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    float *Sw = lds + wv * WS, *xw = Sw + QS + WIN * (3 + HG);
    int *rel_s = reinterpret_cast<int *>(Sw + QS);
    float *w_s = Sw + QS + WIN * 3;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nquads = (N + 3) >> 2, nw = gridDim.x * 4;
    auto load_off = [&](int qd) { return seg_off[min(4 * qd + min(lane, 4), N)]; };              // lanes 0 .. 4: the quad's five bounds
    auto load_x = [&](int qd, float (&xr)[XL]) {
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const int e = lane + 64 * i, sl = e / CG, c = e - sl * CG;
            xr[i] = x[(size_t)min(4 * qd + sl, N - 1) * ldx + (size_t)h0 * D + c];
        }
    };
    int rr[PR][3];
    float ww[PR][HG];
    auto load_entries = [&](int e0, int e1) {      // entries e0 + lane + 64 k < e1 of a window
#pragma unroll
        for (int k = 0; k < PR; ++k) {
            const int e = e0 + lane + 64 * k;
            if (e < e1) {
                rr[k][0] = rel[(size_t)e * 3]; rr[k][1] = rel[(size_t)e * 3 + 1]; rr[k][2] = rel[(size_t)e * 3 + 2];
                const size_t wi = (size_t)(CSC ? seg_edge[e] : e) * h + h0;
#pragma unroll
                for (int g = 0; g < HG; ++g) ww[k][g] = w[wi + g];
            }
        }
    };
    auto store_entries = [&]() {                   // the window: rel_s[local][3], w_s[local][HG]  (lanes past the end write stale values: never read)
#pragma unroll
        for (int k = 0; k < PR; ++k) {
            const int l = lane + 64 * k;
            rel_s[l * 3] = rr[k][0]; rel_s[l * 3 + 1] = rr[k][1]; rel_s[l * 3 + 2] = rr[k][2];
#pragma unroll
            for (int g = 0; g < HG; ++g) w_s[l * HG + g] = ww[k][g];
        }
    };
    // phase-1 role of the lane: (owner ps of the quad, histogram row pj = head * 3 + axis)
    const int ps = lane / NP, pj = lane - ps * NP, pg = pj / 3, pa = pj - 3 * pg;
    const bool p1 = ps < 4;
    float *prow = Sw + (ps * NP + pj) * HS;
    int qd = blockIdx.x * 4 + wv;
    int off0 = 0, off1 = 0;       // bounds (lanes 0 .. 4) of this quad and of the next
    float xr[XL];
    if (qd < nquads) {
        off0 = load_off(qd);
        load_x(qd, xr);
        if (qd + nw < nquads) off1 = load_off(qd + nw);
        load_entries(__shfl(off0, 0, 64), min(__shfl(off0, 4, 64), __shfl(off0, 0, 64) + WIN));
    }
    for (; qd < nquads; qd += nw) {
        const int o0 = __shfl(off0, 0, 64), o4 = __shfl(off0, 4, 64);
        const int pb = __shfl(off0, min(ps, 3), 64), pe = __shfl(off0, min(ps, 3) + 1, 64);     // the lane's owner row [pb, pe)
        for (int e = lane; e < QS / 4; e += 64) reinterpret_cast<float4 *>(Sw)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const int e = lane + 64 * i;
            xw[e] = 4 * qd + e / CG < N ? xr[i] * xscale : 0.f;
        }
        store_entries();
        // everything the NEXT quad needs from global memory: in flight from here on
        const int qn = qd + nw, qnn = qn + nw;
        int off2 = 0;
        if (qn < nquads) {
            load_x(qn, xr);
            const int n0 = __shfl(off1, 0, 64);
            load_entries(n0, min(__shfl(off1, 4, 64), n0 + WIN));
            if (qnn < nquads) off2 = load_off(qnn);
        }
        for (int wb = o0; wb < o4; wb += WIN) {
            if (wb != o0) {                       // a further window of a long quad: loaded in place (rare; the next quad's registers are in use,
                const int we = min(o4, wb + WIN);  // so this one goes straight to LDS)
                for (int l = lane; l < we - wb; l += 64) {
                    const int e = wb + l;
                    rel_s[l * 3] = rel[(size_t)e * 3]; rel_s[l * 3 + 1] = rel[(size_t)e * 3 + 1]; rel_s[l * 3 + 2] = rel[(size_t)e * 3 + 2];
                    const size_t wi = (size_t)(CSC ? seg_edge[e] : e) * h + h0;
#pragma unroll
                    for (int g = 0; g < HG; ++g) w_s[l * HG + g] = w[wi + g];
                }
            }
            if (p1) {
                const int t1 = min(pe, wb + WIN);
                int t = max(pb, wb);
                // four entries at a time: their bins are read together and written back in order; the NEXT four entries' (bin, value)
                // are read from the window before the write-back (the compiler keeps LDS accesses in program order)
                int b[4];
                float v[4];
                auto fetch = [&](int tt) {
                    const int l = tt - wb;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { b[i] = rel_s[(l + i) * 3 + pa]; v[i] = w_s[(l + i) * HG + pg]; }
                };
                if (t + 3 < t1) fetch(t);
                for (; t + 3 < t1; t += 4) {
                    const int b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
                    const float w0 = v[0], w1 = v[1], w2 = v[2], w3 = v[3];
                    const float s0 = prow[b0], s1 = prow[b1], s2 = prow[b2], s3 = prow[b3];
                    if (t + 7 < t1) fetch(t + 4);
                    const float c1 = b1 == b0 ? w0 + w1 : w1;
                    const float c2 = b2 == b1 ? c1 + w2 : (b2 == b0 ? w0 + w2 : w2);
                    const float c3 = b3 == b2 ? c2 + w3 : (b3 == b1 ? c1 + w3 : (b3 == b0 ? w0 + w3 : w3));
                    prow[b0] = s0 + w0; prow[b1] = s1 + c1; prow[b2] = s2 + c2; prow[b3] = s3 + c3;
                }
                for (; t < t1; ++t) prow[rel_s[(t - wb) * 3 + pa]] += w_s[(t - wb) * HG + pg];
            }
        }
        // phase 2: A[i = row][k = owner lk] from the histograms, B[k = owner lk][j = channel] from the owner rows
        float bv[HG];
#pragma unroll
        for (int g = 0; g < HG; ++g) bv[g] = xw[lk * CG + g * D + li];
#pragma unroll
        for (int t = 0; t < NP; ++t)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
                acc[t * 4 + rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(Sw[(lk * NP + t) * HS + rb * 16 + li], bv[t / 3], acc[t * 4 + rb], 0, 0, 0);
        off0 = off1; off1 = off2;
    }
    // the workgroup's slab: the four waves' accumulators added in wave order through LDS (element (tile, v, lane) at (4 tile + v) 64 + lane)
    __syncthreads();
    for (int k = 0; k < 4; ++k) {
        if (wv == k) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    float *p = lds + (4 * t + v) * 64 + lane;
                    *p = k == 0 ? acc[t][v] : *p + acc[t][v];
                }
        }
        __syncthreads();
    }
    float *dst = partial + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 3 * L * CG;
    for (int e = threadIdx.x; e < NT * 256; e += TB) {
        const int ln = e & 63, v = (e >> 6) & 3, t = e >> 8, rb = t & 3, tt = t >> 2;      // D[i = 4 (lane / 16) + v][j = lane % 16]
        const int r = rb * 16 + 4 * (ln >> 4) + v;
        if (r < L) dst[((size_t)(tt % 3) * L + r) * CG + (tt / 3) * D + (ln & 15)] = lds[e];
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
// The chunk's queries are positions q0 .. q0 + nq - 1 of `order` (null: the queries themselves): their edge ranges need not be adjacent,
// so the lanes walk a VIRTUAL edge index over the chunk (pref: exclusive prefix of the row lengths, qst: first edge of every row).
template <int QN>
__device__ __forceinline__ void chunk_rows(int *pref, int *qst, int *qid, const int *__restrict__ offsets, const int *__restrict__ order, int q0, int nq) {
    static_assert(QN <= 64, "one wave scans the chunk's row lengths");
    if (threadIdx.x < 64) {
        const int jj = threadIdx.x;
        const int q = jj < nq ? (order ? order[q0 + jj] : q0 + jj) : 0;
        const int st = jj < nq ? offsets[q] : 0;
        int len = jj < nq ? offsets[q + 1] - st : 0, sum = len;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(sum, o, 64);
            if (jj >= o) sum += u;
        }
        if (jj < QN) { qid[jj] = q; qst[jj] = st; pref[jj + 1] = sum; }
        if (jj == 0) pref[0] = 0;
    }
    __syncthreads();
}

__global__ __launch_bounds__(TB) void k_grad_attn(int N, int h, int L, const float *__restrict__ go, long ldg, const int *__restrict__ offsets,
                                                  const int *__restrict__ index1, const float *__restrict__ v, long ldv, const float *__restrict__ table,
                                                  const int *__restrict__ rel, float *__restrict__ grad_attn, const int *__restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *tb = sm, *gr = tb + 3 * L * D, *P = gr + QF * D;     // tb [3][L][16] | gr [QF][16] | P [QF][3 L]
    int *offs = reinterpret_cast<int *>(P + QF * 3 * L), *qst = offs + QF + 1, *qid = qst + QF;
    const int hh = blockIdx.y, q0 = blockIdx.x * QF, nq = min(QF, N - q0), W = 3 * L;
    chunk_rows<QF>(offs, qst, qid, offsets, order, q0, nq);
    stage_table<1>(tb, table, L, h, hh);
    for (int e = threadIdx.x; e < QF * D; e += TB) gr[e] = e < nq * D ? go[(size_t)qid[e / D] * ldg + (size_t)hh * D + e % D] : 0.f;
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
    const int total = offs[nq];
    int ql = 0;
    for (int t = threadIdx.x; t < total; t += TB) {             // (a lane's virtual edges ascend: the query is found by stepping on)
        while (offs[ql + 1] <= t) ++ql;
        const int m = qst[ql] + (t - offs[ql]);
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
// TK = false: the same pass without the key-side table term -- grad_attn[m, hh] = <grad_out[q(m), hh, :], v[j, hh, :] + T_v(m, hh, :)>
// (attention_step2_with_rel_pos_value_v2 backward, relative_pos_encoding_cuda_kernel_v2.cu:441-470) is this expression with q := grad_out,
// k := v, table_q := table_v.  (Rounds 5-6 factored the table term through per-query projections P_q[a][r]: 3 L x 16 multiply-adds per
// query against 3 x 16 per edge -- more work below 64 edges per row, and rows hold ~31: 298 -> see profiles/r06_wa_table_ab.txt.)
constexpr int QL = 64;
template <bool TK>
__global__ __launch_bounds__(TB) void k_logits_fwd(int N, int h, int L, const float *__restrict__ q, long ldq, const float *__restrict__ k, long ld, float qscale,
                                                   const int *__restrict__ offsets, const int *__restrict__ index1,
                                                   const float *__restrict__ table_q, const float *__restrict__ table_k,
                                                   const int *__restrict__ rel, float *__restrict__ out, const int *__restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *tq = sm, *tk = tq + 3 * L * D, *qs = tk + (TK ? 3 * L * D : 0);     // [3][L][16] (x 2 with the key-side table) | qs [QL][16]
    int *offs = reinterpret_cast<int *>(qs + QL * D), *qst = offs + QL + 1, *qid = qst + QL;
    const int hh = blockIdx.y, q0 = blockIdx.x * QL, nq = min(QL, N - q0);
    chunk_rows<QL>(offs, qst, qid, offsets, order, q0, nq);
    stage_table<1>(tq, table_q, L, h, hh);
    if (TK) stage_table<1>(tk, table_k, L, h, hh);
    for (int e = threadIdx.x; e < QL * D; e += TB) qs[e] = e < nq * D ? q[(size_t)qid[e / D] * ldq + (size_t)hh * D + e % D] * qscale : 0.f;
    __syncthreads();
    const int total = offs[nq];
    int ql = 0;
    for (int t = threadIdx.x; t < total; t += TB) {
        while (offs[ql + 1] <= t) ++ql;
        const int m = qst[ql] + (t - offs[ql]);
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
            const float4 tqv = add4(add4(a0[c4], a1[c4]), a2[c4]);
            if (TK) {
                const float4 tkv = add4(add4(b0[c4], b1[c4]), b2[c4]);
                sum += qv.x * (kv.x + tqv.x) + kv.x * tkv.x;
                sum += qv.y * (kv.y + tqv.y) + kv.y * tkv.y;
                sum += qv.z * (kv.z + tqv.z) + kv.z * tkv.z;
                sum += qv.w * (kv.w + tqv.w) + kv.w * tkv.w;
            } else {
                sum += qv.x * (kv.x + tqv.x);
                sum += qv.y * (kv.y + tqv.y);
                sum += qv.z * (kv.z + tqv.z);
                sum += qv.w * (kv.w + tqv.w);
            }
        }
        out[(size_t)m * h + hh] = sum;
    }
}

// dst[m, :] = src[edge[m], :] for (M, h) edge scalars (the key-ordered copy the CSC passes read sequentially): one thread per element
__global__ __launch_bounds__(TB) void k_permute_edges(long total, int h, const float *__restrict__ src, const int *__restrict__ edge, float *__restrict__ dst) {
    const long t = (long)blockIdx.x * TB + threadIdx.x;
    if (t >= total) return;
    const long m = t / h;
    dst[t] = src[(size_t)edge[m] * h + (int)(t - m * h)];
}

static inline int head_group(int h) { return h % 3 == 0 ? 3 : 1; }
static inline int table_head_group(int h) { return head_group(h); }
static inline int table_grid(int N, int h) {
    const int nchunks = (N + 15) / 16, hgn = h / table_head_group(h);      // (a workgroup = four waves = four owner quads at a time)
    static const int target = [] { const char *e = getenv("PDFOPS_WA_TABLE_GRID"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 512; }();
    int g = target / (hgn < 1 ? 1 : hgn);       // ~512 workgroups in flight over all head groups
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
extern "C" int pdf_wa_segment_rows_ordered(int N, int h, int d, int L, const int *seg_off, const int *seg_edge, const int *other, const int *rel,
                                           const float *w, const float *X, long ldx, float xscale, const float *table, float *out, long ldo,
                                           float oscale, const int *order, void *stream);
extern "C" int pdf_wa_segment_rows(int N, int h, int d, int L, const int *seg_off, const int *seg_edge, const int *other, const int *rel,
                                   const float *w, const float *X, long ldx, float xscale, const float *table, float *out, long ldo, float oscale,
                                   void *stream) {
    return pdf_wa_segment_rows_ordered(N, h, d, L, seg_off, seg_edge, other, rel, w, X, ldx, xscale, table, out, ldo, oscale, nullptr, stream);
}

// The same with a visiting order of the owners (order: a permutation of 0 .. N - 1, or null): workgroups that run side by side then work
// on owners of one window, whose segments gather the same rows (the window's members) -- L2 hits instead of Infinity-Cache traffic.
// Every owner's sum is formed as before: results are identical.
extern "C" int pdf_wa_segment_rows_ordered(int N, int h, int d, int L, const int *seg_off, const int *seg_edge, const int *other, const int *rel,
                                           const float *w, const float *X, long ldx, float xscale, const float *table, float *out, long ldo,
                                           float oscale, const int *order, void *stream) {
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
            N, h, L, seg_off, seg_edge, other, rel, w, X, ldx, xscale, table, out, ldo, oscale, order); } while (0)
#define PDF_WB_ROWS2(HG_, CSC_) do { \
        if (X && table) PDF_WB_ROWS(HG_, CSC_, true, true); else if (X) PDF_WB_ROWS(HG_, CSC_, true, false); else PDF_WB_ROWS(HG_, CSC_, false, true); } while (0)
    if (hg == 3) { if (seg_edge) PDF_WB_ROWS2(3, true); else PDF_WB_ROWS2(3, false); }
    else { if (seg_edge) PDF_WB_ROWS2(1, true); else PDF_WB_ROWS2(1, false); }
#undef PDF_WB_ROWS2
#undef PDF_WB_ROWS
    return pdf_launch_status();
}

// grad_attn (M, h), every element written:  < grad_out[q(m), hh, :], v[index1[m], hh, :] + T(m, hh, :) >
extern "C" int pdf_wa_grad_attn_ordered(int N, int M, int h, int d, int L, const float *grad_out, long ldg, const int *offsets, const int *index1,
                                        const float *v, long ldv, const float *table, const int *rel, float *grad_attn, const int *order, void *stream);
extern "C" int pdf_wa_grad_attn(int N, int M, int h, int d, int L, const float *grad_out, long ldg, const int *offsets, const int *index1,
                                const float *v, long ldv, const float *table, const int *rel, float *grad_attn, void *stream) {
    return pdf_wa_grad_attn_ordered(N, M, h, d, L, grad_out, ldg, offsets, index1, v, ldv, table, rel, grad_attn, nullptr, stream);
}
// (order: the queries a workgroup takes, as for pdf_wa_segment_rows_ordered; every element of grad_attn is the same dot product)
extern "C" int pdf_wa_grad_attn_ordered(int N, int M, int h, int d, int L, const float *grad_out, long ldg, const int *offsets, const int *index1,
                                        const float *v, long ldv, const float *table, const int *rel, float *grad_attn, const int *order, void *stream) {
    if (N < 0 || M < 0 || h < 1 || d < 1 || L < 1 || !grad_out || !offsets || !index1 || !v || !table || !rel || !grad_attn) return PDF_ERR_BAD_ARG;
    if (N == 0 || M == 0) return PDF_OK;
    if (d != wb::D || L > wb::LMAX || (reinterpret_cast<uintptr_t>(v) & 15) || (ldv & 3) || ldv < (long)h * d || ldg < (long)h * d) return PDF_ERR_UNSUPPORTED;
    static const bool factored = [] { const char *e = getenv("PDFOPS_WA_GRAD_ATTN"); return e && e[0] == 'f'; }();   // (A/B: the round-5 form)
    if (!factored) {
        if (reinterpret_cast<uintptr_t>(v) & 15) return PDF_ERR_UNSUPPORTED;
        const size_t lds1 = sizeof(float) * (size_t)(3 * L * wb::D + wb::QL * wb::D) + sizeof(int) * (3 * wb::QL + 1);
        wb::k_logits_fwd<false><<<dim3((unsigned)((N + wb::QL - 1) / wb::QL), (unsigned)h), wb::TB, lds1, static_cast<hipStream_t>(stream)>>>(
            N, h, L, grad_out, ldg, v, ldv, 1.f, offsets, index1, table, nullptr, rel, grad_attn, order);
        return pdf_launch_status();
    }
    const size_t lds = sizeof(float) * (size_t)(3 * L * wb::D + wb::QF * wb::D + wb::QF * 3 * L) + sizeof(int) * (3 * wb::QF + 1);
    wb::k_grad_attn<<<dim3((unsigned)((N + wb::QF - 1) / wb::QF), (unsigned)h), wb::TB, lds, static_cast<hipStream_t>(stream)>>>(
        N, h, L, grad_out, ldg, offsets, index1, v, ldv, table, rel, grad_attn, order);
    return pdf_launch_status();
}

// logits (M, h) = attention_step1_v2(q, k) + dot_prod_with_idx_v3(q, k, table_q, table_k), every element written
extern "C" int pdf_wa_logits_forward_ordered(int N, int M, int h, int d, int L, const float *q, const float *k, long ld, float qscale, const int *offsets,
                                             const int *index1, const float *table_q, const float *table_k, const int *rel, float *out,
                                             const int *order, void *stream);
extern "C" int pdf_wa_logits_forward(int N, int M, int h, int d, int L, const float *q, const float *k, long ld, float qscale, const int *offsets,
                                     const int *index1, const float *table_q, const float *table_k, const int *rel, float *out, void *stream) {
    return pdf_wa_logits_forward_ordered(N, M, h, d, L, q, k, ld, qscale, offsets, index1, table_q, table_k, rel, out, nullptr, stream);
}
extern "C" int pdf_wa_logits_forward_ordered(int N, int M, int h, int d, int L, const float *q, const float *k, long ld, float qscale, const int *offsets,
                                             const int *index1, const float *table_q, const float *table_k, const int *rel, float *out,
                                             const int *order, void *stream) {
    if (N < 0 || M < 0 || h < 1 || d < 1 || L < 1 || !q || !k || !offsets || !index1 || !table_q || !table_k || !rel || !out) return PDF_ERR_BAD_ARG;
    if (N == 0 || M == 0) return PDF_OK;
    if (d != wb::D || L > wb::LMAX || (reinterpret_cast<uintptr_t>(k) & 15) || (ld & 3) || ld < (long)h * d) return PDF_ERR_UNSUPPORTED;
    const size_t lds = sizeof(float) * (size_t)(6 * L * wb::D + wb::QL * wb::D) + sizeof(int) * (3 * wb::QL + 1);
    wb::k_logits_fwd<true><<<dim3((unsigned)((N + wb::QL - 1) / wb::QL), (unsigned)h), wb::TB, lds, static_cast<hipStream_t>(stream)>>>(
        N, h, L, q, ld, k, ld, qscale, offsets, index1, table_q, table_k, rel, out, order);
    return pdf_launch_status();
}

// dst (M, h) = src[edge[m], :]: the edge scalars of a pass in the order of another edge list (edge: ids into src's rows)
extern "C" int pdf_wa_permute_edges(int M, int h, const float *src, const int *edge, float *dst, void *stream) {
    if (M < 0 || h < 1 || (M > 0 && (!src || !edge || !dst))) return PDF_ERR_BAD_ARG;
    if (M == 0) return PDF_OK;
    const long total = (long)M * h;
    wb::k_permute_edges<<<(unsigned)((total + wb::TB - 1) / wb::TB), wb::TB, 0, static_cast<hipStream_t>(stream)>>>(total, h, src, edge, dst);
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
    const int hg = wb::table_head_group(h), hgn = h / hg, G = wb::table_grid(N, h);
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
