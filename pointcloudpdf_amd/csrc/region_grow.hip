// Region growing of the PDF pseudo-label pass (pointcept/recognizers/ours/pointpdf_v1m1_base.py:233-305) and the region's edge list
// (:309-335, ours/utils.py:7-43) without a host in the loop: one workgroup per scene runs ALL growth rounds.
//
// Upstream grows the seed list round by round with host-side control flow: candidates = unique(neighbors[graph]) minus the members,
// ranked by 0.4 * closeness to the region's centroid + 0.6 * similarity of their score to the region's (mean of the scores between the
// region's 10 % and 60 % quantiles), the best 40 % join, until the region's mean score passes `stop` or nothing changes -- three host
// reads per round, ~35 short launches, and the device idles while the host walks through them (rounds 1-4 of this repo kept that
// shape: 6.9 ms of host-paced work per 150k-point scene).  Everything a round needs is a reduction, a k-th order statistic or a
// mask update over the scene's points, so one 1024-thread workgroup keeps the whole loop on the device:
//   state      mult[i] = multiplicity of point i in the region list (the seed list may hold repeats -- drawn with replacement, :206 --
//              and upstream's statistics of the first round count them; a grown region is a set: mult in {0, 1})
//   per round  block reductions in double in a fixed order (length, mean score, centroid), candidate marks (plain stores), the two
//              quantiles and the 40 % cut by an exact 4-pass radix select over order-preserving keys (LDS histograms, integer adds),
//              similarity in upstream's operation order with contraction off, mask update
// The result is the same SET as upstream's loop whenever no decision sits at a float tie (summation order differs from torch's).
// k_region_edges then lists the region's nodes in ascending order and the (row, col, weight) entries of its neighbour graph in
// (row, col) order -- what scipy's csr_matrix holds upstream -- with the counts left in device memory for the graph kernels
// (csrc/graph_prune.hip: pdf_graph_forest_dev / pdf_gmm2_1d_dev).
#include "pdfops_common.h"

namespace rg {

constexpr int T = 1024;
constexpr int NW = T / 64;

__device__ __forceinline__ unsigned okey(float f) {   // float -> unsigned with the same order
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float unkey(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// fixed-order block sums of K doubles: every thread gets the totals
template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double *lds) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double s = v[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if ((threadIdx.x & 63) == 0) lds[(threadIdx.x >> 6) * K + k] = s;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double s = 0.0;
        for (int w = 0; w < NW; ++w) s += lds[w * K + k];
        v[k] = s;
    }
}
__device__ __forceinline__ float block_min(float v, float *lds) {
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_down(v, o, 64));
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = lds[0];
    for (int w = 1; w < NW; ++w) r = fminf(r, lds[w]);
    return r;
}
__device__ __forceinline__ float block_max(float v, float *lds) {
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = lds[0];
    for (int w = 1; w < NW; ++w) r = fmaxf(r, lds[w]);
    return r;
}
__device__ __forceinline__ long long block_sum_ll(long long v, long long *lds) {
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    long long r = 0;
    for (int w = 0; w < NW; ++w) r += lds[w];
    return r;
}

// The k-th smallest (k >= 1, 1-indexed) key of the weighted multiset {key[i] with weight wt(i)}: 4 passes of 8 bits, most significant
// first.  `wt(i)` returns 0 for elements outside the multiset.  Exact (integer histogram).
template <typename W, typename Kf>
__device__ __forceinline__ unsigned radix_select(int n, long long k, W wt, Kf keyf, int *hist /* [256] */, unsigned *bcast /* [2] */) {
    unsigned prefix = 0;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int b = threadIdx.x; b < 256; b += T) hist[b] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += T) {
            const int w = wt(i);
            if (w > 0) {
                const unsigned key = keyf(i);
                if (pass == 0 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(key >> shift) & 255u], w);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            long long acc = 0;
            int b = 0;
            for (; b < 255; ++b) {
                if (acc + hist[b] >= k) break;
                acc += hist[b];
            }
            bcast[0] = (unsigned)b;
            bcast[1] = (unsigned)(k - acc);   // rank inside the bucket
        }
        __syncthreads();
        prefix |= bcast[0] << shift;
        k = (long long)bcast[1];
        __syncthreads();
    }
    return prefix;
}

__device__ __forceinline__ float fsub(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float fadd(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float fmul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float fdivr(float a, float b) { return __fdiv_rn(a, b); }
__device__ __forceinline__ float norm3(float dx, float dy, float dz) { return __fsqrt_rn(fadd(fadd(fmul(dx, dx), fmul(dy, dy)), fmul(dz, dz))); }

// grid = scenes.  Per scene s: points [start[s], start[s] + n), neighbour ids LOCAL to the scene (-1 padded).
//   mult   (N) int32  in: multiplicity of every point in the seed list; out: 1 on the region's points (or the seed multiplicities when the
//                     region never grew)
//   cand   (N) uint8  scratch;  sim (N) float scratch
//   info   (scenes, 4) int32 out: [rounds run, grew (0 / 1), length of the region list, distinct points]
__global__ __launch_bounds__(T) void k_grow(const int *__restrict__ starts, const int *__restrict__ sizes, const float *__restrict__ coord,
                                            const float *__restrict__ score, const long long *__restrict__ neighbors, int nsample,
                                            const float *__restrict__ stop, int slide_window, int max_rounds, int *__restrict__ mult,
                                            unsigned char *__restrict__ cand, float *__restrict__ sim, int *__restrict__ info) {
    __shared__ double dl[NW * 5];
    __shared__ float fl[NW];
    __shared__ long long ll[NW];
    __shared__ int hist[256];
    __shared__ unsigned bc[2];
    __shared__ int wcount[NW];
    const int s = blockIdx.x, t = threadIdx.x;
    const long s0 = starts[s];
    const int n = sizes[s];
    coord += s0 * 3; score += s0; neighbors += s0 * nsample; mult += s0; cand += s0; sim += s0;
    const float stop_s = stop[s];
    int rounds = 0, grew = 0;
    long long L = 0, distinct = 0;
    for (;;) {
        // ---- the region list: length (with repeats), mean score, centroid
        double a[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
        long long dcount = 0;
        for (int i = t; i < n; i += T) {
            const int m = mult[i];
            if (m > 0) {
                const double dm = (double)m;
                a[0] += dm; a[1] += dm * (double)score[i];
                a[2] += dm * (double)coord[3 * i]; a[3] += dm * (double)coord[3 * i + 1]; a[4] += dm * (double)coord[3 * i + 2];
                ++dcount;
            }
        }
        block_sum<5>(a, dl);
        distinct = block_sum_ll(dcount, ll);
        L = (long long)a[0];
        if (L == 0 || rounds >= max_rounds) break;
        const float g_mean = (float)(a[1] / a[0]);
        if (g_mean > stop_s && (double)L > 0.01 * (double)n && L > 50) break;
        const float cx = (float)(a[2] / a[0]), cy = (float)(a[3] / a[0]), cz = (float)(a[4] / a[0]);
        // ---- candidates: neighbours of the members that are not members
        for (int i = t; i < n; i += T) cand[i] = 0;
        __syncthreads();
        for (int i = t; i < n; i += T) {
            if (mult[i] > 0) {
                const long long *row = neighbors + (size_t)i * nsample;
                for (int k = 0; k < nsample; ++k) {
                    const long long nb = row[k];
                    if (nb >= 0 && nb < n) cand[nb] = 1;
                }
            }
        }
        __syncthreads();
        long long nc_l = 0;
        for (int i = t; i < n; i += T) {
            if (mult[i] > 0) cand[i] = 0;
            nc_l += cand[i];
        }
        const long long nc = block_sum_ll(nc_l, ll);
        // ---- the score the candidates are compared with: mean of the region's scores between its 10 % and 60 % quantiles
        float lo, hi;
        if (slide_window) {
            const long long k1 = (long long)((double)L * 0.1), k2 = (long long)((double)L * 0.6);   // int(len * 0.1), int(len * 0.6)
            auto wt = [&](int i) { return mult[i]; };
            auto kf = [&](int i) { return okey(score[i]); };
            lo = unkey(radix_select(n, k1 < 1 ? 1 : k1, wt, kf, hist, bc));
            hi = unkey(radix_select(n, k2 < 1 ? 1 : k2, wt, kf, hist, bc));
        } else {
            float mn = INFINITY, mx = -INFINITY;
            for (int i = t; i < n; i += T)
                if (mult[i] > 0) { mn = fminf(mn, score[i]); mx = fmaxf(mx, score[i]); }
            lo = block_min(mn, fl);
            hi = block_max(mx, fl);
        }
        double r2[2] = {0.0, 0.0};
        for (int i = t; i < n; i += T) {
            const int m = mult[i];
            if (m > 0 && score[i] >= lo && score[i] <= hi) { r2[0] += (double)m; r2[1] += (double)m * (double)score[i]; }
        }
        block_sum<2>(r2, dl);
        const float ref = (float)(r2[1] / r2[0]);
        // ---- similarity of every candidate: 0.4 * (1 - (dist - min) / (max - min + 1e-3)) + 0.6 * exp(-|score - ref|)
        float dmn = INFINITY, dmx = -INFINITY;
        for (int i = t; i < n; i += T) {
            if (cand[i]) {
                const float d = norm3(fsub(coord[3 * i], cx), fsub(coord[3 * i + 1], cy), fsub(coord[3 * i + 2], cz));
                sim[i] = d;
                dmn = fminf(dmn, d); dmx = fmaxf(dmx, d);
            }
        }
        const float dmin = block_min(dmn, fl), dmax = block_max(dmx, fl);
        const float den = fadd(fsub(dmax, dmin), 1e-3f);
        for (int i = t; i < n; i += T) {
            if (cand[i]) {
                const float ds = fsub(1.0f, fdivr(fsub(sim[i], dmin), den));
                const float cs = expf(-fabsf(fsub(score[i], ref)));
                sim[i] = fadd(fmul(0.4f, ds), fmul(0.6f, cs));
            }
        }
        __syncthreads();
        // ---- the best 40 % join: k-th largest similarity = (nc - k + 1)-th smallest
        const long long k = (long long)((double)nc * 0.4);   // int(sim.numel() * 0.4)
        long long added = 0;
        if (k > 0) {
            auto wt = [&](int i) { return (int)cand[i]; };
            auto kf = [&](int i) { return okey(sim[i]); };
            const unsigned cut = radix_select(n, nc - k + 1, wt, kf, hist, bc);
            long long above_l = 0;
            for (int i = t; i < n; i += T) above_l += (cand[i] && okey(sim[i]) > cut) ? 1 : 0;
            const long long above = block_sum_ll(above_l, ll);
            const long long need = k - above;   // >= 1 entries at the cut value join
            long long ties_l = 0;
            for (int i = t; i < n; i += T) ties_l += (cand[i] && okey(sim[i]) == cut) ? 1 : 0;
            const long long ties = block_sum_ll(ties_l, ll);
            if (ties == need) {   // (always, but for exact float ties across the cut)
                for (int i = t; i < n; i += T) cand[i] = (cand[i] && okey(sim[i]) >= cut) ? 2 : 0;   // 2 = joins
            } else {              // of the entries AT the cut, the lowest ids until k are reached: ordered walk in chunks of T
                if (t == 0) bc[0] = 0u;
                __syncthreads();
                for (int base = 0; base < n; base += T) {
                    const int i = base + t;
                    const bool tie = i < n && cand[i] && okey(sim[i]) == cut;
                    const unsigned long long bal = __ballot(tie);
                    if ((t & 63) == 0) wcount[t >> 6] = __popcll(bal);
                    __syncthreads();
                    int before = (int)bc[0];
                    for (int w = 0; w < (t >> 6); ++w) before += wcount[w];
                    before += __popcll(bal & ((1ull << (t & 63)) - 1ull));
                    const bool take = i < n && cand[i] && (okey(sim[i]) > cut || (tie && before < need));
                    if (i < n) cand[i] = take ? 2 : 0;
                    __syncthreads();
                    if (t == 0) { int tot = 0; for (int w = 0; w < NW; ++w) tot += wcount[w]; bc[0] += (unsigned)tot; }
                    __syncthreads();
                }
            }
            __syncthreads();
            added = k;
        }
        ++rounds;
        // ---- upstream: grown = unique(cat(graph, chosen)); `if grown.shape[0] == graph.shape[0]: break` compares the new SET with the old LIST
        if (distinct + added == L) break;
        for (int i = t; i < n; i += T) mult[i] = (mult[i] > 0 || (k > 0 && cand[i] == 2)) ? 1 : 0;
        grew = 1;
        __syncthreads();
    }
    if (t == 0) { info[4 * s] = rounds; info[4 * s + 1] = grew; info[4 * s + 2] = (int)L; info[4 * s + 3] = (int)distinct; }
}

// ---------------------------------------------------------------------------------------------------------------- the region's graph
// Per scene (grid = scenes): node list = the region's points in ascending order (LOCAL ids, int64: what pdf_graph_forest takes) and the
// entries (u, v, w) of its neighbour graph in (row, col) order: v among u's neighbours with v != -1, v != u, v in the region;
// w = mult[u] * (0.4 * dist_sim + 0.6 * conf_sim) (ours/utils.py:7-43; repeated rows of a seed list are SUMMED by upstream's csr_matrix).
// Row minima / maxima of the distance run over ALL of u's neighbour slots with invalid ones (padding, u itself) counted as 0, as upstream's
// masked tensor does.  counts (scenes, 4) int32: [nodes, entries, any -1 padding among the region's rows, smallest id touched].
// Capacities: nodes_out / the entry arrays hold sizes[s] resp. sizes[s] * nsample elements per scene at offset starts[s] (* nsample).
__global__ __launch_bounds__(T) void k_region_edges(const int *__restrict__ starts, const int *__restrict__ sizes, const float *__restrict__ coord,
                                                    const float *__restrict__ msp, const long long *__restrict__ neighbors, int nsample,
                                                    const int *__restrict__ mult, long long *__restrict__ nodes_out, long long *__restrict__ eu,
                                                    long long *__restrict__ ev, float *__restrict__ ew, unsigned char *__restrict__ touched,
                                                    int *__restrict__ counts) {
    __shared__ int wsum[NW], wm[NW], wc[NW];
    __shared__ int carry[2];
    __shared__ int flags[2];
    const int s = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const long s0 = starts[s];
    const int n = sizes[s];
    coord += s0 * 3; msp += s0; neighbors += s0 * nsample; mult += s0; nodes_out += s0; touched += s0;
    eu += s0 * nsample; ev += s0 * nsample; ew += s0 * nsample;
    if (t == 0) { carry[0] = 0; carry[1] = 0; flags[0] = 0; flags[1] = 0x7fffffff; }
    for (int i = t; i < n; i += T) touched[i] = 0;
    __syncthreads();
    // chunks of T points in ascending order: exclusive prefix of (is member, valid entries of the row) inside the workgroup
    for (int base = 0; base < n; base += T) {
        const int i = base + t;
        const int m = i < n ? mult[i] : 0;
        int cnt = 0;
        bool pad = false;
        if (m > 0) {
            const long long *row = neighbors + (size_t)i * nsample;
            for (int k = 0; k < nsample; ++k) {
                const long long nb = row[k];
                if (nb < 0) pad = true;
                else {
                    touched[nb] = 1;
                    if (nb != i && nb < n && mult[nb] > 0) ++cnt;
                }
            }
            touched[i] = 1;
        }
        if (pad) flags[0] = 1;
        // wave-level inclusive scans of (member, cnt)
        int pm = m > 0 ? 1 : 0, pc = cnt;
        for (int o = 1; o < 64; o <<= 1) {
            const int a = __shfl_up(pm, o, 64), b = __shfl_up(pc, o, 64);
            if (lane >= o) { pm += a; pc += b; }
        }
        if (lane == 63) { wm[wv] = pm; wc[wv] = pc; }
        __syncthreads();
        int om = carry[0], oc = carry[1];
        for (int w = 0; w < wv; ++w) { om += wm[w]; oc += wc[w]; }
        const int my_node = om + pm - (m > 0 ? 1 : 0), my_entry = oc + pc - cnt;
        if (m > 0) {
            nodes_out[my_node] = i;
            const long long *row = neighbors + (size_t)i * nsample;
            const float px = coord[3 * i], py = coord[3 * i + 1], pz = coord[3 * i + 2], si = msp[i];
            float dmin = INFINITY, dmax = -INFINITY;
            for (int k = 0; k < nsample; ++k) {
                const long long nb = row[k];
                const bool valid = nb >= 0 && nb != i;
                const long long j = nb >= 0 ? nb : n - 1;   // (upstream indexes coord[-1]: the last point; its distance is masked to 0 anyway)
                const float d = valid ? norm3(fsub(coord[3 * j], px), fsub(coord[3 * j + 1], py), fsub(coord[3 * j + 2], pz)) : 0.f;
                dmin = fminf(dmin, d); dmax = fmaxf(dmax, d);
            }
            const float den = fadd(fsub(dmax, dmin), 1e-3f);
            int e = my_entry;
            for (int k = 0; k < nsample; ++k) {
                const long long nb = row[k];
                if (nb >= 0 && nb != i && nb < n && mult[nb] > 0) {
                    const float d = norm3(fsub(coord[3 * nb], px), fsub(coord[3 * nb + 1], py), fsub(coord[3 * nb + 2], pz));
                    const float ds = fsub(1.0f, fdivr(fsub(d, dmin), den));
                    const float cs = expf(-fabsf(fsub(msp[nb], si)));
                    const float w = fadd(fmul(0.4f, ds), fmul(0.6f, cs));
                    eu[e] = i; ev[e] = nb; ew[e] = m == 1 ? w : fmul((float)m, w);
                    ++e;
                }
            }
        }
        __syncthreads();
        if (t == T - 1) { carry[0] = om + pm; carry[1] = oc + pc; }
        __syncthreads();
    }
    // smallest touched id (upstream's `unique(...)[1:]` drops it when the rows hold no -1 padding)
    int first = 0x7fffffff;
    for (int i = t; i < n; i += T)
        if (touched[i]) { first = i; break; }
    for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_down(first, o, 64));
    if (lane == 0) wsum[wv] = first;
    __syncthreads();
    if (t == 0) {
        int f = wsum[0];
        for (int w = 1; w < NW; ++w) f = min(f, wsum[w]);
        counts[4 * s] = carry[0]; counts[4 * s + 1] = carry[1]; counts[4 * s + 2] = flags[0]; counts[4 * s + 3] = f;
    }
}

// The chosen entries of the spanning forest, in entry order: tu, tv (ids), tw (weights; +inf beyond the tree so that a plain sort of the
// scene's slice brings the tree's weights to the front), tdev (scenes, 2) = [nodes, tree edges] for the second graph kernel.
__global__ __launch_bounds__(T) void k_tree_edges(const int *__restrict__ starts, const int *__restrict__ sizes, int nsample,
                                                  const int *__restrict__ counts, const unsigned char *__restrict__ chosen,
                                                  const long long *__restrict__ eu, const long long *__restrict__ ev, const float *__restrict__ ew,
                                                  long long *__restrict__ tu, long long *__restrict__ tv, float *__restrict__ tw, int *__restrict__ tdev) {
    __shared__ int wcnt[NW];
    __shared__ int carry;
    const int s = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const long s0 = starts[s];
    const int n = sizes[s], E = counts[4 * s + 1];
    chosen += s0 * nsample; eu += s0 * nsample; ev += s0 * nsample; ew += s0 * nsample;
    tu += s0; tv += s0; tw += s0;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < E; base += T) {
        const int e = base + t;
        const bool on = e < E && chosen[e] != 0;
        const unsigned long long bal = __ballot(on);
        if (lane == 0) wcnt[wv] = __popcll(bal);
        __syncthreads();
        int pos = carry;
        for (int w = 0; w < wv; ++w) pos += wcnt[w];
        pos += __popcll(bal & ((1ull << lane) - 1ull));
        if (on && pos < n) { tu[pos] = eu[e]; tv[pos] = ev[e]; tw[pos] = ew[e]; }
        __syncthreads();
        if (t == 0) { int tot = 0; for (int w = 0; w < NW; ++w) tot += wcnt[w]; carry += tot; }
        __syncthreads();
    }
    const int m = min(carry, n);
    for (int i = m + t; i < n; i += T) { tu[i] = 0; tv[i] = 0; tw[i] = INFINITY; }
    if (t == 0) { tdev[2 * s] = counts[4 * s]; tdev[2 * s + 1] = m; }
}

}  // namespace rg

// Region growing of every scene of a batch, all rounds on the device.  starts / sizes (scenes) int32: the scenes' point ranges; neighbors
// (N, nsample) int64 LOCAL ids (-1 padded); stop (scenes) float; mult (N) int32 in / out; cand (N) bytes, sim (N) floats: scratch;
// info (scenes, 4) int32 out [rounds, grew, list length, distinct points].
extern "C" int pdf_region_grow(int scenes, const int *starts, const int *sizes, const float *coord, const float *score, const long long *neighbors,
                               int nsample, const float *stop, int slide_window, int max_rounds, int *mult, unsigned char *cand, float *sim,
                               int *info, void *stream) {
    if (scenes < 0 || nsample < 1 || max_rounds < 0) return PDF_ERR_BAD_ARG;
    if (scenes == 0) return PDF_OK;
    if (!starts || !sizes || !coord || !score || !neighbors || !stop || !mult || !cand || !sim || !info) return PDF_ERR_BAD_ARG;
    rg::k_grow<<<scenes, rg::T, 0, static_cast<hipStream_t>(stream)>>>(starts, sizes, coord, score, neighbors, nsample, stop, slide_window, max_rounds,
                                                                     mult, cand, sim, info);
    return pdf_launch_status();
}

// The region's node list and neighbour-graph entries (see k_region_edges).  nodes_out (N) int64; eu, ev (N * nsample) int64, ew (N * nsample)
// float: scene s writes at starts[s] (* nsample); touched (N) bytes out; counts (scenes, 4) int32 out.
extern "C" int pdf_region_edges(int scenes, const int *starts, const int *sizes, const float *coord, const float *msp, const long long *neighbors,
                                int nsample, const int *mult, long long *nodes_out, long long *eu, long long *ev, float *ew, unsigned char *touched,
                                int *counts, void *stream) {
    if (scenes < 0 || nsample < 1) return PDF_ERR_BAD_ARG;
    if (scenes == 0) return PDF_OK;
    if (!starts || !sizes || !coord || !msp || !neighbors || !mult || !nodes_out || !eu || !ev || !ew || !touched || !counts) return PDF_ERR_BAD_ARG;
    rg::k_region_edges<<<scenes, rg::T, 0, static_cast<hipStream_t>(stream)>>>(starts, sizes, coord, msp, neighbors, nsample, mult, nodes_out, eu, ev, ew,
                                                                             touched, counts);
    return pdf_launch_status();
}

// The forest's entries compacted per scene (see k_tree_edges): tu, tv (N) int64, tw (N) float, tdev (scenes, 2) int32.
extern "C" int pdf_region_tree(int scenes, const int *starts, const int *sizes, int nsample, const int *counts, const unsigned char *chosen,
                               const long long *eu, const long long *ev, const float *ew, long long *tu, long long *tv, float *tw, int *tdev,
                               void *stream) {
    if (scenes < 0 || nsample < 1) return PDF_ERR_BAD_ARG;
    if (scenes == 0) return PDF_OK;
    if (!starts || !sizes || !counts || !chosen || !eu || !ev || !ew || !tu || !tv || !tw || !tdev) return PDF_ERR_BAD_ARG;
    rg::k_tree_edges<<<scenes, rg::T, 0, static_cast<hipStream_t>(stream)>>>(starts, sizes, nsample, counts, chosen, eu, ev, ew, tu, tv, tw, tdev);
    return pdf_launch_status();
}
