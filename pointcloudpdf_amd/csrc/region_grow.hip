// The PDF pseudo-label pass (pointcept/recognizers/ours/pointpdf_v1m1_base.py:188-382, helpers ours/utils.py:7-43) without a host in the
// loop.  Stage by stage, every scene of the batch per launch:
//   k_scene_stats (:190-205)   ml_norm, stop = mean - beta * std of the growth score
//   k_seed_select (:206-207)   the point of a given rank (radix select: stands in for sort + gather), seed multiplicities
//   k_grow        (:233-305)   ALL growth rounds in one workgroup per scene (k_grow_scan: its first form, fallback for huge scenes)
//   k_region_nodes[_from_list] / k_region_rows / k_region_scan (:309-335)   the region's node list and the (row, col, weight) entries of
//                              its neighbour graph -- what scipy's csr_matrix holds upstream
//   k_tree_edges, k_sort_floats   the spanning forest's entries compacted, their weights sorted (csrc/graph_prune.hip picks the forest,
//                              fits the mixture and labels the components: pdf_graph_forest_batch_dev, pdf_gmm2_weak_dev)
//   k_region_mask (:360-380)   component sizes over the touched points, z-score > 2 -> the pseudo mask
// Every size the next stage needs stays in device memory, and every reduction is done here in a fixed order (see k_scene_stats for why
// none is left to torch), so the pass can be recorded into the training step's hipGraph.
//
// Upstream grows the seed list round by round with host-side control flow: candidates = unique(neighbors[graph]) minus the members,
// ranked by 0.4 * closeness to the region's centroid + 0.6 * similarity of their score to the region's (mean of the scores between the
// region's 10 % and 60 % quantiles), the best 40 % join, until the region's mean score passes `stop` or nothing changes -- three host
// reads per round, ~35 short launches, and the device idles while the host walks through them (rounds 1-4 of this repo kept that
// shape: 6.9 ms of host-paced work per 150k-point scene).  Everything a round needs is a reduction, a k-th order statistic or a set
// update, so one 1024-thread workgroup keeps the whole loop on the device:
//   state      mult[i] = multiplicity of point i in the region list (the seed list may hold repeats -- drawn with replacement, :206 --
//              and upstream's statistics of the first round count them; a grown region is a set: mult in {0, 1})
//   per round  block reductions in double in a fixed order (length, mean score, centroid), candidate marks, the two quantiles and the
//              40 % cut by an exact 4-pass radix select over order-preserving keys (LDS histograms, integer adds), similarity in
//              upstream's operation order with contraction off, set update
// The result is the same SET as upstream's loop whenever no decision sits at a float tie (summation order differs from torch's).
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
#pragma unroll 4
        for (int w = 0; w < NW; ++w) s += lds[w * K + k];   // (fully unrolled, the 16 * K loads in flight cost 2 * 16 * K registers)
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
__device__ __forceinline__ unsigned radix_select(int n, long long k, W wt, Kf keyf, int *hist /* [256] */, unsigned *bcast /* [2] */,
                                                 long long *rank_among_equal = nullptr) {
    unsigned prefix = 0;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int b = threadIdx.x; b < 256; b += T) hist[b] = 0;
        __syncthreads();
#pragma unroll 4
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
    if (rank_among_equal) *rank_among_equal = k;   // the wanted element is the k-th (1-indexed) of those equal to the returned key
    return prefix;
}

__device__ __forceinline__ float fsub(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float fadd(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float fmul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float fdivr(float a, float b) { return __fdiv_rn(a, b); }
__device__ __forceinline__ float norm3(float dx, float dy, float dz) { return __fsqrt_rn(fadd(fadd(fmul(dx, dx), fmul(dy, dy)), fmul(dz, dz))); }

// grid = scenes.  Per scene s: points [start[s], start[s] + n); neighbour table: GLOBAL row ids (int32, -1 padded: what
// pdf_radius_neighbors_self writes), made local here.
//   mult   (N) int32  in: multiplicity of every point in the seed list; out: 1 on the region's points (or the seed multiplicities when the
//                     region never grew)
//   lists  (2 N) int32 scratch: the scene's member list | candidate list;  sim (N) float scratch (one value per candidate)
//   info   (scenes, 4) int32 out: [rounds run, grew (0 / 1), length of the region list, distinct points]
// A round touches all n points ONCE (the ascending member list + the member bitmap, by ballots); everything else walks the member list
// (a few thousand entries), the members' neighbour rows, or the candidate list they produce (round 5's first form made ~25 passes over all
// points per round, each 146 dependent iterations per lane at 150k points: 1.05 ms per round).  Membership and candidate marks are two
// bitmaps in LDS (n / 8 bytes each: test-and-set by an LDS atomicOr); the candidate list is appended in arrival order -- nothing that is
// computed from it depends on its order (minima, maxima, counts, radix selects, the tie rule by id) -- while the member list is ascending,
// so the double sums over it run in a fixed order.
__global__ __launch_bounds__(T) void k_grow(const int *__restrict__ starts, const int *__restrict__ sizes, const float *__restrict__ coord,
                                            const float *__restrict__ score, const int *__restrict__ neighbors, int nsample,
                                            const float *__restrict__ stop, int slide_window, int max_rounds, int *__restrict__ mult,
                                            int *__restrict__ lists, float *__restrict__ sim, int *__restrict__ info, int max_points) {
    extern __shared__ unsigned bits[];   // member bitmap [words] | candidate bitmap [words]
    __shared__ double dl[NW * 5];
    __shared__ float fl[NW];
    __shared__ long long ll[NW];
    __shared__ int hist[256];
    __shared__ unsigned bc[2];
    __shared__ int wcount[2][NW];
    __shared__ int ccount;
    const int s = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const long s0 = starts[s];
    const int n = sizes[s];
    if (n > max_points) {   // (the launch sized the bitmaps for max_points)
        if (threadIdx.x == 0) { info[4 * s] = -1; info[4 * s + 1] = 0; info[4 * s + 2] = 0; info[4 * s + 3] = 0; }
        return;
    }
    const int words = (n + 63) / 64 * 2;   // (whole 64-point chunks: a wave's ballot is two words)
    unsigned *mbits = bits, *cbits = bits + words;
    coord += s0 * 3; score += s0; neighbors += s0 * nsample; mult += s0; sim += s0;
    int *mlist = lists + 2 * s0, *clist = mlist + n;
    const float stop_s = stop[s];
    int rounds = 0, grew = 0;
    long long L = 0, distinct = 0;
    for (int w = t; w < words; w += T) cbits[w] = 0u;
    for (;;) {
        // ---- the ONE pass over the scene's points: ascending member list + member bitmap
        int M = 0;
        for (int base = 0, it = 0; base < n; base += T, ++it) {
            const int i = base + t;
            const bool on = i < n && mult[i] > 0;
            const unsigned long long bal = __ballot(on);
            if (lane == 0) {
                wcount[it & 1][wv] = __popcll(bal);
                if (base + 64 * wv < n) { mbits[(base >> 5) + 2 * wv] = (unsigned)bal; mbits[(base >> 5) + 2 * wv + 1] = (unsigned)(bal >> 32); }
            }
            __syncthreads();
            int pos = M, tot = 0;
            for (int w = 0; w < NW; ++w) { const int c = wcount[it & 1][w]; if (w < wv) pos += c; tot += c; }
            if (on) mlist[pos + __popcll(bal & ((1ull << lane) - 1ull))] = i;
            M += tot;
        }
        if (t == 0) ccount = 0;
        __syncthreads();
        // ---- the region list: length (with repeats), mean score, centroid
        double a[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
        for (int j = t; j < M; j += T) {
            const int i = mlist[j];
            const double dm = (double)mult[i];
            a[0] += dm; a[1] += dm * (double)score[i];
            a[2] += dm * (double)coord[3 * i]; a[3] += dm * (double)coord[3 * i + 1]; a[4] += dm * (double)coord[3 * i + 2];
        }
        block_sum<5>(a, dl);
        distinct = M;
        L = (long long)a[0];
        if (L == 0 || rounds >= max_rounds) break;
        const float g_mean = (float)(a[1] / a[0]);
        if (g_mean > stop_s && (double)L > 0.01 * (double)n && L > 50) break;
        const float cx = (float)(a[2] / a[0]), cy = (float)(a[3] / a[0]), cz = (float)(a[4] / a[0]);
        // ---- candidates: neighbours of the members that are not members (one wave per member row, lanes over the slots)
        for (int j = wv; j < M; j += NW) {
            const int *row = neighbors + (size_t)mlist[j] * nsample;
            for (int k = lane; k < nsample; k += 64) {
                const int nb = row[k] < 0 ? -1 : row[k] - (int)s0;
                if (nb >= 0 && nb < n && !((mbits[nb >> 5] >> (nb & 31)) & 1u)) {
                    const unsigned bit = 1u << (nb & 31);
                    if (!(atomicOr(&cbits[nb >> 5], bit) & bit)) clist[atomicAdd(&ccount, 1)] = nb;
                }
            }
        }
        __syncthreads();
        const int nc = ccount;
        // ---- the score the candidates are compared with: mean of the region's scores between its 10 % and 60 % quantiles
        float lo, hi;
        if (slide_window) {
            const long long k1 = (long long)((double)L * 0.1), k2 = (long long)((double)L * 0.6);   // int(len * 0.1), int(len * 0.6)
            auto wt = [&](int j) { return mult[mlist[j]]; };
            auto kf = [&](int j) { return okey(score[mlist[j]]); };
            lo = unkey(radix_select(M, k1 < 1 ? 1 : k1, wt, kf, hist, bc));
            hi = unkey(radix_select(M, k2 < 1 ? 1 : k2, wt, kf, hist, bc));
        } else {
            float mn = INFINITY, mx = -INFINITY;
            for (int j = t; j < M; j += T) { const float v = score[mlist[j]]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
            lo = block_min(mn, fl);
            hi = block_max(mx, fl);
        }
        double r2[2] = {0.0, 0.0};
        for (int j = t; j < M; j += T) {
            const int i = mlist[j];
            const float v = score[i];
            if (v >= lo && v <= hi) { const double dm = (double)mult[i]; r2[0] += dm; r2[1] += dm * (double)v; }
        }
        block_sum<2>(r2, dl);
        const float ref = (float)(r2[1] / r2[0]);
        // ---- similarity of every candidate: 0.4 * (1 - (dist - min) / (max - min + 1e-3)) + 0.6 * exp(-|score - ref|)
        float dmn = INFINITY, dmx = -INFINITY;
        for (int j = t; j < nc; j += T) {
            const int i = clist[j];
            const float d = norm3(fsub(coord[3 * i], cx), fsub(coord[3 * i + 1], cy), fsub(coord[3 * i + 2], cz));
            sim[j] = d;
            dmn = fminf(dmn, d); dmx = fmaxf(dmx, d);
        }
        const float dmin = block_min(dmn, fl), dmax = block_max(dmx, fl);
        const float den = fadd(fsub(dmax, dmin), 1e-3f);
        for (int j = t; j < nc; j += T) {
            const float ds = fsub(1.0f, fdivr(fsub(sim[j], dmin), den));
            const float cs = expf(-fabsf(fsub(score[clist[j]], ref)));
            sim[j] = fadd(fmul(0.4f, ds), fmul(0.6f, cs));
        }
        __syncthreads();
        // ---- the best 40 % join: k-th largest similarity = (nc - k + 1)-th smallest
        const long long k = (long long)((double)nc * 0.4);   // int(sim.numel() * 0.4)
        ++rounds;
        // ---- upstream: grown = unique(cat(graph, chosen)); `if grown.shape[0] == graph.shape[0]: break` compares the new SET with the old LIST
        if (distinct + k == L) break;
        if (k > 0) {
            auto one = [&](int) { return 1; };
            auto kf = [&](int j) { return okey(sim[j]); };
            long long need = 0;   // how many of the entries AT the cut value join (>= 1)
            const unsigned cut = radix_select(nc, (long long)nc - k + 1, one, kf, hist, bc, &need);
            // (rank_among_equal counts from the smallest: `need_low` of the ties lie below the wanted element; the ties that join are the
            //  others -- ties - need_low + 1; upstream's topk keeps, of equal values, the lowest ids: the tie rule below)
            long long ties_l = 0;
            for (int j = t; j < nc; j += T) ties_l += okey(sim[j]) == cut ? 1 : 0;
            const long long ties = block_sum_ll(ties_l, ll);
            const long long join_ties = ties - need + 1;
            int id_cut = 0x7fffffff;
            if (join_ties < ties) {   // (only with exact float ties across the cut) the join_ties lowest ids among the ties
                auto wtie = [&](int j) { return okey(sim[j]) == cut ? 1 : 0; };
                auto kid = [&](int j) { return (unsigned)clist[j]; };
                id_cut = (int)radix_select(nc, join_ties, wtie, kid, hist, bc);
            }
            for (int j = t; j < nc; j += T) {
                const unsigned key = okey(sim[j]);
                if (key > cut || (key == cut && clist[j] <= id_cut)) mult[clist[j]] = 1;
            }
        }
        for (int j = t; j < M; j += T) mult[mlist[j]] = 1;        // (a grown region is a set)
        for (int w = t; w < words; w += T) cbits[w] = 0u;
        grew = 1;
        __syncthreads();
    }
    if (t == 0) { info[4 * s] = rounds; info[4 * s + 1] = grew; info[4 * s + 2] = (int)L; info[4 * s + 3] = (int)distinct; }
}

// The first form of the growth kernel (round 5): every stage of a round as a pass over ALL points of the scene -- kept for scenes whose two
// bitmaps (n / 4 bytes) do not fit the LDS (more than ~600k points).  Same results as k_grow up to the summation order of the region's sums.
//   mult   (N) int32  in: multiplicity of every point in the seed list; out: 1 on the region's points (or the seed multiplicities when the
//                     region never grew)
//   cand   (N) uint8  scratch;  sim (N) float scratch
//   info   (scenes, 4) int32 out: [rounds run, grew (0 / 1), length of the region list, distinct points]
__global__ __launch_bounds__(T) void k_grow_scan(const int *__restrict__ starts, const int *__restrict__ sizes, const float *__restrict__ coord,
                                            const float *__restrict__ score, const int *__restrict__ neighbors, int nsample,
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
#pragma unroll 4
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
#pragma unroll 4
        for (int i = t; i < n; i += T) cand[i] = 0;
        __syncthreads();
#pragma unroll 4
        for (int i = t; i < n; i += T) {
            if (mult[i] > 0) {
                const int *row = neighbors + (size_t)i * nsample;
                for (int k = 0; k < nsample; ++k) {
                    const int nb = row[k] < 0 ? -1 : row[k] - (int)s0;
                    if (nb >= 0 && nb < n) cand[nb] = 1;
                }
            }
        }
        __syncthreads();
        long long nc_l = 0;
#pragma unroll 4
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
#pragma unroll 4
            for (int i = t; i < n; i += T)
                if (mult[i] > 0) { mn = fminf(mn, score[i]); mx = fmaxf(mx, score[i]); }
            lo = block_min(mn, fl);
            hi = block_max(mx, fl);
        }
        double r2[2] = {0.0, 0.0};
#pragma unroll 4
        for (int i = t; i < n; i += T) {
            const int m = mult[i];
            if (m > 0 && score[i] >= lo && score[i] <= hi) { r2[0] += (double)m; r2[1] += (double)m * (double)score[i]; }
        }
        block_sum<2>(r2, dl);
        const float ref = (float)(r2[1] / r2[0]);
        // ---- similarity of every candidate: 0.4 * (1 - (dist - min) / (max - min + 1e-3)) + 0.6 * exp(-|score - ref|)
        float dmn = INFINITY, dmx = -INFINITY;
#pragma unroll 4
        for (int i = t; i < n; i += T) {
            if (cand[i]) {
                const float d = norm3(fsub(coord[3 * i], cx), fsub(coord[3 * i + 1], cy), fsub(coord[3 * i + 2], cz));
                sim[i] = d;
                dmn = fminf(dmn, d); dmx = fmaxf(dmx, d);
            }
        }
        const float dmin = block_min(dmn, fl), dmax = block_max(dmx, fl);
        const float den = fadd(fsub(dmax, dmin), 1e-3f);
#pragma unroll 4
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
#pragma unroll 4
            for (int i = t; i < n; i += T) above_l += (cand[i] && okey(sim[i]) > cut) ? 1 : 0;
            const long long above = block_sum_ll(above_l, ll);
            const long long need = k - above;   // >= 1 entries at the cut value join
            long long ties_l = 0;
#pragma unroll 4
            for (int i = t; i < n; i += T) ties_l += (cand[i] && okey(sim[i]) == cut) ? 1 : 0;
            const long long ties = block_sum_ll(ties_l, ll);
            if (ties == need) {   // (always, but for exact float ties across the cut)
#pragma unroll 4
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
#pragma unroll 4
        for (int i = t; i < n; i += T) mult[i] = (mult[i] > 0 || (k > 0 && cand[i] == 2)) ? 1 : 0;
        grew = 1;
        __syncthreads();
    }
    if (t == 0) { info[4 * s] = rounds; info[4 * s + 1] = grew; info[4 * s + 2] = (int)L; info[4 * s + 3] = (int)distinct; }
}

// ---------------------------------------------------------------------------------------------------------------- before the growth
// Per scene (grid = scenes): what upstream computes on the logits before it grows a region (pointpdf_v1m1_base.py:190-207):
//   ml_norm = (ml - min) / (max - min + 1e-6)   (ml = the row maximum of the logits)
//   stop    = mean(score) - beta * std(score)   (unbiased std; score = msp or ml_norm)
// and the seed multiplicities are cleared (k_seed_select adds to them).  The reductions run in double in a fixed order.  (They are NOT
// left to torch: its multi-block reductions clear their semaphores with a memset node, and replays of a captured step that are queued
// back to back read garbage from them on this stack -- docs/NOTEBOOK.md, round 5.)
__global__ __launch_bounds__(T) void k_scene_stats(const int *__restrict__ starts, const int *__restrict__ sizes, const float *__restrict__ msp,
                                                   const float *__restrict__ ml, int score_is_ml, float beta, float *__restrict__ ml_norm,
                                                   float *__restrict__ stop, int *__restrict__ mult) {
    __shared__ double dl[NW];
    __shared__ float fl[NW];
    const int s = blockIdx.x, t = threadIdx.x;
    const long s0 = starts[s];
    const int n = sizes[s];
    msp += s0; ml += s0; ml_norm += s0; mult += s0;
    float mn = INFINITY, mx = -INFINITY;
#pragma unroll 4
    for (int i = t; i < n; i += T) { mn = fminf(mn, ml[i]); mx = fmaxf(mx, ml[i]); }
    mn = block_min(mn, fl);
    mx = block_max(mx, fl);
    const float den = fadd(fsub(mx, mn), 1e-6f);
    double a[1] = {0.0};
#pragma unroll 4
    for (int i = t; i < n; i += T) {
        const float v = fdivr(fsub(ml[i], mn), den);
        ml_norm[i] = v;
        mult[i] = 0;
        a[0] += (double)(score_is_ml ? v : msp[i]);
    }
    block_sum<1>(a, dl);
    const double mean = n > 0 ? a[0] / (double)n : 0.0;
    double q[1] = {0.0};
#pragma unroll 4
    for (int i = t; i < n; i += T) {
        const double d = (double)(score_is_ml ? fdivr(fsub(ml[i], mn), den) : msp[i]) - mean;
        q[0] += d * d;
    }
    block_sum<1>(q, dl);
    if (t == 0) stop[s] = fsub((float)mean, fmul(beta, (float)sqrt(q[0] / (double)(n - 1))));
}

// The seed list (:206-207: ``torch.sort(src)[1][dice]``) without the sort: workgroup (scene, j) finds the point whose src value has rank
// dice[scene, j] (0-indexed; among equal values the lower id first) by an exact radix select and adds 1 to its multiplicity.
__global__ __launch_bounds__(T) void k_seed_select(const int *__restrict__ starts, const int *__restrict__ sizes, const float *__restrict__ src,
                                                   const long long *__restrict__ dice, int num_seed, int *__restrict__ mult) {
    __shared__ int hist[256];
    __shared__ unsigned bc[2];
    __shared__ int found;
    __shared__ int wcount[NW];
    const int s = blockIdx.x / num_seed, t = threadIdx.x;
    const long s0 = starts[s];
    const int n = sizes[s];
    if (n <= 0) return;
    src += s0; mult += s0;
    long long r = dice[blockIdx.x];
    r = r < 0 ? 0 : (r > n - 1 ? n - 1 : r);
    long long need = 1;
    auto wt = [&](int) { return 1; };
    auto kf = [&](int i) { return okey(src[i]); };
    const unsigned cut = radix_select(n, r + 1, wt, kf, hist, bc, &need);
    if (t == 0) { found = 0x7fffffff; bc[0] = 0u; }
    __syncthreads();
    if (need <= 1) {   // the lowest id among the equal values
        int best = 0x7fffffff;
#pragma unroll 4
        for (int i = t; i < n; i += T)
            if (okey(src[i]) == cut) { best = i; break; }
        if (best != 0x7fffffff) atomicMin(&found, best);
        __syncthreads();
    } else {           // the need-th of them in id order: ordered walk in chunks of T
        for (int base = 0; base < n; base += T) {
            const int i = base + t;
            const bool eq = i < n && okey(src[i]) == cut;
            const unsigned long long bal = __ballot(eq);
            if ((t & 63) == 0) wcount[t >> 6] = __popcll(bal);
            __syncthreads();
            int before = (int)bc[0];
            for (int w = 0; w < (t >> 6); ++w) before += wcount[w];
            before += __popcll(bal & ((1ull << (t & 63)) - 1ull));
            if (eq && before + 1 == need) found = i;
            __syncthreads();
            if (t == 0) { int tot = 0; for (int w = 0; w < NW; ++w) tot += wcount[w]; bc[0] += (unsigned)tot; }
            __syncthreads();
            if ((long long)bc[0] >= need) break;
        }
    }
    if (t == 0 && found != 0x7fffffff) atomicAdd(&mult[found], 1);
}

// ---------------------------------------------------------------------------------------------------------------- the region's graph
// Node list = the region's points in ascending order (LOCAL ids, int64: what pdf_graph_forest takes) and the entries (u, v, w) of its
// neighbour graph in (row, col) order: v among u's neighbours with v != -1, v != u, v in the region;
// w = mult[u] * (0.4 * dist_sim + 0.6 * conf_sim) (ours/utils.py:7-43; repeated rows of a seed list are SUMMED by upstream's csr_matrix).
// Row minima / maxima of the distance run over ALL of u's neighbour slots with invalid ones (padding, u itself) counted as 0, as upstream's
// masked tensor does.  counts (scenes, 4) int32: [nodes, entries, any -1 padding among the region's rows, smallest id touched].
// Capacities: nodes_out / the entry arrays hold sizes[s] resp. sizes[s] * nsample elements per scene at offset starts[s] (* nsample).
// Four short kernels (round 5's first form walked the scene in chunks of 1024 points inside ONE workgroup and did the rows' 64-slot loops on
// the few member lanes of a chunk: 12 ms per 150k-point scene):
//   k_region_nodes   (scenes)          the ascending node list (ballot prefix per chunk), comp / lab = own id, touched = 0
//   k_region_rows    (scenes x RW)     one WAVE per region row, lanes over the neighbour slots: entry count, distance range, touched marks
//   k_region_scan    (scenes)          exclusive prefix of the rows' entry counts
//   k_region_entries (scenes x RW)     the rows again: entries written at their prefix
constexpr int RW = 64;   // workgroups per scene of the row kernels

__global__ __launch_bounds__(T) void k_region_nodes(const int *__restrict__ starts, const int *__restrict__ sizes, const int *__restrict__ mult,
                                                    long long *__restrict__ nodes_out, unsigned char *__restrict__ touched, int *__restrict__ comp,
                                                    int *__restrict__ lab, int *__restrict__ counts) {
    __shared__ int wm[2][NW];
    const int s = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const long s0 = starts[s];
    const int n = sizes[s];
    mult += s0; nodes_out += s0; touched += s0; comp += s0; lab += s0;
    int carry = 0;   // (the same value in every thread: one barrier per chunk, wave counts double-buffered)
    for (int base = 0, it = 0; base < n; base += T, ++it) {
        const int i = base + t;
        const bool on = i < n && mult[i] > 0;
        if (i < n) { touched[i] = 0; comp[i] = i; lab[i] = i; }
        const unsigned long long bal = __ballot(on);
        if (lane == 0) wm[it & 1][wv] = __popcll(bal);
        __syncthreads();
        int pos = carry, tot = 0;
        for (int w = 0; w < NW; ++w) { const int c = wm[it & 1][w]; if (w < wv) pos += c; tot += c; }
        if (on) nodes_out[pos + __popcll(bal & ((1ull << lane) - 1ull))] = i;
        carry += tot;
    }
    if (t == 0) { counts[4 * s] = carry; counts[4 * s + 1] = 0; counts[4 * s + 2] = 0; counts[4 * s + 3] = 0x7fffffff; }
}

// The same from the ascending member list pdf_region_grow left behind (lists: 2 N ints, scene s at 2 * start[s]; info[4 s + 3] = its
// length): a plain copy, RW workgroups per scene instead of a one-workgroup compaction (137 -> ~10 us per 150k-point scene).
__global__ __launch_bounds__(T) void k_region_nodes_from_list(const int *__restrict__ starts, const int *__restrict__ sizes,
                                                              const int *__restrict__ lists, const int *__restrict__ info,
                                                              long long *__restrict__ nodes_out, unsigned char *__restrict__ touched,
                                                              int *__restrict__ comp, int *__restrict__ lab, int *__restrict__ counts) {
    const int s = blockIdx.x / RW, part = blockIdx.x % RW;
    const long s0 = starts[s];
    const int n = sizes[s], M = min(max(info[4 * s + 3], 0), n);
    const int *mlist = lists + 2 * s0;
    for (int i = part * T + threadIdx.x; i < n; i += RW * T) {
        touched[s0 + i] = 0; comp[s0 + i] = i; lab[s0 + i] = i;
        if (i < M) nodes_out[s0 + i] = mlist[i];
    }
    if (part == 0 && threadIdx.x == 0) { counts[4 * s] = M; counts[4 * s + 1] = 0; counts[4 * s + 2] = 0; counts[4 * s + 3] = 0x7fffffff; }
}

// WRITE = false: rowcnt[r] = entries of row r, rowrange[2 r] = (dmin, dmax), touched marks, counts[2] |= padding seen, counts[3] = min id touched
// WRITE = true : the entries of row r at rowcnt[r] (by then the exclusive prefix)
template <bool WRITE>
__global__ __launch_bounds__(T) void k_region_rows(const int *__restrict__ starts, const int *__restrict__ sizes, const float *__restrict__ coord,
                                                   const float *__restrict__ msp, const int *__restrict__ neighbors, int nsample,
                                                   const int *__restrict__ mult, const long long *__restrict__ nodes, int *__restrict__ rowcnt,
                                                   float *__restrict__ rowrange, long long *__restrict__ eu, long long *__restrict__ ev,
                                                   float *__restrict__ ew, unsigned char *__restrict__ touched, int *__restrict__ counts) {
    const int s = blockIdx.x / RW, t = threadIdx.x, lane = t & 63;
    const long s0 = starts[s];
    const int n = sizes[s], M = counts[4 * s];
    coord += s0 * 3; msp += s0; neighbors += s0 * nsample; mult += s0; nodes += s0; rowcnt += s0; rowrange += 2 * s0; touched += s0;
    eu += s0 * nsample; ev += s0 * nsample; ew += s0 * nsample;
    bool pad = false;
    int first = 0x7fffffff;
    for (int r = (blockIdx.x % RW) * NW + (t >> 6); r < M; r += RW * NW) {
        const int i = (int)nodes[r];
        const int *row = neighbors + (size_t)i * nsample;
        const float px = coord[3 * i], py = coord[3 * i + 1], pz = coord[3 * i + 2];
        if (!WRITE) {
            float dmin = INFINITY, dmax = -INFINITY;
            int cnt = 0;
            for (int k0 = 0; k0 < nsample; k0 += 64) {
                const int k = k0 + lane;
                if (k < nsample) {
                    const int nb = row[k] < 0 ? -1 : row[k] - (int)s0;
                    const bool valid = nb >= 0 && nb != i;
                    const int j = nb >= 0 ? nb : n - 1;   // (upstream indexes coord[-1]: the last point; its distance is masked to 0 anyway)
                    const float d = valid ? norm3(fsub(coord[3 * j], px), fsub(coord[3 * j + 1], py), fsub(coord[3 * j + 2], pz)) : 0.f;
                    dmin = fminf(dmin, d); dmax = fmaxf(dmax, d);
                    if (nb < 0) pad = true;
                    else {
                        touched[nb] = 1;
                        first = min(first, nb);
                        if (valid && nb < n && mult[nb] > 0) ++cnt;
                    }
                }
            }
            for (int o = 32; o > 0; o >>= 1) {
                dmin = fminf(dmin, __shfl_xor(dmin, o, 64)); dmax = fmaxf(dmax, __shfl_xor(dmax, o, 64)); cnt += __shfl_xor(cnt, o, 64);
            }
            if (lane == 0) { rowcnt[r] = cnt; rowrange[2 * r] = dmin; rowrange[2 * r + 1] = dmax; touched[i] = 1; first = min(first, i); }
        } else {
            const float dmin = rowrange[2 * r], den = fadd(fsub(rowrange[2 * r + 1], dmin), 1e-3f), si = msp[i];
            const int m = mult[i];
            int e = rowcnt[r];
            for (int k0 = 0; k0 < nsample; k0 += 64) {
                const int k = k0 + lane;
                const int nb = k < nsample ? (row[k] < 0 ? -1 : row[k] - (int)s0) : -1;
                const bool in = nb >= 0 && nb != i && nb < n && mult[nb] > 0;
                const unsigned long long bal = __ballot(in);
                if (in) {
                    const float d = norm3(fsub(coord[3 * nb], px), fsub(coord[3 * nb + 1], py), fsub(coord[3 * nb + 2], pz));
                    const float ds = fsub(1.0f, fdivr(fsub(d, dmin), den));
                    const float cs = expf(-fabsf(fsub(msp[nb], si)));
                    const float w = fadd(fmul(0.4f, ds), fmul(0.6f, cs));
                    const int at = e + __popcll(bal & ((1ull << lane) - 1ull));
                    eu[at] = i; ev[at] = nb; ew[at] = m == 1 ? w : fmul((float)m, w);
                }
                e += __popcll(bal);
            }
        }
    }
    if (!WRITE) {
        if (__ballot(pad)) { if (lane == 0) atomicOr(&counts[4 * s + 2], 1); }
        for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o, 64));
        if (lane == 0 && first != 0x7fffffff) atomicMin(&counts[4 * s + 3], first);
    }
}

__global__ __launch_bounds__(T) void k_region_scan(const int *__restrict__ starts, int *__restrict__ rowcnt, int *__restrict__ counts) {
    __shared__ int wsum[NW];
    __shared__ int carry;
    const int s = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int M = counts[4 * s];
    rowcnt += (long)starts[s];
    if (t == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < M; base += T) {
        const int r = base + t;
        const int c = r < M ? rowcnt[r] : 0;
        int p = c;
        for (int o = 1; o < 64; o <<= 1) {
            const int a = __shfl_up(p, o, 64);
            if (lane >= o) p += a;
        }
        if (lane == 63) wsum[wv] = p;
        __syncthreads();
        int off = carry;
        for (int w = 0; w < wv; ++w) off += wsum[w];
        if (r < M) rowcnt[r] = off + p - c;
        __syncthreads();
        if (t == T - 1) carry = off + p;
        __syncthreads();
    }
    if (t == 0) counts[4 * s + 1] = carry;
}

// The chosen entries of the spanning forest, in entry order: tu, tv (ids), tw (weights), tdev (scenes, 2) = [nodes, tree edges] for the sort,
// the mixture fit and the second graph kernel (nothing is written beyond the tree: a one-workgroup fill of the scene's slices was half of
// this kernel's time).
__global__ __launch_bounds__(T) void k_tree_edges(const int *__restrict__ starts, const int *__restrict__ sizes, int nsample,
                                                  const int *__restrict__ counts, const unsigned char *__restrict__ chosen,
                                                  const long long *__restrict__ eu, const long long *__restrict__ ev, const float *__restrict__ ew,
                                                  long long *__restrict__ tu, long long *__restrict__ tv, float *__restrict__ tw, int *__restrict__ tdev) {
    __shared__ int wcnt[2][NW];
    const int s = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const long s0 = starts[s];
    const int n = sizes[s], E = counts[4 * s + 1];
    chosen += s0 * nsample; eu += s0 * nsample; ev += s0 * nsample; ew += s0 * nsample;
    tu += s0; tv += s0; tw += s0;
    int carry = 0;   // (the same value in every thread: one barrier per chunk, wave counts double-buffered)
    // four consecutive entries per lane and trip (the flags of a scene start at a multiple of nsample: 4-byte loads when nsample % 4 == 0):
    // the walk is a chain of dependent flag loads, one per chunk -- 85 chunks of 1,024 entries at 87k entries, 22 of 4,096
    const bool vec = (nsample & 3) == 0;
    const int step = vec ? 4 * T : T;
    for (int base = 0, it = 0; base < E; base += step, ++it) {
        const int e0 = vec ? base + 4 * t : base + t;
        unsigned f = 0;   // flags of the lane's entries, one bit each
        if (vec) {
            if (e0 < E) {
                const unsigned v = *reinterpret_cast<const unsigned *>(chosen + e0);
#pragma unroll
                for (int q = 0; q < 4; ++q) f |= (e0 + q < E && ((v >> (8 * q)) & 0xffu)) ? 1u << q : 0u;
            }
        } else {
            f = (e0 < E && chosen[e0] != 0) ? 1u : 0u;
        }
        const int c = __popc(f);
        int p = c;   // inclusive prefix of the lane counts inside the wave
        for (int o = 1; o < 64; o <<= 1) {
            const int a = __shfl_up(p, o, 64);
            if (lane >= o) p += a;
        }
        if (lane == 63) wcnt[it & 1][wv] = p;
        __syncthreads();
        int pos = carry + p - c, tot = 0;
        for (int w = 0; w < NW; ++w) { const int cw = wcnt[it & 1][w]; if (w < wv) pos += cw; tot += cw; }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if ((f >> q) & 1u) {
                if (pos < n) { tu[pos] = eu[e0 + q]; tv[pos] = ev[e0 + q]; tw[pos] = ew[e0 + q]; }
                ++pos;
            }
        carry += tot;
    }
    const int m = min(carry, n);   // (entries beyond m are left as they are: every consumer reads tdev[2 s + 1])
    if (t == 0) { tdev[2 * s] = counts[4 * s]; tdev[2 * s + 1] = m; }
}

// Ascending sort of the first m = tdev[2 s + 1] floats of every scene's slice (the spanning tree's weights: the mixture fit below takes
// its quartiles from sorted data and sums in sorted order, as the host form does after torch.sort).  One workgroup per scene, LSD radix
// sort of the order-preserving keys, 8 passes of 4 bits; the rank of a key inside a chunk of T comes from 16 ballots.  out / tmp: N words.
__global__ __launch_bounds__(T) void k_sort_floats(const int *__restrict__ starts, const int *__restrict__ sizes, const int *__restrict__ tdev,
                                                   const float *__restrict__ x, float *__restrict__ out, unsigned *__restrict__ tmp) {
    __shared__ int hist[16];
    __shared__ int base[16];
    __shared__ int wtot[NW * 16];
    const int s = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const long s0 = starts[s];
    const int m = min(tdev[2 * s + 1], sizes[s]);
    x += s0; tmp += s0;
    unsigned *a = reinterpret_cast<unsigned *>(out + s0);
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 4 * pass;
        const unsigned *src = (pass & 1) ? tmp : a;          // pass 0 reads x itself
        unsigned *dst = (pass & 1) ? a : tmp;
        if (t < 16) hist[t] = 0;
        __syncthreads();
        for (int i = t; i < m; i += T) {
            const unsigned key = pass == 0 ? okey(x[i]) : src[i];
            atomicAdd(&hist[(key >> shift) & 15u], 1);
        }
        __syncthreads();
        if (t == 0) {
            int acc = 0;
            for (int b = 0; b < 16; ++b) { base[b] = acc; acc += hist[b]; }
        }
        __syncthreads();
        for (int c0 = 0; c0 < m; c0 += T) {
            const int i = c0 + t;
            const bool on = i < m;
            const unsigned key = on ? (pass == 0 ? okey(x[i]) : src[i]) : 0u;
            const int d = on ? (int)((key >> shift) & 15u) : 16;
            unsigned long long mine = 0ull;
            int tot = 0;
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                const unsigned long long bal = __ballot(d == b);
                if (d == b) mine = bal;
                if (lane == b) tot = __popcll(bal);
            }
            if (lane < 16) wtot[wv * 16 + lane] = tot;
            __syncthreads();
            if (on) {
                int pos = base[d];
                for (int w = 0; w < wv; ++w) pos += wtot[w * 16 + d];
                pos += __popcll(mine & ((1ull << lane) - 1ull));
                dst[pos] = key;
            }
            __syncthreads();
            if (t < 16) { int add = 0; for (int w = 0; w < NW; ++w) add += wtot[w * 16 + t]; base[t] += add; }
            __syncthreads();
        }
    }
    for (int i = t; i < m; i += T) out[s0 + i] = unkey(a[i]);   // (8 passes: the last one wrote `a`)
}

// The pseudo mask of every scene from its second component labelling (pointpdf_v1m1_base.py:360-380): size of every component counted
// over the TOUCHED points (region rows and their neighbours, minus the first entry upstream's unique(...)[1:] drops), components whose
// size lies more than 2 population standard deviations above the mean -> mask.  cnt: N ints of workspace.
__global__ __launch_bounds__(T) void k_region_mask(const int *__restrict__ starts, const int *__restrict__ sizes, const int *__restrict__ lab,
                                                   const unsigned char *__restrict__ touched, const int *__restrict__ counts, int *__restrict__ cnt,
                                                   unsigned char *__restrict__ mask) {
    __shared__ double dl[NW * 2];
    const int s = blockIdx.x, t = threadIdx.x;
    const long s0 = starts[s];
    const int n = sizes[s];
    lab += s0; touched += s0; cnt += s0; mask += s0;
    const int has_pad = counts[4 * s + 2];
    const int first = min(counts[4 * s + 3], n - 1);
#pragma unroll 4
    for (int i = t; i < n; i += T) cnt[i] = 0;
    __syncthreads();
#pragma unroll 4
    for (int i = t; i < n; i += T) {
        // touched = unique(cat([node, node_nn]))[1:]: the first entry is the -1 padding -- or, when no row of the region is padded, the
        // smallest id touched (dropped all the same upstream)
        if (touched[i] && !(i == first && !has_pad)) atomicAdd(&cnt[lab[i]], 1);
    }
    __threadfence();
    __syncthreads();
    double a[2] = {0.0, 0.0};
#pragma unroll 4
    for (int i = t; i < n; i += T) {
        const int c = __hip_atomic_load(&cnt[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (c > 0) { a[0] += 1.0; a[1] += (double)c; }
    }
    block_sum<2>(a, dl);
    const double k = a[0], mean = a[1] / k;
    double q[1] = {0.0};
#pragma unroll 4
    for (int i = t; i < n; i += T) {
        const int c = __hip_atomic_load(&cnt[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (c > 0) q[0] += ((double)c - mean) * ((double)c - mean);
    }
    block_sum<1>(q, dl);
    const double sd = sqrt(q[0] / k);
#pragma unroll 4
    for (int i = t; i < n; i += T) {
        const int c = __hip_atomic_load(&cnt[lab[i]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        mask[i] = (c > 0 && ((double)c - mean) / sd > 2.0) ? 1 : 0;   // z_score_mask_np(area="right", score=2.0)
    }
}

}  // namespace rg

// Before the growth (see k_scene_stats): ml_norm (N) out, stop (scenes) out, mult (N) cleared.  msp / ml (N): the row maxima of the softmax /
// of the logits; score_is_ml: the growth condition runs on ml_norm instead of msp.
extern "C" int pdf_region_stats(int scenes, const int *starts, const int *sizes, const float *msp, const float *ml, int score_is_ml, float beta,
                                float *ml_norm, float *stop, int *mult, void *stream) {
    if (scenes < 0) return PDF_ERR_BAD_ARG;
    if (scenes == 0) return PDF_OK;
    if (!starts || !sizes || !msp || !ml || !ml_norm || !stop || !mult) return PDF_ERR_BAD_ARG;
    rg::k_scene_stats<<<scenes, rg::T, 0, static_cast<hipStream_t>(stream)>>>(starts, sizes, msp, ml, score_is_ml, beta, ml_norm, stop, mult);
    return pdf_launch_status();
}

// The seed list: mult[p] += 1 for the point p of scene s whose src value has rank dice[s * num_seed + j] (see k_seed_select).
extern "C" int pdf_region_seeds(int scenes, const int *starts, const int *sizes, const float *src, const long long *dice, int num_seed, int *mult,
                                void *stream) {
    if (scenes < 0 || num_seed < 0) return PDF_ERR_BAD_ARG;
    if (scenes == 0 || num_seed == 0) return PDF_OK;
    if (!starts || !sizes || !src || !dice || !mult) return PDF_ERR_BAD_ARG;
    rg::k_seed_select<<<scenes * num_seed, rg::T, 0, static_cast<hipStream_t>(stream)>>>(starts, sizes, src, dice, num_seed, mult);
    return pdf_launch_status();
}

// Dynamic LDS the growth kernel may ask for: 150 KB (+ ~2 KB static; the CU has 160 KB) once the runtime accepted the raised limit, else the
// 64 KB default minus the static part.
static size_t grow_lds_limit() {
    constexpr size_t LDS_MAX = 150 * 1024;
    static PdfLdsLimit site;   // per device (pdfops_common.h)
    return pdf_lds_limit_raised(site, reinterpret_cast<const void *>(&rg::k_grow), (int)LDS_MAX) ? LDS_MAX : (size_t)60 * 1024;
}

// Region growing of every scene of a batch, all rounds on the device.  starts / sizes (scenes) int32: the scenes' point ranges; neighbors
// (N, nsample) int32 GLOBAL row ids (-1 padded: pdf_radius_neighbors_self's table); stop (scenes) float; mult (N) int32 in / out; lists
// (2 N) ints, sim (N) floats: scratch; info (scenes, 4) int32 out [rounds, grew, list length, distinct points].  max_points: the largest
// scene of the batch (host value: sizes the LDS bitmaps; larger scenes than it would corrupt the workgroup's LDS -- checked on the device:
// such a scene reports rounds = -1 and keeps its seeds).
extern "C" int pdf_region_grow(int scenes, const int *starts, const int *sizes, int max_points, const float *coord, const float *score,
                               const int *neighbors, int nsample, const float *stop, int slide_window, int max_rounds, int *mult, int *lists,
                               float *sim, int *info, void *stream) {
    if (scenes < 0 || nsample < 1 || max_rounds < 0 || max_points < 0) return PDF_ERR_BAD_ARG;
    if (scenes == 0) return PDF_OK;
    if (!starts || !sizes || !coord || !score || !neighbors || !stop || !mult || !lists || !sim || !info) return PDF_ERR_BAD_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t lds = (size_t)((max_points + 63) / 64 * 2) * 2 * sizeof(unsigned);
    if (lds <= grow_lds_limit())
        rg::k_grow<<<scenes, rg::T, lds, st>>>(starts, sizes, coord, score, neighbors, nsample, stop, slide_window, max_rounds, mult, lists, sim, info,
                                              max_points);
    else
        rg::k_grow_scan<<<scenes, rg::T, 0, st>>>(starts, sizes, coord, score, neighbors, nsample, stop, slide_window, max_rounds, mult,
                                                 reinterpret_cast<unsigned char *>(lists), sim, info);
    return pdf_launch_status();
}

// The region's node list and neighbour-graph entries (see k_region_rows).  nodes_out (N) int64; eu, ev (N * nsample) int64, ew (N * nsample)
// float: scene s writes at starts[s] (* nsample); touched (N) bytes out; comp, lab (N) int32 out: every point's own LOCAL id (what the two
// pdf_graph_forest_dev calls start from); counts (scenes, 4) int32 out; rows_ws: 3 N words of workspace.  lists / grow_info: pdf_region_grow's
// `lists` and `info` of the SAME batch when it ran its LDS form (largest scene <= pdf_region_grow_list_points()), else both NULL.
extern "C" int pdf_region_edges(int scenes, const int *starts, const int *sizes, const float *coord, const float *msp, const int *neighbors,
                                int nsample, const int *mult, const int *lists, const int *grow_info, long long *nodes_out, long long *eu,
                                long long *ev, float *ew, unsigned char *touched, int *comp, int *lab, int *counts, void *rows_ws, long n_total,
                                void *stream) {
    if (scenes < 0 || nsample < 1 || n_total < 0) return PDF_ERR_BAD_ARG;
    if (scenes == 0) return PDF_OK;
    if (!starts || !sizes || !coord || !msp || !neighbors || !mult || !nodes_out || !eu || !ev || !ew || !touched || !comp || !lab || !counts || !rows_ws)
        return PDF_ERR_BAD_ARG;
    if ((lists == nullptr) != (grow_info == nullptr)) return PDF_ERR_BAD_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int *rowcnt = static_cast<int *>(rows_ws);
    float *rowrange = reinterpret_cast<float *>(rowcnt + n_total);
    if (lists)   // (the member list pdf_region_grow's LDS form left behind; its scan form leaves none: the caller passes NULL then)
        rg::k_region_nodes_from_list<<<scenes * rg::RW, rg::T, 0, st>>>(starts, sizes, lists, grow_info, nodes_out, touched, comp, lab, counts);
    else
        rg::k_region_nodes<<<scenes, rg::T, 0, st>>>(starts, sizes, mult, nodes_out, touched, comp, lab, counts);
    rg::k_region_rows<false><<<scenes * rg::RW, rg::T, 0, st>>>(starts, sizes, coord, msp, neighbors, nsample, mult, nodes_out, rowcnt, rowrange, eu, ev, ew,
                                                               touched, counts);
    rg::k_region_scan<<<scenes, rg::T, 0, st>>>(starts, rowcnt, counts);
    rg::k_region_rows<true><<<scenes * rg::RW, rg::T, 0, st>>>(starts, sizes, coord, msp, neighbors, nsample, mult, nodes_out, rowcnt, rowrange, eu, ev, ew,
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

// The first tdev[2 s + 1] floats of every scene's slice of x, ascending, into the same slice of out (see k_sort_floats); tmp: N words.
extern "C" int pdf_sort_floats_dev(int scenes, const int *starts, const int *sizes, const int *tdev, const float *x, float *out, void *tmp,
                                   void *stream) {
    if (scenes < 0) return PDF_ERR_BAD_ARG;
    if (scenes == 0) return PDF_OK;
    if (!starts || !sizes || !tdev || !x || !out || !tmp) return PDF_ERR_BAD_ARG;
    rg::k_sort_floats<<<scenes, rg::T, 0, static_cast<hipStream_t>(stream)>>>(starts, sizes, tdev, x, out, static_cast<unsigned *>(tmp));
    return pdf_launch_status();
}

// The pseudo mask (N bytes, 0 / 1) from the second labelling (see k_region_mask); cnt: N ints of workspace.
extern "C" int pdf_region_mask(int scenes, const int *starts, const int *sizes, const int *lab, const unsigned char *touched, const int *counts,
                               int *cnt, unsigned char *mask, void *stream) {
    if (scenes < 0) return PDF_ERR_BAD_ARG;
    if (scenes == 0) return PDF_OK;
    if (!starts || !sizes || !lab || !touched || !counts || !cnt || !mask) return PDF_ERR_BAD_ARG;
    rg::k_region_mask<<<scenes, rg::T, 0, static_cast<hipStream_t>(stream)>>>(starts, sizes, lab, touched, counts, cnt, mask);
    return pdf_launch_status();
}

// The largest scene (points) for which pdf_region_grow runs its LDS form -- the form that leaves the ascending member list in `lists`.
extern "C" long pdf_region_grow_list_points(void) { return (long)(grow_lds_limit() / (2 * 2 * sizeof(unsigned))) * 64 - 64; }
