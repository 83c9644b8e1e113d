// Bucketed exact farthest point sampling for gfx950.
//
// Same results, bit for bit, as the plain kernel / the reference (sampling_cuda_kernel.cu:14-129): sample j is the
// arg-max of tmp[k] = min_i d(k, sample_i) with the reference's tie rule, encoded as the packed key of sampling.hip.
// What changes is the work per iteration.  The reference touches every point of the scene for every sample
// (25,000 x 100,000 point updates at level 1).  Here the points of a scene are grouped into spatial buckets of 64
// (counting sort by a 15-bit Morton cell of the scene's bounding box), each bucket keeps
//      bounding box | best packed key (max tmp + tie key) | coordinates of that best point
// in LDS, and a new sample s only visits buckets with  d_bb(s, bucket) < max tmp(bucket)  where d_bb is the
// box distance evaluated with the same fp32 operation sequence as d.  Because IEEE rounding is monotonic,
// d_fp32(k, s) >= d_bb_fp32 for every point k of the bucket, so a skipped bucket provably has min(d, tmp) == tmp for
// all its points: pruning is EXACT, no epsilon.  16 buckets form a super-bucket with the same record, checked first.
//
// Execution (three kernels, chosen by the size of the largest scene; all give the same indices):
//   k_fps<K>      one single-wave workgroup per scene, one sample per iteration (small scenes: extra waves only add barriers):
//                 super check (<= 3 per lane) -> bucket check (lanes 0..15) -> 64-lane update of each surviving bucket (one point per
//                 lane, coalesced 1 KiB loads from the L2-resident sorted copy) -> u64 wave max-reductions;
//   k_fps_multi   several samples per ROUND, exactly (accepted prefix known before any update, see the kernel header);
//   k_fps_mw<NW>  the same with one workgroup of NW = 8 / 16 waves per scene: wave 0 selects the round's centres, wave u owns centre u,
//                 buckets claimed once through LDS and dealt round-robin to the waves (scenes >= 3k / 16k points).
// A typical sample touches 1-4 buckets instead of 1,563.  Roofline: latency-bound by design; algorithmic HBM bytes 12N + 4M'
// (SURVEY.md 8d).
#include "pdfops_common.h"
#include <stdlib.h>

extern "C" int pdf_fps_reference_block_log2(int n);

namespace {

constexpr int BSZ = 64;            // points per bucket (= one wave, one point per lane)
constexpr int SUP = 16;            // buckets per super-bucket
constexpr int NB_MAX = 3072;       // buckets per scene held in LDS (=> scenes up to 196,608 points)
constexpr int NS_MAX = NB_MAX / SUP;
constexpr int CELLS = 32768;       // 15-bit Morton cells per scene
constexpr int REL_BITS = 22;
constexpr unsigned REL_MASK = (1u << REL_BITS) - 1u;
constexpr int PB = 256;            // block size of the preparation kernels
constexpr int MSTRIDE = 16;        // dwords of metadata per bucket (box 6, best key 2, best xyz 3, second-best key 2)
constexpr int FPS_MULTI_MIN = 768;  // smallest largest-scene size for the several-samples-per-round kernels (3,072 before the rounds could take their
                                    // candidates bucket by bucket: a 1,562-point scene has 2 super-buckets but 25 buckets)
constexpr int TWO_MAX_SUPERS = 64;  // k_fps_mw: largest scene (in super-buckets) that takes its candidates in two levels
constexpr int BKC_MAX = 256;       // k_fps_mw: scenes with at most this many buckets take their round's candidates bucket by bucket

struct Layout {  // byte offsets into the caller's workspace
    size_t stats, bbox, hist, cursor, cell, pts, kb, meta, total;
    int npad, nbk;
};

__host__ __device__ inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

__host__ __device__ inline Layout make_layout(int b, int n_total) {
    Layout L;
    L.npad = n_total + BSZ * b;
    L.nbk = L.npad / BSZ + b;
    size_t o = 0;
    L.stats = o;  o = align256(o + (size_t)b * 4 * 4);  // per scene: bucket updates, super visits, samples, -
    L.bbox = o;   o = align256(o + (size_t)b * 8 * 4);
    L.hist = o;   o = align256(o + (size_t)b * CELLS * 4);
    L.cursor = o; o = align256(o + (size_t)b * CELLS * 4);
    L.cell = o;   o = align256(o + (size_t)n_total * 4);
    L.pts = o;    o = align256(o + (size_t)L.npad * 16);
    L.kb = o;     o = align256(o + (size_t)L.npad * 4);
    L.meta = o;   o = align256(o + (size_t)L.nbk * MSTRIDE * 4);
    L.total = o;
    return L;
}

struct Scene {
    int start_n, n, start_m, m, pbase, bbase, nb;
};

// Scene table entry from the cumulative offsets (b is small: linear scan).
__device__ inline Scene scene_of(int s, const int *__restrict__ offset, const int *__restrict__ new_offset) {
    Scene sc;
    int pbase = 0, bbase = 0, prev = 0;
    for (int i = 0; i < s; ++i) {
        const int e = offset[i];
        const int nb = (e - prev + BSZ - 1) / BSZ;
        pbase += nb * BSZ;
        bbase += nb;
        prev = e;
    }
    sc.start_n = prev;
    sc.n = offset[s] - prev;
    sc.start_m = s == 0 ? 0 : new_offset[s - 1];
    sc.m = new_offset[s] - sc.start_m;
    sc.pbase = pbase;
    sc.bbase = bbase;
    sc.nb = (sc.n + BSZ - 1) / BSZ;
    return sc;
}

__device__ inline unsigned keybits(int rel, int bs_ref_log2) {
    const unsigned slot = (unsigned)rel & ((1u << bs_ref_log2) - 1u);
    const unsigned rev = bs_ref_log2 ? (__brev(slot) >> (32 - bs_ref_log2)) : 0u;
    return ((~rev & 0x3ffu) << REL_BITS) | (~(unsigned)rel & REL_MASK);
}

__device__ inline float dist_as_written(float x2, float y2, float z2, float x1, float y1, float z1) {
    return pdf_sqdist3(x2 - x1, y2 - y1, z2 - z1);
}

// ---------------------------------------------------------------- preparation kernels
__global__ __launch_bounds__(PB) void k_bbox(const float *__restrict__ xyz, const int *__restrict__ offset, float *__restrict__ bbox) {
    __shared__ float red[6][PB / 64];
    const int s = blockIdx.x;
    const int start = s == 0 ? 0 : offset[s - 1], end = offset[s];
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = start + threadIdx.x; i < end; i += PB)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = xyz[3 * (size_t)i + a];
            lo[a] = fminf(lo[a], v);
            hi[a] = fmaxf(hi[a], v);
        }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            red[a][threadIdx.x >> 6] = lo[a];
            red[3 + a][threadIdx.x >> 6] = hi[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        float l = red[threadIdx.x][0], h = red[3 + threadIdx.x][0];
        for (int w = 1; w < PB / 64; ++w) {
            l = fminf(l, red[threadIdx.x][w]);
            h = fmaxf(h, red[3 + threadIdx.x][w]);
        }
        bbox[s * 8 + threadIdx.x] = l;
        const float ext = h - l;
        bbox[s * 8 + 3 + threadIdx.x] = ext > 0.f ? 32.0f / (ext * 1.0001f) : 0.f;  // cells per unit length
    }
}

__device__ inline unsigned spread5(unsigned v) {  // 5 bits -> every third bit
    v &= 31u;
    v = (v | (v << 8)) & 0x100fu;
    v = (v | (v << 4)) & 0x10c3u;
    v = (v | (v << 2)) & 0x1249u;
    return v;
}

__device__ inline int scene_of_point(int i, const int *__restrict__ offset, int b) {
    int s = 0;
    while (s < b - 1 && i >= offset[s]) ++s;
    return s;
}

__global__ __launch_bounds__(PB) void k_hist(int n_total, int b, const float *__restrict__ xyz, const int *__restrict__ offset,
                                             const float *__restrict__ bbox, unsigned *__restrict__ hist, unsigned *__restrict__ cell) {
    const int i = blockIdx.x * PB + threadIdx.x;
    if (i >= n_total) return;
    const int s = scene_of_point(i, offset, b);
    unsigned q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float t = (xyz[3 * (size_t)i + a] - bbox[s * 8 + a]) * bbox[s * 8 + 3 + a];
        int v = (int)t;
        q[a] = (unsigned)(v < 0 ? 0 : v > 31 ? 31 : v);
    }
    const unsigned c = spread5(q[0]) | (spread5(q[1]) << 1) | (spread5(q[2]) << 2);
    cell[i] = c;
    atomicAdd(&hist[(size_t)s * CELLS + c], 1u);
}

// exclusive scan of the 32768 cell counts of one scene (1024 threads x 32 cells), in place
__global__ __launch_bounds__(1024) void k_scan(unsigned *__restrict__ hist) {
    __shared__ unsigned wsum[16];
    unsigned *h = hist + (size_t)blockIdx.x * CELLS;
    const int t = threadIdx.x;
    unsigned v[32], sum = 0;
#pragma unroll
    for (int u = 0; u < 32; ++u) {
        v[u] = h[t * 32 + u];
        sum += v[u];
    }
    unsigned inc = sum;  // inclusive scan of per-thread sums across the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned up = __shfl_up(inc, o, 64);
        if ((t & 63) >= o) inc += up;
    }
    if ((t & 63) == 63) wsum[t >> 6] = inc;
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < (t >> 6); ++w) base += wsum[w];
    unsigned run = base + inc - sum;
#pragma unroll
    for (int u = 0; u < 32; ++u) {
        h[t * 32 + u] = run;
        run += v[u];
    }
}

__global__ __launch_bounds__(PB) void k_scatter(int n_total, int b, const float *__restrict__ xyz, const int *__restrict__ offset,
                                                const int *__restrict__ new_offset, const unsigned *__restrict__ start,
                                                unsigned *__restrict__ cursor, const unsigned *__restrict__ cell,
                                                float4 *__restrict__ pts, unsigned *__restrict__ kb, int bs_ref_log2) {
    const int i = blockIdx.x * PB + threadIdx.x;
    if (i >= n_total) return;
    const int s = scene_of_point(i, offset, b);
    const Scene sc = scene_of(s, offset, new_offset);
    const unsigned c = cell[i];
    const unsigned r = start[(size_t)s * CELLS + c] + atomicAdd(&cursor[(size_t)s * CELLS + c], 1u);
    const float x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
    // tmp after the first sample (= first point of the scene); the reference starts from tmp = 1e10 (sampling.py:19)
    const float x1 = xyz[3 * (size_t)sc.start_n], y1 = xyz[3 * (size_t)sc.start_n + 1], z1 = xyz[3 * (size_t)sc.start_n + 2];
    const float t = fminf(dist_as_written(x, y, z, x1, y1, z1), 1e10f);
    const size_t dst = (size_t)sc.pbase + r;
    pts[dst] = make_float4(x, y, z, t);
    kb[dst] = keybits(i - sc.start_n, bs_ref_log2);
}

// pad the tail of every scene's last bucket with never-winning points
__global__ void k_pad(int b, const int *__restrict__ offset, const int *__restrict__ new_offset, float4 *__restrict__ pts,
                      unsigned *__restrict__ kb) {
    const int s = blockIdx.x;
    const Scene sc = scene_of(s, offset, new_offset);
    const int r = sc.n + threadIdx.x;
    if (r < sc.nb * BSZ) {
        pts[(size_t)sc.pbase + r] = make_float4(0.f, 0.f, 0.f, -1.f);
        kb[(size_t)sc.pbase + r] = 0u;
    }
}

// one wave per bucket: bounding box, best packed key, coordinates of the best point, second-best key -> MSTRIDE dwords
__global__ __launch_bounds__(64) void k_meta(const float4 *__restrict__ pts, const unsigned *__restrict__ kb, float *__restrict__ meta) {
    const int bk = blockIdx.x;
    const int lane = threadIdx.x;
    const float4 p = pts[(size_t)bk * BSZ + lane];
    const bool valid = p.w >= 0.f;
    float lo[3] = {valid ? p.x : 3.0e38f, valid ? p.y : 3.0e38f, valid ? p.z : 3.0e38f};
    float hi[3] = {valid ? p.x : -3.0e38f, valid ? p.y : -3.0e38f, valid ? p.z : -3.0e38f};
    unsigned long long key = valid ? (((unsigned long long)pdf_f32_ordered(p.w) << 32) | kb[(size_t)bk * BSZ + lane]) : 0ull;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64));
        }
    }
    const unsigned long long best = pdf_wave_max_u64(key);
    const unsigned long long mask = __ballot(key == best && valid);
    const int win = mask ? __ffsll((long long)mask) - 1 : 0;
    const float bx = __shfl(p.x, win, 64), by = __shfl(p.y, win, 64), bz = __shfl(p.z, win, 64);
    const unsigned long long second = pdf_wave_max_u64(lane == win ? 0ull : key);
    if (lane == 0) {
        float *m = meta + (size_t)bk * MSTRIDE;
        m[11] = __uint_as_float((unsigned)(second >> 32));
        m[12] = __uint_as_float((unsigned)second);
        m[0] = lo[0]; m[1] = lo[1]; m[2] = lo[2];
        m[3] = hi[0]; m[4] = hi[1]; m[5] = hi[2];
        m[6] = __uint_as_float((unsigned)(best >> 32));
        m[7] = __uint_as_float((unsigned)best);
        m[8] = bx; m[9] = by; m[10] = bz;
    }
}

// ---------------------------------------------------------------- the sampling kernel (one wave per scene)
struct Rec {  // SoA record arrays in LDS
    float *lox, *loy, *loz, *hix, *hiy, *hiz, *bx, *by, *bz;
    unsigned *khi, *klo;
};

__device__ inline Rec carve(float *base, int n) {
    Rec r;
    r.lox = base; r.loy = base + n; r.loz = base + 2 * n;
    r.hix = base + 3 * n; r.hiy = base + 4 * n; r.hiz = base + 5 * n;
    r.bx = base + 6 * n; r.by = base + 7 * n; r.bz = base + 8 * n;
    r.khi = reinterpret_cast<unsigned *>(base + 9 * n);
    r.klo = reinterpret_cast<unsigned *>(base + 10 * n);
    return r;
}

// box distance with the operation sequence of dist_as_written (monotone rounding => lower bound of every point's d)
__device__ inline float box_dist(const Rec &r, int i, float cx, float cy, float cz) {
    const float gx = fmaxf(fmaxf(r.lox[i] - cx, cx - r.hix[i]), 0.f);
    const float gy = fmaxf(fmaxf(r.loy[i] - cy, cy - r.hiy[i]), 0.f);
    const float gz = fmaxf(fmaxf(r.loz[i] - cz, cz - r.hiz[i]), 0.f);
    return pdf_sqdist3(gx, gy, gz);
}

// wave64 unsigned max through DPP (no LDS round trips): row_shr 1/2/4/8 fold each 16-lane row into its last lane,
// row_bcast:15 / row_bcast:31 carry the row results to lane 63, v_readlane broadcasts.  ~8 VALU ops.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_or0(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
    v = max(v, dpp_or0<0x111, 0xf>(v));
    v = max(v, dpp_or0<0x112, 0xf>(v));
    v = max(v, dpp_or0<0x114, 0xf>(v));
    v = max(v, dpp_or0<0x118, 0xf>(v));
    v = max(v, dpp_or0<0x142, 0xa>(v));
    v = max(v, dpp_or0<0x143, 0xc>(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// arg-max of the packed key (hi = ordered tmp, lo = tie key; lo unique among lanes with hi != 0). Returns the lane.
__device__ __forceinline__ int wave_argmax_key(unsigned hi, unsigned lo, unsigned &mhi, unsigned &mlo) {
    mhi = wave_umax(hi);
    mlo = wave_umax(hi == mhi ? lo : 0u);
    const unsigned long long m = __ballot(hi == mhi && lo == mlo);
    return __ffsll((long long)m) - 1;
}
__device__ __forceinline__ float lane_bcast(float v, int lane) {
    return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), lane));
}
// single-wave workgroup: LDS traffic only needs the wave's own DS queue drained, never the VMEM queue
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Row-wise (16-lane) unsigned max: after the four row_shr folds, lanes 15/31/47/63 hold their row's maximum.
__device__ __forceinline__ unsigned row_umax_bcast(unsigned v, int row) {
    v = max(v, dpp_or0<0x111, 0xf>(v));
    v = max(v, dpp_or0<0x112, 0xf>(v));
    v = max(v, dpp_or0<0x114, 0xf>(v));
    v = max(v, dpp_or0<0x118, 0xf>(v));
    const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)v, 15), r1 = (unsigned)__builtin_amdgcn_readlane((int)v, 31);
    const unsigned r2 = (unsigned)__builtin_amdgcn_readlane((int)v, 47), r3 = (unsigned)__builtin_amdgcn_readlane((int)v, 63);
    return row == 0 ? r0 : row == 1 ? r1 : row == 2 ? r2 : r3;
}

constexpr int FPS_UNROLL = 4;  // bucket loads in flight per wave
#ifndef PDF_FPS_MW_UNROLL
#define PDF_FPS_MW_UNROLL 1   // k_fps_mw's update phase is bound by the CU's vector issue (16 waves, ~150 instructions per bucket), not by the
#endif                        // loads: round 4, levels 100k / 25k / 6250: unroll 1 23.5 / 7.1 / 2.7 ms, 2 24.0 / 7.3 / 2.9, 4 26.6 / 8.6 / 3.1, 8 30.8 / 11.3 / 4.3

// NW waves per scene.  Every wave derives the same sample / active-super / active-bucket lists from the shared LDS
// records (redundantly, no communication); the surviving buckets are dealt round-robin to the waves; two barriers per
// sample fence the bucket-record writes.
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_fps(const int *__restrict__ offset, const int *__restrict__ new_offset,
                                                 float4 *__restrict__ pts, const unsigned *__restrict__ kbs,
                                                 const float *__restrict__ meta, int *__restrict__ idx, int nb_cap,
                                                 unsigned *__restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) float fps_lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int row = lane >> 4, col = lane & 15;
    const Scene sc = scene_of(blockIdx.x, offset, new_offset);
    if (sc.m <= 0) return;
    const int nb = sc.nb, ns = (nb + SUP - 1) / SUP;
    const int ns_cap = (nb_cap + SUP - 1) / SUP;
    Rec B = carve(fps_lds, nb_cap);
    Rec S = carve(fps_lds + 11 * nb_cap, ns_cap);
    // active-super list, double-buffered by sample parity (a fast wave may already build sample j+1's list while a slow
    // one still refreshes the supers of sample j); active-bucket list (consumed between barriers A and B)
    unsigned short *slist_base = reinterpret_cast<unsigned short *>(fps_lds + 11 * nb_cap + 11 * ns_cap);
    const int slist_stride = ((ns_cap + 63) & ~63) + 64;
    unsigned short *blist = slist_base + 2 * slist_stride;  // [nb_cap]
    unsigned n_updates = 0, n_supers = 0;

    // load bucket records, build super records
    for (int i = tid; i < nb; i += 64 * NW) {
        const float *m = meta + (size_t)(sc.bbase + i) * MSTRIDE;
        B.lox[i] = m[0]; B.loy[i] = m[1]; B.loz[i] = m[2];
        B.hix[i] = m[3]; B.hiy[i] = m[4]; B.hiz[i] = m[5];
        B.khi[i] = __float_as_uint(m[6]); B.klo[i] = __float_as_uint(m[7]);
        B.bx[i] = m[8]; B.by[i] = m[9]; B.bz[i] = m[10];
    }
    __syncthreads();
    for (int s = tid; s < ns; s += 64 * NW) {
        float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        unsigned long long best = 0ull;
        int bi = s * SUP;
        for (int i = s * SUP; i < min(nb, s * SUP + SUP); ++i) {
            lo[0] = fminf(lo[0], B.lox[i]); lo[1] = fminf(lo[1], B.loy[i]); lo[2] = fminf(lo[2], B.loz[i]);
            hi[0] = fmaxf(hi[0], B.hix[i]); hi[1] = fmaxf(hi[1], B.hiy[i]); hi[2] = fmaxf(hi[2], B.hiz[i]);
            const unsigned long long k = ((unsigned long long)B.khi[i] << 32) | B.klo[i];
            if (k > best) { best = k; bi = i; }
        }
        S.lox[s] = lo[0]; S.loy[s] = lo[1]; S.loz[s] = lo[2];
        S.hix[s] = hi[0]; S.hiy[s] = hi[1]; S.hiz[s] = hi[2];
        S.khi[s] = (unsigned)(best >> 32); S.klo[s] = (unsigned)best;
        S.bx[s] = B.bx[bi]; S.by[s] = B.by[bi]; S.bz[s] = B.bz[bi];
    }
    __syncthreads();

    if (tid == 0) idx[sc.start_m] = sc.start_n;
    // sample 1 needs no update pass: tmp already holds the distances to sample 0 (k_scatter)
    for (int j = 1; j < sc.m; ++j) {
        // ---- 1. arg-max over super records (every wave, redundantly)
        unsigned bhi = 0u, blo = 0u;
        int bs = 0;
        for (int s = lane; s < ns; s += 64) {
            const unsigned h = S.khi[s], l = S.klo[s];
            if (h > bhi || (h == bhi && l > blo)) { bhi = h; blo = l; bs = s; }
        }
        unsigned whi, wlo;
        const int wl = wave_argmax_key(bhi, blo, whi, wlo);
        const int wsup = __builtin_amdgcn_readlane(bs, wl);
        const float cx = S.bx[wsup], cy = S.by[wsup], cz = S.bz[wsup];
        if (tid == 0) idx[sc.start_m + j] = sc.start_n + (int)(~wlo & REL_MASK);
        if (j == sc.m - 1) break;  // the last sample needs no distance update

        // ---- 2. active supers -> slist
        unsigned short *slist = slist_base + (j & 1) * slist_stride;
        int n_sup = 0;
        for (int s0 = 0; s0 < ns; s0 += 64) {
            const int s = s0 + lane;
            bool act = false;
            if (s < ns) act = pdf_f32_ordered(box_dist(S, s, cx, cy, cz)) < S.khi[s];
            const unsigned long long m = __ballot(act);
            if (act) slist[n_sup + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)s;
            n_sup += __popcll(m);
        }
        lds_fence();
        // ---- 3. active buckets of those supers (4 supers x 16 buckets per pass) -> blist
        int n_bk = 0;
        for (int g = 0; g < n_sup; g += 4) {
            bool bact = false;
            int bk = 0;
            if (g + row < n_sup) {
                bk = (int)slist[g + row] * SUP + col;
                if (bk < nb) bact = pdf_f32_ordered(box_dist(B, bk, cx, cy, cz)) < B.khi[bk];
            }
            const unsigned long long m = __ballot(bact);
            if (bact) blist[n_bk + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)bk;
            n_bk += __popcll(m);
        }
        n_supers += n_sup;
        n_updates += n_bk;
        if (NW > 1) __syncthreads(); else lds_fence();  // A: nobody still reads bucket records / every list is complete

        // ---- 4. update the surviving buckets (dealt round-robin to the waves, FPS_UNROLL loads in flight)
        for (int i0 = wave * FPS_UNROLL; i0 < n_bk; i0 += NW * FPS_UNROLL) {
            float4 p[FPS_UNROLL];
            unsigned kb[FPS_UNROLL];
            int bkid[FPS_UNROLL];
#pragma unroll
            for (int u = 0; u < FPS_UNROLL; ++u) {
                bkid[u] = i0 + u < n_bk ? (int)blist[i0 + u] : -1;
                if (bkid[u] >= 0) {
                    const size_t pos = (size_t)sc.pbase + (size_t)bkid[u] * BSZ + lane;
                    p[u] = pts[pos];
                    kb[u] = kbs[pos];
                }
            }
#pragma unroll
            for (int u = 0; u < FPS_UNROLL; ++u) {
                if (bkid[u] < 0) break;
                unsigned khi = 0u;
                if (p[u].w >= 0.f) {
                    const float d = dist_as_written(p[u].x, p[u].y, p[u].z, cx, cy, cz);
                    if (d < p[u].w) {
                        p[u].w = d;
                        pts[(size_t)sc.pbase + (size_t)bkid[u] * BSZ + lane].w = d;
                    }
                    khi = pdf_f32_ordered(p[u].w);
                }
                unsigned mhi, mlo;
                const int kl = wave_argmax_key(khi, kb[u], mhi, mlo);
                const float nx = lane_bcast(p[u].x, kl), ny = lane_bcast(p[u].y, kl), nz = lane_bcast(p[u].z, kl);
                if (lane == 0) {
                    B.khi[bkid[u]] = mhi; B.klo[bkid[u]] = mlo;
                    B.bx[bkid[u]] = nx; B.by[bkid[u]] = ny; B.bz[bkid[u]] = nz;
                }
            }
        }
        if (NW > 1) __syncthreads(); else lds_fence();  // B: bucket records + tmp stores of every wave are visible

        // ---- 5. refresh the touched super records (every wave, redundantly; 4 supers per pass, one per 16-lane row)
        for (int g = 0; g < n_sup; g += 4) {
            unsigned h16 = 0u, l16 = 0u;
            int sup = -1, bk = 0;
            if (g + row < n_sup) {
                sup = (int)slist[g + row];
                bk = sup * SUP + col;
                if (bk < nb) { h16 = B.khi[bk]; l16 = B.klo[bk]; }
            }
            const unsigned mh = row_umax_bcast(h16, row);
            const unsigned ml = row_umax_bcast(h16 == mh ? l16 : 0u, row);
            const unsigned long long m = __ballot(sup >= 0 && h16 == mh && l16 == ml);
            if (col == 0 && sup >= 0) {
                const int wcol = __ffs((unsigned)(m >> (16 * row)) & 0xffffu) - 1;
                const int wb = sup * SUP + wcol;
                S.khi[sup] = mh; S.klo[sup] = ml;
                S.bx[sup] = B.bx[wb]; S.by[sup] = B.by[wb]; S.bz[sup] = B.bz[wb];
            }
        }
        lds_fence();
    }
    if (tid == 0 && stats) {
        stats[blockIdx.x * 4 + 0] = n_updates;
        stats[blockIdx.x * 4 + 1] = n_supers;
        stats[blockIdx.x * 4 + 2] = (unsigned)sc.m;
        stats[blockIdx.x * 4 + 3] = (unsigned)nb;
    }
}

// ---------------------------------------------------------------- several samples per round (exact)
// The per-sample chain above is pure latency (arg-max -> lists -> one L2 round trip -> records -> two barriers: ~3.6 us).
// k_fps_multi emits up to KMAX samples per round of the same chain.  Let c_1 be the arg-max and c_t (t >= 2) the best
// point outside the super-buckets of c_1..c_{t-1}.  c_t is accepted -- it IS sample j+t-1 of the sequential algorithm -- if
//   (a) key(c_t) > second-best key (before this round) of every super-bucket that holds an earlier candidate, and
//   (b) d(c_u, c_t) >= tmp[c_t] for every earlier candidate c_u (as-written fp32 distance).
// Proof: inserting c_1..c_{t-1} only lowers tmp values.  By (b) tmp[c_t] keeps its value; every other point outside the
// earlier candidates' supers had a smaller key than c_t already; inside those supers every point other than the candidate
// itself is bounded by the super's second-best key, which (a) puts below key(c_t); the candidates themselves drop to
// tmp = 0.  Keys are unique (the low word carries the point's index), so c_t is the unique arg-max after t-1 insertions.
// Both tests use values from before the round, the accepted prefix is known before any update, and the updates
// tmp = min(tmp, d(., c_u)) commute -- one pass over the union of the touched buckets applies them all.
// Records carry the second-best key per bucket and per super-bucket for (a).
struct Rec2 {
    float *lox, *loy, *loz, *hix, *hiy, *hiz, *bx, *by, *bz;
    unsigned *khi, *klo, *k2hi, *k2lo;
};
constexpr int REC2 = 13;
__device__ inline Rec2 carve2(float *base, int n) {
    Rec2 r;
    r.lox = base; r.loy = base + n; r.loz = base + 2 * n;
    r.hix = base + 3 * n; r.hiy = base + 4 * n; r.hiz = base + 5 * n;
    r.bx = base + 6 * n; r.by = base + 7 * n; r.bz = base + 8 * n;
    r.khi = reinterpret_cast<unsigned *>(base + 9 * n);
    r.klo = reinterpret_cast<unsigned *>(base + 10 * n);
    r.k2hi = reinterpret_cast<unsigned *>(base + 11 * n);
    r.k2lo = reinterpret_cast<unsigned *>(base + 12 * n);
    return r;
}
__device__ __forceinline__ float box_dist_v(float lox, float loy, float loz, float hix, float hiy, float hiz, float cx, float cy, float cz) {
    const float gx = fmaxf(fmaxf(lox - cx, cx - hix), 0.f);
    const float gy = fmaxf(fmaxf(loy - cy, cy - hiy), 0.f);
    const float gz = fmaxf(fmaxf(loz - cz, cz - hiz), 0.f);
    return pdf_sqdist3(gx, gy, gz);
}
__device__ __forceinline__ bool key_gt(unsigned ah, unsigned al, unsigned bh, unsigned bl) { return ah > bh || (ah == bh && al > bl); }

template <int NW, int KMAX>
__global__ __launch_bounds__(64 * NW) void k_fps_multi(const int *__restrict__ offset, const int *__restrict__ new_offset,
                                                       float4 *__restrict__ pts, const unsigned *__restrict__ kbs,
                                                       const float *__restrict__ meta, int *__restrict__ idx, int nb_cap,
                                                       unsigned *__restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) float fps_lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int row = lane >> 4, col = lane & 15;
    const Scene sc = scene_of(blockIdx.x, offset, new_offset);
    if (sc.m <= 0) return;
    const int nb = sc.nb, ns = (nb + SUP - 1) / SUP;
    const int ns_cap = (nb_cap + SUP - 1) / SUP;
    Rec2 B = carve2(fps_lds, nb_cap);
    Rec2 S = carve2(fps_lds + REC2 * nb_cap, ns_cap);
    unsigned short *slist_base = reinterpret_cast<unsigned short *>(fps_lds + REC2 * nb_cap + REC2 * ns_cap);
    const int slist_stride = ((ns_cap + 63) & ~63) + 64;
    unsigned short *blist = slist_base + 2 * slist_stride;  // [nb_cap]
    unsigned n_updates = 0, n_supers = 0, n_rounds = 0;
#ifdef FPS_PROFILE
    unsigned long long tc[5] = {0, 0, 0, 0, 0}, t_prev = __builtin_readcyclecounter();
#define FPS_TICK(i_) do { const unsigned long long t_now = __builtin_readcyclecounter(); tc[i_] += t_now - t_prev; t_prev = t_now; } while (0)
#else
#define FPS_TICK(i_) do {} while (0)
#endif

    for (int i = tid; i < nb; i += 64 * NW) {
        const float *m = meta + (size_t)(sc.bbase + i) * MSTRIDE;
        B.lox[i] = m[0]; B.loy[i] = m[1]; B.loz[i] = m[2];
        B.hix[i] = m[3]; B.hiy[i] = m[4]; B.hiz[i] = m[5];
        B.khi[i] = __float_as_uint(m[6]); B.klo[i] = __float_as_uint(m[7]);
        B.bx[i] = m[8]; B.by[i] = m[9]; B.bz[i] = m[10];
        B.k2hi[i] = __float_as_uint(m[11]); B.k2lo[i] = __float_as_uint(m[12]);
    }
    __syncthreads();
    for (int s = tid; s < ns; s += 64 * NW) {
        float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        unsigned long long best = 0ull, second = 0ull;
        int bi = s * SUP;
        for (int i = s * SUP; i < min(nb, s * SUP + SUP); ++i) {
            lo[0] = fminf(lo[0], B.lox[i]); lo[1] = fminf(lo[1], B.loy[i]); lo[2] = fminf(lo[2], B.loz[i]);
            hi[0] = fmaxf(hi[0], B.hix[i]); hi[1] = fmaxf(hi[1], B.hiy[i]); hi[2] = fmaxf(hi[2], B.hiz[i]);
            const unsigned long long k = ((unsigned long long)B.khi[i] << 32) | B.klo[i];
            const unsigned long long k2 = ((unsigned long long)B.k2hi[i] << 32) | B.k2lo[i];
            if (k > best) { second = best > k2 ? best : k2; best = k; bi = i; }
            else if (k > second) second = k;   // (k2 < k <= best: k2 cannot beat k)
        }
        S.lox[s] = lo[0]; S.loy[s] = lo[1]; S.loz[s] = lo[2];
        S.hix[s] = hi[0]; S.hiy[s] = hi[1]; S.hiz[s] = hi[2];
        S.khi[s] = (unsigned)(best >> 32); S.klo[s] = (unsigned)best;
        S.k2hi[s] = (unsigned)(second >> 32); S.k2lo[s] = (unsigned)second;
        S.bx[s] = B.bx[bi]; S.by[s] = B.by[bi]; S.bz[s] = B.bz[bi];
    }
    __syncthreads();

    if (tid == 0) idx[sc.start_m] = sc.start_n;
    int j = 1;
    FPS_TICK(4);
    while (j < sc.m) {
        // ---- 1. candidates (every wave, redundantly): per-lane best over its supers, then up to KMAX wave arg-maxes
        float cx[KMAX], cy[KMAX], cz[KMAX];
        int csup[KMAX];
        unsigned bhi = 0u, blo = 0u;
        int bs = 0;
        for (int s = lane; s < ns; s += 64) {
            const unsigned h = S.khi[s], l = S.klo[s];
            if (key_gt(h, l, bhi, blo)) { bhi = h; blo = l; bs = s; }
        }
        const int kmax = min(KMAX, sc.m - j);
        int a = 0;
        unsigned bound_hi = 0u, bound_lo = 0u;   // max second-best key over the accepted candidates' supers
#pragma unroll
        for (int t = 0; t < KMAX; ++t) {
            if (t >= kmax) break;
            unsigned whi, wlo;
            const int wl = wave_argmax_key(bhi, blo, whi, wlo);
            if (t > 0 && !key_gt(whi, wlo, bound_hi, bound_lo)) break;          // (a)  (also stops at key 0: no valid point left)
            const int wsup = __builtin_amdgcn_readlane(bs, wl);
            const float x = S.bx[wsup], y = S.by[wsup], z = S.bz[wsup];
            bool far = true;                                                       // (b)
#pragma unroll
            for (int u = 0; u < KMAX; ++u)
                if (u < t) far = far && !(pdf_f32_ordered(dist_as_written(x, y, z, cx[u], cy[u], cz[u])) < whi);
            if (!far) break;
            cx[t] = x; cy[t] = y; cz[t] = z; csup[t] = wsup;
            a = t + 1;
            if (tid == 0) idx[sc.start_m + j + t] = sc.start_n + (int)(~wlo & REL_MASK);
            const unsigned s2h = S.k2hi[wsup], s2l = S.k2lo[wsup];
            if (key_gt(s2h, s2l, bound_hi, bound_lo)) { bound_hi = s2h; bound_lo = s2l; }
            if (t + 1 < kmax && lane == wl) {   // the owning lane re-scans its supers without the taken ones
                bhi = 0u; blo = 0u; bs = 0;
                for (int s = lane; s < ns; s += 64) {
                    bool taken = false;
#pragma unroll
                    for (int u = 0; u < KMAX; ++u) if (u <= t) taken = taken || s == csup[u];
                    const unsigned h = S.khi[s], l = S.klo[s];
                    if (!taken && key_gt(h, l, bhi, blo)) { bhi = h; blo = l; bs = s; }
                }
            }
        }
        // a >= 1 here (t = 0 always accepts)
        j += a;
        ++n_rounds;
        if (j >= sc.m) break;  // the last sample needs no distance update
        FPS_TICK(0);

        // ---- 2. active supers (touched by any accepted centre) -> slist
        unsigned short *slist = slist_base + (n_rounds & 1) * slist_stride;
        int n_sup = 0;
        for (int s0 = 0; s0 < ns; s0 += 64) {
            const int s = s0 + lane;
            bool act = false;
            if (s < ns) {
                const float lx = S.lox[s], ly = S.loy[s], lz = S.loz[s], hx = S.hix[s], hy = S.hiy[s], hz = S.hiz[s];
                const unsigned kh = S.khi[s];
#pragma unroll
                for (int u = 0; u < KMAX; ++u)
                    if (u < a) act = act || pdf_f32_ordered(box_dist_v(lx, ly, lz, hx, hy, hz, cx[u], cy[u], cz[u])) < kh;
            }
            const unsigned long long m = __ballot(act);
            if (act) slist[n_sup + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)s;
            n_sup += __popcll(m);
        }
        lds_fence();
        // ---- 3. active buckets of those supers -> blist
        int n_bk = 0;
        for (int g = 0; g < n_sup; g += 4) {
            bool bact = false;
            int bk = 0;
            if (g + row < n_sup) {
                bk = (int)slist[g + row] * SUP + col;
                if (bk < nb) {
                    const float lx = B.lox[bk], ly = B.loy[bk], lz = B.loz[bk], hx = B.hix[bk], hy = B.hiy[bk], hz = B.hiz[bk];
                    const unsigned kh = B.khi[bk];
#pragma unroll
                    for (int u = 0; u < KMAX; ++u)
                        if (u < a) bact = bact || pdf_f32_ordered(box_dist_v(lx, ly, lz, hx, hy, hz, cx[u], cy[u], cz[u])) < kh;
                }
            }
            const unsigned long long m = __ballot(bact);
            if (bact) blist[n_bk + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)bk;
            n_bk += __popcll(m);
        }
        n_supers += n_sup;
        n_updates += n_bk;
        if (NW > 1) __syncthreads(); else lds_fence();  // A
        FPS_TICK(1);

        // ---- 4. update the surviving buckets with every accepted centre
        for (int i0 = wave * FPS_UNROLL; i0 < n_bk; i0 += NW * FPS_UNROLL) {
            float4 p[FPS_UNROLL];
            unsigned kb[FPS_UNROLL];
            int bkid[FPS_UNROLL];
#pragma unroll
            for (int u = 0; u < FPS_UNROLL; ++u) {
                bkid[u] = i0 + u < n_bk ? (int)blist[i0 + u] : -1;
                if (bkid[u] >= 0) {
                    const size_t pos = (size_t)sc.pbase + (size_t)bkid[u] * BSZ + lane;
                    p[u] = pts[pos];
                    kb[u] = kbs[pos];
                }
            }
#pragma unroll
            for (int u = 0; u < FPS_UNROLL; ++u) {
                if (bkid[u] < 0) break;
                unsigned khi = 0u;
                if (p[u].w >= 0.f) {
                    float w = p[u].w;
#pragma unroll
                    for (int v = 0; v < KMAX; ++v)
                        if (v < a) {
                            const float d = dist_as_written(p[u].x, p[u].y, p[u].z, cx[v], cy[v], cz[v]);
                            if (d < w) w = d;
                        }
                    if (w < p[u].w) pts[(size_t)sc.pbase + (size_t)bkid[u] * BSZ + lane].w = w;
                    khi = pdf_f32_ordered(w);
                }
                unsigned mhi, mlo;
                const int kl = wave_argmax_key(khi, kb[u], mhi, mlo);
                const unsigned h2 = lane == kl ? 0u : khi, l2 = lane == kl ? 0u : kb[u];
                const unsigned m2hi = wave_umax(h2);
                const unsigned m2lo = wave_umax(h2 == m2hi ? l2 : 0u);
                const float nx = lane_bcast(p[u].x, kl), ny = lane_bcast(p[u].y, kl), nz = lane_bcast(p[u].z, kl);
                if (lane == 0) {
                    B.khi[bkid[u]] = mhi; B.klo[bkid[u]] = mlo;
                    B.k2hi[bkid[u]] = m2hi; B.k2lo[bkid[u]] = m2lo;
                    B.bx[bkid[u]] = nx; B.by[bkid[u]] = ny; B.bz[bkid[u]] = nz;
                }
            }
        }
        if (NW > 1) __syncthreads(); else lds_fence();  // B
        FPS_TICK(2);

        // ---- 5. refresh the touched super records (best + second-best; 4 supers per pass, one per 16-lane row)
        for (int g = 0; g < n_sup; g += 4) {
            unsigned h16 = 0u, l16 = 0u, h2 = 0u, l2 = 0u;
            int sup = -1, bk = 0;
            if (g + row < n_sup) {
                sup = (int)slist[g + row];
                bk = sup * SUP + col;
                if (bk < nb) { h16 = B.khi[bk]; l16 = B.klo[bk]; h2 = B.k2hi[bk]; l2 = B.k2lo[bk]; }
            }
            const unsigned mh = row_umax_bcast(h16, row);
            const unsigned ml = row_umax_bcast(h16 == mh ? l16 : 0u, row);
            const unsigned long long m = __ballot(sup >= 0 && h16 == mh && l16 == ml);
            const int wcol = __ffs((unsigned)(m >> (16 * row)) & 0xffffu) - 1;
            // second-best of the super: the winning bucket contributes its second-best key, the others their best
            const unsigned sh = col == wcol ? h2 : h16, sl = col == wcol ? l2 : l16;
            const unsigned m2h = row_umax_bcast(sh, row);
            const unsigned m2l = row_umax_bcast(sh == m2h ? sl : 0u, row);
            if (col == 0 && sup >= 0) {
                const int wb = sup * SUP + wcol;
                S.khi[sup] = mh; S.klo[sup] = ml;
                S.k2hi[sup] = m2h; S.k2lo[sup] = m2l;
                S.bx[sup] = B.bx[wb]; S.by[sup] = B.by[wb]; S.bz[sup] = B.bz[wb];
            }
        }
        lds_fence();
        FPS_TICK(3);
    }
    if (tid == 0 && stats) {
        stats[blockIdx.x * 4 + 0] = n_updates;
        stats[blockIdx.x * 4 + 1] = n_supers;
        stats[blockIdx.x * 4 + 2] = (unsigned)sc.m;
        stats[blockIdx.x * 4 + 3] = n_rounds;
#ifdef FPS_PROFILE
        for (int i = 0; i < 4; ++i) stats[blockIdx.x * 4 + i] = (unsigned)(tc[i] >> 10);   // kilo-cycles: candidates | lists | updates | refresh
#endif
    }
#undef FPS_TICK
}

// k_fps_mw, RANK: for every unit (super-bucket or bucket) the units of this wave's slice with a greater key, added to rank[] in LDS; then
// (wave 0, after a barrier) the units of rank < KMAX placed best first into cand[].  N: units per lane.
template <int N, int NW>
__device__ __forceinline__ void rank_count(const Rec2 &U, int nu, unsigned *rank, int lane, int wave) {
    const int js = (nu + NW - 1) / NW, j0 = wave * js, j1 = min(nu, j0 + js);
    unsigned long long mykey[N];
    unsigned above[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int s = lane + 64 * i;
        mykey[i] = s < nu ? (((unsigned long long)U.khi[s] << 32) | U.klo[s]) : ~0ull;
        above[i] = 0u;
    }
    for (int jj = j0; jj < j1; ++jj) {
        const unsigned long long kj = ((unsigned long long)U.khi[jj] << 32) | U.klo[jj];
#pragma unroll
        for (int i = 0; i < N; ++i) above[i] += kj > mykey[i] ? 1u : 0u;
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (above[i]) atomicAdd(&rank[lane + 64 * i], above[i]);   // (above != 0 only for s < nu)
}
template <int N, int KMAX>
__device__ __forceinline__ void rank_place(const Rec2 &U, int nu, unsigned *rank, int *cand, int lane) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int s = lane + 64 * i;
        if (s < nu) {
            const unsigned rk = rank[s];
            rank[s] = 0u;
            if (rk < (unsigned)KMAX && (U.khi[s] | U.klo[s]) != 0u) cand[rk] = s;   // (keys are unique unless 0: ranks too)
        }
    }
}

// ---------------------------------------------------------------- one centre per wave
// k_fps_multi still derives every list redundantly in every wave; measured (s_memtime, 100k-point scene, K = 8): candidate
// selection 11.6k cycles per round, lists 13.7k, updates 24.7k, super refresh 8.8k -- instruction latency of ONE wave,
// not memory.  k_fps_mw keeps the acceptance rule (same proof) and spreads the rest over NW = KMAX waves:
//   phase 1  every wave, redundantly: up to NW accepted centres (registers).
//   phase 2  wave u owns centre u: its active supers (private list), then the active buckets of those supers.  A bucket or
//            super touched by several centres is CLAIMED once (ds exchange of the round tag) and appended to the shared
//            lists through a wave-aggregated LDS counter.
//   barrier A
//   phase 3  the shared bucket list is dealt round-robin to the waves; a bucket gets the centres whose bit is set in its
//            claim word (the owner clears the word).
//   barrier B
//   phase 4  the shared super list is dealt to the waves (4 supers per pass); barrier C.
template <int NW, bool RANK>
__global__ __launch_bounds__(64 * NW) void k_fps_mw(const int *__restrict__ offset, const int *__restrict__ new_offset,
                                                    float4 *__restrict__ pts, const unsigned *__restrict__ kbs,
                                                    const float *__restrict__ meta, int *__restrict__ idx, int nb_cap,
                                                    unsigned *__restrict__ stats, int two_level) {
    constexpr int KMAX = NW;
    constexpr int UNR = PDF_FPS_MW_UNROLL;   // bucket loads in flight per wave
    constexpr int NSL = (NS_MAX + 63) / 64;   // supers per lane
    extern __shared__ __attribute__((aligned(16))) float fps_lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int row = lane >> 4, col = lane & 15;
    const Scene sc = scene_of(blockIdx.x, offset, new_offset);
    if (sc.m <= 0) return;
    const int nb = sc.nb, ns = (nb + SUP - 1) / SUP;
    const int ns_cap = (nb_cap + SUP - 1) / SUP;
    Rec2 B = carve2(fps_lds, nb_cap);
    Rec2 S = carve2(fps_lds + REC2 * nb_cap, ns_cap);
    unsigned *claimB = reinterpret_cast<unsigned *>(fps_lds + REC2 * nb_cap + REC2 * ns_cap);   // [nb_cap] round tag of the claim
    unsigned *claimS = claimB + nb_cap;                                                          // [ns_cap]
    unsigned *cnt = claimS + ns_cap;                                                             // [4]: n_bk[2], n_sup[2] by round parity
    float *cent = reinterpret_cast<float *>(cnt + 4);                                            // [3 * 16] the round's centres
    int *cent_n = reinterpret_cast<int *>(cent + 3 * 16);                                      // [4]
    const int nr_cap = nb_cap <= BKC_MAX ? nb_cap : ns_cap;
    unsigned *rank = reinterpret_cast<unsigned *>(cent_n + 4);                                   // [nr_cap] units with a greater key (RANK)
    int *cand = reinterpret_cast<int *>(rank + nr_cap);                                          // [16] the round's candidate units, best first
    unsigned *c_s2h = reinterpret_cast<unsigned *>(cand + 16), *c_s2l = c_s2h + 16;              // [16] [16] their supers' second-best keys
    unsigned *rank2 = c_s2l + 16;                                                                // [16 * SUP] buckets of the best supers with a greater key
    unsigned short *blist = reinterpret_cast<unsigned short *>(rank2 + 16 * SUP);                // [nb_cap + 64] shared
    unsigned short *slist = blist + nb_cap + 64;                                                 // [ns_cap + 64] shared
    const int psl_stride = ((ns_cap + 63) & ~63) + 64;
    unsigned short *pslist = slist + ns_cap + 64 + wave * psl_stride;                            // private per wave
    unsigned n_updates = 0, n_supers = 0, n_rounds = 0;
#ifdef FPS_PROFILE
    unsigned long long tc[5] = {0, 0, 0, 0, 0}, t_prev = __builtin_readcyclecounter();
#define FPS_TICK(i_) do { const unsigned long long t_now = __builtin_readcyclecounter(); tc[i_] += t_now - t_prev; t_prev = t_now; } while (0)
#else
#define FPS_TICK(i_) do {} while (0)
#endif

    for (int i = tid; i < nb; i += 64 * NW) {
        const float *m = meta + (size_t)(sc.bbase + i) * MSTRIDE;
        B.lox[i] = m[0]; B.loy[i] = m[1]; B.loz[i] = m[2];
        B.hix[i] = m[3]; B.hiy[i] = m[4]; B.hiz[i] = m[5];
        B.khi[i] = __float_as_uint(m[6]); B.klo[i] = __float_as_uint(m[7]);
        B.bx[i] = m[8]; B.by[i] = m[9]; B.bz[i] = m[10];
        B.k2hi[i] = __float_as_uint(m[11]); B.k2lo[i] = __float_as_uint(m[12]);
        claimB[i] = 0u;
    }
    if (tid < 4) cnt[tid] = 0u;
    if (tid < 16) cand[tid] = -1;
    __syncthreads();
    for (int s = tid; s < ns; s += 64 * NW) {
        float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        unsigned long long best = 0ull, second = 0ull;
        int bi = s * SUP;
        for (int i = s * SUP; i < min(nb, s * SUP + SUP); ++i) {
            lo[0] = fminf(lo[0], B.lox[i]); lo[1] = fminf(lo[1], B.loy[i]); lo[2] = fminf(lo[2], B.loz[i]);
            hi[0] = fmaxf(hi[0], B.hix[i]); hi[1] = fmaxf(hi[1], B.hiy[i]); hi[2] = fmaxf(hi[2], B.hiz[i]);
            const unsigned long long k = ((unsigned long long)B.khi[i] << 32) | B.klo[i];
            const unsigned long long k2 = ((unsigned long long)B.k2hi[i] << 32) | B.k2lo[i];
            if (k > best) { second = best > k2 ? best : k2; best = k; bi = i; }
            else if (k > second) second = k;
        }
        S.lox[s] = lo[0]; S.loy[s] = lo[1]; S.loz[s] = lo[2];
        S.hix[s] = hi[0]; S.hiy[s] = hi[1]; S.hiz[s] = hi[2];
        S.khi[s] = (unsigned)(best >> 32); S.klo[s] = (unsigned)best;
        S.k2hi[s] = (unsigned)(second >> 32); S.k2lo[s] = (unsigned)second;
        S.bx[s] = B.bx[bi]; S.by[s] = B.by[bi]; S.bz[s] = B.bz[bi];
        claimS[s] = 0u;
    }
    for (int i = tid; i < nr_cap; i += 64 * NW) rank[i] = 0u;
    for (int i = tid; i < 16 * SUP; i += 64 * NW) rank2[i] = 0u;
    __syncthreads();

    if (tid == 0) idx[sc.start_m] = sc.start_n;
    int j = 1;
    FPS_TICK(4);
    while (j < sc.m) {
        // ---- 1. candidates.  The round's candidates are the KMAX supers with the largest best keys, best first (c_t = the best point
        // outside the supers of c_1..c_{t-1}); tests (a) and (b) of k_fps_multi only use values from before the round, so the accepted
        // prefix is the first t that fails either.  Rounds 2-3 found the candidates one wave arg-max after the other in wave 0 (~1,450
        // cycles each, 16k of the 31k cycles of a level-1 round).  RANK: every wave counts, for every super, the supers of ITS slice with a
        // greater key (broadcast LDS reads, LDS adds); after barrier R wave 0 places the supers of rank < KMAX, lane t takes candidate t
        // and all tests run at once.
        if (RANK) {
            // The candidate UNITS: super-buckets (1,024 points) at level 1; the buckets themselves (64 points) when the scene has at most
            // BKC_MAX of them (levels 2+) -- a scene of 25k points has 25 supers, one of 6,250 has 7, which capped a round at that many
            // centres and let test (a) (second-best key of a 1,024-point unit) end it earlier still.  The proof of k_fps_multi holds for any
            // partition into units that carry their best and second-best key.
            const bool bkc = nb_cap <= BKC_MAX;   // (by the launch's largest scene: the rank array is sized for it; either way the same samples)
            // Larger scenes, two levels: the KMAX best buckets lie inside the KMAX best super-buckets (a super whose best key is below the
            // KMAX-th best bucket key holds none of them), so rank the supers, then the <= KMAX * SUP buckets of the KMAX best supers.
            // (measured: 25k-point scenes 7.2 -> 5.5 ms; 100k-point scenes 22.5 -> 23.3 -- with 98 supers the two extra barriers cost more than
            // the better yield returns: only up to TWO_MAX_SUPERS supers)
            const bool two = !bkc && two_level && ns_cap <= TWO_MAX_SUPERS;
            const Rec2 &U_ = (bkc || two) ? B : S;
            const int nu = bkc ? nb : ns;
            if (bkc) rank_count<BKC_MAX / 64, NW>(B, nu, rank, lane, wave); else rank_count<NSL, NW>(S, nu, rank, lane, wave);
            __syncthreads();  // R: ranks complete
            if (two) {
                if (wave == 0) rank_place<NSL, KMAX>(S, ns, rank, cand, lane);
                __syncthreads();  // R2: the KMAX best supers are listed
                constexpr int NSLOT = KMAX * SUP, NQ = NSLOT / 64, JS = NSLOT / NW;
                unsigned long long mykey[NQ];
                unsigned above[NQ];
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    const int q = lane + 64 * i, sup = cand[q / SUP], bk = sup * SUP + q % SUP;
                    mykey[i] = (sup >= 0 && bk < nb) ? (((unsigned long long)B.khi[bk] << 32) | B.klo[bk]) : 0ull;
                    above[i] = 0u;
                }
                for (int q = wave * JS; q < wave * JS + JS; ++q) {
                    const int sup = cand[q / SUP], bk = sup * SUP + q % SUP;
                    const unsigned long long kj = (sup >= 0 && bk < nb) ? (((unsigned long long)B.khi[bk] << 32) | B.klo[bk]) : 0ull;
#pragma unroll
                    for (int i = 0; i < NQ; ++i) above[i] += kj > mykey[i] ? 1u : 0u;
                }
#pragma unroll
                for (int i = 0; i < NQ; ++i)
                    if (above[i]) atomicAdd(&rank2[lane + 64 * i], above[i]);
                __syncthreads();  // R3: bucket ranks complete
                if (wave == 0) {
                    int mybk[NQ];
#pragma unroll
                    for (int i = 0; i < NQ; ++i) {
                        const int q = lane + 64 * i, sup = cand[q / SUP];
                        mybk[i] = sup >= 0 && sup * SUP + q % SUP < nb ? sup * SUP + q % SUP : -1;
                    }
                    lds_fence();
                    if (lane < 16) cand[lane] = -1;
                    lds_fence();
#pragma unroll
                    for (int i = 0; i < NQ; ++i) {
                        const unsigned rk = rank2[lane + 64 * i];
                        rank2[lane + 64 * i] = 0u;
                        if (mybk[i] >= 0 && rk < (unsigned)KMAX && (B.khi[mybk[i]] | B.klo[mybk[i]]) != 0u) cand[rk] = mybk[i];
                    }
                }
            }
            if (wave == 0) {
                constexpr int G = 64 / KMAX;          // lane groups of test (b): lane = (g, t)
                const int kmax = min(KMAX, sc.m - j);
                if (!two) { if (bkc) rank_place<BKC_MAX / 64, KMAX>(B, nu, rank, cand, lane); else rank_place<NSL, KMAX>(S, nu, rank, cand, lane); }
                lds_fence();
                const int t = lane % KMAX, g = lane / KMAX;
                const int cs = cand[t];
                const bool valid = cs >= 0 && t < kmax;
                const int c0 = cs >= 0 ? cs : 0;
                const unsigned kh = U_.khi[c0], kl = U_.klo[c0];
                const float x = U_.bx[c0], y = U_.by[c0], z = U_.bz[c0];
                if (lane < KMAX) {
                    cent[3 * t + 0] = x; cent[3 * t + 1] = y; cent[3 * t + 2] = z;
                    c_s2h[t] = valid ? U_.k2hi[c0] : 0u; c_s2l[t] = valid ? U_.k2lo[c0] : 0u;
                }
                lds_fence();
                bool fail = !valid;
                {   // (a) the key beats the second-best key of every earlier candidate's super
                    unsigned long long bound = 0ull;
#pragma unroll
                    for (int u = 0; u < KMAX - 1; ++u) {
                        const unsigned long long k2 = ((unsigned long long)c_s2h[u] << 32) | c_s2l[u];
                        if (u < t && k2 > bound) bound = k2;
                    }
                    const unsigned long long key = ((unsigned long long)kh << 32) | kl;
                    fail = fail || (t > 0 && !(key > bound));
                }
#pragma unroll
                for (int uu = 0; uu < KMAX / G; ++uu) {   // (b) no earlier candidate lowers tmp of this one
                    const int u = g + G * uu;
                    const bool near = pdf_f32_ordered(dist_as_written(x, y, z, cent[3 * u + 0], cent[3 * u + 1], cent[3 * u + 2])) < kh;
                    fail = fail || (u < t && near);
                }
                unsigned long long fm = __ballot(fail);
#pragma unroll
                for (int sft = KMAX; sft < 64; sft <<= 1) fm |= fm >> sft;
                const unsigned f = (unsigned)fm & ((1u << KMAX) - 1u) & ~1u;       // (candidate 0 is always accepted)
                const int acc = f ? __ffs(f) - 1 : KMAX;
                if (lane < acc) idx[sc.start_m + j + lane] = sc.start_n + (int)(~kl & REL_MASK);
                if (lane == 0) cent_n[0] = acc;
                if (lane < 16) cand[lane] = -1;
            }
        } else
        // (rounds 2-3) WAVE 0 ONLY, one candidate after the other: each lane keeps the records of ITS supers (s = lane + 64 i) in
        // registers for the round; lane u holds accepted centre u, so test (b) is one distance per lane.
        if (wave == 0) {
            unsigned skh[NSL], skl[NSL], sk2h[NSL], sk2l[NSL];
            float sbx[NSL], sby[NSL], sbz[NSL];
#pragma unroll
            for (int i = 0; i < NSL; ++i) {
                const int s = lane + 64 * i;
                const bool ok = s < ns;
                skh[i] = ok ? S.khi[s] : 0u; skl[i] = ok ? S.klo[s] : 0u;
                sk2h[i] = ok ? S.k2hi[s] : 0u; sk2l[i] = ok ? S.k2lo[s] : 0u;
                sbx[i] = ok ? S.bx[s] : 0.f; sby[i] = ok ? S.by[s] : 0.f; sbz[i] = ok ? S.bz[s] : 0.f;
            }
            const int kmax = min(KMAX, sc.m - j);
            int acc = 0, my_idx = 0;
            float mcx = 0.f, mcy = 0.f, mcz = 0.f;   // lane u: centre u
            unsigned bound_hi = 0u, bound_lo = 0u;
            for (int t = 0; t < kmax; ++t) {
                unsigned bhi = skh[0], blo = skl[0], b2h = sk2h[0], b2l = sk2l[0];
                float bx = sbx[0], by = sby[0], bz = sbz[0];
                int bi = 0;
#pragma unroll
                for (int i = 1; i < NSL; ++i)
                    if (key_gt(skh[i], skl[i], bhi, blo)) { bhi = skh[i]; blo = skl[i]; b2h = sk2h[i]; b2l = sk2l[i]; bx = sbx[i]; by = sby[i]; bz = sbz[i]; bi = i; }
                unsigned whi, wlo;
                const int wl = wave_argmax_key(bhi, blo, whi, wlo);
                if (t > 0 && !key_gt(whi, wlo, bound_hi, bound_lo)) break;          // (a)  (also stops at key 0: nothing left)
                const float x = lane_bcast(bx, wl), y = lane_bcast(by, wl), z = lane_bcast(bz, wl);
                const bool near = lane < t && pdf_f32_ordered(dist_as_written(x, y, z, mcx, mcy, mcz)) < whi;   // (b)
                if (__ballot(near) != 0ull) break;
                const unsigned s2h = (unsigned)__builtin_amdgcn_readlane((int)b2h, wl), s2l = (unsigned)__builtin_amdgcn_readlane((int)b2l, wl);
                if (lane == t) { mcx = x; mcy = y; mcz = z; my_idx = sc.start_n + (int)(~wlo & REL_MASK); }
                acc = t + 1;
                if (key_gt(s2h, s2l, bound_hi, bound_lo)) { bound_hi = s2h; bound_lo = s2l; }
                if (lane == wl) {   // retire the taken super from this lane's slots
#pragma unroll
                    for (int i = 0; i < NSL; ++i) if (i == bi) { skh[i] = 0u; skl[i] = 0u; }
                }
            }
            if (lane < acc) {
                idx[sc.start_m + j + lane] = my_idx;
                cent[3 * lane + 0] = mcx; cent[3 * lane + 1] = mcy; cent[3 * lane + 2] = mcz;
            }
            if (lane == 0) cent_n[0] = acc;
        }
        __syncthreads();  // D: the round's centres are published
        const int a = cent_n[0];
        float cx[KMAX], cy[KMAX], cz[KMAX];
#pragma unroll
        for (int u = 0; u < KMAX; ++u) { cx[u] = cent[3 * u + 0]; cy[u] = cent[3 * u + 1]; cz[u] = cent[3 * u + 2]; }
        j += a;
        ++n_rounds;
        if (j >= sc.m) break;
        FPS_TICK(0);
        const unsigned tag = n_rounds;          // > 0, unique per round
        const int q = (int)(n_rounds & 1u);

        // ---- 2. wave u owns centre u: active supers (private list), active buckets; claims feed the shared lists
        if (wave < a) {
            float mx = cx[0], my = cy[0], mz = cz[0];
#pragma unroll
            for (int u = 1; u < KMAX; ++u) if (u == wave) { mx = cx[u]; my = cy[u]; mz = cz[u]; }
            int n_my = 0;
            for (int s0 = 0; s0 < ns; s0 += 64) {
                const int s = s0 + lane;
                bool act = false;
                if (s < ns) act = pdf_f32_ordered(box_dist_v(S.lox[s], S.loy[s], S.loz[s], S.hix[s], S.hiy[s], S.hiz[s], mx, my, mz)) < S.khi[s];
                const unsigned long long m = __ballot(act);
                if (act) pslist[n_my + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)s;
                n_my += __popcll(m);
                bool fresh = false;
                if (act) fresh = atomicExch(&claimS[s], tag) != tag;
                const unsigned long long mf = __ballot(fresh);
                if (mf) {
                    unsigned base = 0u;
                    if (lane == 0) base = atomicAdd(&cnt[2 + q], (unsigned)__popcll(mf));
                    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                    if (fresh) slist[base + __popcll(mf & ((1ull << lane) - 1ull))] = (unsigned short)s;
                }
            }
            lds_fence();
            for (int g = 0; g < n_my; g += 4) {
                bool bact = false;
                int bk = 0;
                if (g + row < n_my) {
                    bk = (int)pslist[g + row] * SUP + col;
                    if (bk < nb)
                        bact = pdf_f32_ordered(box_dist_v(B.lox[bk], B.loy[bk], B.loz[bk], B.hix[bk], B.hiy[bk], B.hiz[bk], mx, my, mz)) < B.khi[bk];
                }
                bool fresh = false;   // claimB[bk] = mask of the centres that reach the bucket; the first one lists it
                if (bact) fresh = atomicOr(&claimB[bk], 1u << wave) == 0u;
                const unsigned long long mf = __ballot(fresh);
                if (mf) {
                    unsigned base = 0u;
                    if (lane == 0) base = atomicAdd(&cnt[q], (unsigned)__popcll(mf));
                    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                    if (fresh) blist[base + __popcll(mf & ((1ull << lane) - 1ull))] = (unsigned short)bk;
                }
            }
        }
        __syncthreads();  // A: shared lists complete
        FPS_TICK(1);
        const int n_bk = (int)cnt[q], n_sup = (int)cnt[2 + q];
        if (tid == 0) { cnt[q ^ 1] = 0u; cnt[2 + (q ^ 1)] = 0u; }   // next round's counters (nobody touches them before barrier C)
        n_supers += n_sup;
        n_updates += n_bk;

        // ---- 3. update the claimed buckets with every accepted centre.  This phase is bound by the CU's vector issue (16 waves, one
        // scene per CU), and most of a bucket's ~150 instructions were the four 64-lane reductions (best and second-best key) with their DPP
        // wait states.  RANK (round 4): FOUR buckets per pass, one per 16-lane row, four consecutive points per lane -- a lane first
        // reduces its own four points, the rows then need 4-step reductions, for four buckets at once.
        if (RANK) {
            for (int i0 = wave * 4; i0 < n_bk; i0 += NW * 4) {   // (two such sets in flight per wave: measured slower, 22.9 vs 21.7 ms at level 1)
                const int bk = i0 + row < n_bk ? (int)blist[i0 + row] : -1;
                const bool live = bk >= 0;
                const size_t pos = (size_t)sc.pbase + (size_t)(live ? bk : 0) * BSZ + (size_t)col * 4;   // (idle rows read bucket 0, write nothing)
                float4 p[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) p[e] = pts[pos + e];
                const uint4 kb4 = *reinterpret_cast<const uint4 *>(kbs + pos);
                const unsigned klo[4] = {live ? kb4.x : 0u, live ? kb4.y : 0u, live ? kb4.z : 0u, live ? kb4.w : 0u};
                unsigned rem = live ? claimB[bk] : 0u;            // centres that reach the row's bucket
                float w[4] = {p[0].w, p[1].w, p[2].w, p[3].w};
                while (__ballot(rem != 0u) != 0ull) {             // every row takes ITS next centre (padding points carry w < 0: never lowered)
                    const bool on = rem != 0u;
                    const int v = on ? __ffs(rem) - 1 : 0;
                    rem &= rem - 1u;
                    const float ux = cent[3 * v + 0], uy = cent[3 * v + 1], uz = cent[3 * v + 2];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float d = dist_as_written(p[e].x, p[e].y, p[e].z, ux, uy, uz);
                        if (on && d < w[e]) w[e] = d;
                    }
                }
                unsigned long long key[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (live && w[e] < p[e].w) reinterpret_cast<float *>(pts + pos + e)[3] = w[e];
                    key[e] = ((unsigned long long)(live && p[e].w >= 0.f ? pdf_f32_ordered(w[e]) : 0u) << 32) | klo[e];
                }
                // the lane's best and second-best point
                unsigned long long best = key[0], second = 0ull;
                float bx = p[0].x, by = p[0].y, bz = p[0].z;
#pragma unroll
                for (int e = 1; e < 4; ++e) {
                    const bool gt = key[e] > best;
                    const unsigned long long lower = gt ? best : key[e];
                    second = lower > second ? lower : second;
                    best = gt ? key[e] : best;
                    bx = gt ? p[e].x : bx; by = gt ? p[e].y : by; bz = gt ? p[e].z : bz;
                }
                // the row's best point, and the best of the rest (the winning lane contributes its second-best)
                const unsigned bh = (unsigned)(best >> 32), bl = (unsigned)best;
                const unsigned mh = row_umax_bcast(bh, row);
                const unsigned ml = row_umax_bcast(bh == mh ? bl : 0u, row);
                const unsigned long long wm = __ballot(bh == mh && bl == ml);
                const bool iam = col == __ffs((unsigned)(wm >> (16 * row)) & 0xffffu) - 1;
                const unsigned ch = iam ? (unsigned)(second >> 32) : bh, cl = iam ? (unsigned)second : bl;
                const unsigned m2h = row_umax_bcast(ch, row);
                const unsigned m2l = row_umax_bcast(ch == m2h ? cl : 0u, row);
                if (iam && live) {
                    claimB[bk] = 0u;
                    B.khi[bk] = mh; B.klo[bk] = ml;
                    B.k2hi[bk] = m2h; B.k2lo[bk] = m2l;
                    B.bx[bk] = bx; B.by[bk] = by; B.bz[bk] = bz;
                }
            }
        } else
        for (int i0 = wave * UNR; i0 < n_bk; i0 += NW * UNR) {
            float4 p[UNR];
            unsigned kb[UNR];
            int bkid[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                bkid[u] = i0 + u < n_bk ? (int)blist[i0 + u] : -1;
                if (bkid[u] >= 0) {
                    const size_t pos = (size_t)sc.pbase + (size_t)bkid[u] * BSZ + lane;
                    p[u] = pts[pos];
                    kb[u] = kbs[pos];
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {   // (no early exit: the UNR reduction chains are independent and interleave)
                const bool live = bkid[u] >= 0;
                const unsigned msk = live ? (unsigned)__builtin_amdgcn_readfirstlane((int)claimB[bkid[u]]) : 0u;   // centres that reach it
                unsigned khi = 0u;
                if (live && p[u].w >= 0.f) {
                    float w = p[u].w;
#pragma unroll
                    for (int v = 0; v < KMAX; ++v)
                        if ((msk >> v) & 1u) {
                            const float d = dist_as_written(p[u].x, p[u].y, p[u].z, cx[v], cy[v], cz[v]);
                            if (d < w) w = d;
                        }
                    if (w < p[u].w) pts[(size_t)sc.pbase + (size_t)bkid[u] * BSZ + lane].w = w;
                    khi = pdf_f32_ordered(w);
                }
                const unsigned klo = live ? kb[u] : 0u;
                unsigned mhi, mlo;
                const int kl = wave_argmax_key(khi, klo, mhi, mlo);
                const unsigned h2 = lane == kl ? 0u : khi, l2 = lane == kl ? 0u : klo;
                const unsigned m2hi = wave_umax(h2);
                const unsigned m2lo = wave_umax(h2 == m2hi ? l2 : 0u);
                const float nx = lane_bcast(live ? p[u].x : 0.f, kl & 63), ny = lane_bcast(live ? p[u].y : 0.f, kl & 63), nz = lane_bcast(live ? p[u].z : 0.f, kl & 63);
                if (lane == 0 && live) {
                    claimB[bkid[u]] = 0u;
                    B.khi[bkid[u]] = mhi; B.klo[bkid[u]] = mlo;
                    B.k2hi[bkid[u]] = m2hi; B.k2lo[bkid[u]] = m2lo;
                    B.bx[bkid[u]] = nx; B.by[bkid[u]] = ny; B.bz[bkid[u]] = nz;
                }
            }
        }
        __syncthreads();  // B: bucket records + tmp stores visible
        FPS_TICK(2);

        // ---- 4. refresh the claimed supers (dealt to the waves, 4 supers per pass)
        for (int g = 4 * wave; g < n_sup; g += 4 * NW) {
            unsigned h16 = 0u, l16 = 0u, h2 = 0u, l2 = 0u;
            int sup = -1, bk = 0;
            if (g + row < n_sup) {
                sup = (int)slist[g + row];
                bk = sup * SUP + col;
                if (bk < nb) { h16 = B.khi[bk]; l16 = B.klo[bk]; h2 = B.k2hi[bk]; l2 = B.k2lo[bk]; }
            }
            const unsigned mh = row_umax_bcast(h16, row);
            const unsigned ml = row_umax_bcast(h16 == mh ? l16 : 0u, row);
            const unsigned long long m = __ballot(sup >= 0 && h16 == mh && l16 == ml);
            const int wcol = __ffs((unsigned)(m >> (16 * row)) & 0xffffu) - 1;
            const unsigned sh = col == wcol ? h2 : h16, sl = col == wcol ? l2 : l16;
            const unsigned m2h = row_umax_bcast(sh, row);
            const unsigned m2l = row_umax_bcast(sh == m2h ? sl : 0u, row);
            if (col == 0 && sup >= 0) {
                const int wb = sup * SUP + wcol;
                S.khi[sup] = mh; S.klo[sup] = ml;
                S.k2hi[sup] = m2h; S.k2lo[sup] = m2l;
                S.bx[sup] = B.bx[wb]; S.by[sup] = B.by[wb]; S.bz[sup] = B.bz[wb];
            }
        }
        __syncthreads();  // C: super records complete before the next round's candidates
        FPS_TICK(3);
    }
    if (tid == 0 && stats) {
        stats[blockIdx.x * 4 + 0] = n_updates;
        stats[blockIdx.x * 4 + 1] = n_supers;
        stats[blockIdx.x * 4 + 2] = (unsigned)sc.m;
        stats[blockIdx.x * 4 + 3] = n_rounds;
#ifdef FPS_PROFILE
        for (int i = 0; i < 4; ++i) stats[blockIdx.x * 4 + i] = (unsigned)(tc[i] >> 10);   // kilo-cycles: candidates | lists | updates | refresh
#endif
    }
#undef FPS_TICK
}

}  // namespace

extern "C" long pdf_fps_workspace_bytes(int b, int n_total) {
    if (b < 1 || n_total < 0) return -1;
    return (long)make_layout(b, n_total).total;
}

// byte offset of the per-scene work counters (4 x u32: bucket updates, super visits, samples, buckets) in the workspace
extern "C" long pdf_fps_stats_offset(int b, int n_total) {
    if (b < 1 || n_total < 0) return -1;
    return (long)make_layout(b, n_total).stats;
}

extern "C" int pdf_farthest_point_sampling_bucketed(int b, int n, int n_total, const float *xyz, const int *offset,
                                                    const int *new_offset, void *workspace, long workspace_bytes,
                                                    int *idx, void *stream) {
    if (b < 1 || n < 1 || n_total < 1 || !xyz || !offset || !new_offset || !workspace || !idx) return PDF_ERR_BAD_ARG;
    const Layout L = make_layout(b, n_total);
    if (workspace_bytes < (long)L.total) return PDF_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace);
    const int nb_cap = (n + BSZ - 1) / BSZ;
    if (nb_cap > NB_MAX || n > (1 << REL_BITS)) {
        // scene too large for the LDS-resident bucket table: exact plain kernel (tmp scratch carved from the workspace)
        return pdf_farthest_point_sampling(b, n, xyz, offset, new_offset, reinterpret_cast<float *>(ws + L.pts), idx, stream);
    }
    float *bbox = reinterpret_cast<float *>(ws + L.bbox);
    unsigned *hist = reinterpret_cast<unsigned *>(ws + L.hist);
    unsigned *cursor = reinterpret_cast<unsigned *>(ws + L.cursor);
    unsigned *cell = reinterpret_cast<unsigned *>(ws + L.cell);
    float4 *pts = reinterpret_cast<float4 *>(ws + L.pts);
    unsigned *kb = reinterpret_cast<unsigned *>(ws + L.kb);
    float *meta = reinterpret_cast<float *>(ws + L.meta);
    hipError_t e = hipMemsetAsync(ws + L.hist, 0, L.cell - L.hist, s);  // hist + cursor
    if (e != hipSuccess) return (int)e;
    const int lg = pdf_fps_reference_block_log2(n);
    k_bbox<<<b, PB, 0, s>>>(xyz, offset, bbox);
    k_hist<<<pdf_divup(n_total, PB), PB, 0, s>>>(n_total, b, xyz, offset, bbox, hist, cell);
    k_scan<<<b, 1024, 0, s>>>(hist);
    k_scatter<<<pdf_divup(n_total, PB), PB, 0, s>>>(n_total, b, xyz, offset, new_offset, hist, cursor, cell, pts, kb, lg);
    k_pad<<<b, BSZ, 0, s>>>(b, offset, new_offset, pts, kb);
    // the exact bucket count lives on the device (offsets): launch the upper bound sum_b ceil(n_b/64) <= npad/64;
    // surplus waves summarise unused slots that no scene ever reads
    k_meta<<<L.npad / BSZ, 64, 0, s>>>(pts, kb, meta);
    const int ns_cap = (nb_cap + SUP - 1) / SUP;
    const size_t lds = (size_t)(11 * nb_cap + 11 * ns_cap) * 4 + (size_t)(2 * (((ns_cap + 63) & ~63) + 64) + nb_cap + 64) * 2;
    const char *env_nw = getenv("PDFOPS_FPS_NW");  // tuning knob (waves per scene): 1, 2 or 4
    const int nw = env_nw ? atoi(env_nw) : 4;
    unsigned *stats = reinterpret_cast<unsigned *>(ws + L.stats);
    // several samples per round (k_fps_multi) when its larger records fit the 160 KB of LDS (scenes up to ~180k points)
    const size_t lds2 = (size_t)(REC2 * nb_cap + REC2 * ns_cap) * 4 + (size_t)(2 * (((ns_cap + 63) & ~63) + 64) + nb_cap + 64) * 2;
    const char *env_k = getenv("PDFOPS_FPS_K");    // tuning knob (samples per round): 1 = the one-sample kernel, 4 or 8
    const int kmulti = env_k ? atoi(env_k) : (n >= FPS_MULTI_MIN ? 8 : 1);
    // one centre per wave (k_fps_mw<NW>): records + claim tags + shared / private lists
    const char *env_mw = getenv("PDFOPS_FPS_MW");   // tuning knob: 0 = k_fps_multi, 8 / 16 = waves (= centres per round) of k_fps_mw
    // measured (2 scenes, levels 100k / 25k / 6250 / 1562 points): 16 waves 33.0 / 10.0 / 3.9 / 1.7 ms, 8 waves 41.3 / 10.8 / 3.6 / 1.5 ms,
    // the one-sample kernel 91 / 20 / 4.9 / 1.2 ms -> by the size of the largest scene
    const int mw = env_mw ? atoi(env_mw) : (n >= 16384 ? 16 : n >= FPS_MULTI_MIN ? 8 : 0);
    const int mww = mw >= 16 ? 16 : 8;
    const size_t lds3 = (size_t)(REC2 * nb_cap + REC2 * ns_cap + nb_cap + ns_cap + 4 + 3 * 16 + 4 + (nb_cap <= BKC_MAX ? nb_cap : ns_cap) + 3 * 16 + 16 * SUP) * 4 +
                        (size_t)(nb_cap + 64 + ns_cap + 64 + mww * (((ns_cap + 63) & ~63) + 64)) * 2;
    if (kmulti > 1 && mw != 0 && lds3 <= 160 * 1024) {
#define PDF_LAUNCH_FPS_MW(NW_, RANK_)                                                                                     \
    do {                                                                                                                 \
        if (lds3 > 64 * 1024) {                                                                                          \
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_fps_mw<NW_, RANK_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3); \
            if (e != hipSuccess) return (int)e;                                                                          \
        }                                                                                                                \
        k_fps_mw<NW_, RANK_><<<b, 64 * NW_, lds3, s>>>(offset, new_offset, pts, kb, meta, idx, nb_cap, stats, two_lvl);   \
    } while (0)
        static const bool rank_sel = [] { const char *v = getenv("PDFOPS_FPS_RANK"); return !(v && v[0] == '0'); }();   // 0: rounds 2-3 candidate loop (A/B)
        static const int two_lvl = [] { const char *v = getenv("PDFOPS_FPS_TWO_LEVEL"); return (v && v[0] == '0') ? 0 : 1; }();   // 0: super-bucket candidates at level 1 (A/B)
        if (rank_sel) { if (mww == 16) PDF_LAUNCH_FPS_MW(16, true); else PDF_LAUNCH_FPS_MW(8, true); }
        else { if (mww == 16) PDF_LAUNCH_FPS_MW(16, false); else PDF_LAUNCH_FPS_MW(8, false); }
#undef PDF_LAUNCH_FPS_MW
        return pdf_launch_status();
    }
    if (kmulti > 1 && lds2 <= 160 * 1024) {
#define PDF_LAUNCH_FPS_MULTI(NW_, K_)                                                                                    \
    do {                                                                                                                 \
        if (lds2 > 64 * 1024) {                                                                                          \
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_fps_multi<NW_, K_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2); \
            if (e != hipSuccess) return (int)e;                                                                          \
        }                                                                                                                \
        k_fps_multi<NW_, K_><<<b, 64 * NW_, lds2, s>>>(offset, new_offset, pts, kb, meta, idx, nb_cap, stats);            \
    } while (0)
        if (kmulti <= 4) { if (nw == 8) PDF_LAUNCH_FPS_MULTI(8, 4); else PDF_LAUNCH_FPS_MULTI(4, 4); }
        else { if (nw == 8) PDF_LAUNCH_FPS_MULTI(8, 8); else PDF_LAUNCH_FPS_MULTI(4, 8); }
#undef PDF_LAUNCH_FPS_MULTI
        return pdf_launch_status();
    }
#define PDF_LAUNCH_FPS(NW_)                                                                                              \
    do {                                                                                                                 \
        if (lds > 64 * 1024) {                                                                                           \
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_fps<NW_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return (int)e;                                                                          \
        }                                                                                                                \
        k_fps<NW_><<<b, 64 * NW_, lds, s>>>(offset, new_offset, pts, kb, meta, idx, nb_cap, stats);                       \
    } while (0)
    if (nw == 1) PDF_LAUNCH_FPS(1);
    else if (nw == 2) PDF_LAUNCH_FPS(2);
    else PDF_LAUNCH_FPS(4);
#undef PDF_LAUNCH_FPS
    return pdf_launch_status();
}
