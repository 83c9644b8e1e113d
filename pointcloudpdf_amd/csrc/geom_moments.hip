// Per-scene moments of the relative coordinates of a neighbour table -- the feature-independent half of a BatchNorm.
//
// The PointTransformerLayer's geometry branch starts with Linear(3, 3) on rel = p[idx[i, j]] - p[i] followed by a train-mode
// BatchNorm over all (point, neighbour) rows (point_transformer_seg.py:27-29, 52-57).  Its batch statistics are a closed form of the
// weights and of NINE sums that depend on the coordinates and the kNN table only:
//     sum t1[a]   = W[a,:] . S + rows b[a]                      S = sum rel        (3)
//     sum t1[a]^2 = W[a,:] M W[a,:]^T + 2 b[a] W[a,:] . S + rows b[a]^2       M = sum rel rel^T  (6 distinct)
// so the geometry pre-pass computes S and M once per table (here, per scene, so that a pre-pass over several batches can hand every
// batch the sums of its own scenes), and the layer's first statistics pass + its finalizer (2 of the 7 forward launches of a layer) are
// replaced by ~60 flops in the prologue of the next pass (fused_layer.h, bnp_from_moments).
// rel is formed in fp32 exactly as the layer kernels form it (0 for idx < 0 rows, which the reference keeps as rows); sums in fp64.
// out (b, 9) double = [Sx Sy Sz | Mxx Mxy Mxz Myy Myz Mzz], written.  Bound: HBM (idx + gathered coordinates), ~10 us.
#include "pdfops_common.h"

namespace {

constexpr int MB = 256;

__device__ __forceinline__ int scene_of(const int *__restrict__ offset, int b, long i) {   // first s with offset[s] > i
    int lo = 0, hi = b - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((long)offset[mid] > i) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// Workgroup g -> (scene, chunk of MB rows inside the scene): every workgroup lies inside ONE scene, so its nine sums go to its own slot
// and the per-scene totals are formed in a fixed order by k_rel_moments_sum (round 3: no atomics -- the statistics are bit-reproducible).
__device__ __forceinline__ bool block_rows(const int *__restrict__ offset, int b, int g, int *scene, long *first, long *last, int *g0, int *g1) {
    int cum = 0;
    long prev = 0;
    for (int s = 0; s < b; ++s) {
        const long end = offset[s];
        const int nb = (int)((end - prev + MB - 1) / MB);
        if (g < cum + nb) {
            *scene = s; *first = prev + (long)(g - cum) * MB; *last = min(*first + MB, end); *g0 = cum; *g1 = cum + nb;
            return true;
        }
        cum += nb;
        prev = end;
    }
    return false;
}

// p = source points (indexed by idx), q = query points (row i of idx belongs to q[i]; q == p for a self table)
__global__ __launch_bounds__(MB) void k_rel_moments(long n, int k, const float *__restrict__ p, const float *__restrict__ q, const int *__restrict__ idx,
                                                    const int *__restrict__ offset, int b, double *__restrict__ part) {
    __shared__ double red[MB / 64][9];
    int scene, g0, g1;
    long first, last;
    const bool live = block_rows(offset, b, (int)blockIdx.x, &scene, &first, &last, &g0, &g1);
    const long i = first + threadIdx.x;
    double acc[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) acc[e] = 0.0;
    if (live && i < last) {
        const float px = q[i * 3], py = q[i * 3 + 1], pz = q[i * 3 + 2];
        for (int j = 0; j < k; ++j) {
            const int nb = idx[i * k + j];
            const long nc = nb >= 0 ? nb : 0;
            const float qx = p[nc * 3], qy = p[nc * 3 + 1], qz = p[nc * 3 + 2];
            const float rx = nb >= 0 ? qx - px : 0.f, ry = nb >= 0 ? qy - py : 0.f, rz = nb >= 0 ? qz - pz : 0.f;
            const double x = rx, y = ry, z = rz;
            acc[0] += x; acc[1] += y; acc[2] += z;
            acc[3] += x * x; acc[4] += x * y; acc[5] += x * z; acc[6] += y * y; acc[7] += y * z; acc[8] += z * z;
        }
    }
#pragma unroll
    for (int e = 0; e < 9; ++e) {
        double v = acc[e];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][e] = v;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        double v = 0.0;
        for (int w = 0; w < MB / 64; ++w) v += red[w][threadIdx.x];
        part[(size_t)blockIdx.x * 9 + threadIdx.x] = v;   // (workgroups beyond the last scene store zeros)
    }
}

// out[s][e] = sum of the slots of scene s in workgroup order.  One workgroup per scene; lane l adds slots l, l + MB, ...
__global__ __launch_bounds__(MB) void k_rel_moments_sum(const int *__restrict__ offset, int b, const double *__restrict__ part, double *__restrict__ out) {
    __shared__ double red[MB][9];
    const int s = blockIdx.x;
    int g0 = 0;
    long prev = 0;
    for (int t = 0; t < s; ++t) { g0 += (int)((offset[t] - prev + MB - 1) / MB); prev = offset[t]; }
    const int g1 = g0 + (int)((offset[s] - prev + MB - 1) / MB);
    double acc[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) acc[e] = 0.0;
    for (int g = g0 + threadIdx.x; g < g1; g += MB)
#pragma unroll
        for (int e = 0; e < 9; ++e) acc[e] += part[(size_t)g * 9 + e];
#pragma unroll
    for (int e = 0; e < 9; ++e) red[threadIdx.x][e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 9) {
        double v = 0.0;
        for (int t = 0; t < MB; ++t) v += red[t][threadIdx.x];
        out[(size_t)s * 9 + threadIdx.x] = v;
    }
}

}  // namespace

// workspace (doubles) of the two entry points below: one slot of nine sums per workgroup
extern "C" long pdf_knn_rel_moments_ws_doubles(int b, long m) { return ((m + MB - 1) / MB + (b > 0 ? b : 0)) * 9; }

// The same for a table whose m queries (new_xyz, scene ends new_offset) differ from its source points (TransitionDown's grouping):
// rel = xyz[idx[i, j]] - new_xyz[i].  out (b, 9) is WRITTEN (no zeroing); ws: pdf_knn_rel_moments_ws_doubles(b, m) doubles.
extern "C" int pdf_knn_rel_moments_q(int b, long m, int nsample, const float *xyz, const float *new_xyz, const int *new_offset, const int *idx,
                                     double *out, double *ws, void *stream) {
    if (b < 1 || m < 0 || nsample < 1 || !new_offset || !out || (m > 0 && (!xyz || !new_xyz || !idx || !ws))) return PDF_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (m == 0) {
        hipError_t e = hipMemsetAsync(out, 0, sizeof(double) * 9 * (size_t)b, s);
        return e == hipSuccess ? PDF_OK : (int)e;
    }
    const unsigned grid = (unsigned)((m + MB - 1) / MB + b);   // every scene rounds its rows up to whole workgroups
    k_rel_moments<<<grid, MB, 0, s>>>(m, nsample, xyz, new_xyz, idx, new_offset, b, ws);
    k_rel_moments_sum<<<(unsigned)b, MB, 0, s>>>(new_offset, b, ws, out);
    return pdf_launch_status();
}
extern "C" int pdf_knn_rel_moments(int b, long n, int nsample, const float *xyz, const int *offset, const int *idx, double *out, double *ws,
                                   void *stream) {
    return pdf_knn_rel_moments_q(b, n, nsample, xyz, xyz, offset, idx, out, ws, stream);
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// Morton keys of a batch's points, scene by scene: key = scene << 30 | 30-bit Morton code of the point on a 1024^3 grid over ITS SCENE's
// bounding box.  A stable sort of the keys is the VISITING order of the forward gathers / layer passes (geometry.Geometry.order; nothing
// is stored in that order).  Per-scene boxes make a scene's order a function of that scene alone: a batch gets the same order -- and with
// it the same rounding of every per-workgroup partial sum -- whether its pre-pass ran alone or inside a group of batches.
namespace {
__global__ __launch_bounds__(MB) void k_scene_bounds(const float *__restrict__ xyz, const int *__restrict__ offset, float *__restrict__ bounds) {
    __shared__ float red[6][MB / 64];
    const int s = blockIdx.x;
    const long start = s == 0 ? 0 : offset[s - 1], end = offset[s];
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (long i = start + threadIdx.x; i < end; i += MB)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = xyz[3 * i + a];
            lo[a] = fminf(lo[a], v);
            hi[a] = fmaxf(hi[a], v);
        }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64)); }
        if ((threadIdx.x & 63) == 0) { red[a][threadIdx.x >> 6] = lo[a]; red[3 + a][threadIdx.x >> 6] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float l[3], ext = 0.f;
        for (int a = 0; a < 3; ++a) {
            float h = red[3 + a][0];
            l[a] = red[a][0];
            for (int w = 1; w < MB / 64; ++w) { l[a] = fminf(l[a], red[a][w]); h = fmaxf(h, red[3 + a][w]); }
            ext = fmaxf(ext, end > start ? h - l[a] : 0.f);
        }
        bounds[4 * s + 0] = l[0]; bounds[4 * s + 1] = l[1]; bounds[4 * s + 2] = l[2];
        bounds[4 * s + 3] = fmaxf(ext / 1023.0f, 1e-9f);   // cell edge
    }
}
__device__ __forceinline__ unsigned spread10(unsigned v) {   // 10 bits -> every third bit
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    return (v | (v << 2)) & 0x09249249u;
}
__global__ __launch_bounds__(MB) void k_scene_morton(long n, int b, const float *__restrict__ xyz, const int *__restrict__ offset,
                                                     const float *__restrict__ bounds, long long *__restrict__ keys) {
    const long i = (long)blockIdx.x * MB + threadIdx.x;
    if (i >= n) return;
    const int s = scene_of(offset, b, i);
    const float cell = bounds[4 * s + 3];
    unsigned q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int v = (int)((xyz[3 * i + a] - bounds[4 * s + a]) / cell);
        q[a] = (unsigned)(v < 0 ? 0 : (v > 1023 ? 1023 : v));
    }
    keys[i] = ((long long)s << 30) | (long long)(spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2));
}
}  // namespace

// keys (n) int64, bounds (4 b floats of scratch: per scene lo x / y / z and the cell edge), both written.
extern "C" int pdf_scene_morton_keys(long n, int b, const float *xyz, const int *offset, float *bounds, long long *keys, void *stream) {
    if (n < 0 || b < 1 || !offset || !bounds || (n > 0 && (!xyz || !keys))) return PDF_ERR_BAD_ARG;
    if (n == 0) return PDF_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    k_scene_bounds<<<(unsigned)b, MB, 0, s>>>(xyz, offset, bounds);
    k_scene_morton<<<(unsigned)((n + MB - 1) / MB), MB, 0, s>>>(n, b, xyz, offset, bounds, keys);
    return pdf_launch_status();
}
