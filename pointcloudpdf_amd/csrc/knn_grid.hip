// Grid-accelerated exact kNN for gfx950.
//
// Result contract: bit-identical to pdf_knn_query / the reference scan (knn_query_cuda_kernel.cu:60-104).
// The reference's brute force is O(M * N_scene); its result is "the nsample smallest as-written fp32 distances in
// ascending order", with ties resolved by heap history.  This path finds the same nsample (+1) smallest distances
// through a uniform grid over the scene and hands every query whose answer could depend on tie-breaking (two equal
// distances among the best nsample+1) -- or whose scene holds fewer than nsample+1 points -- to the exact scan kernel.
// With real-valued coordinates that is ~never; with grid-snapped coordinates it is every query (still exact).
//
//   k_grid_setup : per scene bounding box -> cell size from volume / count (target 4 points per cell), dims <= 2^20 cells
//   k_grid_hist / k_grid_scan / k_grid_scatter : counting sort of the source points by cell -> float4 {x,y,z,idx}
//   k_grid_query : one lane per query, rings of cells around the query's cell until the (nsample+1)-th best distance
//                  is certified by the distance to the searched cube's faces; top list in registers (static insertion)
//   knn_scan_kernel (knn_query.hip) on the redo list
// Work per query ~ a few hundred candidate distances instead of N_scene.  Bound: latency / VALU; HBM bytes 12N+12M+8Mk.
#include "pdfops_common.h"
#include <cstdio>
#include <cstdlib>

extern "C" int pdf_knn_query_list(int m, int nsample, const float *xyz, const float *new_xyz, const int *offset,
                                  const int *new_offset, int b, int *idx, float *dist2, const int *qlist,
                                  const int *qcount, void *stream);

namespace kg {

constexpr int CAP_CELLS = 1 << 20;  // cells per scene
constexpr int PB = 256;

struct SceneGrid {      // 16 floats / ints per scene in the workspace
    float minx, miny, minz, inv_h;
    float h;
    int nx, ny, nz;
    int start, n;       // source point range
    int cell_base;      // offset of this scene's cells in the cell arrays
    int pad[5];
};

struct Layout {
    size_t grid, cell_start, cursor, cell_of, sorted, redo, total;
};
__host__ __device__ inline size_t al(size_t v) { return (v + 255) & ~(size_t)255; }
__host__ __device__ inline Layout make_layout(int b, int n, int m) {
    Layout L;
    size_t o = 0;
    L.grid = o;       o = al(o + (size_t)b * sizeof(SceneGrid));
    L.cell_start = o; o = al(o + ((size_t)b * CAP_CELLS + 1) * 4);
    L.cursor = o;     o = al(o + (size_t)b * CAP_CELLS * 4);
    L.cell_of = o;    o = al(o + (size_t)n * 4);
    L.sorted = o;     o = al(o + (size_t)n * 16);
    L.redo = o;       o = al(o + ((size_t)m + 4) * 4);
    L.total = o;
    return L;
}

__global__ __launch_bounds__(PB) void k_grid_setup(const float *__restrict__ xyz, const int *__restrict__ offset, SceneGrid *__restrict__ grids, float ppc,
                                                   float min_cell) {
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
        if ((threadIdx.x & 63) == 0) { red[a][threadIdx.x >> 6] = lo[a]; red[3 + a][threadIdx.x >> 6] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float l[3], h[3];
        for (int a = 0; a < 3; ++a) {
            l[a] = red[a][0]; h[a] = red[3 + a][0];
            for (int w = 1; w < PB / 64; ++w) { l[a] = fminf(l[a], red[a][w]); h[a] = fmaxf(h[a], red[3 + a][w]); }
        }
        const int n = end - start;
        float ext[3];
        for (int a = 0; a < 3; ++a) ext[a] = n > 0 ? fmaxf(h[a] - l[a], 1e-6f) : 1.f;
        // target ~ppc points per cell if the points filled the box; never more than CAP_CELLS cells
        float cell = cbrtf(ext[0] * ext[1] * ext[2] * ppc / (float)(n > 0 ? n : 1));
        const float longest = fmaxf(ext[0], fmaxf(ext[1], ext[2]));
        cell = fmaxf(fmaxf(cell, longest / 1000.f), min_cell);   // (radius queries ask for cell >= radius: 27 cells cover the ball)
        int nx, ny, nz;
        while (true) {
            nx = (int)(ext[0] / cell) + 1; ny = (int)(ext[1] / cell) + 1; nz = (int)(ext[2] / cell) + 1;
            if ((long)nx * ny * nz < CAP_CELLS) break;  // strict: slot [ncell] holds the end sentinel
            cell *= 1.26f;
        }
        SceneGrid g;
        g.minx = l[0]; g.miny = l[1]; g.minz = l[2];
        g.h = cell; g.inv_h = 1.0f / cell;
        g.nx = nx; g.ny = ny; g.nz = nz;
        g.start = start; g.n = n;
        g.cell_base = s * CAP_CELLS;
        grids[s] = g;
    }
}

__device__ __forceinline__ int cell_coord(float v, float lo, float inv_h, int n) {
    int c = (int)floorf((v - lo) * inv_h);
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

__device__ __forceinline__ int scene_of(int i, const int *__restrict__ offset, int b) {
    int s = 0;
    while (s < b - 1 && i >= offset[s]) ++s;
    return s;
}

__global__ __launch_bounds__(PB) void k_grid_hist(int n, int b, const float *__restrict__ xyz, const int *__restrict__ offset,
                                                  const SceneGrid *__restrict__ grids, unsigned *__restrict__ count, int *__restrict__ cell_of) {
    const int i = blockIdx.x * PB + threadIdx.x;
    if (i >= n) return;
    const int s = scene_of(i, offset, b);
    const SceneGrid g = grids[s];
    const int cx = cell_coord(xyz[3 * (size_t)i], g.minx, g.inv_h, g.nx);
    const int cy = cell_coord(xyz[3 * (size_t)i + 1], g.miny, g.inv_h, g.ny);
    const int cz = cell_coord(xyz[3 * (size_t)i + 2], g.minz, g.inv_h, g.nz);
    const int c = g.cell_base + (cz * g.ny + cy) * g.nx + cx;
    cell_of[i] = c;
    atomicAdd(&count[c], 1u);
}

// exclusive scan of one scene's cell counts (in place) + the scene's point base; one 1024-thread block per scene
__global__ __launch_bounds__(1024) void k_grid_scan(const SceneGrid *__restrict__ grids, unsigned *__restrict__ cells) {
    __shared__ unsigned wsum[16];
    __shared__ unsigned carry;
    const SceneGrid g = grids[blockIdx.x];
    const int ncell = g.nx * g.ny * g.nz;
    unsigned *h = cells + g.cell_base;
    const int t = threadIdx.x;
    if (t == 0) carry = (unsigned)g.start;
    __syncthreads();
    for (int base = 0; base < ncell; base += 1024 * 8) {
        unsigned v[8], sum = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = base + t * 8 + u;
            v[u] = c < ncell ? h[c] : 0u;
            sum += v[u];
        }
        unsigned inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned up = __shfl_up(inc, o, 64);
            if ((t & 63) >= o) inc += up;
        }
        if ((t & 63) == 63) wsum[t >> 6] = inc;
        __syncthreads();
        unsigned wbase = 0;
        for (int w = 0; w < (t >> 6); ++w) wbase += wsum[w];
        unsigned run = carry + wbase + inc - sum;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = base + t * 8 + u;
            if (c < ncell) h[c] = run;
            run += v[u];
        }
        __syncthreads();
        if (t == 1023) carry = run;
        __syncthreads();
    }
    if (t == 0) h[ncell] = carry;  // end sentinel (next scene's base or the slack slot)
}

__global__ __launch_bounds__(PB) void k_grid_scatter(int n, const float *__restrict__ xyz, const int *__restrict__ cell_of,
                                                     const unsigned *__restrict__ cell_start, unsigned *__restrict__ cursor,
                                                     float4 *__restrict__ sorted) {
    const int i = blockIdx.x * PB + threadIdx.x;
    if (i >= n) return;
    const int c = cell_of[i];
    const unsigned pos = cell_start[c] + atomicAdd(&cursor[c], 1u);
    sorted[pos] = make_float4(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], __int_as_float(i));
}

// sorted insertion into a (d, i) list of KP1 entries kept in registers (static indices only).  Round 4: ordered by DISTANCE alone, branch-free
// -- new d[s] = med3(d[s-1], d, d[s]) (the list is ascending), the index follows the two comparisons d < d[s-1], d < d[s]: 4 vector
// operations per slot instead of ~8 with the (distance, index) order and its early exits (the query kernels are bound by the vector issue of
// exactly this routine).  Equal distances may now sit in either order -- but a query with two equal distances among its K+1 best goes to the
// exact scan anyway (need_redo), and an equal distance that stays OUTSIDE the list changes neither the K results nor that test.
template <int KP1>
__device__ __forceinline__ void insert(float (&bd)[KP1], int (&bi)[KP1], float d, int i) {
    if (!(d < bd[KP1 - 1])) return;
    bool below = d < bd[0];          // d < (old) bd[s]
    float prev = bd[0];              // old bd[s - 1]
    int previ = bi[0];
    bi[0] = below ? i : bi[0];
    bd[0] = fminf(d, bd[0]);
#pragma unroll
    for (int s = 1; s < KP1; ++s) {
        const float cur = bd[s];
        const int curi = bi[s];
        const bool here = d < cur;
        bd[s] = __builtin_amdgcn_fmed3f(prev, d, cur);
        bi[s] = below ? previ : (here ? i : curi);
        below = here; prev = cur; previ = curi;
    }
}

// SELF: the queries ARE the source points (self kNN of a level: 5 of the 13 tables, most of the query work).  Thread t then
// takes the t-th point of the cell-sorted copy instead of the t-th point in memory order: the 64 lanes of a wave sit in the
// same or adjacent cells, walk the same rings and read the same cell lists (coherent loads, little divergence).
// COUNT (measurement builds of the launch, pdf_knn_query_ws_counted): every lane counts the candidate distances it evaluates; the wave's
// total goes to *pairs with one atomic per wave -- the work the kernel really does, as opposed to the m * n_scene pairs of the brute force
// it replaces (bench.py / tools/ops_roofline.py price the kernel against the fp32 vector peak with THIS count).
//
// Round 4: the walk is organised for memory-level parallelism.  Rounds 1-3 visited the shell of radius R cell by cell -- two dependent
// loads (the cell's range) and then one dependent load per point, ~27-125 cells per query, most of them empty on surface-like scenes: a
// wave spent ~400 us in ~800 serialised L2 round trips (0.2 % of the vector peak on the pairs it evaluated).  Now
//   * the shell is walked ROW by row (dz, dy in -R..R: wave-uniform trip counts, out-of-grid rows are empty ranges): the cells of a row
//     on a z- or y-face are contiguous in the sorted copy -- ONE range [start(x0), start(x1 + 1)) for up to 2R+1 cells -- and an interior
//     row contributes its two end cells;
//   * the next row's ranges are requested before the current row's points are scanned;
//   * points are loaded four at a time (clamped addresses, guarded inserts).
// The candidates and the (distance, index)-ordered list are what they were: results unchanged bit for bit.
struct RowRanges { unsigned a0, a1, b0, b1; };   // [a0, a1) and [b0, b1) of the sorted copy

template <int KP1, bool COUNT>
__device__ __forceinline__ void scan_range(unsigned p0, unsigned p1, const float4 *__restrict__ sorted, float qx, float qy, float qz,
                                           float (&bd)[KP1], int (&bi)[KP1], unsigned &evaluated) {
    if (COUNT) evaluated += p1 - p0;
    for (unsigned p = p0; p < p1; p += 4) {
        const unsigned last = p1 - 1;
        const float4 v0 = sorted[p], v1 = sorted[min(p + 1, last)], v2 = sorted[min(p + 2, last)], v3 = sorted[min(p + 3, last)];
        insert<KP1>(bd, bi, pdf_sqdist3(qx - v0.x, qy - v0.y, qz - v0.z), __float_as_int(v0.w));
        if (p + 1 < p1) insert<KP1>(bd, bi, pdf_sqdist3(qx - v1.x, qy - v1.y, qz - v1.z), __float_as_int(v1.w));
        if (p + 2 < p1) insert<KP1>(bd, bi, pdf_sqdist3(qx - v2.x, qy - v2.y, qz - v2.z), __float_as_int(v2.w));
        if (p + 3 < p1) insert<KP1>(bd, bi, pdf_sqdist3(qx - v3.x, qy - v3.y, qz - v3.z), __float_as_int(v3.w));
    }
}

// ranges of row (dz, dy) of the shell of radius R around cell (cx, cy, cz)
__device__ __forceinline__ RowRanges row_ranges(const SceneGrid &g, const unsigned *__restrict__ cs, int cx, int cy, int cz, int R, int dz, int dy) {
    RowRanges r = {0u, 0u, 0u, 0u};
    const int z = cz + dz, y = cy + dy;
    if (z < 0 || z >= g.nz || y < 0 || y >= g.ny) return r;
    const unsigned *row = cs + (z * g.ny + y) * g.nx;
    const bool face = dz == R || dz == -R || dy == R || dy == -R;   // (wave-uniform)
    if (face) {
        r.a0 = row[max(cx - R, 0)]; r.a1 = row[min(cx + R, g.nx - 1) + 1];
    } else {   // R >= 1: the two x-faces
        if (cx - R >= 0) { r.a0 = row[cx - R]; r.a1 = row[cx - R + 1]; }
        if (cx + R < g.nx) { r.b0 = row[cx + R]; r.b1 = row[cx + R + 1]; }
    }
    return r;
}

template <int KP1, bool SELF, bool COUNT>
__global__ __launch_bounds__(PB) void k_grid_query(int m, int b, const float *__restrict__ new_xyz, const int *__restrict__ new_offset,
                                                   const SceneGrid *__restrict__ grids, const unsigned *__restrict__ cell_start,
                                                   const float4 *__restrict__ sorted, int *__restrict__ idx,
                                                   float *__restrict__ dist2, int *__restrict__ redo, unsigned long long *pairs) {
    constexpr int K = KP1 - 1;
    const int t_ = blockIdx.x * PB + threadIdx.x;
    unsigned evaluated = 0;
    const bool live = t_ < m;                 // (no early return: the row walk below keeps the wave's trip counts uniform)
    int q = live ? t_ : m - 1;
    float qx, qy, qz;
    if (SELF) {
        const float4 me = sorted[q];   // the sorted copy keeps the scenes in order: position t_ belongs to scene_of(t_)
        q = __float_as_int(me.w);
        qx = me.x; qy = me.y; qz = me.z;
    } else {
        qx = new_xyz[3 * (size_t)q]; qy = new_xyz[3 * (size_t)q + 1]; qz = new_xyz[3 * (size_t)q + 2];
    }
    const int s = scene_of(q, new_offset, b);
    const SceneGrid g = grids[s];
    bool need_redo = g.n < KP1;  // placeholders / not enough points for the tie test: exact scan
    float bd[KP1];
    int bi[KP1];
#pragma unroll
    for (int t = 0; t < KP1; ++t) { bd[t] = 3.0e38f; bi[t] = 0x7fffffff; }
    const int cx = cell_coord(qx, g.minx, g.inv_h, g.nx), cy = cell_coord(qy, g.miny, g.inv_h, g.ny), cz = cell_coord(qz, g.minz, g.inv_h, g.nz);
    const unsigned *cs = cell_start + g.cell_base;
    const int rmax = max(g.nx, max(g.ny, g.nz));
    bool done = need_redo || !live;
    for (int R = 0; __builtin_amdgcn_ballot_w64(!done) != 0ull; ++R) {
        if (!done) {
            const int side = 2 * R + 1, rows = side * side;
            RowRanges cur = row_ranges(g, cs, cx, cy, cz, R, -R, -R);
            for (int e = 0; e < rows; ++e) {
                RowRanges nxt = {0u, 0u, 0u, 0u};
                if (e + 1 < rows) nxt = row_ranges(g, cs, cx, cy, cz, R, (e + 1) / side - R, (e + 1) % side - R);
                scan_range<KP1, COUNT>(cur.a0, cur.a1, sorted, qx, qy, qz, bd, bi, evaluated);
                scan_range<KP1, COUNT>(cur.b0, cur.b1, sorted, qx, qy, qz, bd, bi, evaluated);
                cur = nxt;
            }
            // certified radius: distance from the query to the nearest face of the searched cube that is not a grid wall
            float rc = 3.0e38f;
            if (cx - R > 0) rc = fminf(rc, qx - (g.minx + (float)(cx - R) * g.h));
            if (cx + R < g.nx - 1) rc = fminf(rc, (g.minx + (float)(cx + R + 1) * g.h) - qx);
            if (cy - R > 0) rc = fminf(rc, qy - (g.miny + (float)(cy - R) * g.h));
            if (cy + R < g.ny - 1) rc = fminf(rc, (g.miny + (float)(cy + R + 1) * g.h) - qy);
            if (cz - R > 0) rc = fminf(rc, qz - (g.minz + (float)(cz - R) * g.h));
            if (cz + R < g.nz - 1) rc = fminf(rc, (g.minz + (float)(cz + R + 1) * g.h) - qz);
            // slack for the rounding of cell edges and distances (relative + a few ulps of the coordinate magnitude)
            const float rs = fmaxf(rc * 0.999f - 2e-6f * (fabsf(qx) + fabsf(qy) + fabsf(qz) + g.h), 0.f);
            done = rc >= 3.0e38f                // the cube is the whole grid
                   || bd[KP1 - 1] < rs * rs     // the K+1 best are all inside the certified ball
                   || R >= rmax;
        }
    }
    if (!live) return;
    // ties among the K+1 best distances make the reference's answer depend on its heap history: exact scan instead
#pragma unroll
    for (int t = 0; t < K; ++t) need_redo |= (bd[t] == bd[t + 1]);
    if (COUNT) {   // a per-lane atomic: measurement launches only
        if (evaluated) atomicAdd(pairs, (unsigned long long)evaluated);
    }
    if (need_redo) {
        const int slot = atomicAdd(&redo[0], 1);
        redo[4 + slot] = q;
        return;
    }
    int *oi = idx + (size_t)q * K;
    float *od = dist2 + (size_t)q * K;
#pragma unroll
    for (int t = 0; t < K; ++t) { oi[t] = bi[t]; od[t] = bd[t]; }
}

// ---------------------------------------------------------------- fixed-radius neighbours over the same grid
// "The first nsample points of the scene, in index order, with d2 < radius^2 (or d2 <= 1e-5), padded with -1" -- the contract of
// pdf_random_ball_query walked along the identity permutation (the table behind the PDF pseudo-label pass, pseudo_label.py) --
// for SELF queries, from the 27 cells around the query instead of the whole scene.  One wave per query: the cells' points are
// read 64 at a time, accepted ones (as-written fp32 distance: same values as the scan) are appended to an LDS list, and the
// nsample smallest INDICES are selected by rank counting (indices are unique), which also puts them in index order.  A query
// whose ball holds more than RQ_CAP points falls back to the in-order scan of its scene inside the same wave.
constexpr int RQ_WAVES = 4, RQ_CAP = 1024;

__global__ __launch_bounds__(64 * RQ_WAVES) void k_grid_radius_self(int n, int b, int nsample, float radius, const float *__restrict__ xyz,
                                                                    const int *__restrict__ offset, const SceneGrid *__restrict__ grids,
                                                                    const unsigned *__restrict__ cell_start, const float4 *__restrict__ sorted,
                                                                    int *__restrict__ idx, float *__restrict__ dist2) {
    __shared__ int s_idx[RQ_WAVES][RQ_CAP];
    __shared__ float s_d2[RQ_WAVES][RQ_CAP];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = blockIdx.x * RQ_WAVES + wave;
    if (q >= n) return;   // wave-uniform, no block barrier below
    int *ci = s_idx[wave];
    float *cd = s_d2[wave];
    const int sc = scene_of(q, offset, b);
    const SceneGrid g = grids[sc];
    const float qx = xyz[3 * (size_t)q], qy = xyz[3 * (size_t)q + 1], qz = xyz[3 * (size_t)q + 2];
    const float r2 = radius * radius;
    const int cx = cell_coord(qx, g.minx, g.inv_h, g.nx), cy = cell_coord(qy, g.miny, g.inv_h, g.ny), cz = cell_coord(qz, g.minz, g.inv_h, g.nz);
    int cnt = 0;   // wave-uniform
    bool overflow = false;
    for (int dz = -1; dz <= 1 && !overflow; ++dz)
        for (int dy = -1; dy <= 1 && !overflow; ++dy) {
            const int y = cy + dy, z = cz + dz;
            if (y < 0 || y >= g.ny || z < 0 || z >= g.nz) continue;
            const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);           // the three x-cells are contiguous in the sorted copy
            const int c0 = g.cell_base + (z * g.ny + y) * g.nx + x0;
            const unsigned beg = cell_start[c0], end = cell_start[c0 + (x1 - x0) + 1];
            for (unsigned base = beg; base < end; base += 64) {
                const unsigned p = base + lane;
                bool ok = false;
                float d2 = 0.f;
                int id = -1;
                if (p < end) {
                    const float4 v = sorted[p];
                    id = __float_as_int(v.w);
                    d2 = pdf_sqdist3(qx - v.x, qy - v.y, qz - v.z);
                    ok = d2 <= 1e-5f || d2 < r2;
                }
                const unsigned long long mask = __builtin_amdgcn_ballot_w64(ok);
                const int pos = cnt + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
                if (ok && pos < RQ_CAP) { ci[pos] = id; cd[pos] = d2; }
                cnt += __builtin_popcountll(mask);
                if (cnt > RQ_CAP) { overflow = true; break; }
            }
        }
    int *oi = idx + (size_t)q * nsample;
    float *od = dist2 + (size_t)q * nsample;
    if (overflow) {   // in-order scan of the scene (random_ball_query_kernel with the identity permutation)
        int c2 = 0;
        for (int base = g.start; base < g.start + g.n && c2 < nsample; base += 64) {
            const int i = base + lane;
            bool ok = false;
            float d2 = 0.f;
            if (i < g.start + g.n) {
                const float x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
                d2 = pdf_sqdist3(qx - x, qy - y, qz - z);
                ok = d2 <= 1e-5f || d2 < r2;
            }
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(ok);
            const int pos = c2 + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
            if (ok && pos < nsample) { oi[pos] = i; od[pos] = d2; }
            c2 += __builtin_popcountll(mask);
        }
        if (c2 > nsample) c2 = nsample;
        for (int s = c2 + lane; s < nsample; s += 64) { oi[s] = -1; od[s] = 1e10f; }
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // rank of every candidate among the candidates' indices; ranks below nsample are the answer, already in index order
    for (int c = lane; c < cnt; c += 64) {
        const int mine = ci[c];
        int rank = 0;
        for (int o = 0; o < cnt; ++o) rank += ci[o] < mine ? 1 : 0;
        if (rank < nsample) { oi[rank] = mine; od[rank] = cd[c]; }
    }
    for (int s = min(cnt, nsample) + lane; s < nsample; s += 64) { oi[s] = -1; od[s] = 1e10f; }
}

}  // namespace kg

extern "C" long pdf_knn_workspace_bytes(int b, int n, int m) {
    if (b < 1 || n < 0 || m < 0) return -1;
    return (long)kg::make_layout(b, n, m).total;
}

// nsample values served by the grid path (others fall back to the scan inside pdf_knn_query_ws)
extern "C" int pdf_knn_grid_supported(int nsample) { return nsample == 3 || nsample == 8 || nsample == 16; }

// the grid of the SOURCE points (bounding boxes, cell size, counting sort of the points by cell) in `workspace`
static int knn_grid_build(int n, const float *xyz, const int *offset, int b, void *workspace, long workspace_bytes, void *stream) {
    const kg::Layout L = kg::make_layout(b, n, 0);
    if (!workspace || workspace_bytes < (long)L.total) return PDF_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace);
    kg::SceneGrid *grids = reinterpret_cast<kg::SceneGrid *>(ws + L.grid);
    unsigned *cell_start = reinterpret_cast<unsigned *>(ws + L.cell_start);
    unsigned *cursor = reinterpret_cast<unsigned *>(ws + L.cursor);
    int *cell_of = reinterpret_cast<int *>(ws + L.cell_of);
    float4 *sorted = reinterpret_cast<float4 *>(ws + L.sorted);
    hipError_t e = hipMemsetAsync(ws + L.cell_start, 0, L.cell_of - L.cell_start, s);  // counts + cursors
    if (e != hipSuccess) return (int)e;
    // cell size: `ppc` points per cell if the points filled the bounding box (they lie on surfaces, so occupied cells hold more)
    static const float ppc_env = [] { const char *v = getenv("PDFOPS_KNN_PPC"); return v ? (float)atof(v) : 0.f; }();
    const float ppc = ppc_env > 0.f ? ppc_env : 1.0f;   // measured on 12 x 100k-point scenes: 1 beats 4 by 25 % at level 1, equal below
    kg::k_grid_setup<<<b, kg::PB, 0, s>>>(xyz, offset, grids, ppc, 0.f);
    kg::k_grid_hist<<<pdf_divup(n, kg::PB), kg::PB, 0, s>>>(n, b, xyz, offset, grids, cell_start, cell_of);
    kg::k_grid_scan<<<b, 1024, 0, s>>>(grids, cell_start);
    kg::k_grid_scatter<<<pdf_divup(n, kg::PB), kg::PB, 0, s>>>(n, xyz, cell_of, cell_start, cursor, sorted);
    return pdf_launch_status();
}

// the queries over a grid built by knn_grid_build in the same workspace (any number of query sets, any supported nsample)
static int knn_query_grid(int m, int nsample, int n, const float *xyz, const float *new_xyz, const int *offset, const int *new_offset, int b,
                          int *idx, float *dist2, void *workspace, long workspace_bytes, unsigned long long *pairs, void *stream) {
    const kg::Layout L = kg::make_layout(b, n, m);
    if (!workspace || workspace_bytes < (long)L.total) return PDF_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace);
    kg::SceneGrid *grids = reinterpret_cast<kg::SceneGrid *>(ws + L.grid);
    unsigned *cell_start = reinterpret_cast<unsigned *>(ws + L.cell_start);
    float4 *sorted = reinterpret_cast<float4 *>(ws + L.sorted);
    int *redo = reinterpret_cast<int *>(ws + L.redo);
    hipError_t e = hipMemsetAsync(redo, 0, 16, s);
    if (e != hipSuccess) return (int)e;
    const int grid = pdf_divup(m, kg::PB);
    const bool self = new_xyz == xyz && new_offset == offset && m == n && getenv("PDFOPS_KNN_NO_SELF") == nullptr;
#define PDF_KQ2(KP1_, SELF_, COUNT_) kg::k_grid_query<KP1_, SELF_, COUNT_><<<grid, kg::PB, 0, s>>>(m, b, new_xyz, new_offset, grids, cell_start, sorted, idx, dist2, redo, pairs)
#define PDF_KQ(KP1_) do { if (pairs) { if (self) PDF_KQ2(KP1_, true, true); else PDF_KQ2(KP1_, false, true); } \
                          else       { if (self) PDF_KQ2(KP1_, true, false); else PDF_KQ2(KP1_, false, false); } } while (0)
    if (nsample == 3) PDF_KQ(4);
    else if (nsample == 8) PDF_KQ(9);
    else PDF_KQ(17);
#undef PDF_KQ
#undef PDF_KQ2
    int rc = pdf_launch_status();
    if (rc != PDF_OK) return rc;
    if (getenv("PDFOPS_KNN_DEBUG")) {   // diagnostics: size of the exact-scan redo list (synchronises)
        int cnt = -1;
        (void)hipMemcpyAsync(&cnt, redo, sizeof(int), hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
        fprintf(stderr, "[pdfops] knn grid: m=%d nsample=%d b=%d redo=%d\n", m, nsample, b, cnt);
    }
    return pdf_knn_query_list(m, nsample, xyz, new_xyz, offset, new_offset, b, idx, dist2, redo + 4, redo, stream);
}

static int knn_query_ws(int m, int nsample, int n, const float *xyz, const float *new_xyz, const int *offset,
                        const int *new_offset, int b, int *idx, float *dist2, void *workspace,
                        long workspace_bytes, unsigned long long *pairs, void *stream) {
    if (m == 0) return PDF_OK;   // (0-size tensors carry null pointers: not an argument error)
    if (m < 0 || n < 0 || b < 1 || !xyz || !new_xyz || !offset || !new_offset || !idx || !dist2) return PDF_ERR_BAD_ARG;
    if (nsample < 1 || nsample > 128) return PDF_ERR_NSAMPLE;
    if (!pdf_knn_grid_supported(nsample) || b > 64)
        return pairs ? PDF_ERR_UNSUPPORTED : pdf_knn_query(m, nsample, xyz, new_xyz, offset, new_offset, b, idx, dist2, stream);
    if (!workspace || workspace_bytes < (long)kg::make_layout(b, n, m).total) return PDF_ERR_BAD_ARG;
    const int rc = knn_grid_build(n, xyz, offset, b, workspace, workspace_bytes, stream);
    if (rc != PDF_OK) return rc;
    return knn_query_grid(m, nsample, n, xyz, new_xyz, offset, new_offset, b, idx, dist2, workspace, workspace_bytes, pairs, stream);
}

// The two halves on their own: ONE grid per set of source points, any number of query sets over it (the geometry pre-pass asks 2-3 tables of
// every level: 14 grid builds per pre-pass were 2.0 + 0.6 of its 17 ms that are not farthest-point sampling).  The workspace of the build
// must be at least pdf_knn_workspace_bytes(b, n, m) for the largest m queried over it.  nsample must be one pdf_knn_grid_supported accepts.
extern "C" int pdf_knn_grid_build(int n, const float *xyz, const int *offset, int b, void *workspace, long workspace_bytes, void *stream) {
    if (n < 1 || b < 1 || b > 64 || !xyz || !offset) return PDF_ERR_BAD_ARG;
    return knn_grid_build(n, xyz, offset, b, workspace, workspace_bytes, stream);
}
extern "C" int pdf_knn_query_grid(int m, int nsample, int n, const float *xyz, const float *new_xyz, const int *offset, const int *new_offset, int b,
                                  int *idx, float *dist2, void *workspace, long workspace_bytes, void *stream) {
    if (m == 0) return PDF_OK;
    if (m < 0 || n < 1 || b < 1 || b > 64 || !xyz || !new_xyz || !offset || !new_offset || !idx || !dist2) return PDF_ERR_BAD_ARG;
    if (!pdf_knn_grid_supported(nsample)) return PDF_ERR_NSAMPLE;
    return knn_query_grid(m, nsample, n, xyz, new_xyz, offset, new_offset, b, idx, dist2, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int pdf_knn_query_ws(int m, int nsample, int n, const float *xyz, const float *new_xyz, const int *offset,
                                const int *new_offset, int b, int *idx, float *dist2, void *workspace,
                                long workspace_bytes, void *stream) {
    return knn_query_ws(m, nsample, n, xyz, new_xyz, offset, new_offset, b, idx, dist2, workspace, workspace_bytes, nullptr, stream);
}
// the same launch sequence with the grid kernel counting the candidate distances it evaluates: *pairs (device, zeroed by the caller)
// += that count.  Measurement aid (the count costs a per-lane atomic); PDF_ERR_UNSUPPORTED where the grid path does not apply.
extern "C" int pdf_knn_query_ws_counted(int m, int nsample, int n, const float *xyz, const float *new_xyz, const int *offset,
                                        const int *new_offset, int b, int *idx, float *dist2, void *workspace,
                                        long workspace_bytes, unsigned long long *pairs, void *stream) {
    if (!pairs) return PDF_ERR_BAD_ARG;
    return knn_query_ws(m, nsample, n, xyz, new_xyz, offset, new_offset, b, idx, dist2, workspace, workspace_bytes, pairs, stream);
}


namespace kg {
__global__ __launch_bounds__(256) void k_zero_words(unsigned *__restrict__ p, size_t words) {
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < words && i < ((size_t)blockIdx.x + 1) * 1024; i += 256) p[i] = 0u;
}
}  // namespace kg

// Fixed-radius neighbour table of a batch with itself: idx (n, nsample) = the first nsample points of the query's scene, in index
// order, within `radius` (the query included), -1 padded; dist2 = squared distances (1e10 padded).  Same results as
// pdf_random_ball_query with order = identity and min_radius = 0.  Workspace: pdf_knn_workspace_bytes(b, n, 0).
extern "C" int pdf_radius_neighbors_self(int n, int nsample, float radius, const float *xyz, const int *offset, int b, int *idx,
                                         float *dist2, void *workspace, long workspace_bytes, void *stream) {
    if (n == 0) return PDF_OK;
    if (n < 0 || b < 1 || b > 64 || !xyz || !offset || !idx || !dist2 || !(radius > 0.f)) return PDF_ERR_BAD_ARG;
    if (nsample < 1 || nsample > kg::RQ_CAP) return PDF_ERR_NSAMPLE;
    const kg::Layout L = kg::make_layout(b, n, 0);
    if (!workspace || workspace_bytes < (long)L.total) return PDF_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace);
    kg::SceneGrid *grids = reinterpret_cast<kg::SceneGrid *>(ws + L.grid);
    unsigned *cell_start = reinterpret_cast<unsigned *>(ws + L.cell_start);
    unsigned *cursor = reinterpret_cast<unsigned *>(ws + L.cursor);
    int *cell_of = reinterpret_cast<int *>(ws + L.cell_of);
    float4 *sorted = reinterpret_cast<float4 *>(ws + L.sorted);
    // (a zero-fill KERNEL, not hipMemsetAsync: this table is part of the pseudo-label pass, which a captured step records into its graph;
    //  a graph holding a memset node faulted at replay -- "write access to a read-only page" -- once any device-to-device copy ran
    //  between capture and replay on ROCm 7.2: tools/scratch reproduction in docs/NOTEBOOK.md, round 5)
    {
        const size_t words = (size_t)(L.cell_of - L.cell_start) / 4;
        kg::k_zero_words<<<pdf_divup((long)words, 1024), 256, 0, s>>>(reinterpret_cast<unsigned *>(ws + L.cell_start), words);
    }
    kg::k_grid_setup<<<b, kg::PB, 0, s>>>(xyz, offset, grids, 1.0f, radius * 1.0001f);   // cell >= radius
    kg::k_grid_hist<<<pdf_divup(n, kg::PB), kg::PB, 0, s>>>(n, b, xyz, offset, grids, cell_start, cell_of);
    kg::k_grid_scan<<<b, 1024, 0, s>>>(grids, cell_start);
    kg::k_grid_scatter<<<pdf_divup(n, kg::PB), kg::PB, 0, s>>>(n, xyz, cell_of, cell_start, cursor, sorted);
    kg::k_grid_radius_self<<<pdf_divup(n, kg::RQ_WAVES), 64 * kg::RQ_WAVES, 0, s>>>(n, b, nsample, radius, xyz, offset, grids, cell_start, sorted, idx, dist2);
    return pdf_launch_status();
}
