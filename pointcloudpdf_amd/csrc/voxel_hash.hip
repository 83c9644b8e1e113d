// Voxel keys of GridSample for gfx950 -- replaces the numpy front half of
// pointcept/datasets/transform.py:813-823 + fnv_hash_vec (:911-925) for a whole batch of scenes:
//     scaled     = coord / grid_size                    (float64: float32 coordinates divided by a 0-d float64 array promote
//                                                        under NumPy >= 2; `f32 = 1` evaluates the NumPy 1.x float32 variant)
//     grid_coord = floor(scaled).astype(int) - min over the scene
//     key        = FNV over the three uint64 grid coordinates: h = 14695981039346656037; per axis h *= 1099511628211; h ^= g
//                  (multiply THEN xor -- FNV-1 order, although upstream's docstring says "FNV64-1A"; the code is what counts)
// One lane per point: 12 B in, 24 B (int64 grid) + 8 B (key) out -- HBM-bound, no reuse.  The scene of a point comes from a
// scan of the (small) offset array; the per-scene minimum grid coordinate is an input (floor is monotone, so it is the floor of
// the scene's minimum coordinate / grid_size, computed by the host code from a segmented min).
#include "pdfops_common.h"

namespace {

__global__ __launch_bounds__(256) void k_grid_hash(long n, int b, const float *__restrict__ coord, const int *__restrict__ offset,
                                                   double gx, double gy, double gz, int f32, const long long *__restrict__ min_grid,
                                                   long long *__restrict__ grid, unsigned long long *__restrict__ key) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int s = 0;
    while (s < b - 1 && i >= offset[s]) ++s;
    const float cx = coord[3 * i], cy = coord[3 * i + 1], cz = coord[3 * i + 2];
    long long g[3];
    if (f32) {   // NumPy 1.x value-based casting: the division stays in float32
        g[0] = (long long)floorf(cx / (float)gx); g[1] = (long long)floorf(cy / (float)gy); g[2] = (long long)floorf(cz / (float)gz);
    } else {
        g[0] = (long long)floor((double)cx / gx); g[1] = (long long)floor((double)cy / gy); g[2] = (long long)floor((double)cz / gz);
    }
    unsigned long long h = 14695981039346656037ull;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        g[a] -= min_grid[3 * s + a];
        grid[3 * i + a] = g[a];
        h *= 1099511628211ull;
        h ^= (unsigned long long)g[a];
    }
    key[i] = h;
}

}  // namespace

// coord (n,3) f32, offset (b) cumulative ends, min_grid (b,3) int64 -> grid (n,3) int64 (scene-relative), key (n) uint64.
// Algorithmic bytes: 12 n + 24 n + 8 n.
extern "C" int pdf_grid_hash(long n, int b, const float *coord, const int *offset, double gx, double gy, double gz, int f32,
                             const long long *min_grid, long long *grid, unsigned long long *key, void *stream) {
    if (n == 0) return PDF_OK;
    if (n < 0 || b < 1 || !coord || !offset || !min_grid || !grid || !key || !(gx > 0.0) || !(gy > 0.0) || !(gz > 0.0)) return PDF_ERR_BAD_ARG;
    k_grid_hash<<<pdf_divup(n, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(n, b, coord, offset, gx, gy, gz, f32, min_grid, grid, key);
    return pdf_launch_status();
}
