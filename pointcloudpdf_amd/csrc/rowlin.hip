// Dense per-point Linear layers on the matrix cores (the only GEMM-shaped work on this path).
//
//   forward / input gradient :  Y[n,o] (+)= sum_k f(X[n,k]) * Wt(k,o) + bias[o]      f(x) = relu?(x*scale[k] + shift[k]) or x
//   weight gradient          :  dW[o,k] += sum_n G[n,o] * f(X[n,k]),  db[o] += sum_n G[n,o]
//
// f folds the BatchNorm-affine + ReLU that precedes the Linear in the Bottleneck (point_transformer_seg.py:184-192), so the
// normalised activation is never written to HBM; an optional epilogue emits per-column sum / sum-of-squares partials of Y for
// the BatchNorm that follows.  Both kernels use v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate: bit-equal to an fmaf chain,
// 157 TF peak), because the reference computes these layers in fp32 and the 1e-4 bar is on fp32 results.
// N ~ 10^5 rows with 3..512 channels: HBM-bound (read X once per 64 output columns, write Y once); MFMA keeps the ~0.4 MFLOP/pt
// off the VALU.  Shapes are arbitrary (edges are zero-padded in LDS), X / Y / G carry a row stride.
#include "pdfops_common.h"
#include <cstdlib>

namespace rl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 128, BN = 64, BK = 32, LDT = 33;

struct FwdArgs {
    long N;
    int K, O;              // reduction width, output width
    const float *X; long ldx;
    const float *W; long wso, wsk;   // Wt(k, o) = W[o * wso + k * wsk]
    const float *bias;     // (O) or null
    const float *scale, *shift;      // (K) prologue coefficients (PRE)
    int relu;
    float *Y; long ldy;
    int accumulate;        // Y += ... instead of Y = ...
    float *partial;        // [gridDim.x][2 * O] column sum | sum of squares (STATS)
};

template <bool PRE, bool STATS>
__global__ __launch_bounds__(256) void k_rowlin(FwdArgs a) {
    __shared__ float xs[BM * LDT];
    __shared__ float ws[BN * LDT];
    __shared__ float red[4][2][BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long row0 = (long)blockIdx.x * BM;
    const int o0 = blockIdx.y * BN;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const int kk = tid & 31, rr = tid >> 5;
    for (int k0 = 0; k0 < a.K; k0 += BK) {
        const int k = k0 + kk;
        float sc = 1.f, sh = 0.f;
        if (PRE && k < a.K) { sc = a.scale[k]; sh = a.shift[k]; }
#pragma unroll
        for (int i = 0; i < BM / 8; ++i) {
            const int r = rr + 8 * i;
            const long row = row0 + r;
            float v = 0.f;
            if (row < a.N && k < a.K) {
                v = a.X[row * a.ldx + k];
                if (PRE) { v = v * sc + sh; if (a.relu) v = fmaxf(v, 0.f); }
            }
            xs[r * LDT + kk] = v;
        }
#pragma unroll
        for (int i = 0; i < BN / 8; ++i) {
            const int o = rr + 8 * i;
            float v = 0.f;
            if (o0 + o < a.O && k < a.K) v = a.W[(long)(o0 + o) * a.wso + (long)k * a.wsk];
            ws[o * LDT + kk] = v;
        }
        __syncthreads();
        const float *xa = xs + (wave * 32 + (lane & 31)) * LDT + (lane >> 5);
        const float *wb0 = ws + (lane & 31) * LDT + (lane >> 5);
        const float *wb1 = ws + (32 + (lane & 31)) * LDT + (lane >> 5);
#pragma unroll
        for (int q = 0; q < BK; q += 2) {
            const float av = xa[q];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wb0[q], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wb1[q], acc1, 0, 0, 0);
        }
        __syncthreads();
    }
    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int col = o0 + half * 32 + (lane & 31);
        const float bv = (a.bias && col < a.O) ? a.bias[col] : 0.f;
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const long row = row0 + wave * 32 + i;
            if (row < a.N && col < a.O) {
                float v = (half ? acc1[r] : acc0[r]) + bv;
                float *dst = a.Y + row * a.ldy + col;
                if (a.accumulate) v += *dst;
                *dst = v;
                s += v;
                ss += v * v;
            }
        }
        if (STATS) {
            s += __shfl_xor(s, 32, 64);
            ss += __shfl_xor(ss, 32, 64);
            if (lane < 32) { red[wave][0][half * 32 + lane] = s; red[wave][1][half * 32 + lane] = ss; }
        }
    }
    if (STATS) {
        __syncthreads();
        if (tid < BN && o0 + tid < a.O) {
            const float s = red[0][0][tid] + red[1][0][tid] + red[2][0][tid] + red[3][0][tid];
            const float ss = red[0][1][tid] + red[1][1][tid] + red[2][1][tid] + red[3][1][tid];
            a.partial[(size_t)blockIdx.x * 2 * a.O + o0 + tid] = s;
            a.partial[(size_t)blockIdx.x * 2 * a.O + a.O + o0 + tid] = ss;
        }
    }
}

struct WArgs {
    long N;
    int K, O;
    const float *G; long ldg;        // (N, O) upstream gradient
    const float *X; long ldx;        // (N, K) layer input (before the folded affine + ReLU)
    const float *scale, *shift;
    int relu;
    float *dW;                       // (O, K), written by the slab reduction
    float *db;                       // (O) or null
    long rows_per_block;
    float *slab;                     // [gridDim.y][gridDim.x][32 * 32]: one block of dW per WORKGROUP (rl2::k_slab_reduce sums them in order)
    float *bslab;                    // [ceil(O / 32)][gridDim.x][32]: column sums of G (k-block 0)
};

// One wave = one 32x32 block of dW over a row range: A operand = G^T (lanes along o), B operand = f(X) (lanes along k);
// both are read straight from global memory, coalesced 128 B per half-wave, two rows per MFMA.
template <bool PRE>
__global__ __launch_bounds__(256) void k_wgrad(WArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kblocks = (a.K + 31) / 32;
    const int o0 = (blockIdx.y / kblocks) * 32, c0 = (blockIdx.y % kblocks) * 32;
    const int ch = lane & 31, par = lane >> 5;
    const int oc = o0 + ch, cc = c0 + ch;
    const bool ov = oc < a.O, cv = cc < a.K;
    float sc = 1.f, sh = 0.f;
    if (PRE && cv) { sc = a.scale[cc]; sh = a.shift[cc]; }
    const long rb = (long)blockIdx.x * a.rows_per_block;
    const long re = min(rb + a.rows_per_block, a.N);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float gsum = 0.f;
    for (long r0 = rb + 8 * wave; r0 < re; r0 += 32) {  // 8 rows (4 MFMAs) per wave and trip, waves interleaved
        float g[4], x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long r = r0 + 2 * u + par;
            g[u] = (r < re && ov) ? a.G[r * a.ldg + oc] : 0.f;
            float v = 0.f;
            if (r < re && cv) {
                v = a.X[r * a.ldx + cc];
                if (PRE) { v = v * sc + sh; if (a.relu) v = fmaxf(v, 0.f); }
            }
            x[u] = v;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(g[u], x[u], acc, 0, 0, 0);
            gsum += g[u];
        }
    }
    // the four waves' blocks added in wave order through LDS: ONE slab per workgroup (round 6: a slab per wave was 4x the workspace
    // traffic and 4x the reducer's reads -- the fixed cost of this kernel at a few thousand rows; 46 -> 36 us at 160k x 48 -> 48,
    // 27 -> 23 us at 10k x 192 -> 192.  Measured and dropped: 8 / 16 loads in flight per lane instead of 4 (+-0 / slower); every wave its
    // own 32 x 32 block of a 64 x 64 region over all rows of the range, sharing operand rows through L1 (wide layers -12 %, the rest +20 %))
    __shared__ float comb[4][1024 + 32];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);  // o index inside the block
        comb[wave][i * 32 + ch] = acc[r];
    }
    gsum += __shfl_xor(gsum, 32, 64);
    if (lane < 32) comb[wave][1024 + ch] = gsum;
    __syncthreads();
    const size_t sidx = blockIdx.x, nsl = gridDim.x;
    float *slab = a.slab + ((size_t)blockIdx.y * nsl + sidx) * 1024;
#pragma unroll
    for (int e = threadIdx.x; e < 1024; e += 256) slab[e] = ((comb[0][e] + comb[1][e]) + comb[2][e]) + comb[3][e];
    if (a.db && c0 == 0 && threadIdx.x < 32)
        a.bslab[((size_t)(o0 / 32) * nsl + sidx) * 32 + threadIdx.x] =
            ((comb[0][1024 + threadIdx.x] + comb[1][1024 + threadIdx.x]) + comb[2][1024 + threadIdx.x]) + comb[3][1024 + threadIdx.x];
}

}  // namespace rl

// streaming kernels (rowlin2.hip): cover the channel widths of the Bottleneck; the tiled kernels above take every other shape
namespace rl2 {
int try_forward(long n, int k, int o, int nin, int nout, const float *const *x, long ldx, const float *const *w, int transpose_w,
                const float *const *bias, const float *scale, const float *shift, int relu, float *const *y, long ldy,
                int accumulate, float *partial, int mma, hipStream_t s, const float *roww = nullptr, long rws = 0, const float *bx = nullptr,
                long ldb = 0, const float *bcoef = nullptr, int brelu = 0, int *partial_rows = nullptr, long ldw = 0, void *handoff = nullptr);
long stats_rows_floats(long n, int o);
int try_wgrad(long n, int k, int o, int ng, const float *const *g, long ldg, const float *x, long ldx, const float *scale,
              const float *shift, int relu, float *const *dw, float *const *db, float *ws, int mma, hipStream_t s, const float *roww = nullptr, long rws = 0);
long wgrad_ws_floats(long n, int k, int o, int ng);
constexpr int WG_MAXG = 5;   // (as in rowlin2_impl.h)
struct RArgs {
    const float *slab, *bslab;
    float *dW[WG_MAXG], *db[WG_MAXG];
    int B, tiles_k, tiles, otiles, split, K, O;
};
int try_wgrad_group(long n, int k, int o, int ng, const float *const *g, long ldg, const float *const *x, long ldx, const float *const *scale,
                    const float *const *shift, const int *relu, float *const *dw, float *const *db, float *ws, int mma, hipStream_t s);
void launch_slab_reduce(const RArgs &a, int ng, bool any_bias, hipStream_t s);
int stats_rows(long n);
}  // namespace rl2

// split of the tiled weight-gradient kernel (shapes outside the streaming kernels)
static inline void tiled_wg_plan(long n, int k, int o, int *blocks_oc, long *split, long *rows_per_block) {
    *blocks_oc = ((o + 31) / 32) * ((k + 31) / 32);
    long sp = (2048 + *blocks_oc - 1) / *blocks_oc;           // ~2048 workgroups in flight
    const long max_split = (n + 255) / 256;                   // at least 256 rows per workgroup
    if (sp > max_split) sp = max_split;
    if (sp < 1) sp = 1;
    *rows_per_block = ((n + sp - 1) / sp + 31) / 32 * 32;
    *split = (n + *rows_per_block - 1) / *rows_per_block;
}

static inline bool rowlin_streams(int k, int o) {
    return (k == 32 || k == 64 || k == 128 || k == 256 || k == 512) && o % 16 == 0 && getenv("PDFOPS_ROWLIN_TILED") == nullptr;
}

extern "C" int pdf_rowlin_partial_rows(long n, int k, int o) {
    return rowlin_streams(k, o) ? rl2::stats_rows(n) : (int)((n + rl::BM - 1) / rl::BM);
}
extern "C" long pdf_rowlin_partial_floats(long n, int o) {   // rows of either kernel + the handoff scratch of an in-kernel finalize
    const long a = (n + rl::BM - 1) / rl::BM * 2 * (long)o, b = rl2::stats_rows_floats(n, o);
    return (a > b ? a : b) + (long)PDF_HO_FLOATS;
}

// Y (n, o; row stride ldy) (+)= f(X (n, k; row stride ldx)) * Wt + bias.  transpose_w = 0: W is (o, k) row-major (forward);
// 1: W is (k, o) row-major, i.e. the layer's own (out, in) weight used for the input gradient dX = G W.
extern "C" int pdf_rowlin_forward(long n, int k, int o, const float *x, long ldx, const float *w, int transpose_w,
                                  const float *bias, const float *scale, const float *shift, int relu, float *y, long ldy,
                                  int accumulate, float *partial, int mma_input, void *stream) {
    if (mma_input < 0 || mma_input > 2) return PDF_ERR_BAD_ARG;
    if (n < 1 || k < 1 || o < 1 || !x || !w || !y || ldx < k || ldy < o) return PDF_ERR_BAD_ARG;
    if (rowlin_streams(k, o)) {
        if (rl2::try_forward(n, k, o, 1, 1, &x, ldx, &w, transpose_w, &bias, scale, shift, relu, &y, ldy, accumulate, partial,
                             mma_input, static_cast<hipStream_t>(stream)))
            return pdf_launch_status();
        if (partial) return PDF_ERR_BAD_ARG;  // statistics layout is tied to the streaming kernel for these shapes (needs 16-byte alignment)
    }
    // k = 1024 (the TransitionUp head's Linear(2 * 512, 512), point_transformer_seg.py:131-136) as ONE launch of the streaming kernel over
    // the two 512-wide column windows of x and W (its multi-input form: one accumulation chain across both windows, bias at the end --
    // the association of a single 1024-long dot product).  The tiled kernel below walks such a k serially in a handful of workgroups:
    // 274 us at 780 x 1024 -> 512 against ~20 us.
    if (!transpose_w && !partial && !scale && k == 1024 && rowlin_streams(512, o) && !(ldx & 3) && !(ldy & 3)) {
        const float *xs[2] = {x, x + 512}, *ws[2] = {w, w + 512};
        if (rl2::try_forward(n, 512, o, 2, 1, xs, ldx, ws, 0, &bias, nullptr, nullptr, 0, &y, ldy, accumulate, nullptr, mma_input, static_cast<hipStream_t>(stream),
                             nullptr, 0, nullptr, 0, nullptr, 0, nullptr, k))
            return pdf_launch_status();
    }
    rl::FwdArgs a;
    a.N = n; a.K = k; a.O = o; a.X = x; a.ldx = ldx; a.W = w;
    a.wso = transpose_w ? 1 : k; a.wsk = transpose_w ? o : 1;
    a.bias = bias; a.scale = scale; a.shift = shift; a.relu = relu; a.Y = y; a.ldy = ldy; a.accumulate = accumulate;
    a.partial = partial;
    const dim3 grid((unsigned)((n + rl::BM - 1) / rl::BM), (unsigned)((o + rl::BN - 1) / rl::BN));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool pre = scale != nullptr, stats = partial != nullptr;
    if (pre && stats) rl::k_rowlin<true, true><<<grid, 256, 0, s>>>(a);
    else if (pre) rl::k_rowlin<true, false><<<grid, 256, 0, s>>>(a);
    else if (stats) rl::k_rowlin<false, true><<<grid, 256, 0, s>>>(a);
    else rl::k_rowlin<false, false><<<grid, 256, 0, s>>>(a);
    return pdf_launch_status();
}

// Workspace of the weight-gradient entry points, in floats: every workgroup stores its partial block of dW into its own slab, a second
// launch sums the slabs in a fixed order -- no float atomics, bit-reproducible gradients (ng = number of gradients of a _multi call).
extern "C" long pdf_rowlin_wgrad_ws_floats(long n, int k, int o, int ng) {
    if (n < 1 || k < 1 || o < 1 || ng < 1) return 0;
    int blocks_oc; long split, rpb;
    tiled_wg_plan(n, k, o, &blocks_oc, &split, &rpb);
    const long tiled = (long)blocks_oc * split * 4 * 1024 + (long)((o + 31) / 32) * split * 4 * 32;   // (_multi falls back to ng single calls)
    const long streaming = rowlin_streams(k, o) ? rl2::wgrad_ws_floats(n, k, o, ng) : 0;
    return tiled > streaming ? tiled : streaming;
}

extern "C" int pdf_bn_coef_from_partial(const float *partial, int rows, long n, int c, const float *gamma, const float *beta,
                                        float *running_mean, float *running_var, float eps, float momentum, float *coef, void *stream);

// pdf_rowlin_forward with statistics + the coefficients of the train-mode BatchNorm that follows (coef = scale | shift | mean | rstd, 4 o
// floats; running statistics updated): the product's epilogue emits per-row-block column sums, the finalizer launch turns them into the
// coefficients (fixed order: bit-reproducible).
extern "C" int pdf_rowlin_forward_bn(long n, int k, int o, const float *x, long ldx, const float *w, const float *bias, const float *scale,
                                     const float *shift, int relu, float *y, long ldy, float *partial, const float *gamma, const float *beta,
                                     float *running_mean, float *running_var, float eps, float momentum, float *coef, int mma_input, void *stream) {
    if (mma_input < 0 || mma_input > 2) return PDF_ERR_BAD_ARG;
    if (n < 1 || k < 1 || o < 1 || !x || !w || !y || !partial || !gamma || !beta || !coef || ldx < k || ldy < o) return PDF_ERR_BAD_ARG;
    if (rowlin_streams(k, o)) {
        int rows = 0;
        if (rl2::try_forward(n, k, o, 1, 1, &x, ldx, &w, 0, &bias, scale, shift, relu, &y, ldy, 0, partial, mma_input, static_cast<hipStream_t>(stream),
                             nullptr, 0, nullptr, 0, nullptr, 0, &rows)) {
            return pdf_bn_coef_from_partial(partial, rows, n, o, gamma, beta, running_mean, running_var, eps, momentum, coef, stream);
        }
    }
    const int rc = pdf_rowlin_forward(n, k, o, x, ldx, w, 0, bias, scale, shift, relu, y, ldy, 0, partial, mma_input, stream);
    if (rc) return rc;
    return pdf_bn_coef_from_partial(partial, pdf_rowlin_partial_rows(n, k, o), n, o, gamma, beta, running_mean, running_var, eps, momentum, coef, stream);
}

// Internal (csrc/block.hip): pdf_rowlin_forward with the statistics rows of its output and the handoff scratch of the CONSUMER's in-kernel
// BatchNorm finalize zeroed by the product's launch (pdfops_common.h: PdfRowsBn).  Streaming shapes only: returns PDF_ERR_UNSUPPORTED
// otherwise (the caller then takes the finalizer-launch path).  *rows = number of partial rows written.
int pdf_rowlin_forward_stats_ho(long n, int k, int o, const float *x, long ldx, const float *w, const float *bias, const float *scale,
                                const float *shift, int relu, float *y, long ldy, float *rows_out, void *handoff, int *rows, int mma_input,
                                void *stream) {
    if (!rowlin_streams(k, o)) return PDF_ERR_UNSUPPORTED;
    if (!rl2::try_forward(n, k, o, 1, 1, &x, ldx, &w, 0, &bias, scale, shift, relu, &y, ldy, 0, rows_out, mma_input, static_cast<hipStream_t>(stream),
                          nullptr, 0, nullptr, 0, nullptr, 0, rows, 0, handoff))
        return PDF_ERR_UNSUPPORTED;
    return pdf_launch_status();
}

// dW (o, k) = G^T f(X), db (o) = column sums of G (db may be null); both are WRITTEN.  ws: pdf_rowlin_wgrad_ws_floats(n, k, o, 1) floats.
extern "C" int pdf_rowlin_wgrad(long n, int k, int o, const float *g, long ldg, const float *x, long ldx,
                                const float *scale, const float *shift, int relu, float *dw, float *db, float *ws, int mma_input, void *stream) {
    if (mma_input < 0 || mma_input > 2) return PDF_ERR_BAD_ARG;
    if (n < 1 || k < 1 || o < 1 || !g || !x || !dw || !ws || ldg < o || ldx < k) return PDF_ERR_BAD_ARG;
    if (rowlin_streams(k, o) && rl2::try_wgrad(n, k, o, 1, &g, ldg, x, ldx, scale, shift, relu, &dw, &db, ws, mma_input, static_cast<hipStream_t>(stream)))
        return pdf_launch_status();
    rl::WArgs a;
    a.N = n; a.K = k; a.O = o; a.G = g; a.ldg = ldg; a.X = x; a.ldx = ldx; a.scale = scale; a.shift = shift; a.relu = relu;
    a.dW = dw; a.db = db;
    int blocks_oc; long split;
    tiled_wg_plan(n, k, o, &blocks_oc, &split, &a.rows_per_block);
    a.slab = ws;
    a.bslab = ws + (size_t)blocks_oc * split * 1024;
    const dim3 grid((unsigned)split, (unsigned)blocks_oc);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (scale) rl::k_wgrad<true><<<grid, 256, 0, s>>>(a);
    else rl::k_wgrad<false><<<grid, 256, 0, s>>>(a);
    rl2::RArgs r;
    r.slab = a.slab; r.bslab = a.bslab; r.B = 32; r.tiles_k = (k + 31) / 32; r.tiles = blocks_oc; r.otiles = (o + 31) / 32; r.split = (int)split;
    r.K = k; r.O = o;
    r.dW[0] = dw; r.db[0] = db; r.dW[1] = r.dW[2] = nullptr; r.db[1] = r.db[2] = nullptr;
    rl2::launch_slab_reduce(r, 1, db != nullptr, s);
    return pdf_launch_status();
}

// Several Linear layers over the same rows in one launch (channel widths 32..256, see rowlin2.hip):
//   nin = 1, nout = 1..3 :  y[i] = f(x[0]) Wt[i] + bias[i]              (q, k, v projections from one read of x)
//   nin = 1..3, nout = 1 :  y[0] (+)= sum_i x[i] Wt[i] (+ bias[0])      (input gradient of the three projections)
// Shapes the streaming kernels do not cover are issued as single-layer launches.
extern "C" int pdf_rowlin_multi(long n, int k, int o, int nin, int nout, const float *const *x, long ldx, const float *const *w,
                                int transpose_w, const float *const *bias, const float *scale, const float *shift, int relu,
                                float *const *y, long ldy, int accumulate, int mma_input, void *stream) {
    if (mma_input < 0 || mma_input > 2) return PDF_ERR_BAD_ARG;
    if (n < 1 || k < 1 || o < 1 || !x || !w || !y || nin < 1 || nout < 1 || nin > 3 || nout > 3 || (nin > 1 && nout > 1)) return PDF_ERR_BAD_ARG;
    if (rowlin_streams(k, o) && rl2::try_forward(n, k, o, nin, nout, x, ldx, w, transpose_w, bias, scale, shift, relu, y, ldy, accumulate,
                                                 nullptr, mma_input, static_cast<hipStream_t>(stream)))
        return pdf_launch_status();
    int rc = 0;
    if (nin == 1) {
        for (int i = 0; i < nout && rc == 0; ++i)
            rc = pdf_rowlin_forward(n, k, o, x[0], ldx, w[i], transpose_w, bias ? bias[i] : nullptr, scale, shift, relu, y[i], ldy, accumulate, nullptr, mma_input, stream);
    } else {
        for (int i = 0; i < nin && rc == 0; ++i)
            rc = pdf_rowlin_forward(n, k, o, x[i], ldx, w[i], transpose_w, (i == 0 && bias) ? bias[0] : nullptr, scale, shift, relu, y[0], ldy,
                                    accumulate || i > 0, nullptr, mma_input, stream);
    }
    return rc;
}

// Input gradient y = sum_i x[i] W[i] (W (k, o) row-major, nin = 1..3) of Linear layers that read a BatchNorm(+ReLU) output, with that
// BatchNorm's backward sums as an epilogue: partial receives *partial_rows rows of [sum g' | sum g' xhat] (2 o floats each; g' = y
// masked by the ReLU of bx * scale + shift, xhat = (bx - mean) * rstd; bcoef = [scale | shift | mean | rstd]) - what
// pdf_bn_act_backward's first pass would compute from a second read of y and bx.  partial must hold pdf_rowlin_partial_floats(n, o).
// PDF_ERR_UNSUPPORTED for shapes outside the streaming kernels (the caller then runs the two separate passes).
extern "C" int pdf_rowlin_dgrad_bstats(long n, int k, int o, int nin, const float *const *x, long ldx, const float *const *w, float *y, long ldy,
                                       const float *bx, long ldb, const float *bcoef, int brelu, float *partial, int *partial_rows,
                                       int mma_input, void *stream) {
    if (mma_input < 0 || mma_input > 2) return PDF_ERR_BAD_ARG;
    if (n < 1 || k < 1 || o < 1 || !x || !w || !y || nin < 1 || nin > 3 || !bx || !bcoef || !partial || !partial_rows) return PDF_ERR_BAD_ARG;
    if (!rowlin_streams(k, o)) return PDF_ERR_UNSUPPORTED;
    float *ys[1] = {y};
    if (!rl2::try_forward(n, k, o, nin, 1, x, ldx, w, 1, nullptr, nullptr, nullptr, 0, ys, ldy, 0, partial, mma_input, static_cast<hipStream_t>(stream),
                          nullptr, 0, bx, ldb, bcoef, brelu, partial_rows))
        return PDF_ERR_UNSUPPORTED;
    return pdf_launch_status();
}

// dW[i] = G[i]^T f(X), db[i] = column sums of G[i] for up to three gradients sharing the layer input X (written).
// ws: pdf_rowlin_wgrad_ws_floats(n, k, o, ng) floats.
extern "C" int pdf_rowlin_wgrad_multi(long n, int k, int o, int ng, const float *const *g, long ldg, const float *x, long ldx,
                                      const float *scale, const float *shift, int relu, float *const *dw, float *const *db,
                                      float *ws, int mma_input, void *stream) {
    if (mma_input < 0 || mma_input > 2) return PDF_ERR_BAD_ARG;
    if (n < 1 || k < 1 || o < 1 || ng < 1 || ng > 3 || !g || !x || !dw || !ws) return PDF_ERR_BAD_ARG;
    if (rowlin_streams(k, o) && rl2::try_wgrad(n, k, o, ng, g, ldg, x, ldx, scale, shift, relu, dw, db, ws, mma_input, static_cast<hipStream_t>(stream)))
        return pdf_launch_status();
    int rc = 0;   // (stream order: the next call reuses the workspace after this call's reduction has read it)
    for (int i = 0; i < ng && rc == 0; ++i) rc = pdf_rowlin_wgrad(n, k, o, g[i], ldg, x, ldx, scale, shift, relu, dw[i], db ? db[i] : nullptr, ws, mma_input, stream);
    return rc;
}

// Up to five weight gradients of ONE shape with their OWN inputs in one launch + one reduction (a Bottleneck's linear3, q / k / v and
// linear1 products): dW_i (o, k) = G_i^T f_i(X_i), f_i = relu_i?(x * scale_i + shift_i) where scale_i is non-null, else the identity;
// db_i = column sums of G_i where non-null.  Everything is WRITTEN.  Streaming shapes only (k, o in 32..512 by 32: PDF_ERR_UNSUPPORTED
// otherwise -- the caller then issues pdf_rowlin_wgrad per matrix).  ws: pdf_rowlin_wgrad_ws_floats(n, k, o, ng) floats.
extern "C" int pdf_rowlin_wgrad_group(long n, int k, int o, int ng, const float *const *g, long ldg, const float *const *x, long ldx,
                                      const float *const *scale, const float *const *shift, const int *relu, float *const *dw,
                                      float *const *db, float *ws, int mma_input, void *stream) {
    if (mma_input < 0 || mma_input > 2) return PDF_ERR_BAD_ARG;
    if (n < 1 || k < 1 || o < 1 || ng < 1 || ng > rl2::WG_MAXG || !g || !x || !scale || !shift || !relu || !dw || !ws) return PDF_ERR_BAD_ARG;
    for (int i = 0; i < ng; ++i) if (!g[i] || !x[i] || !dw[i] || (scale[i] && !shift[i])) return PDF_ERR_BAD_ARG;
    if (!rowlin_streams(k, o) || !rl2::try_wgrad_group(n, k, o, ng, g, ldg, x, ldx, scale, shift, relu, dw, db, ws, mma_input, static_cast<hipStream_t>(stream)))
        return PDF_ERR_UNSUPPORTED;
    return pdf_launch_status();
}

// Row-weighted variants for the streaming shapes (csrc/transition_down.hip: Gram matrices x^T diag(cnt) x and the dense part of
// the input gradient cnt .* (x Q)): y = roww[n] * (x Wt) (+)= ..., dW += sum_n roww[n] g[n]^T x[n].  roww has element stride rws.
extern "C" int pdf_rowlin_forward_roww(long n, int k, int o, const float *x, long ldx, const float *w, int transpose_w, float *y, long ldy,
                                       int accumulate, const float *roww, long rws, int mma_input, void *stream) {
    if (mma_input < 0 || mma_input > 2) return PDF_ERR_BAD_ARG;
    if (n < 1 || !x || !w || !y || !rowlin_streams(k, o)) return PDF_ERR_UNSUPPORTED;
    const float *bias = nullptr;
    if (!rl2::try_forward(n, k, o, 1, 1, &x, ldx, &w, transpose_w, &bias, nullptr, nullptr, 0, &y, ldy, accumulate, nullptr,
                          mma_input, static_cast<hipStream_t>(stream), roww, rws))
        return PDF_ERR_UNSUPPORTED;
    return pdf_launch_status();
}
extern "C" int pdf_rowlin_wgrad_roww(long n, int k, int o, const float *g, long ldg, const float *x, long ldx, float *dw, const float *roww,
                                     long rws, float *ws, int mma_input, void *stream) {
    if (mma_input < 0 || mma_input > 2) return PDF_ERR_BAD_ARG;
    if (n < 1 || !g || !x || !dw || !ws || !rowlin_streams(k, o)) return PDF_ERR_UNSUPPORTED;
    float *db = nullptr;
    if (!rl2::try_wgrad(n, k, o, 1, &g, ldg, x, ldx, nullptr, nullptr, 0, &dw, &db, ws, mma_input, static_cast<hipStream_t>(stream), roww, rws))
        return PDF_ERR_UNSUPPORTED;
    return pdf_launch_status();
}
