// Per-point normalisation layers for gfx950: train/eval BatchNorm1d over (N, C) rows fused with its ReLU and the
// residual add of the Bottleneck (point_transformer_seg.py:184-192: relu(bn1(.)), relu(bn2(.)), relu(bn3(.) + identity)).
// torch runs each of these as 2 + 1 (+1) kernels forward and 2 + 1 (+1) backward at ~1 TB/s; here a BatchNorm is
//   forward : k_bn_stats (1 read) -> k_bn_finalize (tiny) -> k_bn_apply (1-2 reads, 1 write; affine + residual + ReLU)
//   backward: k_bn_bwd_reduce (2-3 reads) -> k_colsum (tiny) -> k_bn_bwd_apply (2-3 reads, 1-2 writes)
// streaming float4 at HBM rate.  The ReLU mask and x_hat are recomputed from the saved pre-norm input, so no
// activation-sized mask or normalised copy is stored.  Bound: HBM.  Algorithmic bytes: 4NC per tensor pass.
#include "pdfops_common.h"
#include <cstdlib>

namespace pw {

constexpr int PB = 256;
constexpr int MAXB = 1024;

static inline int grid_rows(long n, int c) {
    // one thread = one float4 of a row; a block covers PB*4/c rows per sweep
    const long rows_per_block = (long)PB * 4 / c > 0 ? (long)PB * 4 / c : 1;
    // rows per lane: 8 where the tensor is large enough to be a bandwidth problem, 4 on the short levels (a lane's rows are a chain of
    // load -> accumulate round trips: at 3,124 x 256 eight of them ARE the kernel; measured 8 / 4 / 2 / 1 rows: 3.03 / 2.87 / 2.88 / 2.88 ms
    // per step for these kernels + their reducers, which read one partial row per workgroup)
    static const int short_rows = [] { const char *v = getenv("PDFOPS_BN_ROWS_SHORT"); const int x = v ? atoi(v) : 0; return x > 0 ? x : 4; }();
    const long per = n * c >= (1L << 22) ? 8 : short_rows;
    long g = (n + rows_per_block * per - 1) / (rows_per_block * per);
    if (g > MAXB) g = MAXB;
    if (g < 1) g = 1;
    return (int)g;
}

// partial[block][2c] = per-channel sum | sum of squares over the block's rows.   c % 4 == 0, c <= 1024
__global__ __launch_bounds__(PB) void k_bn_stats(long n, int c, const float *__restrict__ x, float *partial) {
    extern __shared__ float red[];  // [PB][8]
    const int tpr = c / 4;                    // threads per row
    const int rpb = PB / tpr;                 // rows per block sweep (c <= 1024 -> tpr <= 256)
    const int tr = threadIdx.x / tpr, tc = threadIdx.x % tpr;
    float s[4] = {0.f, 0.f, 0.f, 0.f}, ss[4] = {0.f, 0.f, 0.f, 0.f};
    if (tr < rpb) {
        for (long r = (long)blockIdx.x * rpb + tr; r < n; r += (long)gridDim.x * rpb) {
            const float4 v = *reinterpret_cast<const float4 *>(x + r * c + tc * 4);
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
            ss[0] += v.x * v.x; ss[1] += v.y * v.y; ss[2] += v.z * v.z; ss[3] += v.w * v.w;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[threadIdx.x * 8 + k] = s[k]; red[threadIdx.x * 8 + 4 + k] = ss[k]; }
    __syncthreads();
    if (threadIdx.x < tpr) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float a = 0.f;
            for (int rr = 0; rr < rpb; ++rr) a += red[(rr * tpr + threadIdx.x) * 8 + k];
            const int ch = threadIdx.x * 4 + (k & 3);
            partial[(size_t)blockIdx.x * 2 * c + (k < 4 ? ch : c + ch)] = a;
        }
    }
}

// y = [relu]( x * scale + shift [+ res] )
// (RES as a template parameter: a load behind `if (res)` waits for the loads before it and is waited for on its own)
template <bool RES>
__global__ __launch_bounds__(PB) void k_bn_apply(long n4, int c4, const float4 *__restrict__ x, const float4 *__restrict__ scale,
                                                 const float4 *__restrict__ shift, const float4 *__restrict__ res, int relu,
                                                 float4 *__restrict__ y) {
    for (long e = (long)blockIdx.x * PB + threadIdx.x; e < n4; e += (long)gridDim.x * PB) {
        const int cc = (int)(e % c4);
        const float4 v = x[e], sc = scale[cc], sh = shift[cc];
        const float4 r = RES ? res[e] : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 o = make_float4(v.x * sc.x + sh.x + r.x, v.y * sc.y + sh.y + r.y, v.z * sc.z + sh.z + r.z, v.w * sc.w + sh.w + r.w);
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        y[e] = o;
    }
}

// The same with the BatchNorm's finalize inside this kernel (pdfops_common.h: pdf_bn_coef_inkernel): the statistics arrive as the
// producer's partial rows, the first workgroups reduce them, everyone picks the coefficients up through the granules.
template <bool RES>
__global__ __launch_bounds__(PB) void k_bn_apply_rows(long n4, int c4, const float4 *__restrict__ x, PdfRowsBn b, const float4 *__restrict__ res,
                                                      int relu, float4 *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // scale | shift (2 c) | reducer scratch | ticket word
    const int c = 4 * c4;
    double *red = reinterpret_cast<double *>(lds + 2 * c);
    unsigned *word = reinterpret_cast<unsigned *>(red + 2 * (PB / 16) * 17);
    pdf_bn_coef_inkernel<PB>(b, lds, red, word);
    const float4 *scale = reinterpret_cast<const float4 *>(lds), *shift = reinterpret_cast<const float4 *>(lds + c);
    for (long e = (long)blockIdx.x * PB + threadIdx.x; e < n4; e += (long)gridDim.x * PB) {
        const int cc = (int)(e % c4);
        const float4 v = x[e], sc = scale[cc], sh = shift[cc];
        const float4 r = RES ? res[e] : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 o = make_float4(v.x * sc.x + sh.x + r.x, v.y * sc.y + sh.y + r.y, v.z * sc.z + sh.z + r.z, v.w * sc.w + sh.w + r.w);
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        y[e] = o;
    }
}

// partial[block][2c] = sum g' | sum g' * xhat,   g' = gy masked by the ReLU (recomputed from x, res)
template <bool RES>
__global__ __launch_bounds__(PB) void k_bn_bwd_reduce(long n, int c, const float *__restrict__ gy, const float *__restrict__ x,
                                                      const float *__restrict__ res, const float *__restrict__ scale,
                                                      const float *__restrict__ shift, const float *__restrict__ mean,
                                                      const float *__restrict__ rstd, int relu, float *partial) {
    extern __shared__ float red[];
    const int tpr = c / 4, rpb = PB / tpr;
    const int tr = threadIdx.x / tpr, tc = threadIdx.x % tpr;
    float s[4] = {0.f, 0.f, 0.f, 0.f}, sx[4] = {0.f, 0.f, 0.f, 0.f};
    if (tr < rpb) {
        float sc[4], sh[4], mu[4], rs[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { sc[k] = scale[tc * 4 + k]; sh[k] = shift[tc * 4 + k]; mu[k] = mean[tc * 4 + k]; rs[k] = rstd[tc * 4 + k]; }
        for (long r = (long)blockIdx.x * rpb + tr; r < n; r += (long)gridDim.x * rpb) {
            const float4 g4 = *reinterpret_cast<const float4 *>(gy + r * c + tc * 4);
            const float4 x4 = *reinterpret_cast<const float4 *>(x + r * c + tc * 4);
            const float4 r4 = RES ? *reinterpret_cast<const float4 *>(res + r * c + tc * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float g[4] = {g4.x, g4.y, g4.z, g4.w}, xv[4] = {x4.x, x4.y, x4.z, x4.w}, rv[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float pre = xv[k] * sc[k] + sh[k] + rv[k];
                const float gm = (!relu || pre > 0.f) ? g[k] : 0.f;
                s[k] += gm;
                sx[k] += gm * ((xv[k] - mu[k]) * rs[k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[threadIdx.x * 8 + k] = s[k]; red[threadIdx.x * 8 + 4 + k] = sx[k]; }
    __syncthreads();
    if (threadIdx.x < tpr) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float a = 0.f;
            for (int rr = 0; rr < rpb; ++rr) a += red[(rr * tpr + threadIdx.x) * 8 + k];
            const int ch = threadIdx.x * 4 + (k & 3);
            partial[(size_t)blockIdx.x * 2 * c + (k < 4 ? ch : c + ch)] = a;
        }
    }
}

// gx = scale * (g' - sums[ch]/n - xhat * sums[c+ch]/n) ; gres = g'
template <bool RES>
__global__ __launch_bounds__(PB) void k_bn_bwd_apply(long n4, int c4, const float4 *__restrict__ gy, const float4 *__restrict__ x,
                                                     const float4 *__restrict__ res, const float4 *__restrict__ scale,
                                                     const float4 *__restrict__ shift, const float4 *__restrict__ mean,
                                                     const float4 *__restrict__ rstd, const float4 *__restrict__ sums, float inv_n,
                                                     int relu, float4 *__restrict__ gx, float4 *__restrict__ gres) {
    for (long e = (long)blockIdx.x * PB + threadIdx.x; e < n4; e += (long)gridDim.x * PB) {
        const int cc = (int)(e % c4);
        const float4 g4 = gy[e], x4 = x[e], sc = scale[cc], sh = shift[cc], mu = mean[cc], rs = rstd[cc];
        const float4 s1 = sums[cc], s2 = sums[c4 + cc];
        const float4 r4 = RES ? res[e] : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 gm, o;
#define PW_ONE(f)                                                                     \
        {                                                                             \
            const float pre = x4.f * sc.f + sh.f + r4.f;                              \
            gm.f = (!relu || pre > 0.f) ? g4.f : 0.f;                                 \
            const float xh = (x4.f - mu.f) * rs.f;                                    \
            o.f = sc.f * (gm.f - s1.f * inv_n - xh * s2.f * inv_n);                   \
        }
        PW_ONE(x) PW_ONE(y) PW_ONE(z) PW_ONE(w)
#undef PW_ONE
        gx[e] = o;
        if (gres) gres[e] = gm;
    }
}

// eval-mode backward: gx = scale * g' (running statistics are constants)
template <bool RES>
__global__ __launch_bounds__(PB) void k_bn_bwd_eval(long n4, int c4, const float4 *__restrict__ gy, const float4 *__restrict__ x,
                                                    const float4 *__restrict__ res, const float4 *__restrict__ scale,
                                                    const float4 *__restrict__ shift, int relu, float4 *__restrict__ gx,
                                                    float4 *__restrict__ gres) {
    for (long e = (long)blockIdx.x * PB + threadIdx.x; e < n4; e += (long)gridDim.x * PB) {
        const int cc = (int)(e % c4);
        const float4 g4 = gy[e], x4 = x[e], sc = scale[cc], sh = shift[cc];
        const float4 r4 = RES ? res[e] : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 gm;
        gm.x = (!relu || x4.x * sc.x + sh.x + r4.x > 0.f) ? g4.x : 0.f;
        gm.y = (!relu || x4.y * sc.y + sh.y + r4.y > 0.f) ? g4.y : 0.f;
        gm.z = (!relu || x4.z * sc.z + sh.z + r4.z > 0.f) ? g4.z : 0.f;
        gm.w = (!relu || x4.w * sc.w + sh.w + r4.w > 0.f) ? g4.w : 0.f;
        gx[e] = make_float4(gm.x * sc.x, gm.y * sc.y, gm.z * sc.z, gm.w * sc.w);
        if (gres) gres[e] = gm;
    }
}

static inline int grid_elems(long n4) {
    long g = (n4 + PB - 1) / PB;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace pw

namespace fl {  // defined in fused_layer.hip
void launch_bn_finalize(const float *partial, int rows, int nch, double count, const float *gamma, const float *beta, float eps,
                        float momentum, float *running_mean, float *running_var, float *scale, float *shift, float *mean_out,
                        float *rstd_out, hipStream_t s);
void launch_bn_eval(int nch, const float *gamma, const float *beta, float eps, const float *running_mean, const float *running_var,
                    float *scale, float *shift, float *mean_out, float *rstd_out, hipStream_t s);
void launch_colsum(const float *partial, int rows, int width, float *out, hipStream_t s);
}  // namespace fl

extern "C" int pdf_bn_supported(int c) { return c >= 4 && c % 4 == 0 && c <= 1024 && (1024 % c == 0); }
extern "C" long pdf_bn_partial_floats(long n, int c) { return (long)pw::grid_rows(n, c) * 2 * c + (long)PDF_HO_FLOATS; }   // rows [grid][2c] (+ handoff scratch)

// Forward of BatchNorm1d (+residual)(+ReLU) over (n, c).  coef (4c floats) receives [scale | shift | mean | rstd].
// training: batch statistics (running stats updated when non-null); else running statistics.
extern "C" int pdf_bn_act_forward(long n, int c, const float *x, const float *res, const float *gamma, const float *beta,
                                  float *running_mean, float *running_var, int training, float eps, float momentum,
                                  int relu, float *coef, float *partial, float *y, void *stream) {
    if (n < 1 || !x || !gamma || !beta || !coef || !y) return PDF_ERR_BAD_ARG;
    if (!pdf_bn_supported(c)) return PDF_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    float *scale = coef, *shift = coef + c, *mean = coef + 2 * c, *rstd = coef + 3 * c;
    if (training) {
        if (!partial) return PDF_ERR_BAD_ARG;
        const int g = pw::grid_rows(n, c);
        pw::k_bn_stats<<<g, pw::PB, pw::PB * 8 * sizeof(float), s>>>(n, c, x, partial);
        fl::launch_bn_finalize(partial, g, c, (double)n, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, mean, rstd, s);
    } else {
        if (!running_mean || !running_var) return PDF_ERR_BAD_ARG;
        fl::launch_bn_eval(c, gamma, beta, eps, running_mean, running_var, scale, shift, mean, rstd, s);
    }
    const long n4 = n * (c / 4);
    (res ? pw::k_bn_apply<true> : pw::k_bn_apply<false>)<<<pw::grid_elems(n4), pw::PB, 0, s>>>(n4, c / 4, reinterpret_cast<const float4 *>(x),
                                                        reinterpret_cast<const float4 *>(scale), reinterpret_cast<const float4 *>(shift),
                                                        reinterpret_cast<const float4 *>(res), relu, reinterpret_cast<float4 *>(y));
    return pdf_launch_status();
}

// Backward.  sums (2c floats) receives [d beta | d gamma] = [sum g' | sum g' * xhat]; gres may be null.
// presummed > 0: partial already holds that many rows of [sum g' | sum g' xhat] (epilogue of pdf_rowlin_dgrad_bstats); the reduction pass is skipped.
static int bn_act_backward(long n, int c, const float *gy, const float *x, const float *res, const float *coef,
                           int training, int relu, float *partial, float *sums, int sums_zeroed, float *gx, float *gres, void *stream,
                           int presummed = 0) {
    (void)sums_zeroed;
    if (n < 1 || !gy || !x || !coef || !partial || !sums || !gx) return PDF_ERR_BAD_ARG;
    if (!pdf_bn_supported(c)) return PDF_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float *scale = coef, *shift = coef + c, *mean = coef + 2 * c, *rstd = coef + 3 * c;
    const int g = pw::grid_rows(n, c);
    const long n4 = n * (c / 4);
    if (presummed > 0) {
        fl::launch_colsum(partial, presummed, 2 * c, sums, s);
    } else {
        (res ? pw::k_bn_bwd_reduce<true> : pw::k_bn_bwd_reduce<false>)<<<g, pw::PB, pw::PB * 8 * sizeof(float), s>>>(
            n, c, gy, x, res, scale, shift, mean, rstd, relu, partial);
        fl::launch_colsum(partial, g, 2 * c, sums, s);
    }
#define F4(p) reinterpret_cast<const float4 *>(p)
    if (training)
        (res ? pw::k_bn_bwd_apply<true> : pw::k_bn_bwd_apply<false>)<<<pw::grid_elems(n4), pw::PB, 0, s>>>(n4, c / 4, F4(gy), F4(x), F4(res), F4(scale), F4(shift), F4(mean), F4(rstd),
                                                                F4(sums), (float)(1.0 / (double)n), relu, reinterpret_cast<float4 *>(gx),
                                                                reinterpret_cast<float4 *>(gres));
    else
        (res ? pw::k_bn_bwd_eval<true> : pw::k_bn_bwd_eval<false>)<<<pw::grid_elems(n4), pw::PB, 0, s>>>(n4, c / 4, F4(gy), F4(x), F4(res), F4(scale), F4(shift), relu,
                                                               reinterpret_cast<float4 *>(gx), reinterpret_cast<float4 *>(gres));
#undef F4
    return pdf_launch_status();
}

extern "C" int pdf_bn_act_backward(long n, int c, const float *gy, const float *x, const float *res, const float *coef,
                                   int training, int relu, float *partial, float *sums, float *gx, float *gres, void *stream) {
    return bn_act_backward(n, c, gy, x, res, coef, training, relu, partial, sums, 0, gx, gres, stream);
}
// same with the reduction pass already done by the producer of gy (pdf_rowlin_dgrad_bstats): partial holds partial_rows rows of 2c sums
extern "C" int pdf_bn_act_backward_presummed(long n, int c, const float *gy, const float *x, const float *coef, int training, int relu,
                                             const float *partial, int partial_rows, float *sums, float *gx, void *stream) {
    if (partial_rows < 1 || !partial) return PDF_ERR_BAD_ARG;
    return bn_act_backward(n, c, gy, x, nullptr, coef, training, relu, const_cast<float *>(partial), sums, 0, gx, nullptr, stream, partial_rows);
}
// Coefficients of a train-mode BatchNorm whose column statistics were produced by a GEMM epilogue (rowlin STATS).
extern "C" int pdf_bn_coef_from_partial(const float *partial, int rows, long n, int c, const float *gamma, const float *beta,
                                        float *running_mean, float *running_var, float eps, float momentum, float *coef,
                                        void *stream) {
    if (!partial || rows < 1 || n < 1 || c < 1 || !gamma || !beta || !coef) return PDF_ERR_BAD_ARG;
    fl::launch_bn_finalize(partial, rows, c, (double)n, gamma, beta, eps, momentum, running_mean, running_var, coef, coef + c,
                           coef + 2 * c, coef + 3 * c, static_cast<hipStream_t>(stream));
    return pdf_launch_status();
}

// y = relu?(x * scale + shift + res) with given coefficients (no statistics pass)
extern "C" int pdf_bn_apply(long n, int c, const float *x, const float *res, const float *coef, int relu, float *y, void *stream) {
    if (n < 1 || !x || !coef || !y) return PDF_ERR_BAD_ARG;
    if (!pdf_bn_supported(c)) return PDF_ERR_UNSUPPORTED;
    const long n4 = n * (c / 4);
    (res ? pw::k_bn_apply<true> : pw::k_bn_apply<false>)<<<pw::grid_elems(n4), pw::PB, 0, static_cast<hipStream_t>(stream)>>>(
        n4, c / 4, reinterpret_cast<const float4 *>(x), reinterpret_cast<const float4 *>(coef), reinterpret_cast<const float4 *>(coef + c),
        reinterpret_cast<const float4 *>(res), relu, reinterpret_cast<float4 *>(y));
    return pdf_launch_status();
}

// y = relu?(bn(x) + res) for a train-mode BatchNorm whose statistics are `nrows` partial rows [sum | sum of squares] of a producer that
// zeroed `handoff` (PDF_HO_WORDS(c) 32-bit words): finalize + apply in ONE launch; coef (4 c) is written for the backward.
int pdf_bn_apply_rows(long n, int c, const float *x, const float *res, const float *rows, int nrows, const float *gamma, const float *beta,
                      float *running_mean, float *running_var, float eps, float momentum, float *coef, void *handoff, int relu, float *y,
                      void *stream) {
    if (n < 1 || !x || !rows || nrows < 1 || !gamma || !beta || !coef || !handoff || !y) return PDF_ERR_BAD_ARG;
    if (!pdf_bn_supported(c)) return PDF_ERR_UNSUPPORTED;
    PdfRowsBn b;
    b.rows = rows; b.nrows = nrows; b.c = c; b.count = (double)n; b.gamma = gamma; b.beta = beta; b.running_mean = running_mean;
    b.running_var = running_var; b.eps = eps; b.momentum = momentum; b.coef = coef;
    b.gran = static_cast<unsigned long long *>(handoff); b.sync = reinterpret_cast<unsigned *>(b.gran + 2 * (size_t)c);
    const long n4 = n * (c / 4);
    const size_t lds = sizeof(float) * 2 * c + sizeof(double) * 2 * (pw::PB / 16) * 17 + 16;
    (res ? pw::k_bn_apply_rows<true> : pw::k_bn_apply_rows<false>)<<<pw::grid_elems(n4), pw::PB, lds, static_cast<hipStream_t>(stream)>>>(
        n4, c / 4, reinterpret_cast<const float4 *>(x), b, reinterpret_cast<const float4 *>(res), relu, reinterpret_cast<float4 *>(y));
    return pdf_launch_status();
}

// Coefficients (scale|shift|mean|rstd) of a BatchNorm over (n, c) rows without applying it: statistics pass + finalize
// (training) or running statistics (eval).  Used when the affine + ReLU is folded into the consumer GEMM's prologue.
extern "C" int pdf_bn_coef(long n, int c, const float *x, const float *gamma, const float *beta, float *running_mean,
                           float *running_var, int training, float eps, float momentum, float *coef, float *partial, void *stream) {
    if (n < 1 || !gamma || !beta || !coef) return PDF_ERR_BAD_ARG;
    if (!pdf_bn_supported(c)) return PDF_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (training) {
        if (!partial || !x) return PDF_ERR_BAD_ARG;
        const int g = pw::grid_rows(n, c);
        pw::k_bn_stats<<<g, pw::PB, pw::PB * 8 * sizeof(float), s>>>(n, c, x, partial);
        fl::launch_bn_finalize(partial, g, c, (double)n, gamma, beta, eps, momentum, running_mean, running_var, coef, coef + c, coef + 2 * c, coef + 3 * c, s);
    } else {
        if (!running_mean || !running_var) return PDF_ERR_BAD_ARG;
        fl::launch_bn_eval(c, gamma, beta, eps, running_mean, running_var, coef, coef + c, coef + 2 * c, coef + 3 * c, s);
    }
    return pdf_launch_status();
}

// Only the two BatchNorm-backward column sums [sum g' | sum g' * xhat] (g' = gy masked by the ReLU of x * scale + shift), no input
// gradient: used by csrc/transition_down.hip with coefficients that express the max-pooled output.
extern "C" int pdf_bn_bwd_sums(long n, int c, const float *gy, const float *x, const float *coef, int relu, float *partial, float *sums,
                               void *stream) {
    if (n < 1 || !gy || !x || !coef || !partial || !sums) return PDF_ERR_BAD_ARG;
    if (!pdf_bn_supported(c)) return PDF_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int g = pw::grid_rows(n, c);
    pw::k_bn_bwd_reduce<false><<<g, pw::PB, pw::PB * 8 * sizeof(float), s>>>(n, c, gy, x, nullptr, coef, coef + c, coef + 2 * c, coef + 3 * c, relu, partial);
    fl::launch_colsum(partial, g, 2 * c, sums, s);
    return pdf_launch_status();
}
