// Host-side orchestration of the Bottleneck's dense halves (point_transformer_seg.py:184-192) around the attention layer:
//
//   pre  : z1 = x W1^T                     (rowlin, epilogue = column statistics for bn1)
//          q,k,v = relu(bn1(z1)) W{q,k,v}^T + b   (rowlin with the norm folded into the prologue; relu(bn1(z1)) never hits HBM)
//   post : z3 = relu(bn2(t)) W3^T           (statistics pass on t, folded prologue, epilogue statistics for bn3)
//          y  = relu(bn3(z3) + x)
//
// One C call per half and direction: the step is launch/host bound (~3000 launches), and issuing the 6-12 kernels of a half
// from C instead of from ~30 Python-level tensor ops removes most of that host time.  Everything here is launch sequencing;
// the kernels live in rowlin.hip / pointwise.hip.  Scratch and saved activations are caller-owned (no allocation).
#include "pdfops_common.h"

namespace {
struct Err {
    int rc = 0;
    Err &operator<<(int r) { if (rc == 0 && r != 0) rc = r; return *this; }
};
}  // namespace

// coef from GEMM-epilogue partials (training) or running statistics (eval)
extern "C" int pdf_bn_coef_eval_or_partial(const float *partial, int rows, long n, int c, const float *gamma, const float *beta,
                                           float *running_mean, float *running_var, int training, float eps, float momentum,
                                           float *coef, void *stream) {
    if (training) return pdf_bn_coef_from_partial(partial, rows, n, c, gamma, beta, running_mean, running_var, eps, momentum, coef, stream);
    return pdf_bn_coef(n, c, nullptr, gamma, beta, running_mean, running_var, 0, eps, momentum, coef, nullptr, stream);
}

// p[]: x, W1, gamma1, beta1, rm1, rv1, Wq, bq, Wk, bk, Wv, bv           (inputs)
//      z1, coef1 (4c), xq, xk, xv, partial (pdf_rowlin_partial_floats(n, c))   (outputs / scratch)
extern "C" int pdf_block_pre_forward(long n, int c, void *const *p, int training, float eps, float momentum, void *stream) {
    if (n < 1 || !p) return PDF_ERR_BAD_ARG;
    if (!pdf_bn_supported(c)) return PDF_ERR_UNSUPPORTED;
    const float *x = (const float *)p[0], *W1 = (const float *)p[1], *g1 = (const float *)p[2], *b1 = (const float *)p[3];
    float *rm1 = (float *)p[4], *rv1 = (float *)p[5];
    float *z1 = (float *)p[12], *coef1 = (float *)p[13], *partial = (float *)p[17];
    Err e;
    e << pdf_rowlin_forward(n, c, c, x, c, W1, 0, nullptr, nullptr, nullptr, 0, z1, c, 0, training ? partial : nullptr, stream);
    e << pdf_bn_coef_eval_or_partial(partial, pdf_rowlin_partial_rows(n, c, c), n, c, g1, b1, rm1, rv1, training, eps, momentum, coef1, stream);
    const float *xs[1] = {z1}, *ws[3] = {(const float *)p[6], (const float *)p[8], (const float *)p[10]};
    const float *bs[3] = {(const float *)p[7], (const float *)p[9], (const float *)p[11]};
    float *ys[3] = {(float *)p[14], (float *)p[15], (float *)p[16]};
    e << pdf_rowlin_multi(n, c, c, 1, 3, xs, c, ws, 0, bs, coef1, coef1 + c, 1, ys, c, 0, stream);
    return e.rc;
}

// p[]: x, z1, coef1, W1, Wq, Wk, Wv, gxq, gxk, gxv                                  (inputs)
//      gx, grads [dW1 (c*c) | dbeta1 (c) | dgamma1 (c) | {dW (c*c), db (c)} x q,k,v]   (outputs; grads zeroed here)
//      dy (n*c), partial (pdf_bn_partial_floats(n, c))                                    (scratch; p[14] unused)
extern "C" int pdf_block_pre_backward(long n, int c, void *const *p, int training, void *stream) {
    if (n < 1 || !p) return PDF_ERR_BAD_ARG;
    const float *x = (const float *)p[0], *z1 = (const float *)p[1], *coef1 = (const float *)p[2], *W1 = (const float *)p[3];
    float *gx = (float *)p[10], *grads = (float *)p[11], *dy = (float *)p[12], *partial = (float *)p[13];
    const long cc = (long)c * c;
    float *dW1 = grads, *db1 = grads + cc, *dqkv = db1 + 2 * c;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t he = hipMemsetAsync(grads, 0, sizeof(float) * (size_t)(cc + 2 * c + 3 * (cc + c)), s);
    if (he != hipSuccess) return (int)he;
    Err e;
    const float *gs[3] = {(const float *)p[7], (const float *)p[8], (const float *)p[9]};
    const float *ws[3] = {(const float *)p[4], (const float *)p[5], (const float *)p[6]};
    float *dws[3] = {dqkv, dqkv + (cc + c), dqkv + 2 * (cc + c)}, *dbs[3] = {dqkv + cc, dqkv + (cc + c) + cc, dqkv + 2 * (cc + c) + cc};
    float *ys[1] = {dy};
    e << pdf_rowlin_multi(n, c, c, 3, 1, gs, c, ws, 1, nullptr, nullptr, nullptr, 0, ys, c, 0, stream);
    e << pdf_rowlin_wgrad_multi(n, c, c, 3, gs, c, z1, c, coef1, coef1 + c, 1, dws, dbs, stream);
    // bn1 backward in place on dy (elementwise: same index read and written); its column sums ARE [d beta1 | d gamma1]
    e << pdf_bn_act_backward(n, c, dy, z1, nullptr, coef1, training, 1, partial, db1, dy, nullptr, stream);
    e << pdf_rowlin_forward(n, c, c, dy, c, W1, 1, nullptr, nullptr, nullptr, 0, gx, c, 0, nullptr, stream);
    e << pdf_rowlin_wgrad(n, c, c, dy, c, x, c, nullptr, nullptr, 0, dW1, nullptr, stream);
    return e.rc;
}

// p[]: t, x (identity), gamma2, beta2, rm2, rv2, W3, gamma3, beta3, rm3, rv3        (inputs)
//      coef2 (4c), z3, coef3 (4c), y, partial (max(pdf_bn_partial_floats, pdf_rowlin_partial_floats))   (outputs / scratch)
extern "C" int pdf_block_post_forward(long n, int c, void *const *p, int training, float eps, float momentum, void *stream) {
    if (n < 1 || !p) return PDF_ERR_BAD_ARG;
    if (!pdf_bn_supported(c)) return PDF_ERR_UNSUPPORTED;
    const float *t = (const float *)p[0], *x = (const float *)p[1];
    float *coef2 = (float *)p[11], *z3 = (float *)p[12], *coef3 = (float *)p[13], *y = (float *)p[14], *partial = (float *)p[15];
    Err e;
    e << pdf_bn_coef(n, c, t, (const float *)p[2], (const float *)p[3], (float *)p[4], (float *)p[5], training, eps, momentum, coef2, partial, stream);
    e << pdf_rowlin_forward(n, c, c, t, c, (const float *)p[6], 0, nullptr, coef2, coef2 + c, 1, z3, c, 0, training ? partial : nullptr, stream);
    e << pdf_bn_coef_eval_or_partial(partial, pdf_rowlin_partial_rows(n, c, c), n, c, (const float *)p[7], (const float *)p[8], (float *)p[9],
                                     (float *)p[10], training, eps, momentum, coef3, stream);
    e << pdf_bn_apply(n, c, z3, x, coef3, 1, y, stream);
    return e.rc;
}

// p[]: gy, t, x, z3, coef2, coef3, W3                                               (inputs)
//      gt, gres, grads [dW3 (c*c) | dbeta2 | dgamma2 | dbeta3 | dgamma3]             (outputs; grads zeroed here)
//      da (n*c), partial                                                              (scratch; p[12] unused)
extern "C" int pdf_block_post_backward(long n, int c, void *const *p, int training, void *stream) {
    if (n < 1 || !p) return PDF_ERR_BAD_ARG;
    const float *gy = (const float *)p[0], *t = (const float *)p[1], *x = (const float *)p[2], *z3 = (const float *)p[3];
    const float *coef2 = (const float *)p[4], *coef3 = (const float *)p[5], *W3 = (const float *)p[6];
    float *gt = (float *)p[7], *gres = (float *)p[8], *grads = (float *)p[9], *da = (float *)p[10], *partial = (float *)p[11];
    const long cc = (long)c * c;
    float *dW3 = grads, *db2 = grads + cc, *db3 = db2 + 2 * c;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t he = hipMemsetAsync(grads, 0, sizeof(float) * (size_t)(cc + 4 * c), s);
    if (he != hipSuccess) return (int)he;
    Err e;
    // bn3 backward: gz3 -> da (scratch), gres; column sums land in the gradient slots [d beta | d gamma]
    e << pdf_bn_act_backward(n, c, gy, z3, x, coef3, training, 1, partial, db3, da, gres, stream);
    e << pdf_rowlin_wgrad(n, c, c, da, c, t, c, coef2, coef2 + c, 1, dW3, nullptr, stream);
    e << pdf_rowlin_forward(n, c, c, da, c, W3, 1, nullptr, nullptr, nullptr, 0, gt, c, 0, nullptr, stream);
    // bn2 backward in place on gt
    e << pdf_bn_act_backward(n, c, gt, t, nullptr, coef2, training, 1, partial, db2, gt, nullptr, stream);
    return e.rc;
}
