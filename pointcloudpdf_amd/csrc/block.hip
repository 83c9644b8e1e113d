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
extern "C" int pdf_block_pre_forward(long n, int c, void *const *p, int training, float eps, float momentum, int mma_input, void *stream) {
    if (n < 1 || !p) return PDF_ERR_BAD_ARG;
    if (!pdf_bn_supported(c)) return PDF_ERR_UNSUPPORTED;
    const float *x = (const float *)p[0], *W1 = (const float *)p[1], *g1 = (const float *)p[2], *b1 = (const float *)p[3];
    float *rm1 = (float *)p[4], *rv1 = (float *)p[5];
    float *z1 = (float *)p[12], *coef1 = (float *)p[13], *partial = (float *)p[17];
    Err e;
    if (training) {
        e << pdf_rowlin_forward_bn(n, c, c, x, c, W1, nullptr, nullptr, nullptr, 0, z1, c, partial, g1, b1, rm1, rv1, eps, momentum, coef1, mma_input, stream);
    } else {
        e << pdf_rowlin_forward(n, c, c, x, c, W1, 0, nullptr, nullptr, nullptr, 0, z1, c, 0, nullptr, mma_input, stream);
        e << pdf_bn_coef_eval_or_partial(partial, 0, n, c, g1, b1, rm1, rv1, 0, eps, momentum, coef1, stream);
    }
    const float *xs[1] = {z1}, *ws[3] = {(const float *)p[6], (const float *)p[8], (const float *)p[10]};
    const float *bs[3] = {(const float *)p[7], (const float *)p[9], (const float *)p[11]};
    float *ys[3] = {(float *)p[14], (float *)p[15], (float *)p[16]};
    e << pdf_rowlin_multi(n, c, c, 1, 3, xs, c, ws, 0, bs, coef1, coef1 + c, 1, ys, c, 0, mma_input, stream);
    return e.rc;
}

// p[]: x, z1, coef1, W1, Wq, Wk, Wv, gxq, gxk, gxv                                  (inputs)
//      gx, grads [dW1 (c*c) | dbeta1 (c) | dgamma1 (c) | {dW (c*c), db (c)} x q,k,v]   (outputs; every element is WRITTEN: no zeroing)
//      dy (n*c), partial (max(pdf_bn_partial_floats(n, c), pdf_rowlin_partial_floats(n, c)): the dgrad epilogue's rows)   (scratch)
//      p[14] = workspace of the weight-gradient slabs, pdf_rowlin_wgrad_ws_floats(n, c, c, 3) floats
static bool inkernel_bn() {   // PDFOPS_INKERNEL_BN=0: every BatchNorm finalize as its own reducer launch (rounds 1-3)
    static const bool on = [] { const char *v = getenv("PDFOPS_INKERNEL_BN"); return !(v && v[0] == '0'); }();
    return on;
}
static bool dgrad_bstats() {   // PDFOPS_DGRAD_BSTATS=0: BatchNorm-backward sums from their own pass instead of the dgrad epilogue (A/B)
    static const bool on = [] { const char *v = getenv("PDFOPS_DGRAD_BSTATS"); return !(v && v[0] == '0'); }();
    return on;
}

// defer_wgrad: the three weight-gradient products (q / k / v, linear1) are left to the caller (pdf_bottleneck_backward issues them together
// with linear3's in ONE grouped launch); their inputs -- g_xq / g_xk / g_xv, z1 + coef1, dy, x -- stay untouched until then.
static int block_pre_backward(long n, int c, void *const *p, int training, int accumulate_gx, int mma_input, void *stream, bool defer_wgrad = false) {
    if (n < 1 || !p || !p[14]) return PDF_ERR_BAD_ARG;
    const float *x = (const float *)p[0], *z1 = (const float *)p[1], *coef1 = (const float *)p[2], *W1 = (const float *)p[3];
    float *gx = (float *)p[10], *grads = (float *)p[11], *dy = (float *)p[12], *partial = (float *)p[13], *wslab = (float *)p[14];
    const long cc = (long)c * c;
    float *dW1 = grads, *db1 = grads + cc, *dqkv = db1 + 2 * c;
    Err e;
    const float *gs[3] = {(const float *)p[7], (const float *)p[8], (const float *)p[9]};
    const float *ws[3] = {(const float *)p[4], (const float *)p[5], (const float *)p[6]};
    float *dws[3] = {dqkv, dqkv + (cc + c), dqkv + 2 * (cc + c)}, *dbs[3] = {dqkv + cc, dqkv + (cc + c) + cc, dqkv + 2 * (cc + c) + cc};
    float *ys[1] = {dy};
    if (!defer_wgrad) e << pdf_rowlin_wgrad_multi(n, c, c, 3, gs, c, z1, c, coef1, coef1 + c, 1, dws, dbs, wslab, mma_input, stream);   // g_xq / g_xk / g_xv are final here
    // bn1 backward in place on dy (elementwise: same index read and written); its column sums ARE [d beta1 | d gamma1].  The sums come
    // out of the input-gradient product's epilogue where the streaming kernel covers the shape, else from a pass over dy and z1.
    int prow = 0;
    const int rc1 = dgrad_bstats() ? pdf_rowlin_dgrad_bstats(n, c, c, 3, gs, c, ws, dy, c, z1, c, coef1, 1, partial, &prow, mma_input, stream) : PDF_ERR_UNSUPPORTED;
    if (rc1 == PDF_ERR_UNSUPPORTED) {
        e << pdf_rowlin_multi(n, c, c, 3, 1, gs, c, ws, 1, nullptr, nullptr, nullptr, 0, ys, c, 0, mma_input, stream);
        e << pdf_bn_act_backward(n, c, dy, z1, nullptr, coef1, training, 1, partial, db1, dy, nullptr, stream);   // (the atomic variant: 782 blocks on 64 addresses, +13 us)
    } else {
        e << rc1;
        e << pdf_bn_act_backward_presummed(n, c, dy, z1, coef1, training, 1, partial, prow, db1, dy, stream);
    }
    if (!defer_wgrad) e << pdf_rowlin_wgrad(n, c, c, dy, c, x, c, nullptr, nullptr, 0, dW1, nullptr, wslab, mma_input, stream);   // dW1 needs the finished dy (after the q/k/v reduction in stream order)
    e << pdf_rowlin_forward(n, c, c, dy, c, W1, 1, nullptr, nullptr, nullptr, 0, gx, c, accumulate_gx, nullptr, mma_input, stream);
    return e.rc;
}

extern "C" int pdf_block_pre_backward(long n, int c, void *const *p, int training, int mma_input, void *stream) {
    return block_pre_backward(n, c, p, training, 0, mma_input, stream, false);
}

// p[]: t, x (identity), gamma2, beta2, rm2, rv2, W3, gamma3, beta3, rm3, rv3        (inputs)
//      coef2 (4c), z3, coef3 (4c), y, partial (max(pdf_bn_partial_floats, pdf_rowlin_partial_floats))   (outputs / scratch)
// t_stat_rows > 0: `partial` already holds that many rows [sum t | sum t^2] (epilogue of the attention layer's last pass,
// pdf_pt_layer_forward_m): bn2's coefficients come from them instead of a statistics pass over t.
static int block_post_forward(long n, int c, void *const *p, int training, float eps, float momentum, int t_stat_rows, int mma_input, void *stream) {
    if (n < 1 || !p) return PDF_ERR_BAD_ARG;
    if (!pdf_bn_supported(c)) return PDF_ERR_UNSUPPORTED;
    const float *t = (const float *)p[0], *x = (const float *)p[1];
    float *coef2 = (float *)p[11], *z3 = (float *)p[12], *coef3 = (float *)p[13], *y = (float *)p[14], *partial = (float *)p[15];
    Err e;
    if (training && t_stat_rows > 0)
        e << pdf_bn_coef_from_partial(partial, t_stat_rows, n, c, (const float *)p[2], (const float *)p[3], (float *)p[4], (float *)p[5], eps, momentum, coef2, stream);
    else
        e << pdf_bn_coef(n, c, t, (const float *)p[2], (const float *)p[3], (float *)p[4], (float *)p[5], training, eps, momentum, coef2, partial, stream);
    bool applied = false;
    if (training && inkernel_bn()) {   // z3 product (+ statistics rows) -> ONE launch: bn3 finalize + affine + residual + ReLU
        int rows = 0;
        const int rc = pdf_rowlin_forward_stats_ho(n, c, c, t, c, (const float *)p[6], nullptr, coef2, coef2 + c, 1, z3, c, partial + PDF_HO_FLOATS, partial,
                                                   &rows, mma_input, stream);
        if (rc == PDF_OK) {
            e << pdf_bn_apply_rows(n, c, z3, x, partial + PDF_HO_FLOATS, rows, (const float *)p[7], (const float *)p[8], (float *)p[9], (float *)p[10], eps,
                                   momentum, coef3, partial, 1, y, stream);
            applied = true;
        } else if (rc != PDF_ERR_UNSUPPORTED) {
            e << rc;
        }
    }
    if (applied) return e.rc;
    if (training) {
        e << pdf_rowlin_forward_bn(n, c, c, t, c, (const float *)p[6], nullptr, coef2, coef2 + c, 1, z3, c, partial, (const float *)p[7], (const float *)p[8],
                                   (float *)p[9], (float *)p[10], eps, momentum, coef3, mma_input, stream);
    } else {
        e << pdf_rowlin_forward(n, c, c, t, c, (const float *)p[6], 0, nullptr, coef2, coef2 + c, 1, z3, c, 0, nullptr, mma_input, stream);
        e << pdf_bn_coef_eval_or_partial(partial, 0, n, c, (const float *)p[7], (const float *)p[8], (float *)p[9], (float *)p[10], 0, eps, momentum, coef3, stream);
    }
    e << pdf_bn_apply(n, c, z3, x, coef3, 1, y, stream);
    return e.rc;
}
extern "C" int pdf_block_post_forward(long n, int c, void *const *p, int training, float eps, float momentum, int mma_input, void *stream) {
    return block_post_forward(n, c, p, training, eps, momentum, 0, mma_input, stream);
}

// p[]: gy, t, x, z3, coef2, coef3, W3                                               (inputs)
//      gt, gres, grads [dW3 (c*c) | dbeta2 | dgamma2 | dbeta3 | dgamma3]             (outputs; every element is WRITTEN: no zeroing)
//      da (n*c), partial (max(pdf_bn_partial_floats(n, c), pdf_rowlin_partial_floats(n, c)))   (scratch)
//      p[12] = workspace of the weight-gradient slabs, pdf_rowlin_wgrad_ws_floats(n, c, c, 1) floats
static int block_post_backward(long n, int c, void *const *p, int training, int mma_input, void *stream, bool defer_wgrad = false) {
    if (n < 1 || !p || !p[12]) return PDF_ERR_BAD_ARG;
    const float *gy = (const float *)p[0], *t = (const float *)p[1], *x = (const float *)p[2], *z3 = (const float *)p[3];
    const float *coef2 = (const float *)p[4], *coef3 = (const float *)p[5], *W3 = (const float *)p[6];
    float *gt = (float *)p[7], *gres = (float *)p[8], *grads = (float *)p[9], *da = (float *)p[10], *partial = (float *)p[11];
    const long cc = (long)c * c;
    float *dW3 = grads, *db2 = grads + cc, *db3 = db2 + 2 * c;
    Err e;
    // bn3 backward: gz3 -> da (scratch), gres; column sums land in the gradient slots [d beta | d gamma]
    e << pdf_bn_act_backward(n, c, gy, z3, x, coef3, training, 1, partial, db3, da, gres, stream);
    if (!defer_wgrad) e << pdf_rowlin_wgrad(n, c, c, da, c, t, c, coef2, coef2 + c, 1, dW3, nullptr, (float *)p[12], mma_input, stream);
    // bn2 backward in place on gt (sums from the product's epilogue, as in block_pre_backward)
    int prow = 0;
    const float *das[1] = {da}, *w3s[1] = {W3};
    const int rc2 = dgrad_bstats() ? pdf_rowlin_dgrad_bstats(n, c, c, 1, das, c, w3s, gt, c, t, c, coef2, 1, partial, &prow, mma_input, stream) : PDF_ERR_UNSUPPORTED;
    if (rc2 == PDF_ERR_UNSUPPORTED) {
        e << pdf_rowlin_forward(n, c, c, da, c, W3, 1, nullptr, nullptr, nullptr, 0, gt, c, 0, nullptr, mma_input, stream);
        e << pdf_bn_act_backward(n, c, gt, t, nullptr, coef2, training, 1, partial, db2, gt, nullptr, stream);
    } else {
        e << rc2;
        e << pdf_bn_act_backward_presummed(n, c, gt, t, coef2, training, 1, partial, prow, db2, gt, stream);
    }
    return e.rc;
}

extern "C" int pdf_block_post_backward(long n, int c, void *const *p, int training, int mma_input, void *stream) {
    return block_post_backward(n, c, p, training, mma_input, stream, false);
}

// ---------------------------------------------------------------------------------------------------------------------
// The whole Bottleneck (point_transformer_seg.py:184-192: linear1-bn1-relu, PointTransformerLayer, bn2-relu, linear3-bn3,
// + identity, relu) as ONE host call per direction: pre half, fused attention layer (fused_layer*.hip), post half.
// Forward table p[]:
//    0 x | 1 W1 2 gamma1 3 beta1 4 rm1 5 rv1 | 6 Wq 7 bq 8 Wk 9 bk 10 Wv 11 bv | 12 coord 13 knn idx
//   14-21 layer weights (Wp1 bp1 Wp2 bp2 Ww1 bw1 Ww2 bw2) | 22-27 layer norm params (g_p b_p g_1 b_1 g_2 b_2)
//   28-33 layer norm buffers (rm/rv x3) | 34 gamma2 35 beta2 36 rm2 37 rv2 | 38 W3 | 39 gamma3 40 beta3 41 rm3 42 rv3
//   saved / outputs: 43 z1 44 coef1 45 xq 46 xk 47 xv 48 layer bn (2T) 49 layer saved (2T) 50 H 51 t 52 coef2 53 z3 54 coef3 55 y
//   56 scratch (max of pdf_rowlin_partial_floats, pdf_bn_partial_floats, pdf_pt_layer_partial_floats) | 57 visiting order of the points (or null)
//   58 the batch's relative-coordinate sums of the kNN table (9 doubles, pdf_knn_rel_moments) or null
extern "C" int pdf_bottleneck_forward(long n, int nsample, int c, void *const *p, int training, float eps, float momentum, int storage_bf16, int mma_input, void *stream) {
    if (n < 1 || !p) return PDF_ERR_BAD_ARG;
    Err e;
    void *pre[18] = {p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10], p[11], p[43], p[44], p[45], p[46], p[47], p[56]};
    e << pdf_block_pre_forward(n, c, pre, training, eps, momentum, mma_input, stream);
    const float *weights[8], *bn_params[6];
    float *bn_buffers[6];
    for (int i = 0; i < 8; ++i) weights[i] = (const float *)p[14 + i];
    for (int i = 0; i < 6; ++i) { bn_params[i] = (const float *)p[22 + i]; bn_buffers[i] = (float *)p[28 + i]; }
    // bn2's statistics as an epilogue of the attention layer's last pass (PDFOPS_P4_STATS=0: their own pass over t, as in rounds 1-3)
    static const bool p4_stats = [] { const char *v = getenv("PDFOPS_P4_STATS"); return !(v && v[0] == '0'); }();
    int t_rows = 0;
    e << pdf_pt_layer_forward_m((int)n, nsample, c, (const float *)p[45], (const float *)p[46], (const float *)p[47], (const float *)p[12],
                                (const int *)p[13], weights, bn_params, bn_buffers, training, eps, momentum, (float *)p[48], (float *)p[49],
                                (float *)p[50], (float *)p[56], (float *)p[51], storage_bf16, (const int *)p[57], (const double *)p[58],
                                p4_stats ? &t_rows : nullptr, stream);
    void *post[16] = {p[51], p[0], p[34], p[35], p[36], p[37], p[38], p[39], p[40], p[41], p[42], p[52], p[53], p[54], p[55], p[56]};
    e << block_post_forward(n, c, post, training, eps, momentum, e.rc == 0 ? t_rows : 0, mma_input, stream);
    return e.rc;
}

// Backward table p[]:
//    0 gy | 1 x 2 z1 3 coef1 4 W1 5 Wq 6 Wk 7 Wv | 8 coord 9 knn idx 10-17 layer weights 18 layer bn 19 layer saved 20 H
//   21 xq 22 xk 23 xv | 24 t 25 z3 26 coef2 27 coef3 28 W3
//   outputs: 29 gx (n*c; = identity branch + linear1 branch) 30 grads of the pre half (block_pre_backward layout)
//            [30 | 31 | 32 must be laid out contiguously in this order: 30 and 31 are zeroed with one memset]
//            31 grads of the post half (pdf_block_post_backward layout) 32 layer sums (pdf_pt_layer_bwd_sums_floats(c))
//   scratch: 33 gt 34 da / dy 35 gxq 36 gxk 37 gxv (n*c each) 38 G2 (n*nsample*c/8) 39 G3 (n*nsample*3)
//            40 partial (max of pdf_bn_partial_floats, pdf_rowlin_partial_floats, pdf_pt_layer_bwd_partial_floats)
//            41 Wsm (n*nsample*c/8) 42 GR (n*nsample*c) | inverse kNN table: 43 inv_off (n+1) 44 inv_entry, entry_base
//            45 dy (n*c; separate from 34: the forked dW3 kernel may still be reading `da` when the pre half starts)
//   46: the batch's relative-coordinate sums of the kNN table (9 doubles, pdf_knn_rel_moments) or null | 47-48: unused
//             49 visiting order of the points (or null)
//   50 workspace of the weight-gradient slabs: max(pdf_rowlin_wgrad_ws_floats(n, c, c, 5), pdf_rowlin_wgrad_ws_floats(n, c, c, 3) +
//      pdf_rowlin_wgrad_ws_floats(n, c, c, 1)) floats (grouped launch / one product at a time)
// No gradient slot is accumulated into any more (slab reductions and column sums WRITE): the memset of round 2 is gone.
extern "C" int pdf_bottleneck_backward(long n, int nsample, int c, void *const *p, int training, int entry_base, int storage_bf16, int mma_input, void *stream) {
    if (n < 1 || !p) return PDF_ERR_BAD_ARG;
    Err e;
    if (!p[50]) return PDF_ERR_BAD_ARG;
    // The five c x c weight gradients of the block (linear3, q / k / v, linear1) as ONE grouped launch + ONE slab reduction at the end
    // (pdf_rowlin_wgrad_group; PDFOPS_WGRAD_GROUP=0: five launches + three reductions where each product's inputs become final, as in
    // rounds 2-3).  Their inputs are scratch / saved tensors nothing else in the block overwrites: da (34), t (24), g_xq / g_xk / g_xv
    // (35-37), z1 (2), dy (45), x (1).
    static const bool group = [] { const char *v = getenv("PDFOPS_WGRAD_GROUP"); return !(v && v[0] == '0'); }();
    float *ws_post = (float *)p[50], *ws_pre = ws_post + pdf_rowlin_wgrad_ws_floats(n, c, c, 1);
    void *post[13] = {p[0], p[24], p[1], p[25], p[26], p[27], p[28], p[33], p[29], p[31], p[34], p[40], ws_post};
    e << block_post_backward(n, c, post, training, mma_input, stream, group);
    const float *weights[8];
    for (int i = 0; i < 8; ++i) weights[i] = (const float *)p[10 + i];
    e << pdf_pt_layer_backward((int)n, nsample, c, (const float *)p[21], (const float *)p[22], (const float *)p[23], (const float *)p[8],
                               (const int *)p[9], weights, (const float *)p[18], (const float *)p[19], (const float *)p[20],
                               (const float *)p[33], (float *)p[35], (float *)p[36], (float *)p[37], (float *)p[38], (float *)p[39],
                               (float *)p[41], (float *)p[42], (const int *)p[43], (const int *)p[44], entry_base,
                               (float *)p[40], (float *)p[32], storage_bf16, (const int *)p[49], (const double *)p[46], stream);
    void *pre[15] = {p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[35], p[36], p[37], p[29], p[30], p[45], p[40], ws_pre};
    e << block_pre_backward(n, c, pre, training, 1, mma_input, stream, group);   // gx += dy W1 on top of the identity branch
    if (group && e.rc == 0) {
        const long cc = (long)c * c;
        float *gpre = (float *)p[30], *gpost = (float *)p[31];                  // block_pre_backward / block_post_backward gradient layouts
        float *dW1 = gpre, *dqkv = gpre + cc + 2 * c, *dW3 = gpost;
        const float *coef1 = (const float *)p[3], *coef2 = (const float *)p[26];
        const float *g[5] = {(const float *)p[34], (const float *)p[35], (const float *)p[36], (const float *)p[37], (const float *)p[45]};
        const float *x[5] = {(const float *)p[24], (const float *)p[2], (const float *)p[2], (const float *)p[2], (const float *)p[1]};
        const float *sc[5] = {coef2, coef1, coef1, coef1, nullptr}, *sh[5] = {coef2 + c, coef1 + c, coef1 + c, coef1 + c, nullptr};
        const int relu[5] = {1, 1, 1, 1, 0};
        float *dw[5] = {dW3, dqkv, dqkv + (cc + c), dqkv + 2 * (cc + c), dW1};
        float *db[5] = {nullptr, dqkv + cc, dqkv + (cc + c) + cc, dqkv + 2 * (cc + c) + cc, nullptr};
        const int rc = pdf_rowlin_wgrad_group(n, c, c, 5, g, c, x, c, sc, sh, relu, dw, db, ws_post, mma_input, stream);
        if (rc == PDF_ERR_UNSUPPORTED) {   // widths outside the streaming kernels: one product at a time
            e << pdf_rowlin_wgrad(n, c, c, g[0], c, x[0], c, sc[0], sh[0], 1, dw[0], nullptr, ws_post, mma_input, stream);
            e << pdf_rowlin_wgrad_multi(n, c, c, 3, g + 1, c, x[1], c, sc[1], sh[1], 1, dw + 1, db + 1, ws_pre, mma_input, stream);
            e << pdf_rowlin_wgrad(n, c, c, g[4], c, x[4], c, nullptr, nullptr, 0, dw[4], nullptr, ws_pre, mma_input, stream);
        } else {
            e << rc;
        }
    }
    return e.rc;
}

// ---------------------------------------------------------------------------------------------------------------------
// Linear (+ bias) -> BatchNorm1d -> ReLU on (n, k) rows as one host call per direction (TransitionUp.linear1/linear2, the cls
// and confidence heads: point_transformer_seg.py:131-147, 229-234; pt_v1.py:17-24).  Streaming shapes only (k in 32..512,
// o % 16 == 0): the GEMM's epilogue emits the column statistics, so the norm costs one extra pass (apply) instead of three.
// Forward p[]: 0 x 1 W (o,k) 2 bias 3 gamma 4 beta 5 rm 6 rv | 7 z (n,o) 8 coef (4o) 9 y (n,o) 10 partial (pdf_rowlin_partial_floats(n,o))
extern "C" int pdf_linbn_forward(long n, int k, int o, void *const *p, int training, int relu, float eps, float momentum, int mma_input, void *stream) {
    if (n < 1 || !p) return PDF_ERR_BAD_ARG;
    if (!pdf_bn_supported(o)) return PDF_ERR_UNSUPPORTED;
    Err e;
    float *z = (float *)p[7], *coef = (float *)p[8], *partial = (float *)p[10];
    if (training) {
        e << pdf_rowlin_forward_bn(n, k, o, (const float *)p[0], k, (const float *)p[1], (const float *)p[2], nullptr, nullptr, 0, z, o, partial,
                                   (const float *)p[3], (const float *)p[4], (float *)p[5], (float *)p[6], eps, momentum, coef, mma_input, stream);
    } else {
        e << pdf_rowlin_forward(n, k, o, (const float *)p[0], k, (const float *)p[1], 0, (const float *)p[2], nullptr, nullptr, 0, z, o, 0, nullptr, mma_input, stream);
        e << pdf_bn_coef_eval_or_partial(partial, 0, n, o, (const float *)p[3], (const float *)p[4], (float *)p[5], (float *)p[6], 0, eps, momentum, coef, stream);
    }
    e << pdf_bn_apply(n, o, z, nullptr, coef, relu, (float *)p[9], stream);
    return e.rc;
}

// Backward p[]: 0 gy 1 x 2 z 3 coef 4 W | 5 gx (n,k) or null 6 grads [dW (o*k) | db (o) | dbeta (o) | dgamma (o)] (written)
//               7 gz (n,o) scratch 8 partial (pdf_bn_partial_floats(n,o)) 9 weight-gradient slabs (pdf_rowlin_wgrad_ws_floats(n, k, o, 1))
extern "C" int pdf_linbn_backward(long n, int k, int o, void *const *p, int training, int relu, int mma_input, void *stream) {
    if (n < 1 || !p || !p[9]) return PDF_ERR_BAD_ARG;
    float *grads = (float *)p[6], *gz = (float *)p[7];
    const long ok = (long)o * k;
    Err e;
    e << pdf_bn_act_backward(n, o, (const float *)p[0], (const float *)p[2], nullptr, (const float *)p[3], training, relu, (float *)p[8],
                             grads + ok + o, gz, nullptr, stream);                                                    // [dbeta | dgamma]
    if (p[5]) e << pdf_rowlin_forward(n, o, k, gz, o, (const float *)p[4], 1, nullptr, nullptr, nullptr, 0, (float *)p[5], k, 0, nullptr, mma_input, stream);
    e << pdf_rowlin_wgrad(n, k, o, gz, o, (const float *)p[1], k, nullptr, nullptr, 0, grads, grads + ok, (float *)p[9], mma_input, stream);    // dW, db
    return e.rc;
}
