/*
 * pdfops.h -- C ABI of libpdfops.so, the MI355X (gfx950) replacement for the reference's
 * libs/pointops CUDA extension (`pointops._C`).
 *
 * Each entry point replaces one `extern "C"` launcher of the reference (file:line cited per
 * function, paths relative to the reference root).  Signatures keep the reference's parameter
 * lists and add
 *   - `b`      : the scene count, where the reference inferred it by scanning `offset`,
 *   - `stream` : a hipStream_t passed as void* (the reference launched on the legacy stream),
 * and return an int status: 0 = ok, PDF_ERR_* (<0) = argument error, >0 = hipError_t.
 *
 * Contract (same as the reference unless noted):
 *   - all pointers are DEVICE pointers owned by the caller; the library never allocates,
 *     never synchronises and keeps NO mutable state -- no globals, no per-stream tables (re-entrant, any thread, any stream; since
 *     ABI 4 the reduced-precision mode of the Linear products is a per-call argument, `mma_input` below).  The only process-wide inputs
 *     are a few read-only PDFOPS_* environment variables that select kernel variants / grid caps for experiments and tests
 *     (PDFOPS_FPS_MW, PDFOPS_KNN_*, PDFOPS_PT_BLOCKS_*, PDFOPS_PT_CAP_<pass>; defaults are the measured best), read once per process;
 *   - float data is fp32, indices/offsets are int32, `offset` arrays hold CUMULATIVE scene ends;
 *   - kNN placeholder for scenes with fewer than `nsample` points: idx = -1, dist2 = 1e10;
 *   - outputs written by plain stores need no initialisation (grouping/subtraction/
 *     interpolation/aggregation forward, knn, fps, aggregation grad_position);
 *     outputs that are ACCUMULATED into must be pre-zeroed by the caller exactly as the
 *     reference's Python wrappers do: every grad_* scatter target and the attention_* forward outputs;
 *   - gather kernels treat idx < 0 as a zero row (the reference dereferences it: undefined).
 */
#ifndef PDFOPS_H
#define PDFOPS_H

#ifdef __cplusplus
extern "C" {
#endif

#define PDF_OK 0
#define PDF_ERR_BAD_ARG (-1)      /* null pointer / negative size */
#define PDF_ERR_NSAMPLE (-2)      /* nsample outside 1..128 (reference: best_dist[128], knn_query_cuda_kernel.cu:82) */
#define PDF_ERR_UNSUPPORTED (-3)  /* shape outside what the kernels are built for */

/* ABI version of THIS header.  pdf_abi_version() returns the version the loaded library was built with: a caller built against another
 * version must refuse to call (parameter lists differ; pointcloudpdf_amd/_native.py does).  History: 1 = rounds 1-3 (signatures changed
 * without a bump, see ADVICE of round 3); 4 = round 4: `mma_input` per call on the pdf_rowlin_ / pdf_block_ / pdf_bottleneck_ / pdf_linbn_ /
 * pdf_td_ entries, pdf_set_mma_input / pdf_get_mma_input / pdf_tickets_* removed, pdf_sgd_step takes a found-inf flag;
 * 5 = round 5: pdf_wa_* (atomic-free window-attention backward, fused logits), pdf_layernorm_*, pdf_region_* / *_dev (sync-free pseudo-label pass) added. */
#define PDF_ABI_VERSION 6
int pdf_abi_version(void);
const char *pdf_build_info(void);
/* Arithmetic of the squared distance in this library's geometry kernels (kNN, ball query, FPS): 0 = the reference's expression as
 * written in IEEE fp32 (libpdfops.so), 1 = fmaf(dz,dz,fmaf(dy,dy,dx*dx)), 2 = fmaf(dz,dz,fmaf(dx,dx,dy*dy)) (libpdfops_fma{1,2}.so:
 * the contractions an `nvcc -O2` build of knn_query_cuda_kernel.cu:92 / sampling_cuda_kernel.cu:54 may compute; csrc/pdfops_common.h). */
int pdf_dist_fma_mode(void);
/* `mma_input` (argument of every entry that runs the streaming per-point Linear products): the reduced-precision variant the reference
 * trains this path with (torch.cuda.amp.autocast: configs/s3dis/openseg-pt-v1-0-msp.py:6, pointcept/engines/train.py:340-363).
 * 0: fp32 operands, v_mfma_f32_16x16x4_f32 -- the parity path; 1: the operands of the products (forward, input gradient, weight gradient;
 * channel widths 32..512) are rounded to fp16 in registers and multiplied with v_mfma_f32_16x16x16_f16; 2: the same with bfloat16.
 * Tensors stay fp32 in memory and every accumulation is fp32 in all modes (what autocast does to nn.Linear, minus the rounding of the
 * OUTPUT to half).  Shapes the streaming kernels do not cover run in fp32 whatever the mode.  Any other value: PDF_ERR_BAD_ARG. */
#define PDF_MMA_F32 0
#define PDF_MMA_F16 1
#define PDF_MMA_BF16 2

/* replaces knn_query_cuda_launcher, libs/pointops/src/knn_query/knn_query_cuda_kernel.h:13
 * (kernel knn_query_cuda_kernel.cu:60-104).  idx (m,nsample), dist2 (m,nsample) = SQUARED distances. */
int pdf_knn_query(int m, int nsample, const float *xyz, const float *new_xyz,
                  const int *offset, const int *new_offset, int b,
                  int *idx, float *dist2, void *stream);

/* Same results through a uniform grid (exact: queries whose answer could depend on the reference's tie-breaking
 * are re-run by the scan kernel).  n = number of source points, workspace from pdf_knn_workspace_bytes(b, n, m). */
long pdf_knn_workspace_bytes(int b, int n, int m);
int pdf_knn_grid_supported(int nsample);
int pdf_knn_query_ws(int m, int nsample, int n, const float *xyz, const float *new_xyz, const int *offset,
                     const int *new_offset, int b, int *idx, float *dist2, void *workspace, long workspace_bytes,
                     void *stream);
/* The two halves of pdf_knn_query_ws on their own: one grid per set of SOURCE points (pdf_knn_grid_build), any number of query sets and
 * nsample values (3 / 8 / 16) over it (pdf_knn_query_grid).  workspace_bytes >= pdf_knn_workspace_bytes(b, n, m) for the largest m. */
int pdf_knn_grid_build(int n, const float *xyz, const int *offset, int b, void *workspace, long workspace_bytes, void *stream);
int pdf_knn_query_grid(int m, int nsample, int n, const float *xyz, const float *new_xyz, const int *offset, const int *new_offset, int b,
                       int *idx, float *dist2, void *workspace, long workspace_bytes, void *stream);
/* measurement aid: the same with the grid kernel counting the candidate distances it evaluates (*pairs, device counter zeroed by the
 * caller, += count).  The grid prunes by design, so "pairs per second" of this path must be quoted on EVALUATED pairs, not on the
 * m * n_scene pairs of the brute force it replaces. */
int pdf_knn_query_ws_counted(int m, int nsample, int n, const float *xyz, const float *new_xyz, const int *offset,
                             const int *new_offset, int b, int *idx, float *dist2, void *workspace, long workspace_bytes,
                             unsigned long long *pairs, void *stream);
/* the exact scan restricted to qlist[0 .. *qcount) (device pointers); qlist == NULL: all m queries */
int pdf_knn_query_list(int m, int nsample, const float *xyz, const float *new_xyz, const int *offset,
                       const int *new_offset, int b, int *idx, float *dist2, const int *qlist, const int *qcount,
                       void *stream);

/* replaces ball_query_cuda_launcher, libs/pointops/src/ball_query/ball_query_cuda_kernel.h:9-17 (kernel
 * ball_query_cuda_kernel.cu:58-123): points with d2 <= 1e-5 or min_radius^2 <= d2 < max_radius^2, collected in index
 * order, permuted by the reference's heap_sort, then the first candidates (<= nsample, padded with idx -1 / dist2 1e10) or
 * every (count / nsample)-th one (dist2 then holds the candidate INDEX as float, as upstream :120).  The reference's
 * per-thread list holds 2048 candidates and overruns beyond; here the first 2048 accepted points are kept.
 * nsample <= 2048, min_radius < max_radius. */
int pdf_ball_query(int m, int nsample, float min_radius, float max_radius, const float *xyz, const float *new_xyz,
                   const int *offset, const int *new_offset, int b, int *idx, float *dist2, void *stream);

/* replaces random_ball_query_cuda_launcher, libs/pointops/src/random_ball_query/random_ball_query_cuda_kernel.h:9-17
 * (kernel random_ball_query_cuda_kernel.cu:58-108): the first nsample accepted points along the caller's per-scene
 * permutation `order` (n ints, global row ids), padded with idx -1 / dist2 1e10. */
int pdf_random_ball_query(int m, int nsample, float min_radius, float max_radius, const int *order, const float *xyz,
                          const float *new_xyz, const int *offset, const int *new_offset, int b, int *idx, float *dist2,
                          void *stream);

/* replaces farthest_point_sampling_cuda_launcher, libs/pointops/src/sampling/sampling_cuda_kernel.h:13
 * (kernel sampling_cuda_kernel.cu:14-129).  `n` = size of the largest scene (fixes the reference's
 * block size opt_n_threads(n), cuda_utils.h:11-14, which fixes its arg-max tie rule).
 * `tmp` (N floats) is scratch and need NOT be pre-filled (the reference wants 1e10). */
int pdf_farthest_point_sampling(int b, int n, const float *xyz, const int *offset,
                                const int *new_offset, float *tmp, int *idx, void *stream);

/* log2 of the block size the reference would launch for a largest-scene size n (cuda_utils.h:11-14) */
int pdf_fps_reference_block_log2(int n);

/* Bucketed exact FPS (same results as pdf_farthest_point_sampling, far fewer point updates).
 * Host-side sizes: n_total = offset[b-1].  Workspace from pdf_fps_workspace_bytes(). */
long pdf_fps_workspace_bytes(int b, int n_total);
long pdf_fps_stats_offset(int b, int n_total); /* 4 x u32 per scene: bucket updates, super visits, samples, buckets */
int pdf_farthest_point_sampling_bucketed(int b, int n, int n_total, const float *xyz, const int *offset,
                                         const int *new_offset, void *workspace, long workspace_bytes,
                                         int *idx, void *stream);

/* replaces grouping_{forward,backward}_cuda_launcher, libs/pointops/src/grouping/grouping_cuda_kernel.h:14-15 */
int pdf_grouping_forward(int m, int nsample, int c, const float *input, const int *idx, float *output, void *stream);
int pdf_grouping_backward(int m, int nsample, int c, const float *grad_output, const int *idx, float *grad_input, void *stream);
/* The forward gathers with an optional VISITING ORDER of the queries (`order`: a permutation of 0 .. m-1, e.g. the Morton order of the
 * query points that the geometry pre-pass keeps per level; NULL = storage order): identical outputs.  Neighbouring queries share most
 * of their rows, and the kernels give every XCD one contiguous stretch of the order, so the repeats hit that XCD's L2 instead of
 * crossing the fabric (csrc/gather_ops.hip: grouping2 forward 51 % -> 69 % of the HBM peak).  The reference-ABI entries above are
 * these with order = NULL. */
int pdf_grouping_forward_ordered(int m, int nsample, int c, const float *input, const int *idx, const int *order, float *output, void *stream);
int pdf_group_forward_ordered(int m, int nsample, int c, int with_xyz, const float *feat, const float *xyz, const float *new_xyz,
                              const int *idx, const int *order, float *output, void *stream);
int pdf_interpolation_forward_ordered(int n, int c, int k, const float *input, const int *idx, const float *weight, const int *order,
                                      float *output, void *stream);
int pdf_subtraction_forward_ordered(int n, int nsample, int c, const float *input1, const float *input2, const int *idx, const int *order,
                                    float *output, void *stream);
int pdf_aggregation_forward_ordered(int n, int nsample, int c, int w_c, const float *input, const float *position, const float *weight,
                                    const int *idx, const int *order, float *output, void *stream);

/* replaces interpolation_{forward,backward}_cuda_launcher, libs/pointops/src/interpolation/interpolation_cuda_kernel.h:14-15 */
int pdf_interpolation_forward(int n, int c, int k, const float *input, const int *idx, const float *weight, float *output, void *stream);
int pdf_interpolation_backward(int n, int c, int k, const float *grad_output, const int *idx, const float *weight, float *grad_input, void *stream);

/* replaces subtraction_{forward,backward}_cuda_launcher, libs/pointops/src/subtraction/subtraction_cuda_kernel.h:14-15 */
int pdf_subtraction_forward(int n, int nsample, int c, const float *input1, const float *input2, const int *idx, float *output, void *stream);
int pdf_subtraction_backward(int n, int nsample, int c, const int *idx, const float *grad_output, float *grad_input1, float *grad_input2, void *stream);

/* replaces aggregation_{forward,backward}_cuda_launcher, libs/pointops/src/aggregation/aggregation_cuda_kernel.h:14-15 */
int pdf_aggregation_forward(int n, int nsample, int c, int w_c, const float *input, const float *position,
                            const float *weight, const int *idx, float *output, void *stream);
int pdf_aggregation_backward(int n, int nsample, int c, int w_c, const float *input, const float *position,
                             const float *weight, const int *idx, const float *grad_output,
                             float *grad_input, float *grad_position, float *grad_weight, void *stream);

/* replace attention_{relation,fusion}_step_{forward,backward}_cuda_launcher,
 * libs/pointops/src/attention/attention_cuda_kernel.h:31-53 */
int pdf_attention_relation_step_forward(int m, int g, int c, const float *query, const float *key, const float *weight,
                                        const int *index_target, const int *index_refer, float *output, void *stream);
int pdf_attention_relation_step_backward(int m, int g, int c, const float *query, float *grad_query,
                                         const float *key, float *grad_key, const float *weight, float *grad_weight,
                                         const int *index_target, const int *index_refer, const float *grad_output, void *stream);
int pdf_attention_fusion_step_forward(int m, int g, int c, const float *weight, const float *value,
                                      const int *index_target, const int *index_refer, float *output, void *stream);
int pdf_attention_fusion_step_backward(int m, int g, int c, const float *weight, float *grad_weight,
                                       const float *value, float *grad_value,
                                       const int *index_target, const int *index_refer, const float *grad_output, void *stream);

/* ---- fused entry points of our own (no reference launcher; they fuse the reference's Python-side
 *      compositions so the (m,nsample,c) temporaries never reach HBM) ---- */

/* pointops.grouping(idx, feat, xyz, new_xyz, with_xyz) -- libs/pointops/functions/grouping.py:36-60.
 * output (m, nsample, 3*with_xyz + c): [ (xyz[idx]-new_xyz) masked by idx>=0 | feat[idx] ], idx<0 -> zeros. */
int pdf_group_forward(int m, int nsample, int c, int with_xyz, const float *feat, const float *xyz,
                      const float *new_xyz, const int *idx, float *output, void *stream);
/* gradient w.r.t. feat only (xyz carries no gradient on this path); grad_feat (n,c) pre-zeroed. */
int pdf_group_backward(int m, int nsample, int c, int with_xyz, const float *grad_output, const int *idx,
                       float *grad_feat, void *stream);

/* inverse-distance weights of pointops.interpolation -- libs/pointops/functions/interpolation.py:14-17:
 * weight[n,j] = (1/(sqrt(dist2[n,j])+1e-8)) / sum_j(...) */
int pdf_interpolation_weights(int n, int k, const float *dist2, float *weight, void *stream);

/* Fused PointTransformerLayer (point_transformer_seg.py:45-78 incl. the three BatchNorm-as-LayerNorm norms,
 * pointcept/models/point_transformer/utils.py:7-14), forward.  Supported: nsample in {8,16}, c in {32,64,128}
 * (pdf_pt_layer_supported).  weights[8] = Wp1,bp1,Wp2,bp2,Ww1,bw1,Ww2,bw2 (host array of device pointers);
 * bn_params[6] = gamma/beta of linear_p[1], linear_w[0], linear_w[3]; bn_buffers[6] = their running mean/var
 * (updated in place when training).  bn (2*(3+c+c/8) floats) receives the scale/shift of the three norms,
 * saved (same size) their batch mean / rstd (training), H (n*nsample*c/8) the pre-norm attention hidden,
 * partial = pdf_pt_layer_partial_floats(...) floats of scratch, out (n,c). */
int pdf_pt_layer_supported(int nsample, int c);
long pdf_pt_layer_partial_floats(int n, int nsample, int c);
int pdf_pt_layer_forward(int n, int nsample, int c, const float *xq, const float *xk, const float *xv,
                         const float *p, const int *idx, const float *const *weights,
                         const float *const *bn_params, float *const *bn_buffers, int training, float eps,
                         float momentum, float *bn, float *saved, float *H, float *partial, float *out,
                         int storage_bf16, const int *order, void *stream);
/* The same with the batch's relative-coordinate sums (pdf_knn_rel_moments, summed over the batch's scenes: 9 doubles in device
 * memory) or NULL: in train mode the geometry branch's BatchNorm then comes from them (5 launches per layer instead of 7).
 * out_stat_rows (host int, may be NULL): when given, the last pass also leaves *out_stat_rows partial rows [sum out (c) | sum out^2 (c)]
 * in `partial` (train mode; 0 in eval mode) -- the statistics of a BatchNorm that reads `out` (the Bottleneck's bn2,
 * point_transformer_seg.py:187) for pdf_bn_coef_from_partial, instead of a statistics pass over `out`. */
int pdf_pt_layer_forward_m(int n, int nsample, int c, const float *xq, const float *xk, const float *xv,
                           const float *p, const int *idx, const float *const *weights,
                           const float *const *bn_params, float *const *bn_buffers, int training, float eps,
                           float momentum, float *bn, float *saved, float *H, float *partial, float *out,
                           int storage_bf16, const int *order, const double *moments, int *out_stat_rows, void *stream);

/* Backward of the fused PointTransformerLayer (train mode).  gxq / gxk / gxv are overwritten: the scatters of g_xk and g_xv run as
 * segmented gathers over the INVERSE of the kNN table (inv_off (n+1), inv_entry, entry_base -- see pdf_seg_sum_rows), so they are
 * deterministic and atomics-free; Wsm (n*nsample*c/8) and GR (n*nsample*c) are scratch (softmax weights, g_r rows).  sums needs
 * pdf_pt_layer_bwd_sums_floats(c) + 2*(3+c+c/8) floats and returns the parameter-gradient sections documented in
 * csrc/fused_layer.hip.  storage_bf16 (forward AND backward of a layer must agree): the row arrays only the layer itself reads --
 * H (saved), G2 / Wsm / GR (scratch) -- hold bfloat16 (round-to-nearest-even) in the first half of the same buffers; every sum and
 * product stays fp32.  The reduced-precision variant behind `bench.py --storage bf16` (the reference trains under AMP). */
long pdf_pt_layer_bwd_partial_floats(int n, int nsample, int c);
long pdf_pt_layer_bwd_sums_floats(int c);
int pdf_pt_layer_backward(int n, int nsample, int c, const float *xq, const float *xk, const float *xv,
                          const float *p, const int *idx, const float *const *weights, const float *bn,
                          const float *saved, const float *H, const float *gout, float *gxq, float *gxk,
                          float *gxv, float *G2, float *G3, float *Wsm, float *GR, const int *inv_off, const int *inv_entry,
                          int entry_base, float *partial, float *sums, int storage_bf16, const int *order, const double *moments,
                          void *stream);   /* moments: the batch's relative-coordinate sums (pdf_knn_rel_moments) or NULL: with them d Wp1 / d bp1 come in
                                              closed form from the sums of the third pass (no fourth pass over the rows) */

/* BatchNorm1d over (n, c) rows fused with the residual add and ReLU that follow it in the Bottleneck
 * (point_transformer_seg.py:184-192).  c must be a power of two in 4..1024.  coef (4c floats) = scale|shift|mean|rstd,
 * partial = pdf_bn_partial_floats(n, c) floats of scratch; backward: sums (2c) receives [d beta | d gamma]. */
int pdf_bn_supported(int c);
long pdf_bn_partial_floats(long n, int c);
int pdf_bn_act_forward(long n, int c, const float *x, const float *res, const float *gamma, const float *beta,
                       float *running_mean, float *running_var, int training, float eps, float momentum, int relu,
                       float *coef, float *partial, float *y, void *stream);
int pdf_bn_act_backward(long n, int c, const float *gy, const float *x, const float *res, const float *coef,
                        int training, int relu, float *partial, float *sums, float *gx, float *gres, void *stream);
/* coefficients only (statistics pass + finalize, or running statistics) */
int pdf_bn_coef(long n, int c, const float *x, const float *gamma, const float *beta, float *running_mean,
                float *running_var, int training, float eps, float momentum, float *coef, float *partial, void *stream);
/* y = relu?(x * scale + shift + res) with coefficients computed elsewhere (no statistics pass) */
int pdf_bn_apply(long n, int c, const float *x, const float *res, const float *coef, int relu, float *y, void *stream);

/* Dense per-point Linear on the fp32 matrix cores (csrc/rowlin.hip).  y (+)= f(x) Wt + bias with f = identity or the
 * folded BatchNorm-affine (+ReLU) of the producer (scale/shift over the k input channels); partial (optional,
 * pdf_rowlin_partial_floats) receives per-row-block column sums / sums of squares of y for the BatchNorm that follows.
 * transpose_w = 1 computes the input gradient dX = G W with the layer's (out, in) weight. */
long pdf_rowlin_partial_floats(long n, int o);
int pdf_rowlin_partial_rows(long n, int k, int o);
int pdf_rowlin_forward(long n, int k, int o, const float *x, long ldx, const float *w, int transpose_w,
                       const float *bias, const float *scale, const float *shift, int relu, float *y, long ldy,
                       int accumulate, float *partial, int mma_input, void *stream);
/* dW (o,k) = G^T f(X), db (o) = column sums of G (db may be NULL): both WRITTEN.  No float atomics: every workgroup stores its
 * partial block into a slab of ws (pdf_rowlin_wgrad_ws_floats(n, k, o, ng) floats, ng = 1 here), a second launch sums the slabs in a
 * fixed order -- bit-reproducible gradients (the reference's atomicAdd scatters are not: grouping_cuda_kernel.cu:16-25). */
long pdf_rowlin_wgrad_ws_floats(long n, int k, int o, int ng);
int pdf_rowlin_wgrad(long n, int k, int o, const float *g, long ldg, const float *x, long ldx,
                     const float *scale, const float *shift, int relu, float *dw, float *db, float *ws, int mma_input, void *stream);
/* BatchNorm coefficients from column partials [rows][2c] (sum | sum of squares): coef = scale|shift|mean|rstd */
int pdf_bn_coef_from_partial(const float *partial, int rows, long n, int c, const float *gamma, const float *beta,
                             float *running_mean, float *running_var, float eps, float momentum, float *coef, void *stream);

/* Several Linear layers over the same rows in one launch: nin = 1, nout <= 3 (y[i] = f(x[0]) Wt[i] + bias[i]: the q/k/v
 * projections, point_transformer_seg.py:47-49) or nin <= 3, nout = 1 (y[0] (+)= sum_i x[i] Wt[i]: their input gradient);
 * pdf_rowlin_wgrad_multi: weight / bias gradients of up to three layers sharing the input x. */
int pdf_rowlin_multi(long n, int k, int o, int nin, int nout, const float *const *x, long ldx, const float *const *w,
                     int transpose_w, const float *const *bias, const float *scale, const float *shift, int relu,
                     float *const *y, long ldy, int accumulate, int mma_input, void *stream);
int pdf_rowlin_wgrad_multi(long n, int k, int o, int ng, const float *const *g, long ldg, const float *x, long ldx,
                           const float *scale, const float *shift, int relu, float *const *dw, float *const *db, float *ws, int mma_input, void *stream);
/* Up to five weight gradients of ONE shape (n, k, o) with their OWN inputs in one launch + one slab reduction -- the five c x c products
 * of a Bottleneck backward (linear3, q / k / v, linear1; point_transformer_seg.py:184-192): dW_i = G_i^T f_i(X_i), f_i = relu_i?(x *
 * scale_i + shift_i) where scale_i is non-null, else the identity; db_i (nullable, db itself may be null) = column sums of G_i.  Written,
 * not accumulated.  Streaming shapes only (k, o multiples of 32 up to 512): PDF_ERR_UNSUPPORTED otherwise.
 * ws: pdf_rowlin_wgrad_ws_floats(n, k, o, ng) floats. */
int pdf_rowlin_wgrad_group(long n, int k, int o, int ng, const float *const *g, long ldg, const float *const *x, long ldx,
                           const float *const *scale, const float *const *shift, const int *relu, float *const *dw,
                           float *const *db, float *ws, int mma_input, void *stream);

/* Rigid KPConv of the StratifiedTransformer stem (stratified_transformer_v1m1_origin.py:582-662 over torch_points3d's KPConvLayer): the
 * part that is not a matrix product.  weighted[n, k, c] = sum_m max(0, 1 - |support[nb[n, m]] - query[n] - k_points[k]| / extent) *
 * x[nb[n, m], c] (overwritten; nb = -1: no neighbour); out = weighted (N, KP * C_in) times weight (KP * C_in, C_out) is a pdf_rowlin_forward.
 * pdf_kpconv_scatter is its adjoint: grad_x[nb[n, m], c] += sum_k w grad_weighted[n, k, c] (grad_x zeroed by the caller; float atomics).
 * KP <= 16, C_in <= 16 (pdf_kpconv_supported), else PDF_ERR_UNSUPPORTED. */
int pdf_kpconv_supported(int kp, int cin);
int pdf_kpconv_gather(int n, int m, int kp, int cin, const float *query, const float *support, const int *neighbors, const float *x,
                      const float *k_points, float extent, float *weighted, void *stream);
int pdf_kpconv_scatter(int n, int m, int kp, int cin, const float *query, const float *support, const int *neighbors,
                       const float *grad_weighted, const float *k_points, float extent, float *grad_x, void *stream);
/* Input gradient y = sum_i x[i] W[i] (nin <= 3; W (k, o) = the layers' own (out, in) weights) of Linear layers reading a
 * BatchNorm(+ReLU) output bx -> bn -> relu, with that BatchNorm's backward sums as the product's epilogue: partial
 * (pdf_rowlin_partial_floats(n, o)) receives *partial_rows rows of [sum g' | sum g' xhat]; pdf_bn_act_backward_presummed then
 * finishes the BatchNorm backward without re-reading y and bx for the reduction.  PDF_ERR_UNSUPPORTED outside the streaming
 * shapes (callers then use pdf_rowlin_multi + pdf_bn_act_backward). */
int pdf_rowlin_dgrad_bstats(long n, int k, int o, int nin, const float *const *x, long ldx, const float *const *w, float *y, long ldy,
                            const float *bx, long ldb, const float *bcoef, int brelu, float *partial, int *partial_rows,
                            int mma_input, void *stream);
/* pdf_rowlin_forward + the coefficients (scale | shift | mean | rstd) of the train-mode BatchNorm behind it (column sums in the
 * product's epilogue + one finalizer launch). */
int pdf_rowlin_forward_bn(long n, int k, int o, const float *x, long ldx, const float *w, const float *bias, const float *scale,
                          const float *shift, int relu, float *y, long ldy, float *partial, const float *gamma, const float *beta,
                          float *running_mean, float *running_var, float eps, float momentum, float *coef, int mma_input, void *stream);
int pdf_bn_act_backward_presummed(long n, int c, const float *gy, const float *x, const float *coef, int training, int relu,
                                  const float *partial, int partial_rows, float *sums, float *gx, void *stream);

/* Dense halves of the Bottleneck (point_transformer_seg.py:184-192) as single host calls (csrc/block.hip documents the
 * pointer tables p[]): pre = linear1 + bn1 + ReLU + q/k/v projections, post = bn2 + ReLU + linear3 + bn3 + residual + ReLU. */
int pdf_bn_coef_eval_or_partial(const float *partial, int rows, long n, int c, const float *gamma, const float *beta,
                                float *running_mean, float *running_var, int training, float eps, float momentum,
                                float *coef, void *stream);
int pdf_block_pre_forward(long n, int c, void *const *p, int training, float eps, float momentum, int mma_input, void *stream);
int pdf_block_pre_backward(long n, int c, void *const *p, int training, int mma_input, void *stream);
int pdf_block_post_forward(long n, int c, void *const *p, int training, float eps, float momentum, int mma_input, void *stream);
int pdf_block_post_backward(long n, int c, void *const *p, int training, int mma_input, void *stream);
/* The whole Bottleneck (both halves + the fused attention layer) as one call per direction; tables in csrc/block.hip. */
int pdf_bottleneck_forward(long n, int nsample, int c, void *const *p, int training, float eps, float momentum, int storage_bf16, int mma_input, void *stream);
int pdf_bottleneck_backward(long n, int nsample, int c, void *const *p, int training, int entry_base, int storage_bf16, int mma_input, void *stream);
/* Linear (+ bias) -> BatchNorm1d -> (ReLU) as one call per direction (TransitionUp, heads); tables in csrc/block.hip. */
int pdf_linbn_forward(long n, int k, int o, void *const *p, int training, int relu, float eps, float momentum, int mma_input, void *stream);
int pdf_linbn_backward(long n, int k, int o, void *const *p, int training, int relu, int mma_input, void *stream);

/* Row-weighted variants of the streaming Linear kernels and the bare BatchNorm-backward sums (building blocks of pdf_td_*). */
int pdf_rowlin_forward_roww(long n, int k, int o, const float *x, long ldx, const float *w, int transpose_w, float *y, long ldy,
                            int accumulate, const float *roww, long rws, int mma_input, void *stream);
int pdf_rowlin_wgrad_roww(long n, int k, int o, const float *g, long ldg, const float *x, long ldx, float *dw, const float *roww,
                          long rws, float *ws, int mma_input, void *stream);
int pdf_bn_bwd_sums(long n, int c, const float *gy, const float *x, const float *coef, int relu, float *partial, float *sums,
                    void *stream);

/* Fused TransitionDown with stride (point_transformer_seg.py:96-119): grouping with relative coordinates + Linear(3+cin, cout) +
 * train/eval BatchNorm + ReLU + max over the 16 neighbours, forward and backward, without the (m, 16, .) intermediates.
 * Geometry-only inputs (memoised per Geometry): idx (m,16), rel4 (m,16,4) masked relative coordinates, Z (n,32) = [sum of the
 * relative coordinates of the rows gathering point j (3) | their count | 0...], consts (16) = [sum rel^T rel (9) | sum rel (3)].
 * Pointer tables: csrc/transition_down.hip. */
int pdf_td_supported(int nsample, int cin, int cout);
int pdf_td_tables(long m, int b, const float *p_src, const float *p_new, const int *idx, const int *new_offset, float *rel4, float *Z,
                  float *scene_sums, long n, const int *inv_off, const int *inv_entry, int entry_base,
                  void *stream);   /* Z (n,32) and scene_sums (b,16) zeroed by the caller; inv_* = the inverse of idx (or NULL: float atomics) */
long pdf_td_gram_floats(int cin);
long pdf_td_fwd_scratch_floats(long n, int cin);
long pdf_td_bwd_scratch_floats(long m, int cin, int cout);
int pdf_td_forward(long n, long m, int cin, int cout, void *const *p, int training, float eps, float momentum, int mma_input, void *stream);
int pdf_td_backward(long n, long m, int cin, int cout, void *const *p, int entry_base, int mma_input, void *stream);

/* ---- scatter-adds as segmented gathers over an inverse neighbour table (csrc/seg_gather.hip; no reference counterpart: the reference
 * scatters with atomicAdd -- grouping_cuda_kernel.cu:20-25, interpolation_cuda_kernel.cu:27-33, subtraction_cuda_kernel.cu:24-30,
 * aggregation_cuda_kernel.cu:30-39).  For a table idx (m, nsample) over n rows: inv_off (n + 1) ascending positions into inv_entry;
 * inv_entry holds entry ids e + entry_base (e = i * nsample + j) grouped by destination idx[e], ascending inside a destination
 * (fixed summation order); entries with idx < 0 lie outside every segment.  pdf_subtraction_backward / pdf_aggregation_backward
 * accept a NULL scatter target (grad_input2 / grad_input) when the caller forms it with these. */
int pdf_seg_sum_rows(long n, int c, const float *src, const int *inv_off, const int *inv_entry, int entry_base, float scale, float *out,
                     void *stream);
/* self tables: inverse-segment sums (scaled) and every point's own-row sums of src (n * nsample, c) in one walk (subtraction backward) */
int pdf_seg_sum_rows_own(long n, int c, int nsample, const float *src, const int *inv_off, const int *inv_entry, int entry_base, float scale,
                         const int *order, float *out, float *out_own, void *stream);
int pdf_seg_sum_rows_strided(long n, int c, const float *src, long src_stride, const int *inv_off, const int *inv_entry, int entry_base,
                             float scale, float *out, void *stream);
int pdf_seg_sum_rows_x(long n, int c, const float *src, long src_stride, int src_bf16, const int *inv_off, const int *inv_entry, int entry_base,
                       float scale, float *out, void *stream);     /* src rows fp32 or bfloat16 */
int pdf_seg_sum_weighted_x(long n, int c, int nsample, int w_c, const float *src, const float *w, int w_bf16, const int *inv_off,
                           const int *inv_entry, int entry_base, const int *order, float *out, void *stream);   /* w fp32 or bfloat16 */
int pdf_seg_sum_weighted_ordered(long n, int c, int nsample, int w_c, const float *src, const float *w, const int *inv_off,
                                 const int *inv_entry, int entry_base, const int *order, float *out, void *stream);   /* order: visiting order of the n destinations or NULL */
int pdf_seg_sum_weighted(long n, int c, int nsample, int w_c, const float *src, const float *w, const int *inv_off, const int *inv_entry,
                         int entry_base, float *out, void *stream);

/* Cross-entropy with an ignore label, mean over the counted rows (pointcept/models/losses/misc.py:14-39 as configured on
 * this path).  Forward: loss[0], acc (pdf_ce_workspace_floats() floats; the first two = [sum, count], the rest per-workgroup partial
 * sums added in a fixed order: no atomics, nothing to zero), grad (n*c) = softmax - onehot (0 on ignored rows); a target that is neither
 * `ignore` nor in [0, c) makes the loss NaN (torch raises a device-side assert there).  Backward: grad_out = dlogits * gy[0] /
 * count; dlogits (the forward's `grad`) is only read, so the node can be differentiated more than once; grad_out may alias it. */
long pdf_ce_workspace_floats(void);
int pdf_ce_forward(long n, int c, const float *logits, const long *target, long ignore, float *grad, float *acc, float *loss,
                   void *stream);
int pdf_ce_backward(long n, int c, const float *dlogits, const float *acc, const float *gy, float *grad_out, void *stream);

/* Per-scene sums of the relative coordinates rel = xyz[idx[i, j]] - xyz[i] of a SELF neighbour table (rows with idx < 0: rel = 0):
 * out (b, 9) double = [Sx Sy Sz | Mxx Mxy Mxz Myy Myz Mzz], WRITTEN (every workgroup lies inside one scene and stores its nine sums
 * into its own slot of ws -- pdf_knn_rel_moments_ws_doubles(b, n) doubles --, a second launch adds a scene's slots in order: no
 * atomics, bit-reproducible statistics).  The train-mode BatchNorm after the layer's
 * Linear(3, 3) (point_transformer_seg.py:27-29) is a closed form of these and the weights: pdf_pt_layer_forward_m takes the batch's sums
 * and skips its first statistics pass (csrc/geom_moments.hip). */
long pdf_knn_rel_moments_ws_doubles(int b, long m);
int pdf_knn_rel_moments(int b, long n, int nsample, const float *xyz, const int *offset, const int *idx, double *out, double *ws, void *stream);
/* the same for m queries new_xyz (scene ends new_offset) over other source points xyz: TransitionDown's grouping table */
int pdf_knn_rel_moments_q(int b, long m, int nsample, const float *xyz, const float *new_xyz, const int *new_offset, const int *idx,
                          double *out, double *ws, void *stream);

/* Morton keys of the points of a batch, scene by scene (visiting order of the forward gathers and layer passes: a stable sort of these keys;
 * pointcloudpdf_amd/geometry.py: Geometry.order -- no reference counterpart, nothing is stored in that order): keys[i] = scene(i) << 30 |
 * 30-bit Morton code of point i on a 1024^3 grid over ITS scene's bounding box.  bounds: 4 b floats of scratch (written). */
int pdf_scene_morton_keys(long n, int b, const float *xyz, const int *offset, float *bounds, long long *keys, void *stream);

/* SGD with momentum and weight decay (torch.optim.SGD, dampening 0, no Nesterov -- the optimizer the reference's configs build,
 * pointcept/utils/optimizer.py + configs/s3dis/openseg-pt-v1-0-*.py) over every parameter tensor in ONE launch.  tab: ntensors records
 * {float *param; const float *grad; float *momentum; long length} in device memory; chunks: nchunks {tensor, chunk} int32 pairs, one
 * per pdf_sgd_chunk() values of a tensor.  param / momentum are updated in place.
 * found_inf (device float, may be NULL): non-zero = the step is skipped as a whole (a GradScaler step that saw a non-finite gradient). */
int pdf_sgd_chunk(void);
int pdf_sgd_step(int nchunks, const void *tab, const int *chunks, float lr, float momentum, float weight_decay, const float *found_inf,
                 void *stream);
/* Dynamic loss scaling of the reference's AMP training (torch.cuda.amp.GradScaler, pointcept/engines/train.py:340-363) with every decision
 * on the device, so that a captured step replays: pdf_grad_unscale multiplies every gradient of the table (pdf_sgd_step's layout) by
 * *inv_scale in place and sets *found_inf = 1 when any value is inf / nan (found_inf must be 0 on entry); pdf_scaler_update applies
 * GradScaler.update(): found_inf -> scale *= backoff_factor, growth_tracker = 0; else growth_tracker += 1 and after growth_interval clean
 * steps scale *= growth_factor; leaves inv_scale = 1 / scale and found_inf = 0.  All four words are caller-owned device scalars. */
int pdf_grad_unscale(int nchunks, const void *tab, const int *chunks, const float *inv_scale, float *found_inf, void *stream);
int pdf_scaler_update(float *scale, float *inv_scale, int *growth_tracker, float *found_inf, float growth_factor, float backoff_factor,
                      int growth_interval, void *stream);

/* ---- libs/pointops2 window attention: the CSR-by-query v2 / v3 launchers of
 * libs/pointops2/src/attention_v2/attention_cuda_kernel_v2.h and libs/pointops2/src/rpe_v2/relative_pos_encoding_cuda_kernel_v2.h,
 * same parameter lists + stream.  N = number of queries (index0_offsets has N + 1 entries), M = edges, C = h * d,
 * tables are (L, h, d, 3), rel_idx is (M, 3).  n_max (longest edge list) is accepted for parity and not needed.
 * Overwritten outputs: attn / output / grad_q / grad_attn.  Pre-zeroed by the caller: grad_k, grad_v, grad_table*. */
int pdf_attention_step1_forward_v2(int N, int M, int h, int C, unsigned n_max, const float *q, const float *k,
                                   const int *index0_offsets, const int *index1, float *attn, void *stream);
int pdf_attention_step1_backward_v2(int N, int M, int h, int C, unsigned n_max, const float *grad_out, const int *index0_offsets,
                                    const int *index1, const float *q, const float *k, float *grad_q, float *grad_k, void *stream);
int pdf_dot_prod_with_idx_forward_v3(int N, int M, int h, int hdim, unsigned n_max, const float *q, const int *index_q_offsets,
                                     const float *k, const int *index_k, const float *table_q, const float *table_k,
                                     const int *rel_idx, float *output, void *stream);
int pdf_dot_prod_with_idx_backward_v3(int N, int M, int h, int hdim, unsigned n_max, const float *grad_out, const float *q,
                                      const int *index_q_offsets, const float *k, const int *index_k, const float *table_q,
                                      const float *table_k, const int *rel_idx, float *grad_q, float *grad_k, float *grad_table_q,
                                      float *grad_table_k, void *stream);
int pdf_attention_step2_with_rel_pos_value_forward_v2(int N, int M, int h, int hdim, unsigned n_max, const float *attn, const float *v,
                                                      const int *index0_offsets, const int *index1, const float *table,
                                                      const int *rel_idx, float *output, void *stream);
int pdf_attention_step2_with_rel_pos_value_backward_v2(int N, int M, int h, int hdim, unsigned n_max, const float *grad_out,
                                                       const int *index0_offsets, const int *index1, const float *attn, const float *v,
                                                       const float *table, const int *rel_idx, float *grad_attn, float *grad_v,
                                                       float *grad_table, void *stream);

/* the three table ops with the table length L (rows of the tables) passed explicitly: per-head kernels that keep the head's
 * table slabs and the table gradients in LDS (fall back to the entry points above when d is not a power of two <= 64 or the
 * slabs exceed 64 KB) */
int pdf_dot_prod_with_idx_forward_v3_l(int N, int M, int h, int hdim, int L, const float *q, const int *index_q_offsets, const float *k,
                                       const int *index_k, const float *table_q, const float *table_k, const int *rel_idx, float *output,
                                       void *stream);
int pdf_dot_prod_with_idx_backward_v3_l(int N, int M, int h, int hdim, int L, const float *grad_out, const float *q,
                                        const int *index_q_offsets, const float *k, const int *index_k, const float *table_q,
                                        const float *table_k, const int *rel_idx, float *grad_q, float *grad_k, float *grad_table_q,
                                        float *grad_table_k, void *stream);
int pdf_attention_step2_with_rel_pos_value_backward_v2_l(int N, int M, int h, int hdim, int L, const float *grad_out,
                                                         const int *index0_offsets, const int *index1, const float *attn, const float *v,
                                                         const float *table, const int *rel_idx, float *grad_attn, float *grad_v,
                                                         float *grad_table, void *stream);

/* ---- GridSample voxel keys: replaces the numpy front half of pointcept/datasets/transform.py:813-823 and
 * fnv_hash_vec (:911-925) for a batch of scenes.  coord (n,3) f32, offset (b) cumulative ends, min_grid (b,3) int64 = per-scene
 * floor(min coord / grid) -> grid (n,3) int64 scene-relative voxel coordinates, key (n) uint64 FNV keys.  f32 = 0: float64
 * division (NumPy >= 2 promotion), f32 = 1: float32 division (NumPy 1.x). */
int pdf_grid_hash(long n, int b, const float *coord, const int *offset, double gx, double gy, double gz, int f32,
                  const long long *min_grid, long long *grid, unsigned long long *key, void *stream);

/* ---- test-time fragment voting (pointcept/engines/test.py:218-229, 243-251): pred[index[r], :] +=
 * softmax(logits[r, :]); score_sum[index[r]] += score[r]; score_cnt[index[r]] += 1.  index (n) int64, distinct within a call
 * (one point per voxel of a GridSample test fragment); score may be NULL. */
int pdf_vote_accumulate(long n, int c, const float *logits, const float *score, const long *index, float *pred,
                        float *score_sum, float *score_cnt, void *stream);

/* Fixed-radius neighbour table of a batch with itself over the kNN grid (cells >= radius): idx (n, nsample) = the first nsample
 * points of the query's scene in index order within `radius`, -1 / 1e10 padded -- the results of pdf_random_ball_query with the
 * identity permutation and min_radius 0 (stands in for torch_points_kernels.ball_query(..., mode="partial_dense") in the PDF
 * pseudo-label pass, pointpdf_v1m1_base.py:122-130).  Workspace: pdf_knn_workspace_bytes(b, n, 0); b <= 64. */
int pdf_radius_neighbors_self(int n, int nsample, float radius, const float *xyz, const int *offset, int b, int *idx, float *dist2,
                              void *workspace, long workspace_bytes, void *stream);

/* The graph stage of the PDF pseudo-label pass (pointpdf_v1m1_base.py:309-380: scipy.sparse.csgraph.minimum_spanning_tree over the
 * region's neighbour similarities, :340; connected_components of the weak tree edges, :360) for ONE scene, one workgroup.
 * u, v (E directed entries, node ids < n), w (E weights, or NULL: all equal), active (E flags, or NULL: all), nodes (n_nodes ids,
 * repeats allowed, covering every endpoint of an active entry; an entry with an endpoint outside the list or outside [0, n) is ignored).  chosen (E bytes, or NULL) = 1 for the entries of the minimum
 * spanning forest under the strict order (weight, entry index); comp (n ints, written at the listed nodes only) = the root of the
 * node's connected component (one of its node ids).  workspace: pdf_graph_forest_workspace_bytes(n, E, n_nodes) bytes, 8-byte
 * aligned.  Lists of up to 12,288 nodes keep the per-round state in LDS. */
long pdf_graph_forest_workspace_bytes(long n, long E, long n_nodes);
int pdf_graph_forest(long n, int E, const long long *u, const long long *v, const float *w, const unsigned char *active,
                     const long long *nodes, int n_nodes, int *comp, unsigned char *chosen, void *workspace, long workspace_bytes,
                     void *stream);

/* Two-component 1-D Gaussian mixture by EM in double on m SORTED values: stands in for sklearn.mixture.GaussianMixture(n_components=2)
 * .fit(tree weights) (pointpdf_v1m1_base.py:343-345; reg_covar 1e-6, tol on the mean log-likelihood, max_iter) with a deterministic
 * start (quartiles, then 2-means) instead of sklearn's randomly seeded k-means.  resp: 2 m doubles of scratch; out (8 doubles): means,
 * variances, weights of the two components, iterations run, final mean log-likelihood. */
int pdf_gmm2_1d(int m, const float *sorted_x, double *resp, double *out, int iters, double tol, double reg, void *stream);

/* Softmax over the edges of every query, per head (x, y: (M, h); index0_offsets: N + 1 entries; h <= 64): stands in for
 * torch_scatter.scatter_softmax(src, index_0, dim=0) in WindowAttention.forward (stratified_transformer_v1m1_origin.py:322-324;
 * torch_scatter is an unvendored dependency).  backward: grad_x = y * (grad_y - sum over the query of y * grad_y). */
int pdf_segment_softmax_forward(int N, int M, int h, const int *index0_offsets, const float *x, float *y, void *stream);
int pdf_segment_softmax_backward(int N, int M, int h, const int *index0_offsets, const float *y, const float *grad_y, float *grad_x,
                                 void *stream);

/* Atomic-free building blocks of the three backward launchers above (csrc/window_attention_bwd.hip), for head dim 16, L <= 64.  The
 * reference scatters with one atomicAdd per (edge, channel) (attention_cuda_kernel_v2.cu:50-93, relative_pos_encoding_cuda_kernel_v2.cu:
 * 287-340, 441-484); here every such sum is a segmented sum over the edge list grouped by query (the CSR arrays of the forward) or by key
 * (the transposed list the caller builds once per edge table: key offsets, edge ids grouped by key, and the query / rel_idx of those
 * edges -- pointcloudpdf_amd/_native.py: window_csc), in ascending edge order: bit-reproducible, outputs written (not accumulated).
 *   pdf_wa_segment_rows: out[n, c] = sum_{e in [seg_off[n], seg_off[n+1])} w[eid(e), c / 16] * (X[other[e], c] + T(rel[e])[c]),
 *       eid(e) = seg_edge ? seg_edge[e] : e;  X (rows, with `other`) and table (with `rel`, (M, 3) in SEGMENT order) are each optional.
 *   pdf_wa_table_grad:   grad_table[r, c, a] = sum_n x[n, c] * sum_{e in seg(n), rel[e][a] == r} w[eid(e), c / 16]   (x = the segment
 *       owner's row: q or grad_out over the CSR list, k over the CSC list); ws: pdf_wa_table_grad_ws_floats(N, h, L) floats.
 *   pdf_wa_grad_attn:    grad_attn[m, hh] = <grad_out[q(m), hh, :], v[index1[m], hh, :] + T(m, hh, :)>.
 * Row operands carry a row stride in floats (ldx / ldo / ldg / ldv / ld >= h * 16: q, k, v may be slices of the (N, 3 C) rows the qkv Linear
 * writes, gradients may be written into slices of one (N, 3 C) buffer) and the scales that WindowAttention applies to the query
 * (xscale on gathered / owner rows, oscale on the result, qscale on the query rows of the logits).
 * PDF_ERR_UNSUPPORTED for other head dims / longer tables: the caller keeps the atomic launchers above. */
int pdf_wa_segment_rows(int N, int h, int d, int L, const int *seg_off, const int *seg_edge, const int *other, const int *rel,
                        const float *w, const float *X, long ldx, float xscale, const float *table, float *out, long ldo, float oscale, void *stream);
/* pdf_wa_segment_rows with a visiting order of the owners (a permutation of 0 .. N - 1, or null = storage order): same sums, same results;
 * owners of one window side by side gather the same rows (stratified.BasicLayer.window_tables hands the window-sorted order over). */
int pdf_wa_segment_rows_ordered(int N, int h, int d, int L, const int *seg_off, const int *seg_edge, const int *other, const int *rel,
                                const float *w, const float *X, long ldx, float xscale, const float *table, float *out, long ldo, float oscale,
                                const int *order, void *stream);
long pdf_wa_table_grad_ws_floats(int N, int h, int L);
int pdf_wa_table_grad(int N, int h, int d, int L, const int *seg_off, const int *seg_edge, const int *rel, const float *w,
                      const float *x, long ldx, float xscale, float *ws, float *grad_table, void *stream);
int pdf_wa_grad_attn(int N, int M, int h, int d, int L, const float *grad_out, long ldg, const int *offsets, const int *index1, const float *v,
                     long ldv, const float *table, const int *rel, float *grad_attn, void *stream);
/* attention_step1_forward_cuda_launcher_v2 + dot_prod_with_idx_forward_cuda_launcher_v3 in one pass over the key rows (their sum is what
 * WindowAttention.forward feeds the softmax, stratified_transformer_v1m1_origin.py:300-321):
 * logits[m, hh] = <q[q(m), hh], k[index1[m], hh] + T_q(m, hh)> + <k[index1[m], hh], T_k(m, hh)>. */
int pdf_wa_logits_forward(int N, int M, int h, int d, int L, const float *q, const float *k, long ld, float qscale, const int *offsets,
                          const int *index1, const float *table_q, const float *table_k, const int *rel, float *out, void *stream);
/* dst (M, h) = src[edge[m], :]: edge scalars (attention weights / logit gradients) brought into the order of the key-grouped edge list. */
int pdf_wa_permute_edges(int M, int h, const float *src, const int *edge, float *dst, void *stream);
/* pdf_wa_logits_forward / pdf_wa_grad_attn with the queries a workgroup takes given by a visiting order (see pdf_wa_segment_rows_ordered);
 * every output element is the same expression: identical results. */
int pdf_wa_logits_forward_ordered(int N, int M, int h, int d, int L, const float *q, const float *k, long ld, float qscale, const int *offsets,
                                  const int *index1, const float *table_q, const float *table_k, const int *rel, float *out, const int *order,
                                  void *stream);
int pdf_wa_grad_attn_ordered(int N, int M, int h, int d, int L, const float *grad_out, long ldg, const int *offsets, const int *index1,
                             const float *v, long ldv, const float *table, const int *rel, float *grad_attn, const int *order, void *stream);

/* Edge tables of the StratifiedTransformer's window partitions (stratified_transformer_v1m1_origin.py:45-127 get_indice_pairs / grid_sample +
 * the stable sort by query of :468-536, and WindowAttention's quantised relative positions :282-292), built per QUERY instead of by pair
 * expansion + an edge-sized sort (csrc/window_edges.hip).  Inputs: int64 keys per point (fine-window key kf, coarse-window key kc, packed
 * fine-window cell wk), kf in ascending order (kf_sorted) with the point ids in that order (order_f, ties ascending), and the same for the
 * m FPS-downsampled points by coarse key (kcd_sorted, order_cd, wkd = wk[order_cd]).
 *   pdf_window_edges_count : count (n) int32 = row length of every query, seg (n, 4) int32 (16-byte aligned) = its two segments
 *   pdf_window_edges_fill  : offsets (n + 1) = exclusive scan of count -> index0 (E) int64 ascending, index1 (E) int32, rel (E, 3) int32 or
 *                            null; *flag |= 1 if a quantised offset leaves [0, vmax] (the reference asserts that range) */
/* The keys themselves (the elementwise part of :468-499 -- torch_geometric's voxel_grid on the plain coordinates for even blocks, on the
 * coordinates shifted by half a window from the batch minimum for odd ones -- and the fine-window cell of :91-94), one launch per partition:
 * xyz (n, 3), ends (scenes) int32 scene ends, lo / hi (3) float32 DEVICE = per-axis minimum / maximum of xyz -> kf, kc, wk (n) int64.
 * Same floating-point steps as torch.div(.., rounding_mode="floor" / "trunc") on float32 tensors: bit-identical keys. */
int pdf_window_keys(int n, const float *xyz, const int *ends, int scenes, const float *lo, const float *hi, float window_size, int parity,
                    long long *kf, long long *kc, long long *wk, void *stream);
int pdf_window_edges_count(int n, const long long *kf_sorted, const long long *kf, int m, const long long *kcd_sorted, const long long *kc,
                           const long long *wk, const long long *wkd, int *count, int *seg, void *stream);
int pdf_window_edges_fill(int n, const int *offsets, const int *seg, const int *order_f, const int *order_cd, const long long *wk,
                          const long long *wkd, const float *xyz, float c2w, float qs, int vmax, long long *index0, int *index1, int *rel,
                          int *flag, void *stream);

/* The PDF pseudo-label pass without a host in the loop (pointcept/recognizers/ours/pointpdf_v1m1_base.py:190-380; csrc/region_grow.hip,
 * csrc/graph_prune.hip).  All scenes of a batch per call; starts / sizes (scenes) int32 = the scenes' point ranges; neighbors (N, nsample)
 * int32 GLOBAL row ids, -1 padded (the table pdf_radius_neighbors_self writes); every size the next stage needs stays in device memory, and
 * every reduction is done here (torch's multi-block reductions are not replay-safe on this stack: docs/NOTEBOOK.md, round 5):
 *   pdf_region_stats : :190-205 -- ml_norm = (ml - min) / (max - min + 1e-6), stop = mean(score) - beta * std(score) per scene; clears mult
 *   pdf_region_seeds : :206-207 -- mult[p] += 1 for the point whose src value has rank dice[s, j] (radix select; stands in for sort + gather)
 *   pdf_region_grow  : all growth rounds of :233-305 on mult (N) int32 (in: multiplicities of the seed list, out: the region);
 *                      lists: 2 N ints of scratch; max_points: the largest scene (host value: sizes the LDS bitmaps);
 *                      info (scenes, 4) = [rounds, grew, list length, distinct points]
 *   pdf_region_edges : the region's ascending node list and the (row, col, weight) entries of its neighbour graph (:309-335, ours/utils.py:7-43)
 *                      at capacity sizes[s] / sizes[s] * nsample per scene; counts (scenes, 4) = [nodes, entries, any -1 padding, smallest id
 *                      touched]; comp / lab (N) = every point's own local id; rows_ws: 3 * n_total words; lists / grow_info:
 *                      pdf_region_grow's `lists` / `info` when its largest scene <= pdf_region_grow_list_points(), else NULL, NULL
 *   pdf_graph_forest_dev / pdf_gmm2_1d_dev: pdf_graph_forest / pdf_gmm2_1d with the sizes read from device memory ([nodes, entries] / [m])
 *   pdf_region_tree  : the forest's chosen entries compacted in entry order (tdev[s, 1] of them per scene), tdev (scenes, 2) = [nodes, tree edges]
 *   pdf_sort_floats_dev: the first tdev[s, 1] weights of every scene ascending (what the mixture fit takes); tmp: N words
 *   pdf_gmm2_weak_dev: :343-358 for every scene -- the two-component fit of the sorted weights (fit (scenes, 8) doubles as pdf_gmm2_1d) and
 *                      weak[e] = tw[e] < mean - 2 * covariance of the component with the larger mean (N bytes); resp: 2 N doubles
 *   pdf_region_mask  : :360-380 -- component sizes over the touched points, z-score > 2 -> mask (N bytes); cnt: N ints of workspace */
int pdf_region_stats(int scenes, const int *starts, const int *sizes, const float *msp, const float *ml, int score_is_ml, float beta,
                     float *ml_norm, float *stop, int *mult, void *stream);
int pdf_region_seeds(int scenes, const int *starts, const int *sizes, const float *src, const long long *dice, int num_seed, int *mult,
                     void *stream);
int pdf_region_grow(int scenes, const int *starts, const int *sizes, int max_points, const float *coord, const float *score,
                    const int *neighbors, int nsample, const float *stop, int slide_window, int max_rounds, int *mult, int *lists, float *sim,
                    int *info, void *stream);
int pdf_region_edges(int scenes, const int *starts, const int *sizes, const float *coord, const float *msp, const int *neighbors,
                     int nsample, const int *mult, const int *lists, const int *grow_info, long long *nodes_out, long long *eu, long long *ev,
                     float *ew, unsigned char *touched, int *comp, int *lab, int *counts, void *rows_ws, long n_total, void *stream);
long pdf_region_grow_list_points(void);   /* largest scene for which pdf_region_grow leaves the ascending member list in `lists` */
int pdf_region_tree(int scenes, const int *starts, const int *sizes, int nsample, const int *counts, const unsigned char *chosen,
                    const long long *eu, const long long *ev, const float *ew, long long *tu, long long *tv, float *tw, int *tdev, void *stream);
int pdf_graph_forest_dev(long n, int E, const long long *u, const long long *v, const float *w, const unsigned char *active,
                         const long long *nodes, int n_nodes, const int *dev, int *comp, unsigned char *chosen, void *workspace,
                         long workspace_bytes, void *stream);
int pdf_gmm2_1d_dev(int m_cap, const float *sorted_x, const int *m_dev, double *resp, double *out, int iters, double tol, double reg, void *stream);
/* pdf_graph_forest_dev for the scenes of a batch in ONE launch (grid = scenes).  starts / sizes: HOST arrays; u, v, w, active, chosen
 * (N * stride) / nodes, comp (N): the batch's arrays, scene s at starts[s] (* stride); dev + s * dev_stride = the scene's [nodes, entries];
 * workspace: the scenes' pdf_graph_forest_workspace_bytes(size, size * stride, size), each rounded up to 8 bytes, back to back. */
int pdf_graph_forest_batch_dev(int scenes, const int *starts, const int *sizes, int stride, const long long *u, const long long *v,
                               const float *w, const unsigned char *active, const long long *nodes, const int *dev, int dev_stride,
                               int *comp, unsigned char *chosen, void *workspace, long workspace_bytes, void *stream);
int pdf_sort_floats_dev(int scenes, const int *starts, const int *sizes, const int *tdev, const float *x, float *out, void *tmp, void *stream);
int pdf_gmm2_weak_dev(int scenes, const int *starts, const int *sizes, const int *tdev, const float *sorted_x, const float *tw, double *resp,
                      double *fit, unsigned char *weak, int iters, double tol, double reg, void *stream);
int pdf_region_mask(int scenes, const int *starts, const int *sizes, const int *lab, const unsigned char *touched, const int *counts, int *cnt,
                    unsigned char *mask, void *stream);

/* The TransitionUp head's per-scene context (point_transformer_seg.py:148-161: ``x_b.sum(0, True) / cnt`` and ``.repeat(cnt, 1)``) as one
 * launch each with a fixed summation order (csrc/scene_rows.hip; torch's multi-workgroup reduction is not replay-safe on this stack).
 * offset (scenes) int32 = the scenes' end rows; mean != 0: divide by the scene's row count (sum: the forward's mean / repeat: its backward). */
int pdf_scene_sum_rows(int scenes, const int *offset, int c, const float *x, long ldx, int mean, float *out, void *stream);
int pdf_scene_repeat_rows(int scenes, const int *offset, long n, int c, const float *rows, int mean, float *out, void *stream);

/* torch.nn.LayerNorm over the channel dim of (n, c) rows (StratifiedTransformer's norms: stratified_transformer_v1m1_origin.py:123-139,
 * 366-368, 566-569) as one pass per direction (csrc/layernorm.hip): forward saves mean / rstd (n each); backward writes gx, dgamma, dbeta
 * (partial: pdf_layernorm_partial_floats(n, c) floats; fixed summation order).  c % 4 == 0, c <= 512, 16-byte aligned pointers. */
int pdf_layernorm_supported(int c);
long pdf_layernorm_partial_floats(long n, int c);
int pdf_layernorm_forward(long n, int c, const float *x, const float *gamma, const float *beta, float eps, float *y, float *mean,
                          float *rstd, void *stream);
int pdf_layernorm_backward(long n, int c, const float *gy, const float *x, const float *mean, const float *rstd, const float *gamma,
                           float *gx, float *partial, float *dgamma, float *dbeta, void *stream);

/* Staging copy for hipGraph replay (pointcloudpdf_amd/engine.py: CapturedStep; no reference counterpart -- the reference issues its
 * step from Python): nseg (src -> dst, nbytes) segments in ONE launch.  src_offset (device int32, may be NULL): element offset (4-byte
 * units) added to src, read on the device; every 4-byte element has sub_const + *sub (device int32, may be NULL) subtracted. */
#define PDF_COPY_MAX_SEGS 72
typedef struct PdfCopySeg {
    const void *src;
    void *dst;
    long nbytes;            /* multiple of 4 */
    const int *src_offset;
    const int *sub;
    int sub_const;
    int src_elems;          /* > 0: number of 4-byte elements of the array src points into (reads of a device-offset window are clamped) */
} PdfCopySeg;
int pdf_stage_copy(int nseg, const PdfCopySeg *segs, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PDFOPS_H */
