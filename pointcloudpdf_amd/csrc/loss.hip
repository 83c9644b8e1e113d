// Cross-entropy over (N, C) logits with an ignore label, mean over the counted rows -- nn.CrossEntropyLoss as configured on
// this path (losses/misc.py:14-39: weight=None, label_smoothing=0, reduction="mean", ignore_index=-1).
// torch runs log_softmax + an nll reduction that is a single-block kernel (173 us forward + 147 us backward for 200k x 13);
// here one lane owns one row (C <= 64 logits in registers): max / log-sum-exp / picked logit, a block reduction into the block's own
// slot and a one-workgroup sum over the slots (sum of losses, number of counted rows; fixed order, no atomics).  The forward also leaves softmax - onehot (zero on ignored rows) in
// `grad`, so the backward is one scaled copy.  Bound: HBM (8NC bytes forward).
#include "pdfops_common.h"

namespace {

constexpr int LB = 256;
constexpr int PDF_CE_HEAD = 4, PDF_CE_MAX_BLOCKS = 1024;   // acc = [sum, count, -, -] + one [sum | count] pair per workgroup

__global__ __launch_bounds__(LB) void k_ce_fwd(long n, int c, const float *__restrict__ logits, const long *__restrict__ target,
                                               long ignore, float *__restrict__ grad, float *__restrict__ acc) {
    __shared__ float red[2][LB / 64];
    float loss = 0.f, cnt = 0.f;
    for (long r = (long)blockIdx.x * LB + threadIdx.x; r < n; r += (long)gridDim.x * LB) {
        const float *x = logits + r * c;
        const long t = target[r];
        float m = x[0];
        for (int j = 1; j < c; ++j) m = fmaxf(m, x[j]);
        float s = 0.f;
        for (int j = 0; j < c; ++j) s += __expf(x[j] - m);
        const float lse = m + __logf(s);
        const bool counted = t != ignore && t >= 0 && t < c;
        if (counted) { loss += lse - x[t]; cnt += 1.f; }
        // a label that is neither the ignore value nor a class id (torch: device-side assert) poisons the loss instead of being
        // dropped silently: a mis-sized head / label map shows up as a NaN loss on the first step, without a host sync
        else if (t != ignore) loss += __builtin_nanf("");
        float *g = grad + r * c;
        const float inv = 1.f / s;
        for (int j = 0; j < c; ++j) g[j] = counted ? __expf(x[j] - m) * inv - (j == t ? 1.f : 0.f) : 0.f;
    }
    loss = pdf_wave_sum_f32(loss);
    cnt = pdf_wave_sum_f32(cnt);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = loss; red[1][threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {   // the block's [sum | count] goes to its own slot: k_ce_mean adds the slots in a fixed order (no float atomics)
        float a = 0.f, b = 0.f;
        for (int w = 0; w < LB / 64; ++w) { a += red[0][w]; b += red[1][w]; }
        acc[PDF_CE_HEAD + 2 * blockIdx.x] = a;
        acc[PDF_CE_HEAD + 2 * blockIdx.x + 1] = b;
    }
}

// acc[PDF_CE_HEAD + 2 g] = the blocks' [sum of losses, counted rows] -> acc[0..1] = their totals, out = mean loss (NaN when nothing is
// counted, as torch).  One workgroup; lane l adds blocks l, l + 256, ...; the 256 partial sums are combined in lane order.
__global__ __launch_bounds__(LB) void k_ce_mean(float *__restrict__ acc, int blocks, float *__restrict__ out) {
    __shared__ float red[2][LB];
    float a = 0.f, b = 0.f;
    for (int g = threadIdx.x; g < blocks; g += LB) { a += acc[PDF_CE_HEAD + 2 * g]; b += acc[PDF_CE_HEAD + 2 * g + 1]; }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = b;
    __syncthreads();
    if (threadIdx.x != 0) return;
    a = 0.f; b = 0.f;
    for (int t = 0; t < LB; ++t) { a += red[0][t]; b += red[1][t]; }
    acc[0] = a; acc[1] = b;
    out[0] = a / b;
}

// grad_logits = (softmax - onehot) * gy / count.  The forward's buffer is only READ: a second backward over the same graph
// (retain_graph, torch.autograd.grad followed by backward) must see the unscaled values again.
__global__ __launch_bounds__(LB) void k_ce_bwd(long total, const float *__restrict__ dlogits, const float *__restrict__ acc,
                                               const float *__restrict__ gy, float *__restrict__ out) {
    const float scale = gy[0] / acc[1];
    for (long e = (long)blockIdx.x * LB + threadIdx.x; e < total; e += (long)gridDim.x * LB) out[e] = dlogits[e] * scale;
}

// Test-time fragment voting (engines/test.py:218-229, 243-251): pred[index[r], :] += softmax(logits[r, :]) and the running
// sum / count behind scatter_mean(score, index).  The points of ONE fragment are distinct (one per voxel), so a call needs no
// atomics; fragments are accumulated by successive calls on the same stream.  lane = row, logits in registers.
__global__ __launch_bounds__(LB) void k_vote(long n, int c, const float *__restrict__ logits, const float *__restrict__ score,
                                             const long *__restrict__ index, float *__restrict__ pred, float *__restrict__ score_sum,
                                             float *__restrict__ score_cnt) {
    for (long r = (long)blockIdx.x * LB + threadIdx.x; r < n; r += (long)gridDim.x * LB) {
        const float *x = logits + r * c;
        float m = x[0];
        for (int j = 1; j < c; ++j) m = fmaxf(m, x[j]);
        float s = 0.f;
        for (int j = 0; j < c; ++j) s += __expf(x[j] - m);
        const float inv = 1.f / s;
        const long dst = index[r];
        float *p = pred + dst * c;
        for (int j = 0; j < c; ++j) p[j] += __expf(x[j] - m) * inv;
        if (score) { score_sum[dst] += score[r]; score_cnt[dst] += 1.f; }
    }
}

}  // namespace

// pred (N_full, c), score_sum / score_cnt (N_full) accumulate one fragment: logits (n, c), score (n) or NULL, index (n) int64 with
// DISTINCT entries.  Replaces the python accumulation of engines/test.py:218-229 + the scatter_mean inputs of :243-251.
extern "C" int pdf_vote_accumulate(long n, int c, const float *logits, const float *score, const long *index, float *pred,
                                   float *score_sum, float *score_cnt, void *stream) {
    if (n == 0) return PDF_OK;
    if (n < 0 || c < 1 || !logits || !index || !pred || (score && (!score_sum || !score_cnt))) return PDF_ERR_BAD_ARG;
    long g = (n + LB - 1) / LB;
    if (g > 2048) g = 2048;
    k_vote<<<(unsigned)g, LB, 0, static_cast<hipStream_t>(stream)>>>(n, c, logits, score, index, pred, score_sum, score_cnt);
    return pdf_launch_status();
}

extern "C" long pdf_ce_workspace_floats(void) { return PDF_CE_HEAD + 2L * PDF_CE_MAX_BLOCKS; }

// loss (1 float) = mean over rows with target != ignore of -log softmax(logits)[target]; grad (n*c) receives softmax - onehot;
// acc (pdf_ce_workspace_floats() floats) receives [sum, count] in its first two floats (the rest: per-workgroup partial sums, added in a
// fixed order: the loss is bit-reproducible).  Nothing needs zeroing.
extern "C" int pdf_ce_forward(long n, int c, const float *logits, const long *target, long ignore, float *grad, float *acc,
                              float *loss, void *stream) {
    if (n < 1 || c < 1 || !logits || !target || !grad || !acc || !loss) return PDF_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    long g = (n + LB - 1) / LB;
    if (g > PDF_CE_MAX_BLOCKS) g = PDF_CE_MAX_BLOCKS;
    k_ce_fwd<<<(unsigned)g, LB, 0, s>>>(n, c, logits, target, ignore, grad, acc);
    k_ce_mean<<<1, LB, 0, s>>>(acc, (int)g, loss);
    return pdf_launch_status();
}

extern "C" int pdf_ce_backward(long n, int c, const float *dlogits, const float *acc, const float *gy, float *grad_out, void *stream) {
    if (n < 1 || c < 1 || !dlogits || !acc || !gy || !grad_out) return PDF_ERR_BAD_ARG;
    long g = (n * c + LB - 1) / LB;
    if (g > 2048) g = 2048;
    k_ce_bwd<<<(unsigned)g, LB, 0, static_cast<hipStream_t>(stream)>>>(n * c, dlogits, acc, gy, grad_out);
    return pdf_launch_status();
}
