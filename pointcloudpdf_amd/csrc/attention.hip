// Edge-list attention steps for gfx950 -- replaces libs/pointops/src/attention/attention_cuda_kernel.cu:9-147.
//
// The reference launches one thread per (edge, group, channel) on a (ceil(m/512), g, c) grid and funnels
// every product through atomicAdd.  Here one lane owns an (edge, group) pair and walks the c channels of
// that group (contiguous in memory), so the relation-step forward and the fusion-step grad_weight need no
// atomics at all; only true scatters (rows addressed through index_target / index_refer) use the hardware
// fp32 atomic.  grad_weight[c] of the relation step is reduced per wave before touching memory.
// HBM-bound; no in-tree caller in the reference (API completeness).
#include "pdfops_common.h"

namespace {

constexpr int AB = 256;
static inline int agrid(long total) {
    long g = (total + AB - 1) / AB;
    if (g > PDF_MAX_BLOCKS) g = PDF_MAX_BLOCKS;
    if (g < 1) g = 1;
    return (int)g;
}

// attention_cuda_kernel.cu:9-24 : out[r,g] += sum_c q[tgt[r],g,c] * k[ref[r],g,c] * w[c]
__global__ __launch_bounds__(AB) void rel_fwd_kernel(long m, int g, int c, const float *__restrict__ query,
                                                     const float *__restrict__ key, const float *__restrict__ weight,
                                                     const int *__restrict__ index_target,
                                                     const int *__restrict__ index_refer, float *__restrict__ output) {
    const long total = m * g;
    for (long e = (long)blockIdx.x * AB + threadIdx.x; e < total; e += (long)gridDim.x * AB) {
        const long r = e / g;
        const int gi = (int)(e - r * g);
        const float *q = query + ((long)index_target[r] * g + gi) * c;
        const float *k = key + ((long)index_refer[r] * g + gi) * c;
        float acc = 0.f;
        for (int ci = 0; ci < c; ++ci) acc += q[ci] * k[ci] * weight[ci];
        output[e] += acc;  // pre-zeroed accumulate target (reference contract, attention.py:33)
    }
}

// attention_cuda_kernel.cu:26-47
__global__ __launch_bounds__(AB) void rel_bwd_kernel(long m, int g, int c, const float *__restrict__ query,
                                                     float *__restrict__ grad_query, const float *__restrict__ key,
                                                     float *__restrict__ grad_key, const float *__restrict__ weight,
                                                     float *__restrict__ grad_weight,
                                                     const int *__restrict__ index_target,
                                                     const int *__restrict__ index_refer,
                                                     const float *__restrict__ grad_output) {
    const long total = m * g;
    const long span = (long)gridDim.x * AB;
    // every lane of a wave runs the same number of iterations so the wave reduction below is convergent
    const long iters = (total + span - 1) / span;
    for (long it = 0; it < iters; ++it) {
        const long e = it * span + (long)blockIdx.x * AB + threadIdx.x;
        const bool valid = e < total;
        const long r = valid ? e / g : 0;
        const int gi = valid ? (int)(e - r * g) : 0;
        const long qo = ((long)index_target[r] * g + gi) * c;
        const long ko = ((long)index_refer[r] * g + gi) * c;
        const float go = valid ? grad_output[e] : 0.f;
        for (int ci = 0; ci < c; ++ci) {
            const float qv = query[qo + ci], kv = key[ko + ci], wv = weight[ci];
            if (valid) {
                pdf_atomic_add(grad_query + qo + ci, go * kv * wv);
                pdf_atomic_add(grad_key + ko + ci, go * qv * wv);
            }
            const float gw = pdf_wave_sum_f32(valid ? go * kv * qv : 0.f);
            if (pdf_lane() == 0) pdf_atomic_add(grad_weight + ci, gw);
        }
    }
}

// attention_cuda_kernel.cu:50-66 : out[tgt[r],g,c] += w[r,g] * v[ref[r],g,c]
__global__ __launch_bounds__(AB) void fus_fwd_kernel(long m, int g, int c, const float *__restrict__ weight,
                                                     const float *__restrict__ value,
                                                     const int *__restrict__ index_target,
                                                     const int *__restrict__ index_refer, float *__restrict__ output) {
    const long total = m * g * c;
    for (long e = (long)blockIdx.x * AB + threadIdx.x; e < total; e += (long)gridDim.x * AB) {
        const long rg = e / c;
        const int ci = (int)(e - rg * c);
        const long r = rg / g;
        const int gi = (int)(rg - r * g);
        const float f = weight[rg] * value[((long)index_refer[r] * g + gi) * c + ci];
        pdf_atomic_add(output + ((long)index_target[r] * g + gi) * c + ci, f);
    }
}

// attention_cuda_kernel.cu:69-86
__global__ __launch_bounds__(AB) void fus_bwd_kernel(long m, int g, int c, const float *__restrict__ weight,
                                                     float *__restrict__ grad_weight, const float *__restrict__ value,
                                                     float *__restrict__ grad_value,
                                                     const int *__restrict__ index_target,
                                                     const int *__restrict__ index_refer,
                                                     const float *__restrict__ grad_output) {
    const long total = m * g;
    for (long e = (long)blockIdx.x * AB + threadIdx.x; e < total; e += (long)gridDim.x * AB) {
        const long r = e / g;
        const int gi = (int)(e - r * g);
        const long oo = ((long)index_target[r] * g + gi) * c;
        const long vo = ((long)index_refer[r] * g + gi) * c;
        const float w = weight[e];
        float gw = 0.f;
        for (int ci = 0; ci < c; ++ci) {
            const float go = grad_output[oo + ci];
            gw += go * value[vo + ci];
            pdf_atomic_add(grad_value + vo + ci, go * w);
        }
        grad_weight[e] += gw;  // pre-zeroed accumulate target (attention.py:100)
    }
}

}  // namespace

extern "C" int pdf_attention_relation_step_forward(int m, int g, int c, const float *query, const float *key,
                                                   const float *weight, const int *index_target,
                                                   const int *index_refer, float *output, void *stream) {
    if (m < 0 || g < 1 || c < 1 || !query || !key || !weight || !index_target || !index_refer || !output) return PDF_ERR_BAD_ARG;
    if (m == 0) return PDF_OK;
    rel_fwd_kernel<<<agrid((long)m * g), AB, 0, static_cast<hipStream_t>(stream)>>>(m, g, c, query, key, weight, index_target, index_refer, output);
    return pdf_launch_status();
}

extern "C" int pdf_attention_relation_step_backward(int m, int g, int c, const float *query, float *grad_query,
                                                    const float *key, float *grad_key, const float *weight,
                                                    float *grad_weight, const int *index_target,
                                                    const int *index_refer, const float *grad_output, void *stream) {
    if (m < 0 || g < 1 || c < 1 || !query || !grad_query || !key || !grad_key || !weight || !grad_weight ||
        !index_target || !index_refer || !grad_output)
        return PDF_ERR_BAD_ARG;
    if (m == 0) return PDF_OK;
    rel_bwd_kernel<<<agrid((long)m * g), AB, 0, static_cast<hipStream_t>(stream)>>>(m, g, c, query, grad_query, key, grad_key, weight,
                                                                                    grad_weight, index_target, index_refer, grad_output);
    return pdf_launch_status();
}

extern "C" int pdf_attention_fusion_step_forward(int m, int g, int c, const float *weight, const float *value,
                                                 const int *index_target, const int *index_refer, float *output,
                                                 void *stream) {
    if (m < 0 || g < 1 || c < 1 || !weight || !value || !index_target || !index_refer || !output) return PDF_ERR_BAD_ARG;
    if (m == 0) return PDF_OK;
    fus_fwd_kernel<<<agrid((long)m * g * c), AB, 0, static_cast<hipStream_t>(stream)>>>(m, g, c, weight, value, index_target, index_refer, output);
    return pdf_launch_status();
}

extern "C" int pdf_attention_fusion_step_backward(int m, int g, int c, const float *weight, float *grad_weight,
                                                  const float *value, float *grad_value, const int *index_target,
                                                  const int *index_refer, const float *grad_output, void *stream) {
    if (m < 0 || g < 1 || c < 1 || !weight || !grad_weight || !value || !grad_value || !index_target ||
        !index_refer || !grad_output)
        return PDF_ERR_BAD_ARG;
    if (m == 0) return PDF_OK;
    fus_bwd_kernel<<<agrid((long)m * g), AB, 0, static_cast<hipStream_t>(stream)>>>(m, g, c, weight, grad_weight, value, grad_value,
                                                                                    index_target, index_refer, grad_output);
    return pdf_launch_status();
}
