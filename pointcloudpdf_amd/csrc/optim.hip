// SGD with momentum and weight decay over ALL parameter tensors of the model in ONE launch (torch.optim.SGD semantics, dampening = 0,
// no Nesterov: engines/defaults build the reference's optimizer from configs/_base_/default_runtime + the dataset config: SGD,
// momentum 0.9, weight_decay 1e-4):   g' = g + wd * p;   buf = momentum * buf + g';   p -= lr * buf.
// torch's fused multi-tensor SGD packs ~24 tensors per launch: 13 launches and 275 us per step for this model's 304 tensors (8.5 M
// values, 170 MB of traffic = 21 us at the HBM peak).  Here the host hands over a table of (param, grad, momentum, length) and a
// chunk list; one workgroup = one chunk of CH values of one tensor.  Bound: HBM.
#include "pdfops_common.h"

namespace {

constexpr int OB = 256, CH = 4096;   // threads per workgroup, values per chunk

struct SgdTensor { float *p; const float *g; float *m; long n; };

__global__ __launch_bounds__(OB) void k_sgd(const SgdTensor *__restrict__ tab, const int2 *__restrict__ chunks, float lr, float momentum,
                                            float wd, const float *__restrict__ found_inf) {
    if (found_inf && *found_inf != 0.f) return;        // (uniform: a GradScaler step with a non-finite gradient changes nothing)
    const int2 c = chunks[blockIdx.x];                 // (tensor, chunk inside the tensor)
    const SgdTensor t = tab[c.x];
    const long base = (long)c.y * CH;
    const long left = t.n - base;
    float *p = t.p + base, *m = t.m + base;
    const float *g = t.g + base;
    if (left >= CH && ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m)) & 15) == 0) {
        float4 pv[CH / (4 * OB)], gv[CH / (4 * OB)], mv[CH / (4 * OB)];
#pragma unroll
        for (int k = 0; k < CH / (4 * OB); ++k) {
            const int e = threadIdx.x + k * OB;
            pv[k] = reinterpret_cast<const float4 *>(p)[e]; gv[k] = reinterpret_cast<const float4 *>(g)[e]; mv[k] = reinterpret_cast<const float4 *>(m)[e];
        }
#pragma unroll
        for (int k = 0; k < CH / (4 * OB); ++k) {
            const int e = threadIdx.x + k * OB;
            float4 b;
            b.x = momentum * mv[k].x + (gv[k].x + wd * pv[k].x); b.y = momentum * mv[k].y + (gv[k].y + wd * pv[k].y);
            b.z = momentum * mv[k].z + (gv[k].z + wd * pv[k].z); b.w = momentum * mv[k].w + (gv[k].w + wd * pv[k].w);
            reinterpret_cast<float4 *>(m)[e] = b;
            reinterpret_cast<float4 *>(p)[e] = make_float4(pv[k].x - lr * b.x, pv[k].y - lr * b.y, pv[k].z - lr * b.z, pv[k].w - lr * b.w);
        }
    } else {
        const long n = left < CH ? left : CH;
        for (long e = threadIdx.x; e < n; e += OB) {
            const float pe = p[e];
            const float b = momentum * m[e] + (g[e] + wd * pe);
            m[e] = b;
            p[e] = pe - lr * b;
        }
    }
}

// GradScaler.unscale_ (torch/amp/grad_scaler.py: _unscale_grads_ -> _amp_foreach_non_finite_check_and_unscale_) over the same tables:
// g *= inv_scale in place; found_inf = 1 if any gradient value is inf / nan (every workgroup that sees one stores the same 1.0f: no
// atomics, no ordering needed).  found_inf must be 0 on entry (pdf_scaler_update leaves it so).
__global__ __launch_bounds__(OB) void k_unscale(const SgdTensor *__restrict__ tab, const int2 *__restrict__ chunks,
                                                const float *__restrict__ inv_scale, float *found_inf) {
    const int2 c = chunks[blockIdx.x];
    const SgdTensor t = tab[c.x];
    const long base = (long)c.y * CH;
    const long left = t.n - base;
    float *g = const_cast<float *>(t.g) + base;
    const float is = *inv_scale;
    bool bad = false;
    if (left >= CH && (reinterpret_cast<uintptr_t>(g) & 15) == 0) {
        float4 gv[CH / (4 * OB)];
#pragma unroll
        for (int k = 0; k < CH / (4 * OB); ++k) gv[k] = reinterpret_cast<const float4 *>(g)[threadIdx.x + k * OB];
#pragma unroll
        for (int k = 0; k < CH / (4 * OB); ++k) {
            bad = bad || !(isfinite(gv[k].x) && isfinite(gv[k].y) && isfinite(gv[k].z) && isfinite(gv[k].w));
            reinterpret_cast<float4 *>(g)[threadIdx.x + k * OB] = make_float4(gv[k].x * is, gv[k].y * is, gv[k].z * is, gv[k].w * is);
        }
    } else {
        const long n = left < CH ? left : CH;
        for (long e = threadIdx.x; e < n; e += OB) {
            const float v = g[e];
            bad = bad || !isfinite(v);
            g[e] = v * is;
        }
    }
    if (__syncthreads_or(bad) && threadIdx.x == 0) *found_inf = 1.f;
}

// GradScaler.update (torch/amp/grad_scaler.py: _amp_update_scale_): found_inf -> scale *= backoff, tracker = 0; else tracker += 1 and
// at `interval` consecutive clean steps scale *= growth, tracker = 0.  Leaves inv_scale = 1 / scale and found_inf = 0 for the next step.
__global__ void k_scaler_update(float *scale, float *inv_scale, int *tracker, float *found_inf, float growth, float backoff, int interval) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float s = *scale;
    if (*found_inf != 0.f) {
        s *= backoff;
        *tracker = 0;
    } else {
        const int t = *tracker + 1;
        if (t >= interval) {
            const float grown = s * growth;
            if (isfinite(grown)) s = grown;   // (torch keeps the scale when growing would overflow)
            *tracker = 0;
        } else {
            *tracker = t;
        }
    }
    *scale = s;
    *inv_scale = 1.f / s;
    *found_inf = 0.f;
}

}  // namespace

extern "C" int pdf_sgd_chunk(void) { return CH; }

// tab: ntensors x {param*, grad*, momentum*, length} (device, 32 bytes each); chunks: nchunks x {tensor, chunk} int32 pairs (device).
// found_inf (device float, nullable): non-zero = the whole step is skipped (GradScaler.step with a non-finite gradient).
extern "C" int pdf_sgd_step(int nchunks, const void *tab, const int *chunks, float lr, float momentum, float weight_decay, const float *found_inf,
                            void *stream) {
    if (nchunks == 0) return PDF_OK;
    if (nchunks < 0 || !tab || !chunks) return PDF_ERR_BAD_ARG;
    k_sgd<<<nchunks, OB, 0, static_cast<hipStream_t>(stream)>>>(static_cast<const SgdTensor *>(tab), reinterpret_cast<const int2 *>(chunks), lr, momentum,
                                                              weight_decay, found_inf);
    return pdf_launch_status();
}

extern "C" int pdf_grad_unscale(int nchunks, const void *tab, const int *chunks, const float *inv_scale, float *found_inf, void *stream) {
    if (nchunks == 0) return PDF_OK;
    if (nchunks < 0 || !tab || !chunks || !inv_scale || !found_inf) return PDF_ERR_BAD_ARG;
    k_unscale<<<nchunks, OB, 0, static_cast<hipStream_t>(stream)>>>(static_cast<const SgdTensor *>(tab), reinterpret_cast<const int2 *>(chunks), inv_scale,
                                                                  found_inf);
    return pdf_launch_status();
}

extern "C" int pdf_scaler_update(float *scale, float *inv_scale, int *growth_tracker, float *found_inf, float growth_factor, float backoff_factor,
                                 int growth_interval, void *stream) {
    if (!scale || !inv_scale || !growth_tracker || !found_inf || !(growth_factor >= 1.f) || !(backoff_factor > 0.f && backoff_factor <= 1.f) ||
        growth_interval < 1)
        return PDF_ERR_BAD_ARG;
    k_scaler_update<<<1, 64, 0, static_cast<hipStream_t>(stream)>>>(scale, inv_scale, growth_tracker, found_inf, growth_factor, backoff_factor, growth_interval);
    return pdf_launch_status();
}
