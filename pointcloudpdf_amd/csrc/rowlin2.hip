// Streaming per-point Linear layers, fp32 operands (MP = 0) + the precision-independent helpers and the dispatch on the call's
// product-input mode (`mma_input` of the C entry points, include/pdfops.h).  Kernels: rowlin2_impl.h.
#define RL2_MAIN_TU
#include "rowlin2_impl.h"

namespace rl2 {
template int try_forward_mp<0>(long, int, int, int, int, const float *const *, long, const float *const *, int, const float *const *, const float *,
                               const float *, int, float *const *, long, int, float *, hipStream_t, const float *, long, const float *, long,
                               const float *, int, int *, long, void *);
template int try_wgrad_mp<0>(long, int, int, int, const float *const *, long, const float *, long, const float *, const float *, int,
                             float *const *, float *const *, float *, hipStream_t, const float *, long);
template int try_wgrad_group_mp<0>(long, int, int, int, const float *const *, long, const float *const *, long, const float *const *, const float *const *, const int *,
                                   float *const *, float *const *, float *, hipStream_t);
extern template int try_wgrad_group_mp<1>(long, int, int, int, const float *const *, long, const float *const *, long, const float *const *, const float *const *, const int *,
                                   float *const *, float *const *, float *, hipStream_t);
extern template int try_wgrad_group_mp<2>(long, int, int, int, const float *const *, long, const float *const *, long, const float *const *, const float *const *, const int *,
                                   float *const *, float *const *, float *, hipStream_t);
extern template int try_forward_mp<1>(long, int, int, int, int, const float *const *, long, const float *const *, int, const float *const *, const float *,
                                      const float *, int, float *const *, long, int, float *, hipStream_t, const float *, long, const float *, long,
                                      const float *, int, int *, long, void *);
extern template int try_forward_mp<2>(long, int, int, int, int, const float *const *, long, const float *const *, int, const float *const *, const float *,
                                      const float *, int, float *const *, long, int, float *, hipStream_t, const float *, long, const float *, long,
                                      const float *, int, int *, long, void *);
extern template int try_wgrad_mp<1>(long, int, int, int, const float *const *, long, const float *, long, const float *, const float *, int,
                                    float *const *, float *const *, float *, hipStream_t, const float *, long);
extern template int try_wgrad_mp<2>(long, int, int, int, const float *const *, long, const float *, long, const float *, const float *, int,
                                    float *const *, float *const *, float *, hipStream_t, const float *, long);

// returns 1 when a streaming kernel took the job, 0 when the shape is not covered (caller falls back to the tiled kernel)
int try_forward(long n, int k, int o, int nin, int nout, const float *const *x, long ldx, const float *const *w, int transpose_w,
                const float *const *bias, const float *scale, const float *shift, int relu, float *const *y, long ldy,
                int accumulate, float *partial, int mma, hipStream_t s, const float *roww, long rws, const float *bx, long ldb, const float *bcoef,
                int brelu, int *partial_rows, long ldw, void *handoff) {
#define RL2_FWD(MP_) try_forward_mp<MP_>(n, k, o, nin, nout, x, ldx, w, transpose_w, bias, scale, shift, relu, y, ldy, accumulate, partial, s, roww, rws, \
                                         bx, ldb, bcoef, brelu, partial_rows, ldw, handoff)
    switch (mma) {
    case 1: return RL2_FWD(1);
    case 2: return RL2_FWD(2);
    default: return RL2_FWD(0);
    }
#undef RL2_FWD
}

int try_wgrad(long n, int k, int o, int ng, const float *const *g, long ldg, const float *x, long ldx, const float *scale,
              const float *shift, int relu, float *const *dw, float *const *db, float *ws, int mma, hipStream_t s, const float *roww, long rws) {
    switch (mma) {
    case 1: return try_wgrad_mp<1>(n, k, o, ng, g, ldg, x, ldx, scale, shift, relu, dw, db, ws, s, roww, rws);
    case 2: return try_wgrad_mp<2>(n, k, o, ng, g, ldg, x, ldx, scale, shift, relu, dw, db, ws, s, roww, rws);
    default: return try_wgrad_mp<0>(n, k, o, ng, g, ldg, x, ldx, scale, shift, relu, dw, db, ws, s, roww, rws);
    }
}

int try_wgrad_group(long n, int k, int o, int ng, const float *const *g, long ldg, const float *const *x, long ldx, const float *const *scale,
                    const float *const *shift, const int *relu, float *const *dw, float *const *db, float *ws, int mma, hipStream_t s) {
    switch (mma) {
    case 1: return try_wgrad_group_mp<1>(n, k, o, ng, g, ldg, x, ldx, scale, shift, relu, dw, db, ws, s);
    case 2: return try_wgrad_group_mp<2>(n, k, o, ng, g, ldg, x, ldx, scale, shift, relu, dw, db, ws, s);
    default: return try_wgrad_group_mp<0>(n, k, o, ng, g, ldg, x, ldx, scale, shift, relu, dw, db, ws, s);
    }
}
}  // namespace rl2
