// Streaming per-point Linear layers with fp16 product operands (MP = 1): kernels and dispatch in rowlin2_impl.h / rowlin2.hip.
#include "rowlin2_impl.h"

namespace rl2 {
template int try_forward_mp<1>(long, int, int, int, int, const float *const *, long, const float *const *, int, const float *const *, const float *,
                               const float *, int, float *const *, long, int, float *, hipStream_t, const float *, long, const float *, long,
                               const float *, int, int *, long, void *);
template int try_wgrad_mp<1>(long, int, int, int, const float *const *, long, const float *, long, const float *, const float *, int,
                             float *const *, float *const *, float *, hipStream_t, const float *, long);
template int try_wgrad_group_mp<1>(long, int, int, int, const float *const *, long, const float *const *, long, const float *const *, const float *const *, const int *,
                                   float *const *, float *const *, float *, hipStream_t);
}  // namespace rl2
