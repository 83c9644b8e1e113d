// Bucketed exact farthest point sampling (placeholder wiring: forwards to the plain kernel until the
// bucket-pruned kernel lands; the entry points and workspace contract are final).
#include "pdfops_common.h"

extern "C" long pdf_fps_workspace_bytes(int b, int n_total) {
    if (b < 1 || n_total < 0) return -1;
    return (long)n_total * 4 + 256;
}

extern "C" int pdf_farthest_point_sampling_bucketed(int b, int n, int n_total, const float *xyz, const int *offset,
                                                    const int *new_offset, void *workspace, long workspace_bytes,
                                                    int *idx, void *stream) {
    if (!workspace || workspace_bytes < pdf_fps_workspace_bytes(b, n_total)) return PDF_ERR_BAD_ARG;
    return pdf_farthest_point_sampling(b, n, xyz, offset, new_offset, static_cast<float *>(workspace), idx, stream);
}
