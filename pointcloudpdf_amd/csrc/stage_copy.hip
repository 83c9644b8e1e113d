// Staging copy for hipGraph replay of a captured training step (engine.CapturedStep): the ~70 coordinate-only tables of a batch
// (geometry.Geometry: level coordinates, kNN / interpolation / TransitionDown tables, inverse tables, visiting orders, coordinate
// sums) move from wherever the pre-pass left them -- row slices of a grouped pre-pass's arrays -- into the fixed-address buffer the
// captured kernels read, in ONE launch.  A segment may carry the two fix-ups a slice of a group's inverse table needs (csrc/
// seg_gather.hip): a source window that starts at a position only known on the device (the slice's first offset) and an int32 value
// subtracted from every element (that same offset for the offset array, the batch's first entry id for the entry list).
// Bound: HBM (read + write of ~110 MB per 2 x 100k-point batch: ~45 us).
#include "pdfops_common.h"

namespace {

constexpr int SB = 256, CHUNK = 16384;   // threads per workgroup, bytes per workgroup

struct Table {
    PdfCopySeg seg[PDF_COPY_MAX_SEGS];
    unsigned first_chunk[PDF_COPY_MAX_SEGS + 1];
    int nseg;
};

__global__ __launch_bounds__(SB) void k_stage_copy(const Table t) {
    int lo = 0, hi = t.nseg;   // last segment whose first chunk is <= blockIdx.x
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (t.first_chunk[mid] <= blockIdx.x) lo = mid; else hi = mid;
    }
    const PdfCopySeg s = t.seg[lo];
    const long base = (long)(blockIdx.x - t.first_chunk[lo]) * CHUNK;
    const long left = s.nbytes - base;
    const long n = left < CHUNK ? left : CHUNK;
    const long soff = s.src_offset ? 4L * (long)(*s.src_offset) : 0L;
    const int sub = s.sub_const + (s.sub ? *s.sub : 0);
    const char *src = static_cast<const char *>(s.src) + soff + base;
    char *dst = static_cast<char *>(s.dst) + base;
    if (sub == 0 && !s.src_offset && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0 && n == CHUNK) {
        int4 v[CHUNK / (16 * SB)];
#pragma unroll
        for (int k = 0; k < CHUNK / (16 * SB); ++k) v[k] = reinterpret_cast<const int4 *>(src)[threadIdx.x + k * SB];
#pragma unroll
        for (int k = 0; k < CHUNK / (16 * SB); ++k) reinterpret_cast<int4 *>(dst)[threadIdx.x + k * SB] = v[k];
    } else {   // 4-byte elements (every table is int32 / float32 / float64): tails, unaligned slices, the subtracting segments
        const long n4 = n >> 2;
        // a window that starts at a device-side offset must not leave its source array (src_elems; the tail of such a window is never
        // read by the consumer: entries of placeholder rows).  The clamp is on the ABSOLUTE element index: a chunk that lies wholly past
        // the array re-reads the array's last element (round 3 clamped relative to the chunk and read s.src[soff + base] itself then).
        const int *arr = static_cast<const int *>(s.src);
        const long first = (soff + base) >> 2;
        const long avail_last = s.src_elems > 0 ? (long)s.src_elems - 1 : first + n4 - 1;
        for (long e = threadIdx.x; e < n4; e += SB) {
            long a = first + e;
            a = a <= avail_last ? a : avail_last;
            a = a > 0 ? a : 0;
            reinterpret_cast<int *>(dst)[e] = arr[a] - sub;
        }
    }
}

}  // namespace

extern "C" int pdf_stage_copy(int nseg, const PdfCopySeg *segs, void *stream) {
    if (nseg == 0) return PDF_OK;
    if (nseg < 0 || !segs) return PDF_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int at = 0; at < nseg; at += PDF_COPY_MAX_SEGS) {
        Table t;
        t.nseg = nseg - at < PDF_COPY_MAX_SEGS ? nseg - at : PDF_COPY_MAX_SEGS;
        unsigned chunks = 0;
        for (int i = 0; i < t.nseg; ++i) {
            t.seg[i] = segs[at + i];
            if (t.seg[i].nbytes < 0 || (t.seg[i].nbytes & 3) || (t.seg[i].nbytes && (!t.seg[i].src || !t.seg[i].dst))) return PDF_ERR_BAD_ARG;
            t.first_chunk[i] = chunks;
            chunks += (unsigned)((t.seg[i].nbytes + CHUNK - 1) / CHUNK);
        }
        t.first_chunk[t.nseg] = chunks;
        if (chunks) k_stage_copy<<<chunks, SB, 0, s>>>(t);
    }
    return pdf_launch_status();
}
