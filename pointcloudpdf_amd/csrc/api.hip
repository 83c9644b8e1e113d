// Version / build probes of libpdfops.so (callable without a GPU).
#include "pdfops_common.h"

#define PDF_ABI_VERSION 1

extern "C" int pdf_abi_version(void) { return PDF_ABI_VERSION; }

extern "C" const char *pdf_build_info(void) {
    return "libpdfops abi=1 target=gfx950 wave64 hipcc " __VERSION__;
}

// Input precision of the matrix-core products of the streaming Linear kernels (csrc/rowlin2_impl.h: Mma): 0 = fp32 operands (default; the
// parity path), 1 = operands rounded to fp16, 2 = to bfloat16 -- fp32 storage and fp32 accumulation in every mode.  Process-wide, like
// torch.backends.cuda.matmul's switches: read when a launch is issued (a captured graph keeps the mode it was captured with).
#include <atomic>
namespace { std::atomic<int> g_mma_input{0}; }
extern "C" int pdf_set_mma_input(int mode) {
    if (mode < 0 || mode > 2) return PDF_ERR_BAD_ARG;
    g_mma_input.store(mode, std::memory_order_relaxed);
    return PDF_OK;
}
extern "C" int pdf_get_mma_input(void) { return g_mma_input.load(std::memory_order_relaxed); }
int pdf_mma_input_mode() { return g_mma_input.load(std::memory_order_relaxed); }

// Ticket arrays of the in-launch reductions (pdfops_common.h: pdf_tail_sum), one per stream, caller-owned and zero-initialised
// (PDF_TICKET_WORDS 32-bit words): the only state this library keeps, an association table -- nothing is allocated or freed here.
#include <mutex>
namespace {
struct Bound { hipStream_t s; unsigned *w; };
Bound g_bound[64];
int g_nbound = 0;
std::mutex g_bound_mu;
}  // namespace

extern "C" int pdf_tickets_words(void) { return PDF_TICKET_WORDS; }

extern "C" int pdf_tickets_bind(void *stream, void *words) {
    std::lock_guard<std::mutex> lock(g_bound_mu);
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int i = 0; i < g_nbound; ++i)
        if (g_bound[i].s == s) { g_bound[i].w = static_cast<unsigned *>(words); return PDF_OK; }
    if (g_nbound == 64) return PDF_ERR_UNSUPPORTED;
    g_bound[g_nbound++] = Bound{s, static_cast<unsigned *>(words)};
    return PDF_OK;
}

unsigned *pdf_tickets_for(hipStream_t s, long n) {
    // OPT-IN (PDFOPS_TAIL=1).  Measured on MI355X, round 3, the BatchNorm statistics / backward-sum tails of pw::k_bn_stats, pw::k_bn_bwd_reduce
    // and rl2::k_fwd (~150 reducer launches folded per step): 19.2 ms per step with the tails against 18.1 ms with the separate 5-us reducer
    // launches, same box, back to back -- the agent-scope release every workgroup needs (buffer_wbl2: an L2-wide write-back while the
    // kernel is still streaming its own output) costs more than the launch boundary it saves.  Kept for the record and for A/B runs.
    // PDFOPS_TAIL_MAXN=<rows>: tails only for launches over at most that many rows (the small levels, where a launch writes little).
    static const bool on = [] { const char *v = getenv("PDFOPS_TAIL"); return v && v[0] == '1'; }();
    static const long maxn = [] { const char *v = getenv("PDFOPS_TAIL_MAXN"); return v ? atol(v) : 0L; }();
    if (!on && n > maxn) return nullptr;
    std::lock_guard<std::mutex> lock(g_bound_mu);
    for (int i = 0; i < g_nbound; ++i)
        if (g_bound[i].s == s) return g_bound[i].w;
    return nullptr;
}
