// Version / build probes of libpdfops.so (callable without a GPU).
#include "pdfops_common.h"

#define PDF_ABI_VERSION 1

extern "C" int pdf_abi_version(void) { return PDF_ABI_VERSION; }

extern "C" const char *pdf_build_info(void) {
    return "libpdfops abi=1 target=gfx950 wave64 hipcc " __VERSION__;
}

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

unsigned *pdf_tickets_for(hipStream_t s) {
    // OPT-IN (PDFOPS_TAIL=1).  Measured on MI355X, round 3, the BatchNorm statistics / backward-sum tails of pw::k_bn_stats, pw::k_bn_bwd_reduce
    // and rl2::k_fwd (~150 reducer launches folded per step): 19.2 ms per step with the tails against 18.1 ms with the separate 5-us reducer
    // launches, same box, back to back -- the agent-scope release every workgroup needs (buffer_wbl2: an L2-wide write-back while the
    // kernel is still streaming its own output) costs more than the launch boundary it saves.  Kept for the record and for A/B runs.
    static const bool on = [] { const char *v = getenv("PDFOPS_TAIL"); return v && v[0] == '1'; }();
    if (!on) return nullptr;
    std::lock_guard<std::mutex> lock(g_bound_mu);
    for (int i = 0; i < g_nbound; ++i)
        if (g_bound[i].s == s) return g_bound[i].w;
    return nullptr;
}
