// Version / build probes of libpdfops.so (callable without a GPU).  The library keeps NO mutable state: no globals, no caches, no
// per-stream tables (rounds 2-3 had a process-wide product-input mode and a stream -> ticket-word table; both are gone -- the mode is a
// per-call argument of the pdf_rowlin_* family, the in-launch reducer tails were measured slower and removed, docs/NOTEBOOK.md).
#include "pdfops_common.h"

#define PDF_STR2(x) #x
#define PDF_STR(x) PDF_STR2(x)

extern "C" int pdf_abi_version(void) { return PDF_ABI_VERSION; }

extern "C" const char *pdf_build_info(void) {
    return "libpdfops abi=" PDF_STR(PDF_ABI_VERSION) " target=gfx950 wave64 hipcc " __VERSION__;
}
