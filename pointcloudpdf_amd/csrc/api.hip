// Version / build probes of libpdfops.so (callable without a GPU).
#include "pdfops_common.h"

#define PDF_ABI_VERSION 1

extern "C" int pdf_abi_version(void) { return PDF_ABI_VERSION; }

extern "C" const char *pdf_build_info(void) {
    return "libpdfops abi=1 target=gfx950 wave64 hipcc " __VERSION__;
}
