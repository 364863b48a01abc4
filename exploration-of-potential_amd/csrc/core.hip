// ep24 - error plumbing of the C ABI.
#include "common.h"

static thread_local char g_err[512] = "";

void ep24_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* ep24_last_error(void) { return g_err; }
extern "C" int ep24_abi_version(void) { return EP24_ABI_VERSION; }
// 1 in the diagnostic build that carries the measured-and-lost kernel variants (make variants: the ring without a patch, the narrow
// ring tile, the 32 x 32 x 16 consumers, the weight gradient as a ring), 0 in the product library (round 5: they left it)
extern "C" int ep24_ab_variants(void) {
#ifdef EP24_AB_VARIANTS
    return 1;
#else
    return 0;
#endif
}
