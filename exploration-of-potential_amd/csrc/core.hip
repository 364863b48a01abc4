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
