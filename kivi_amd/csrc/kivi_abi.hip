// Error plumbing and version of the C ABI (include/kivi_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "kivi_common.h"

static thread_local char g_err[512] = "";

void kivi_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" int kivi_abi_version(void) { return KIVI_ABI_VERSION; }
extern "C" const char* kivi_last_error(void) { return g_err; }
