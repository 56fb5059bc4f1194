// Error reporting + ABI version for libgeoformer_hip.so
#include <stdarg.h>

#include "gf_common.h"

static thread_local char g_err[512] = "";

void gf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int gf_abi_version(void) { return 1; }
extern "C" const char* gf_last_error(void) { return g_err; }
