// Error plumbing + build identification for libdin_hip.so
#include "din_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void din_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {
int din_abi_version(void) { return DIN_ABI_VERSION; }
const char* din_last_error_string(void) { return g_err; }
const char* din_build_arch(void) { return "gfx950"; }
}
