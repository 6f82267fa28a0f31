// Thread-local error text + ABI version for libuc_hip.so.
#include "common.h"

static thread_local char g_uc_err[512] = {0};

void uc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_uc_err, sizeof(g_uc_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* uc_last_error(void) { return g_uc_err; }
extern "C" int uc_abi_version(void) { return UC_ABI_VERSION; }
