// Thread-local error text + version for libgdm_hip.so.
#include "gdm_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void gdm_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* gdm_last_error(void) { return g_err; }
extern "C" int gdm_version(void) { return 1; }
