// Thread-local error message + ABI version for libstep_hip.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/step_hip.h"

static thread_local char g_err[512] = "";

void step_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* step_last_error(void) { return g_err; }
extern "C" int step_abi_version(void) { return 10; }
