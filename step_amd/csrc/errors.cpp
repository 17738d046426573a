// Thread-local error message + ABI version for libstep_hip.
#include <stdarg.h>
#include <stdio.h>
#include <mutex>
#include <hip/hip_runtime.h>
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

// The dynamic-LDS limit of a kernel is an attribute per (kernel, DEVICE): raised once for each pair, under a lock (first launches may come
// from two host threads at once: the autograd worker and the main thread), never on every launch (a launch inside a stream capture stays a
// plain kernel node).  `what` names the caller in the error message.  Returns 0 or STEP_ERR_HIP (-2).
int step_raise_lds_once(const void* kernel, int bytes, const char* what) {
    static std::mutex mu;
    static const void* seen[256];
    static int seen_dev[256], nseen = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    for (int i = 0; i < nseen; ++i)
        if (seen[i] == kernel && seen_dev[i] == dev) return 0;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
        step_set_error("%s: cannot raise the dynamic LDS limit to %d bytes", what, bytes);
        return -2;
    }
    if (nseen < 256) { seen[nseen] = kernel; seen_dev[nseen] = dev; ++nseen; }
    return 0;
}
