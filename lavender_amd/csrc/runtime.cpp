// Error reporting + launch checking shared by all kernels (host side only).
#include "common.h"
#include <stdarg.h>
#include "../../include/lavender_hip.h"

static thread_local char g_err[512] = "";

extern "C" void lav_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* lav_last_error(void) { return g_err; }
extern "C" int lav_abi_version(void) { return 1; }

int lav_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        lav_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return LAV_E_LAUNCH;
    }
    return LAV_OK;
}
