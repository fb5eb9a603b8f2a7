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
extern "C" int lav_abi_version(void) { return 5; }

// Split-K workspaces, LDS-size attributes, window tables and the optimizer's partial-sum buffer are process-wide (keyed by
// stream at most): ONE device per process, the deployment model of this library (one rank per GPU).  Every launch checks it,
// so a process that switches devices gets an error instead of a foreign workspace.
int lav_check_launch(const char* what) {
    static int first_device = -1;
    int dev = -1;
    if (hipGetDevice(&dev) == hipSuccess) {
        if (first_device < 0) first_device = dev;
        else if (dev != first_device) {
            lav_set_error("%s: called on device %d after device %d: liblavender_hip keeps per-process state and supports one device per "
                          "process (run one rank per GPU)", what, dev, first_device);
            return LAV_E_UNSUPPORTED;
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        lav_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return LAV_E_LAUNCH;
    }
    return LAV_OK;
}
