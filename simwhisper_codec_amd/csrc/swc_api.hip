// Library-level entry points: version, error text, device probe.
#include <stdarg.h>
#include "swc_common.h"

static thread_local char g_err[512] = "";

void swc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int swc_version(void) { return 110; }  // 1.10: + fp8 mode, swc_cast_fp8, swc_gather_rows

extern "C" const char* swc_last_error(void) { return g_err; }

extern "C" int swc_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        swc_set_error("swc_device_count: no HIP device (%s)", hipGetErrorString(e));
        return SWC_E_NODEV;
    }
    return n;
}
