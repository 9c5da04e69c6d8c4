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

// per calling thread: where the producers of split-f16 / fp8 activations report clipping (device memory, caller-owned)
static thread_local unsigned* g_sat = nullptr;
unsigned* swc_sat_counter() { return g_sat; }
extern "C" int swc_set_saturation_counter(uint32_t* dev_counters) {
    g_sat = dev_counters;
    return SWC_OK;
}

extern "C" int swc_version(void) { return 206; }  // 2.06: swc_layer_tail takes operand_dtype; 2.05: swc_convnext_block takes operand_dtype (SWC_BF16 | SWC_F16); 2.04: swc_proj_ln (+ _pack, _stream_bytes); 2.03: swc_convnext_pack folds gamma into the stream; 2.02: swc_mlp_block / swc_layer_tail + their packed operand streams (2.01: swc_convnext_block takes per-utterance frame limits; 2.00: saturation counter, fused ConvNeXt, packed layouts)

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
