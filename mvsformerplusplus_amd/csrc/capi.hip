// Error plumbing and ABI identification for libmvs_hip.so (see include/mvs_hip.h).
#include "mvs_common.h"

#include <stdarg.h>
#include <stdio.h>

namespace mvs {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return MVS_ERR_LAUNCH;
    }
    return MVS_OK;
}

// one reader per translation unit that stores fp16 activations (the counter is a per-unit device global, mvs_common.h)
unsigned int sat_read_conv(int reset);
unsigned int sat_read_gather(int reset);
unsigned int sat_read_gather_keep(int reset);
unsigned int sat_read_gather_agg(int reset);
unsigned int sat_read_gather_agg16(int reset);
unsigned int sat_read_warp(int reset);

}  // namespace mvs

extern "C" unsigned long long mvs_f16_saturation_count(int reset) {
    return (unsigned long long)mvs::sat_read_conv(reset) + mvs::sat_read_gather(reset) + mvs::sat_read_gather_keep(reset) +
           mvs::sat_read_gather_agg(reset) + mvs::sat_read_gather_agg16(reset) + mvs::sat_read_warp(reset);
}

extern "C" int mvs_abi_version(void) { return MVS_ABI_VERSION; }
extern "C" const char* mvs_last_error(void) { return mvs::g_err; }
