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

}  // namespace mvs

extern "C" int mvs_abi_version(void) { return MVS_ABI_VERSION; }
extern "C" const char* mvs_last_error(void) { return mvs::g_err; }
