// Error bookkeeping + version for libekaid_hip.so.
#include "common.h"
#include "../../include/ekaid_hip.h"
#include <cstdarg>
#include <cstdio>

namespace ekaid {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return EKAID_ERR_LAUNCH;
    }
    return EKAID_OK;
}
}  // namespace ekaid

extern "C" int ekaid_version(void) { return EKAID_ABI_VERSION; }
extern "C" const char* ekaid_last_error(void) { return ekaid::g_err; }
