// Error plumbing of the C ABI: thread-local last-error text, launch checking.  No exceptions cross the ABI.
#include "dbw_common.h"
#include "../../include/dbw_hip.h"

#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void dbw_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int dbw_check_launch(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        dbw_set_error("%s: %s", what, hipGetErrorString(e));
        return DBW_ERR_LAUNCH;
    }
    return DBW_OK;
}

extern "C" const char *dbw_last_error(void) { return g_err; }
extern "C" int dbw_bin_subcursors(void) { return DBW_BIN_SUBCURSORS; }
extern "C" int dbw_abi_version(void) { return DBW_ABI_VERSION; }      // (history: include/dbw_hip.h)
