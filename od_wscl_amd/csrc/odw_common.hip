#include "odw_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void odw_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

ODW_EXPORT const char* odw_last_error(void) { return g_err; }
ODW_EXPORT int odw_version(void) { return 1; }
