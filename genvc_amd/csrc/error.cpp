#include <stdarg.h>
#include <stdio.h>

#include "../../include/genvc_hip.h"

namespace gvc {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace gvc

extern "C" int gvc_version(void) { return 100; }
extern "C" const char* gvc_last_error(void) { return gvc::g_err; }
