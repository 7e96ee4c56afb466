// Error plumbing of the C ABI (include/tinynerf_hip.h).
#include "tn_common.h"
#include <stdarg.h>

namespace tn {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace tn

extern "C" const char *tn_last_error_string(void) { return tn::g_err; }
extern "C" int tn_abi_version(void) { return TN_ABI_VERSION; }
