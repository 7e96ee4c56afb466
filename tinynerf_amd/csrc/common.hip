// Error plumbing of the C ABI (include/tinynerf_hip.h).
#include "tn_common.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include <string.h>

namespace tn {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
// One line per slow path and process: a call that is VALID but lands on a general-shape fallback kernel (several times slower than the
// kernels the reference's configurations take) says so once on stderr (unless TN_QUIET is set) and through tn_last_warning_string().
static char g_warn[512] = "";
static std::mutex g_warn_mu;
static unsigned g_warned = 0;
void warn_once(int id, const char *fmt, ...) {
    std::lock_guard<std::mutex> lk(g_warn_mu);
    if (id >= 0 && id < 32) {
        if (g_warned & (1u << id)) return;
        g_warned |= 1u << id;
    }
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_warn, sizeof(g_warn), fmt, ap);
    va_end(ap);
    if (!getenv("TN_QUIET")) fprintf(stderr, "tinynerf_hip: %s\n", g_warn);
}
}  // namespace tn

extern "C" const char *tn_last_error_string(void) { return tn::g_err; }
extern "C" const char *tn_last_warning_string(void) {
    // a copy taken under the writer's mutex into a buffer of the calling thread: warn_once() may rewrite g_warn at any time
    static thread_local char copy[sizeof(tn::g_warn)];
    std::lock_guard<std::mutex> lk(tn::g_warn_mu);
    memcpy(copy, tn::g_warn, sizeof(copy));
    copy[sizeof(copy) - 1] = 0;
    return copy;
}
extern "C" int tn_abi_version(void) { return TN_ABI_VERSION; }
