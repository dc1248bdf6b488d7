// common.cpp — pfa_last_error / pfa_version.
#include <cstdarg>
#include <cstdio>

#include "../../include/pufferlib_amd.h"

namespace pfa {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace pfa

extern "C" int pfa_version(void) { return 1; }
extern "C" const char *pfa_last_error(void) { return pfa::g_err; }
