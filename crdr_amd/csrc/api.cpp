// Library-level entry points: version, architecture, thread-local error string.
#include <stdarg.h>
#include <stdio.h>

#include "crdr_hip.h"

namespace crdr {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace crdr

extern "C" const char* crdr_last_error(void) { return crdr::g_err; }
extern "C" int crdr_version(void) { return 100; }
extern "C" const char* crdr_arch(void) { return "gfx950"; }
