// Thread-local error message for the C ABI (include/tbe_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/tbe_hip.h"

namespace tbe {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace tbe

extern "C" const char* tbe_last_error(void) { return tbe::g_err; }
extern "C" int32_t tbe_abi_version(void) { return 2; }
