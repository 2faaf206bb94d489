// Version / error reporting of the gcnx C ABI.
#include <stdarg.h>
#include <string.h>

#include "gcnx_common.h"

namespace gcnx {

static thread_local char g_error[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}

const char *get_error() { return g_error; }

}  // namespace gcnx

extern "C" int gcnx_abi_version(void) { return GCNX_ABI_VERSION; }

extern "C" const char *gcnx_last_error(void) { return gcnx::get_error(); }
