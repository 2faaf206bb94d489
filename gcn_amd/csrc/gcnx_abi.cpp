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

// 0: large dense products (and the row-forming of the factored SpMM) run on the
// bf16 matrix pipe with exactly split operands, fp32 accuracy; 1: fp32-input
// MFMA chains only.  Thread-local, like the error text.
static thread_local int g_dense_arith = 0;
int dense_arith_mode() { return g_dense_arith; }

}  // namespace gcnx

extern "C" int gcnx_abi_version(void) { return GCNX_ABI_VERSION; }

extern "C" const char *gcnx_last_error(void) { return gcnx::get_error(); }

extern "C" int gcnx_dense_arith(int mode) {
  const int before = gcnx::g_dense_arith;
  if (mode == 0 || mode == 1) gcnx::g_dense_arith = mode;
  return before;
}
