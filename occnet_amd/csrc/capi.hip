// C-ABI plumbing shared by every entry point: ABI version + thread-local error message.
#include <stdarg.h>
#include "common.h"

namespace occ {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace occ

extern "C" int occ_abi_version(void) { return 2; }
extern "C" const char* occ_last_error(void) { return occ::g_err; }
