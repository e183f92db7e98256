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

extern "C" int occ_abi_version(void) { return 3; }

// q[i] = occ::fdiv(a[i], d[i]) — the gathers' quotient (common.h) exposed so that the tests can measure it against IEEE division
namespace occ {
__global__ __launch_bounds__(256) void fdiv_selftest_kernel(const float* __restrict__ a, const float* __restrict__ d,
                                                            float* __restrict__ q, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) q[i] = fdiv(a[i], d[i]);
}
}  // namespace occ
extern "C" int occ_selftest_fdiv_f32(const float* a, const float* d, float* q, int64_t n, void* stream) {
  OCC_CHECK_ARG(a && d && q && n > 0, "selftest_fdiv: bad argument");
  hipLaunchKernelGGL(occ::fdiv_selftest_kernel, dim3(1024), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a, d, q, (long)n);
  OCC_CHECK_LAUNCH("selftest_fdiv");
  return OCC_OK;
}
extern "C" const char* occ_last_error(void) { return occ::g_err; }
