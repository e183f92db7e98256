// Row gather / gather-sum for the training path of SpatialCrossAttention (gfx950).
//
//   out[b][r][:] = sum_{k < K, index[r][k] >= 0} x[b][index[r][k]][:]
//
// With K = 1 this is the reference's per-camera rebatch (spatial_cross_attention.py:145-153: visible BEV queries
// copied into padded per-camera rows; index -1 = padding -> zeros); with index = the inverse map (for every BEV
// query the <= K padded rows that hold it) it is the scatter back into the BEV slots (:165-167) — and each is the
// other's gradient.  ATen does the pair with index_select + a float-atomic index_add (0.43 ms per call on MI355X:
// gfx950 retires float atomics at ~82 G/s); as a gather-sum both directions are plain coalesced copies, and the sum
// order is fixed (deterministic).
#include "common.h"

namespace occ {

__global__ __launch_bounds__(256) void rows_gather_sum_kernel(const float* __restrict__ x, long x_batch_stride,
                                                              const int64_t* __restrict__ index, int K,
                                                              float* __restrict__ out, long rows_out, int F4,
                                                              int rows_in) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long r = gid / F4;
  if (r >= rows_out) return;
  const int c = (int)(gid - r * F4);
  const int b = blockIdx.y;
  const float4* xb = reinterpret_cast<const float4*>(x + (long)b * x_batch_stride);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < K; ++k) {
    const long src = index[r * K + k];
    if (src >= 0 && src < rows_in) {
      const float4 v = xb[src * F4 + c];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  reinterpret_cast<float4*>(out + ((long)b * rows_out) * F4 * 4)[r * F4 + c] = acc;
}

}  // namespace occ

extern "C" int occ_rows_gather_sum_f32(const float* x, int64_t x_batch_stride, const int64_t* index, int K,
                                       float* out, int B, int64_t rows_out, int64_t rows_in, int F, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(x && index && out, "rows_gather_sum: null pointer argument");
  OCC_CHECK_ARG(B > 0 && B < 65536 && rows_out > 0 && rows_in > 0 && rows_in < (1L << 31) && K > 0 && F > 0,
                "rows_gather_sum: bad dimension (B=%d rows_out=%lld rows_in=%lld K=%d F=%d)", B,
                (long long)rows_out, (long long)rows_in, K, F);
  OCC_CHECK_ARG(F % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 15) == 0 && x_batch_stride % 4 == 0,
                "rows_gather_sum: rows must be 16-byte aligned multiples of 4 floats");
  const int F4 = F / 4;
  const long threads = rows_out * F4;
  OCC_CHECK_ARG((threads + 255) / 256 < (1L << 31), "rows_gather_sum: too many rows");
  hipLaunchKernelGGL(rows_gather_sum_kernel, dim3((unsigned)((threads + 255) / 256), (unsigned)B), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, (long)x_batch_stride, index, K, out, (long)rows_out,
                     F4, (int)rows_in);
  OCC_CHECK_LAUNCH("rows_gather_sum");
  return OCC_OK;
}
