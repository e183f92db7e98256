// Backbone tail: y = relu(y + bias[c] (+ residual)) in place on an NHWC bf16 activation.
//
// Not part of the hand-written hot path (the image backbone stays stock MIOpen convolutions); this one
// streaming kernel replaces the three elementwise launches PyTorch issues after every convolution of the
// reference's ResNet-50 (mmdet ResNet Bottleneck: BatchNorm -> ReLU, and BatchNorm -> += identity -> ReLU)
// once eval-mode BatchNorm has been folded into the convolution weights.  8 bf16 (16 B) per lane,
// channel index from the flat offset, fp32 arithmetic, round-to-nearest-even back to bf16.  HBM-bound:
// 2 (3 with residual) x 2 bytes per element.
#include "common.h"

namespace occ {

__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);                                                   // RNE
  return (unsigned short)(u >> 16);
}

__global__ __launch_bounds__(256) void bias_act_nhwc_bf16_kernel(
    uint4* __restrict__ x, const float* __restrict__ bias, const uint4* __restrict__ residual, long n_vec,
    int C, int relu) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec;
       i += (long)gridDim.x * blockDim.x) {
    const int c0 = (int)((i * 8) % C);
    uint4 v = x[i];
    uint4 r = make_uint4(0, 0, 0, 0);
    if (residual) r = residual[i];
    const float4 b0 = *reinterpret_cast<const float4*>(bias + c0);
    const float4 b1 = *reinterpret_cast<const float4*>(bias + c0 + 4);
    const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    unsigned vv[4] = {v.x, v.y, v.z, v.w}, rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float lo = bf16_to_f32((unsigned short)(vv[k] & 0xffffu)) + bb[2 * k];
      float hi = bf16_to_f32((unsigned short)(vv[k] >> 16)) + bb[2 * k + 1];
      if (residual) {
        lo += bf16_to_f32((unsigned short)(rr[k] & 0xffffu));
        hi += bf16_to_f32((unsigned short)(rr[k] >> 16));
      }
      if (relu) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
      vv[k] = pack_bf16x2_rne(lo, hi);
    }
    x[i] = make_uint4(vv[0], vv[1], vv[2], vv[3]);
  }
}

}  // namespace occ

extern "C" int occ_bias_act_nhwc_bf16(void* x, const float* bias, const void* residual, int64_t rows,
                                      int C, int relu, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(x && bias, "bias_act_nhwc_bf16: null pointer argument");
  OCC_CHECK_ARG(rows > 0 && C > 0, "bias_act_nhwc_bf16: bad dimension");
  if (C % 8) {
    set_error("bias_act_nhwc_bf16: channel count %d is not a multiple of 8", C);
    return OCC_E_UNSUPPORTED;
  }
  const long n_vec = rows * (long)C / 8;
  long blocks = (n_vec + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(bias_act_nhwc_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<uint4*>(x), bias,
                     reinterpret_cast<const uint4*>(residual), n_vec, C, relu);
  OCC_CHECK_LAUNCH("bias_act_nhwc_bf16");
  return OCC_OK;
}
