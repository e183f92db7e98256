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

// Stem tail: out = maxpool3x3/s2/p1( relu(y + bias) ) in ONE pass over the raw convolution output
// (= relu(max(y) + bias): bias and ReLU are monotonic), instead of the in-place bias/ReLU pass plus PyTorch's
// pooling kernel.  One lane = one output pixel x 8 channels: nine 16-byte loads (padding taps skipped, as
// max_pool2d pads with -inf), fp32 max, one 16-byte store.  HBM-bound on reading y once.
__global__ __launch_bounds__(256) void bias_relu_maxpool_nhwc_bf16_kernel(
    const uint4* __restrict__ y, const float* __restrict__ bias, uint4* __restrict__ out, long n_vec, int H,
    int W, int C8, int Ho, int Wo) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_vec) return;
  const int cg = (int)(i % C8);
  long r = i / C8;
  const int ox = (int)(r % Wo); r /= Wo;
  const int oy = (int)(r % Ho);
  const long n = r / Ho;
  float m[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) m[k] = -INFINITY;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
        const uint4 v = y[((n * H + iy) * W + ix) * C8 + cg];
        const unsigned vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          m[2 * k] = fmaxf(m[2 * k], __uint_as_float(vv[k] << 16));
          m[2 * k + 1] = fmaxf(m[2 * k + 1], __uint_as_float(vv[k] & 0xffff0000u));
        }
      }
    }
  const float4 b0 = *reinterpret_cast<const float4*>(bias + cg * 8);
  const float4 b1 = *reinterpret_cast<const float4*>(bias + cg * 8 + 4);
  out[i] = make_uint4(pack_bf16x2_rne(fmaxf(m[0] + b0.x, 0.f), fmaxf(m[1] + b0.y, 0.f)),
                      pack_bf16x2_rne(fmaxf(m[2] + b0.z, 0.f), fmaxf(m[3] + b0.w, 0.f)),
                      pack_bf16x2_rne(fmaxf(m[4] + b1.x, 0.f), fmaxf(m[5] + b1.y, 0.f)),
                      pack_bf16x2_rne(fmaxf(m[6] + b1.z, 0.f), fmaxf(m[7] + b1.w, 0.f)));
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

extern "C" int occ_bias_relu_maxpool_nhwc_bf16(const void* y, const float* bias, void* out, int batch, int H,
                                               int W, int C, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(y && bias && out, "bias_relu_maxpool_nhwc_bf16: null pointer argument");
  OCC_CHECK_ARG(batch > 0 && H > 0 && W > 0 && C > 0, "bias_relu_maxpool_nhwc_bf16: bad dimension");
  if (C % 8) {
    set_error("bias_relu_maxpool_nhwc_bf16: channel count %d is not a multiple of 8", C);
    return OCC_E_UNSUPPORTED;
  }
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;      // floor((H + 2 - 3) / 2) + 1
  const long n_vec = (long)batch * Ho * Wo * (C / 8);
  hipLaunchKernelGGL(bias_relu_maxpool_nhwc_bf16_kernel, dim3((unsigned)((n_vec + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const uint4*>(y), bias,
                     reinterpret_cast<uint4*>(out), n_vec, H, W, C / 8, Ho, Wo);
  OCC_CHECK_LAUNCH("bias_relu_maxpool_nhwc_bf16");
  return OCC_OK;
}
