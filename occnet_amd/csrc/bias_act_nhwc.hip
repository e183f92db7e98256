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

// Backward of the tail above for the TRAINING step's autocast backbone (conv -> + bias (+ residual) -> ReLU as one autograd
// node, plugin/backbone.py): g = relu ? (y > 0 ? grad_y : 0) : grad_y, and the bias gradient = column sums of g, in ONE pass
// over the activation — ATen runs threshold_backward, a bf16 column reduction and (at a residual join) an add, three passes.
// A thread keeps ONE 8-channel group for all its rows (grid-stride in whole rows), so the column sums are 8 fp32 registers;
// the block's threads of one group meet in LDS, every block writes one row of `partial` (blocks x C), and the second kernel
// adds the rows in a fixed order: deterministic, no atomics.  relu == 0: nothing is written back (g = grad_y), y is not read.
__global__ __launch_bounds__(256) void bias_act_bwd_nhwc_bf16_kernel(
    const uint4* __restrict__ gy, const uint4* __restrict__ y, uint4* __restrict__ g, float* __restrict__ partial,
    long rows, int CG, int relu) {
  __shared__ float red[256 * 8];
  const int t = threadIdx.x;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int cg = 0;
  if (CG <= 256) {
    const int rpi = 256 / CG;                       // rows per iteration of one block (CG divides 256 or the tail lanes idle)
    cg = t % CG;
    const int rsub = t / CG;
    if (rsub < rpi) {
      for (long r = (long)blockIdx.x * rpi + rsub; r < rows; r += (long)gridDim.x * rpi) {
        const long i = r * CG + cg;
        const uint4 a = gy[i];
        unsigned aa[4] = {a.x, a.y, a.z, a.w};
        if (relu) {
          const uint4 v = y[i];
          const unsigned vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            // y is a ReLU output: > 0 iff the bf16 pattern is non-zero and its sign bit clear
            const unsigned lo = vv[k] & 0xffffu, hi = vv[k] >> 16;
            const unsigned mlo = (lo != 0u && !(lo & 0x8000u)) ? 0xffffu : 0u;
            const unsigned mhi = (hi != 0u && !(hi & 0x8000u)) ? 0xffff0000u : 0u;
            aa[k] &= (mlo | mhi);
          }
          g[i] = make_uint4(aa[0], aa[1], aa[2], aa[3]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          acc[2 * k] += __uint_as_float(aa[k] << 16);
          acc[2 * k + 1] += __uint_as_float(aa[k] & 0xffff0000u);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[t * 8 + k] = acc[k];
    __syncthreads();
    if (t < CG) {
      float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int rs = 0; rs < rpi; ++rs)
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += red[(rs * CG + t) * 8 + k];
      float* dst = partial + ((long)blockIdx.x * CG + t) * 8;
      *reinterpret_cast<float4*>(dst) = make_float4(s[0], s[1], s[2], s[3]);
      *reinterpret_cast<float4*>(dst + 4) = make_float4(s[4], s[5], s[6], s[7]);
    }
  }
}

// 256 threads = 8 row groups x 32 channels: row group r adds the partial rows r, r + 8, .. (independent loads, four in
// flight), the 8 sums of a channel meet in LDS in a fixed order
__global__ __launch_bounds__(256) void bias_grad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                               int blocks, int C) {
  __shared__ float red[8][32];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < C) {
    int b = rg;
    for (; b + 24 < blocks; b += 32) {
      s0 += partial[(long)b * C + c];
      s1 += partial[(long)(b + 8) * C + c];
      s2 += partial[(long)(b + 16) * C + c];
      s3 += partial[(long)(b + 24) * C + c];
    }
    for (; b < blocks; b += 8) s0 += partial[(long)b * C + c];
  }
  red[rg][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && c < C) {
    float s = red[0][cl];
#pragma unroll
    for (int r = 1; r < 8; ++r) s += red[r][cl];
    out[c] = s;
  }
}

// Weight side of the training step's folded convolutions (plugin/backbone.py ConvBNActFunction): eval-mode BatchNorm folded
// into the convolution weight, and the fold's chain rule, ONE launch each — as ATen ops the fold is six launches per
// convolution in forward and seven in backward (x 45 convolutions per step, each a few microseconds of kernel and as many of
// launch).  One block per output channel o, n = Cin * kh * kw weights:
//   forward:  s = gamma[o] * rstd[o];  wf[o][.] = W[o][.] * s (fp32, OIHW, the own kernels' pack routines read it);
//             w16[o][kh][kw][i] = bf16(wf) (channels_last, MIOpen's operand in backward);  b[o] = beta[o] - gamma[o] * mean_rstd[o]
//   backward: dW[o][.] = gw[o][.] * s;  dgamma[o] = rstd[o] * sum(gw[o][.] * W[o][.]) - mean_rstd[o] * gb[o]
//             (gw: MIOpen's bf16 weight gradient, element strides given; the sum is a fixed-order tree: deterministic)
__global__ __launch_bounds__(256) void conv_bn_fold_fwd_kernel(
    const float* __restrict__ W, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ rstd, const float* __restrict__ mean_rstd, float* __restrict__ wf,
    unsigned short* __restrict__ w16, float* __restrict__ b, int I, int KK) {
  const int o = blockIdx.x;
  const float s = gamma[o] * rstd[o];
  const long n = (long)I * KK;
  for (long e = threadIdx.x; e < n; e += blockDim.x) {
    const float v = W[(long)o * n + e] * s;
    wf[(long)o * n + e] = v;
    const int i = (int)(e / KK), k = (int)(e - (long)i * KK);
    w16[(long)o * n + (long)k * I + i] = bf16_rne(v);
  }
  if (threadIdx.x == 0) b[o] = fmaf(-gamma[o], mean_rstd[o], beta[o]);
}

__global__ __launch_bounds__(256) void conv_bn_fold_bwd_kernel(
    const unsigned short* __restrict__ gw, long so, long si, long sk_h, long sk_w, const float* __restrict__ W,
    const float* __restrict__ gamma, const float* __restrict__ rstd, const float* __restrict__ mean_rstd,
    const float* __restrict__ gb, float* __restrict__ dW, float* __restrict__ dgamma, int I, int KH, int KW) {
  __shared__ float red[256];
  const int o = blockIdx.x;
  const float s = gamma[o] * rstd[o];
  const int KK = KH * KW;
  const long n = (long)I * KK;
  float dot = 0.f;
  for (long e = threadIdx.x; e < n; e += blockDim.x) {
    const int i = (int)(e / KK), k = (int)(e - (long)i * KK);
    const int kh = k / KW, kw = k - kh * KW;
    const float g = bf16_to_f32(gw[(long)o * so + (long)i * si + (long)kh * sk_h + (long)kw * sk_w]);
    dW[(long)o * n + e] = g * s;
    dot = fmaf(g, W[(long)o * n + e], dot);
  }
  red[threadIdx.x] = dot;
  __syncthreads();
  for (int d = 128; d >= 1; d >>= 1) {
    if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) dgamma[o] = rstd[o] * red[0] - mean_rstd[o] * gb[o];
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

extern "C" int64_t occ_bias_act_bwd_partial_floats(int64_t rows, int C) {
  if (rows <= 0 || C <= 0 || C % 8 || C > 2048) return 0;
  const int CG = C / 8, rpi = 256 / CG;
  long blocks = (rows + rpi - 1) / rpi;
  if (blocks > 512) blocks = 512;
  return blocks * (int64_t)C;
}

extern "C" int occ_bias_act_bwd_nhwc_bf16(const void* grad_y, const void* y, void* g, float* partial, float* bias_grad,
                                          int64_t rows, int C, int relu, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(grad_y && partial && bias_grad && (!relu || (y && g)), "bias_act_bwd_nhwc_bf16: null pointer argument");
  OCC_CHECK_ARG(rows > 0 && C > 0, "bias_act_bwd_nhwc_bf16: bad dimension");
  if (C % 8 || C > 2048) {
    set_error("bias_act_bwd_nhwc_bf16: channel count %d must be a multiple of 8 and at most 2048", C);
    return OCC_E_UNSUPPORTED;
  }
  const int CG = C / 8, rpi = 256 / CG;
  long blocks = (rows + rpi - 1) / rpi;
  if (blocks > 512) blocks = 512;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(bias_act_bwd_nhwc_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, st,
                     reinterpret_cast<const uint4*>(grad_y), reinterpret_cast<const uint4*>(y), reinterpret_cast<uint4*>(g),
                     partial, (long)rows, CG, relu);
  OCC_CHECK_LAUNCH("bias_act_bwd_nhwc_bf16");
  hipLaunchKernelGGL(bias_grad_reduce_kernel, dim3((unsigned)((C + 31) / 32)), dim3(256), 0, st, partial, bias_grad,
                     (int)blocks, C);
  OCC_CHECK_LAUNCH("bias_act_bwd_nhwc_bf16 (reduce)");
  return OCC_OK;
}

extern "C" int occ_conv_bn_fold_fwd_f32(const float* weight, const float* gamma, const float* beta, const float* rstd,
                                        const float* mean_rstd, float* w_folded, void* w_folded_bf16_nhwc, float* bias,
                                        int Cout, int Cin, int KH, int KW, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(weight && gamma && beta && rstd && mean_rstd && w_folded && w_folded_bf16_nhwc && bias,
                "conv_bn_fold_fwd: null pointer argument");
  OCC_CHECK_ARG(Cout > 0 && Cin > 0 && KH > 0 && KW > 0, "conv_bn_fold_fwd: bad dimension");
  hipLaunchKernelGGL(conv_bn_fold_fwd_kernel, dim3((unsigned)Cout), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), weight,
                     gamma, beta, rstd, mean_rstd, w_folded, reinterpret_cast<unsigned short*>(w_folded_bf16_nhwc), bias, Cin,
                     KH * KW);
  OCC_CHECK_LAUNCH("conv_bn_fold_fwd");
  return OCC_OK;
}

extern "C" int occ_conv_bn_fold_bwd_f32(const void* grad_w_bf16, int64_t stride_o, int64_t stride_i, int64_t stride_h,
                                        int64_t stride_w, const float* weight, const float* gamma, const float* rstd,
                                        const float* mean_rstd, const float* grad_bias, float* grad_weight, float* grad_gamma,
                                        int Cout, int Cin, int KH, int KW, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(grad_w_bf16 && weight && gamma && rstd && mean_rstd && grad_bias && grad_weight && grad_gamma,
                "conv_bn_fold_bwd: null pointer argument");
  OCC_CHECK_ARG(Cout > 0 && Cin > 0 && KH > 0 && KW > 0, "conv_bn_fold_bwd: bad dimension");
  hipLaunchKernelGGL(conv_bn_fold_bwd_kernel, dim3((unsigned)Cout), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const unsigned short*>(grad_w_bf16), (long)stride_o, (long)stride_i, (long)stride_h,
                     (long)stride_w, weight, gamma, rstd, mean_rstd, grad_bias, grad_weight, grad_gamma, Cin, KH, KW);
  OCC_CHECK_LAUNCH("conv_bn_fold_bwd");
  return OCC_OK;
}
