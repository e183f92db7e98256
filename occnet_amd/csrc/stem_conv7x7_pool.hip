// ResNet stem in one kernel (outside the hand-written hot path, like the other backbone kernels):
//   out = max_pool2d( relu( conv2d(x, W 7x7, stride 2, pad 3) + b ), 3, stride 2, pad 1 )
// x (N, 3, H, W) fp32 NCHW as the pipeline hands the normalised images over; out (N, H/4, W/4, 64) bf16 NHWC.
// Replaces the bf16/NHWC conversion pass, MIOpen's igemm + tensor helper and the bias/ReLU/max-pool pass: the
// 64-channel stride-2 map (285 MB at 6 x 928 x 1600) is never written to HBM.
//
// Block = 4 waves, one 8 x 8 tile of POOLED pixels: 17 x 17 convolution pixels, 39 x 39 input pixels.
//   stage   input tile -> LDS as [39 rows][40 px][4 ch] bf16 (8-byte pixels: channel 3 and pixel 39 are zero,
//           outside the image is zero = the convolution's padding)
//   conv    implicit GEMM on v_mfma_f32_32x32x16_bf16: M = 289 conv pixels (10 row tiles), N = 64, K = 7 rows x
//           (8 px x 4 ch) = 224: for one kernel row the 32 operands of a conv pixel are 64 CONTIGUOUS bytes of the
//           LDS tile (16-byte aligned because pixels are 8 bytes and the stride is 2), weights zero at the pad
//           positions (kx = 7, c = 3).  The wave's 14 weight fragments live in registers for the whole kernel
//           (occ_mfma_pack_b_frag_bf16 of the (64, 224) matrix).  Weights are the row operand, so a lane holds one
//           pixel x 4 consecutive channels: bias + ReLU + bf16 -> 8-byte LDS stores into the conv tile (zero for
//           conv pixels outside the map: pooling pads with -inf, and relu >= 0 makes 0 equivalent)
//   pool    one lane = one pooled pixel x 8 channels: nine 16-byte LDS reads, max, one 16-byte global store.
#include "common.h"

namespace occ {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kStP = 8;                        // pooled tile edge
constexpr int kStC = 2 * kStP + 1;             // conv tile edge (17)
constexpr int kStI = 2 * kStC + 5;             // input tile edge (39)
constexpr int kStIW = 40;                      // input tile row pitch in pixels (pixel 39 = zero pad)
constexpr int kStIn = kStI * kStIW * 8;        // 12 480 B
constexpr int kStNC = kStC * kStC;             // 289 conv pixels
constexpr int kStMS = 144;                     // conv pixel slot: 64 bf16 + 16 B pad
constexpr int kStConv = 320 * kStMS;           // 10 row tiles of 32 slots

// U8 (the input side of the path): x is the RAW camera image, uint8 HWC (N, Hs, Ws, 3), and the
// pipeline's NormalizeMultiviewImage ((x[to_rgb ? 2 - c : c] - mean[c]) * (1 / std[c]) in float32 — mmcv.imnormalize_
// subtracts the mean and MULTIPLIES by the reciprocal of std, it does not divide) and PadMultiViewImage
// (zeros to H x W = the next multiple of 32; reference P/datasets/pipelines/transform_3d.py:31-45,82-94) happen
// while the tile is staged: the 107 MB fp32 tensor of the six images is never built, 26 MB of uint8 cross PCIe
struct StemNorm { float mean[3], stdinv[3]; int Hs, Ws, to_rgb; };

template <bool U8>
__global__ __launch_bounds__(256, 2) void stem_conv7x7_pool_kernel(
    const void* __restrict__ x_, const uint4* __restrict__ wfrag, const float* __restrict__ bias,
    uint4* __restrict__ out, int H, int W, int Hc, int Wc, int Hp, int Wp, int tiles_x, int tiles_y,
    StemNorm nrm) {
  __shared__ __attribute__((aligned(16))) char lds[kStIn + kStConv];
  char* const sIn = lds;
  char* const sCv = lds + kStIn;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vi = lane & 31, kb = lane >> 5;
  int bid = blockIdx.x;
  const int tx_i = bid % tiles_x; bid /= tiles_x;
  const int ty_i = bid % tiles_y;
  const int img = bid / tiles_y;
  const int py0 = ty_i * kStP, px0 = tx_i * kStP;             // pooled tile origin
  const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;             // conv tile origin (may be -1)
  const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;             // input tile origin

  // this wave's 14 weight fragments (column tile wave & 1), requested first
  const int nt = wave & 1, mt0 = 5 * (wave >> 1);
  uint4 wf[14];
#pragma unroll
  for (int s = 0; s < 14; ++s) wf[s] = wfrag[((long)s * 2 + nt) * 64 + lane];
  float4 bv[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const float4*>(bias + nt * 32 + 8 * g + 4 * kb);

  // ---- stage the input tile: 4 x 39 x 40 bf16 items (channel-major for coalesced plane reads) ---------
  {
    const long plane = (long)H * W;
    const float* xi = reinterpret_cast<const float*>(x_) + (long)img * 3 * plane;
    const unsigned char* xu = reinterpret_cast<const unsigned char*>(x_) + (long)img * nrm.Hs * nrm.Ws * 3;
    constexpr int ITEMS = 4 * kStI * kStIW;                    // 6 240
    constexpr int NJ = (ITEMS + 255) / 256;                    // 25 per thread
    // all loads first (unconditional, clamped), then the conversions and LDS stores: a load inside the
    // conditional store would serialise 25 memory round trips
    float v[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int idx = min(tid + 256 * j, ITEMS - 1);
      const int c = idx / (kStI * kStIW), r = idx % (kStI * kStIW);
      const int ly = r / kStIW, lx = r % kStIW;
      const int cc = c < 3 ? c : 0, cy = min(max(iy0 + ly, 0), H - 1), cx = min(max(ix0 + lx, 0), W - 1);
      if (U8) {
        const int sy = min(cy, nrm.Hs - 1), sx = min(cx, nrm.Ws - 1), sc = nrm.to_rgb ? 2 - cc : cc;
        const float raw = (float)xu[((long)sy * nrm.Ws + sx) * 3 + sc];
        const float mu = cc == 0 ? nrm.mean[0] : (cc == 1 ? nrm.mean[1] : nrm.mean[2]);
        const float si = cc == 0 ? nrm.stdinv[0] : (cc == 1 ? nrm.stdinv[1] : nrm.stdinv[2]);
        v[j] = (cy < nrm.Hs && cx < nrm.Ws) ? (raw - mu) * si : 0.f;       // bottom / right pad: normalised zeros
      } else {
        v[j] = xi[cc * plane + (long)cy * W + cx];
      }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int idx = tid + 256 * j;
      if (idx < ITEMS) {
        const int c = idx / (kStI * kStIW), r = idx % (kStI * kStIW);
        const int ly = r / kStIW, lx = r % kStIW;
        const int iy = iy0 + ly, ix = ix0 + lx;
        const bool in = c < 3 && lx < kStI && iy >= 0 && iy < H && ix >= 0 && ix < W;
        *reinterpret_cast<unsigned short*>(sIn + (ly * kStIW + lx) * 8 + c * 2) = in ? bf16_rne(v[j]) : (unsigned short)0;
      }
    }
  }
  __syncthreads();

  // ---- convolution: 5 row tiles x 1 column tile per wave, K = 14 k-steps ------------------------------
  int abase[5];
  bool cvalid[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int q = (mt0 + j) * 32 + vi;                          // conv pixel of this lane in row tile j
    const int qc = q < kStNC ? q : kStNC - 1;
    const int cyl = qc / kStC, cxl = qc - cyl * kStC;
    abase[j] = ((2 * cyl) * kStIW + 2 * cxl) * 8 + kb * 16;
    const int cy = cy0 + cyl, cx = cx0 + cxl;
    cvalid[j] = q < kStNC && cy >= 0 && cy < Hc && cx >= 0 && cx < Wc;
  }
  f32x16 acc[5];
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
  for (int s = 0; s < 14; ++s) {
    const int koff = (s >> 1) * (kStIW * 8) + (s & 1) * 32;     // kernel row, first / second 4 pixels
    const bf16x8 w = __builtin_bit_cast(bf16x8, wf[s]);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(sIn + abase[j] + koff);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, a, acc[j], 0, 0, 0);     // D[channel][pixel]
    }
  }
  // bias + ReLU -> bf16 conv tile (zero where the conv pixel does not exist)
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int q = (mt0 + j) * 32 + vi;
    if (q < kStNC) {
      const float m = cvalid[j] ? 1.f : 0.f;
      char* dst = sCv + q * kStMS + (nt * 32 + 4 * kb) * 2;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float v0 = fmaxf(acc[j][4 * g + 0] + bv[g].x, 0.f) * m, v1 = fmaxf(acc[j][4 * g + 1] + bv[g].y, 0.f) * m;
        const float v2 = fmaxf(acc[j][4 * g + 2] + bv[g].z, 0.f) * m, v3 = fmaxf(acc[j][4 * g + 3] + bv[g].w, 0.f) * m;
        *reinterpret_cast<uint2*>(dst + 16 * g) = make_uint2(pack_bf16x2_rne(v0, v1), pack_bf16x2_rne(v2, v3));
      }
    }
  }
  __syncthreads();

  // ---- 3x3 / stride 2 max pooling out of the conv tile: 64 pooled pixels x 8 channel groups -----------
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int item = tid + 256 * j;
    const int cg = item & 7, p = item >> 3;
    const int pyl = p >> 3, pxl = p & 7;
    const int py = py0 + pyl, px = px0 + pxl;
    if (py < Hp && px < Wp) {
      float m[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) m[k] = 0.f;                   // every conv value is >= 0
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const uint4 v = *reinterpret_cast<const uint4*>(sCv + ((2 * pyl + dy) * kStC + 2 * pxl + dx) * kStMS + cg * 16);
          const unsigned vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            m[2 * k] = fmaxf(m[2 * k], __uint_as_float(vv[k] << 16));
            m[2 * k + 1] = fmaxf(m[2 * k + 1], __uint_as_float(vv[k] & 0xffff0000u));
          }
        }
      // (values are bf16 already: the max of bf16 numbers re-packs exactly)
      out[(((long)img * Hp + py) * Wp + px) * 8 + cg] =
          make_uint4(pack_bf16x2_rne(m[0], m[1]), pack_bf16x2_rne(m[2], m[3]), pack_bf16x2_rne(m[4], m[5]),
                     pack_bf16x2_rne(m[6], m[7]));
    }
  }
}

}  // namespace occ

extern "C" int occ_stem_conv7x7_pool_f32_bf16(const float* x, const void* weight_frag, const float* bias,
                                              void* out, int batch, int H, int W, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(x && weight_frag && bias && out, "stem_conv7x7_pool: null pointer argument");
  OCC_CHECK_ARG(batch > 0 && H > 0 && W > 0, "stem_conv7x7_pool: bad dimension");
  const int Hc = (H - 1) / 2 + 1, Wc = (W - 1) / 2 + 1;        // floor((H + 6 - 7) / 2) + 1
  const int Hp = (Hc - 1) / 2 + 1, Wp = (Wc - 1) / 2 + 1;      // floor((Hc + 2 - 3) / 2) + 1
  const int tiles_x = (Wp + kStP - 1) / kStP, tiles_y = (Hp + kStP - 1) / kStP;
  hipLaunchKernelGGL(stem_conv7x7_pool_kernel<false>, dim3((unsigned)((long)batch * tiles_x * tiles_y)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, reinterpret_cast<const uint4*>(weight_frag), bias,
                     reinterpret_cast<uint4*>(out), H, W, Hc, Wc, Hp, Wp, tiles_x, tiles_y, StemNorm{});
  OCC_CHECK_LAUNCH("stem_conv7x7_pool");
  return OCC_OK;
}

// Raw camera images in: x (batch, Hs, Ws, 3) uint8 HWC; the stem sees (x[to_rgb ? 2-c : c] - mean[c]) * (1 / std[c])
// zero-padded bottom/right to H x W (H >= Hs, W >= Ws) — NormalizeMultiviewImage + PadMultiViewImage fused into
// the tile staging.  mean / std: 3 host floats each, in the network's channel order.
extern "C" int occ_stem_conv7x7_pool_u8_bf16(const uint8_t* x, const void* weight_frag, const float* bias,
                                             void* out, int batch, int Hs, int Ws, int H, int W,
                                             const float* mean, const float* std, int to_rgb, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(x && weight_frag && bias && out && mean && std, "stem_conv7x7_pool_u8: null pointer argument");
  OCC_CHECK_ARG(batch > 0 && Hs > 0 && Ws > 0 && H >= Hs && W >= Ws, "stem_conv7x7_pool_u8: bad dimension");
  OCC_CHECK_ARG(std[0] != 0.f && std[1] != 0.f && std[2] != 0.f, "stem_conv7x7_pool_u8: zero std");
  const int Hc = (H - 1) / 2 + 1, Wc = (W - 1) / 2 + 1;
  const int Hp = (Hc - 1) / 2 + 1, Wp = (Wc - 1) / 2 + 1;
  const int tiles_x = (Wp + kStP - 1) / kStP, tiles_y = (Hp + kStP - 1) / kStP;
  StemNorm nrm;
  for (int c = 0; c < 3; ++c) { nrm.mean[c] = mean[c]; nrm.stdinv[c] = (float)(1.0 / (double)std[c]); }   // stdinv in f64 (mmcv), used in f32
  nrm.Hs = Hs; nrm.Ws = Ws; nrm.to_rgb = to_rgb ? 1 : 0;
  hipLaunchKernelGGL(stem_conv7x7_pool_kernel<true>, dim3((unsigned)((long)batch * tiles_x * tiles_y)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, reinterpret_cast<const uint4*>(weight_frag), bias,
                     reinterpret_cast<uint4*>(out), H, W, Hc, Wc, Hp, Wp, tiles_x, tiles_y, nrm);
  OCC_CHECK_LAUNCH("stem_conv7x7_pool_u8");
  return OCC_OK;
}
