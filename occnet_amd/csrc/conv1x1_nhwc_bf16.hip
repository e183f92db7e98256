// Backbone 1x1 convolutions (NHWC bf16) as a GEMM on the gfx950 bf16 matrix cores with the whole tail
// fused: out = relu?( x . W^T + bias (+ residual) ), bf16 in / f32 accumulate / bf16 out.
//
// NOT part of the hand-written hot path (SURVEY.md §2 row 8 keeps the image backbone on stock MIOpen): the
// 3x3 / 7x7 convolutions stay MIOpen's.  A 1x1 convolution on an NHWC activation is exactly the row-major
// GEMM the encoder's Linear kernel already implements, so the reference's ResNet-50 bottleneck
// (mmdet ResNet: conv1 1x1 -> BN -> ReLU, conv3 1x1 -> BN -> += identity -> ReLU, downsample 1x1 stride 2
// -> BN) and the FPN laterals reuse its structure: eval BatchNorm folded into the weights, bias + residual +
// ReLU in the epilogue, so the activation makes ONE trip through HBM per layer instead of three
// (MIOpen kernel + its zero-fill / cast helpers + the elementwise tail).
//
// Decomposition: block = 4 waves x 32*RT rows (pixels; 64 or 128) x 128*NT output channels; K chunk = 32 input channels;
// the rows x 32 activation chunk is staged once per block (double buffered in LDS, one barrier per chunk) and
// every wave stages its own 32*NT x 32 weight slice — all plain 16-byte copies, 80-byte row stride =
// conflict-free ds_read_b128; next chunk's loads in flight during the MFMAs (v_mfma_f32_32x32x16_bf16, 2
// k-steps per chunk).
// Strided 1x1 convolutions (the downsample branch) only change which input pixel a row reads.
#include "common.h"

namespace occ {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kCLD = 80;   // LDS row stride in bytes: 32 bf16 (64 B) + 16 B pad

__device__ __forceinline__ float c1_bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short c1_f32_to_bf16(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

template <int NT, int RT>
__global__ __launch_bounds__(256) void conv1x1_nhwc_bf16_kernel(
    const uint4* __restrict__ x, const uint4* __restrict__ w, const float* __restrict__ bias,
    const unsigned short* __restrict__ residual, unsigned short* __restrict__ out, long M, int N, int K,
    int Hin, int Win, int Hout, int Wout, int stride, int relu) {
  constexpr int BM = 32 * RT, BN = 128 * NT, WR = 32 * NT, OLD = BN + 4, AP = RT / 2;   // AP: A pieces per thread
  constexpr int A_BYTES = BM * kCLD, W_BYTES = WR * kCLD;
  // LDS: the activation chunk is staged ONCE per block (double buffered, one barrier per chunk) and read by
  // all four waves; every wave keeps a private region for its own weight slice
  constexpr int STAGE_BYTES = 2 * A_BYTES + 4 * W_BYTES, OUT_BYTES = 32 * OLD * 4;
  __shared__ __attribute__((aligned(16))) char lds[STAGE_BYTES > OUT_BYTES ? STAGE_BYTES : OUT_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vi = lane & 31, kb = lane >> 5;
  char* sW = lds + 2 * A_BYTES + wave * W_BYTES;
  const long m0 = (long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int nw0 = n0 + wave * WR;
  const int KQ = K / 8;            // uint4 (8 bf16) per row

  f32x16 acc[RT][NT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rt][t][r] = 0.f;

  // A: thread -> (row = tid/4, 16-byte piece = tid%4) of the block's 64 rows; W: lane -> (row = lane/4 +
  // 16*it, piece = lane%4) of the wave's slice.  Unconditional clamped loads.
  const int arow = tid >> 2, sp = tid & 3, srow = lane >> 2;
  long aofs[AP];
#pragma unroll
  for (int ap = 0; ap < AP; ++ap) {
    long m = m0 + arow + 64 * ap;
    if (m >= M) m = M - 1;
    long pix = m;
    if (stride != 1) {           // output pixel (n, yo, xo) reads input pixel (n, yo*stride, xo*stride)
      const long hw = (long)Hout * Wout;
      const long n = m / hw, rem = m % hw;
      const int yo = (int)(rem / Wout), xo = (int)(rem % Wout);
      pix = (n * Hin + (long)yo * stride) * Win + (long)xo * stride;
    }
    aofs[ap] = pix * KQ + sp;
  }
  long wofs[2 * NT];
#pragma unroll
  for (int it = 0; it < 2 * NT; ++it) {
    const int n = nw0 + srow + 16 * it;
    wofs[it] = (long)(n < N ? n : N - 1) * KQ + sp;
  }
  uint4 va0, va1, vw0, vw1, vw2, vw3;
#define OCC_C1_ISSUE(K0)                                                                          \
  {                                                                                               \
    const long kq = (K0) / 8;                                                                     \
    va0 = x[aofs[0] + kq];                                                                        \
    if (AP == 2) va1 = x[aofs[AP - 1] + kq];                                                      \
    vw0 = w[wofs[0] + kq]; vw1 = w[wofs[1] + kq];                                                 \
    if (NT == 2) { vw2 = w[wofs[2 * NT - 2] + kq]; vw3 = w[wofs[2 * NT - 1] + kq]; }              \
  }

  OCC_C1_ISSUE(0)
  int buf = 0;
  for (int k0 = 0; k0 < K; k0 += 32, buf ^= 1) {
    char* sA = lds + buf * A_BYTES;
    *reinterpret_cast<uint4*>(sA + arow * kCLD + sp * 16) = va0;
    if (AP == 2) *reinterpret_cast<uint4*>(sA + (arow + 64) * kCLD + sp * 16) = va1;
    *reinterpret_cast<uint4*>(sW + (srow) * kCLD + sp * 16) = vw0;
    *reinterpret_cast<uint4*>(sW + (srow + 16) * kCLD + sp * 16) = vw1;
    if (NT == 2) {
      *reinterpret_cast<uint4*>(sW + (srow + 32) * kCLD + sp * 16) = vw2;
      *reinterpret_cast<uint4*>(sW + (srow + 48) * kCLD + sp * 16) = vw3;
    }
    __syncthreads();   // chunk visible to every wave; the other A buffer is free for the next iteration
    OCC_C1_ISSUE(k0 + 32 < K ? k0 + 32 : k0)
    bf16x8 af[RT][2], wf[NT][2];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        af[rt][ks] = *reinterpret_cast<const bf16x8*>(sA + (rt * 32 + vi) * kCLD + ks * 32 + kb * 16);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        wf[t][ks] = *reinterpret_cast<const bf16x8*>(sW + (t * 32 + vi) * kCLD + ks * 32 + kb * 16);
    wave_lds_sync();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc[rt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt][ks], wf[t][ks], acc[rt][t], 0, 0, 0);
  }
#undef OCC_C1_ISSUE

  // ---- epilogue, 32 rows at a time through an LDS transpose: bias, residual, ReLU, bf16 store --------
  const int c = lane * 4;
  const bool col_live = c < BN && n0 + c < N;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col_live) bv = *reinterpret_cast<const float4*>(bias + n0 + c);
  float* sO = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        sO[((r & 3) + 8 * (r >> 2) + 4 * kb) * OLD + (wave * NT + t) * 32 + vi] = acc[rt][t][r];
    __syncthreads();
    // all 8 residual rows of this wave are requested before any is consumed (clamped, unconditional):
    // a load inside the row loop would serialise 8 dependent memory round trips per pass
    uint2 rv[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      long m = m0 + rt * 32 + wave * 8 + rr;
      if (m >= M) m = M - 1;
      rv[rr] = make_uint2(0u, 0u);
      if (residual != nullptr && col_live) rv[rr] = *reinterpret_cast<const uint2*>(residual + m * N + n0 + c);
    }
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int row = wave * 8 + rr;
      const long m = m0 + rt * 32 + row;
      if (m < M && col_live) {
        float4 v = *reinterpret_cast<const float4*>(sO + row * OLD + c);
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        v.x += c1_bf16_to_f32((unsigned short)(rv[rr].x & 0xffffu));
        v.y += c1_bf16_to_f32((unsigned short)(rv[rr].x >> 16));
        v.z += c1_bf16_to_f32((unsigned short)(rv[rr].y & 0xffffu));
        v.w += c1_bf16_to_f32((unsigned short)(rv[rr].y >> 16));
        if (relu) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        const uint2 o = make_uint2((unsigned)c1_f32_to_bf16(v.x) | ((unsigned)c1_f32_to_bf16(v.y) << 16),
                                   (unsigned)c1_f32_to_bf16(v.z) | ((unsigned)c1_f32_to_bf16(v.w) << 16));
        *reinterpret_cast<uint2*>(out + m * N + n0 + c) = o;
      }
    }
  }
}

}  // namespace occ

extern "C" int occ_conv1x1_nhwc_bf16(const void* x, const void* weight, const float* bias,
                                     const void* residual, void* out, int batch, int Hin, int Win,
                                     int Cin, int Cout, int stride, int relu, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(x && weight && bias && out, "conv1x1_nhwc_bf16: null pointer argument");
  OCC_CHECK_ARG(batch > 0 && Hin > 0 && Win > 0 && Cin > 0 && Cout > 0 && stride > 0,
                "conv1x1_nhwc_bf16: bad dimension");
  if (Cin % 32 || Cout % 8) {
    set_error("conv1x1_nhwc_bf16: no kernel for Cin=%d Cout=%d (need Cin %% 32 == 0, Cout %% 8 == 0)", Cin,
              Cout);
    return OCC_E_UNSUPPORTED;
  }
  const int Hout = (Hin - 1) / stride + 1, Wout = (Win - 1) / stride + 1;
  const long M = (long)batch * Hout * Wout;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define OCC_C1_LAUNCH(NTT, RTT, BNN)                                                                \
  hipLaunchKernelGGL((conv1x1_nhwc_bf16_kernel<NTT, RTT>),                                          \
                     dim3((unsigned)((M + 32 * RTT - 1) / (32 * RTT)), (unsigned)((Cout + BNN - 1) / BNN)), \
                     dim3(256), 0, st, reinterpret_cast<const uint4*>(x),                           \
                     reinterpret_cast<const uint4*>(weight), bias,                                  \
                     reinterpret_cast<const unsigned short*>(residual),                             \
                     reinterpret_cast<unsigned short*>(out), M, Cout, Cin, Hin, Win, Hout, Wout, stride, relu)
  // 64-row blocks.  128-row blocks (RT = 4: weight slice staged once per 128 pixels) were measured on the
  // ResNet-50 shapes and are not faster (87.4 vs 88.9 samples/s end to end: 2 instead of 4 blocks per CU).
  const bool big = false;
  if (Cout <= 128) {
    if (big) OCC_C1_LAUNCH(1, 4, 128); else OCC_C1_LAUNCH(1, 2, 128);
  } else {
    if (big) OCC_C1_LAUNCH(2, 4, 256); else OCC_C1_LAUNCH(2, 2, 256);
  }
#undef OCC_C1_LAUNCH
  OCC_CHECK_LAUNCH("conv1x1_nhwc_bf16");
  return OCC_OK;
}
