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
// Decomposition: block = 4 waves x 32*RT rows (pixels; 64 or 128) x 128*NT output channels; K chunk = 32 or 64 input channels;
// the rows x 32 activation chunk is staged once per block (double buffered in LDS, one barrier per chunk) and
// every wave stages its own 32*NT x 32 weight slice — all plain 16-byte copies, 80-byte row stride =
// conflict-free ds_read_b128; next chunk's loads in flight during the MFMAs (v_mfma_f32_32x32x16_bf16, 2
// k-steps per chunk).
// Strided 1x1 convolutions (the downsample branch) only change which input pixel a row reads.  The weight is
// given chunk-major [Cin/32][Cout][32], so the slice a wave stages per chunk is contiguous.
#include "common.h"

namespace occ {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// LDS row stride in bytes for a K chunk of KC bf16: KC*2 + 16 B pad (80 / 144: conflict-free ds_read_b128)
template <int KC> struct C1Geom { static constexpr int LD = KC * 2 + 16, PIECES = KC / 8; };

__device__ __forceinline__ float c1_bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

template <int NT, int RT, int KC>
__global__ __launch_bounds__(256) void conv1x1_nhwc_bf16_kernel(
    const uint4* __restrict__ x, const uint4* __restrict__ w, const float* __restrict__ bias,
    const unsigned short* __restrict__ residual, unsigned short* __restrict__ out, long M, int N, int K,
    int Hin, int Win, int Hout, int Wout, int stride, int relu) {
  constexpr int kCLD = C1Geom<KC>::LD, PC = C1Geom<KC>::PIECES;    // PC 16-byte pieces per row per chunk
  constexpr int BM = 32 * RT, BN = 128 * NT, WR = 32 * NT, OLD = BN + 4;
  constexpr int AP = BM * PC / 256, WP = WR * PC / 64;   // A pieces per thread, W pieces per lane
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

  // A: thread item j -> (row = j / PC, piece = j % PC), j = tid + 256*ap; W: lane item j -> (row, piece) of the
  // wave's slice, j = lane + 64*wq.  Unconditional clamped loads, named registers (max 4 A + 8 W pieces).
  static_assert(AP <= 4 && WP <= 8, "staging register budget");
  long aofs[AP];
  int adst[AP];
#pragma unroll
  for (int ap = 0; ap < AP; ++ap) {
    const int j = tid + 256 * ap;
    const int row = j / PC, piece = j % PC;
    long m = m0 + row;
    if (m >= M) m = M - 1;
    long pix = m;
    if (stride != 1) {           // output pixel (n, yo, xo) reads input pixel (n, yo*stride, xo*stride)
      const long hw = (long)Hout * Wout;
      const long n = m / hw, rem = m % hw;
      const int yo = (int)(rem / Wout), xo = (int)(rem % Wout);
      pix = (n * Hin + (long)yo * stride) * Win + (long)xo * stride;
    }
    aofs[ap] = pix * KQ + piece;
    adst[ap] = row * kCLD + piece * 16;
  }
  long wofs[WP];
  int wdst[WP];
#pragma unroll
  for (int wq = 0; wq < WP; ++wq) {
    const int j = lane + 64 * wq;
    const int row = j / PC, piece = j % PC;
    const int n = nw0 + row;
    wofs[wq] = (long)(n < N ? n : N - 1) * PC + piece;     // chunk-major weights: + chunk * N * PC
    wdst[wq] = row * kCLD + piece * 16;
  }
  // Two register sets (S = 0 / 1) hold the next TWO K chunks: with few blocks per CU the kernel is bound by
  // the number of loads in flight, not by bandwidth, so the prefetch runs two chunks ahead.
  uint4 va0_0, va1_0, va2_0, va3_0, vw0_0, vw1_0, vw2_0, vw3_0, vw4_0, vw5_0, vw6_0, vw7_0;
  uint4 va0_1, va1_1, va2_1, va3_1, vw0_1, vw1_1, vw2_1, vw3_1, vw4_1, vw5_1, vw6_1, vw7_1;
#define OCC_C1_ISSUE(S, K0)                                                                       \
  {                                                                                               \
    const long kq = (K0) / 8;                                                                     \
    const long wq_ = (long)((K0) / KC) * N * PC;                                                  \
    va0_##S = x[aofs[0] + kq];                                                                    \
    if (AP > 1) va1_##S = x[aofs[AP > 1 ? 1 : 0] + kq];                                           \
    if (AP > 2) { va2_##S = x[aofs[AP > 2 ? 2 : 0] + kq]; va3_##S = x[aofs[AP > 3 ? 3 : 0] + kq]; } \
    vw0_##S = w[wofs[0] + wq_]; vw1_##S = w[wofs[1] + wq_];                                         \
    if (WP > 2) { vw2_##S = w[wofs[WP > 2 ? 2 : 0] + wq_]; vw3_##S = w[wofs[WP > 3 ? 3 : 0] + wq_]; } \
    if (WP > 4) {                                                                                 \
      vw4_##S = w[wofs[WP > 4 ? 4 : 0] + wq_]; vw5_##S = w[wofs[WP > 5 ? 5 : 0] + wq_];             \
      vw6_##S = w[wofs[WP > 6 ? 6 : 0] + wq_]; vw7_##S = w[wofs[WP > 7 ? 7 : 0] + wq_];             \
    }                                                                                             \
  }
  // one K chunk: registers of set S -> LDS (A buffer BUF), barrier, refill set S with chunk K_NEXT, MFMAs
#define OCC_C1_STEP(S, BUF, K_NEXT)                                                               \
  {                                                                                               \
    char* sA = lds + (BUF) * A_BYTES;                                                             \
    *reinterpret_cast<uint4*>(sA + adst[0]) = va0_##S;                                            \
    if (AP > 1) *reinterpret_cast<uint4*>(sA + adst[AP > 1 ? 1 : 0]) = va1_##S;                   \
    if (AP > 2) {                                                                                 \
      *reinterpret_cast<uint4*>(sA + adst[AP > 2 ? 2 : 0]) = va2_##S;                             \
      *reinterpret_cast<uint4*>(sA + adst[AP > 3 ? 3 : 0]) = va3_##S;                             \
    }                                                                                             \
    *reinterpret_cast<uint4*>(sW + wdst[0]) = vw0_##S;                                            \
    *reinterpret_cast<uint4*>(sW + wdst[1]) = vw1_##S;                                            \
    if (WP > 2) {                                                                                 \
      *reinterpret_cast<uint4*>(sW + wdst[WP > 2 ? 2 : 0]) = vw2_##S;                             \
      *reinterpret_cast<uint4*>(sW + wdst[WP > 3 ? 3 : 0]) = vw3_##S;                             \
    }                                                                                             \
    if (WP > 4) {                                                                                 \
      *reinterpret_cast<uint4*>(sW + wdst[WP > 4 ? 4 : 0]) = vw4_##S;                             \
      *reinterpret_cast<uint4*>(sW + wdst[WP > 5 ? 5 : 0]) = vw5_##S;                             \
      *reinterpret_cast<uint4*>(sW + wdst[WP > 6 ? 6 : 0]) = vw6_##S;                             \
      *reinterpret_cast<uint4*>(sW + wdst[WP > 7 ? 7 : 0]) = vw7_##S;                             \
    }                                                                                             \
    __syncthreads(); /* chunk visible to every wave; the other A buffer is free */                \
    OCC_C1_ISSUE(S, K_NEXT)                                                                       \
    _Pragma("unroll") for (int kh = 0; kh < KC / 32; ++kh) {                                      \
      bf16x8 af[RT][2], wf[NT][2];                                                                \
      _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                           \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                          \
          af[rt][ks] = *reinterpret_cast<const bf16x8*>(sA + (rt * 32 + vi) * kCLD + kh * 64 + ks * 32 + kb * 16); \
      _Pragma("unroll") for (int t = 0; t < NT; ++t)                                              \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                          \
          wf[t][ks] = *reinterpret_cast<const bf16x8*>(sW + (t * 32 + vi) * kCLD + kh * 64 + ks * 32 + kb * 16); \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                            \
        _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                         \
          _Pragma("unroll") for (int t = 0; t < NT; ++t)                                          \
            acc[rt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt][ks], wf[t][ks], acc[rt][t], 0, 0, 0); \
    }                                                                                             \
    wave_lds_sync();                                                                              \
  }

  // every block walks the K chunks in a rotated order (start depends on the block): blocks launched
  // together would otherwise request the same weight chunk and same-stride activation columns at the
  // same time (L2 channel hot-spotting); the K sum is order-independent up to f32 rounding
  const int NCHK = K / KC;
  const int rot = (int)((blockIdx.x * 5u + blockIdx.y * 3u) % (unsigned)NCHK);
#define OCC_C1_K(CI) ((((CI) + rot) % NCHK) * KC)    /* chunk index -> k offset (clamped index re-reads) */
  OCC_C1_ISSUE(0, OCC_C1_K(0))
  OCC_C1_ISSUE(1, OCC_C1_K(NCHK > 1 ? 1 : 0))
  for (int ci = 0; ci < NCHK; ci += 2) {
    OCC_C1_STEP(0, 0, OCC_C1_K(min(ci + 2, NCHK - 1)))
    if (ci + 1 < NCHK) OCC_C1_STEP(1, 1, OCC_C1_K(min(ci + 3, NCHK - 1)))
  }
#undef OCC_C1_K
#undef OCC_C1_STEP
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
        const uint2 o = make_uint2(pack_bf16x2_rne(v.x, v.y), pack_bf16x2_rne(v.z, v.w));
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
#define OCC_C1_LAUNCH_(NTT, RTT, BNN, KCC)                                                          \
  hipLaunchKernelGGL((conv1x1_nhwc_bf16_kernel<NTT, RTT, KCC>),                                     \
                     dim3((unsigned)((M + 32 * RTT - 1) / (32 * RTT)), (unsigned)((Cout + BNN - 1) / BNN)), \
                     dim3(256), 0, st, reinterpret_cast<const uint4*>(x),                           \
                     reinterpret_cast<const uint4*>(weight), bias,                                  \
                     reinterpret_cast<const unsigned short*>(residual),                             \
                     reinterpret_cast<unsigned short*>(out), M, Cout, Cin, Hin, Win, Hout, Wout, stride, relu)
  // K chunk of 32 input channels (64 was measured and is not faster); the prefetch runs two chunks ahead
#define OCC_C1_LAUNCH(NTT, RTT, BNN)                                                                \
  do {                                                                                              \
    OCC_C1_LAUNCH_(NTT, RTT, BNN, 32); /* KC = 64 measured: not faster */                          \
  } while (0)
  // 64-row blocks.  128-row blocks (RT = 4: weight slice staged once per 128 pixels) were measured on the
  // ResNet-50 shapes and are not faster (87.4 vs 88.9 samples/s end to end: 2 instead of 4 blocks per CU).
  const bool big = false;
  if (Cout <= 128) {
    if (big) OCC_C1_LAUNCH(1, 4, 128); else OCC_C1_LAUNCH(1, 2, 128);
  } else {
    if (big) OCC_C1_LAUNCH(2, 4, 256); else OCC_C1_LAUNCH(2, 2, 256);
  }
#undef OCC_C1_LAUNCH
#undef OCC_C1_LAUNCH_
  OCC_CHECK_LAUNCH("conv1x1_nhwc_bf16");
  return OCC_OK;
}
