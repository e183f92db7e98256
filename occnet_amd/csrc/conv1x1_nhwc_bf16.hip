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
// Decomposition: block = 4 waves x 64 rows (pixels) x 128*NT output channels; K chunk = 32 input channels.
// The 64 x 32 activation chunk is staged once per block (double buffered in LDS, one barrier per chunk, 80-byte
// row stride = conflict-free ds_read_b128).  The weight is given in MFMA B-fragment order
// (occ_mfma_pack_b_frag_bf16: [K/16][Cout/32][lane][8 bf16]) and goes global -> registers directly, three chunks
// deep; v_mfma_f32_32x32x16_bf16, 2 k-steps per chunk.
// Strided 1x1 convolutions (the downsample branch) only change which input pixel a row reads.
#include "common.h"

namespace occ {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));


__device__ __forceinline__ float c1_bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

template <int NT, int RT>
__global__ __launch_bounds__(256) void conv1x1_nhwc_bf16_kernel(
    const uint4* __restrict__ x, const uint4* __restrict__ w, const float* __restrict__ bias,
    const unsigned short* __restrict__ residual, unsigned short* __restrict__ out, long M, int N, int K,
    int Hin, int Win, int Hout, int Wout, int stride, int relu, int res_up) {
  constexpr int KC = 32, kCLD = KC * 2 + 16, PC = KC / 8;           // 80-byte LDS rows, 4 pieces of 16 B
  constexpr int BM = 32 * RT, BN = 128 * NT, WR = 32 * NT, OLD = BN + 4;
  constexpr int AP = BM * PC / 256;                                 // A pieces per thread per chunk
  constexpr int A_BYTES = BM * kCLD;
  // LDS carries ONLY the activation chunk (staged once per block, double buffered, one barrier per chunk, read
  // by all four waves).  The weights never touch it: they are pre-packed in MFMA B-fragment order, so a wave's
  // operand for (k-step, column tile) is one coalesced 1 KB global load straight into registers (each wave owns
  // its own 32*NT columns — there is nothing to share; going through LDS was only a layout transposer and made
  // the kernel LDS-bound: 13 KB of LDS traffic per wave and chunk for 8 MFMAs, now 5 KB).
  constexpr int STAGE_BYTES = 2 * A_BYTES, OUT_BYTES = 32 * OLD * 4;
  __shared__ __attribute__((aligned(16))) char lds[STAGE_BYTES > OUT_BYTES ? STAGE_BYTES : OUT_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vi = lane & 31, kb = lane >> 5;
  const long m0 = (long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int KQ = K / 8;            // uint4 (8 bf16) per row
  const int NT32 = N / 32;         // 32-column tiles in the packed weight

  f32x16 acc[RT][NT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rt][t][r] = 0.f;

  // A: thread item j -> (row = j / PC, piece = j % PC), j = tid + 256*ap.  Unconditional clamped loads.
  static_assert(AP >= 1 && AP <= 2 && NT <= 2, "staging register budget");
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
  // this wave's column tiles in the packed weight (clamped: a wave past the last tile recomputes it, its
  // columns are masked in the epilogue)
  const int nt0 = min((n0 + wave * WR) / 32, NT32 - 1), nt1 = min((n0 + wave * WR) / 32 + (NT - 1), NT32 - 1);
  const long wl0 = (long)nt0 * 64 + lane, wl1 = (long)nt1 * 64 + lane;
  // Registers: two activation sets (prefetch two chunks ahead) and three weight sets (the set of chunk c+2 is
  // requested while the MFMAs of chunk c still read theirs); loads are issued per chunk as one group
  // {A(c+2), W(c+2)} so the in-order vmcnt wait for chunk c never drains younger prefetches.
  uint4 va0_0, va1_0, va0_1, va1_1;
  uint4 wa0_0, wa1_0, wb0_0, wb1_0, wa0_1, wa1_1, wb0_1, wb1_1, wa0_2, wa1_2, wb0_2, wb1_2;   // w{ks a/b}{t}_{set}
#define OCC_C1_ISSUE_A(S, K0)                                                                     \
  {                                                                                               \
    const long kq = (K0) / 8;                                                                     \
    va0_##S = x[aofs[0] + kq];                                                                    \
    if (AP > 1) va1_##S = x[aofs[AP > 1 ? 1 : 0] + kq];                                           \
  }
#define OCC_C1_ISSUE_W(S, K0)                                                                     \
  {                                                                                               \
    const long k0 = (long)((K0) / 16) * NT32 * 64, k1 = k0 + (long)NT32 * 64;                     \
    wa0_##S = w[k0 + wl0]; wb0_##S = w[k1 + wl0];                                                 \
    if (NT > 1) { wa1_##S = w[k0 + wl1]; wb1_##S = w[k1 + wl1]; }                                 \
  }
  // one K chunk: A registers of set SA -> LDS buffer BUF, barrier, request chunk c+2 (A into set SA, W into
  // set SWN), MFMAs of chunk c with the weights of set SW
#define OCC_C1_STEP(SA, SW, SWN, BUF, K_NEXT)                                                     \
  {                                                                                               \
    char* sA = lds + (BUF) * A_BYTES;                                                             \
    *reinterpret_cast<uint4*>(sA + adst[0]) = va0_##SA;                                           \
    if (AP > 1) *reinterpret_cast<uint4*>(sA + adst[AP > 1 ? 1 : 0]) = va1_##SA;                  \
    __syncthreads(); /* chunk visible to every wave; the other A buffer is free */                \
    OCC_C1_ISSUE_A(SA, K_NEXT)                                                                    \
    OCC_C1_ISSUE_W(SWN, K_NEXT)                                                                   \
    bf16x8 af[RT][2];                                                                             \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                             \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                            \
        af[rt][ks] = *reinterpret_cast<const bf16x8*>(sA + (rt * 32 + vi) * kCLD + ks * 32 + kb * 16); \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) {                                           \
      acc[rt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt][0], __builtin_bit_cast(bf16x8, wa0_##SW), acc[rt][0], 0, 0, 0); \
      if (NT > 1) acc[rt][NT - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt][0], __builtin_bit_cast(bf16x8, wa1_##SW), acc[rt][NT - 1], 0, 0, 0); \
    }                                                                                             \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) {                                           \
      acc[rt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt][1], __builtin_bit_cast(bf16x8, wb0_##SW), acc[rt][0], 0, 0, 0); \
      if (NT > 1) acc[rt][NT - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt][1], __builtin_bit_cast(bf16x8, wb1_##SW), acc[rt][NT - 1], 0, 0, 0); \
    }                                                                                             \
  }

  // every block walks the K chunks in a rotated order (start depends on the block): blocks launched
  // together would otherwise request the same weight chunk and same-stride activation columns at the
  // same time (L2 channel hot-spotting); the K sum is order-independent up to f32 rounding
  const int NCHK = K / KC;
  const int rot = (int)((blockIdx.x * 5u + blockIdx.y * 3u) % (unsigned)NCHK);
#define OCC_C1_K(CI) ((((CI) < NCHK ? (CI) : NCHK - 1) + rot) % NCHK * KC)   /* chunk index -> k offset (clamped) */
  OCC_C1_ISSUE_A(0, OCC_C1_K(0))
  OCC_C1_ISSUE_W(0, OCC_C1_K(0))
  OCC_C1_ISSUE_A(1, OCC_C1_K(1))
  OCC_C1_ISSUE_W(1, OCC_C1_K(1))
  for (int ci = 0; ci < NCHK; ci += 6) {
    OCC_C1_STEP(0, 0, 2, 0, OCC_C1_K(ci + 2))
    if (ci + 1 < NCHK) OCC_C1_STEP(1, 1, 0, 1, OCC_C1_K(ci + 3))
    if (ci + 2 < NCHK) OCC_C1_STEP(0, 2, 1, 0, OCC_C1_K(ci + 4))
    if (ci + 3 < NCHK) OCC_C1_STEP(1, 0, 2, 1, OCC_C1_K(ci + 5))
    if (ci + 4 < NCHK) OCC_C1_STEP(0, 1, 0, 0, OCC_C1_K(ci + 6))
    if (ci + 5 < NCHK) OCC_C1_STEP(1, 2, 1, 1, OCC_C1_K(ci + 7))
  }
#undef OCC_C1_K
#undef OCC_C1_STEP
#undef OCC_C1_ISSUE_A
#undef OCC_C1_ISSUE_W

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
      if (residual != nullptr && col_live) {
        long rrow = m;
        if (res_up) {   // residual = a map of half the resolution, nearest-upsampled x2 (FPN top-down path)
          const long hw = (long)Hout * Wout;
          const long n = m / hw, rem = m - n * hw;
          const int yo = (int)(rem / Wout), xo = (int)(rem - (long)yo * Wout);
          rrow = (n * (Hout >> 1) + (yo >> 1)) * (Wout >> 1) + (xo >> 1);
        }
        rv[rr] = *reinterpret_cast<const uint2*>(residual + rrow * N + n0 + c);
      }
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
                                     int Cin, int Cout, int stride, int relu, int residual_upsample2,
                                     void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(x && weight && bias && out, "conv1x1_nhwc_bf16: null pointer argument");
  OCC_CHECK_ARG(batch > 0 && Hin > 0 && Win > 0 && Cin > 0 && Cout > 0 && stride > 0,
                "conv1x1_nhwc_bf16: bad dimension");
  if (Cin % 32 || Cout % 32) {
    set_error("conv1x1_nhwc_bf16: no kernel for Cin=%d Cout=%d (need Cin %% 32 == 0, Cout %% 32 == 0)", Cin,
              Cout);
    return OCC_E_UNSUPPORTED;
  }
  const int Hout = (Hin - 1) / stride + 1, Wout = (Win - 1) / stride + 1;
  OCC_CHECK_ARG(!residual_upsample2 || (residual && Hout % 2 == 0 && Wout % 2 == 0),
                "conv1x1_nhwc_bf16: an upsampled residual needs even output sizes (exact x2 nearest upsampling)");
  const long M = (long)batch * Hout * Wout;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define OCC_C1_LAUNCH(NTT, RTT, BNN)                                                                \
  hipLaunchKernelGGL((conv1x1_nhwc_bf16_kernel<NTT, RTT>),                                          \
                     dim3((unsigned)((M + 32 * RTT - 1) / (32 * RTT)), (unsigned)((Cout + BNN - 1) / BNN)), \
                     dim3(256), 0, st, reinterpret_cast<const uint4*>(x),                           \
                     reinterpret_cast<const uint4*>(weight), bias,                                  \
                     reinterpret_cast<const unsigned short*>(residual),                             \
                     reinterpret_cast<unsigned short*>(out), M, Cout, Cin, Hin, Win, Hout, Wout, stride, relu,  \
                     residual_upsample2)
  // 64-row blocks (128-row blocks were measured on the ResNet-50 shapes and are not faster)
  if (Cout <= 128) OCC_C1_LAUNCH(1, 2, 128); else OCC_C1_LAUNCH(2, 2, 256);
#undef OCC_C1_LAUNCH
  OCC_CHECK_LAUNCH("conv1x1_nhwc_bf16");
  return OCC_OK;
}
