// nn.Linear with fused epilogues on the gfx950 bf16 matrix cores at (near-)fp32 accuracy: "bf16x3".
//
// Same contract and call sites as linear_mfma.hip (occ_linear_f32) — this is its fast variant.  Every f32
// operand x is split into two bf16 numbers, hi = bf16(x), lo = bf16(x - hi) (16 mantissa bits together),
// and  A.W^T ~= Ah.Wh^T + Ah.Wl^T + Al.Wh^T  is accumulated in f32 by v_mfma_f32_32x32x16_bf16: three
// bf16 MFMAs (32 cycles each per 16 k) replace eight f32 MFMAs (64 cycles each per 2 k): 5.3x less
// matrix-pipe time; the dropped Al.Wl term and the split's rounding bound the relative error of a product
// at 2^-16 (1.5e-5; the f32 kernel: 6e-8), far inside the path's 1e-3 parity budget.  gfx950 has no
// xf32/TF32 matrix instruction, so this is the only way to get fp32-like GEMMs off the 157 TFLOP/s f32 rate.
//
// Layout: the weight is split and packed ONCE on the device (occ_linear_pack_weight_bf16x3) in MFMA
// B-fragment order,  packed[K/16][ceil(N/32)][hi | lo][lane][8 bf16]  (columns padded with zeros to a multiple
// of 32; 4 bytes per weight like f32): the operand of a wave for (k-step, column tile, plane) is ONE coalesced
// 1 KB global load straight into registers.  Every wave owns its own 32*NT columns, so the weights never touch
// LDS (as a layout transposer it made the kernel LDS-bound).  Activations are split on the fly while staged.
// Decomposition: block = 4 waves x (32*RT rows) x (128*NT columns); per 16-k chunk the activation rows are
// split and staged once per block (hi + lo planes, double buffered, one barrier per chunk, 48-byte row stride =
// conflict-free ds_read_b128); activation loads run two chunks ahead, weight loads three (issued as one group
// per chunk, so the in-order vmcnt wait for chunk c never drains younger prefetches); blocks walk K in a
// rotated order (L2 channel hot-spotting); epilogue (bias, ReLU, residual, two-pass LayerNorm) on row-major
// rows through an LDS transpose.
#include "common.h"

namespace occ {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kXBK = 16;          // k per chunk = one 32x32x16 MFMA step
constexpr int kXLD = 48;          // LDS row stride in BYTES: 16 bf16 (32 B) + 16 B pad

__device__ __forceinline__ unsigned short x3_bf16_rne(float f) { return bf16_rne(f); }
// two values at once: hi pair = cvt_pk(x0, x1), lo pair = cvt_pk(x0 - hi0, x1 - hi1)   (6 VALU ops per pair)
__device__ __forceinline__ void x3_split(float x, unsigned short& hi, unsigned short& lo) {
  hi = x3_bf16_rne(x);
  lo = x3_bf16_rne(x - __uint_as_float((unsigned)hi << 16));
}
__device__ __forceinline__ void x3_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  hi = pack_bf16x2_rne(x0, x1);
  lo = pack_bf16x2_rne(x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u));
}
__device__ __forceinline__ float x3_wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// (N, K) f32 -> packed[K/16][ceil(N/32)][plane: hi, lo][lane][8] bf16 (zero columns beyond N)
__global__ void linear_pack_weight_bf16x3_kernel(const float* __restrict__ w,
                                                 unsigned short* __restrict__ packed, long n_elem,
                                                 int K, int N) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;     // one (hi, lo) pair per thread
  if (idx >= n_elem) return;
  const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
  const long rest = idx >> 9;
  const int nt32 = (N + 31) / 32;
  const int nt = (int)(rest % nt32), ks = (int)(rest / nt32);
  const int n = nt * 32 + (lane & 31), k = ks * 16 + (lane >> 5) * 8 + j;
  unsigned short hi = 0, lo = 0;
  if (n < N) x3_split(w[(long)n * K + k], hi, lo);
  unsigned short* dst = packed + ((rest * 2) * 64 + lane) * 8 + j;
  dst[0] = hi;
  dst[64 * 8] = lo;
}

template <int NT, int RT, bool ADD>
__global__ __launch_bounds__(256) void linear_bf16x3_kernel(
    const float* __restrict__ a1, long lda1, int K1, const float* __restrict__ a2,
    const float* __restrict__ a2add, long lda2, int K2, const uint4* __restrict__ wp,
    const float* __restrict__ bias, int act, const float* __restrict__ residual, long ldres,
    const float* __restrict__ ln_g, const float* __restrict__ ln_b, float ln_eps,
    float* __restrict__ out, long ldo, int M, int N) {
  constexpr int BM = 32 * RT, BN = 128 * NT, WR = 32 * NT, OLD = BN + 4;
  constexpr int A_BYTES = BM * kXLD;                                 // one plane (hi or lo)
  // LDS: only the activation chunk (hi + lo planes), split and staged ONCE per block (double buffered, one
  // barrier per chunk) and read by all four waves
  constexpr int STAGE_BYTES = 2 * (2 * A_BYTES), OUT_BYTES = 32 * OLD * 4;
  __shared__ __attribute__((aligned(16))) char lds[STAGE_BYTES > OUT_BYTES ? STAGE_BYTES : OUT_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vi = lane & 31, kb = lane >> 5;
  const long m0 = (long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int K = K1 + K2;
  const int NT32 = (N + 31) / 32;

  f32x16 acc[RT][NT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rt][t][r] = 0.f;

  // A: thread -> (row = tid/4, 4 k = tid%4) of the block's rows (threads beyond BM rows idle in staging).
  // All loads unconditional (clamped indices, 0/1-scaled addend alias), named registers — see linear_mfma.hip.
  const int arow = tid >> 2, sp = tid & 3;
  const bool a_live = arow < BM;
  long am = m0 + (a_live ? arow : 0);
  if (am >= M) am = (long)M - 1;
  // this wave's column tiles in the packed weight (clamped: a wave past the last tile recomputes it, its
  // columns are masked in the epilogue); uint4 index of (tile, plane hi, lane), the lo plane is +64
  static_assert(NT <= 2, "weight register budget");
  const int nt0 = min((n0 + wave * WR) / 32, NT32 - 1), nt1 = min((n0 + wave * WR) / 32 + (NT - 1), NT32 - 1);
  const long wl0 = (long)nt0 * 128 + lane, wl1 = (long)nt1 * 128 + lane;
  // two activation sets (two chunks ahead), three weight sets (the set of chunk c+2 is requested while the
  // MFMAs of chunk c still read theirs)
  float4 va_0, va_1, vd_0 = make_float4(0.f, 0.f, 0.f, 0.f), vd_1 = vd_0;
  float as_0 = 0.f, as_1 = 0.f;                    // 1 when the chunk carries the addend (ADD only)
  uint4 wh0_0, wl0_0, wh1_0, wl1_0, wh0_1, wl0_1, wh1_1, wl1_1, wh0_2, wl0_2, wh1_2, wl1_2;
#define OCC_X3_ISSUE_A(S, K0)                                                                     \
  {                                                                                               \
    const int k0_ = (K0);                                                                         \
    const bool seg2 = k0_ >= K1;                                                                  \
    const float* ab = (seg2 ? a2 + (k0_ - K1) : a1 + k0_) + sp * 4;                               \
    const long lda = seg2 ? lda2 : lda1;                                                          \
    const bool add = seg2 && a2add != nullptr;                                                    \
    const float* addb = add ? a2add + (k0_ - K1) + sp * 4 : ab;                                   \
    as_##S = add ? 1.f : 0.f;                                                                     \
    va_##S = *reinterpret_cast<const float4*>(ab + am * lda);                                     \
    if (ADD) vd_##S = *reinterpret_cast<const float4*>(addb + am * lda); /* compile-time */       \
  }
#define OCC_X3_ISSUE_W(S, K0)                                                                     \
  {                                                                                               \
    const long kc = (long)((K0) / 16) * NT32 * 128;                                               \
    wh0_##S = wp[kc + wl0]; wl0_##S = wp[kc + wl0 + 64];                                          \
    if (NT > 1) { wh1_##S = wp[kc + wl1]; wl1_##S = wp[kc + wl1 + 64]; }                          \
  }
  // one 16-k chunk: A registers of set SA split into LDS buffer BUF, barrier, request chunk c+2 (A into set
  // SA, W into set SWN), MFMAs of chunk c with the weights of set SW (small terms first)
#define OCC_X3_STEP(SA, SW, SWN, BUF, K_NEXT)                                                     \
  {                                                                                               \
    char* sAh = lds + (BUF) * 2 * A_BYTES;                                                        \
    char* sAl = sAh + A_BYTES;                                                                    \
    if (a_live) {   /* f32 x4 (+ addend) -> 4 hi bf16 + 4 lo bf16 into the two A planes */        \
      const float f0 = ADD ? fmaf(as_##SA, vd_##SA.x, va_##SA.x) : va_##SA.x;                     \
      const float f1 = ADD ? fmaf(as_##SA, vd_##SA.y, va_##SA.y) : va_##SA.y;                     \
      const float f2 = ADD ? fmaf(as_##SA, vd_##SA.z, va_##SA.z) : va_##SA.z;                     \
      const float f3 = ADD ? fmaf(as_##SA, vd_##SA.w, va_##SA.w) : va_##SA.w;                     \
      unsigned h01, h23, l01, l23;                                                                \
      x3_split2(f0, f1, h01, l01); x3_split2(f2, f3, h23, l23);                                   \
      *reinterpret_cast<uint2*>(sAh + arow * kXLD + sp * 8) = make_uint2(h01, h23);               \
      *reinterpret_cast<uint2*>(sAl + arow * kXLD + sp * 8) = make_uint2(l01, l23);               \
    }                                                                                             \
    __syncthreads();   /* chunk visible to every wave; the other A buffer is free */              \
    OCC_X3_ISSUE_A(SA, K_NEXT)                                                                    \
    OCC_X3_ISSUE_W(SWN, K_NEXT)                                                                   \
    bf16x8 ah[RT], al[RT];                                                                        \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) {                                           \
      ah[rt] = *reinterpret_cast<const bf16x8*>(sAh + (rt * 32 + vi) * kXLD + kb * 16);           \
      al[rt] = *reinterpret_cast<const bf16x8*>(sAl + (rt * 32 + vi) * kXLD + kb * 16);           \
    }                                                                                             \
    /* term-major order (small terms first): back-to-back MFMAs on ONE accumulator wait out the 64-cycle result  \
       latency (issue: 32); walking the RT*NT tiles per term keeps the pipe fed */                               \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) {                                           \
      acc[rt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[rt], __builtin_bit_cast(bf16x8, wh0_##SW), acc[rt][0], 0, 0, 0); \
      if (NT > 1) acc[rt][NT - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[rt], __builtin_bit_cast(bf16x8, wh1_##SW), acc[rt][NT - 1], 0, 0, 0); \
    }                                                                                             \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) {                                           \
      acc[rt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[rt], __builtin_bit_cast(bf16x8, wl0_##SW), acc[rt][0], 0, 0, 0); \
      if (NT > 1) acc[rt][NT - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[rt], __builtin_bit_cast(bf16x8, wl1_##SW), acc[rt][NT - 1], 0, 0, 0); \
    }                                                                                             \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) {                                           \
      acc[rt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[rt], __builtin_bit_cast(bf16x8, wh0_##SW), acc[rt][0], 0, 0, 0); \
      if (NT > 1) acc[rt][NT - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[rt], __builtin_bit_cast(bf16x8, wh1_##SW), acc[rt][NT - 1], 0, 0, 0); \
    }                                                                                             \
  }

  // every block walks the K chunks in a rotated order (see conv1x1_nhwc_bf16.hip): blocks launched together
  // would otherwise request the same weight chunk / same-stride activation columns at the same time
  const int NCHK = K / kXBK;
  const int rot = (int)((blockIdx.x * 5u + blockIdx.y * 3u) % (unsigned)NCHK);
#define OCC_X3_K(CI) ((((CI) < NCHK ? (CI) : NCHK - 1) + rot) % NCHK * kXBK)   /* chunk index -> k offset */
  OCC_X3_ISSUE_A(0, OCC_X3_K(0))
  OCC_X3_ISSUE_W(0, OCC_X3_K(0))
  OCC_X3_ISSUE_A(1, OCC_X3_K(1))
  OCC_X3_ISSUE_W(1, OCC_X3_K(1))
  for (int ci = 0; ci < NCHK; ci += 6) {
    OCC_X3_STEP(0, 0, 2, 0, OCC_X3_K(ci + 2))
    if (ci + 1 < NCHK) OCC_X3_STEP(1, 1, 0, 1, OCC_X3_K(ci + 3))
    if (ci + 2 < NCHK) OCC_X3_STEP(0, 2, 1, 0, OCC_X3_K(ci + 4))
    if (ci + 3 < NCHK) OCC_X3_STEP(1, 0, 2, 1, OCC_X3_K(ci + 5))
    if (ci + 4 < NCHK) OCC_X3_STEP(0, 1, 0, 0, OCC_X3_K(ci + 6))
    if (ci + 5 < NCHK) OCC_X3_STEP(1, 2, 1, 1, OCC_X3_K(ci + 7))
  }
#undef OCC_X3_K
#undef OCC_X3_STEP
#undef OCC_X3_ISSUE_A
#undef OCC_X3_ISSUE_W

  // ---- epilogue, 32 rows at a time: accumulators -> LDS row-major tile -> 8 rows per wave ------------
  const int c = lane * 4;
  const bool col_live = c < BN && n0 + c < N;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), gv = bv, bev = bv;
  if (col_live) {
    if (bias) bv = *reinterpret_cast<const float4*>(bias + n0 + c);
    if (ln_g) {
      gv = *reinterpret_cast<const float4*>(ln_g + n0 + c);
      bev = *reinterpret_cast<const float4*>(ln_b + n0 + c);
    }
  }
  const float inv_n = fdiv(1.f, (float)N);     // (no `/` on fp32 in device code: common.h)
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
    // a load inside the row loop would serialise 8 dependent memory round trips
    float4 rres[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      long m = m0 + rt * 32 + wave * 8 + rr;
      if (m >= M) m = M - 1;
      rres[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (residual != nullptr && col_live)
        rres[rr] = *reinterpret_cast<const float4*>(residual + m * ldres + n0 + c);
    }
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int row = wave * 8 + rr;
      const long m = m0 + rt * 32 + row;
      if (m >= M) break;                       // wave-uniform
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (col_live) {
        v = *reinterpret_cast<const float4*>(sO + row * OLD + c);
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        if (act == 1) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        v.x += rres[rr].x; v.y += rres[rr].y; v.z += rres[rr].z; v.w += rres[rr].w;
      }
      if (ln_g) {                              // LayerNorm over the N columns (N <= BN, one column block)
        const float mean = x3_wave_sum(col_live ? (v.x + v.y) + (v.z + v.w) : 0.f) * inv_n;
        const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
        const float var = x3_wave_sum(col_live ? (dx * dx + dy * dy) + (dz * dz + dw * dw) : 0.f) * inv_n;
        const float rstd = rsqrtf(var + ln_eps);
        v.x = dx * rstd * gv.x + bev.x; v.y = dy * rstd * gv.y + bev.y;
        v.z = dz * rstd * gv.z + bev.z; v.w = dw * rstd * gv.w + bev.w;
      }
      if (col_live) *reinterpret_cast<float4*>(out + m * ldo + n0 + c) = v;
    }
  }
}


}  // namespace occ

extern "C" int occ_linear_pack_weight_bf16x3(const float* weight, void* packed, int N, int K,
                                             void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(weight && packed, "linear_pack_weight_bf16x3: null pointer argument");
  OCC_CHECK_ARG(N > 0 && K > 0, "linear_pack_weight_bf16x3: bad dimension");
  if (K % 16) {
    set_error("linear_pack_weight_bf16x3: K=%d is not a multiple of 16", K);
    return OCC_E_UNSUPPORTED;
  }
  const long n = (long)((N + 31) / 32) * 32 * K;      // (hi, lo) pairs incl. the zero columns
  hipLaunchKernelGGL(linear_pack_weight_bf16x3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), weight,
                     reinterpret_cast<unsigned short*>(packed), n, K, N);
  OCC_CHECK_LAUNCH("linear_pack_weight_bf16x3");
  return OCC_OK;
}

extern "C" int occ_linear_bf16x3_f32(const float* a1, int64_t lda1, int K1, const float* a2,
                                     const float* a2_add, int64_t lda2, int K2,
                                     const void* weight_packed, const float* bias, int act,
                                     const float* residual, int64_t ldres, const float* ln_gamma,
                                     const float* ln_beta, float ln_eps, float* out, int64_t ldo, int M,
                                     int N, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(a1 && weight_packed && out, "linear_bf16x3: null pointer argument");
  OCC_CHECK_ARG(M > 0 && N > 0 && K1 > 0 && K2 >= 0, "linear_bf16x3: bad dimension (M=%d N=%d K1=%d K2=%d)",
                M, N, K1, K2);
  OCC_CHECK_ARG((K2 == 0) == (a2 == nullptr), "linear_bf16x3: a2 must be given exactly when K2 > 0");
  OCC_CHECK_ARG(!a2_add || a2, "linear_bf16x3: a2_add without a2");
  OCC_CHECK_ARG(act == 0 || act == 1, "linear_bf16x3: act must be 0 (none) or 1 (ReLU)");
  OCC_CHECK_ARG((ln_gamma == nullptr) == (ln_beta == nullptr),
                "linear_bf16x3: ln_gamma and ln_beta go together");
  OCC_CHECK_ARG(lda1 >= K1 && (K2 == 0 || lda2 >= K2) && ldo >= N && (!residual || ldres >= N),
                "linear_bf16x3: leading dimension smaller than the row");
  if (K1 % kXBK || K2 % kXBK || N % 4 || lda1 % 4 || lda2 % 4 || ldo % 4 || ldres % 4 ||
      (ln_gamma && N > 256)) {
    set_error("linear_bf16x3: no kernel for K1=%d K2=%d N=%d (need K %% 16 == 0, N %% 4 == 0, 16-byte "
              "aligned rows, N <= 256 with LayerNorm)", K1, K2, N);
    return OCC_E_UNSUPPORTED;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const uint4* wp = reinterpret_cast<const uint4*>(weight_packed);
  // two row tiles per wave (W slice staged once per 64 rows: a third less L1 traffic per flop — the
  // kernel's bound) whenever 64-row blocks still give every CU at least two of them
  const bool rt2 = ((long)(M + 63) / 64) * ((N + 255) / 256) >= 2L * 256;
#define OCC_X3_LAUNCH_(NTT, RTT, BNN, ADDD)                                                         \
  hipLaunchKernelGGL((linear_bf16x3_kernel<NTT, RTT, ADDD>),                                        \
                     dim3((unsigned)((M + 32 * RTT - 1) / (32 * RTT)), (unsigned)((N + BNN - 1) / BNN)), \
                     dim3(256), 0, st, a1, (long)lda1, K1, a2, a2_add, (long)lda2, K2, wp, bias, act, \
                     residual, (long)ldres, ln_gamma, ln_beta, ln_eps, out, (long)ldo, M, N)
#define OCC_X3_LAUNCH(NTT, RTT, BNN)                                                                \
  do {                                                                                              \
    if (a2_add) OCC_X3_LAUNCH_(NTT, RTT, BNN, true); else OCC_X3_LAUNCH_(NTT, RTT, BNN, false);      \
  } while (0)
  if (N <= 128) {
    if (rt2) OCC_X3_LAUNCH(1, 2, 128); else OCC_X3_LAUNCH(1, 1, 128);
  } else {
    if (rt2) OCC_X3_LAUNCH(2, 2, 256); else OCC_X3_LAUNCH(2, 1, 256);
  }
#undef OCC_X3_LAUNCH
#undef OCC_X3_LAUNCH_
  OCC_CHECK_LAUNCH("linear_bf16x3");
  return OCC_OK;
}
