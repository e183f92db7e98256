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
// Layout: the weight is split and packed ONCE on the device (occ_linear_pack_weight_bf16x3):
//   packed[K/16][n][ hi[16] | lo[16] ] bf16  — the same 4 bytes per weight as f32; CHUNK-major, so the
// slice a wave stages per K chunk is contiguous (rows at a 1-2 KB stride would all land on a few L2
// channels while every block walks K in lockstep); staging is a plain 16-byte-per-lane copy into LDS.
// Activations are split on the fly while staged.
// Decomposition: block = 4 waves x (32*RT rows) x (128*NT columns); per 16-k chunk the activation rows are
// split and staged once per block (double buffered, one barrier per chunk) and every wave stages its own W
// slice into a private LDS region (48-byte row stride = conflict-free ds_read_b128); next chunk's global
// loads in flight during the MFMAs; epilogue (bias, ReLU, residual, two-pass LayerNorm) on row-major rows
// through an LDS transpose.
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

// (N, K) f32 -> packed[K/16][n][hi16 | lo16] bf16
__global__ void linear_pack_weight_bf16x3_kernel(const float* __restrict__ w,
                                                 unsigned short* __restrict__ packed, long n_elem,
                                                 int K, int N) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_elem) return;
  const long n = idx / K;
  const int k = (int)(idx % K);
  unsigned short hi, lo;
  x3_split(w[idx], hi, lo);
  unsigned short* dst = packed + ((long)(k / 16) * N + n) * 32 + (k % 16);
  dst[0] = hi;
  dst[16] = lo;
}

template <int NT, int RT, bool ADD>
__global__ __launch_bounds__(256) void linear_bf16x3_kernel(
    const float* __restrict__ a1, long lda1, int K1, const float* __restrict__ a2,
    const float* __restrict__ a2add, long lda2, int K2, const uint4* __restrict__ wp,
    const float* __restrict__ bias, int act, const float* __restrict__ residual, long ldres,
    const float* __restrict__ ln_g, const float* __restrict__ ln_b, float ln_eps,
    float* __restrict__ out, long ldo, int M, int N) {
  constexpr int BM = 32 * RT, BN = 128 * NT, WR = 32 * NT, OLD = BN + 4;
  constexpr int A_BYTES = BM * kXLD, W_BYTES = WR * kXLD;            // one plane (hi or lo)
  // LDS: the activation chunk (hi + lo planes) is split and staged ONCE per block (double buffered, one
  // barrier per chunk) and read by all four waves; every wave keeps a private region for its weight slice
  constexpr int STAGE_BYTES = 2 * (2 * A_BYTES) + 4 * (2 * W_BYTES), OUT_BYTES = 32 * OLD * 4;
  __shared__ __attribute__((aligned(16))) char lds[STAGE_BYTES > OUT_BYTES ? STAGE_BYTES : OUT_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vi = lane & 31, kb = lane >> 5;
  char* sWh = lds + 4 * A_BYTES + wave * 2 * W_BYTES;
  char* sWl = sWh + W_BYTES;
  const long m0 = (long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int nw0 = n0 + wave * WR;
  const int K = K1 + K2;

  f32x16 acc[RT][NT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rt][t][r] = 0.f;

  // A: thread -> (row = tid/4, 4 k = tid%4) of the block's rows (threads beyond BM rows idle in staging);
  // W: lane -> (row = lane/4 + 16*it, 16-byte piece = lane%4) of the wave's slice.  All loads unconditional
  // (clamped indices, 0/1-scaled addend alias) and held in named registers — see linear_mfma.hip.
  const int arow = tid >> 2, sp = tid & 3, srow = lane >> 2;
  const bool a_live = arow < BM;
  long am = m0 + (a_live ? arow : 0);
  if (am >= M) am = (long)M - 1;
  long wofs[2 * NT];   // uint4 index of this lane's piece in chunk 0
#pragma unroll
  for (int it = 0; it < 2 * NT; ++it) {
    const int n = nw0 + srow + 16 * it;
    wofs[it] = (long)(n < N ? n : N - 1) * 4 + sp;
  }
  float4 va;
  float4 vd = make_float4(0.f, 0.f, 0.f, 0.f);     // addend (ADD only)
  uint4 vw0, vw1, vw2, vw3;                        // W rows it = 0..3 (it >= 2 only for NT == 2)
  float addscale = 0.f;
#define OCC_X3_ISSUE(K0)                                                                          \
  {                                                                                               \
    const int k0_ = (K0);                                                                         \
    const bool seg2 = k0_ >= K1;                                                                  \
    const float* ab = (seg2 ? a2 + (k0_ - K1) : a1 + k0_) + sp * 4;                               \
    const long lda = seg2 ? lda2 : lda1;                                                          \
    const bool add = seg2 && a2add != nullptr;                                                    \
    const float* addb = add ? a2add + (k0_ - K1) + sp * 4 : ab;                                   \
    addscale = add ? 1.f : 0.f;                                                                   \
    va = *reinterpret_cast<const float4*>(ab + am * lda);                                         \
    if (ADD) vd = *reinterpret_cast<const float4*>(addb + am * lda); /* compile-time */           \
    const long kc4 = (long)(k0_ / 16) * N * 4; /* chunk-major packed weights */                   \
    vw0 = wp[wofs[0] + kc4];                                                                      \
    vw1 = wp[wofs[1] + kc4];                                                                      \
    if (NT == 2) {                                                                                \
      vw2 = wp[wofs[2 * NT - 2] + kc4];                                                           \
      vw3 = wp[wofs[2 * NT - 1] + kc4];                                                           \
    }                                                                                             \
  }
#define OCC_X3_PUT_W(V, ROW)                                                                      \
  *reinterpret_cast<uint4*>((sp < 2 ? sWh : sWl) + (ROW) * kXLD + (sp & 1) * 16) = V;

  // every block walks the K chunks in a rotated order (see conv1x1_nhwc_bf16.hip): blocks launched together
  // would otherwise request the same weight chunk / same-stride activation columns at the same time
  const int NCHK = K / kXBK;
  const int rot = (int)((blockIdx.x * 5u + blockIdx.y * 3u) % (unsigned)NCHK);
  OCC_X3_ISSUE((rot % NCHK) * kXBK)
  int buf = 0;
  for (int ci = 0; ci < NCHK; ++ci, buf ^= 1) {
    char* sAh = lds + buf * 2 * A_BYTES;
    char* sAl = sAh + A_BYTES;
    if (a_live) {   // f32 x4 (+ addend) -> 4 hi bf16 + 4 lo bf16 into the two A planes
      const float f0 = ADD ? fmaf(addscale, vd.x, va.x) : va.x, f1 = ADD ? fmaf(addscale, vd.y, va.y) : va.y;
      const float f2 = ADD ? fmaf(addscale, vd.z, va.z) : va.z, f3 = ADD ? fmaf(addscale, vd.w, va.w) : va.w;
      unsigned h01, h23, l01, l23;
      x3_split2(f0, f1, h01, l01); x3_split2(f2, f3, h23, l23);
      *reinterpret_cast<uint2*>(sAh + arow * kXLD + sp * 8) = make_uint2(h01, h23);
      *reinterpret_cast<uint2*>(sAl + arow * kXLD + sp * 8) = make_uint2(l01, l23);
    }
    OCC_X3_PUT_W(vw0, srow)
    OCC_X3_PUT_W(vw1, srow + 16)
    if (NT == 2) {
      OCC_X3_PUT_W(vw2, srow + 32)
      OCC_X3_PUT_W(vw3, srow + 48)
    }
    __syncthreads();   // chunk visible to every wave; the other A buffer is free for the next iteration
    OCC_X3_ISSUE((((ci + 1 < NCHK ? ci + 1 : ci) + rot) % NCHK) * kXBK)   // unconditional prefetch

    bf16x8 ah[RT], al[RT], wh[NT], wl[NT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      ah[rt] = *reinterpret_cast<const bf16x8*>(sAh + (rt * 32 + vi) * kXLD + kb * 16);
      al[rt] = *reinterpret_cast<const bf16x8*>(sAl + (rt * 32 + vi) * kXLD + kb * 16);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      wh[t] = *reinterpret_cast<const bf16x8*>(sWh + (t * 32 + vi) * kXLD + kb * 16);
      wl[t] = *reinterpret_cast<const bf16x8*>(sWl + (t * 32 + vi) * kXLD + kb * 16);
    }
    wave_lds_sync();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int t = 0; t < NT; ++t) {   // small terms first
        acc[rt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[rt], wh[t], acc[rt][t], 0, 0, 0);
        acc[rt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[rt], wl[t], acc[rt][t], 0, 0, 0);
        acc[rt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[rt], wh[t], acc[rt][t], 0, 0, 0);
      }
  }
#undef OCC_X3_ISSUE
#undef OCC_X3_PUT_W

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
  const float inv_n = 1.f / (float)N;
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
  const long n = (long)N * K;
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
