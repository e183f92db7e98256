// nn.Linear with fused epilogues on the gfx950 bf16 matrix cores, bf16x3 arithmetic — round-2 kernel ("x3s":
// operands SHARED through LDS by a 160-row block).
//
// Same contract, weight packing and call sites as linear_bf16x3.hip (occ_linear_bf16x3_f32): out =
// LayerNorm(residual + act([a1 | a2 (+ a2_add)] . W^T + bias)), A.W^T ~= Al.Wh^T + Ah.Wl^T + Ah.Wh^T in f32
// accumulation (product error <= 2^-16).  What changed, from the round-1 measurements (profiles/
// r01_pmc_derived_hotpath_v9.txt: MFMA 30 % busy; r02_e2e trace: 23-65 us per launch against 6-19 us of matrix time):
//   * one barrier per 32 k (30 MFMAs per wave in between) instead of one per 16 k (12 MFMAs);
//   * the weight chunk is staged ONCE per block in LDS (fragment order, 32 KB per 32 k) and read by its wave with
//     ds_read_b128 — the 64-row blocks of round 1 each streamed the whole weight matrix through the texture path;
//   * 160 rows per block: the encoder's M = 40 000 (and 160 000) rows are exactly 250 (1 000) blocks — one block per
//     CU, one round, no tail — and the weights are re-read from L2 2.5x less often;
//   * 8 waves, each owning one 32-column tile and all five 32-row tiles (80 accumulator registers): an A fragment
//     is reused for one MFMA triple, a B fragment for five.
// Decomposition: block = 160 rows x 256 columns, 512 threads.  Per chunk: every thread fetches its share of the
// next chunk (A: 8 fp32 of one row, split into hi/lo bf16 on the way to LDS; B: 4 x 16 bytes of packed weights)
// into registers, the MFMAs of the current chunk run from the other LDS buffer, the registers are written to LDS,
// one barrier.  A rows are padded to 80 bytes (conflict-free ds_read_b128 across the 32 rows of a fragment).
// Epilogue as in round 1: accumulators -> LDS row-major 32-row tile -> bias, ReLU, residual, two-pass LayerNorm
// by one wave per 4 rows -> global.
#include "common.h"

namespace occ {

typedef float s_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 s_bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kSBM = 160, kSBK = 32, kSRT = kSBM / 32, kSALD = 80, kSThreads = 512;
constexpr int kSAPlane = kSBM * kSALD;                 // one plane (hi or lo) of an A chunk
constexpr int kSAStage = 2 * kSAPlane;
constexpr int kSBStage = 2 * 8 * 2 * 1024;             // 2 k-steps x 8 column tiles x (hi, lo) x 1 KB
constexpr int kSStage = kSAStage + kSBStage;           // 58 368 B
constexpr int kSOLD = 256 + 4;                         // epilogue tile row stride (floats)
constexpr int kSLds = 2 * kSStage;                     // 116 736 B (the 33 KB epilogue tile reuses it)

__device__ __forceinline__ void s_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  hi = pack_bf16x2_rne(x0, x1);
  lo = pack_bf16x2_rne(x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u));
}
__device__ __forceinline__ float s_wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

template <bool ADD>
__global__ __launch_bounds__(kSThreads, 2) void linear_x3s_kernel(
    const float* __restrict__ a1, long lda1, int K1, const float* __restrict__ a2,
    const float* __restrict__ a2add, long lda2, int K2, const uint4* __restrict__ wp,
    const float* __restrict__ bias, int act, const float* __restrict__ residual, long ldres,
    const float* __restrict__ ln_g, const float* __restrict__ ln_b, float ln_eps,
    float* __restrict__ out, long ldo, int M, int N) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vi = lane & 31, kb = lane >> 5;
  const long m0 = (long)blockIdx.x * kSBM;
  const int n0 = blockIdx.y * 256;
  const int K = K1 + K2, NT32 = (N + 31) / 32, NCHK = K / kSBK;
  const int nt0 = n0 / 32;

  s_f32x16 acc[kSRT];
#pragma unroll
  for (int rt = 0; rt < kSRT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;

  // A staging role: thread -> (row tid/4 of rows 0..127, 8 k at (tid%4)*8); threads 0..127 also rows 128..159
  const int arow = tid >> 2, asp = tid & 3;
  const bool extra = tid < 4 * (kSBM - 128);
  long am = m0 + arow, ame = m0 + 128 + arow;
  if (am >= M) am = (long)M - 1;
  if (ame >= M) ame = (long)M - 1;
  // B staging role: 4 x 16 bytes; piece r: linear index idx = tid + 512 r in the chunk's [k-step][tile][plane][lane]
  // (named scalars, not arrays: captured by the lambdas below an array stays an alloca and lives in scratch)
  auto bsrc_of = [&](int r) {
    const int idx = tid + kSThreads * r;
    const int s = idx >> 10, rem = idx & 1023;
    const int tile = min(nt0 + (rem >> 7), NT32 - 1);        // clamped: masked in the epilogue
    return ((long)s * NT32 + tile) * 128 + (rem & 127);     // + (k0/16) * NT32 * 128 per chunk
  };
  const long bsrc0 = bsrc_of(0), bsrc1 = bsrc_of(1), bsrc2 = bsrc_of(2), bsrc3 = bsrc_of(3);

  float4 va0, va1, ve0, ve1, vd0, vd1, vf0, vf1;
  uint4 vb0, vb1, vb2, vb3;
  bool cur_add = false;
  auto issue = [&](int k0) {
    // weights first: hipcc's waitcnt pass believes the previous iteration's loads into these registers may still
    // be pending and guards their reuse with counted waits — issued after the activation loads those waits drain
    // the loads just issued (a full memory round trip per chunk)
    const long kc = (long)(k0 / 16) * NT32 * 128;
    vb0 = wp[kc + bsrc0]; vb1 = wp[kc + bsrc1]; vb2 = wp[kc + bsrc2]; vb3 = wp[kc + bsrc3];
    const bool seg2 = k0 >= K1;
    const float* ab = (seg2 ? a2 + (k0 - K1) : a1 + k0) + asp * 8;
    const long lda = seg2 ? lda2 : lda1;
    va0 = *reinterpret_cast<const float4*>(ab + am * lda);
    va1 = *reinterpret_cast<const float4*>(ab + am * lda + 4);
    if (extra) {
      ve0 = *reinterpret_cast<const float4*>(ab + ame * lda);
      ve1 = *reinterpret_cast<const float4*>(ab + ame * lda + 4);
    }
    cur_add = ADD && seg2 && a2add != nullptr;
    if (ADD && cur_add) {                                   // block-uniform
      const float* db = a2add + (k0 - K1) + asp * 8;
      vd0 = *reinterpret_cast<const float4*>(db + am * lda);
      vd1 = *reinterpret_cast<const float4*>(db + am * lda + 4);
      if (extra) {
        vf0 = *reinterpret_cast<const float4*>(db + ame * lda);
        vf1 = *reinterpret_cast<const float4*>(db + ame * lda + 4);
      }
    }
  };
  auto put_row = [&](char* sA, int row, float4 x0, float4 x1) {
    uint4 h, l;
    s_split2(x0.x, x0.y, h.x, l.x); s_split2(x0.z, x0.w, h.y, l.y);
    s_split2(x1.x, x1.y, h.z, l.z); s_split2(x1.z, x1.w, h.w, l.w);
    *reinterpret_cast<uint4*>(sA + row * kSALD + asp * 16) = h;
    *reinterpret_cast<uint4*>(sA + kSAPlane + row * kSALD + asp * 16) = l;
  };
  auto commit = [&](int buf) {                              // registers of the fetched chunk -> LDS buffer `buf`
    char* sA = lds + buf * kSStage;
    char* sB = sA + kSAStage;
    if (ADD && cur_add) {
      va0.x += vd0.x; va0.y += vd0.y; va0.z += vd0.z; va0.w += vd0.w;
      va1.x += vd1.x; va1.y += vd1.y; va1.z += vd1.z; va1.w += vd1.w;
      if (extra) {
        ve0.x += vf0.x; ve0.y += vf0.y; ve0.z += vf0.z; ve0.w += vf0.w;
        ve1.x += vf1.x; ve1.y += vf1.y; ve1.z += vf1.z; ve1.w += vf1.w;
      }
    }
    put_row(sA, arow, va0, va1);
    if (extra) put_row(sA, 128 + arow, ve0, ve1);
    *reinterpret_cast<uint4*>(sB + (tid + kSThreads * 0) * 16) = vb0;
    *reinterpret_cast<uint4*>(sB + (tid + kSThreads * 1) * 16) = vb1;
    *reinterpret_cast<uint4*>(sB + (tid + kSThreads * 2) * 16) = vb2;
    *reinterpret_cast<uint4*>(sB + (tid + kSThreads * 3) * 16) = vb3;
  };

  // blocks walk K in a rotated order (round 1: blocks launched together otherwise hit the same weight chunk /
  // the same activation columns at the same time)
  const int rot = (int)((blockIdx.x * 5u + blockIdx.y * 3u) % (unsigned)NCHK);
  auto kof = [&](int ci) { return ((ci + rot) % NCHK) * kSBK; };

  issue(kof(0));
  commit(0);
  __syncthreads();
  for (int ci = 0; ci < NCHK; ++ci) {
    const int buf = ci & 1;
    if (ci + 1 < NCHK) issue(kof(ci + 1));
    // pin the three phases: left alone, the scheduler sinks each weight load next to its LDS store (one register
    // quad, s_waitcnt vmcnt(0) after every load) and the chunk pays four serial L2 round trips before its MFMAs
    __builtin_amdgcn_sched_barrier(0);
    const char* sA = lds + buf * kSStage;
    const char* sB = sA + kSAStage;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const s_bf16x8 bh = *reinterpret_cast<const s_bf16x8*>(sB + ((s * 8 + wave) * 2 + 0) * 1024 + lane * 16);
      const s_bf16x8 bl = *reinterpret_cast<const s_bf16x8*>(sB + ((s * 8 + wave) * 2 + 1) * 1024 + lane * 16);
#pragma unroll
      for (int rt = 0; rt < kSRT; ++rt) {
        const s_bf16x8 ah = *reinterpret_cast<const s_bf16x8*>(sA + (rt * 32 + vi) * kSALD + s * 32 + kb * 16);
        const s_bf16x8 al =
            *reinterpret_cast<const s_bf16x8*>(sA + kSAPlane + (rt * 32 + vi) * kSALD + s * 32 + kb * 16);
        acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[rt], 0, 0, 0);    // small terms first
        acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[rt], 0, 0, 0);
        acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[rt], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (ci + 1 < NCHK) commit(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue, 32 rows at a time: accumulators -> LDS row-major tile -> 4 rows per wave ----------------------
  const int c = lane * 4;
  const bool col_live = n0 + c < N;
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
  for (int rt = 0; rt < kSRT; ++rt) {
    if (rt) __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r)
      sO[((r & 3) + 8 * (r >> 2) + 4 * kb) * kSOLD + wave * 32 + vi] = acc[rt][r];
    __syncthreads();
    float4 rres[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      long m = m0 + rt * 32 + wave * 4 + rr;
      if (m >= M) m = M - 1;
      rres[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (residual != nullptr && col_live)
        rres[rr] = *reinterpret_cast<const float4*>(residual + m * ldres + n0 + c);
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int row = wave * 4 + rr;
      const long m = m0 + rt * 32 + row;
      if (m >= M) break;                       // wave-uniform
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (col_live) {
        v = *reinterpret_cast<const float4*>(sO + row * kSOLD + c);
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        if (act == 1) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        v.x += rres[rr].x; v.y += rres[rr].y; v.z += rres[rr].z; v.w += rres[rr].w;
      }
      if (ln_g) {                              // LayerNorm over the N columns (N <= 256: one column block)
        const float mean = s_wave_sum(col_live ? (v.x + v.y) + (v.z + v.w) : 0.f) * inv_n;
        const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
        const float var = s_wave_sum(col_live ? (dx * dx + dy * dy) + (dz * dz + dw * dw) : 0.f) * inv_n;
        const float rstd = rsqrtf(var + ln_eps);
        v.x = dx * rstd * gv.x + bev.x; v.y = dy * rstd * gv.y + bev.y;
        v.z = dz * rstd * gv.z + bev.z; v.w = dw * rstd * gv.w + bev.w;
      }
      if (col_live) *reinterpret_cast<float4*>(out + m * ldo + n0 + c) = v;
    }
  }
}

}  // namespace occ

// Same signature and packed-weight format as occ_linear_bf16x3_f32.  Needs K1 % 32 == 0 and K2 % 32 == 0
// (OCC_E_UNSUPPORTED otherwise: the caller falls back to the 16-k kernel).
extern "C" int occ_linear_bf16x3s_f32(const float* a1, int64_t lda1, int K1, const float* a2,
                                      const float* a2_add, int64_t lda2, int K2, const void* weight_packed,
                                      const float* bias, int act, const float* residual, int64_t ldres,
                                      const float* ln_gamma, const float* ln_beta, float ln_eps, float* out,
                                      int64_t ldo, int M, int N, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(a1 && weight_packed && out, "linear_bf16x3s: null pointer argument");
  OCC_CHECK_ARG(M > 0 && N > 0 && K1 > 0 && K2 >= 0, "linear_bf16x3s: bad dimension (M=%d N=%d K1=%d K2=%d)", M, N,
                K1, K2);
  OCC_CHECK_ARG((K2 == 0) == (a2 == nullptr), "linear_bf16x3s: a2 must be given exactly when K2 > 0");
  OCC_CHECK_ARG(!a2_add || a2, "linear_bf16x3s: a2_add without a2");
  OCC_CHECK_ARG(act == 0 || act == 1, "linear_bf16x3s: act must be 0 (none) or 1 (ReLU)");
  OCC_CHECK_ARG((ln_gamma == nullptr) == (ln_beta == nullptr), "linear_bf16x3s: ln_gamma and ln_beta go together");
  OCC_CHECK_ARG(lda1 >= K1 && (K2 == 0 || lda2 >= K2) && ldo >= N && (!residual || ldres >= N),
                "linear_bf16x3s: leading dimension smaller than the row");
  if (K1 % kSBK || K2 % kSBK || N % 4 || lda1 % 4 || lda2 % 4 || ldo % 4 || ldres % 4 || (ln_gamma && N > 256)) {
    set_error("linear_bf16x3s: no kernel for K1=%d K2=%d N=%d (need K %% 32 == 0, N %% 4 == 0, 16-byte aligned rows, "
              "N <= 256 with LayerNorm)", K1, K2, N);
    return OCC_E_UNSUPPORTED;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linear_x3s_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kSLds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linear_x3s_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kSLds);
    attr_done = true;
  }
  const dim3 grid((unsigned)((M + kSBM - 1) / kSBM), (unsigned)((N + 255) / 256));
  const uint4* wp = reinterpret_cast<const uint4*>(weight_packed);
  if (a2_add)
    hipLaunchKernelGGL(linear_x3s_kernel<true>, grid, dim3(kSThreads), kSLds, st, a1, (long)lda1, K1, a2, a2_add,
                       (long)lda2, K2, wp, bias, act, residual, (long)ldres, ln_gamma, ln_beta, ln_eps, out,
                       (long)ldo, M, N);
  else
    hipLaunchKernelGGL(linear_x3s_kernel<false>, grid, dim3(kSThreads), kSLds, st, a1, (long)lda1, K1, a2, a2_add,
                       (long)lda2, K2, wp, bias, act, residual, (long)ldres, ln_gamma, ln_beta, ln_eps, out,
                       (long)ldo, M, N);
  OCC_CHECK_LAUNCH("linear_bf16x3s");
  return OCC_OK;
}
