// nn.Linear on the gfx950 f32 matrix cores with the encoder's elementwise tail fused into the epilogue:
//   out = LayerNorm( residual + act( [A1 | A2 (+ A2add)] @ W^T + bias ) )        (every stage optional)
//
// Replaces, per BEVFormer encoder layer (reference files under projects/mmdet3d_plugin/bevformer/modules/):
//   temporal_self_attention.py:197-201  cat([value[:bs], query(+query_pos)], -1) -> sampling_offsets /
//                                        attention_weights Linears   (two K segments, addend on the 2nd)
//   temporal_self_attention.py:204, spatial_cross_attention.py:334   value_proj
//   spatial_cross_attention.py:338-341                               query-side Linears
//   temporal_self_attention.py:266-272, spatial_cross_attention.py:173-175  output_proj + residual
//   encoder.py:377-404 (mmcv FFN + LayerNorm)  Linear+ReLU, Linear+residual,
//                                                                     and the LayerNorm after each op
// v_mfma_f32_32x32x2_f32 is exact f32 (an fmaf chain), so the only difference to the reference's
// ATen GEMM is the order of the K sum.
//
// Block = 4 waves, tile BM = 32 rows x BN = 128*NT columns (NT 32-column MFMA tiles per wave; wave w
// owns columns [w*32*NT, (w+1)*32*NT)).  The main loop has NO block barrier: every wave stages what it
// contracts — the 32 x 16 A chunk (redundantly per wave, it comes from L1/L2) and its own 32*NT x 16
// slice of W — into a private LDS region (20-float row stride = 80 B: the 32 rows a ds_read_b128
// touches land on distinct 16-byte bank slots), so the 16 waves a CU holds free-run against each other
// and hide one another's global-load and LDS round trips; the next chunk's global loads are issued
// before the current chunk's MFMAs.  k-pairing as in the Conv3d kernel: lanes 0-31 contract the chunk's
// first 8 k, lanes 32-63 the last 8, so a lane's fragment is 8 contiguous floats of one row.
// Epilogue (one __syncthreads): the accumulators are transposed through LDS into row-major; every wave
// then owns 8 rows, 4 consecutive columns per lane (coalesced float4 traffic for bias / residual /
// output) and the LayerNorm statistics are two wave reductions per row (two-pass variance, as ATen).
#include "common.h"

namespace occ {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kLinBM = 32, kLinBK = 16, kLinLD = 20;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

template <int NT>
__global__ __launch_bounds__(256) void linear_mfma_kernel(
    const float* __restrict__ a1, long lda1, int K1, const float* __restrict__ a2,
    const float* __restrict__ a2add, long lda2, int K2, const float* __restrict__ w,
    const float* __restrict__ bias, int act, const float* __restrict__ residual, long ldres,
    const float* __restrict__ ln_g, const float* __restrict__ ln_b, float ln_eps,
    float* __restrict__ out, long ldo, int M, int N) {
  constexpr int BN = 128 * NT, LD = kLinLD, OLD = BN + 4, WR = 32 * NT;   // W rows per wave
  constexpr int WAVE_FLOATS = (kLinBM + WR) * LD;
  constexpr int STAGE_FLOATS = 4 * WAVE_FLOATS, OUT_FLOATS = kLinBM * OLD;
  __shared__ __attribute__((aligned(16))) float lds[STAGE_FLOATS > OUT_FLOATS ? STAGE_FLOATS : OUT_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vi = lane & 31, kh = lane >> 5;
  float* sA = lds + wave * WAVE_FLOATS;
  float* sW = sA + kLinBM * LD;
  const long m0 = (long)blockIdx.x * kLinBM;
  const int n0 = blockIdx.y * BN;
  const int nw0 = n0 + wave * WR;          // first W row (output column) of this wave
  const int K = K1 + K2;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // staging roles inside the wave: lane -> (row = lane/4 + 16*it, 16-byte part = lane%4)
  const int srow = lane >> 2, spart = lane & 3;
  // All loads are UNCONDITIONAL (row / column indices clamped instead of predicated, the addend read
  // from a valid alias and scaled by 0/1): a branch around a load makes hipcc drain vmcnt(0) at the
  // join and serialises the prefetch.  Rows >= M and columns >= N compute garbage that is never stored.
  // The prefetch registers are named scalars, not arrays: loop-carried arrays indexed by a
  // template-dependent trip count were left in scratch by hipcc (ROCm 7.2).
  const long mr0 = m0 + srow, mr1 = m0 + srow + 16;
  const long arow0 = mr0 < M ? mr0 : (long)M - 1, arow1 = mr1 < M ? mr1 : (long)M - 1;
  const long wo0 = (long)min(nw0 + srow, N - 1) * K + spart * 4;
  const long wo1 = (long)min(nw0 + srow + 16, N - 1) * K + spart * 4;
  const long wo2 = (long)min(nw0 + srow + 32, N - 1) * K + spart * 4;   // NT == 2 only
  const long wo3 = (long)min(nw0 + srow + 48, N - 1) * K + spart * 4;
  float4 va0, va1, vd0, vd1, vw0, vw1, vw2, vw3;
  float addscale = 0.f;
#define OCC_LIN_ISSUE_LOADS(K0)                                                                   \
  {                                                                                               \
    const int k0_ = (K0);                                                                         \
    const bool seg2 = k0_ >= K1; /* wave-uniform scalar selects */                                \
    const float* ab = (seg2 ? a2 + (k0_ - K1) : a1 + k0_) + spart * 4;                            \
    const long lda = seg2 ? lda2 : lda1;                                                          \
    const bool add = seg2 && a2add != nullptr;                                                    \
    const float* addb = add ? a2add + (k0_ - K1) + spart * 4 : ab;                                \
    addscale = add ? 1.f : 0.f;                                                                   \
    va0 = *reinterpret_cast<const float4*>(ab + arow0 * lda);                                     \
    va1 = *reinterpret_cast<const float4*>(ab + arow1 * lda);                                     \
    vd0 = *reinterpret_cast<const float4*>(addb + arow0 * lda);                                   \
    vd1 = *reinterpret_cast<const float4*>(addb + arow1 * lda);                                   \
    vw0 = *reinterpret_cast<const float4*>(w + wo0 + k0_);                                        \
    vw1 = *reinterpret_cast<const float4*>(w + wo1 + k0_);                                        \
    if (NT == 2) {                                                                                \
      vw2 = *reinterpret_cast<const float4*>(w + wo2 + k0_);                                      \
      vw3 = *reinterpret_cast<const float4*>(w + wo3 + k0_);                                      \
    }                                                                                             \
  }

  OCC_LIN_ISSUE_LOADS(0)
  for (int k0 = 0; k0 < K; k0 += kLinBK) {
    // registers (chunk k0) -> this wave's LDS region; its previous fragment reads have completed
    va0.x = fmaf(addscale, vd0.x, va0.x); va0.y = fmaf(addscale, vd0.y, va0.y);
    va0.z = fmaf(addscale, vd0.z, va0.z); va0.w = fmaf(addscale, vd0.w, va0.w);
    va1.x = fmaf(addscale, vd1.x, va1.x); va1.y = fmaf(addscale, vd1.y, va1.y);
    va1.z = fmaf(addscale, vd1.z, va1.z); va1.w = fmaf(addscale, vd1.w, va1.w);
    *reinterpret_cast<float4*>(sA + srow * LD + spart * 4) = va0;
    *reinterpret_cast<float4*>(sA + (srow + 16) * LD + spart * 4) = va1;
    *reinterpret_cast<float4*>(sW + srow * LD + spart * 4) = vw0;
    *reinterpret_cast<float4*>(sW + (srow + 16) * LD + spart * 4) = vw1;
    if (NT == 2) {
      *reinterpret_cast<float4*>(sW + (srow + 32) * LD + spart * 4) = vw2;
      *reinterpret_cast<float4*>(sW + (srow + 48) * LD + spart * 4) = vw3;
    }
    wave_lds_sync();
    // next chunk's loads stay in flight during this chunk's MFMAs (the last iteration re-reads its own
    // chunk: an unconditional prefetch keeps the schedule branch-free)
    OCC_LIN_ISSUE_LOADS(k0 + kLinBK < K ? k0 + kLinBK : k0)

    float af[8], bf[NT][8];
#pragma unroll
    for (int s4 = 0; s4 < 2; ++s4) {
      const float4 x = *reinterpret_cast<const float4*>(sA + vi * LD + kh * 8 + s4 * 4);
      af[s4 * 4 + 0] = x.x; af[s4 * 4 + 1] = x.y; af[s4 * 4 + 2] = x.z; af[s4 * 4 + 3] = x.w;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s4 = 0; s4 < 2; ++s4) {
        const float4 x = *reinterpret_cast<const float4*>(sW + (t * 32 + vi) * LD + kh * 8 + s4 * 4);
        bf[t][s4 * 4 + 0] = x.x; bf[t][s4 * 4 + 1] = x.y; bf[t][s4 * 4 + 2] = x.z; bf[t][s4 * 4 + 3] = x.w;
      }
    wave_lds_sync();   // fragment reads retire before the next iteration overwrites the region
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[t][s], acc[t], 0, 0, 0);
  }

#undef OCC_LIN_ISSUE_LOADS

  // ---- epilogue: accumulators -> LDS row-major tile -> 8 rows per wave ------------------------------
  __syncthreads();
  float* sO = lds;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      sO[((r & 3) + 8 * (r >> 2) + 4 * kh) * OLD + (wave * NT + t) * 32 + vi] = acc[t][r];
  __syncthreads();

  const int c = lane * 4;                  // this lane's 4 columns inside the block tile
  const bool col_live = c < BN && n0 + c < N;   // N % 4 == 0 is required by the host wrapper
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), gv = bv, bev = bv;
  if (col_live) {
    if (bias) bv = *reinterpret_cast<const float4*>(bias + n0 + c);
    if (ln_g) {
      gv = *reinterpret_cast<const float4*>(ln_g + n0 + c);
      bev = *reinterpret_cast<const float4*>(ln_b + n0 + c);
    }
  }
  const float inv_n = fdiv(1.f, (float)N);     // (no `/` on fp32 in device code: common.h)
  // all 8 residual rows of this wave are requested before any is consumed (clamped, unconditional):
  // a load inside the row loop would serialise 8 dependent memory round trips
  float4 rres[8];
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) {
    long m = m0 + wave * 8 + rr;
    if (m >= M) m = M - 1;
    rres[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (residual != nullptr && col_live)
      rres[rr] = *reinterpret_cast<const float4*>(residual + m * ldres + n0 + c);
  }
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) {
    const int row = wave * 8 + rr;
    const long m = m0 + row;
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
      const float mean = wave_sum(col_live ? (v.x + v.y) + (v.z + v.w) : 0.f) * inv_n;
      const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
      const float var = wave_sum(col_live ? (dx * dx + dy * dy) + (dz * dz + dw * dw) : 0.f) * inv_n;
      const float rstd = rsqrtf(var + ln_eps);
      v.x = dx * rstd * gv.x + bev.x; v.y = dy * rstd * gv.y + bev.y;
      v.z = dz * rstd * gv.z + bev.z; v.w = dw * rstd * gv.w + bev.w;
    }
    if (col_live) *reinterpret_cast<float4*>(out + m * ldo + n0 + c) = v;
  }
}

}  // namespace occ

extern "C" int occ_linear_f32(const float* a1, int64_t lda1, int K1, const float* a2,
                              const float* a2_add, int64_t lda2, int K2, const float* weight,
                              const float* bias, int act, const float* residual, int64_t ldres,
                              const float* ln_gamma, const float* ln_beta, float ln_eps, float* out,
                              int64_t ldo, int M, int N, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(a1 && weight && out, "linear: null pointer argument");
  OCC_CHECK_ARG(M > 0 && N > 0 && K1 > 0 && K2 >= 0, "linear: bad dimension (M=%d N=%d K1=%d K2=%d)", M,
                N, K1, K2);
  OCC_CHECK_ARG((K2 == 0) == (a2 == nullptr), "linear: a2 must be given exactly when K2 > 0");
  OCC_CHECK_ARG(!a2_add || a2, "linear: a2_add without a2");
  OCC_CHECK_ARG(act == 0 || act == 1, "linear: act must be 0 (none) or 1 (ReLU)");
  OCC_CHECK_ARG((ln_gamma == nullptr) == (ln_beta == nullptr), "linear: ln_gamma and ln_beta go together");
  OCC_CHECK_ARG(lda1 >= K1 && (K2 == 0 || lda2 >= K2) && ldo >= N && (!residual || ldres >= N),
                "linear: leading dimension smaller than the row");
  if (K1 % kLinBK || K2 % kLinBK || N % 4 || lda1 % 4 || lda2 % 4 || ldo % 4 || ldres % 4 ||
      (ln_gamma && N > 256)) {
    set_error("linear: no MFMA kernel for K1=%d K2=%d N=%d (need K %% 16 == 0, N %% 4 == 0, 16-byte "
              "aligned rows, N <= 256 with LayerNorm)", K1, K2, N);
    return OCC_E_UNSUPPORTED;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const unsigned gx = (unsigned)((M + kLinBM - 1) / kLinBM);
  if (N <= 128) {
    hipLaunchKernelGGL(linear_mfma_kernel<1>, dim3(gx, (unsigned)((N + 127) / 128)), dim3(256), 0, st,
                       a1, (long)lda1, K1, a2, a2_add, (long)lda2, K2, weight, bias, act, residual,
                       (long)ldres, ln_gamma, ln_beta, ln_eps, out, (long)ldo, M, N);
  } else {
    hipLaunchKernelGGL(linear_mfma_kernel<2>, dim3(gx, (unsigned)((N + 255) / 256)), dim3(256), 0, st,
                       a1, (long)lda1, K1, a2, a2_add, (long)lda2, K2, weight, bias, act, residual,
                       (long)ldres, ln_gamma, ln_beta, ln_eps, out, (long)ldo, M, N);
  }
  OCC_CHECK_LAUNCH("linear");
  return OCC_OK;
}
