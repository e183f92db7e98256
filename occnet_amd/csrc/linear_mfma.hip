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
// owns columns [w*32*NT, (w+1)*32*NT)).  Per K chunk of 32: A tile (32 x 32) and W tile (BN x 32) are
// staged row-major into LDS with a 36-float row stride (144 B: the 32 rows a ds_read_b128 touches land
// on distinct 16-byte bank slots).  k-pairing as in the Conv3d kernel: lanes 0-31 contract the chunk's
// first 16 k, lanes 32-63 the last 16, so a lane's fragment is 16 contiguous floats of one row.
// Epilogue: the accumulators are transposed through LDS into row-major; every wave then owns 8 rows,
// 4 consecutive columns per lane (coalesced float4 traffic for bias / residual / output) and the
// LayerNorm statistics are two wave reductions per row (two-pass variance, as ATen's kernel).
#include "common.h"

namespace occ {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kLinBM = 32, kLinBK = 32, kLinLD = 36;

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
  constexpr int BN = 128 * NT, LD = kLinLD, OLD = BN + 4;
  constexpr int STAGE_FLOATS = (kLinBM + BN) * LD, OUT_FLOATS = kLinBM * OLD;
  __shared__ __attribute__((aligned(16))) float lds[STAGE_FLOATS > OUT_FLOATS ? STAGE_FLOATS : OUT_FLOATS];
  float* sA = lds;
  float* sW = lds + kLinBM * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vi = lane & 31, kh = lane >> 5;
  const long m0 = (long)blockIdx.x * kLinBM;
  const int n0 = blockIdx.y * BN;
  const int K = K1 + K2;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // staging roles: thread -> (row, 16-byte part); A: 1 float4 per thread, W: BN/32 float4 per thread
  const int srow = tid >> 3, spart = tid & 7;
  for (int k0 = 0; k0 < K; k0 += kLinBK) {
    float4 va = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m0 + srow < M) {
      if (k0 < K1) {
        va = *reinterpret_cast<const float4*>(a1 + (m0 + srow) * lda1 + k0 + spart * 4);
      } else {
        const long o = (m0 + srow) * lda2 + (k0 - K1) + spart * 4;
        va = *reinterpret_cast<const float4*>(a2 + o);
        if (a2add) {
          const float4 vb = *reinterpret_cast<const float4*>(a2add + o);
          va.x += vb.x; va.y += vb.y; va.z += vb.z; va.w += vb.w;
        }
      }
    }
    float4 vw[BN / 32];
#pragma unroll
    for (int it = 0; it < BN / 32; ++it) {
      const int n = n0 + srow + 32 * it;
      vw[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < N) vw[it] = *reinterpret_cast<const float4*>(w + (long)n * K + k0 + spart * 4);
    }
    if (k0) __syncthreads();   // previous chunk's fragments have been read
    *reinterpret_cast<float4*>(sA + srow * LD + spart * 4) = va;
#pragma unroll
    for (int it = 0; it < BN / 32; ++it)
      *reinterpret_cast<float4*>(sW + (srow + 32 * it) * LD + spart * 4) = vw[it];
    __syncthreads();

    float af[16], bf[NT][16];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const float4 x = *reinterpret_cast<const float4*>(sA + vi * LD + kh * 16 + s4 * 4);
      af[s4 * 4 + 0] = x.x; af[s4 * 4 + 1] = x.y; af[s4 * 4 + 2] = x.z; af[s4 * 4 + 3] = x.w;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const float4 x = *reinterpret_cast<const float4*>(sW + ((wave * NT + t) * 32 + vi) * LD +
                                                          kh * 16 + s4 * 4);
        bf[t][s4 * 4 + 0] = x.x; bf[t][s4 * 4 + 1] = x.y; bf[t][s4 * 4 + 2] = x.z; bf[t][s4 * 4 + 3] = x.w;
      }
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[t][s], acc[t], 0, 0, 0);
  }

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
  const float inv_n = 1.f / (float)N;
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
      if (residual) {
        const float4 rv = *reinterpret_cast<const float4*>(residual + m * ldres + n0 + c);
        v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
      }
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
    set_error("linear: no MFMA kernel for K1=%d K2=%d N=%d (need K %% 32 == 0, N %% 4 == 0, 16-byte "
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
