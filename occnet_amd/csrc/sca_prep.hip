// Sampling-location / attention-weight preparation of SpatialCrossAttention's TRAINING path, forward and backward,
// one kernel each (gfx950).
//
// Reference (spatial_cross_attention.py:338-373 on the per-camera rebatched queries): attention_weights Linear ->
// softmax over L*P, sampling_offsets Linear / (W_l, H_l) + the z-anchor reference point (point p pairs with anchor
// p % Z) -> sampling_locations; ATen runs that as ~8 elementwise launches over (6 x 9 900 x 768)-sized tensors plus
// ~12 in backward, per layer.  Here the two Linears have already been applied once per BEV query (`proj`, the
// projected-rebatch order of SpatialCrossAttention._unfused_slots), and
//   forward : one wave per padded (camera, row): fetch the query's 768 projections through row_to_query (the
//             rebatch itself), softmax, normalise, add the anchor -> loc (rows, M, L, P, 2), attn (rows, M, L, P)
//             in exactly the layout ms_deform_attn_forward takes;
//   backward: one wave per BEV query: over the <= K rows that hold it (query_to_rows) accumulate
//             d offsets = grad_loc / (W_l, H_l) and d logits = attn * (grad_attn - sum(attn * grad_attn)) -> d proj.
// Both directions are gathers (no atomics) with a fixed summation order.
#include "common.h"

namespace occ {

constexpr int kPrepWaves = 4;

template <int L, int P>
__global__ __launch_bounds__(256) void sca_prep_fwd_kernel(
    const float* __restrict__ proj, long proj_batch_stride, int proj_ld, int n_off,
    const int64_t* __restrict__ row_to_query, const float* __restrict__ ref_rb,
    const int64_t* __restrict__ shapes, float* __restrict__ loc, float* __restrict__ attn, int B, long R, int Z) {
  constexpr int M = 8, LP = L * P, K = M * LP / 64;
  static_assert(LP == 32, "one softmax group = half a wave");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long wg = (long)blockIdx.x * kPrepWaves + wave;
  if (wg >= (long)B * R) return;
  const int b = (int)(wg / R);
  const long r = wg - (long)b * R;
  const long q = row_to_query[r];
  const int s = lane % LP, l = s / P, p = s % P;
  const float Hl = (float)shapes[2 * l], Wl = (float)shapes[2 * l + 1];
  const float2 rxy = *reinterpret_cast<const float2*>(ref_rb + (((long)b * R + r) * Z + (p % Z)) * 2);
  const float* prow = proj + (long)b * proj_batch_stride + (q < 0 ? 0 : q) * (long)proj_ld;
  float* lrow = loc + ((long)b * R + r) * (M * LP * 2);
  float* arow = attn + ((long)b * R + r) * (M * LP);
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int idx = lane + 64 * k;                       // = m * LP + s
    float x = 0.f;
    float2 o = make_float2(0.f, 0.f);
    if (q >= 0) {                                        // wave-uniform; a padded row is the image of zeros
      x = prow[n_off + idx];
      o = *reinterpret_cast<const float2*>(prow + 2 * idx);
    }
    float mx = x;
#pragma unroll
    for (int d = LP / 2; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    const float e = expf(x - mx);
    float sum = e;
#pragma unroll
    for (int d = LP / 2; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
    arow[idx] = fdiv(e, sum);                  // (fdiv, not `/`: see common.h)
    *reinterpret_cast<float2*>(lrow + 2 * idx) = make_float2(rxy.x + fdiv(o.x, Wl), rxy.y + fdiv(o.y, Hl));
  }
}

template <int L, int P>
__global__ __launch_bounds__(256) void sca_prep_bwd_kernel(
    const float* __restrict__ grad_loc, const float* __restrict__ grad_attn, const float* __restrict__ attn,
    const int64_t* __restrict__ query_to_rows, int Kq, const int64_t* __restrict__ shapes,
    float* __restrict__ dproj, int proj_ld, int n_off, int B, long R, long Q) {
  constexpr int M = 8, LP = L * P, K = M * LP / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long wg = (long)blockIdx.x * kPrepWaves + wave;
  if (wg >= (long)B * Q) return;
  const int b = (int)(wg / Q);
  const long q = wg - (long)b * Q;
  const int s = lane % LP, l = s / P;
  const float Hl = (float)shapes[2 * l], Wl = (float)shapes[2 * l + 1];
  float gx[K], gy[K], gl[K];
#pragma unroll
  for (int k = 0; k < K; ++k) gx[k] = gy[k] = gl[k] = 0.f;
  for (int kk = 0; kk < Kq; ++kk) {
    const long r = query_to_rows[q * Kq + kk];
    if (r < 0) continue;                                 // wave-uniform
    const long row = (long)b * R + r;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int idx = lane + 64 * k;
      const float a = attn[row * (M * LP) + idx];
      const float ga = grad_attn[row * (M * LP) + idx];
      float dot = a * ga;
#pragma unroll
      for (int d = LP / 2; d >= 1; d >>= 1) dot += __shfl_xor(dot, d);
      gl[k] += a * (ga - dot);
      const float2 g2 = *reinterpret_cast<const float2*>(grad_loc + (row * (M * LP) + idx) * 2);
      gx[k] += fdiv(g2.x, Wl);
      gy[k] += fdiv(g2.y, Hl);
    }
  }
  float* drow = dproj + ((long)b * Q + q) * proj_ld;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int idx = lane + 64 * k;
    *reinterpret_cast<float2*>(drow + 2 * idx) = make_float2(gx[k], gy[k]);
    drow[n_off + idx] = gl[k];
  }
}

}  // namespace occ

extern "C" int occ_sca_prep_forward_f32(const float* proj, int64_t proj_batch_stride, int proj_ld,
                                        const int64_t* row_to_query, const float* ref_rb,
                                        const int64_t* spatial_shapes, float* loc, float* attn, int B, int64_t R,
                                        int M, int L, int P, int Z, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(proj && row_to_query && ref_rb && spatial_shapes && loc && attn, "sca_prep_forward: null pointer");
  OCC_CHECK_ARG(B > 0 && R > 0 && Z > 0 && P % Z == 0, "sca_prep_forward: bad dimension (B=%d R=%lld Z=%d P=%d)", B,
                (long long)R, Z, P);
  if (M != 8 || L != 4 || P != 8) {
    set_error("sca_prep_forward: no kernel for M=%d L=%d P=%d", M, L, P);
    return OCC_E_UNSUPPORTED;
  }
  OCC_CHECK_ARG(proj_ld >= M * L * P * 3 && proj_ld % 2 == 0, "sca_prep_forward: proj rows shorter than 3*M*L*P");
  const long waves = (long)B * R;
  hipLaunchKernelGGL((sca_prep_fwd_kernel<4, 8>), dim3((unsigned)((waves + kPrepWaves - 1) / kPrepWaves)), dim3(256),
                     0, reinterpret_cast<hipStream_t>(stream), proj, (long)proj_batch_stride, proj_ld, M * L * P * 2,
                     row_to_query, ref_rb, spatial_shapes, loc, attn, B, (long)R, Z);
  OCC_CHECK_LAUNCH("sca_prep_forward");
  return OCC_OK;
}

extern "C" int occ_sca_prep_backward_f32(const float* grad_loc, const float* grad_attn, const float* attn,
                                         const int64_t* query_to_rows, int Kq, const int64_t* spatial_shapes,
                                         float* dproj, int proj_ld, int B, int64_t R, int64_t Q, int M, int L,
                                         int P, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(grad_loc && grad_attn && attn && query_to_rows && spatial_shapes && dproj,
                "sca_prep_backward: null pointer");
  OCC_CHECK_ARG(B > 0 && R > 0 && Q > 0 && Kq > 0, "sca_prep_backward: bad dimension");
  if (M != 8 || L != 4 || P != 8) {
    set_error("sca_prep_backward: no kernel for M=%d L=%d P=%d", M, L, P);
    return OCC_E_UNSUPPORTED;
  }
  OCC_CHECK_ARG(proj_ld >= M * L * P * 3 && proj_ld % 2 == 0, "sca_prep_backward: proj rows shorter than 3*M*L*P");
  const long waves = (long)B * Q;
  hipLaunchKernelGGL((sca_prep_bwd_kernel<4, 8>), dim3((unsigned)((waves + kPrepWaves - 1) / kPrepWaves)), dim3(256),
                     0, reinterpret_cast<hipStream_t>(stream), grad_loc, grad_attn, attn, query_to_rows, Kq,
                     spatial_shapes, dproj, proj_ld, M * L * P * 2, B, (long)R, (long)Q);
  OCC_CHECK_LAUNCH("sca_prep_backward");
  return OCC_OK;
}
