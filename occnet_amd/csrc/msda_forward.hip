// Multi-scale deformable attention forward for gfx950 — the drop-in for mmcv's
// `ms_deform_attn_forward` (call sites: projects/mmdet3d_plugin/bevformer/modules/
// multi_scale_deformable_attn_function.py:42-48,118-124).
//
// Layout: value (B, S, M, D) keeps one pixel's M*D floats contiguous, so a bilinear corner of one
// head is one 4*D-byte row (128 B at D=32).  Fast kernel (D == 32): a 64-lane wave owns 8
// consecutive (b,q,m) items; the 8 lanes of a group hold 4 channels each (one 16-byte load per
// lane = one full 128-byte corner row per group, 8 rows = 1 KiB per wave instruction).
//   phase 1: the wave resolves all 8*L*P samples of its items once (4 corner weights * attention
//            weight, 4 element offsets) and parks them in LDS (32 B per sample);
//   phase 2: each group walks its item's samples: 2 ds_read_b128 + 4 global_load_dwordx4 + 16 FMA
//            per sample, 4 samples (16 loads) in flight per lane; no cross-lane reduction, the
//            wave writes 8 x 128 B of output.
// Any other D uses the scalar kernel (one thread per output element, mmcv's own decomposition).
#include "common.h"

namespace occ {

constexpr int kWavesPerBlock = 4;

// BUF: corner rows through buffer loads over the WHOLE value tensor (it must be smaller than kOobOffset bytes): a corner
// outside its map carries an out-of-range offset and returns 0 without a memory request — no dummy load of row 0, so
// a NaN / Inf there cannot leak into border samples (0 * Inf).  Larger tensors keep the pointer form.
template <bool BUF>
__global__ __launch_bounds__(256) void msda_fwd_d32_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lstart, const float* __restrict__ loc,
    const float* __restrict__ attn, float* __restrict__ out, int S, int M, int L, int Lq, int P,
    long n_items, unsigned value_bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int D = 32;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int LP = L * P;
  const int LPp = LP + 1;  // +32 B per item: neighbouring groups land on different LDS banks
  static_assert(sizeof(SampleParam) == sizeof(SampleParamB), "one LDS layout for both forms");
  SampleParam* sp = reinterpret_cast<SampleParam*>(smem) + (size_t)wave * 8 * LPp;
  SampleParamB* spb = reinterpret_cast<SampleParamB*>(smem) + (size_t)wave * 8 * LPp;
  const long item0 = ((long)blockIdx.x * kWavesPerBlock + wave) * 8;
  if (item0 >= n_items) return;
  const int row_stride = M * D;

  for (int i = lane; i < 8 * LP; i += 64) {
    const int g = i / LP, s = i - g * LP;
    const long item = item0 + g;
    const bool live = item < n_items;
    const int l = s / P;
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const int st = (int)lstart[l];
    const long si = (live ? item : n_items - 1) * LP + s;
    const float2 xy = *reinterpret_cast<const float2*>(loc + si * 2);
    const float a = attn[si];
    if (BUF) {
      SampleParamB p;
      bilinear_setup_b(xy.x, xy.y, a, H, W, st, (unsigned)row_stride * 4u, kOobOffset, lane_flag(live), p);
      spb[g * LPp + s] = p;
    } else {
      SampleParam p;
      bilinear_setup(xy.x, xy.y, a, H, W, st, row_stride, lane_flag(live), p);
      sp[g * LPp + s] = p;
    }
  }
  wave_lds_sync();

  const int g = lane >> 3, c4 = lane & 7;
  const long item = item0 + g;
  if (item < n_items) {
    const long m = item % M;
    const long b = item / ((long)M * Lq);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (BUF) {
      const __amdgpu_buffer_rsrc_t rsrc = uniform_rsrc(value, value_bytes);
      const unsigned lane_off = (unsigned)((b * (long)S * row_stride + m * D + c4 * 4) * 4);
      acc = gather_samples_buf<4>(rsrc, lane_off, spb + g * LPp, LP, acc);
    } else {
      const float* vb = value + b * (long)S * row_stride + m * D + c4 * 4;
      acc = gather_samples<4>(vb, sp + g * LPp, LP, acc);
    }
    *reinterpret_cast<float4*>(out + item * D + c4 * 4) = acc;
  }
}

// Scalar fallback: one thread per output element (b, q, m, c); any D / L / P.
__global__ __launch_bounds__(256) void msda_fwd_scalar_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lstart, const float* __restrict__ loc,
    const float* __restrict__ attn, float* __restrict__ out, int S, int M, int D, int L, int Lq,
    int P, long n_out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_out) return;
  const int c = (int)(idx % D);
  const long item = idx / D;  // (b*Lq + q)*M + m
  const int m = (int)(item % M);
  const long b = item / ((long)M * Lq);
  const long row_stride = (long)M * D;
  const float* vb = value + b * S * row_stride + (long)m * D + c;
  float col = 0.f;
  for (int l = 0; l < L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const long st = lstart[l];
    for (int p = 0; p < P; ++p) {
      const long si = (item * L + l) * P + p;
      const float loc_w = loc[si * 2], loc_h = loc[si * 2 + 1];
      const float weight = attn[si];
      const BilinearTerms t = bilinear_terms(loc_w, loc_h, H, W, 1);
      // unconditional loads: a corner outside the map (or of a sample that is not admitted) reads element 0 of the batch entry
      // and is replaced by 0 (select, not multiply: 0 * Inf)
      const long base = (st + (long)t.h_low * W + t.w_low) * row_stride;
      const float r1 = vb[t.c[0] ? base : 0], r2 = vb[t.c[1] ? base + row_stride : 0];
      const float r3 = vb[t.c[2] ? base + (long)W * row_stride : 0], r4 = vb[t.c[3] ? base + (long)(W + 1) * row_stride : 0];
      const float v1 = t.c[0] ? r1 : 0.f, v2 = t.c[1] ? r2 : 0.f, v3 = t.c[2] ? r3 : 0.f, v4 = t.c[3] ? r4 : 0.f;
      const float val = t.hh * t.hw * v1 + t.hh * t.lw * v2 + t.lh * t.hw * v3 + t.lh * t.lw * v4;
      col += t.adm ? val * weight : 0.f;
    }
  }
  out[idx] = col;
}

}  // namespace occ

extern "C" int occ_ms_deform_attn_forward_f32(const float* value, const int64_t* spatial_shapes,
                                              const int64_t* level_start_index,
                                              const float* sampling_loc, const float* attn_weight,
                                              float* out, int B, int S, int M, int D, int L, int Lq,
                                              int P, int im2col_step, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(value && spatial_shapes && level_start_index && sampling_loc && attn_weight && out,
                "ms_deform_attn_forward: null pointer argument");
  OCC_CHECK_ARG(B > 0 && S > 0 && M > 0 && D > 0 && L > 0 && Lq > 0 && P > 0,
                "ms_deform_attn_forward: non-positive dimension (B=%d S=%d M=%d D=%d L=%d Lq=%d P=%d)",
                B, S, M, D, L, Lq, P);
  OCC_CHECK_ARG(im2col_step > 0, "ms_deform_attn_forward: im2col_step must be positive");
  const int step = B < im2col_step ? B : im2col_step;
  OCC_CHECK_ARG(B % step == 0, "ms_deform_attn_forward: batch(%d) must divide im2col_step(%d)", B,
                step);
  OCC_CHECK_ARG((long)S * M * D < (1L << 31),
                "ms_deform_attn_forward: one batch entry of value exceeds 2^31 elements");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long n_items = (long)B * Lq * M;
  const int LP = L * P;
  const size_t lds = (size_t)kWavesPerBlock * 8 * (LP + 1) * sizeof(SampleParam);
  if (D == 32 && lds <= 64 * 1024) {
    const long waves = (n_items + 7) / 8;
    const long blocks = (waves + kWavesPerBlock - 1) / kWavesPerBlock;
    const long value_bytes = (long)B * S * M * D * 4;
    if (value_bytes < (long)kOobOffset)
      hipLaunchKernelGGL(msda_fwd_d32_kernel<true>, dim3((unsigned)blocks), dim3(256), lds, st, value,
                         spatial_shapes, level_start_index, sampling_loc, attn_weight, out, S, M, L,
                         Lq, P, n_items, (unsigned)value_bytes);
    else
      hipLaunchKernelGGL(msda_fwd_d32_kernel<false>, dim3((unsigned)blocks), dim3(256), lds, st, value,
                         spatial_shapes, level_start_index, sampling_loc, attn_weight, out, S, M, L,
                         Lq, P, n_items, 0u);
  } else {
    const long n_out = n_items * D;
    const long blocks = (n_out + 255) / 256;
    hipLaunchKernelGGL(msda_fwd_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, st, value,
                       spatial_shapes, level_start_index, sampling_loc, attn_weight, out, S, M, D,
                       L, Lq, P, n_out);
  }
  OCC_CHECK_LAUNCH("ms_deform_attn_forward");
  return OCC_OK;
}
