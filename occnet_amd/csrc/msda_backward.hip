// Multi-scale deformable attention backward for gfx950 — the drop-in for mmcv's
// `ms_deform_attn_backward` (call sites: projects/mmdet3d_plugin/bevformer/modules/
// multi_scale_deformable_attn_function.py:74-84,150-160).  Arithmetic of mmcv's
// ms_deformable_col2im, per sample (b,q,m,l,p) with top = grad_out[b,q,m,:]:
//   grad_attn      = sum_c top_c * bilinear_c
//   grad_loc_x     = W * sum_c top_c * attn * (hh*(v2-v1) + lh*(v4-v3))      (only in-range corners)
//   grad_loc_y     = H * sum_c top_c * attn * (hw*(v3-v1) + lw*(v4-v2))
//   grad_value[k] += top_c * attn * w_k                                     (atomic, in-range corners)
// The three grad tensors arrive pre-zeroed (the caller's contract); grad_value is accumulated with
// atomics, the other two are overwritten (each sample has exactly one writer), as mmcv does.
//
// Decomposition (D == 32): 8 lanes x 4 channels per (b,q,m) item, 8 items per wave — the forward's
// layout, so a corner is one 128-byte row per group; the channel sums are 3-step DPP/shuffle
// reductions inside the 8-lane group.  Other D: one thread per (item, channel), mmcv's own shape,
// with the channel sums reduced through LDS.
#include "common.h"

namespace occ {

__device__ __forceinline__ float group8_sum(float v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  return v;
}

__global__ __launch_bounds__(256) void msda_bwd_d32_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lstart, const float* __restrict__ loc,
    const float* __restrict__ attn, const float* __restrict__ grad_out,
    float* __restrict__ grad_value, float* __restrict__ grad_loc, float* __restrict__ grad_attn,
    int S, int M, int L, int Lq, int P, long n_items) {
  constexpr int D = 32;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long item = gid >> 3;
  const int c4 = (int)(gid & 7);
  if (item >= n_items) return;   // whole 8-lane groups leave together; shuffles stay inside a group
  const int m = (int)(item % M);
  const long b = item / ((long)M * Lq);
  const long row_stride = (long)M * D;
  const long voff = b * (long)S * row_stride + (long)m * D + c4 * 4;
  const float* vb = value + voff;
  float* gvb = grad_value + voff;
  const float4 top = *reinterpret_cast<const float4*>(grad_out + item * D + c4 * 4);
  // an item whose 32 output gradients are all zero (the padded rebatch rows of SpatialCrossAttention's
  // autograd path, a quarter of all rows) contributes nothing: the caller pre-zeroed the three grad tensors
  const bool nz = top.x != 0.f || top.y != 0.f || top.z != 0.f || top.w != 0.f;
  const unsigned long long any = __ballot(nz);
  if (((any >> (threadIdx.x & 56)) & 0xffull) == 0ull) return;
  const int LP = L * P;
  for (int s = 0; s < LP; ++s) {
    const int l = s / P;
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const long st = lstart[l];
    const long si = item * LP + s;
    const float2 xy = *reinterpret_cast<const float2*>(loc + si * 2);
    const float a = attn[si];
    const float h_im = xy.y * (float)H - 0.5f;
    const float w_im = xy.x * (float)W - 0.5f;
    float g_attn = 0.f, g_x = 0.f, g_y = 0.f;
    if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {   // group-uniform
      const float hf = floorf(h_im), wf = floorf(w_im);
      const int h_low = (int)hf, w_low = (int)wf, h_high = h_low + 1, w_high = w_low + 1;
      const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
      const bool t_ok = h_low >= 0, b_ok = h_high <= H - 1, l_ok = w_low >= 0, r_ok = w_high <= W - 1;
      const long base = (st + (long)h_low * W + w_low) * row_stride;
      float4 v1 = make_float4(0.f, 0.f, 0.f, 0.f), v2 = v1, v3 = v1, v4 = v1;
      if (t_ok && l_ok) v1 = *reinterpret_cast<const float4*>(vb + base);
      if (t_ok && r_ok) v2 = *reinterpret_cast<const float4*>(vb + base + row_stride);
      if (b_ok && l_ok) v3 = *reinterpret_cast<const float4*>(vb + base + (long)W * row_stride);
      if (b_ok && r_ok) v4 = *reinterpret_cast<const float4*>(vb + base + (long)(W + 1) * row_stride);
      const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
      const float tx[4] = {top.x, top.y, top.z, top.w};
      const float a1[4] = {v1.x, v1.y, v1.z, v1.w}, a2[4] = {v2.x, v2.y, v2.z, v2.w};
      const float a3[4] = {v3.x, v3.y, v3.z, v3.w}, a4[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float ta = tx[k] * a;
        g_attn += tx[k] * (w1 * a1[k] + w2 * a2[k] + w3 * a3[k] + w4 * a4[k]);
        g_x += ta * (hh * (a2[k] - a1[k]) + lh * (a4[k] - a3[k]));
        g_y += ta * (hw * (a3[k] - a1[k]) + lw * (a4[k] - a2[k]));
        if (t_ok && l_ok) unsafeAtomicAdd(gvb + base + k, ta * w1);
        if (t_ok && r_ok) unsafeAtomicAdd(gvb + base + row_stride + k, ta * w2);
        if (b_ok && l_ok) unsafeAtomicAdd(gvb + base + (long)W * row_stride + k, ta * w3);
        if (b_ok && r_ok) unsafeAtomicAdd(gvb + base + (long)(W + 1) * row_stride + k, ta * w4);
      }
      g_x *= (float)W;
      g_y *= (float)H;
    }
    g_attn = group8_sum(g_attn);
    g_x = group8_sum(g_x);
    g_y = group8_sum(g_y);
    if (c4 == 0) {
      grad_attn[si] = g_attn;
      grad_loc[si * 2] = g_x;
      grad_loc[si * 2 + 1] = g_y;
    }
  }
}

// Generic D: one thread per (item, channel); reductions over the item's D threads through LDS.
__global__ __launch_bounds__(256) void msda_bwd_generic_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lstart, const float* __restrict__ loc,
    const float* __restrict__ attn, const float* __restrict__ grad_out,
    float* __restrict__ grad_value, float* __restrict__ grad_loc, float* __restrict__ grad_attn,
    int S, int M, int D, int L, int Lq, int P, long n_items) {
  // block = (256 / D) items x D channels (D divides 256) or one item per block with D <= 256 threads
  extern __shared__ float red[];   // 3 * blockDim.x
  const int per_block = blockDim.x / D;
  const int li = threadIdx.x / D, c = threadIdx.x % D;
  const long item = (long)blockIdx.x * per_block + li;
  const bool live = item < n_items && li < per_block;
  const int LP = L * P;
  const long row_stride = (long)M * D;
  long voff = 0;
  float top = 0.f;
  if (live) {
    const int m = (int)(item % M);
    const long b = item / ((long)M * Lq);
    voff = b * (long)S * row_stride + (long)m * D + c;
    top = grad_out[item * D + c];
  }
  for (int s = 0; s < LP; ++s) {
    float g_attn = 0.f, g_x = 0.f, g_y = 0.f;
    if (live) {
      const int l = s / P;
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const long st = lstart[l];
      const long si = item * LP + s;
      const float loc_w = loc[si * 2], loc_h = loc[si * 2 + 1], a = attn[si];
      const float h_im = loc_h * (float)H - 0.5f, w_im = loc_w * (float)W - 0.5f;
      if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
        const float hf = floorf(h_im), wf = floorf(w_im);
        const int h_low = (int)hf, w_low = (int)wf, h_high = h_low + 1, w_high = w_low + 1;
        const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
        const bool t_ok = h_low >= 0, b_ok = h_high <= H - 1, l_ok = w_low >= 0, r_ok = w_high <= W - 1;
        const long base = voff + (st + (long)h_low * W + w_low) * row_stride;
        float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
        const float ta = top * a;
        if (t_ok && l_ok) { v1 = value[base]; unsafeAtomicAdd(grad_value + base, ta * hh * hw); }
        if (t_ok && r_ok) { v2 = value[base + row_stride];
                            unsafeAtomicAdd(grad_value + base + row_stride, ta * hh * lw); }
        if (b_ok && l_ok) { v3 = value[base + (long)W * row_stride];
                            unsafeAtomicAdd(grad_value + base + (long)W * row_stride, ta * lh * hw); }
        if (b_ok && r_ok) { v4 = value[base + (long)(W + 1) * row_stride];
                            unsafeAtomicAdd(grad_value + base + (long)(W + 1) * row_stride, ta * lh * lw); }
        g_attn = top * (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4);
        g_x = ta * (hh * (v2 - v1) + lh * (v4 - v3)) * (float)W;
        g_y = ta * (hw * (v3 - v1) + lw * (v4 - v2)) * (float)H;
      }
    }
    red[threadIdx.x] = g_attn;
    red[blockDim.x + threadIdx.x] = g_x;
    red[2 * blockDim.x + threadIdx.x] = g_y;
    __syncthreads();
    if (live && c == 0) {
      float sa = 0.f, sx = 0.f, sy = 0.f;
      for (int k = 0; k < D; ++k) {
        sa += red[li * D + k];
        sx += red[blockDim.x + li * D + k];
        sy += red[2 * blockDim.x + li * D + k];
      }
      const long si = item * LP + s;
      grad_attn[si] = sa;
      grad_loc[si * 2] = sx;
      grad_loc[si * 2 + 1] = sy;
    }
    __syncthreads();
  }
}

}  // namespace occ

extern "C" int occ_ms_deform_attn_backward_f32(
    const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
    const float* sampling_loc, const float* attn_weight, const float* grad_output, float* grad_value,
    float* grad_sampling_loc, float* grad_attn_weight, int B, int S, int M, int D, int L, int Lq,
    int P, int im2col_step, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(value && spatial_shapes && level_start_index && sampling_loc && attn_weight &&
                    grad_output && grad_value && grad_sampling_loc && grad_attn_weight,
                "ms_deform_attn_backward: null pointer argument");
  OCC_CHECK_ARG(B > 0 && S > 0 && M > 0 && D > 0 && L > 0 && Lq > 0 && P > 0,
                "ms_deform_attn_backward: non-positive dimension");
  OCC_CHECK_ARG(im2col_step > 0, "ms_deform_attn_backward: im2col_step must be positive");
  const int step = B < im2col_step ? B : im2col_step;
  OCC_CHECK_ARG(B % step == 0, "ms_deform_attn_backward: batch(%d) must divide im2col_step(%d)", B,
                step);
  OCC_CHECK_ARG(D <= 256, "ms_deform_attn_backward: channels per head (%d) above 256", D);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long n_items = (long)B * Lq * M;
  if (D == 32) {
    const long threads = n_items * 8;
    hipLaunchKernelGGL(msda_bwd_d32_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st,
                       value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                       grad_output, grad_value, grad_sampling_loc, grad_attn_weight, S, M, L, Lq, P,
                       n_items);
  } else {
    const int per_block = 256 / D > 0 ? 256 / D : 1;
    const int threads = per_block * D;
    const long blocks = (n_items + per_block - 1) / per_block;
    hipLaunchKernelGGL(msda_bwd_generic_kernel, dim3((unsigned)blocks), dim3(threads),
                       3 * threads * sizeof(float), st, value, spatial_shapes, level_start_index,
                       sampling_loc, attn_weight, grad_output, grad_value, grad_sampling_loc,
                       grad_attn_weight, S, M, D, L, Lq, P, n_items);
  }
  OCC_CHECK_LAUNCH("ms_deform_attn_backward");
  return OCC_OK;
}
