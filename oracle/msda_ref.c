/* Plain-C restatement of the deformable-attention forward arithmetic — TEST INFRASTRUCTURE
 * (oracle; see oracle/__init__.py).  Follows the per-sample arithmetic of mmcv-full 1.4.x
 * ms_deformable_im2col, the operator the reference binds at
 * projects/mmdet3d_plugin/bevformer/modules/multi_scale_deformable_attn_function.py:10-12,118-124.
 * One (b,q,m) row at a time, fp32, OpenMP over rows; used as a second oracle for large shapes and
 * as a multi-core CPU baseline.  Build: make -C oracle
 */
#include <math.h>
#include <stdint.h>

void msda_forward_ref_f32(const float* value, const int64_t* shapes, const int64_t* lstart,
                          const float* loc, const float* attn, float* out, int B, int S, int M,
                          int D, int L, int Lq, int P) {
  const long n_items = (long)B * Lq * M;
  const long row = (long)M * D;
#pragma omp parallel for schedule(static)
  for (long item = 0; item < n_items; ++item) {
    const int m = (int)(item % M);
    const long b = item / ((long)M * Lq);
    const float* vb = value + b * S * row + (long)m * D;
    float* o = out + item * D;
    for (int c = 0; c < D; ++c) o[c] = 0.f;
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const long st = lstart[l];
      for (int p = 0; p < P; ++p) {
        const long si = (item * L + l) * P + p;
        const float loc_w = loc[2 * si], loc_h = loc[2 * si + 1], weight = attn[si];
        const float h_im = loc_h * H - 0.5f, w_im = loc_w * W - 0.5f;
        if (!(h_im > -1 && w_im > -1 && h_im < H && w_im < W)) continue;
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const int h_high = h_low + 1, w_high = w_low + 1;
        const float lh = h_im - h_low, lw = w_im - w_low, hh = 1 - lh, hw = 1 - lw;
        const float* p1 = (h_low >= 0 && w_low >= 0) ? vb + (st + (long)h_low * W + w_low) * row : 0;
        const float* p2 = (h_low >= 0 && w_high <= W - 1) ? vb + (st + (long)h_low * W + w_high) * row : 0;
        const float* p3 = (h_high <= H - 1 && w_low >= 0) ? vb + (st + (long)h_high * W + w_low) * row : 0;
        const float* p4 = (h_high <= H - 1 && w_high <= W - 1) ? vb + (st + (long)h_high * W + w_high) * row : 0;
        const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
        for (int c = 0; c < D; ++c) {
          const float val = w1 * (p1 ? p1[c] : 0.f) + w2 * (p2 ? p2[c] : 0.f) +
                            w3 * (p3 ? p3[c] : 0.f) + w4 * (p4 ? p4[c] : 0.f);
          o[c] += val * weight;
        }
      }
    }
  }
}
