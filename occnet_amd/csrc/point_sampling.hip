// Pillar reference points -> camera pixels + visibility, one thread per (camera, batch, query, z).
//
// Restates BEVFormerEncoder.point_sampling (reference: projects/mmdet3d_plugin/bevformer/modules/
// encoder.py:92-151) without materialising the (Z,B,NC,Nq,4,4) repeated matrices (:120-126):
//   p_metres = ref*(range) + min (:107-112); p_cam = (lidar2img @ ego2lidar) @ [p,1] (:126);
//   mask = z > 1e-5 (:129); uv = xy / max(z,1e-5) (:130-131); uv /= (img_w, img_h) (:133-134);
//   mask &= 0<v<1 & 0<u<1 (:136-139); outputs permuted to (NC,B,Nq,Z,*) (:146-149).
// fp32 throughout ("This function must use fp32!!!", :91).  Also emits one visibility word per
// (batch, query): bit c = any z-anchor visible in camera c — the information
// SpatialCrossAttention.forward rebuilds with nonzero() per layer (spatial_cross_attention.py:136-141,169-171).
#include "common.h"

namespace occ {

__global__ __launch_bounds__(256) void point_sampling_kernel(
    const float* __restrict__ ref_3d, const float* __restrict__ lidar2img,
    const float* __restrict__ ego2lidar, float rx, float ry, float rz, float x0, float y0, float z0,
    float img_h, float img_w, float* __restrict__ ref_cam, uint8_t* __restrict__ bev_mask,
    uint32_t* __restrict__ vis_bits, int B, int NC, int Nq, int Z) {
  // thread -> (b, q); loops cameras and anchors so the visibility word needs no atomics
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * Nq) return;
  const int b = (int)(idx / Nq), q = (int)(idx % Nq);
  uint32_t bits = 0;
  for (int c = 0; c < NC; ++c) {
    // T = lidar2img[b,c] @ ego2lidar  (k-ordered fma chain per element)
    const float* A = lidar2img + ((long)b * NC + c) * 16;
    float T[3][4];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = __fmul_rn(A[i * 4 + 0], ego2lidar[0 * 4 + j]);
        s = fmaf(A[i * 4 + 1], ego2lidar[1 * 4 + j], s);
        s = fmaf(A[i * 4 + 2], ego2lidar[2 * 4 + j], s);
        s = fmaf(A[i * 4 + 3], ego2lidar[3 * 4 + j], s);
        T[i][j] = s;
      }
    int any = 0;                                     // lane predicates as 0 / 1 VGPR integers (common.h: lane_flag)
    for (int z = 0; z < Z; ++z) {
      const float* p = ref_3d + (((long)b * Z + z) * Nq + q) * 3;
      const float X = __fadd_rn(__fmul_rn(p[0], rx), x0);
      const float Y = __fadd_rn(__fmul_rn(p[1], ry), y0);
      const float Zm = __fadd_rn(__fmul_rn(p[2], rz), z0);
      float cam[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        float s = __fmul_rn(T[i][0], X);
        s = fmaf(T[i][1], Y, s);
        s = fmaf(T[i][2], Zm, s);
        s = fmaf(T[i][3], 1.0f, s);
        cam[i] = s;
      }
      const float eps = 1e-5f;
      int m = lane_flag(cam[2] > eps);
      const float den = fmaxf(cam[2], eps);
      float u = cam[0] / den, v = cam[1] / den;
      u = u / img_w;
      v = v / img_h;
      m = m & lane_flag(v > 0.0f) & lane_flag(v < 1.0f) & lane_flag(u < 1.0f) & lane_flag(u > 0.0f);
      const long o = (((long)c * B + b) * Nq + q) * Z + z;
      ref_cam[o * 2 + 0] = u;
      ref_cam[o * 2 + 1] = v;
      bev_mask[o] = (uint8_t)m;
      any |= m;
    }
    bits |= (uint32_t)any << c;
  }
  if (vis_bits) vis_bits[idx] = bits;
}

}  // namespace occ

extern "C" int occ_point_sampling_f32(const float* ref_3d, const float* lidar2img,
                                      const float* ego2lidar, const float* pc_range, float img_h,
                                      float img_w, float* ref_cam, uint8_t* bev_mask,
                                      uint32_t* vis_bits, int B, int NC, int Nq, int Z,
                                      void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(ref_3d && lidar2img && ego2lidar && pc_range && ref_cam && bev_mask,
                "point_sampling: null pointer argument");
  OCC_CHECK_ARG(B > 0 && NC > 0 && NC <= 32 && Nq > 0 && Z > 0,
                "point_sampling: bad dimension (B=%d NC=%d Nq=%d Z=%d)", B, NC, Nq, Z);
  OCC_CHECK_ARG(img_h > 0 && img_w > 0, "point_sampling: bad image shape");
  // python-float (double) range arithmetic rounded once to f32, as `tensor * (pc[3]-pc[0]) + pc[0]`
  const float rx = (float)((double)pc_range[3] - (double)pc_range[0]);
  const float ry = (float)((double)pc_range[4] - (double)pc_range[1]);
  const float rz = (float)((double)pc_range[5] - (double)pc_range[2]);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long n = (long)B * Nq;
  hipLaunchKernelGGL(point_sampling_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                     ref_3d, lidar2img, ego2lidar, rx, ry, rz, pc_range[0], pc_range[1],
                     pc_range[2], img_h, img_w, ref_cam, bev_mask, vis_bits, B, NC, Nq, Z);
  OCC_CHECK_LAUNCH("point_sampling");
  return OCC_OK;
}
