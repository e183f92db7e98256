// Ray casting through an occupancy grid for the RayIoU metric — gfx950 version of the reference's only
// native component, `dvr.render_forward` (tools/ray_iou/lib/dvr/dvr.cu:70-319 kernel, :329-388 host
// wrapper; called from projects/mmdet3d_plugin/datasets/ray_metrics.py:116-123 with phase "test").
//
// Per ray: Amanatides-Woo voxel traversal from the origin toward the end point (double precision, the
// reference computes in double on float inputs) and report
//   pred_dist   = ray parameter at which the ray LEAVES the first traversed in-grid voxel whose occupancy
//                 is > 0.5 — or, if none is, the last in-grid voxel it traversed,
//   gt_dist     = |end - origin|  ("test" phase: not clamped; "train": min(gt, exit distance)),
//   coord_index = (x, y, z) of that voxel,
// leaving (-1, -1, (0,0,0)) when the ray never enters the grid.  The reference first records the whole
// path into per-thread arrays (int3 path[1446] + 3 x double[1446] = 52 KB of scratch per thread) and
// then scans it; only the first occupied voxel or the last visited one can be reported, so this kernel
// keeps two registers instead and stops at the first occupied voxel.  The iteration budget is the
// reference's (MAX_STEP = 1000 -> at most 1001 traversal steps).  Floating-point contraction is OFF so
// the arithmetic is the plain IEEE double sequence of the C oracle (oracle/dvr_ref.c) — bit-exact.
#include <float.h>
#include "common.h"

namespace occ {

#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void dvr_render_forward_kernel(
    const float* __restrict__ sigma, const float* __restrict__ origin,
    const float* __restrict__ points, const float* __restrict__ tindex,
    float* __restrict__ pred_dist, float* __restrict__ gt_dist, float* __restrict__ coord_index, int T,
    int vzsize, int vysize, int vxsize, int M, int pstride, int train_phase) {
#pragma clang fp contract(off)
  const int n = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= M) return;
  const float tf = tindex[(long)n * M + c];
  if (tf < 0.f) return;                       // padded ray
  const int t = (int)tf;
  const int ts = (T == 1) ? 0 : t;
  const float* o = origin + ((long)n * T + t) * 3;       // origin is (N, T, 3); t indexes it
  const float* e = points + ((long)n * M + c) * pstride;
  const double xo = o[0], yo = o[1], zo = o[2];
  const double xe = e[0], ye = e[1], ze = e[2];
  int vx = (int)xo, vy = (int)yo, vz = (int)zo;
  const double rx = xe - xo, ry = ye - yo, rz = ze - zo;
  double gt_d = sqrt(rx * rx + ry * ry + rz * rz);
  const double dx = rx / gt_d, dy = ry / gt_d, dz = rz / gt_d;
  const int stepX = (dx >= 0) ? 1 : -1, stepY = (dy >= 0) ? 1 : -1, stepZ = (dz >= 0) ? 1 : -1;
  const double bx = vx + (stepX < 0 ? 0 : 1), by = vy + (stepY < 0 ? 0 : 1),
               bz = vz + (stepZ < 0 ? 0 : 1);
  double tMaxX = (dx != 0) ? (bx - xo) / dx : DBL_MAX;
  double tMaxY = (dy != 0) ? (by - yo) / dy : DBL_MAX;
  double tMaxZ = (dz != 0) ? (bz - zo) / dz : DBL_MAX;
  const double tDeltaX = (dx != 0) ? stepX / dx : DBL_MAX;
  const double tDeltaY = (dy != 0) ? stepY / dy : DBL_MAX;
  const double tDeltaZ = (dz != 0) ? stepZ / dz : DBL_MAX;
  const float* grid = sigma + ((long)n * T + ts) * vzsize * vysize * vxsize;

  bool was_inside = false, hit = false;
  double last_d = 0.0, hit_d = 0.0;           // exit distance of the last in-grid voxel / of the hit voxel
  int lx = 0, ly = 0, lz = 0, hx = 0, hy = 0, hz = 0;
  for (int step = 0; step <= 1000; ++step) {  // MAX_STEP = 1000: the reference runs steps 0..1000
    const bool inside = (0 <= vx && vx < vxsize) && (0 <= vy && vy < vysize) && (0 <= vz && vz < vzsize);
    if (!inside && was_inside) break;         // left the grid: never comes back
    const int cx = vx, cy = vy, cz = vz;
    double d;
    if (tMaxX < tMaxY) {
      if (tMaxX < tMaxZ) { d = tMaxX; vx += stepX; tMaxX += tDeltaX; }
      else { d = tMaxZ; vz += stepZ; tMaxZ += tDeltaZ; }
    } else {
      if (tMaxY < tMaxZ) { d = tMaxY; vy += stepY; tMaxY += tDeltaY; }
      else { d = tMaxZ; vz += stepZ; tMaxZ += tDeltaZ; }
    }
    if (inside) {
      was_inside = true;
      last_d = d; lx = cx; ly = cy; lz = cz;
      if (!hit && grid[((long)cz * vysize + cy) * vxsize + cx] > 0.5f) {
        hit = true; hit_d = d; hx = cx; hy = cy; hz = cz;
        if (!train_phase) break;              // "test": nothing after the first occupied voxel matters
      }
    }
  }
  if (!was_inside) return;                    // outputs keep their -1 / 0 initialisation
  if (train_phase) gt_d = gt_d < last_d ? gt_d : last_d;   // clamp to the grid exit distance (:296-298)
  if (hit) { last_d = hit_d; lx = hx; ly = hy; lz = hz; }
  const long oi = (long)n * M + c;
  pred_dist[oi] = (float)last_d;
  gt_dist[oi] = (float)gt_d;
  coord_index[oi * 3 + 0] = (float)lx;
  coord_index[oi * 3 + 1] = (float)ly;
  coord_index[oi * 3 + 2] = (float)lz;
}

__global__ void dvr_fill_kernel(float* __restrict__ p, float v, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

}  // namespace occ

extern "C" int occ_dvr_render_forward_f32(const float* sigma, const float* origin, const float* points,
                                          const float* tindex, float* pred_dist, float* gt_dist,
                                          float* coord_index, int N, int T, int Z, int Y, int X, int M,
                                          int point_stride, int train_phase, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(sigma && origin && points && tindex && pred_dist && gt_dist && coord_index,
                "dvr_render_forward: null pointer argument");
  OCC_CHECK_ARG(N > 0 && T > 0 && Z > 0 && Y > 0 && X > 0 && M > 0,
                "dvr_render_forward: bad dimension (N=%d T=%d Z=%d Y=%d X=%d M=%d)", N, T, Z, Y, X, M);
  OCC_CHECK_ARG(point_stride >= 3, "dvr_render_forward: points need at least 3 coordinates per ray");
  OCC_CHECK_ARG(train_phase == 0 || train_phase == 1, "dvr_render_forward: phase must be 0 (test) or 1 (train)");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long nm = (long)N * M;
  // reference host wrapper: pred/gt = -ones, coord_index = zeros (dvr.cu:353-356)
  hipLaunchKernelGGL(dvr_fill_kernel, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, st, pred_dist, -1.f, nm);
  hipLaunchKernelGGL(dvr_fill_kernel, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, st, gt_dist, -1.f, nm);
  hipLaunchKernelGGL(dvr_fill_kernel, dim3((unsigned)((3 * nm + 255) / 256)), dim3(256), 0, st, coord_index, 0.f, 3 * nm);
  hipLaunchKernelGGL(dvr_render_forward_kernel, dim3((unsigned)((M + 255) / 256), (unsigned)N), dim3(256), 0,
                     st, sigma, origin, points, tindex, pred_dist, gt_dist, coord_index, T, Z, Y, X, M,
                     point_stride, train_phase);
  OCC_CHECK_LAUNCH("dvr_render_forward");
  return OCC_OK;
}
