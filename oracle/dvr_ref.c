/* Plain-C restatement of the reference's DVR ray caster — TEST INFRASTRUCTURE (oracle; see
 * oracle/__init__.py).  Follows tools/ray_iou/lib/dvr/dvr.cu:70-319 (`render_forward_cuda_kernel`) step
 * by step, including the recorded path / distance arrays and the final scan for the first voxel with
 * occupancy > 0.5, and :353-356 for the output initialisation.  Double arithmetic on float inputs, as the
 * reference.  Compile WITHOUT floating-point contraction (oracle/Makefile passes -ffp-contract=off).
 *
 * Pinning status: "parity unpinned" — the reference component is CUDA (needs nvcc + the torch extension
 * API, unbuildable here) and ships no test vectors; the HIP kernel is checked bit-exactly against this
 * restatement only.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>

#define MAX_D 1446
#define MAX_STEP 1000

void dvr_render_forward_ref(const float* sigma, const float* origin, const float* points,
                            const float* tindex, float* pred_dist, float* gt_dist, float* coord_index,
                            int N, int T, int vzsize, int vysize, int vxsize, int M, int pstride,
                            int train_phase) {
  int (*path)[3] = malloc(sizeof(int[3]) * MAX_D);
  double* d = malloc(sizeof(double) * MAX_D);
  for (long i = 0; i < (long)N * M; ++i) { pred_dist[i] = -1.f; gt_dist[i] = -1.f; }
  for (long i = 0; i < (long)N * M * 3; ++i) coord_index[i] = 0.f;
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < M; ++c) {
      const float tf = tindex[(long)n * M + c];
      if (tf < 0) continue;
      const int t = (int)tf;
      const int ts = (T == 1) ? 0 : t;
      const float* o = origin + ((long)n * T + t) * 3;
      const float* e = points + ((long)n * M + c) * pstride;
      const double xo = o[0], yo = o[1], zo = o[2];
      const double xe = e[0], ye = e[1], ze = e[2];
      int vx = (int)xo, vy = (int)yo, vz = (int)zo;
      const double rx = xe - xo, ry = ye - yo, rz = ze - zo;
      double gt_d = sqrt(rx * rx + ry * ry + rz * rz);
      const double dx = rx / gt_d, dy = ry / gt_d, dz = rz / gt_d;
      const int stepX = (dx >= 0) ? 1 : -1, stepY = (dy >= 0) ? 1 : -1, stepZ = (dz >= 0) ? 1 : -1;
      const double nbx = vx + (stepX < 0 ? 0 : 1), nby = vy + (stepY < 0 ? 0 : 1),
                   nbz = vz + (stepZ < 0 ? 0 : 1);
      double tMaxX = (dx != 0) ? (nbx - xo) / dx : DBL_MAX;
      double tMaxY = (dy != 0) ? (nby - yo) / dy : DBL_MAX;
      double tMaxZ = (dz != 0) ? (nbz - zo) / dz : DBL_MAX;
      const double tDeltaX = (dx != 0) ? stepX / dx : DBL_MAX;
      const double tDeltaY = (dy != 0) ? stepY / dy : DBL_MAX;
      const double tDeltaZ = (dz != 0) ? stepZ / dz : DBL_MAX;
      const float* grid = sigma + ((long)n * T + ts) * vzsize * vysize * vxsize;
      int step = 0, count = 0, was_inside = 0;
      while (1) {
        const int inside = (0 <= vx && vx < vxsize) && (0 <= vy && vy < vysize) && (0 <= vz && vz < vzsize);
        if (inside) {
          was_inside = 1;
          path[count][0] = vx; path[count][1] = vy; path[count][2] = vz;
        } else if (was_inside) {
          break;
        }
        double _d;
        if (tMaxX < tMaxY) {
          if (tMaxX < tMaxZ) { _d = tMaxX; vx += stepX; tMaxX += tDeltaX; }
          else { _d = tMaxZ; vz += stepZ; tMaxZ += tDeltaZ; }
        } else {
          if (tMaxY < tMaxZ) { _d = tMaxY; vy += stepY; tMaxY += tDeltaY; }
          else { _d = tMaxZ; vz += stepZ; tMaxZ += tDeltaZ; }
        }
        if (inside) { d[count] = _d; count++; }
        step++;
        if (step > MAX_STEP) break;
      }
      if (count > 0) {
        double exp_d = d[count - 1];
        int x = path[count - 1][0], y = path[count - 1][1], z = path[count - 1][2];
        for (int i = 0; i < count; ++i) {
          const double occ = grid[((long)path[i][2] * vysize + path[i][1]) * vxsize + path[i][0]];
          if (occ > 0.5) { exp_d = d[i]; x = path[i][0]; y = path[i][1]; z = path[i][2]; break; }
        }
        const double max_d = d[count - 1];
        if (train_phase == 1) gt_d = gt_d < max_d ? gt_d : max_d;
        const long oi = (long)n * M + c;
        pred_dist[oi] = (float)exp_d;
        gt_dist[oi] = (float)gt_d;
        coord_index[oi * 3 + 0] = (float)x;
        coord_index[oi * 3 + 1] = (float)y;
        coord_index[oi * 3 + 2] = (float)z;
      }
    }
  free(path);
  free(d);
}
