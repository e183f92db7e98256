// Training-step glue of a BEVFormerLayer: y = LayerNorm(dropout(x) + residual) and its backward, one launch each.
//
// The reference's layer runs  attention / FFN -> Dropout(0.1) -> + identity -> LayerNorm  (encoder.py:377-404 with the module
// tails spatial_cross_attention.py:173-175, temporal_self_attention.py:270-272 and mmcv's FFN); as ATen ops that is a dropout
// kernel, an add and a LayerNorm forward, and in backward LayerNorm's two kernels, a masked scale and the adds of the
// residual fork: five to six passes over the (40 000, 256) activation per site, twelve sites per step.  Here:
//   forward : z = x * keep(i) / (1 - p) + residual;  y = (z - mean) * rstd * gamma + beta;   writes z, y, (mean, rstd)
//   backward: gz = rstd * (gy*gamma - mean_c(gy*gamma) - zhat * mean_c(gy*gamma*zhat));  g_residual = gz;
//             g_x = gz * keep(i) / (1 - p);  dgamma = sum_rows gy * zhat;  dbeta = sum_rows gy
// keep(i) is a counter-based hash of (seed, element index) against the drop probability: the mask is never stored, the
// backward regenerates it (the seed is drawn on the host from torch's CPU generator: reproducible under torch.manual_seed,
// no device sync).  One wave per row (C = 256: a lane holds 4 consecutive columns = one 16-byte piece of the row), row sums
// by shuffle; the column sums of the backward stay in registers over the block's rows, meet in LDS and leave as one row of
// partial sums per block (fixed-order second stage: csrc/bias_act_nhwc.hip's reduce — deterministic).  HBM-bound.
#include "common.h"

namespace occ {

__device__ __forceinline__ unsigned ln_hash(unsigned i, unsigned s0, unsigned s1) {
  unsigned h = i ^ s0;                                  // murmur3's finaliser over (index ^ seed), salted once more
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  h += s1;
  h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
  return h;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

template <bool DROP>
__global__ __launch_bounds__(256) void dropout_add_ln_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, unsigned s0, unsigned s1, unsigned thresh, float scale,
    float* __restrict__ z, float* __restrict__ y, float2* __restrict__ stats, long rows) {
  constexpr int C = 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float4 g4 = *reinterpret_cast<const float4*>(gamma + lane * 4), b4 = *reinterpret_cast<const float4*>(beta + lane * 4);
  for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
    const long o = r * C + lane * 4;
    float4 v = *reinterpret_cast<const float4*>(x + o);
    const float4 q = *reinterpret_cast<const float4*>(res + o);
    if (DROP) {
      const unsigned i = (unsigned)o;
      v.x = ln_hash(i, s0, s1) >= thresh ? v.x * scale : 0.f;
      v.y = ln_hash(i + 1, s0, s1) >= thresh ? v.y * scale : 0.f;
      v.z = ln_hash(i + 2, s0, s1) >= thresh ? v.z * scale : 0.f;
      v.w = ln_hash(i + 3, s0, s1) >= thresh ? v.w * scale : 0.f;
    }
    v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    const float mean = wave_sum((v.x + v.y) + (v.z + v.w)) * (1.f / C);
    const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
    const float var = wave_sum((dx * dx + dy * dy) + (dz * dz + dw * dw)) * (1.f / C);
    const float rstd = rsqrtf(var + eps);
    *reinterpret_cast<float4*>(z + o) = v;
    *reinterpret_cast<float4*>(y + o) = make_float4(fmaf(dx * rstd, g4.x, b4.x), fmaf(dy * rstd, g4.y, b4.y),
                                                    fmaf(dz * rstd, g4.z, b4.z), fmaf(dw * rstd, g4.w, b4.w));
    if (lane == 0) stats[r] = make_float2(mean, rstd);
  }
}

template <bool DROP>
__global__ __launch_bounds__(256) void dropout_add_ln_bwd_kernel(
    const float* __restrict__ gy, const float* __restrict__ z, const float2* __restrict__ stats,
    const float* __restrict__ gamma, unsigned s0, unsigned s1, unsigned thresh, float scale, float* __restrict__ gx,
    float* __restrict__ gres, float* __restrict__ partial, long rows) {
  constexpr int C = 256;
  __shared__ float red[4][2][C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float4 g4 = *reinterpret_cast<const float4*>(gamma + lane * 4);
  float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg;
  for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
    const long o = r * C + lane * 4;
    const float4 g = *reinterpret_cast<const float4*>(gy + o);
    const float4 v = *reinterpret_cast<const float4*>(z + o);
    const float2 st = stats[r];
    const float hx = (v.x - st.x) * st.y, hy = (v.y - st.x) * st.y, hz = (v.z - st.x) * st.y, hw = (v.w - st.x) * st.y;
    const float wx = g.x * g4.x, wy = g.y * g4.y, wz = g.z * g4.z, ww = g.w * g4.w;
    const float c1 = wave_sum((wx + wy) + (wz + ww)) * (1.f / C);
    const float c2 = wave_sum((wx * hx + wy * hy) + (wz * hz + ww * hw)) * (1.f / C);
    float4 t = make_float4(st.y * (wx - c1 - hx * c2), st.y * (wy - c1 - hy * c2), st.y * (wz - c1 - hz * c2),
                           st.y * (ww - c1 - hw * c2));
    *reinterpret_cast<float4*>(gres + o) = t;
    if (DROP) {
      const unsigned i = (unsigned)o;
      t.x = ln_hash(i, s0, s1) >= thresh ? t.x * scale : 0.f;
      t.y = ln_hash(i + 1, s0, s1) >= thresh ? t.y * scale : 0.f;
      t.z = ln_hash(i + 2, s0, s1) >= thresh ? t.z * scale : 0.f;
      t.w = ln_hash(i + 3, s0, s1) >= thresh ? t.w * scale : 0.f;
      *reinterpret_cast<float4*>(gx + o) = t;
    }
    dg.x = fmaf(g.x, hx, dg.x); dg.y = fmaf(g.y, hy, dg.y); dg.z = fmaf(g.z, hz, dg.z); dg.w = fmaf(g.w, hw, dg.w);
    db.x += g.x; db.y += g.y; db.z += g.z; db.w += g.w;
  }
  *reinterpret_cast<float4*>(&red[wave][0][lane * 4]) = dg;
  *reinterpret_cast<float4*>(&red[wave][1][lane * 4]) = db;
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int k = i / C, c = i - k * C;
    partial[(long)blockIdx.x * 2 * C + i] = (red[0][k][c] + red[1][k][c]) + (red[2][k][c] + red[3][k][c]);
  }
}

// second stage (same shape as csrc/bias_act_nhwc.hip's): 8 row groups x 32 columns per block, fixed order
__global__ __launch_bounds__(256) void ln_colsum_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                               int blocks, int C2) {
  __shared__ float red[8][32];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < C2) {
    int b = rg;
    for (; b + 24 < blocks; b += 32) {
      s0 += partial[(long)b * C2 + c];
      s1 += partial[(long)(b + 8) * C2 + c];
      s2 += partial[(long)(b + 16) * C2 + c];
      s3 += partial[(long)(b + 24) * C2 + c];
    }
    for (; b < blocks; b += 8) s0 += partial[(long)b * C2 + c];
  }
  red[rg][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && c < C2) {
    float s = red[0][cl];
#pragma unroll
    for (int r = 1; r < 8; ++r) s += red[r][cl];
    out[c] = s;
  }
}

static int ln_train_blocks(long rows) {
  long b = (rows + 3) / 4;
  return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}

}  // namespace occ

// the keep decision of element i (tests restate it on the host): hash(i, seed) >= drop_threshold
extern "C" unsigned occ_ln_dropout_hash(unsigned index, unsigned seed_lo, unsigned seed_hi) {
  unsigned h = index ^ seed_lo;
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  h += seed_hi;
  h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
  return h;
}

extern "C" int64_t occ_dropout_add_ln_bwd_partial_floats(int64_t rows, int C) {
  if (rows <= 0 || C != 256) return 0;
  return (int64_t)occ::ln_train_blocks(rows) * 2 * C;
}

extern "C" int occ_dropout_add_ln_fwd_f32(const float* x, const float* residual, const float* gamma, const float* beta,
                                          float eps, float p_drop, uint64_t seed, float* z, float* y, float* stats,
                                          int64_t rows, int C, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(x && residual && gamma && beta && z && y && stats, "dropout_add_ln_fwd: null pointer argument");
  OCC_CHECK_ARG(rows > 0 && p_drop >= 0.f && p_drop < 1.f, "dropout_add_ln_fwd: bad argument (rows=%ld p=%g)", (long)rows,
                (double)p_drop);
  if (C != 256 || rows * (long)C >= (1L << 32)) {
    set_error("dropout_add_ln_fwd: C=%d rows=%ld: the kernel covers C = 256 and fewer than 2^32 elements", C, (long)rows);
    return OCC_E_UNSUPPORTED;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const unsigned thresh = (unsigned)((double)p_drop * 4294967296.0);
  const float scale = 1.f / (1.f - p_drop);
  const dim3 grid((unsigned)ln_train_blocks(rows));
  if (p_drop > 0.f)
    hipLaunchKernelGGL(dropout_add_ln_fwd_kernel<true>, grid, dim3(256), 0, st, x, residual, gamma, beta, eps, (unsigned)seed,
                       (unsigned)(seed >> 32), thresh, scale, z, y, reinterpret_cast<float2*>(stats), (long)rows);
  else
    hipLaunchKernelGGL(dropout_add_ln_fwd_kernel<false>, grid, dim3(256), 0, st, x, residual, gamma, beta, eps, 0u, 0u, 0u, 1.f,
                       z, y, reinterpret_cast<float2*>(stats), (long)rows);
  OCC_CHECK_LAUNCH("dropout_add_ln_fwd");
  return OCC_OK;
}

extern "C" int occ_dropout_add_ln_bwd_f32(const float* grad_y, const float* z, const float* stats, const float* gamma,
                                          float p_drop, uint64_t seed, float* grad_x, float* grad_residual, float* partial,
                                          float* grad_gamma_beta, int64_t rows, int C, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(grad_y && z && stats && gamma && grad_residual && partial && grad_gamma_beta && (p_drop == 0.f || grad_x),
                "dropout_add_ln_bwd: null pointer argument");
  OCC_CHECK_ARG(rows > 0 && p_drop >= 0.f && p_drop < 1.f, "dropout_add_ln_bwd: bad argument (rows=%ld p=%g)", (long)rows,
                (double)p_drop);
  if (C != 256 || rows * (long)C >= (1L << 32)) {
    set_error("dropout_add_ln_bwd: C=%d rows=%ld: the kernel covers C = 256 and fewer than 2^32 elements", C, (long)rows);
    return OCC_E_UNSUPPORTED;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const unsigned thresh = (unsigned)((double)p_drop * 4294967296.0);
  const float scale = 1.f / (1.f - p_drop);
  const int blocks = ln_train_blocks(rows);
  if (p_drop > 0.f)
    hipLaunchKernelGGL(dropout_add_ln_bwd_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, grad_y, z,
                       reinterpret_cast<const float2*>(stats), gamma, (unsigned)seed, (unsigned)(seed >> 32), thresh, scale, grad_x,
                       grad_residual, partial, (long)rows);
  else
    hipLaunchKernelGGL(dropout_add_ln_bwd_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, grad_y, z,
                       reinterpret_cast<const float2*>(stats), gamma, 0u, 0u, 0u, 1.f, grad_x, grad_residual, partial, (long)rows);
  OCC_CHECK_LAUNCH("dropout_add_ln_bwd");
  hipLaunchKernelGGL(ln_colsum_reduce_kernel, dim3((unsigned)((2 * C + 31) / 32)), dim3(256), 0, st, partial, grad_gamma_beta,
                     blocks, 2 * C);
  OCC_CHECK_LAUNCH("dropout_add_ln_bwd (reduce)");
  return OCC_OK;
}
