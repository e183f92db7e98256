// Weight / bias gradient of a Linear on the bf16 matrix cores (bf16x3 split, f32 accumulation) for gfx950.
//
//   dW[n][k] = sum_m dY[m][n] * X[m][k]        db[n] = sum_m dY[m][n]
//
// The training-side partner of linear_bf16x3.hip (reference: the autograd of the nn.Linear call sites of a
// BEVFormerLayer — temporal_self_attention.py:197-209,266-272, spatial_cross_attention.py:334-341,173-175, mmcv FFN —
// which the reference leaves to ATen's addmm backward).  dX needs no kernel of its own: it is
// occ_linear_bf16x3_f32(dY, W^T).
//
// Shape of the problem: M = 40 000 .. 185 000 rows is the REDUCTION dimension, the output is small (N, K <= 768).
// The reduction is split over `chunks` row ranges (grid.y); every block owns one 128 x 128 output tile of one
// chunk, writes its partial tile, and a second kernel adds the partials in a fixed order (deterministic: no float
// atomics — gfx950 retires them at ~82 G/s, and rank-to-rank bit equality of gradients is worth keeping).
//
// MFMA operand layout makes LDS unnecessary: for v_mfma_f32_32x32x16_bf16 lane l holds, of the A operand, row
// l % 32 and the 8 reduction indices 8 * (l / 32) .. +7 — here: column n of dY and 8 consecutive rows m.  A dword
// load per (lane, m) therefore reads two fully used 128-byte segments per wave instruction (lanes 0-31: row m,
// lanes 32-63: row m + 8), and the 8 loaded values ARE the fragment once split into hi / lo bf16.  Same for X as
// the B operand.  Three MFMAs per tile pair and 16 rows: dYl.Xh + dYh.Xl + dYh.Xh, term-major over the wave's four
// independent accumulator tiles.
#include "common.h"

namespace occ {

typedef float wg_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned wg_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wg_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  hi = pack_bf16x2_rne(x0, x1);
  lo = pack_bf16x2_rne(x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u));
}

// 8 consecutive-m values of one column -> hi / lo fragments
__device__ __forceinline__ void wg_frag(const float (&v)[8], wg_bf16x8& hi, wg_bf16x8& lo) {
  wg_u32x4 h, l;
  unsigned a, b;
  wg_split2(v[0], v[1], a, b); h.x = a; l.x = b;
  wg_split2(v[2], v[3], a, b); h.y = a; l.y = b;
  wg_split2(v[4], v[5], a, b); h.z = a; l.z = b;
  wg_split2(v[6], v[7], a, b); h.w = a; l.w = b;
  hi = __builtin_bit_cast(wg_bf16x8, h);
  lo = __builtin_bit_cast(wg_bf16x8, l);
}

// rows m .. m+15 of this wave's two dY column tiles and two X column tiles -> raw registers.  Rows are clamped to
// M - 1 so the load is always legal; `rows_left` < 16 zeroes the rows beyond the chunk (wave-uniform branch).
__device__ __forceinline__ void wg_load(const float* __restrict__ pa0, const float* __restrict__ pa1,
                                        const float* __restrict__ pb0, const float* __restrict__ pb1, long lda,
                                        long ldb, int m, int g, int M, int rows_left, float (&a0)[8],
                                        float (&a1)[8], float (&b0)[8], float (&b1)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row = m + 8 * g + j;
    const long r = row < M ? row : M - 1;
    a0[j] = pa0[r * lda];
    a1[j] = pa1[r * lda];
    b0[j] = pb0[r * ldb];
    b1[j] = pb1[r * ldb];
  }
  if (rows_left < 16) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = 8 * g + j < rows_left;
      a0[j] = ok ? a0[j] : 0.f;
      a1[j] = ok ? a1[j] : 0.f;
      b0[j] = ok ? b0[j] : 0.f;
      b1[j] = ok ? b1[j] : 0.f;
    }
  }
}

__device__ __forceinline__ void wg_step(const float (&a0)[8], const float (&a1)[8], const float (&b0)[8],
                                        const float (&b1)[8], wg_f32x16 (&acc)[4], float& s0, float& s1) {
  wg_bf16x8 ah0, al0, ah1, al1, bh0, bl0, bh1, bl1;
  wg_frag(a0, ah0, al0);
  wg_frag(a1, ah1, al1);
  wg_frag(b0, bh0, bl0);
  wg_frag(b1, bh1, bl1);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s0 += a0[j];
    s1 += a1[j];
  }
  acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, bh0, acc[0], 0, 0, 0);
  acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, bh1, acc[1], 0, 0, 0);
  acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, bh0, acc[2], 0, 0, 0);
  acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, bh1, acc[3], 0, 0, 0);
  acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bl0, acc[0], 0, 0, 0);
  acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bl1, acc[1], 0, 0, 0);
  acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bl0, acc[2], 0, 0, 0);
  acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bl1, acc[3], 0, 0, 0);
  acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh0, acc[0], 0, 0, 0);
  acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh1, acc[1], 0, 0, 0);
  acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh0, acc[2], 0, 0, 0);
  acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh1, acc[3], 0, 0, 0);
}

// grid (tiles_n * tiles_k, chunks); block = 4 waves as 2 x 2 over a 128 (n) x 128 (k) tile
__global__ __launch_bounds__(256, 2) void linear_wgrad_x3_kernel(
    const float* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx, float* __restrict__ part_w,
    float* __restrict__ part_b, int M, int N, int K, int MC, int tiles_k) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, g = lane >> 5;
  const int tile_n = blockIdx.x / tiles_k, tile_k = blockIdx.x - tile_n * tiles_k;
  const int c = blockIdx.y;
  const int n0 = tile_n * 128 + (wave >> 1) * 64, k0 = tile_k * 128 + (wave & 1) * 64;
  const int m_begin = c * MC;
  const int m_end = m_begin + MC < M ? m_begin + MC : M;

  // column pointers; columns beyond N / K are clamped for the load and dropped at the store
  const int na0 = n0 + col, na1 = n0 + 32 + col, kb0 = k0 + col, kb1 = k0 + 32 + col;
  const float* pa0 = dy + (na0 < N ? na0 : N - 1);
  const float* pa1 = dy + (na1 < N ? na1 : N - 1);
  const float* pb0 = x + (kb0 < K ? kb0 : K - 1);
  const float* pb1 = x + (kb1 < K ? kb1 : K - 1);

  wg_f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float s0 = 0.f, s1 = 0.f;

  float a0[8], a1[8], b0[8], b1[8], c0[8], c1[8], d0[8], d1[8];
  int m = m_begin;
  if (m < m_end) wg_load(pa0, pa1, pb0, pb1, lddy, ldx, m, g, M, m_end - m, a0, a1, b0, b1);
  // two steps per iteration: the next 16 rows are requested before the current ones are consumed
  while (m < m_end) {
    const int m1 = m + 16;
    if (m1 < m_end) wg_load(pa0, pa1, pb0, pb1, lddy, ldx, m1, g, M, m_end - m1, c0, c1, d0, d1);
    wg_step(a0, a1, b0, b1, acc, s0, s1);
    if (m1 >= m_end) break;
    const int m2 = m1 + 16;
    if (m2 < m_end) wg_load(pa0, pa1, pb0, pb1, lddy, ldx, m2, g, M, m_end - m2, a0, a1, b0, b1);
    wg_step(c0, c1, d0, d1, acc, s0, s1);
    m = m2;
  }

  // partial tile: D[i][col], i = 8 * (r / 4) + 4 * g + r % 4  (rows = n, columns = k)
  float* pw = part_w + (long)c * N * K;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int nb = n0 + (t >> 1) * 32, kk = k0 + (t & 1) * 32 + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = nb + 8 * (r >> 2) + 4 * g + (r & 3);
      if (n < N && kk < K) pw[(long)n * K + kk] = acc[t][r];
    }
  }
  if (part_b && tile_k == 0 && (wave & 1) == 0) {
    s0 += __shfl_xor(s0, 32);
    s1 += __shfl_xor(s1, 32);
    if (g == 0) {
      if (na0 < N) part_b[(long)c * N + na0] = s0;
      if (na1 < N) part_b[(long)c * N + na1] = s1;
    }
  }
}

// ---- N <= 32 outputs, optional Conv3d addressing ----------------------------------------------------------------------
// Weight gradient of the decoder's 3 x 3 x 3 convolutions (reference transformer_occ.py:106-126, autograd of nn.Conv3d):
//   dW[co][tap][ci] = sum over voxels v of dY[v][co] * X[v + shift(tap)][ci]
// On ZERO-PADDED copies of x and dy (one halo voxel on every side, the same (Y+2, X+2, Z+2) grid for both) a tap is a
// constant row offset, so this is dW = dY^T . Xcol with the virtual im2col matrix Xcol[r][tap * Cin + ci] =
// Xpad[r + off(tap)][ci]: the lane that owns a column only adds its tap's offset to its base pointer.  N = 32 output
// channels = ONE dY column tile: a block is 4 waves x (32 n x 64 k) = 256 columns (the 128 x 128 tiling above would idle
// three quarters of its lanes), dY is read once per 256 columns instead of once per tap (27 separate launches of the
// kernel above took 7 ms per training step).  conv_cin = 0: plain linear addressing for N <= 32.
__global__ __launch_bounds__(256, 2) void linear_wgrad_n32_kernel(
    const float* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx, float* __restrict__ part_w, int M,
    int N, int K, int MC, int conv_cin, int xp2, int zp2) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, g = lane >> 5;
  const int c = blockIdx.y;
  const int k0 = blockIdx.x * 256 + wave * 64;
  const int m_begin = c * MC;
  const int m_end = m_begin + MC < M ? m_begin + MC : M;
  const float* pa = dy + (col < N ? col : N - 1);
  const float* pb[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    int kk = k0 + 32 * t + col;
    if (kk >= K) kk = K - 1;
    if (conv_cin > 0) {
      const int tap = kk / conv_cin, ci = kk - tap * conv_cin;
      const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
      pb[t] = x + (long)(((ky - 1) * xp2 + (kx - 1)) * zp2 + (kz - 1)) * ldx + ci;
    } else {
      pb[t] = x + kk;
    }
  }
  wg_f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  auto load = [&](int m, float (&a)[8], float (&b0)[8], float (&b1)[8]) {
    const int left = m_end - m;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = m + 8 * g + j;
      const long r = row < M ? row : M - 1;
      a[j] = pa[r * lddy];
      b0[j] = pb[0][r * ldx];
      b1[j] = pb[1][r * ldx];
    }
    if (left < 16) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool ok = 8 * g + j < left;
        a[j] = ok ? a[j] : 0.f;           // a zero dY row contributes nothing whatever X holds there
      }
    }
  };
  auto step = [&](const float (&a)[8], const float (&b0)[8], const float (&b1)[8]) {
    wg_bf16x8 ah, al, bh0, bl0, bh1, bl1;
    wg_frag(a, ah, al);
    wg_frag(b0, bh0, bl0);
    wg_frag(b1, bh1, bl1);
    // FOUR terms here (al.bl too: products exact to ~2^-24): a convolution weight's gradient is a sum over every voxel of
    // the grid of terms that largely cancel (|sum| < |term|); with three terms the 2^-16 product rounding left a noise
    // floor of 0.2 % of the largest gradient on decoder.0.conv.weight (round 4, tests/test_gpu_training.py).  The kernel
    // is load-bound: the two extra MFMAs per step are free.
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bl0, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bl1, acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh0, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh1, acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl0, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl1, acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh0, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh1, acc[1], 0, 0, 0);
  };
  float a0[8], b0[8], b1[8], a1[8], d0[8], d1[8];
  int m = m_begin;
  if (m < m_end) load(m, a0, b0, b1);
  while (m < m_end) {
    const int m1 = m + 16;
    if (m1 < m_end) load(m1, a1, d0, d1);
    step(a0, b0, b1);
    if (m1 >= m_end) break;
    const int m2 = m1 + 16;
    if (m2 < m_end) load(m2, a0, b0, b1);
    step(a1, d0, d1);
    m = m2;
  }
  float* pw = part_w + (long)c * N * K;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int kk = k0 + 32 * t + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = 8 * (r >> 2) + 4 * g + (r & 3);
      if (n < N && kk < K) pw[(long)n * K + kk] = acc[t][r];
    }
  }
}

// dW = sum_c partial[c], db likewise, in a fixed order: a block owns 64 consecutive outputs, its 4 waves each add
// every 4th chunk (4 independent loads in flight per lane), and the four wave sums are combined in wave order through
// LDS.  (One thread per output walking all chunks serially ran at 0.7 TB/s: as long as the MFMA kernel itself.)
__global__ __launch_bounds__(256) void linear_wgrad_reduce_kernel(const float* __restrict__ part_w,
                                                                  const float* __restrict__ part_b,
                                                                  float* __restrict__ dw, float* __restrict__ db,
                                                                  long NK, int N, int chunks) {
  __shared__ float red[4][64];
  const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long n_w_blocks = (NK + 63) / 64;
  const bool is_w = (long)blockIdx.x < n_w_blocks;
  const float* src = is_w ? part_w : part_b;
  const long stride = is_w ? NK : (long)N;
  const long i = (is_w ? (long)blockIdx.x : (long)blockIdx.x - n_w_blocks) * 64 + o;
  const bool ok = i < stride;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (ok) {
    int c = g;
    for (; c + 12 < chunks; c += 16) {
      s0 += src[(long)c * stride + i];
      s1 += src[(long)(c + 4) * stride + i];
      s2 += src[(long)(c + 8) * stride + i];
      s3 += src[(long)(c + 12) * stride + i];
    }
    for (; c < chunks; c += 4) s0 += src[(long)c * stride + i];
  }
  red[g][o] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && ok) {
    const float r = ((red[0][o] + red[1][o]) + red[2][o]) + red[3][o];
    if (is_w) dw[i] = r; else db[i] = r;
  }
}

static void wgrad_plan(int M, int N, int K, int& tiles_n, int& tiles_k, int& chunks, int& MC) {
  tiles_n = (N + 127) / 128;
  tiles_k = (K + 127) / 128;
  const int tiles = tiles_n * tiles_k;
  // 512 blocks = two per CU, all resident at once (640 left a quarter-full second round); every chunk costs one
  // N x K partial written and read back, so small outputs take what fills the chip and no more
  int want = (512 + tiles - 1) / tiles;
  const int max_chunks = (M + 63) / 64;              // at least 4 MFMA steps per block
  if (want > max_chunks) want = max_chunks;
  if (want < 1) want = 1;
  MC = ((M + want - 1) / want + 15) / 16 * 16;
  chunks = (M + MC - 1) / MC;
}

}  // namespace occ

extern "C" int64_t occ_linear_wgrad_workspace_bytes(int M, int N, int K) {
  using namespace occ;
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  int tn, tk, chunks, MC;
  wgrad_plan(M, N, K, tn, tk, chunks, MC);
  return (int64_t)chunks * ((int64_t)N * K + N) * 4;
}

extern "C" int occ_linear_wgrad_bf16x3_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, float* dw,
                                           float* db, void* workspace, int64_t workspace_bytes, int M, int N,
                                           int K, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(dy && x && dw && workspace, "linear_wgrad: null pointer argument");
  OCC_CHECK_ARG(M > 0 && N > 0 && K > 0, "linear_wgrad: bad dimension (M=%d N=%d K=%d)", M, N, K);
  OCC_CHECK_ARG(lddy >= N && ldx >= K, "linear_wgrad: row strides smaller than a row");
  OCC_CHECK_ARG(workspace_bytes >= occ_linear_wgrad_workspace_bytes(M, N, K),
                "linear_wgrad: workspace too small (%lld bytes, need %lld)", (long long)workspace_bytes,
                (long long)occ_linear_wgrad_workspace_bytes(M, N, K));
  OCC_CHECK_ARG(((uintptr_t)workspace & 15) == 0, "linear_wgrad: workspace must be 16-byte aligned");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int tn, tk, chunks, MC;
  wgrad_plan(M, N, K, tn, tk, chunks, MC);
  float* part_w = reinterpret_cast<float*>(workspace);
  float* part_b = part_w + (long)chunks * N * K;
  hipLaunchKernelGGL(linear_wgrad_x3_kernel, dim3((unsigned)(tn * tk), (unsigned)chunks), dim3(256), 0, st, dy,
                     (long)lddy, x, (long)ldx, part_w, db ? part_b : nullptr, M, N, K, MC, tk);
  OCC_CHECK_LAUNCH("linear_wgrad");
  const long NK = (long)N * K;
  const long red_blocks = (NK + 63) / 64 + (db ? (N + 63) / 64 : 0);
  hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3((unsigned)red_blocks), dim3(256), 0, st, part_w,
                     part_b, dw, db, NK, N, chunks);
  OCC_CHECK_LAUNCH("linear_wgrad_reduce");
  return OCC_OK;
}


// Weight gradient of a 3x3x3 / stride 1 / pad 1 Conv3d with 32 output channels (linear_wgrad_n32_kernel): dy_pad
// (B, Y+2, X+2, Z+2, 32) and x_pad (B, Y+2, X+2, Z+2, Cin) are ZERO-PADDED channels-last copies of the output gradient and
// of the input; dw (32, 27, Cin) f32 = dW[co][kz*9 + ky*3 + kx][ci] (the caller permutes to torch's (32, Cin, 3, 3, 3)).
// workspace: occ_conv3d_wgrad_workspace_bytes(B, Z, Y, X, Cin) bytes, 16-byte aligned.  Deterministic (fixed-order
// reduction of the row chunks).
extern "C" int64_t occ_conv3d_wgrad_workspace_bytes(int B, int Z, int Y, int X, int Cin) {
  using namespace occ;
  if (B <= 0 || Z <= 0 || Y <= 0 || X <= 0 || Cin <= 0) return 0;
  const long R = (long)B * (Y + 2) * (X + 2) * (Z + 2), omax = (long)(X + 2) * (Z + 2) + (Z + 2) + 1;
  const long M = R - 2 * omax;
  if (M <= 0 || M >= (1L << 31)) return 0;
  const int K = 27 * Cin, tiles = (K + 255) / 256;
  int want = (512 + tiles - 1) / tiles;
  const long max_chunks = (M + 63) / 64;
  if (want > max_chunks) want = (int)max_chunks;
  if (want < 1) want = 1;
  return (int64_t)want * 32 * K * 4;
}

extern "C" int occ_conv3d_wgrad_bf16x3_f32(const float* dy_pad, const float* x_pad, float* dw, void* workspace,
                                           int64_t workspace_bytes, int B, int Z, int Y, int X, int Cin, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(dy_pad && x_pad && dw && workspace, "conv3d_wgrad: null pointer argument");
  OCC_CHECK_ARG(B > 0 && Z > 0 && Y > 0 && X > 0 && Cin > 0, "conv3d_wgrad: bad dimension");
  const int64_t need = occ_conv3d_wgrad_workspace_bytes(B, Z, Y, X, Cin);
  OCC_CHECK_ARG(need > 0, "conv3d_wgrad: grid beyond the 32-bit row range");
  OCC_CHECK_ARG(workspace_bytes >= need && ((uintptr_t)workspace & 15) == 0,
                "conv3d_wgrad: workspace too small (%lld bytes, need %lld) or not 16-byte aligned",
                (long long)workspace_bytes, (long long)need);
  const long R = (long)B * (Y + 2) * (X + 2) * (Z + 2), omax = (long)(X + 2) * (Z + 2) + (Z + 2) + 1;
  const int M = (int)(R - 2 * omax), K = 27 * Cin, tiles = (K + 255) / 256;
  const int chunks = (int)(need / ((int64_t)32 * K * 4));
  const int MC = ((M + chunks - 1) / chunks + 15) / 16 * 16;
  const int used = (M + MC - 1) / MC;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float* part = reinterpret_cast<float*>(workspace);
  // rows [omax, R - omax): the skipped rows are halo rows of dy_pad (zero); every tap offset stays inside x_pad
  hipLaunchKernelGGL(linear_wgrad_n32_kernel, dim3((unsigned)tiles, (unsigned)used), dim3(256), 0, st,
                     dy_pad + omax * 32, 32L, x_pad + omax * Cin, (long)Cin, part, M, 32, K, MC, Cin, X + 2, Z + 2);
  OCC_CHECK_LAUNCH("conv3d_wgrad");
  const long NK = 32L * K;
  hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3((unsigned)((NK + 63) / 64)), dim3(256), 0, st, part, nullptr, dw,
                     nullptr, NK, 32, used);
  OCC_CHECK_LAUNCH("conv3d_wgrad_reduce");
  return OCC_OK;
}
