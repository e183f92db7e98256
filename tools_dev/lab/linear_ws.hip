// LAB (not built into libocc_amd.so; see tools_dev/lab/README.md for the measurements that rejected it).
// Weight-stationary persistent Linear for the encoder's tall-skinny GEMMs (M = 40 000 BEV queries, K = 256,
// N = 192 / 256 / 512 / 768) on the gfx950 bf16 matrix cores, bf16x3 arithmetic (see linear_bf16x3.hip: hi/lo-split
// operands, f32 accumulation, product error <= 2^-16) — same contract and call sites as occ_linear_bf16x3_f32
// (reference: the nn.Linear / LayerNorm call sites of a BEVFormerLayer, encoder.py:377-404,
// spatial_cross_attention.py:173,334-341, temporal_self_attention.py:198-209,266).
//
// Why another kernel (round 3): linear_bf16x3 gives every 64-row block its own pass over the weights and its own
// 16-chunk barrier chain; with 625 blocks on 256 CUs every block runs exactly once, nothing reaches a steady state and
// the launches sit at ~2x their HBM floor (profiles/r02_linear_probe.txt: 27-69 us against 13-26 us).  Here
//   * ONE block of 8 waves per CU owns a contiguous range of ~157 rows and 256 output columns;
//   * wave w keeps the hi + lo bf16 fragments of ITS 32 columns for all 256 k in registers (128 VGPRs: the weights are
//     read from L2 once per block, not once per 64 rows, and the k loop has no barrier and no weight traffic);
//   * the rows stream through LDS in 64-row tiles: the fp32 tile of step t+1 arrives by LDS-DMA
//     (global_load_lds_dwordx4, no registers) while the matrix cores work on tile t, is split once per block into
//     hi/lo bf16 planes (528-byte row pitch: conflict-free ds_read_b128 of the A fragments), and all 8 waves read
//     the same planes;
//   * the residual rows of tile t are requested right after its k loop, under the epilogue's LDS transpose;
//   * epilogue per tile through an LDS transpose (the planes' space): bias, ReLU, residual, two-pass LayerNorm by
//     one wave per row, 16-byte stores.
// A launch moves every input byte once and writes every output byte once; what is left is HBM time.
#include "common.h"

namespace occ {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int kWsK = 256;                 // k per pass (all of it: weight-stationary)
constexpr int kWsTile = 64;               // rows per tile
constexpr int kWsPitch = kWsK * 2 + 16;   // bytes per row of one bf16 plane (528: 132 dwords = 4 mod 64 banks)
constexpr int kWsPlane = kWsTile * kWsPitch;            // 33 792
constexpr int kWsRaw = kWsTile * kWsK * 4;              // 65 536: the fp32 tile as the DMA lands it
constexpr int kWsOLd = 256 + 4;                         // floats per row of the epilogue tile
static_assert(kWsTile * kWsOLd * 4 <= 2 * kWsPlane, "the epilogue tile overlays the two planes");

__device__ __forceinline__ void ws_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  hi = pack_bf16x2_rne(x0, x1);
  lo = pack_bf16x2_rne(x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u));
}
__device__ __forceinline__ float ws_wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// DMA of one 64-row fp32 tile (rows row0 .. row0+63, clamped to [.., row_last]) into `raw`: wave w fetches rows
// w, w+8, ..; one instruction = one 1 KB row, lane l's 16 bytes land at raw + r*1024 + 16*l
__device__ __forceinline__ void ws_dma_tile(const float* __restrict__ a, long lda, long row0, long row_last,
                                            char* raw, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < kWsTile / 8; ++j) {
    const int r = wave + 8 * j;
    long m = row0 + r;
    if (m > row_last) m = row_last;
    __builtin_amdgcn_global_load_lds(a + m * lda + lane * 4, (lds_ptr_t)(raw + r * (kWsK * 4)), 16, 0, 0);
  }
}

// raw fp32 tile -> hi / lo bf16 planes (each thread 8 x 16 bytes; a wave reads one full raw row per instruction).
// k-chunk c (16 values) of a row lands in plane slot (c - rot) & 15.
__device__ __forceinline__ void ws_split_tile(const char* raw, char* planes, int tid, int rot = 0) {
#pragma unroll
  for (int j = 0; j < kWsTile * kWsK / 4 / 512; ++j) {
    const int c = tid + 512 * j, r = c >> 6, q = c & 63;
    const float4 f = *reinterpret_cast<const float4*>(raw + r * (kWsK * 4) + q * 16);
    unsigned h01, h23, l01, l23;
    ws_split2(f.x, f.y, h01, l01);
    ws_split2(f.z, f.w, h23, l23);
    const int dq = ((((q >> 2) - rot) & 15) << 2) | (q & 3);
    *reinterpret_cast<uint2*>(planes + r * kWsPitch + dq * 8) = make_uint2(h01, h23);
    *reinterpret_cast<uint2*>(planes + kWsPlane + r * kWsPitch + dq * 8) = make_uint2(l01, l23);
  }
}


// k loop of one 32-row half tile against this wave's 32 columns: A fragments (hi, lo planes) read one k-step ahead,
// three MFMAs per k-step on ONE accumulator (the co-resident wave of the SIMD fills the dependent-issue gaps).  One
// accumulator tile at a time keeps the kernel inside 256 registers with the 128 weight registers AND a fragment
// prefetch — with two tiles live hipcc had 8 registers left for fragments and waited out every LDS read.
// (Every block walks k in its own rotation — register set j and plane slot j hold k-chunk (j + rot) & 15, arranged by
// the weight loads and by ws_split_tile — so the 256 blocks that start together do not ask the same L2 channel for the
// same weight lines at the same time; the loop itself sees static offsets.)
template <bool TRANSPOSED>
__device__ __forceinline__ f32x16 ws_kloop32(const char* pa, const uint4 (&wh)[16], const uint4 (&wl)[16]) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  bf16x8 fh[2], fl[2];
  fh[0] = *reinterpret_cast<const bf16x8*>(pa);
  fl[0] = *reinterpret_cast<const bf16x8*>(pa + kWsPlane);
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    if (ks + 1 < 16) {
      fh[(ks + 1) & 1] = *reinterpret_cast<const bf16x8*>(pa + (ks + 1) * 32);
      fl[(ks + 1) & 1] = *reinterpret_cast<const bf16x8*>(pa + kWsPlane + (ks + 1) * 32);
    }
    const bf16x8 bh = __builtin_bit_cast(bf16x8, wh[ks]), bl = __builtin_bit_cast(bf16x8, wl[ks]);
    const bf16x8 ah = fh[ks & 1], al = fl[ks & 1];
    if (TRANSPOSED) {        // D = W . A^T (weights as the row operand)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, al, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, ah, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah, acc, 0, 0, 0);
    } else {                 // small terms first
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    }
  }
  return acc;
}

__global__ __launch_bounds__(512, 2) void linear_ws_kernel(
    const float* __restrict__ a, long lda, const uint4* __restrict__ wp, int NT32,
    const float* __restrict__ bias, int act, const float* __restrict__ residual, long ldres, int nres,
    const float* __restrict__ ln_g, const float* __restrict__ ln_b, float ln_eps, float* __restrict__ out,
    long ldo, int M, int N, int rows_per_block) {
  __shared__ __attribute__((aligned(16))) char lds[2 * kWsPlane + kWsRaw];
  char* planes = lds;
  char* raw = lds + 2 * kWsPlane;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int vi = lane & 31, kb = lane >> 5;
  const long r_begin = (long)blockIdx.x * rows_per_block;
  long r_end = r_begin + rows_per_block;
  if (r_end > M) r_end = M;
  if (r_begin >= r_end) return;
  const int n0 = blockIdx.y * 256;
  const int ntiles = (int)((r_end - r_begin + kWsTile - 1) / kWsTile);

  ws_dma_tile(a, lda, r_begin, r_end - 1, raw, wave, lane);
  const int rot = (int)((blockIdx.x * 5u + blockIdx.y * 3u) & 15u);

  // this wave's 32 columns: hi / lo fragments of all 16 k-steps (waves past the last column tile idle in the k loop)
  const int nt = n0 / 32 + wave;
  const bool live = nt < NT32;
  uint4 wh[16], wl[16];
  {
    const long base = (long)(live ? nt : NT32 - 1) * 128 + lane;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const long kc = (ks + rot) & 15;
      wh[ks] = wp[kc * NT32 * 128 + base];
      wl[ks] = wp[kc * NT32 * 128 + base + 64];
    }
  }
  const int c = lane * 4;                                   // this lane's 4 columns of the block's 256 in the epilogue
  const bool col_live = n0 + c < N;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), gv = bv, bev = bv;
  if (col_live) {
    if (bias) bv = *reinterpret_cast<const float4*>(bias + n0 + c);
    if (ln_g) {
      gv = *reinterpret_cast<const float4*>(ln_g + n0 + c);
      bev = *reinterpret_cast<const float4*>(ln_b + n0 + c);
    }
  }
  const bool res_live = residual != nullptr && col_live && n0 + c < nres;
  const float inv_n = 1.f / (float)N;
  float* sO = reinterpret_cast<float*>(planes);

  __syncthreads();                     // the DMA of tile 0 has landed (the barrier's release waits vmcnt(0))
  ws_split_tile(raw, planes, tid, rot);
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const long row0 = r_begin + (long)t * kWsTile;
    if (t + 1 < ntiles) ws_dma_tile(a, lda, row0 + kWsTile, r_end - 1, raw, wave, lane);
    // the two 32-row halves one after the other; the results wait in registers until every wave has left the planes
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    if (live) {
      const char* pa = planes + vi * kWsPitch + kb * 16;
      acc0 = ws_kloop32<false>(pa, wh, wl);
      acc1 = ws_kloop32<false>(pa + 32 * kWsPitch, wh, wl);
    }
    __syncthreads();                   // every wave is done with the planes (and the next tile's DMA has landed)
    // residual rows of this tile (8 per wave): requested here, their latency hides under the transpose below — held
    // across the k loop they cost the 32 registers the fragment prefetch needs
    float4 rres[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      long m = row0 + wave * 8 + rr;
      if (m > r_end - 1) m = r_end - 1;
      rres[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (res_live) rres[rr] = *reinterpret_cast<const float4*>(residual + m * ldres + n0 + c);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * kb;
      sO[row * kWsOLd + wave * 32 + vi] = acc0[r];
      sO[(row + 32) * kWsOLd + wave * 32 + vi] = acc1[r];
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int row = wave * 8 + rr;
      const long m = row0 + row;
      if (m >= r_end) break;                   // wave-uniform
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (col_live) {
        v = *reinterpret_cast<const float4*>(sO + row * kWsOLd + c);
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        if (act == 1) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        v.x += rres[rr].x; v.y += rres[rr].y; v.z += rres[rr].z; v.w += rres[rr].w;
      }
      if (ln_g) {                              // LayerNorm over the N columns (N <= 256: one column group)
        const float mean = ws_wave_sum(col_live ? (v.x + v.y) + (v.z + v.w) : 0.f) * inv_n;
        const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
        const float var = ws_wave_sum(col_live ? (dx * dx + dy * dy) + (dz * dz + dw * dw) : 0.f) * inv_n;
        const float rstd = rsqrtf(var + ln_eps);
        v.x = dx * rstd * gv.x + bev.x; v.y = dy * rstd * gv.y + bev.y;
        v.z = dz * rstd * gv.z + bev.z; v.w = dw * rstd * gv.w + bev.w;
      }
      if (col_live) *reinterpret_cast<float4*>(out + m * ldo + n0 + c) = v;
    }
    if (t + 1 < ntiles) {
      __syncthreads();                 // the epilogue tile is consumed before the planes are rewritten
      ws_split_tile(raw, planes, tid, rot);
      __syncthreads();
    }
  }
}

}  // namespace occ

extern "C" int occ_linear_ws_bf16x3_f32(const float* a, int64_t lda, int K, const void* weight_packed,
                                        const float* bias, int act, const float* residual, int64_t ldres,
                                        int residual_cols, const float* ln_gamma, const float* ln_beta, float ln_eps,
                                        float* out, int64_t ldo, int M, int N, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(a && weight_packed && out, "linear_ws: null pointer argument");
  OCC_CHECK_ARG(M > 0 && N > 0, "linear_ws: bad dimension (M=%d N=%d)", M, N);
  OCC_CHECK_ARG(act == 0 || act == 1, "linear_ws: act must be 0 (none) or 1 (ReLU)");
  OCC_CHECK_ARG((ln_gamma == nullptr) == (ln_beta == nullptr), "linear_ws: ln_gamma and ln_beta go together");
  OCC_CHECK_ARG(lda >= K && ldo >= N && (!residual || ldres >= residual_cols),
                "linear_ws: leading dimension smaller than the row");
  if (K != kWsK || N % 4 || lda % 4 || ldo % 4 || (residual && (ldres % 4 || residual_cols % 4)) ||
      (ln_gamma && N > 256) || (reinterpret_cast<uintptr_t>(a) & 15)) {
    set_error("linear_ws: no kernel for K=%d N=%d (need K == 256, N %% 4 == 0, 16-byte aligned rows, N <= 256 with "
              "LayerNorm)", K, N);
    return OCC_E_UNSUPPORTED;
  }
  const int ncg = (N + 255) / 256;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  // one block per CU and round: row ranges so that (row ranges x column groups) ~ the CU count
  int nrb = cus / ncg;
  if (nrb < 1) nrb = 1;
  const int max_rb = (M + kWsTile - 1) / kWsTile;
  if (nrb > max_rb) nrb = max_rb;
  const int rows_per_block = (M + nrb - 1) / nrb;
  nrb = (M + rows_per_block - 1) / rows_per_block;
  hipLaunchKernelGGL(linear_ws_kernel, dim3((unsigned)nrb, (unsigned)ncg), dim3(512), 0,
                     reinterpret_cast<hipStream_t>(stream), a, (long)lda,
                     reinterpret_cast<const uint4*>(weight_packed), (N + 31) / 32, bias, act, residual, (long)ldres,
                     residual ? residual_cols : 0, ln_gamma, ln_beta, ln_eps, out, (long)ldo, M, N, rows_per_block);
  OCC_CHECK_LAUNCH("linear_ws");
  return OCC_OK;
}
