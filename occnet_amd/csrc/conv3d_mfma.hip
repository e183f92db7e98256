// BEV -> voxel lifter + 3x3x3 Conv3d + BatchNorm3d(eval) + ReLU as one implicit-GEMM kernel on the
// gfx950 f32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, bitwise an fmaf chain).
//
// Replaces (reference: projects/mmdet3d_plugin/bevformer/modules/transformer_occ.py):
//   :305-307  bev_embed.permute/view -> (bs, C/Z, Z, H, W)      [the "lifter": channel c of the BEV
//             embedding is feature c // Z at height c % Z; a free view on our (bs, H*W, C) buffer]
//   :106-126  ConvModule(Conv3d k3 p1 no-bias + BN3d + ReLU) x2 (self.decoder)
//   :308      permute(0,4,3,2,1) -> (bs, W, H, Z, C): folded into the output strides of the 2nd conv.
//
// GEMM view: M = voxels, N = Cout = 32, K = 27 taps x Cin.  A 32-voxel row tile is PX = 32/Z
// x-adjacent pillars x all Z heights, so one MFMA D tile (32 voxels x 32 output channels) is whole
// pillars.  A block of 4 waves owns TY x TX pillars (NACC row tiles per wave, B fragments shared by
// the NACC accumulators).  Per phase of CH input channels the block stages the (TY+2)x(TX+2)x(Z+2)
// halo once into LDS — voxel slots of CH+4 floats, pillar stride a multiple of 256 B, so the per-tap
// ds_read_b128 of the 32 voxels of a row tile is bank-conflict free — and walks the 27 taps with
// compile-time LDS offsets.  k-pairing inside a tap: lanes 0-31 contract channels [0,CH/2), lanes
// 32-63 channels [CH/2,CH) (the order of the K sum is free), so a lane's A fragment for one tap is
// CH/2 contiguous floats of ONE voxel and its B fragment CH/2 contiguous floats of the packed weight.
// Zero padding of the convolution = zero-filled halo slots.  Epilogue: y = relu(acc*scale + shift)
// with the eval-mode BatchNorm folded into (scale, shift) per output channel, 128-byte rows per voxel.
#include "common.h"

namespace occ {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int Z, int CH, int TY, int TX>
struct ConvGeom {
  static_assert(32 % Z == 0 && Z % 4 == 0, "Z must be 4, 8, 16 or 32");
  static_assert(CH == 8 || CH == 16, "CH must be 8 or 16");
  static constexpr int PX = 32 / Z;                          // pillars per 32-voxel row tile
  static_assert(TX % PX == 0, "TX must be a multiple of the pillars per row tile");
  static constexpr int VS = CH + 4;                          // floats per voxel slot in LDS
  static constexpr int PS = ((Z + 2) * VS + 63) / 64 * 64;   // floats per pillar (256-B multiple)
  static constexpr int HX = TX + 2, HY = TY + 2;
  static constexpr int LDS_FLOATS = HY * HX * PS;
  static constexpr int ROW_TILES = TY * TX / PX;
  static_assert(ROW_TILES % 4 == 0, "row tiles must split over 4 waves");
  static constexpr int NACC = ROW_TILES / 4;
  static constexpr int TXG = TX / PX;                        // row tiles per pillar row
};

// LAYOUT 0: in[b][y][x][z][Cin]          (channels innermost: output layout of this kernel)
// LAYOUT 1: in[b][y][x][Cin][Z]          (lifter view of the BEV embedding: c = ci*Z + z)
template <int Z, int CH, int TY, int TX, int LAYOUT>
__global__ __launch_bounds__(256) void conv3d_mfma_kernel(
    const float* __restrict__ in, const float* __restrict__ wp, const float* __restrict__ scale,
    const float* __restrict__ shift, float* __restrict__ out, int Y, int X, int Cin, long out_sb,
    long out_sy, long out_sx, int relu, int tiles_x, int tiles_y) {
  using G = ConvGeom<Z, CH, TY, TX>;
  constexpr int VS = G::VS, PS = G::PS, HX = G::HX, HY = G::HY, NACC = G::NACC, H2 = CH / 2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware tile order: hardware block b runs on XCD b % 8, each with a private L2.  Dealt linearly, the eight tiles
  // of a row segment land on eight different L2s and every XCD fetches its own copy of the shared halos (PMC, round 2:
  // 306 MB read for an 82 MB input).  Here XCD x walks the contiguous tile range [x*q + min(x, r), ...) — neighbours in
  // space are neighbours in time on ONE L2 (bijective for any grid size: q = n / 8, r = n % 8).
  int bid;
  {
    const int n = (int)gridDim.x, q = n >> 3, r = n & 7, x = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
    bid = x * q + (x < r ? x : r) + j;
  }
  const int tx_i = bid % tiles_x;
  bid /= tiles_x;
  const int ty_i = bid % tiles_y;
  const int b = bid / tiles_y;
  const int y0 = ty_i * TY, x0 = tx_i * TX;
  const float* inb = in + (long)b * Y * X * Z * Cin;

  // z-halo slots (z = -1 and z = Z) of every halo pillar stay zero for all phases
  for (int i = tid; i < HY * HX * 2 * VS; i += 256) {
    const int pil = i / (2 * VS), rem = i % (2 * VS);
    lds[pil * PS + (rem >= VS ? (Z + 1) * VS : 0) + rem % VS] = 0.f;
  }

  f32x16 acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  const int vi = lane & 31, kh = lane >> 5;
  int abase[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    const int rt = wave * NACC + a;
    const int ty = rt / G::TXG, txg = rt % G::TXG;
    const int px = txg * G::PX + vi / Z, z = vi % Z;
    abase[a] = (ty * HX + px) * PS + z * VS + kh * H2;  // tap (0,0,0) = halo corner (-1,-1,-1)
  }

  const int nph = Cin / CH;
  for (int p = 0; p < nph; ++p) {
    if (p) __syncthreads();  // every wave is done reading the previous phase's halo
    // ---- stage CH channels of the halo ------------------------------------------------------
    if (LAYOUT == 0) {
      constexpr int PARTS = CH / 4, ITEMS = HY * HX * Z * PARTS, ITERS = (ITEMS + 255) / 256;
      float4 v[ITERS];
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int idx = tid + it * 256;
        const int part = idx % PARTS, z = (idx / PARTS) % Z, pil = idx / (PARTS * Z);
        const int gy = y0 + pil / HX - 1, gx = x0 + pil % HX - 1;
        v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < ITEMS && gy >= 0 && gy < Y && gx >= 0 && gx < X)
          v[it] = *reinterpret_cast<const float4*>(inb + (((long)gy * X + gx) * Z + z) * Cin +
                                                   p * CH + part * 4);
      }
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int idx = tid + it * 256;
        const int part = idx % PARTS, z = (idx / PARTS) % Z, pil = idx / (PARTS * Z);
        if (idx < ITEMS)
          *reinterpret_cast<float4*>(lds + pil * PS + (z + 1) * VS + part * 4) = v[it];
      }
    } else {
      constexpr int Z4 = Z / 4, ITEMS = HY * HX * CH * Z4, ITERS = (ITEMS + 255) / 256;
      float4 v[ITERS];
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int idx = tid + it * 256;
        const int z4 = idx % Z4, ci = (idx / Z4) % CH, pil = idx / (Z4 * CH);
        const int gy = y0 + pil / HX - 1, gx = x0 + pil % HX - 1;
        v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < ITEMS && gy >= 0 && gy < Y && gx >= 0 && gx < X)
          v[it] = *reinterpret_cast<const float4*>(inb + ((long)gy * X + gx) * Z * Cin +
                                                   (long)(p * CH + ci) * Z + z4 * 4);
      }
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int idx = tid + it * 256;
        const int z4 = idx % Z4, ci = (idx / Z4) % CH, pil = idx / (Z4 * CH);
        if (idx < ITEMS) {
          float* d = lds + pil * PS + (z4 * 4 + 1) * VS + ci;
          d[0] = v[it].x; d[VS] = v[it].y; d[2 * VS] = v[it].z; d[3 * VS] = v[it].w;
        }
      }
    }
    __syncthreads();

    // ---- 27 taps x CH/2 k-pairs x NACC accumulators ---------------------------------------------
    const float* wq = wp + (((long)p * 27 * 2 + kh) * 32 + vi) * H2;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const int kz = t / 9, ky = (t / 3) % 3, kx = t % 3;
      const int toff = (ky * HX + kx) * PS + kz * VS;
      float bf[H2];
#pragma unroll
      for (int s4 = 0; s4 < H2 / 4; ++s4) {
        const float4 w4 = *reinterpret_cast<const float4*>(wq + (long)t * 2 * 32 * H2 + s4 * 4);
        bf[s4 * 4 + 0] = w4.x; bf[s4 * 4 + 1] = w4.y; bf[s4 * 4 + 2] = w4.z; bf[s4 * 4 + 3] = w4.w;
      }
      float af[NACC][H2];
#pragma unroll
      for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int s4 = 0; s4 < H2 / 4; ++s4) {
          const float4 a4 = *reinterpret_cast<const float4*>(lds + abase[a] + toff + s4 * 4);
          af[a][s4 * 4 + 0] = a4.x; af[a][s4 * 4 + 1] = a4.y;
          af[a][s4 * 4 + 2] = a4.z; af[a][s4 * 4 + 3] = a4.w;
        }
#pragma unroll
      for (int s = 0; s < H2; ++s)
#pragma unroll
        for (int a = 0; a < NACC; ++a)
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][s], bf[s], acc[a], 0, 0, 0);
    }
  }

  // ---- epilogue: BN(eval) + ReLU, one 128-byte row per voxel ---------------------------------------
  const float sc = scale[vi], sh = shift[vi];
  float* outb = out + (long)b * out_sb;
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    const int rt = wave * NACC + a;
    const int ty = rt / G::TXG, txg = rt % G::TXG;
    const int gy = y0 + ty;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
      const int gx = x0 + txg * G::PX + row / Z, z = row % Z;
      float v = fmaf(acc[a][r], sc, sh);
      if (relu) v = fmaxf(v, 0.f);
      if (gy < Y && gx < X) outb[gy * out_sy + gx * out_sx + z * 32 + vi] = v;
    }
  }
}

// (Cout=32, Cin, 3, 3, 3) torch weight -> packed[p][t][kh][co][CH/2], channel = p*CH + kh*CH/2 + s,
// tap t = (kz*3 + ky)*3 + kx.
__global__ void conv3d_pack_weight_kernel(const float* __restrict__ w, float* __restrict__ packed,
                                          int Cin, int CH) {
  const int n = 32 * Cin * 27;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int H2 = CH / 2;
  int r = idx;
  const int s = r % H2; r /= H2;
  const int co = r % 32; r /= 32;
  const int kh = r % 2; r /= 2;
  const int t = r % 27;
  const int p = r / 27;
  const int ci = p * CH + kh * H2 + s;
  packed[idx] = w[((long)co * Cin + ci) * 27 + t];
}

// ------------------------------------------------------------------------------------------------------
// bf16x3 variant (default): same decomposition, but the K contraction runs on the bf16 matrix cores with every
// f32 operand split into hi + lo bf16 (x = hi + lo to 16 mantissa bits) and  a.w ~= al.wh + ah.wl + ah.wh
// accumulated in f32 — three 32-cycle v_mfma_f32_32x32x16_bf16 per tap and 16 channels instead of eight 64-cycle
// v_mfma_f32_32x32x2_f32: 5.3x less matrix-pipe time at a product error <= 2^-16 (the same trade as
// linear_bf16x3.hip; gfx950 has no xf32 instruction).  CH = 16 channels per phase; a halo voxel slot holds
// [16 hi | 16 lo] bf16 + 16 B pad = the same 80 bytes as the f32 kernel's 16 + 4 floats, so the LDS geometry and
// its conflict-free ds_read_b128 pattern carry over; lanes 0-31 / 32-63 supply channels 0-7 / 8-15 of a voxel.
// Weights: packed[phase][tap][hi, lo][k half][co][8] bf16 — one 16-byte load per lane, plane and tap.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void conv3d_pack_weight_bf16x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ packed,
                                                 int Cin) {
  const int n = 32 * Cin * 27;                                 // one (hi, lo) pair per thread
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  int r = idx;
  const int j = r % 8; r /= 8;
  const int co = r % 32; r /= 32;
  const int kh = r % 2; r /= 2;
  const int t = r % 27;
  const int p = r / 27;
  const int ci = p * 16 + kh * 8 + j;
  const float x = w[((long)co * Cin + ci) * 27 + t];
  const unsigned short hi = bf16_rne(x);
  const unsigned short lo = bf16_rne(x - __uint_as_float((unsigned)hi << 16));
  const long base = (((long)(p * 27 + t) * 2) * 2 + kh) * 32 * 8 + co * 8 + j;     // plane 0 (hi)
  packed[base] = hi;
  packed[base + 2 * 32 * 8] = lo;                                                    // plane 1 (lo)
}

// ---- the 27 taps of one 16-channel pass, software-pipelined and PINNED -------------------------------------------------
// As plain loops hipcc sinks every weight load and every fragment read to its use: one or two requests in flight, an
// `s_waitcnt vmcnt(0)` in front of each tap's MFMAs (round 4 ISA reading: the matrix pipe 28 % busy).  Here the hi / lo weight
// fragments of a tap (2 x 1 KB per wave) come through a 4-slot register ring of buffer loads requested THREE taps ahead, the
// operand fragments of tap t + 1 are read from LDS under the MFMAs of tap t, and the issue order is fixed per tap with
// sched_group_barrier (DS NACC . MFMA NACC . DS NACC . MFMA NACC . VMEM 2 . MFMA NACC), the whole loop fenced with
// sched_barrier so that neighbouring code cannot be matched into the groups.  The ring runs across passes: a pass is 27 taps
// + 1 bubble = 28 ring steps, so slot (t & 3) lines up in every pass; the requests of the last three steps are the first three
// taps of the NEXT pass (clamped to the last tap of the weight when there is none).  TR: the MFMA operands swapped
// (D = W . A^T: conv3d_heads_x3_kernel).  LO = byte distance of the lo plane inside a halo voxel slot.
template <int NACC, bool TR, int HX, int PSB, int VSB, int LO>
__device__ __forceinline__ void conv_taps27(f32x16 (&acc)[NACC], const char* ldsb, const int (&abase)[NACC],
                                            occ_u32x4 (&w)[4][2], const __amdgpu_buffer_rsrc_t wr, const int wv,
                                            const int T0, const int Tlast) {
#define OCC_CV_TOFF(t) (((((t) / 3) % 3) * HX + (t) % 3) * PSB + ((t) / 9) * VSB)
#define OCC_CV_REQ(SLOT, TN)                                                                        \
  {                                                                                                \
    const int tn_ = (TN) < Tlast ? (TN) : Tlast;                                                   \
    w[SLOT][0] = __builtin_amdgcn_raw_buffer_load_b128(wr, wv, tn_ * 2048, 0);                     \
    w[SLOT][1] = __builtin_amdgcn_raw_buffer_load_b128(wr, wv, tn_ * 2048 + 1024, 0);              \
  }
  bf16x8 ah[2][NACC], al[2][NACC];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int a = 0; a < NACC; ++a) al[0][a] = *reinterpret_cast<const bf16x8*>(ldsb + abase[a] + OCC_CV_TOFF(0) + LO);
#pragma unroll
  for (int a = 0; a < NACC; ++a) ah[0][a] = *reinterpret_cast<const bf16x8*>(ldsb + abase[a] + OCC_CV_TOFF(0));
  __builtin_amdgcn_sched_group_barrier(0x100, 2 * NACC, 0);
#pragma unroll
  for (int t = 0; t < 27; ++t) {
    if (t + 1 < 27) {
#pragma unroll
      for (int a = 0; a < NACC; ++a)
        al[(t + 1) & 1][a] = *reinterpret_cast<const bf16x8*>(ldsb + abase[a] + OCC_CV_TOFF(t + 1) + LO);
#pragma unroll
      for (int a = 0; a < NACC; ++a)
        ah[(t + 1) & 1][a] = *reinterpret_cast<const bf16x8*>(ldsb + abase[a] + OCC_CV_TOFF(t + 1));
    }
    if (t + 3 != 27) OCC_CV_REQ((t + 3) & 3, t + 3 <= 26 ? T0 + t + 3 : T0 + 27 + (t + 3 - 28))
    const bf16x8 wh = __builtin_bit_cast(bf16x8, w[t & 3][0]), wl = __builtin_bit_cast(bf16x8, w[t & 3][1]);
#pragma unroll
    for (int a = 0; a < NACC; ++a)
      acc[a] = TR ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, al[t & 1][a], acc[a], 0, 0, 0)
                  : __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t & 1][a], wh, acc[a], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < NACC; ++a)
      acc[a] = TR ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, ah[t & 1][a], acc[a], 0, 0, 0)
                  : __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t & 1][a], wl, acc[a], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < NACC; ++a)
      acc[a] = TR ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, ah[t & 1][a], acc[a], 0, 0, 0)
                  : __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t & 1][a], wh, acc[a], 0, 0, 0);
    if (t + 1 < 27) __builtin_amdgcn_sched_group_barrier(0x100, NACC, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NACC, 0);
    if (t + 1 < 27) __builtin_amdgcn_sched_group_barrier(0x100, NACC, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NACC, 0);
    if (t + 3 != 27) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NACC, 0);
  }
  OCC_CV_REQ(2, T0 + 27 + 2)                          // the bubble step: tap 2 of the next pass
  __builtin_amdgcn_sched_barrier(0);
#undef OCC_CV_REQ
#undef OCC_CV_TOFF
}

template <int Z, int TY, int TX, int LAYOUT>
__global__ __launch_bounds__(256) void conv3d_bf16x3_kernel(
    const float* __restrict__ in, const uint4* __restrict__ wp, const float* __restrict__ scale,
    const float* __restrict__ shift, float* __restrict__ out, int Y, int X, int Cin, long out_sb,
    long out_sy, long out_sx, int relu, int tiles_x, int tiles_y) {
  constexpr int CH = 16;
  using G = ConvGeom<Z, CH, TY, TX>;
  constexpr int VSB = G::VS * 4, PSB = G::PS * 4, HX = G::HX, HY = G::HY, NACC = G::NACC;   // bytes
  extern __shared__ __attribute__((aligned(16))) char ldsb[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware tile order: hardware block b runs on XCD b % 8, each with a private L2.  Dealt linearly, the eight tiles
  // of a row segment land on eight different L2s and every XCD fetches its own copy of the shared halos (PMC, round 2:
  // 306 MB read for an 82 MB input).  Here XCD x walks the contiguous tile range [x*q + min(x, r), ...) — neighbours in
  // space are neighbours in time on ONE L2 (bijective for any grid size: q = n / 8, r = n % 8).
  int bid;
  {
    const int n = (int)gridDim.x, q = n >> 3, r = n & 7, x = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
    bid = x * q + (x < r ? x : r) + j;
  }
  const int tx_i = bid % tiles_x;
  bid /= tiles_x;
  const int ty_i = bid % tiles_y;
  const int b = bid / tiles_y;
  const int y0 = ty_i * TY, x0 = tx_i * TX;
  const float* inb = in + (long)b * Y * X * Z * Cin;

  // z-halo slots (z = -1 and z = Z) of every halo pillar stay zero for all phases
  for (int i = tid; i < HY * HX * 2 * (VSB / 16); i += 256) {
    const int pil = i / (2 * (VSB / 16)), rem = i % (2 * (VSB / 16));
    const int part = rem % (VSB / 16);
    *reinterpret_cast<uint4*>(ldsb + pil * PSB + (rem >= VSB / 16 ? (Z + 1) * VSB : 0) + part * 16) =
        make_uint4(0u, 0u, 0u, 0u);
  }

  f32x16 acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  const int vi = lane & 31, kh = lane >> 5;
  int abase[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    const int rt = wave * NACC + a;
    const int ty = rt / G::TXG, txg = rt % G::TXG;
    const int px = txg * G::PX + vi / Z, z = vi % Z;
    abase[a] = (ty * HX + px) * PSB + z * VSB + kh * 16;       // tap (0,0,0) = halo corner (-1,-1,-1), hi plane
  }

  const int nph = Cin / CH;
  // weight ring (conv_taps27): the first three taps are requested before anything else
  occ_u32x4 w[4][2];
  const __amdgpu_buffer_rsrc_t wr = uniform_rsrc(wp, (unsigned)nph * 27u * 2048u);
  const int wv = lane * 16, Tlast = nph * 27 - 1;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int tn = t < Tlast ? t : Tlast;
    w[t][0] = __builtin_amdgcn_raw_buffer_load_b128(wr, wv, tn * 2048, 0);
    w[t][1] = __builtin_amdgcn_raw_buffer_load_b128(wr, wv, tn * 2048 + 1024, 0);
  }
  for (int p = 0; p < nph; ++p) {
    if (p) block_lds_sync();  // every wave is done reading the previous phase's halo
    // ---- stage 16 channels of the halo, split into hi / lo bf16 -------------------------------------
    if (LAYOUT == 0) {
      constexpr int PARTS = CH / 4, ITEMS = HY * HX * Z * PARTS, ITERS = (ITEMS + 255) / 256;
      float4 v[ITERS];
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int idx = tid + it * 256;
        const int part = idx % PARTS, z = (idx / PARTS) % Z, pil = idx / (PARTS * Z);
        const int gy = y0 + pil / HX - 1, gx = x0 + pil % HX - 1;
        v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < ITEMS && gy >= 0 && gy < Y && gx >= 0 && gx < X)
          v[it] = *reinterpret_cast<const float4*>(inb + (((long)gy * X + gx) * Z + z) * Cin +
                                                   p * CH + part * 4);
      }
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int idx = tid + it * 256;
        const int part = idx % PARTS, z = (idx / PARTS) % Z, pil = idx / (PARTS * Z);
        if (idx < ITEMS) {
          const unsigned h01 = pack_bf16x2_rne(v[it].x, v[it].y), h23 = pack_bf16x2_rne(v[it].z, v[it].w);
          const unsigned l01 = pack_bf16x2_rne(v[it].x - __uint_as_float(h01 << 16),
                                               v[it].y - __uint_as_float(h01 & 0xffff0000u));
          const unsigned l23 = pack_bf16x2_rne(v[it].z - __uint_as_float(h23 << 16),
                                               v[it].w - __uint_as_float(h23 & 0xffff0000u));
          char* d = ldsb + pil * PSB + (z + 1) * VSB + part * 8;
          *reinterpret_cast<uint2*>(d) = make_uint2(h01, h23);
          *reinterpret_cast<uint2*>(d + 32) = make_uint2(l01, l23);
        }
      }
    } else {
      constexpr int Z4 = Z / 4, ITEMS = HY * HX * CH * Z4, ITERS = (ITEMS + 255) / 256;
      float4 v[ITERS];
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int idx = tid + it * 256;
        const int z4 = idx % Z4, ci = (idx / Z4) % CH, pil = idx / (Z4 * CH);
        const int gy = y0 + pil / HX - 1, gx = x0 + pil % HX - 1;
        v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < ITEMS && gy >= 0 && gy < Y && gx >= 0 && gx < X)
          v[it] = *reinterpret_cast<const float4*>(inb + ((long)gy * X + gx) * Z * Cin +
                                                   (long)(p * CH + ci) * Z + z4 * 4);
      }
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int idx = tid + it * 256;
        const int z4 = idx % Z4, ci = (idx / Z4) % CH, pil = idx / (Z4 * CH);
        if (idx < ITEMS) {
          const float f[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
          char* d = ldsb + pil * PSB + (z4 * 4 + 1) * VSB + ci * 2;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const unsigned short hi = bf16_rne(f[q]);
            const unsigned short lo = bf16_rne(f[q] - __uint_as_float((unsigned)hi << 16));
            *reinterpret_cast<unsigned short*>(d + q * VSB) = hi;
            *reinterpret_cast<unsigned short*>(d + q * VSB + 32) = lo;
          }
        }
      }
    }
    block_lds_sync();

    // ---- 27 taps x NACC accumulators x 3 MFMAs (small terms first, term-major), weights through the ring ---------------
    conv_taps27<NACC, false, HX, PSB, VSB, 32>(acc, ldsb, abase, w, wr, wv, p * 27, Tlast);
  }

  // ---- epilogue: BN(eval) + ReLU, one 128-byte row per voxel ---------------------------------------
  const float sc = scale[vi], sh = shift[vi];
  float* outb = out + (long)b * out_sb;
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    const int rt = wave * NACC + a;
    const int ty = rt / G::TXG, txg = rt % G::TXG;
    const int gy = y0 + ty;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
      const int gx = x0 + txg * G::PX + row / Z, z = row % Z;
      float v = fmaf(acc[a][r], sc, sh);
      if (relu) v = fmaxf(v, 0.f);
      if (gy < Y && gx < X) outb[gy * out_sy + gx * out_sx + z * 32 + vi] = v;
    }
  }
}

// ---- Cin = 8 on the bf16 matrix cores: TWO TAPS per MFMA step ---------------------------------------------------------
// BASELINE configs[4] (400 x 400 x 32 voxels, pillar_h = 32) lifts 256 / 32 = 8 channels per voxel: too few for the
// 16-k step of v_mfma_f32_32x32x16_bf16 as conv3d_bf16x3_kernel feeds it (lane half = channel half), so round 3 sent this
// convolution to the exact-f32 kernel (108 64-cycle MFMAs per row tile: 0.685 ms).  Here the two halves of the k-step
// are two TAPS: lanes 0-31 contract the 8 channels of tap 2 s, lanes 32-63 those of tap 2 s + 1 — 14 steps for the 27
// taps (the 28th is zero weights), 42 32-cycle MFMAs per row tile.  A halo voxel slot is [8 hi | 8 lo] bf16 + 16 B pad =
// the f32 kernel's CH = 8 geometry (48 bytes: conflict-free ds_read_b128 at 12-bank strides); a lane half adds its own
// tap's compile-time offset.  Weights: packed[step][hi, lo][lane half = tap parity][co][8] bf16.
__global__ void conv3d_pack_weight_bf16x3_c8_kernel(const float* __restrict__ w, unsigned short* __restrict__ packed) {
  const int n = 14 * 2 * 32 * 8;                               // one (hi, lo) pair per thread
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  int r = idx;
  const int j = r % 8; r /= 8;
  const int co = r % 32; r /= 32;
  const int kh = r % 2;
  const int s = r / 2;
  const int t = 2 * s + kh;
  const float x = t < 27 ? w[((long)co * 8 + j) * 27 + t] : 0.f;
  const unsigned short hi = bf16_rne(x);
  const unsigned short lo = bf16_rne(x - __uint_as_float((unsigned)hi << 16));
  const long base = (((long)s * 2) * 2 + kh) * 32 * 8 + co * 8 + j;                 // plane 0 (hi)
  packed[base] = hi;
  packed[base + 2 * 32 * 8] = lo;                                                    // plane 1 (lo)
}

template <int Z, int TY, int TX, int LAYOUT>
__global__ __launch_bounds__(256) void conv3d_bf16x3_c8_kernel(
    const float* __restrict__ in, const uint4* __restrict__ wp, const float* __restrict__ scale,
    const float* __restrict__ shift, float* __restrict__ out, int Y, int X, long out_sb, long out_sy, long out_sx,
    int relu, int tiles_x, int tiles_y) {
  constexpr int CH = 8, Cin = 8;
  using G = ConvGeom<Z, CH, TY, TX>;
  constexpr int VSB = G::VS * 4, PSB = G::PS * 4, HX = G::HX, HY = G::HY, NACC = G::NACC;   // bytes
  extern __shared__ __attribute__((aligned(16))) char ldsb[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid;                                    // XCD-aware tile order (see conv3d_bf16x3_kernel)
  {
    const int n = (int)gridDim.x, q = n >> 3, r = n & 7, x = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
    bid = x * q + (x < r ? x : r) + j;
  }
  const int tx_i = bid % tiles_x;
  bid /= tiles_x;
  const int ty_i = bid % tiles_y;
  const int b = bid / tiles_y;
  const int y0 = ty_i * TY, x0 = tx_i * TX;
  const float* inb = in + (long)b * Y * X * Z * Cin;
  // weight ring: 14 steps x (hi, lo) x 1 KB, requested three steps ahead (see conv_taps27)
  occ_u32x4 w[4][2];
  const __amdgpu_buffer_rsrc_t wr = uniform_rsrc(wp, 14u * 2048u);
  const int wv = lane * 16;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    w[s][0] = __builtin_amdgcn_raw_buffer_load_b128(wr, wv, s * 2048, 0);
    w[s][1] = __builtin_amdgcn_raw_buffer_load_b128(wr, wv, s * 2048 + 1024, 0);
  }

  for (int i = tid; i < HY * HX * 2 * (VSB / 16); i += 256) {          // z-halo slots stay zero
    const int pil = i / (2 * (VSB / 16)), rem = i % (2 * (VSB / 16));
    const int part = rem % (VSB / 16);
    *reinterpret_cast<uint4*>(ldsb + pil * PSB + (rem >= VSB / 16 ? (Z + 1) * VSB : 0) + part * 16) =
        make_uint4(0u, 0u, 0u, 0u);
  }
  // ---- stage the 8 channels of the halo, split into hi / lo bf16 (one phase) --------------------------------------
  // (all requests first, then the conversions: as one loop hipcc waits for every load before it issues the next)
  if (LAYOUT == 0) {
    constexpr int PARTS = CH / 4, ITEMS = HY * HX * Z * PARTS, ITERS = (ITEMS + 255) / 256;
    float4 v[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int idx = tid + it * 256;
      const int part = idx % PARTS, z = (idx / PARTS) % Z, pil = idx / (PARTS * Z);
      const int gy = y0 + pil / HX - 1, gx = x0 + pil % HX - 1;
      v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < ITEMS && gy >= 0 && gy < Y && gx >= 0 && gx < X)
        v[it] = *reinterpret_cast<const float4*>(inb + (((long)gy * X + gx) * Z + z) * Cin + part * 4);
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int idx = tid + it * 256;
      const int part = idx % PARTS, z = (idx / PARTS) % Z, pil = idx / (PARTS * Z);
      if (idx < ITEMS) {
        const float4 u = v[it];
        const unsigned h01 = pack_bf16x2_rne(u.x, u.y), h23 = pack_bf16x2_rne(u.z, u.w);
        const unsigned l01 = pack_bf16x2_rne(u.x - __uint_as_float(h01 << 16), u.y - __uint_as_float(h01 & 0xffff0000u));
        const unsigned l23 = pack_bf16x2_rne(u.z - __uint_as_float(h23 << 16), u.w - __uint_as_float(h23 & 0xffff0000u));
        char* d = ldsb + pil * PSB + (z + 1) * VSB + part * 8;
        *reinterpret_cast<uint2*>(d) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(d + 16) = make_uint2(l01, l23);
      }
    }
  } else {
    constexpr int Z4 = Z / 4, ITEMS = HY * HX * CH * Z4, ITERS = (ITEMS + 255) / 256;
    float4 v[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int idx = tid + it * 256;
      const int z4 = idx % Z4, ci = (idx / Z4) % CH, pil = idx / (Z4 * CH);
      const int gy = y0 + pil / HX - 1, gx = x0 + pil % HX - 1;
      v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < ITEMS && gy >= 0 && gy < Y && gx >= 0 && gx < X)
        v[it] = *reinterpret_cast<const float4*>(inb + ((long)gy * X + gx) * Z * Cin + (long)ci * Z + z4 * 4);
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int idx = tid + it * 256;
      const int z4 = idx % Z4, ci = (idx / Z4) % CH, pil = idx / (Z4 * CH);
      if (idx < ITEMS) {
        const float f[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
        char* d = ldsb + pil * PSB + (z4 * 4 + 1) * VSB + ci * 2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned short hi = bf16_rne(f[q]);
          const unsigned short lo = bf16_rne(f[q] - __uint_as_float((unsigned)hi << 16));
          *reinterpret_cast<unsigned short*>(d + q * VSB) = hi;
          *reinterpret_cast<unsigned short*>(d + q * VSB + 16) = lo;
        }
      }
    }
  }
  block_lds_sync();

  f32x16 acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  const int vi = lane & 31, kh = lane >> 5;
  int abase[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    const int rt = wave * NACC + a;
    const int ty = rt / G::TXG, txg = rt % G::TXG;
    const int px = txg * G::PX + vi / Z, z = vi % Z;
    abase[a] = (ty * HX + px) * PSB + z * VSB;                 // tap (0,0,0) = halo corner (-1,-1,-1), hi plane
  }
  // ---- 14 steps (tap pairs) x NACC accumulators x 3 MFMAs (small terms first, term-major), pinned like conv_taps27 -------
  constexpr auto toff_of = [](int t) { return ((((t / 3) % 3) * HX + t % 3) * PSB + (t / 9) * VSB); };
  int fo[NACC];                                  // this lane half's tap offset is a run-time select of two constants
  bf16x8 ah[2][NACC], al[2][NACC];
  __builtin_amdgcn_sched_barrier(0);
#define OCC_C8_FRAG(BUF, S)                                                                          \
  {                                                                                                \
    const int t0_ = 2 * (S), t1_ = 2 * (S) + 1 < 27 ? 2 * (S) + 1 : 26;   /* the 28th tap has zero weights */ \
    const int toff_ = kh ? toff_of(t1_) : toff_of(t0_);                                            \
    _Pragma("unroll") for (int a = 0; a < NACC; ++a) fo[a] = abase[a] + toff_;                     \
    _Pragma("unroll") for (int a = 0; a < NACC; ++a) al[BUF][a] = *reinterpret_cast<const bf16x8*>(ldsb + fo[a] + 16); \
    _Pragma("unroll") for (int a = 0; a < NACC; ++a) ah[BUF][a] = *reinterpret_cast<const bf16x8*>(ldsb + fo[a]);      \
  }
  OCC_C8_FRAG(0, 0)
  __builtin_amdgcn_sched_group_barrier(0x100, 2 * NACC, 0);
#pragma unroll
  for (int s = 0; s < 14; ++s) {
    if (s + 1 < 14) OCC_C8_FRAG((s + 1) & 1, s + 1)
    if (s + 3 < 14) {
      w[(s + 3) & 3][0] = __builtin_amdgcn_raw_buffer_load_b128(wr, wv, (s + 3) * 2048, 0);
      w[(s + 3) & 3][1] = __builtin_amdgcn_raw_buffer_load_b128(wr, wv, (s + 3) * 2048 + 1024, 0);
    }
    const bf16x8 wh = __builtin_bit_cast(bf16x8, w[s & 3][0]), wl = __builtin_bit_cast(bf16x8, w[s & 3][1]);
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s & 1][a], wh, acc[a], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s & 1][a], wl, acc[a], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s & 1][a], wh, acc[a], 0, 0, 0);
    if (s + 1 < 14) __builtin_amdgcn_sched_group_barrier(0x100, NACC, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NACC, 0);
    if (s + 1 < 14) __builtin_amdgcn_sched_group_barrier(0x100, NACC, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NACC, 0);
    if (s + 3 < 14) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NACC, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
#undef OCC_C8_FRAG

  // ---- epilogue: BN(eval) + ReLU, one 128-byte row per voxel ---------------------------------------
  const float sc = scale[vi], sh = shift[vi];
  float* outb = out + (long)b * out_sb;
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    const int rt = wave * NACC + a;
    const int ty = rt / G::TXG, txg = rt % G::TXG;
    const int gy = y0 + ty;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
      const int gx = x0 + txg * G::PX + row / Z, z = row % Z;
      float v = fmaf(acc[a][r], sc, sh);
      if (relu) v = fmaxf(v, 0.f);
      if (gy < Y && gx < X) outb[gy * out_sy + gx * out_sx + z * 32 + vi] = v;
    }
  }
}

template <int Z, int TY, int TX>
static int launch_conv_x3_c8(const float* in, const void* wp, const float* scale, const float* shift, float* out, int B,
                             int Y, int X, long out_sb, long out_sy, long out_sx, int relu, int layout, hipStream_t st) {
  using G = ConvGeom<Z, 8, TY, TX>;
  const int tiles_x = (X + TX - 1) / TX, tiles_y = (Y + TY - 1) / TY;
  const size_t lds = (size_t)G::LDS_FLOATS * sizeof(float);
  const dim3 grid((unsigned)((long)B * tiles_x * tiles_y));
  const uint4* w4 = reinterpret_cast<const uint4*>(wp);
  hipError_t e;
  if (layout == 0) {
    auto k = conv3d_bf16x3_c8_kernel<Z, TY, TX, 0>;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
      hipLaunchKernelGGL(k, grid, dim3(256), lds, st, in, w4, scale, shift, out, Y, X, out_sb, out_sy, out_sx, relu,
                         tiles_x, tiles_y);
  } else {
    auto k = conv3d_bf16x3_c8_kernel<Z, TY, TX, 1>;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
      hipLaunchKernelGGL(k, grid, dim3(256), lds, st, in, w4, scale, shift, out, Y, X, out_sb, out_sy, out_sx, relu,
                         tiles_x, tiles_y);
  }
  if (e != hipSuccess) {
    set_error("conv3d_bn_relu_bf16x3: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    return OCC_E_LAUNCH;
  }
  OCC_CHECK_LAUNCH("conv3d_bn_relu_bf16x3");
  return OCC_OK;
}

template <int Z, int TY, int TX>
static int launch_conv_x3(const float* in, const void* wp, const float* scale, const float* shift,
                          float* out, int B, int Y, int X, int Cin, long out_sb, long out_sy,
                          long out_sx, int relu, int layout, hipStream_t st) {
  using G = ConvGeom<Z, 16, TY, TX>;
  const int tiles_x = (X + TX - 1) / TX, tiles_y = (Y + TY - 1) / TY;
  const size_t lds = (size_t)G::LDS_FLOATS * sizeof(float);
  const dim3 grid((unsigned)((long)B * tiles_x * tiles_y));
  const uint4* w4 = reinterpret_cast<const uint4*>(wp);
  hipError_t e;
  if (layout == 0) {
    auto k = conv3d_bf16x3_kernel<Z, TY, TX, 0>;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
      hipLaunchKernelGGL(k, grid, dim3(256), lds, st, in, w4, scale, shift, out, Y, X, Cin, out_sb,
                         out_sy, out_sx, relu, tiles_x, tiles_y);
  } else {
    auto k = conv3d_bf16x3_kernel<Z, TY, TX, 1>;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
      hipLaunchKernelGGL(k, grid, dim3(256), lds, st, in, w4, scale, shift, out, Y, X, Cin, out_sb,
                         out_sy, out_sx, relu, tiles_x, tiles_y);
  }
  if (e != hipSuccess) {
    set_error("conv3d_bn_relu_bf16x3: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    return OCC_E_LAUNCH;
  }
  OCC_CHECK_LAUNCH("conv3d_bn_relu_bf16x3");
  return OCC_OK;
}

template <int Z, int CH, int TY, int TX>
static int launch_conv(const float* in, const float* wp, const float* scale, const float* shift,
                       float* out, int B, int Y, int X, int Cin, long out_sb, long out_sy,
                       long out_sx, int relu, int layout, hipStream_t st) {
  using G = ConvGeom<Z, CH, TY, TX>;
  const int tiles_x = (X + TX - 1) / TX, tiles_y = (Y + TY - 1) / TY;
  const size_t lds = (size_t)G::LDS_FLOATS * sizeof(float);
  const dim3 grid((unsigned)((long)B * tiles_x * tiles_y));
  hipError_t e;
  if (layout == 0) {
    auto k = conv3d_mfma_kernel<Z, CH, TY, TX, 0>;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
      hipLaunchKernelGGL(k, grid, dim3(256), lds, st, in, wp, scale, shift, out, Y, X, Cin, out_sb,
                         out_sy, out_sx, relu, tiles_x, tiles_y);
  } else {
    auto k = conv3d_mfma_kernel<Z, CH, TY, TX, 1>;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
      hipLaunchKernelGGL(k, grid, dim3(256), lds, st, in, wp, scale, shift, out, Y, X, Cin, out_sb,
                         out_sy, out_sx, relu, tiles_x, tiles_y);
  }
  if (e != hipSuccess) {
    set_error("conv3d_bn_relu: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    return OCC_E_LAUNCH;
  }
  OCC_CHECK_LAUNCH("conv3d_bn_relu");
  return OCC_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Second decoder convolution + BatchNorm + ReLU + BOTH occupancy heads + the class decode in ONE kernel (SURVEY.md §7
// step 5: "heads fused on the same tile so the 640 000 x 32 activations never round-trip HBM"; reference
// transformer_occ.py:304-321 + bevformer_occ_head.py:210-212).  The convolution is conv3d_bf16x3_kernel<Z, TY, TX, 0>
// with the MFMA operands swapped — D = W . A^T, so a lane holds 16 output channels of ONE voxel: after scale / shift /
// ReLU its registers 0-7 and 8-15, split into hi / lo bf16, ARE the two k-steps of the heads' first contraction
// (k-slot j of lane half g = channel 16 s + 8 (j / 4) + 4 g + j % 4, the order the W1cat fragments are packed in) — no
// LDS transpose, no HBM round trip.  The heads' fragments (32 KB, packed once by conv3d_heads_pack_kernel) replace the
// halo in LDS once the taps are done; the rest is occ_heads_x3_kernel's body (csrc/occ_heads.hip) per 32-voxel tile.
constexpr int kHeadsPackBytes = 16 * 1024 + 16 * 1024 + 128 * 4 + 32 * 4;

// torch.nn.Softplus(beta=1, threshold=20) = max(x, 0) + log(1 + exp(-|x|)), branch-free on the two transcendental
// instructions (v_exp_f32 / v_log_f32 on an argument in (1, 2]: ~1e-7 absolute).  Above the threshold exp(-x) < 2.1e-9 is below
// half an ulp of x, 1 + e rounds to 1 and the expression returns x itself — what torch's cut-off returns.  (As
// `x > 20 ? x : ... __logf(...)` hipcc emitted an exec-mask branch and a 12-instruction refined logarithm per element: the heads
// phase of the fused kernel is VALU-bound.)
__device__ __forceinline__ float cvh_softplus(float x) {
  const float e = __builtin_amdgcn_exp2f(fabsf(x) * -1.44269504088896341f);
  return fmaxf(x, 0.f) + __builtin_amdgcn_logf(1.f + e) * 0.693147180559945309f;
}
__device__ __forceinline__ void cvh_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  hi = pack_bf16x2_rne(x0, x1);
  lo = pack_bf16x2_rne(x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u));
}

// [W1cat fragments [(a*2 + s)*2 + plane][lane][8] | W2cat fragments [(kk*2 + plane)][lane][8] | b1 (128 f32) | b2 (32 f32)]
__global__ void conv3d_heads_pack_kernel(const float* __restrict__ w1o, const float* __restrict__ b1o,
                                         const float* __restrict__ w2o, const float* __restrict__ b2o,
                                         const float* __restrict__ w1f, const float* __restrict__ b1f,
                                         const float* __restrict__ w2f, const float* __restrict__ b2f,
                                         unsigned short* __restrict__ packed, int ncls) {
  constexpr int C = 32, HID = 64;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < 4096) {
    const int j = e & 7, l = (e >> 3) & 63, f = e >> 9;
    const int m = l & 31, gg = l >> 5;
    {   // W1cat: f = a*2 + s; k-slot (s, gg, j) stands for channel 16 s + 8 (j / 4) + 4 gg + j % 4 (the D-register order)
      const int a = f >> 1, sk = f & 1, u = 32 * a + m, k = 16 * sk + 8 * (j >> 2) + 4 * gg + (j & 3);
      const float w = u < HID ? w1o[u * C + k] : w1f[(u - HID) * C + k];
      const unsigned short hi = bf16_rne(w), lo = bf16_rne(w - __uint_as_float((unsigned)hi << 16));
      packed[((f * 2 + 0) * 64 + l) * 8 + j] = hi;
      packed[((f * 2 + 1) * 64 + l) * 8 + j] = lo;
    }
    {   // W2cat: f = kk = 2a + ks; output row m, hidden unit u (occ_heads_x3_kernel's layout)
      const int a = f >> 1, ks = f & 1, u = 32 * a + 16 * ks + 8 * (j >> 2) + 4 * gg + (j & 3);
      float w = 0.f;
      if (m < ncls) { if (u < HID) w = w2o[m * HID + u]; }
      else if (m < ncls + 2) { if (u >= HID) w = w2f[(m - ncls) * HID + (u - HID)]; }
      const unsigned short hi = bf16_rne(w), lo = bf16_rne(w - __uint_as_float((unsigned)hi << 16));
      packed[8192 + ((f * 2 + 0) * 64 + l) * 8 + j] = hi;
      packed[8192 + ((f * 2 + 1) * 64 + l) * 8 + j] = lo;
    }
  }
  float* bp = reinterpret_cast<float*>(packed + 16384);
  if (e < 128) bp[e] = e < HID ? b1o[e] : b1f[e - HID];
  if (e < 32) bp[128 + e] = e < ncls ? b2o[e] : (e < ncls + 2 ? b2f[e - ncls] : 0.f);
}

// NCLS: the class count as a compile-time constant (17: the reference's heads) or 0 = the run-time `ncls`
template <int Z, int TY, int TX, int NCLS>
__global__ __launch_bounds__(256) void conv3d_heads_x3_kernel(
    const float* __restrict__ in, const uint4* __restrict__ wp, const float* __restrict__ scale,
    const float* __restrict__ shift, const uint4* __restrict__ heads_pack, float* __restrict__ occ,
    float* __restrict__ flow, long long* __restrict__ occ_cls, int Y, int X, int ncls, int tiles_x, int tiles_y) {
  constexpr int CH = 16, Cin = 32;
  using G = ConvGeom<Z, CH, TY, TX>;
  constexpr int VSB = G::VS * 4, PSB = G::PS * 4, HX = G::HX, HY = G::HY, NACC = G::NACC;   // bytes
  static_assert(G::LDS_FLOATS * 4 >= kHeadsPackBytes + 4 * 32 * 33 * 4, "the heads' fragments + transposes overlay the halo");
  extern __shared__ __attribute__((aligned(16))) char ldsb[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid;                                    // XCD-aware tile order (see conv3d_bf16x3_kernel)
  {
    const int n = (int)gridDim.x, q = n >> 3, r = n & 7, x = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
    bid = x * q + (x < r ? x : r) + j;
  }
  const int tx_i = bid % tiles_x;
  bid /= tiles_x;
  const int ty_i = bid % tiles_y;
  const int b = bid / tiles_y;
  const int y0 = ty_i * TY, x0 = tx_i * TX;
  const float* inb = in + (long)b * Y * X * Z * Cin;

  for (int i = tid; i < HY * HX * 2 * (VSB / 16); i += 256) {          // z-halo slots stay zero
    const int pil = i / (2 * (VSB / 16)), rem = i % (2 * (VSB / 16));
    const int part = rem % (VSB / 16);
    *reinterpret_cast<uint4*>(ldsb + pil * PSB + (rem >= VSB / 16 ? (Z + 1) * VSB : 0) + part * 16) =
        make_uint4(0u, 0u, 0u, 0u);
  }
  f32x16 acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  const int vi = lane & 31, kh = lane >> 5;
  int abase[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    const int rt = wave * NACC + a;
    const int ty = rt / G::TXG, txg = rt % G::TXG;
    const int px = txg * G::PX + vi / Z, z = vi % Z;
    abase[a] = (ty * HX + px) * PSB + z * VSB + kh * 16;
  }
  occ_u32x4 w[4][2];                          // weight ring (conv_taps27)
  const __amdgpu_buffer_rsrc_t wr = uniform_rsrc(wp, (unsigned)(Cin / CH) * 27u * 2048u);
  const int wv = lane * 16, Tlast = (Cin / CH) * 27 - 1;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    w[t][0] = __builtin_amdgcn_raw_buffer_load_b128(wr, wv, t * 2048, 0);
    w[t][1] = __builtin_amdgcn_raw_buffer_load_b128(wr, wv, t * 2048 + 1024, 0);
  }
#pragma unroll
  for (int p = 0; p < Cin / CH; ++p) {
    if (p) block_lds_sync();
    {
      constexpr int PARTS = CH / 4, ITEMS = HY * HX * Z * PARTS, ITERS = (ITEMS + 255) / 256;
      float4 v[ITERS];
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int idx = tid + it * 256;
        const int part = idx % PARTS, z = (idx / PARTS) % Z, pil = idx / (PARTS * Z);
        const int gy = y0 + pil / HX - 1, gx = x0 + pil % HX - 1;
        v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < ITEMS && gy >= 0 && gy < Y && gx >= 0 && gx < X)
          v[it] = *reinterpret_cast<const float4*>(inb + (((long)gy * X + gx) * Z + z) * Cin + p * CH + part * 4);
      }
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int idx = tid + it * 256;
        const int part = idx % PARTS, z = (idx / PARTS) % Z, pil = idx / (PARTS * Z);
        if (idx < ITEMS) {
          unsigned h01, h23, l01, l23;
          cvh_split2(v[it].x, v[it].y, h01, l01);
          cvh_split2(v[it].z, v[it].w, h23, l23);
          char* d = ldsb + pil * PSB + (z + 1) * VSB + part * 8;
          *reinterpret_cast<uint2*>(d) = make_uint2(h01, h23);
          *reinterpret_cast<uint2*>(d + 32) = make_uint2(l01, l23);
        }
      }
    }
    block_lds_sync();
    conv_taps27<NACC, true, HX, PSB, VSB, 32>(acc, ldsb, abase, w, wr, wv, p * 27, Tlast);
  }

  // ---- the heads' operands replace the halo -------------------------------------------------------------------------
  __syncthreads();
  for (int i = tid; i < kHeadsPackBytes / 16; i += 256) reinterpret_cast<uint4*>(ldsb)[i] = heads_pack[i];
  __syncthreads();
  const bf16x8* W1 = reinterpret_cast<const bf16x8*>(ldsb) + lane;
  const bf16x8* W2 = reinterpret_cast<const bf16x8*>(ldsb + 16384) + lane;
  const float* b1s = reinterpret_cast<const float*>(ldsb + 32768);
  const float* b2s = b1s + 128;
  float* sm = reinterpret_cast<float*>(ldsb + kHeadsPackBytes) + wave * (32 * 33);
  // BatchNorm (eval) scale / shift of this lane's 16 output channels: (r & 3) + 8 (r >> 2) + 4 kh
  float4 scv[4], shv[4];
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    scv[q4] = *reinterpret_cast<const float4*>(scale + 8 * q4 + 4 * kh);
    shv[q4] = *reinterpret_cast<const float4*>(shift + 8 * q4 + 4 * kh);
  }
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    // conv tile a of this wave: 32 voxels = PX pillars x Z heights; lane's voxel vi
    uint4 xh[2], xl[2];
    {
      float f[16];
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        f[4 * q4 + 0] = fmaxf(fmaf(acc[a][4 * q4 + 0], scv[q4].x, shv[q4].x), 0.f);
        f[4 * q4 + 1] = fmaxf(fmaf(acc[a][4 * q4 + 1], scv[q4].y, shv[q4].y), 0.f);
        f[4 * q4 + 2] = fmaxf(fmaf(acc[a][4 * q4 + 2], scv[q4].z, shv[q4].z), 0.f);
        f[4 * q4 + 3] = fmaxf(fmaf(acc[a][4 * q4 + 3], scv[q4].w, shv[q4].w), 0.f);
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        cvh_split2(f[8 * s2 + 0], f[8 * s2 + 1], xh[s2].x, xl[s2].x); cvh_split2(f[8 * s2 + 2], f[8 * s2 + 3], xh[s2].y, xl[s2].y);
        cvh_split2(f[8 * s2 + 4], f[8 * s2 + 5], xh[s2].z, xl[s2].z); cvh_split2(f[8 * s2 + 6], f[8 * s2 + 7], xh[s2].w, xl[s2].w);
      }
    }
    f32x16 h[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) h[t][r] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const bf16x8 bxh = __builtin_bit_cast(bf16x8, xh[s2]), bxl = __builtin_bit_cast(bf16x8, xl[s2]);
#pragma unroll
      for (int t = 0; t < 4; ++t) h[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W1[((t * 2 + s2) * 2 + 1) * 64], bxh, h[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) h[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W1[((t * 2 + s2) * 2 + 0) * 64], bxl, h[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) h[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W1[((t * 2 + s2) * 2 + 0) * 64], bxh, h[t], 0, 0, 0);
    }
    f32x16 o0, o1, o2;
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = o1[r] = o2[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float v[16];
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const float4 bb = *reinterpret_cast<const float4*>(b1s + 32 * t + 8 * q4 + 4 * kh);
        const float t0 = h[t][4 * q4 + 0] + bb.x, t1 = h[t][4 * q4 + 1] + bb.y;
        const float t2 = h[t][4 * q4 + 2] + bb.z, t3 = h[t][4 * q4 + 3] + bb.w;
        if (t < 2) {
          v[4 * q4 + 0] = cvh_softplus(t0); v[4 * q4 + 1] = cvh_softplus(t1);
          v[4 * q4 + 2] = cvh_softplus(t2); v[4 * q4 + 3] = cvh_softplus(t3);
        } else {
          v[4 * q4 + 0] = fmaxf(t0, 0.f); v[4 * q4 + 1] = fmaxf(t1, 0.f);
          v[4 * q4 + 2] = fmaxf(t2, 0.f); v[4 * q4 + 3] = fmaxf(t3, 0.f);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        uint4 hh, hl;
        cvh_split2(v[8 * ks + 0], v[8 * ks + 1], hh.x, hl.x); cvh_split2(v[8 * ks + 2], v[8 * ks + 3], hh.y, hl.y);
        cvh_split2(v[8 * ks + 4], v[8 * ks + 5], hh.z, hl.z); cvh_split2(v[8 * ks + 6], v[8 * ks + 7], hh.w, hl.w);
        const int kk = 2 * t + ks;
        const bf16x8 wh = W2[(kk * 2 + 0) * 64], wl = W2[(kk * 2 + 1) * 64];
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, __builtin_bit_cast(bf16x8, hh), o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, __builtin_bit_cast(bf16x8, hl), o1, 0, 0, 0);
        o2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, __builtin_bit_cast(bf16x8, hh), o2, 0, 0, 0);
      }
    }
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const float4 bb = *reinterpret_cast<const float4*>(b2s + 8 * q4 + 4 * kh);
      const float bq[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * q4 + i;
        sm[vi * 33 + 8 * q4 + 4 * kh + i] = ((o0[r] + o1[r]) + o2[r]) + bq[i];
      }
    }
    wave_lds_sync();
    // output rows: voxel v of the tile = pillar (gy, gx0 + v / Z), height v % Z; outputs are (B, X, Y, Z, .)
    const int rt = wave * NACC + a;
    const int gy = y0 + rt / G::TXG, gx0 = x0 + (rt % G::TXG) * G::PX;
    if (gy < Y) {
      // a pillar's Z x ncls logits are one contiguous, 16-byte aligned run of the output (Z % 4 == 0): quad stores,
      // element -> (height, class) by a constant division when NCLS is known
      const int nc = NCLS ? NCLS : ncls;
      const int quads = Z * nc / 4;                     // per pillar
      for (int f = lane; f < (32 / Z) * quads; f += 64) {
        const int pl = f / quads, e0 = 4 * (f - pl * quads), gx = gx0 + pl;
        float q4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int e = e0 + i, z = e / nc, ch = e - z * nc;
          q4[i] = sm[(pl * Z + z) * 33 + ch];
        }
        if (gx < X)
          *reinterpret_cast<float4*>(occ + (((long)b * X + gx) * Y + gy) * Z * nc + e0) = make_float4(q4[0], q4[1], q4[2], q4[3]);
      }
      {
        const int v = lane >> 1, gx = gx0 + v / Z;
        if (gx < X) flow[((((long)b * X + gx) * Y + gy) * Z + v % Z) * 2 + (lane & 1)] = sm[v * 33 + ncls + (lane & 1)];
      }
      if (occ_cls != nullptr && lane < 32) {      // decode: argmax of the logits, first index on ties, 0 for a NaN row
        const int gx = gx0 + lane / Z;
        float best = sm[lane * 33];
        int arg = 0;
        bool nan = best != best;
        for (int ch = 1; ch < ncls; ++ch) {
          const float x = sm[lane * 33 + ch];
          nan |= x != x;
          if (x > best) { best = x; arg = ch; }
        }
        if (gx < X) occ_cls[(((long)b * X + gx) * Y + gy) * Z + lane % Z] = nan ? 0 : arg;
      }
    }
    wave_lds_sync();
  }
}

}  // namespace occ

extern "C" int occ_conv3d_channel_block(int Cin) { return Cin % 16 == 0 ? 16 : (Cin % 8 == 0 ? 8 : 0); }

extern "C" int occ_conv3d_pack_weight_f32(const float* weight, float* packed, int Cin, int Cout,
                                          void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(weight && packed, "conv3d_pack_weight: null pointer argument");
  if (Cout != 32 || occ_conv3d_channel_block(Cin) == 0) {
    set_error("conv3d_pack_weight: no MFMA kernel for Cin=%d Cout=%d (need Cout=32, Cin %% 8 == 0)",
              Cin, Cout);
    return OCC_E_UNSUPPORTED;
  }
  const int n = 32 * Cin * 27;
  hipLaunchKernelGGL(conv3d_pack_weight_kernel, dim3((n + 255) / 256), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), weight, packed, Cin,
                     occ_conv3d_channel_block(Cin));
  OCC_CHECK_LAUNCH("conv3d_pack_weight");
  return OCC_OK;
}

extern "C" int occ_conv3d_bn_relu_f32(const float* in, const float* w_packed, const float* scale,
                                      const float* shift, float* out, int B, int Z, int Y, int X,
                                      int Cin, int Cout, int in_layout, int64_t out_stride_b,
                                      int64_t out_stride_y, int64_t out_stride_x, int relu,
                                      void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(in && w_packed && scale && shift && out, "conv3d_bn_relu: null pointer argument");
  OCC_CHECK_ARG(B > 0 && Z > 0 && Y > 0 && X > 0 && Cin > 0 && Cout > 0,
                "conv3d_bn_relu: bad dimension (B=%d Z=%d Y=%d X=%d Cin=%d Cout=%d)", B, Z, Y, X, Cin,
                Cout);
  OCC_CHECK_ARG(in_layout == 0 || in_layout == 1, "conv3d_bn_relu: in_layout must be 0 or 1");
  OCC_CHECK_ARG((long)Y * X * Z * (Cin > Cout ? Cin : Cout) < (1L << 31),
                "conv3d_bn_relu: one batch entry exceeds 2^31 elements");
  const int CH = occ_conv3d_channel_block(Cin);
  if (Cout != 32 || CH == 0 || !(Z == 4 || Z == 8 || Z == 16 || Z == 32)) {
    set_error("conv3d_bn_relu: no MFMA kernel for Z=%d Cin=%d Cout=%d", Z, Cin, Cout);
    return OCC_E_UNSUPPORTED;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define OCC_CONV_CASE(ZZ, CC, TTY, TTX)                                                            \
  if (Z == ZZ && CH == CC)                                                                         \
    return launch_conv<ZZ, CC, TTY, TTX>(in, w_packed, scale, shift, out, B, Y, X, Cin,            \
                                         (long)out_stride_b, (long)out_stride_y,                   \
                                         (long)out_stride_x, relu, in_layout, st);
  OCC_CONV_CASE(16, 16, 2, 8)
  OCC_CONV_CASE(16, 8, 2, 8)
  OCC_CONV_CASE(32, 16, 2, 4)
  OCC_CONV_CASE(32, 8, 2, 4)
  OCC_CONV_CASE(8, 16, 2, 16)
  OCC_CONV_CASE(8, 8, 2, 16)
  OCC_CONV_CASE(4, 16, 2, 16)
  OCC_CONV_CASE(4, 8, 2, 16)
#undef OCC_CONV_CASE
  set_error("conv3d_bn_relu: no MFMA kernel for Z=%d CH=%d", Z, CH);
  return OCC_E_UNSUPPORTED;
}

extern "C" int occ_conv3d_pack_weight_bf16x3(const float* weight, void* packed, int Cin, int Cout,
                                             void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(weight && packed, "conv3d_pack_weight_bf16x3: null pointer argument");
  if (Cout == 32 && Cin == 8) {          // two taps per MFMA step: 14 steps x (hi, lo) x 2 lane halves x 32 x 8 bf16
    hipLaunchKernelGGL(conv3d_pack_weight_bf16x3_c8_kernel, dim3((14 * 2 * 32 * 8 + 255) / 256), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), weight, reinterpret_cast<unsigned short*>(packed));
    OCC_CHECK_LAUNCH("conv3d_pack_weight_bf16x3");
    return OCC_OK;
  }
  if (Cout != 32 || Cin <= 0 || Cin % 16) {
    set_error("conv3d_pack_weight_bf16x3: no kernel for Cin=%d Cout=%d (need Cout=32, Cin == 8 or Cin %% 16 == 0)", Cin, Cout);
    return OCC_E_UNSUPPORTED;
  }
  const int n = 32 * Cin * 27;
  hipLaunchKernelGGL(conv3d_pack_weight_bf16x3_kernel, dim3((n + 255) / 256), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), weight, reinterpret_cast<unsigned short*>(packed), Cin);
  OCC_CHECK_LAUNCH("conv3d_pack_weight_bf16x3");
  return OCC_OK;
}

extern "C" int occ_conv3d_bn_relu_bf16x3_f32(const float* in, const void* w_packed, const float* scale,
                                             const float* shift, float* out, int B, int Z, int Y, int X,
                                             int Cin, int Cout, int in_layout, int64_t out_stride_b,
                                             int64_t out_stride_y, int64_t out_stride_x, int relu,
                                             void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(in && w_packed && scale && shift && out, "conv3d_bn_relu_bf16x3: null pointer argument");
  OCC_CHECK_ARG(B > 0 && Z > 0 && Y > 0 && X > 0 && Cin > 0 && Cout > 0,
                "conv3d_bn_relu_bf16x3: bad dimension (B=%d Z=%d Y=%d X=%d Cin=%d Cout=%d)", B, Z, Y, X, Cin,
                Cout);
  OCC_CHECK_ARG(in_layout == 0 || in_layout == 1, "conv3d_bn_relu_bf16x3: in_layout must be 0 or 1");
  OCC_CHECK_ARG((long)Y * X * Z * (Cin > Cout ? Cin : Cout) < (1L << 31),
                "conv3d_bn_relu_bf16x3: one batch entry exceeds 2^31 elements");
  if (Cout != 32 || (Cin % 16 && Cin != 8) || !(Z == 4 || Z == 8 || Z == 16 || Z == 32)) {
    set_error("conv3d_bn_relu_bf16x3: no kernel for Z=%d Cin=%d Cout=%d", Z, Cin, Cout);
    return OCC_E_UNSUPPORTED;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (Cin == 8) {
#define OCC_CONV_C8(ZZ, TTY, TTX)                                                                  \
    if (Z == ZZ)                                                                                   \
      return launch_conv_x3_c8<ZZ, TTY, TTX>(in, w_packed, scale, shift, out, B, Y, X, (long)out_stride_b, \
                                             (long)out_stride_y, (long)out_stride_x, relu, in_layout, st);
    OCC_CONV_C8(32, 2, 4)
    OCC_CONV_C8(16, 2, 8)
    OCC_CONV_C8(8, 2, 16)
    OCC_CONV_C8(4, 2, 16)
#undef OCC_CONV_C8
    return OCC_E_UNSUPPORTED;
  }
#define OCC_CONV_CASE(ZZ, TTY, TTX)                                                                \
  if (Z == ZZ)                                                                                     \
    return launch_conv_x3<ZZ, TTY, TTX>(in, w_packed, scale, shift, out, B, Y, X, Cin,             \
                                        (long)out_stride_b, (long)out_stride_y,                    \
                                        (long)out_stride_x, relu, in_layout, st);
  OCC_CONV_CASE(16, 2, 8)
  OCC_CONV_CASE(32, 2, 4)
  OCC_CONV_CASE(8, 2, 16)
  OCC_CONV_CASE(4, 2, 16)
#undef OCC_CONV_CASE
  return OCC_E_UNSUPPORTED;
}

extern "C" int occ_conv3d_heads_pack(const float* w1_occ, const float* b1_occ, const float* w2_occ, const float* b2_occ,
                                     const float* w1_flow, const float* b1_flow, const float* w2_flow,
                                     const float* b2_flow, void* packed, int C, int hidden, int num_classes,
                                     void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(w1_occ && b1_occ && w2_occ && b2_occ && w1_flow && b1_flow && w2_flow && b2_flow && packed,
                "conv3d_heads_pack: null pointer argument");
  if (C != 32 || hidden != 64 || num_classes <= 0 || num_classes + 2 > 32) {
    set_error("conv3d_heads_pack: no fused kernel for C=%d hidden=%d num_classes=%d", C, hidden, num_classes);
    return OCC_E_UNSUPPORTED;
  }
  hipLaunchKernelGGL(conv3d_heads_pack_kernel, dim3(16), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w1_occ,
                     b1_occ, w2_occ, b2_occ, w1_flow, b1_flow, w2_flow, b2_flow,
                     reinterpret_cast<unsigned short*>(packed), num_classes);
  OCC_CHECK_LAUNCH("conv3d_heads_pack");
  return OCC_OK;
}

extern "C" int64_t occ_conv3d_heads_pack_bytes(void) { return occ::kHeadsPackBytes; }

extern "C" int occ_conv3d_heads_decode_bf16x3_f32(const float* in, const void* w_packed, const float* scale,
                                                  const float* shift, const void* heads_packed, float* occ_out,
                                                  float* flow_out, int64_t* occ_cls_out, int B, int Z, int Y, int X,
                                                  int Cin, int num_classes, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(in && w_packed && scale && shift && heads_packed && occ_out && flow_out,
                "conv3d_heads_decode: null pointer argument");
  OCC_CHECK_ARG(B > 0 && Y > 0 && X > 0 && num_classes > 0, "conv3d_heads_decode: bad dimension");
  OCC_CHECK_ARG((long)Y * X * Z * 32 < (1L << 31), "conv3d_heads_decode: one batch entry exceeds 2^31 elements");
  OCC_CHECK_ARG(((uintptr_t)occ_out & 15) == 0, "conv3d_heads_decode: occ_out must be 16-byte aligned (quad stores)");
  if ((Z != 16 && Z != 32) || Cin != 32 || num_classes + 2 > 32) {
    set_error("conv3d_heads_decode: no fused kernel for Z=%d Cin=%d num_classes=%d", Z, Cin, num_classes);
    return OCC_E_UNSUPPORTED;
  }
  // Z = 16: 2 x 8 pillars per block (two pillars per 32-voxel row tile); Z = 32 (BASELINE configs[4]): 2 x 4 pillars,
  // one pillar per row tile — the same 8 row tiles / 2 accumulators per wave, 66 KB of halo
  auto launch = [&](auto k, int TY, int TX, size_t lds) -> int {
    const int tiles_x = (X + TX - 1) / TX, tiles_y = (Y + TY - 1) / TY;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("conv3d_heads_decode: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return OCC_E_LAUNCH;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)((long)B * tiles_x * tiles_y)), dim3(256), lds, reinterpret_cast<hipStream_t>(stream),
                       in, reinterpret_cast<const uint4*>(w_packed), scale, shift,
                       reinterpret_cast<const uint4*>(heads_packed), occ_out, flow_out,
                       reinterpret_cast<long long*>(occ_cls_out), Y, X, num_classes, tiles_x, tiles_y);
    OCC_CHECK_LAUNCH("conv3d_heads_decode");
    return OCC_OK;
  };
  if (Z == 16) {
    if (num_classes == 17)
      return launch(conv3d_heads_x3_kernel<16, 2, 8, 17>, 2, 8, (size_t)ConvGeom<16, 16, 2, 8>::LDS_FLOATS * sizeof(float));
    return launch(conv3d_heads_x3_kernel<16, 2, 8, 0>, 2, 8, (size_t)ConvGeom<16, 16, 2, 8>::LDS_FLOATS * sizeof(float));
  }
  if (num_classes == 17)
    return launch(conv3d_heads_x3_kernel<32, 2, 4, 17>, 2, 4, (size_t)ConvGeom<32, 16, 2, 4>::LDS_FLOATS * sizeof(float));
  return launch(conv3d_heads_x3_kernel<32, 2, 4, 0>, 2, 4, (size_t)ConvGeom<32, 16, 2, 4>::LDS_FLOATS * sizeof(float));
}
