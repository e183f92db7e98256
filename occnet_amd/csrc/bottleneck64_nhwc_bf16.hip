// One whole ResNet bottleneck with 64 mid channels (the three stride-4 blocks of the reference's ResNet-50
// `layer1`: mmdet Bottleneck, style='pytorch') as ONE kernel on NHWC bf16 activations:
//
//   c1  = relu(conv1x1(x,  W1) + b1)                      Cin -> 64          (on the tile's 10 x 18 halo)
//   c2  = relu(conv3x3(c1, W2) + b2)                      64  -> 64, pad 1
//   out = relu(conv1x1(c2, W3) + b3 + identity)           64  -> 256
//   identity = x (Cin = 256)  or  conv1x1(x, Wds) + bds (Cin = 64, first block: folded into the third GEMM
//   as 64 extra K columns, b3 := b3 + bds)
//
// NOT part of the hand-written hot path.  Why it exists: at stride 4 the 6 x 232 x 400
// pixel maps make every layer of the block HBM-bound (rocprofv3 PMC: the separate 1x1 kernels move 1.14 GB
// per block, 0.57 GB of it for 64-channel intermediates and the second read of the identity), so the only
// way down is to keep c1 / c2 in LDS: x is read once (+41 % halo overlap, mostly L2 hits), out written once.
//
// Block = 4 waves x (8 x 16 output pixels).  All three weight matrices are pre-packed in MFMA B-fragment
// order (occ_mfma_pack_b_frag_bf16: [k-step][32-col tile][lane][8 bf16]), so a wave's operand is ONE coalesced
// 1 KB global load per (k-step, column tile) that never touches LDS (L1/L2 hits: 0.2 MB of weights shared
// by every block); LDS carries only activations:
//   phase A  x halo chunk (192 pixel slots x 32 ch, double buffered, 80-byte slots) -> c1 halo (10 x 18 pixel
//            slots of 144 B, zero outside the image = conv2's padding);  GEMM 192 x 64 x Cin, wave = 3 row
//            tiles x 1 column tile
//   phase B  9 taps x 4 k-steps straight out of the c1 halo -> c2 tile (128 x 144 B); wave = 2 x 1 tiles
//   phase C  c2 (+ x centre pixels when downsampling) x W3 -> 128 x 256; wave = 4 row tiles x 2 column tiles,
//            epilogue through the same LDS transpose as conv1x1_nhwc_bf16.hip (bias, identity, ReLU, bf16).
// 75 KB LDS -> two blocks per CU, so one block's loads overlap the other's MFMAs.
#include "common.h"

namespace occ {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kBnTH = 8, kBnTW = 16, kBnHH = 10, kBnHW = 18, kBnNP = kBnHH * kBnHW;   // 180 halo pixels
constexpr int kBnXS = 80;                    // bytes per x-chunk pixel slot (32 bf16 + 16 pad)
constexpr int kBnXA = 192 * kBnXS;           // one x chunk buffer (6 row tiles of 32 halo pixels)
constexpr int kBnMS = 144;                   // bytes per 64-channel pixel slot (128 + 16 pad)
constexpr int kBnH1ROW = kBnHW * kBnMS;      // 2592
constexpr int kBnH1 = kBnHH * kBnH1ROW;      // 25 920
constexpr int kBnC2 = 128 * kBnMS;           // 18 432
constexpr int kBnOLD = 256 + 4;              // epilogue transpose row stride (floats)

__device__ __forceinline__ float bn_bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short bn_f32_to_bf16(float f) { return bf16_rne(f); }

// f32 row-major weight (N, K) -> bf16 MFMA B-fragment order: packed[((ks * N/32 + nt) * 64 + lane) * 8 + j]
// = w[nt*32 + (lane & 31)][ks*16 + (lane >> 5)*8 + j]   (v_mfma_f32_32x32x16_bf16 operand of lane `lane`)
__global__ void mfma_pack_b_frag_kernel(const float* __restrict__ w, unsigned short* __restrict__ packed,
                                        int N, int K) {
  const long n_el = (long)N * K;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_el) return;
  const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
  const long rest = idx >> 9;
  const int nts = N / 32;
  const int nt = (int)(rest % nts), ks = (int)(rest / nts);
  const int n = nt * 32 + (lane & 31), k = ks * 16 + (lane >> 5) * 8 + j;
  packed[idx] = bn_f32_to_bf16(w[(long)n * K + k]);
}

template <int CIN, bool DS>
__global__ __launch_bounds__(256, 2) void bottleneck64_nhwc_bf16_kernel(
    const uint4* __restrict__ x, const uint4* __restrict__ w1p, const float* __restrict__ b1,
    const uint4* __restrict__ w2p, const float* __restrict__ b2, const uint4* __restrict__ w3p,
    const float* __restrict__ b3, unsigned short* __restrict__ out, int H, int W, int tiles_x, int tiles_y) {
  static_assert(CIN % 32 == 0 && (!DS || CIN == 64), "downsample variant keeps both x chunks in LDS");
  constexpr int NCH = CIN / 32, CQ = CIN / 8;
  constexpr int STAGE_BYTES = 2 * kBnXA + kBnH1 + kBnC2;              // 75 072
  __shared__ __attribute__((aligned(16))) char lds[STAGE_BYTES];
  static_assert(32 * kBnOLD * 4 <= 2 * kBnXA + kBnH1, "epilogue transpose aliases the x chunks + c1 halo");
  char* const sH1 = lds + 2 * kBnXA;
  char* const sC2 = sH1 + kBnH1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vi = lane & 31, kb = lane >> 5;
  int bid = blockIdx.x;
  const int tx_i = bid % tiles_x; bid /= tiles_x;
  const int ty_i = bid % tiles_y;
  const int img = bid / tiles_y;
  const int y0 = ty_i * kBnTH, x0 = tx_i * kBnTW;

  // ================= phase A: c1 = relu(x . W1^T + b1) on the 180 halo pixels ==========================
  // x staging roles: 180 pixels x 4 pieces of 16 B per chunk = 720 items over 256 threads
  long hofs0, hofs1, hofs2;
  int hdst0, hdst1, hdst2;
  bool hin0, hin1, hin2, hlive0, hlive1, hlive2;
#define OCC_BN_ROLE(K, OFS, DST, IN, LIVE)                                                        \
  {                                                                                               \
    const int idx = tid + 256 * (K);                                                              \
    LIVE = idx < kBnNP * 4;                                                                       \
    const int p = LIVE ? idx >> 2 : 0, piece = idx & 3;                                           \
    const int hy = p / kBnHW, hx = p % kBnHW;                                                     \
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;                                                 \
    IN = LIVE && iy >= 0 && iy < H && ix >= 0 && ix < W;                                          \
    const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);                           \
    OFS = (((long)img * H + cy) * W + cx) * CQ + piece;                                           \
    DST = p * kBnXS + piece * 16;                                                                 \
  }
  OCC_BN_ROLE(0, hofs0, hdst0, hin0, hlive0)
  OCC_BN_ROLE(1, hofs1, hdst1, hin1, hlive1)
  OCC_BN_ROLE(2, hofs2, hdst2, hin2, hlive2)
#undef OCC_BN_ROLE
  const unsigned hm0 = hin0 ? 0xffffffffu : 0u, hm1 = hin1 ? 0xffffffffu : 0u, hm2 = hin2 ? 0xffffffffu : 0u;
  // four register sets: the x prefetch runs four chunks ahead (two blocks per CU and ~2 us of HBM latency
  // under load: with two chunks in flight the phase was latency-bound)
  uint4 vh0_0, vh1_0, vh2_0, vh0_1, vh1_1, vh2_1, vh0_2, vh1_2, vh2_2, vh0_3, vh1_3, vh2_3;
#define OCC_BN_ISSUE_X(S, CH)                                                                     \
  {                                                                                               \
    const long cq = (long)(CH) * 4;                                                               \
    vh0_##S = x[hofs0 + cq]; vh1_##S = x[hofs1 + cq]; vh2_##S = x[hofs2 + cq];                    \
  }
  // chunk order rotated per block (L2 channel hot-spotting, see conv1x1_nhwc_bf16.hip); the downsample
  // variant keeps chunk c in buffer c for phase C
  const int rot = DS ? 0 : (int)((blockIdx.x * 5u) % (unsigned)NCH);
#define OCC_BN_CH(CI) (((CI) + rot) % NCH)

  const int ntA = wave & 1, mtA0 = 3 * (wave >> 1);
  const int ntB = wave & 1, mtB0 = 2 * (wave >> 1);
  // The MFMAs of phases A and B are issued with the WEIGHTS as the row operand: D[channel][pixel], so a lane
  // ends up with ONE pixel (column = lane & 31) and four groups of 4 consecutive channels
  // (8g + 4*(lane>>5) + 0..3): the bf16 intermediates go to LDS as 8-byte stores of v_cvt_pk pairs and the
  // per-pixel bookkeeping (in-image test) is done once per row tile, not once per element.
  // Biases of this lane's channels, requested first (an in-order vmcnt wait on them later would otherwise
  // also wait for every weight prefetch issued in between)
  float4 bA[4], bB[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    bA[g] = *reinterpret_cast<const float4*>(b1 + ntA * 32 + 8 * g + 4 * kb);
    bB[g] = *reinterpret_cast<const float4*>(b2 + ntB * 32 + 8 * g + 4 * kb);
  }
  f32x16 acc1[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[j][r] = 0.f;
  // W1 fragments travel with the x chunk of the same index (four sets in flight): vmcnt retires in order,
  // a weight load younger than the x prefetches would drain them all
  uint4 wa[4][2];
#define OCC_BN_ISSUE_W1(SLOT, CH)                                                                 \
  {                                                                                               \
    wa[SLOT][0] = w1p[(((long)(CH) * 2 + 0) * 2 + ntA) * 64 + lane];                              \
    wa[SLOT][1] = w1p[(((long)(CH) * 2 + 1) * 2 + ntA) * 64 + lane];                              \
  }
#define OCC_BN_ISSUE_XW(S, CH) { OCC_BN_ISSUE_W1(S, CH) OCC_BN_ISSUE_X(S, CH) }
  OCC_BN_ISSUE_XW(0, OCC_BN_CH(0))
  OCC_BN_ISSUE_XW(1, OCC_BN_CH(NCH > 1 ? 1 : 0))
  if (NCH > 2) {
    OCC_BN_ISSUE_XW(2, OCC_BN_CH(NCH > 2 ? 2 : 0))
    OCC_BN_ISSUE_XW(3, OCC_BN_CH(NCH > 3 ? 3 : 0))
  }
#define OCC_BN_STEP_A(S, CI)                                                                      \
  {                                                                                               \
    char* sX = lds + ((CI) & 1) * kBnXA;                                                          \
    if (hlive0) *reinterpret_cast<uint4*>(sX + hdst0) = make_uint4(vh0_##S.x & hm0, vh0_##S.y & hm0, vh0_##S.z & hm0, vh0_##S.w & hm0); \
    if (hlive1) *reinterpret_cast<uint4*>(sX + hdst1) = make_uint4(vh1_##S.x & hm1, vh1_##S.y & hm1, vh1_##S.z & hm1, vh1_##S.w & hm1); \
    if (hlive2) *reinterpret_cast<uint4*>(sX + hdst2) = make_uint4(vh2_##S.x & hm2, vh2_##S.y & hm2, vh2_##S.z & hm2, vh2_##S.w & hm2); \
    const bf16x8 wf0 = __builtin_bit_cast(bf16x8, wa[S][0]), wf1 = __builtin_bit_cast(bf16x8, wa[S][1]); \
    __syncthreads();                                                                              \
    if ((CI) + 4 < NCH) OCC_BN_ISSUE_XW(S, OCC_BN_CH((CI) + 4 < NCH ? (CI) + 4 : NCH - 1))        \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                               \
      const bf16x8 af0 = *reinterpret_cast<const bf16x8*>(sX + ((mtA0 + j) * 32 + vi) * kBnXS + kb * 16);      \
      const bf16x8 af1 = *reinterpret_cast<const bf16x8*>(sX + ((mtA0 + j) * 32 + vi) * kBnXS + 32 + kb * 16); \
      acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf0, af0, acc1[j], 0, 0, 0);              \
      acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf1, af1, acc1[j], 0, 0, 0);              \
    }                                                                                             \
  }
#pragma unroll
  for (int ci = 0; ci < NCH; ci += 4) {
    OCC_BN_STEP_A(0, ci)
    if (ci + 1 < NCH) OCC_BN_STEP_A(1, ci + 1)
    if (ci + 2 < NCH) OCC_BN_STEP_A(2, ci + 2)
    if (ci + 3 < NCH) OCC_BN_STEP_A(3, ci + 3)
  }
#undef OCC_BN_STEP_A
#undef OCC_BN_ISSUE_XW
#undef OCC_BN_ISSUE_W1
#undef OCC_BN_ISSUE_X
#undef OCC_BN_CH

  // first W2 fragments in flight while c1 is written
  constexpr int PFB = 12;
  uint4 wr[PFB];
#pragma unroll
  for (int s = 0; s < PFB; ++s) wr[s] = w2p[((long)s * 2 + ntB) * 64 + lane];

  // c1 halo: bias + ReLU, zero outside the image (conv2's zero padding applies to c1, not to x); the halo
  // slot of pixel p = hy*18 + hx is simply p * 144 bytes
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int p = (mtA0 + j) * 32 + vi;
    const int hy = p / kBnHW, hx = p - hy * kBnHW;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    const float m = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? 1.f : 0.f;
    if (p < kBnNP) {
      char* dst = sH1 + p * kBnMS + (ntA * 32 + 4 * kb) * 2;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float v0 = fmaxf(acc1[j][4 * g + 0] + bA[g].x, 0.f) * m, v1 = fmaxf(acc1[j][4 * g + 1] + bA[g].y, 0.f) * m;
        const float v2 = fmaxf(acc1[j][4 * g + 2] + bA[g].z, 0.f) * m, v3 = fmaxf(acc1[j][4 * g + 3] + bA[g].w, 0.f) * m;
        *reinterpret_cast<uint2*>(dst + 16 * g) = make_uint2(pack_bf16x2_rne(v0, v1), pack_bf16x2_rne(v2, v3));
      }
    }
  }
  __syncthreads();

  // ================= phase B: c2 = relu(conv3x3(c1) + b2), K = 9 taps x 64 ================================
  f32x16 acc2[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
  const int ab0 = (2 * mtB0 + (vi >> 4)) * kBnH1ROW + (vi & 15) * kBnMS + kb * 16;
  const int ab1 = ab0 + 2 * kBnH1ROW;
  {
    bf16x8 a0 = *reinterpret_cast<const bf16x8*>(sH1 + ab0), a1 = *reinterpret_cast<const bf16x8*>(sH1 + ab1);
#pragma unroll
    for (int s = 0; s < 36; ++s) {
      const bf16x8 wf = __builtin_bit_cast(bf16x8, wr[s % PFB]);
      // the ring keeps PFB weight loads in flight; the scheduling barrier keeps hipcc from sinking them
      // behind the MFMAs (it would otherwise run the phase with two loads in flight, L2-latency-bound)
      if (s + PFB < 36) wr[s % PFB] = w2p[((long)(s + PFB) * 2 + ntB) * 64 + lane];
      bf16x8 a0n = a0, a1n = a1;
      if (s + 1 < 36) {   // next k-step's pixel fragments while this step's MFMAs run
        const int tap = (s + 1) >> 2, ks = (s + 1) & 3;
        const int toff = (tap / 3) * kBnH1ROW + (tap % 3) * kBnMS + ks * 32;
        a0n = *reinterpret_cast<const bf16x8*>(sH1 + ab0 + toff);
        a1n = *reinterpret_cast<const bf16x8*>(sH1 + ab1 + toff);
      }
      acc2[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, a0, acc2[0], 0, 0, 0);
      acc2[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, a1, acc2[1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      a0 = a0n; a1 = a1n;
    }
  }
  // W3 fragments (ring of 4 k-steps) in flight while c2 is written
  constexpr int KS3 = DS ? 8 : 4;
  uint4 w3r[4][2];
#define OCC_BN_ISSUE_W3(SLOT, KS)                                                                 \
  {                                                                                               \
    w3r[SLOT][0] = w3p[((long)(KS) * 8 + 2 * wave + 0) * 64 + lane];                              \
    w3r[SLOT][1] = w3p[((long)(KS) * 8 + 2 * wave + 1) * 64 + lane];                              \
  }
  OCC_BN_ISSUE_W3(0, 0)
  OCC_BN_ISSUE_W3(1, 1)
  OCC_BN_ISSUE_W3(2, 2)
  OCC_BN_ISSUE_W3(3, 3)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    char* dst = sC2 + ((mtB0 + j) * 32 + vi) * kBnMS + (ntB * 32 + 4 * kb) * 2;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float v0 = fmaxf(acc2[j][4 * g + 0] + bB[g].x, 0.f), v1 = fmaxf(acc2[j][4 * g + 1] + bB[g].y, 0.f);
      const float v2 = fmaxf(acc2[j][4 * g + 2] + bB[g].z, 0.f), v3 = fmaxf(acc2[j][4 * g + 3] + bB[g].w, 0.f);
      *reinterpret_cast<uint2*>(dst + 16 * g) = make_uint2(pack_bf16x2_rne(v0, v1), pack_bf16x2_rne(v2, v3));
    }
  }
  __syncthreads();

  // ================= phase C: out = relu(c2 . W3^T (+ x . Wds^T) + b3 (+ x)) ==============================
  // (pixels as the row operand again: the epilogue wants D[pixel][channel])
  f32x16 acc3[4][2];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc3[rt][t][r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < KS3; ++ks) {
    const bf16x8 wf0 = __builtin_bit_cast(bf16x8, w3r[ks & 3][0]);
    const bf16x8 wf1 = __builtin_bit_cast(bf16x8, w3r[ks & 3][1]);
    if (ks + 4 < KS3) OCC_BN_ISSUE_W3(ks & 3, ks + 4)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      bf16x8 af;
      if (ks < 4) {
        af = *reinterpret_cast<const bf16x8*>(sC2 + (rt * 32 + vi) * kBnMS + ks * 32 + kb * 16);
      } else {   // downsample: the centre pixels of the x halo, chunk (ks-4)/2 still sits in its buffer
        const int p = (2 * rt + (vi >> 4) + 1) * kBnHW + (vi & 15) + 1;
        af = *reinterpret_cast<const bf16x8*>(lds + ((ks - 4) >> 1) * kBnXA + p * kBnXS + ((ks - 4) & 1) * 32 + kb * 16);
      }
      acc3[rt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, wf0, acc3[rt][0], 0, 0, 0);
      acc3[rt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, wf1, acc3[rt][1], 0, 0, 0);
    }
  }
#undef OCC_BN_ISSUE_W3

  // ---- epilogue, one 32-pixel row tile (two image rows) at a time through an LDS transpose -------------
  const int c = lane * 4;
  const float4 bv = *reinterpret_cast<const float4*>(b3 + c);
  float* sO = reinterpret_cast<float*>(lds);
  const unsigned short* xs = reinterpret_cast<const unsigned short*>(x);
  // identity = x (Cin == 256): the 8 rows a wave adds in pass rt are requested one pass ahead
  uint2 rv[8], rn[8];
#define OCC_BN_ISSUE_RES(DST, RT)                                                                 \
  _Pragma("unroll") for (int rr = 0; rr < 8; ++rr) {                                              \
    DST[rr] = make_uint2(0u, 0u);                                                                 \
    if (!DS) {                                                                                    \
      const int row = wave * 8 + rr;                                                              \
      const int cy = min(y0 + 2 * (RT) + (row >> 4), H - 1), cx = min(x0 + (row & 15), W - 1);    \
      DST[rr] = *reinterpret_cast<const uint2*>(xs + (((long)img * H + cy) * W + cx) * CIN + c);  \
    }                                                                                             \
  }
  OCC_BN_ISSUE_RES(rn, 0)
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        sO[((r & 3) + 8 * (r >> 2) + 4 * kb) * kBnOLD + (wave * 2 + t) * 32 + vi] = acc3[rt][t][r];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) rv[rr] = rn[rr];
    if (rt + 1 < 4) OCC_BN_ISSUE_RES(rn, rt + 1)
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int row = wave * 8 + rr;
      const int oy = y0 + 2 * rt + (row >> 4), ox = x0 + (row & 15);
      if (oy < H && ox < W) {
        float4 v = *reinterpret_cast<const float4*>(sO + row * kBnOLD + c);
        v.x += bv.x + bn_bf16_to_f32((unsigned short)(rv[rr].x & 0xffffu));
        v.y += bv.y + bn_bf16_to_f32((unsigned short)(rv[rr].x >> 16));
        v.z += bv.z + bn_bf16_to_f32((unsigned short)(rv[rr].y & 0xffffu));
        v.w += bv.w + bn_bf16_to_f32((unsigned short)(rv[rr].y >> 16));
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        const uint2 o = make_uint2(pack_bf16x2_rne(v.x, v.y), pack_bf16x2_rne(v.z, v.w));
        *reinterpret_cast<uint2*>(out + (((long)img * H + oy) * W + ox) * 256 + c) = o;
      }
    }
  }
#undef OCC_BN_ISSUE_RES
}

}  // namespace occ

extern "C" int occ_mfma_pack_b_frag_bf16(const float* weight, void* packed, int N, int K, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(weight && packed, "mfma_pack_b_frag_bf16: null pointer argument");
  OCC_CHECK_ARG(N > 0 && K > 0, "mfma_pack_b_frag_bf16: bad dimension");
  if (N % 32 || K % 16) {
    set_error("mfma_pack_b_frag_bf16: N=%d K=%d (need N %% 32 == 0, K %% 16 == 0)", N, K);
    return OCC_E_UNSUPPORTED;
  }
  const long n = (long)N * K;
  hipLaunchKernelGGL(mfma_pack_b_frag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), weight, reinterpret_cast<unsigned short*>(packed),
                     N, K);
  OCC_CHECK_LAUNCH("mfma_pack_b_frag_bf16");
  return OCC_OK;
}

extern "C" int occ_bottleneck64_nhwc_bf16(const void* x, const void* w1_frag, const float* b1,
                                          const void* w2_frag, const float* b2, const void* w3_frag,
                                          const float* b3, void* out, int batch, int H, int W, int Cin,
                                          int downsample, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(x && w1_frag && b1 && w2_frag && b2 && w3_frag && b3 && out,
                "bottleneck64_nhwc_bf16: null pointer argument");
  OCC_CHECK_ARG(batch > 0 && H > 0 && W > 0, "bottleneck64_nhwc_bf16: bad dimension");
  if (!((Cin == 256 && !downsample) || (Cin == 64 && downsample))) {
    set_error("bottleneck64_nhwc_bf16: no kernel for Cin=%d downsample=%d (256/identity or 64/projection)", Cin,
              downsample);
    return OCC_E_UNSUPPORTED;
  }
  const int tiles_x = (W + kBnTW - 1) / kBnTW, tiles_y = (H + kBnTH - 1) / kBnTH;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const unsigned gx = (unsigned)((long)batch * tiles_x * tiles_y);
#define OCC_BN_LAUNCH(CINN, DSS)                                                                     \
  hipLaunchKernelGGL((bottleneck64_nhwc_bf16_kernel<CINN, DSS>), dim3(gx), dim3(256), 0, st,            \
                     reinterpret_cast<const uint4*>(x), reinterpret_cast<const uint4*>(w1_frag), b1,    \
                     reinterpret_cast<const uint4*>(w2_frag), b2, reinterpret_cast<const uint4*>(w3_frag), \
                     b3, reinterpret_cast<unsigned short*>(out), H, W, tiles_x, tiles_y)
  if (downsample) OCC_BN_LAUNCH(64, true); else OCC_BN_LAUNCH(256, false);
#undef OCC_BN_LAUNCH
  OCC_CHECK_LAUNCH("bottleneck64_nhwc_bf16");
  return OCC_OK;
}
