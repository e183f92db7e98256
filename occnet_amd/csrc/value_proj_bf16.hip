// SCA value projection straight off the backbone's bf16 NHWC feature maps:
//
//   value[cam, start_l + i, :] = (feat_l[cam, i, :] + cams_embeds[cam] + level_embeds[l]) . W^T + b
//                              =  feat_l[cam, i, :] . W^T  +  gbias[l][cam, :]
//
// with gbias = (cams_embeds + level_embeds[l]) . W^T + b precomputed per (level, camera) — the projection is
// linear, so the reference's feature flatten (`transformer_occ.py:204-222`: permute + two embedding adds into a
// (num_cam, sum hw, bs, C) fp32 tensor) and `MSDeformableAttention3D.value_proj`
// (`spatial_cross_attention.py:366`) collapse into this one GEMM per level: the 189 MB fp32 flatten buffer is
// never written nor read (the activation operand is the 94 MB of bf16 maps), and because the activation IS a
// bf16 number its low split part is zero — two MFMAs per k-step (Ah.Wl + Ah.Wh) instead of bf16x3's three, with
// the same hi/lo weight pack as occ_linear_bf16x3_f32 (product error <= 2^-17 of |a.w|).
//
// Structure = conv1x1_nhwc_bf16.hip: block = 4 waves x 64 rows x 128*NT columns, 32-k activation chunks staged
// once per block in LDS (double buffered, one barrier per chunk), fragment-ordered weights global -> registers
// one chunk ahead, K order rotated per block; epilogue through an LDS transpose: + group bias, fp32 rows written
// at out[(g * out_group_rows + out_row0 + i) * ldo] for row m = g * rows_per_group + i (g = camera image).
// All FPN levels go in ONE launch (segment table by value): the small levels alone are launch/ramp-bound
// (13 us for 2 250 rows) and now overlap the large one.
// Ablation on the base shape (184 950 x 256 -> 256, 64-row blocks, 104 us): without the weight loads 77 us,
// without the MFMAs 61 us, without the stores or without the activation loads 84 us each — the weight re-read
// per block through L1/TA (256 KB per block against 32 KB of activations) and the poorly overlapped MFMA burst
// are what separates it from the 39 us of a plain bf16 -> fp32 widening copy of the same tensors; 128-row blocks
// halve the former (102 us).  Pinning the prefetch with sched_barrier and an 8-chunk activation prefetch were
// measured: no gain.
#include <cstdlib>
#include <hip/hip_fp16.h>
#include "common.h"

namespace occ {


// f32 pair -> packed fp16 pair, SATURATING: |x| > 65 504 becomes +-65 504 instead of +-Inf (one v_med3_f32 per value).
// The projected SCA value maps are stored in fp16 (sca_fused.hip); an Inf there would turn every bilinear sample that
// touches the pixel into Inf / NaN (w * Inf, Inf - Inf), a clamp only caps one outlier value.  Random-init and the
// synthetic features never get near the limit; a trained checkpoint's FPN outputs might.
__device__ __forceinline__ unsigned vp_sat_half2(float a, float b) {
  const __half2 h = __floats2half2_rn(__builtin_amdgcn_fmed3f(a, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f));
  return __builtin_bit_cast(unsigned, h);
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// one launch covers every FPN level: segment l = the (rows_l, K) pixel matrix of level l, whose groups (camera
// images, rows_per_group_l = h_l*w_l rows each) land at out_row0_l inside the per-camera blocks of the output
constexpr int kVpMaxSeg = 8;
struct VpSegments {
  const uint4* a[kVpMaxSeg];
  const float* gbias[kVpMaxSeg];
  long rows[kVpMaxSeg], rows_per_group[kVpMaxSeg], out_row0[kVpMaxSeg], lda8[kVpMaxSeg];
  int first_block[kVpMaxSeg + 1];
  int n;
};

// OUTH: the projected value is written as fp16 (opt-in fp16-value mode of the SCA gather) instead of fp32
template <int NT, int RT, bool OUTH>
__global__ __launch_bounds__(256, 2) void value_proj_bf16_kernel(
    VpSegments seg, const uint4* __restrict__ wp, int bias_groups, void* __restrict__ out_, long ldo, int N,
    int K, long out_group_rows, int ncb, int plane_cols, long plane_stride, const float* __restrict__ out_scale) {
  float* __restrict__ out = reinterpret_cast<float*>(out_);
  // (row block rb, column block n0).  One projection: the grid's x / y.  Stacked projections (ncb > 1, several layers'
  // weights along N): the ncb column blocks of a row block must read its 32-64 KB of feature rows from the SAME L2, so
  // they are dealt as linear id = (rb / 8) * 8 ncb + cb * 8 + rb % 8 — hardware block id % 8 picks the XCD, which is
  // rb % 8 for all of them, and they are dispatched within 8 ncb blocks of each other: HBM sees the rows once.
  int rb = (int)blockIdx.x, n0 = (int)blockIdx.y * (128 * NT);
  if (ncb > 1) {
    const int id = (int)blockIdx.x, grp = id / (8 * ncb), within = id % (8 * ncb);
    rb = grp * 8 + (within & 7);
    n0 = (within >> 3) * (128 * NT);
    if (rb >= seg.first_block[kVpMaxSeg]) return;       // padding of the last group of 8 row blocks
  }
  int si = 0;
#pragma unroll
  for (int i = 1; i < kVpMaxSeg; ++i)
    if (i < seg.n && rb >= seg.first_block[i]) si = i;
  const uint4* __restrict__ a = seg.a[si];
  const float* __restrict__ gbias = seg.gbias[si];
  const long M = seg.rows[si], rows_per_group = seg.rows_per_group[si], out_row0 = seg.out_row0[si];
  const long lda8 = seg.lda8[si];
  constexpr int KC = 32, kLD = KC * 2 + 16, PC = KC / 8;
  constexpr int BM = 32 * RT, BN = 128 * NT, WR = 32 * NT, OLD = BN + 4;
  constexpr int A_BYTES = BM * kLD;
  constexpr int STAGE_BYTES = 2 * A_BYTES, OUT_BYTES = 32 * OLD * 4;
  __shared__ __attribute__((aligned(16))) char lds[STAGE_BYTES > OUT_BYTES ? STAGE_BYTES : OUT_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vi = lane & 31, kb = lane >> 5;
  const long m0 = (long)(rb - seg.first_block[si]) * BM;     // first_block counts BM-row blocks
  const int NT32 = (N + 31) / 32;

  f32x16 acc[RT][NT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rt][t][r] = 0.f;

  // A: thread -> (row = tid / 4, 16-byte piece = tid % 4) of the 64 x 32 chunk; unconditional clamped loads
  constexpr int AP = BM * PC / 256;                  // activation pieces per thread per chunk (1 or 2)
  static_assert((AP == 1 || AP == 2) && NT <= 2, "staging register budget");
  const int arow = tid / PC, apiece = tid % PC;      // second piece (AP == 2): row + 64
  long am = m0 + arow, am2 = m0 + arow + 64;
  if (am >= M) am = M - 1;
  if (am2 >= M) am2 = M - 1;
  const long aofs = am * lda8 + apiece, aofs2 = am2 * lda8 + apiece;
  const int adst = arow * kLD + apiece * 16;
  // this wave's column tiles in the packed weight (hi plane; the lo plane is +64 uint4)
  const int nt0 = min((n0 + wave * WR) / 32, NT32 - 1), nt1 = min((n0 + wave * WR) / 32 + (NT - 1), NT32 - 1);
  const long wl0 = (long)nt0 * 128 + lane, wl1 = (long)nt1 * 128 + lane;
  // two activation sets (chunks c+1, c+2 in flight), two weight sets (chunk c in use, c+1 in flight).  (Requesting
  // the block's whole 64 x K activation tile up front — 8 chunk registers — was measured: not faster.)
  uint4 va_0, va_1, vb_0, vb_1;      // vb: the second 64 rows (RT == 4)
  uint4 ha0_0, la0_0, hb0_0, lb0_0, ha1_0, la1_0, hb1_0, lb1_0;     // {h,l}{k-step a,b}{tile}_{set}
  uint4 ha0_1, la0_1, hb0_1, lb0_1, ha1_1, la1_1, hb1_1, lb1_1;
#define OCC_VP_ISSUE_A(S, K0) { va_##S = a[aofs + (K0) / 8]; if (AP > 1) vb_##S = a[aofs2 + (K0) / 8]; }
#define OCC_VP_ISSUE_W(S, K0)                                                                     \
  {                                                                                               \
    const long k0 = (long)((K0) / 16) * NT32 * 128, k1 = k0 + (long)NT32 * 128;                   \
    ha0_##S = wp[k0 + wl0]; la0_##S = wp[k0 + wl0 + 64];                                          \
    hb0_##S = wp[k1 + wl0]; lb0_##S = wp[k1 + wl0 + 64];                                          \
    if (NT > 1) {                                                                                 \
      ha1_##S = wp[k0 + wl1]; la1_##S = wp[k0 + wl1 + 64];                                        \
      hb1_##S = wp[k1 + wl1]; lb1_##S = wp[k1 + wl1 + 64];                                        \
    }                                                                                             \
  }
#define OCC_VP_MFMA(RTI, T, AF, WREG) \
  acc[RTI][T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF, __builtin_bit_cast(bf16x8, WREG), acc[RTI][T], 0, 0, 0);
  // chunk c: activation set SA -> LDS buffer BUF, barrier, request A(c+2) into set SA and W(c+1) into the other
  // weight set, MFMAs with weight set SW (small term first)
#define OCC_VP_STEP(SA, BUF, SW, SWN, K_A_NEXT, K_W_NEXT)                                         \
  {                                                                                               \
    char* sA = lds + (BUF) * A_BYTES;                                                             \
    *reinterpret_cast<uint4*>(sA + adst) = va_##SA;                                               \
    if (AP > 1) *reinterpret_cast<uint4*>(sA + adst + 64 * kLD) = vb_##SA;                        \
    __syncthreads();                                                                              \
    OCC_VP_ISSUE_A(SA, K_A_NEXT)                                                                  \
    OCC_VP_ISSUE_W(SWN, K_W_NEXT)                                                                 \
    bf16x8 af[RT][2];                                                                             \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                             \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                            \
        af[rt][ks] = *reinterpret_cast<const bf16x8*>(sA + (rt * 32 + vi) * kLD + ks * 32 + kb * 16); \
    /* term-major order: two MFMAs on ONE accumulator back to back wait out the 64-cycle result latency (issue  \
       is 32 cycles); walking all tiles per term puts RT*NT instructions between them */                         \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) {                                           \
      OCC_VP_MFMA(rt, 0, af[rt][0], la0_##SW) if (NT > 1) { OCC_VP_MFMA(rt, NT - 1, af[rt][0], la1_##SW) } }   \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) {                                           \
      OCC_VP_MFMA(rt, 0, af[rt][0], ha0_##SW) if (NT > 1) { OCC_VP_MFMA(rt, NT - 1, af[rt][0], ha1_##SW) } }   \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) {                                           \
      OCC_VP_MFMA(rt, 0, af[rt][1], lb0_##SW) if (NT > 1) { OCC_VP_MFMA(rt, NT - 1, af[rt][1], lb1_##SW) } }   \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) {                                           \
      OCC_VP_MFMA(rt, 0, af[rt][1], hb0_##SW) if (NT > 1) { OCC_VP_MFMA(rt, NT - 1, af[rt][1], hb1_##SW) } }   \
  }
  const int NCHK = K / KC;
  const int rot = (int)((blockIdx.x * 5u + blockIdx.y * 3u) % (unsigned)NCHK);
#define OCC_VP_K(CI) ((((CI) < NCHK ? (CI) : NCHK - 1) + rot) % NCHK * KC)
  OCC_VP_ISSUE_A(0, OCC_VP_K(0))
  OCC_VP_ISSUE_W(0, OCC_VP_K(0))
  OCC_VP_ISSUE_A(1, OCC_VP_K(1))
  for (int ci = 0; ci < NCHK; ci += 2) {
    OCC_VP_STEP(0, 0, 0, 1, OCC_VP_K(ci + 2), OCC_VP_K(ci + 1))
    if (ci + 1 < NCHK) OCC_VP_STEP(1, 1, 1, 0, OCC_VP_K(ci + 3), OCC_VP_K(ci + 2))
  }
#undef OCC_VP_K
#undef OCC_VP_STEP
#undef OCC_VP_MFMA
#undef OCC_VP_ISSUE_W
#undef OCC_VP_ISSUE_A

  // ---- epilogue, 32 rows at a time through an LDS transpose: + group bias, fp32 rows ---------------------
  const int c = lane * 4;
  const bool col_live = c < BN && n0 + c < N;
  float* sO = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        sO[((r & 3) + 8 * (r >> 2) + 4 * kb) * OLD + (wave * NT + t) * 32 + vi] = acc[rt][t][r];
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int row = wave * 8 + rr;
      const long m = m0 + rt * 32 + row;
      if (m < M && col_live) {
        const long g = m / rows_per_group, i = m - g * rows_per_group;
        float4 v = *reinterpret_cast<const float4*>(sO + row * OLD + c);
        if (gbias != nullptr) {
          const float4 b = *reinterpret_cast<const float4*>(gbias + (g % bias_groups) * N + n0 + c);
          v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        // column n lives in output plane n / plane_cols (one plane per stacked projection), column n % plane_cols
        const int nn = n0 + c;
        if (out_scale != nullptr) {      // the plane's power-of-two range scale (value_range.hip): exact
          const float s = out_scale[nn / plane_cols];
          v.x *= s; v.y *= s; v.z *= s; v.w *= s;
        }
        const long eo = (long)(nn / plane_cols) * plane_stride + (g * out_group_rows + out_row0 + i) * ldo + nn % plane_cols;
        if (OUTH) {
          // fp16 maps are the SCA gather's operand: pixel-PAIR layout [group][pix >> 1][head][pix & 1][32] (sca_fused.hip),
          // pix = out_row0 + i, head = column / 32; rows are contiguous (ldo == plane_cols, checked by the launcher)
          const long pix = out_row0 + i;
          const int pc = nn % plane_cols;
          const long ep = (long)(nn / plane_cols) * plane_stride + (g * out_group_rows + (pix & ~1L)) * ldo +
                          (pc >> 5) * 64 + (pix & 1) * 32 + (pc & 31);
          *reinterpret_cast<uint2*>(reinterpret_cast<__half*>(out_) + ep) =
              make_uint2(vp_sat_half2(v.x, v.y), vp_sat_half2(v.z, v.w));
        } else {
          *reinterpret_cast<float4*>(out + eo) = v;
        }
      }
    }
  }
}


// ---- activation-resident variant (K = 256, N % 256 == 0: the stacked SCA value projections) --------------------------
// The tiled kernel above meets a block barrier every 32 k with two waves per SIMD: its weight loads, activation loads,
// MFMAs and stores each cost 20-40 % of the launch when removed alone (ablation in the header) — they add up instead of
// overlapping.  Here a block owns 128 rows for ALL N columns:
//   * its 128 x 256 bf16 activation tile (64 KB) goes global -> LDS once, by LDS-DMA (no registers), 16-byte pieces
//     XOR-swizzled by row so the MFMA A-fragment reads (ds_read_b128, row pitch 512 B) are bank-conflict-free;
//   * after that ONE barrier the four waves never synchronise again: wave w walks the column passes (256 columns per
//     pass, the wave's 64 of them as two 32-column tiles x four 32-row tiles = 8 accumulators), streaming its hi/lo
//     weight fragments from L2 through a 4-deep register ring that runs ahead across k-steps AND across passes, so the
//     epilogue of pass p runs under the weight loads of pass p + 1 and under the other waves' MFMAs;
//   * the MFMAs are TRANSPOSED (weights as the row operand): a lane ends up with 16 columns of ONE output row, four
//     consecutive ones per register quad, so the epilogue is a packed convert + one ds_write_b64 per quad into a
//     2.5 KB per-wave scratch and 16-byte row-segment stores out of it; the accumulators START at the group bias
//     instead of zero, so there is no bias pass (first cut, lane = column: a convert, a select, an add and a 2-byte
//     LDS write per element — 1 080 VALU instructions per pass beside 256 MFMAs, profiles/r03_vproj_resident_pmc.txt);
//   * the feature rows are read from HBM exactly once per launch, whatever the number of stacked projections.
// Base shape (184 950 rows, four stacked 256-column projections, fp16 out): 0.40 ms tiled -> 0.22 ms (870 TFLOP/s of
// bf16 MFMA work, profiles/r03_vproj_probe.txt).  Starting every other 256 blocks half a pass late (so that the two
// blocks of a CU do not sit in the MFMA phase / the epilogue together) was measured: no change (r03_vproj_stagger.txt).
constexpr int kVprRows = 128, kVprK = 256, kVprPitch = 80, kVprScratch = 32 * kVprPitch;
constexpr int kVprTileBytes = kVprRows * kVprK * 2;
constexpr int kVprLdsBytes = kVprTileBytes + 4 * kVprScratch;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// OFMT: 0 = fp32 rows, 1 = fp16 pixel pairs, 2 = q16 pixel pairs (block floating point, common.h: 16 bits per element, one
// 4-bit exponent per 16-byte piece of 8 channels)
template <int OFMT>
__global__ __launch_bounds__(256, 2) void value_proj_resident_kernel(
    VpSegments seg, const uint4* __restrict__ wp, int bias_groups, void* __restrict__ out_, long ldo, int N,
    long out_group_rows, int plane_cols, long plane_stride, const float* __restrict__ out_scale) {
  extern __shared__ __attribute__((aligned(16))) char vlds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vi = lane & 31, kb = lane >> 5;
  const int rb = (int)blockIdx.x;
  int si = 0;
#pragma unroll
  for (int i = 1; i < kVpMaxSeg; ++i)
    if (i < seg.n && rb >= seg.first_block[i]) si = i;
  const uint4* __restrict__ a = seg.a[si];
  const float* __restrict__ gbias = seg.gbias[si];
  // row counts fit an int (the launcher checks): 32-bit index arithmetic keeps the scalar unit's divisions short
  const int M = (int)seg.rows[si], rpg = (int)seg.rows_per_group[si];
  const long out_row0 = seg.out_row0[si], lda8 = seg.lda8[si];
  const int m0 = (rb - seg.first_block[si]) * kVprRows;
  const int NT32 = N / 32, NP = N / 256;

  // activation tile: DMA instruction j of wave w fills tile rows 2 (16 w + j) and + 1 (lanes 0-31 / 32-63); LDS slot s
  // of row r holds the row's 16-byte piece s ^ (r & 31)
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int r = (wave * 16 + j) * 2 + kb;
    int m = m0 + r;
    if (m >= M) m = M - 1;
    __builtin_amdgcn_global_load_lds(a + (long)m * lda8 + (vi ^ (r & 31)), (lds_ptr_t)(vlds + (wave * 16 + j) * 1024), 16, 0, 0);
  }

  // weight ring: slot (step & 3) = {hi tile 0, lo tile 0, hi tile 1, lo tile 1} of flat step = pass * 16 + k-step
  // (buffer loads: one lane-offset VGPR for all of them, the step's offset in an SGPR, hi/lo/tile in the immediate)
  occ_u32x4 w[4][4];
  const __amdgpu_buffer_rsrc_t wr = uniform_rsrc(wp, (unsigned)(kVprK / 16) * (unsigned)NT32 * 2048u);
  const int wv = (wave * 2 * 128 + lane) * 16;
  const int kstep_bytes = NT32 * 2048;
#define OCC_VPR_LOAD(SLOT, PASS, KS)                                                               \
  {                                                                                               \
    const int so = (KS) * kstep_bytes + (PASS) * (8 * 2048);                                      \
    w[SLOT][0] = __builtin_amdgcn_raw_buffer_load_b128(wr, wv, so, 0);                            \
    w[SLOT][1] = __builtin_amdgcn_raw_buffer_load_b128(wr, wv + 1024, so, 0);                     \
    w[SLOT][2] = __builtin_amdgcn_raw_buffer_load_b128(wr, wv + 2048, so, 0);                     \
    w[SLOT][3] = __builtin_amdgcn_raw_buffer_load_b128(wr, wv + 3072, so, 0);                     \
  }
  OCC_VPR_LOAD(0, 0, 0)
  OCC_VPR_LOAD(1, 0, 1)
  OCC_VPR_LOAD(2, 0, 2)
  __syncthreads();                                  // the tile has landed (the barrier waits for the DMA)

  // rows of this block may belong to two groups (camera images): g0 up to `boundary`, g0 + 1 from there
  const int g0 = m0 / rpg;
  const int boundary = min((g0 + 1) * rpg - m0, kVprRows);
  const float* __restrict__ bias0 = gbias ? gbias + (long)(g0 % bias_groups) * N : nullptr;
  const float* __restrict__ bias1 = gbias ? gbias + (long)((g0 + 1) % bias_groups) * N : nullptr;
  // output row of tile row r: orow0 + r (group g0) or orow1 + r (group g0 + 1)
  const long orow0 = (long)g0 * out_group_rows + out_row0 + (m0 - g0 * rpg);
  const long orow1 = orow0 + out_group_rows - rpg;
  // the same split into (first row of the group's block, pixel index inside it) for the fp16 pixel-pair layout
  // (a group's block is < 2 GB: 32-bit offsets inside it)
  const unsigned row_bytes = (unsigned)ldo * 2u;
  const long gbytes0 = (long)g0 * out_group_rows * row_bytes, gbytes1 = gbytes0 + out_group_rows * row_bytes;
  const int pix0 = (int)out_row0 + (m0 - g0 * rpg), pix1 = pix0 - rpg;
  char* scratch = vlds + kVprTileBytes + wave * kVprScratch;

  // A fragments of k-step ks (the same for every pass): double buffered, read one step ahead
  bf16x8 af[2][4];
  // (slot of piece 2 ks + kb in row vi = (2 ks) ^ ((kb ^ vi) & 31): one XOR per step on an address the compiler cannot
  // see through — it would otherwise keep all 16 steps' addresses in registers across the pass loop)
  unsigned abase = (unsigned)(vi * 512 + ((kb ^ vi) & 31) * 16);
#define OCC_VPR_AFRAG(BUF, KS)                                                                     \
  {                                                                                               \
    asm volatile("" : "+v"(abase));                                                               \
    const char* ap = vlds + (abase ^ (unsigned)((KS) * 32));                                      \
    _Pragma("unroll") for (int rt = 0; rt < 4; ++rt)                                              \
      af[BUF][rt] = *reinterpret_cast<const bf16x8*>(ap + rt * (32 * 512));                       \
  }
  OCC_VPR_AFRAG(0, 0)
#pragma unroll 1
  for (int p = 0; p < NP; ++p) {
    const int nw = p * 256 + wave * 64;              // the wave's first column of this pass
    // D[column][row]: lane (vi, kb) holds row rt * 32 + vi, register 4 q + i = column 8 q + 4 kb + i of tile t.
    // The accumulators start at the row's group bias.
    f32x16 acc[4][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0;
        if (gbias) {
          c0 = *reinterpret_cast<const float4*>(bias0 + nw + t * 32 + 8 * q + 4 * kb);
          c1 = *reinterpret_cast<const float4*>(bias1 + nw + t * 32 + 8 * q + 4 * kb);
        }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          const bool second = rt * 32 + vi >= boundary;
          acc[rt][t][4 * q + 0] = second ? c1.x : c0.x;
          acc[rt][t][4 * q + 1] = second ? c1.y : c0.y;
          acc[rt][t][4 * q + 2] = second ? c1.z : c0.z;
          acc[rt][t][4 * q + 3] = second ? c1.w : c0.w;
        }
      }
    const int pn = p + 1 < NP ? p + 1 : p;           // the ring runs into the next pass (last pass: a harmless re-read)
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (ks + 3 < 16) OCC_VPR_LOAD((ks + 3) & 3, p, ks + 3)
      else OCC_VPR_LOAD((ks + 3) & 3, pn, ks + 3 - 16)
      OCC_VPR_AFRAG((ks + 1) & 1, (ks + 1) & 15)
      // small term first; 8 accumulators between the two MFMAs on one accumulator cover the result latency
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          acc[rt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[ks & 3][2 * t + 1]), af[ks & 1][rt],
                                                               acc[rt][t], 0, 0, 0);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          acc[rt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[ks & 3][2 * t]), af[ks & 1][rt],
                                                               acc[rt][t], 0, 0, 0);
      // pin the software pipeline (hipcc otherwise sinks every ring request down to its use: load, vmcnt(0), MFMA)
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);   // ring requests of step s + 3
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // A fragments of step s + 1
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }

    // ---- epilogue of the pass: CPR columns of a 32-row tile at a time through the wave's scratch (row pitch 80 B) ------
    constexpr bool OUTH = OFMT != 0;
    constexpr int EB = OUTH ? 2 : 4, CPR = OUTH ? 32 : 16;      // bytes / element, columns per round (64-byte row segments)
    int elane = lane;                               // opaque per pass: the store offsets are recomputed here instead of
    asm volatile("" : "+v"(elane));                 // living (spilled) across the k loop
    const int plane = nw / plane_cols, pcol = nw - plane * plane_cols;     // the wave's 64 columns lie in one plane
    // the plane's power-of-two range scale (value_range.hip; exact): the fp16 rows of a plane cannot saturate
    const float osc = out_scale != nullptr ? out_scale[plane] : 1.f;
    // q16: rows are stored under s / 2 (common.h); lanes kb = 0 hold a piece's elements 0-3 (0, 1 tagged with the exponent)
    const int q_eosc = OFMT == 2 ? __builtin_amdgcn_frexp_expf(osc) - 2 : 0;
    const int q_km = kb == 0 ? 3 : 0, q_sh = kb == 0 ? 2 : 0;
    const float q_mulS = kb == 0 ? 0.25f : 1.f;
    char* const obase = reinterpret_cast<char*>(out_) + ((long)plane * plane_stride + pcol) * EB;
    char* const pbase = reinterpret_cast<char*>(out_) + (long)plane * plane_stride * EB;     // plane base (pair layout)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int rd = 0; rd < 32 / CPR; ++rd) {
          if (OFMT == 2) {
            // q16 (common.h): ONE exponent for the whole 64-byte head row (32 channels = this lane's 16 + lane ^ 32's 16) —
            // a coarser group than the format's 16-byte piece, a valid encoding; per-piece exponents cost 44 more VALU
            // instructions per tile here.  A 16-byte piece = this lane's 4 channels (kb = 0: elements 0-3, the first two
            // carry the exponent) and lane + 32's 4 (elements 4-7).
            float m = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r += 2) m = fmaxf(m, fmaxf(fabsf(acc[rt][t][r]), fabsf(acc[rt][t][r + 1])));
            {
              const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
              m = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            const int E = q16_group_exponent(m, q_eosc);
            const float f = ldexpf(1.f, q_eosc + 15 - E);
            const int r0 = E & q_km, r1 = (E >> 2) & q_km;
            const float fS = f * q_mulS, c0 = -0.25f * (float)r0, c1 = -0.25f * (float)r1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              *reinterpret_cast<uint2*>(scratch + vi * kVprPitch + 16 * q + 8 * kb) =
                  make_uint2(q16_pair_tagged(acc[rt][t][4 * q + 0], acc[rt][t][4 * q + 1], fS, c0, c1, r0, r1, q_sh),
                             q16_pair(acc[rt][t][4 * q + 2], acc[rt][t][4 * q + 3], f));
            }
          } else if (OUTH) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              *reinterpret_cast<uint2*>(scratch + vi * kVprPitch + 16 * q + 8 * kb) =
                  make_uint2(vp_sat_half2(acc[rt][t][4 * q + 0] * osc, acc[rt][t][4 * q + 1] * osc),
                             vp_sat_half2(acc[rt][t][4 * q + 2] * osc, acc[rt][t][4 * q + 3] * osc));
            }
          } else {
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
              const int q = 2 * rd + qq;
              *reinterpret_cast<float4*>(scratch + vi * kVprPitch + 32 * qq + 16 * kb) =
                  make_float4(acc[rt][t][4 * q + 0] * osc, acc[rt][t][4 * q + 1] * osc, acc[rt][t][4 * q + 2] * osc,
                              acc[rt][t][4 * q + 3] * osc);
            }
          }
          wave_lds_sync();
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int row = (elane >> 2) + 16 * j, piece = elane & 3;
            const int rloc = rt * 32 + row;
            const uint4 v = *reinterpret_cast<const uint4*>(scratch + row * kVprPitch + piece * 16);
            if (m0 + rloc < M) {
              if (OUTH) {
                // pixel-pair layout of the fp16 maps (sca_fused.hip): [group][pix >> 1][head][pix & 1][32]; the wave's
                // tile t is head (pcol + 32 t) / 32 of the plane, a 16-byte piece = 8 of its 32 channels
                const bool second = rloc >= boundary;
                const int pix = (second ? pix1 : pix0) + rloc;
                const unsigned inb = (unsigned)(pix & ~1) * row_bytes + (unsigned)(((pcol >> 5) + t) * 128 + (pix & 1) * 64 + piece * 16);
                *reinterpret_cast<uint4*>(pbase + (second ? gbytes1 : gbytes0) + inb) = v;
              } else {
                const long orow = (rloc >= boundary ? orow1 : orow0) + rloc;
                *reinterpret_cast<uint4*>(obase + (orow * ldo + t * 32 + rd * CPR) * EB + piece * 16) = v;
              }
            }
          }
          wave_lds_sync();
        }
      }
    }
  }
#undef OCC_VPR_AFRAG
#undef OCC_VPR_LOAD
}

}  // namespace occ

static int value_proj_bf16_launch(int n_segments, const void* const* a, const int64_t* lda,
                                  const int64_t* rows, const int64_t* rows_per_group,
                                  const int64_t* out_row0, const float* const* group_bias,
                                  int bias_groups, const void* weight_packed, void* out, int64_t ldo,
                                  int K, int N, int64_t out_group_rows, int out_fmt /* 0 f32, 1 f16 pairs, 2 q16 pairs */,
                                  void* stream, int plane_cols = 0, int64_t plane_stride = 0,
                                  const float* out_scale = nullptr) {
  const bool out_f16 = out_fmt != 0;        // 16-bit elements in the pixel-pair layout
  using namespace occ;
  OCC_CHECK_ARG(a && lda && rows && rows_per_group && out_row0 && weight_packed && out,
                "value_proj_bf16: null pointer argument");
  OCC_CHECK_ARG(n_segments > 0 && n_segments <= kVpMaxSeg, "value_proj_bf16: 1..%d segments", kVpMaxSeg);
  if (plane_cols <= 0) { plane_cols = N; plane_stride = 0; }
  OCC_CHECK_ARG(K > 0 && N > 0 && out_group_rows >= 0 && ldo >= plane_cols && N % plane_cols == 0,
                "value_proj_bf16: bad dimension");
  OCC_CHECK_ARG(!group_bias || bias_groups > 0, "value_proj_bf16: bias_groups must be positive");
  // fp16 output = the SCA gather's pixel-pair layout: contiguous rows of whole 32-channel heads, even blocks
  OCC_CHECK_ARG(!out_f16 || (ldo == plane_cols && plane_cols % 32 == 0 && out_group_rows % 2 == 0 &&
                             out_group_rows * ldo * 2 < (1L << 31)),
                "value_proj_bf16: fp16 output needs ldo == columns, columns %% 32 == 0 and an even out_group_rows");
  if (K % 32 || N % 4 || ldo % 4) {
    set_error("value_proj_bf16: no kernel for K=%d N=%d (need K %% 32 == 0, N %% 4 == 0, 16-byte aligned rows)",
              K, N);
    return OCC_E_UNSUPPORTED;
  }
  // 128-row blocks (four row tiles per wave: the wave's weight fragments, which stream through L1 per block —
  // 256 KB per block at K = N = 256 against 32-64 KB of activations — are reused twice as often) whenever that
  // still gives every CU two blocks
  long total_rows = 0;
  for (int i = 0; i < n_segments; ++i) total_rows += rows[i];
  // the activation-resident kernel: K = 256, whole 256-column passes, at most two groups per 128-row block, and a
  // plane never split inside a wave's 64 columns.  OCC_VPROJ_RESIDENT=0 (development switch) keeps the tiled kernel.
  static const bool resident_on = [] { const char* e = getenv("OCC_VPROJ_RESIDENT"); return !(e && e[0] == '0'); }();
  bool resident = resident_on && K == kVprK && N % 256 == 0 && plane_cols % 64 == 0 && ldo % 8 == 0 &&
                  (plane_stride == 0 || plane_stride % 8 == 0);
  for (int i = 0; i < n_segments && resident; ++i)
    resident = rows_per_group[i] >= kVprRows && rows[i] < (1L << 30) && rows_per_group[i] < (1L << 30);
  const int bm = resident ? kVprRows : (total_rows / 128) * ((N + 255) / 256) >= 2L * 256 ? 128 : 64;
  VpSegments seg;
  long blocks = 0;
  for (int i = 0; i < kVpMaxSeg; ++i) {
    const int j = i < n_segments ? i : n_segments - 1;       // unused slots alias the last segment
    OCC_CHECK_ARG(a[j] && rows[j] > 0 && rows_per_group[j] > 0 && out_row0[j] >= 0 && lda[j] >= K,
                  "value_proj_bf16: bad segment %d", j);
    if (lda[j] % 8) {
      set_error("value_proj_bf16: segment %d row stride %ld is not 16-byte aligned", j, (long)lda[j]);
      return OCC_E_UNSUPPORTED;
    }
    seg.a[i] = reinterpret_cast<const uint4*>(a[j]);
    seg.gbias[i] = group_bias ? group_bias[j] : nullptr;
    seg.rows[i] = rows[j];
    seg.rows_per_group[i] = rows_per_group[j];
    seg.out_row0[i] = out_row0[j];
    seg.lda8[i] = lda[j] / 8;
    seg.first_block[i] = (int)blocks;
    if (i < n_segments) blocks += (rows[j] + bm - 1) / bm;
  }
  seg.first_block[kVpMaxSeg] = (int)blocks;
  seg.n = n_segments;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (resident) {
    auto launch = [&](auto kern) -> hipError_t {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         kVprLdsBytes);
      if (e == hipSuccess)
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), kVprLdsBytes, st, seg,
                           reinterpret_cast<const uint4*>(weight_packed), bias_groups, out, (long)ldo, N,
                           (long)out_group_rows, plane_cols, (long)plane_stride, out_scale);
      return e;
    };
    const hipError_t e = out_fmt == 2 ? launch(value_proj_resident_kernel<2>)
                         : out_fmt == 1 ? launch(value_proj_resident_kernel<1>) : launch(value_proj_resident_kernel<0>);
    if (e != hipSuccess) {
      set_error("value_proj_bf16: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return OCC_E_LAUNCH;
    }
    OCC_CHECK_LAUNCH("value_proj_bf16");
    return OCC_OK;
  }
  if (out_fmt == 2) {
    set_error("value_proj_bf16: q16 output exists on the activation-resident kernel only (K = 256, N %% 256 == 0, groups of >= "
              "128 rows): project to fp32 and encode with occ_sca_rows_encode_q16");
    return OCC_E_UNSUPPORTED;
  }
  // stacked projections (planes): the column blocks of a row block are dealt to one XCD (see the kernel)
  const bool walk = plane_stride != 0 && plane_cols % 256 == 0;
  const long groups8 = (blocks + 7) / 8;
#define OCC_VP_LAUNCH__(NTT, BNN, RTT, HH)                                                          \
  hipLaunchKernelGGL((value_proj_bf16_kernel<NTT, RTT, HH>),                                        \
                     walk ? dim3((unsigned)(groups8 * 8 * ((N + BNN - 1) / BNN)), 1)                \
                          : dim3((unsigned)blocks, (unsigned)((N + BNN - 1) / BNN)),                \
                     dim3(256), 0, st, seg, reinterpret_cast<const uint4*>(weight_packed), bias_groups, out, \
                     (long)ldo, N, K, (long)out_group_rows, walk ? (N + BNN - 1) / BNN : 1, plane_cols, \
                     (long)plane_stride, out_scale)
#define OCC_VP_LAUNCH_(NTT, BNN, RTT) do { if (out_f16) OCC_VP_LAUNCH__(NTT, BNN, RTT, true); else OCC_VP_LAUNCH__(NTT, BNN, RTT, false); } while (0)
#define OCC_VP_LAUNCH(NTT, BNN) do { if (bm == 128) OCC_VP_LAUNCH_(NTT, BNN, 4); else OCC_VP_LAUNCH_(NTT, BNN, 2); } while (0)
  if (N <= 128) OCC_VP_LAUNCH(1, 128); else OCC_VP_LAUNCH(2, 256);
#undef OCC_VP_LAUNCH
#undef OCC_VP_LAUNCH_
#undef OCC_VP_LAUNCH__
  OCC_CHECK_LAUNCH("value_proj_bf16");
  return OCC_OK;
}

extern "C" int occ_value_proj_bf16_f32(int n_segments, const void* const* a, const int64_t* lda,
                                       const int64_t* rows, const int64_t* rows_per_group,
                                       const int64_t* out_row0, const float* const* group_bias,
                                       int bias_groups, const void* weight_packed, float* out, int64_t ldo,
                                       int K, int N, int64_t out_group_rows, void* stream) {
  return value_proj_bf16_launch(n_segments, a, lda, rows, rows_per_group, out_row0, group_bias, bias_groups,
                                weight_packed, out, ldo, K, N, out_group_rows, 0, stream);
}

// same, output written as fp16 IN THE PIXEL-PAIR ORDER of occ_sca_fused_forward_f16v (ldo in fp16 elements); the symbol was
// occ_value_proj_bf16_f16 while the rows were in pixel order (round 2) — renamed with the layout so that a C caller
// linked against the old contract fails at link time instead of reading permuted rows
extern "C" int occ_value_proj_bf16_f16pairs(int n_segments, const void* const* a, const int64_t* lda,
                                       const int64_t* rows, const int64_t* rows_per_group,
                                       const int64_t* out_row0, const float* const* group_bias,
                                       int bias_groups, const void* weight_packed, void* out, int64_t ldo,
                                       int K, int N, int64_t out_group_rows, const float* out_scale, void* stream) {
  return value_proj_bf16_launch(n_segments, a, lda, rows, rows_per_group, out_row0, group_bias, bias_groups,
                                weight_packed, out, ldo, K, N, out_group_rows, 1, stream, 0, 0, out_scale);
}

// same, output as q16 pixel pairs (block floating point: common.h); OCC_E_UNSUPPORTED where only the tiled kernel applies
// (the caller projects to fp32 and encodes with occ_sca_rows_encode_q16)
extern "C" int occ_value_proj_bf16_q16pairs(int n_segments, const void* const* a, const int64_t* lda,
                                       const int64_t* rows, const int64_t* rows_per_group,
                                       const int64_t* out_row0, const float* const* group_bias,
                                       int bias_groups, const void* weight_packed, void* out, int64_t ldo,
                                       int K, int N, int64_t out_group_rows, const float* out_scale, void* stream) {
  return value_proj_bf16_launch(n_segments, a, lda, rows, rows_per_group, out_row0, group_bias, bias_groups,
                                weight_packed, out, ldo, K, N, out_group_rows, 2, stream, 0, 0, out_scale);
}

// Several projections of the SAME rows in one launch (the four encoder layers' SCA value projections depend on the
// camera features only): weight_packed = pack of the (n_planes * plane_cols, K) stacked weights, group_bias[s]
// (bias_groups, n_planes * plane_cols); projection p writes plane p of `out` (plane_stride elements apart, rows of ldo
// elements, plane_cols columns) exactly as occ_value_proj_bf16_f16pairs / _f32 would.  The column blocks of a row block run
// on one XCD back to back, so the feature maps are read from HBM once instead of once per layer.  plane_cols % 256 == 0.
extern "C" int occ_value_proj_bf16_planes(int n_segments, const void* const* a, const int64_t* lda,
                                          const int64_t* rows, const int64_t* rows_per_group,
                                          const int64_t* out_row0, const float* const* group_bias,
                                          int bias_groups, const void* weight_packed, void* out, int out_f16,
                                          int64_t ldo, int K, int n_planes, int plane_cols, int64_t plane_stride,
                                          int64_t out_group_rows, const float* out_scale, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(n_planes > 0 && plane_cols > 0 && plane_stride > 0, "value_proj_bf16_planes: bad plane geometry");
  if (plane_cols % 256) {
    set_error("value_proj_bf16_planes: plane_cols=%d is not a multiple of 256", plane_cols);
    return OCC_E_UNSUPPORTED;
  }
  return value_proj_bf16_launch(n_segments, a, lda, rows, rows_per_group, out_row0, group_bias, bias_groups,
                                weight_packed, out, ldo, K, n_planes * plane_cols, out_group_rows,
                                out_f16 == 2 ? 2 : out_f16 != 0 ? 1 : 0, stream, plane_cols, plane_stride, out_scale);
}

// fp32 value rows -> q16 pixel pairs (block floating point: common.h fma8q / q16_*), the operand of occ_sca_fused_forward_q16v,
// for value maps that were projected in fp32 (feature inputs other than the backbone's bf16 NHWC maps; shapes the resident
// projection does not cover).  v: (groups, S, C) fp32 contiguous, C = heads * 32; out: (groups, S + (S & 1), C) int16 in the
// pair order [group][pix >> 1][head][pix & 1][32] (the pad row of an odd S is not written); scale: one DEVICE float s (a power
// of two with max|v| * s <= 2^15 — ext.f16_range_scaled's rule), NULL = 1.
namespace occ {
__global__ __launch_bounds__(256) void sca_rows_encode_q16_kernel(const float* __restrict__ v, uint4* __restrict__ out,
                                                                  const float* __restrict__ scale, long groups, int S,
                                                                  int C) {
  const int pieces = C / 8;
  const long n = groups * S * pieces;
  const int eosc = __builtin_amdgcn_frexp_expf(scale != nullptr ? *scale : 1.f) - 2;      // log2(s / 2)
  const int Sp = S + (S & 1);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int pc = (int)(i % pieces);
    const long row = i / pieces;
    const int pix = (int)(row % S);
    const long g = row / S;
    const float4 a = *reinterpret_cast<const float4*>(v + row * C + pc * 8);
    const float4 b = *reinterpret_cast<const float4*>(v + row * C + pc * 8 + 4);
    const float u[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(u[j]));
    const int E = q16_group_exponent(m, eosc);
    const float f = ldexpf(1.f, eosc + 15 - E);
    const int r0 = E & 3, r1 = E >> 2;
    uint4 o;
    o.x = q16_pair_tagged(u[0], u[1], f * 0.25f, -0.25f * (float)r0, -0.25f * (float)r1, r0, r1, 2);
    o.y = q16_pair(u[2], u[3], f);
    o.z = q16_pair(u[4], u[5], f);
    o.w = q16_pair(u[6], u[7], f);
    const int head = pc >> 2, piece = pc & 3;
    // 16-byte units: a pixel pair holds C / 32 heads x 8 units (2 pixels x 4 pieces)
    out[((g * (Sp / 2) + (pix >> 1)) * (C / 32) + head) * 8 + (pix & 1) * 4 + piece] = o;
  }
}
}  // namespace occ

extern "C" int occ_sca_rows_encode_q16(const float* v, void* out, const float* scale, int64_t groups, int S, int C,
                                       void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(v && out, "sca_rows_encode_q16: null pointer argument");
  OCC_CHECK_ARG(groups > 0 && S > 0 && C > 0, "sca_rows_encode_q16: bad dimension");
  if (C % 32) {
    set_error("sca_rows_encode_q16: C=%d is not a multiple of 32 (whole heads)", C);
    return OCC_E_UNSUPPORTED;
  }
  const long n = groups * S * (C / 8);
  long blocks = (n + 255) / 256;
  blocks = blocks > 8192 ? 8192 : blocks;
  hipLaunchKernelGGL(sca_rows_encode_q16_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), v,
                     reinterpret_cast<uint4*>(out), scale, (long)groups, S, C);
  OCC_CHECK_LAUNCH("sca_rows_encode_q16");
  return OCC_OK;
}
