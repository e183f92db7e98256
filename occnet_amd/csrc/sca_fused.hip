// Fused spatial cross-attention gather for gfx950.
//
// Replaces, in one launch per encoder layer (reference: projects/mmdet3d_plugin/bevformer/modules/
// spatial_cross_attention.py):
//   :136-153  per-camera visible-query index lists + rebatch gather (nonzero() host sync, padding)
//   :338-373  softmax over L*P logits, offsets/(W_l,H_l), + z-anchor reference point (anchor p % Z)
//   :386-396  ms_deform_attn_forward on (bs*6, max_len) padded rows
//   :165-173  scatter-add into slots, count of visible cameras, divide
// Query-major instead of camera-major: one 64-lane wave owns one BEV query (all 8 heads, 8 lanes x 4
// channels per head) and loops over the cameras that see it (batch 0's mask decides, as the
// reference does), so nothing is padded, nothing is scattered and the camera mean is a register
// reduction in the reference's camera order.  The softmax and the normalised offsets do not depend on
// the camera and are computed once per query.  Per (query, camera): 256 samples are resolved by the
// wave (4 per lane) into LDS, then every 8-lane group gathers its head's 32 samples:
// 128 x 16-byte loads per lane, 16 in flight.
//
// fp16 value rows (sca_fused_h_kernel, default for inference since round 3): the projected value maps are stored as fp16
// (the value projection's fp16 epilogue), so one head's 32 channels of one pixel are 64 bytes = FOUR lanes x 16 bytes.
// The texture path retires ~one 1 KB wave load per ~21 clocks whatever the lanes ask for (tools_dev/ta_probe.hip:
// 47-50 B/clk/CU for random 128-byte rows, random 64-byte rows fetched 8 B per lane, and fully coalesced 1 KB alike), so
// the cost of a row is the number of LANES it occupies: with 16 B per lane a wave instruction fetches 16 fp16 rows
// instead of 8 fp32 rows.  The wave's 16 lane-groups = 8 heads x 2 halves of the head's L*P samples; every lane keeps 8
// fp32 accumulators (v_fma_mix_f32 widens the fp16 operand inside the FMA) and the two halves meet in one cross-lane
// add at the end.  Round 2's fp16 experiment kept 8 lanes x 8 B per row — the same number of wave loads per row — and
// measured no gain for exactly this reason.  Sampling arithmetic, weights and accumulation stay fp32; the value elements
// carry 11 significant bits (end-to-end effect at full size: tests/test_gpu_fullsize.py, bench parity leg).  Its loads
// run through a ROLLING window (the loads of sample j + DEPTH are requested as soon as sample j is consumed) instead of
// batches that drain to zero before the next batch is requested; its softmax keeps a head's consecutive samples in one
// lane (6 cross-lane operations per query instead of 40).  The fp32-row kernel (sca_fused_kernel) is round 2's, unchanged.
// PIXEL-PAIR LAYOUT of the fp16 maps: value_f16[b * NC + c][pix >> 1][head][pix & 1][32] — the two x-neighbours (2k, 2k + 1)
// of one head share one 128-byte line (a 64-byte row alone is half a line: every corner cost a whole L1 miss for half
// its bytes).  The wide PMC sweep (profiles/r03_pmc_wide_sca_linear.txt) shows what binds the gather: 65 % of its L1
// line accesses miss, the TCP sits 51 % of the launch in PENDING stall (its outstanding-miss capacity / the 232-cycle
// L2 round trip) and the TD 96 % busy.  With pairs the left / right corners of a sample hit the same line half the
// time: L1 -> L2 requests 31.3 M -> 21.3 M per launch, 0.178 -> 0.156 ms (profiles/r03_sca_pair_layout.txt).  The
// value projection's fp16 epilogue writes this layout (csrc/value_proj_bf16.hip); S must be even (padded).
// Round 6, what bounds it now (profiles/r06_c6_sca_fine_levels_only.txt): with the COARSE half of the samples (levels 2-3, half of
// every head's 32) not gathered at all the launch drops from 206 to 166 us — so the lead of serving those levels from an
// LDS-resident copy (a camera / head-major pass) is capped at ~40 us minus what the LDS pass itself costs; the other ~165 us are
// the fine-level gather (~60), the 164 MB stream of query-Linear outputs and results (~42), the prologue and the per-camera
// set-up.  Not pursued: the restructuring needs per-camera query lists, a partial-sum side buffer and 8 waves per CU.
#include <cstdlib>
#include "common.h"

namespace occ {

constexpr int kScaWaves = 4;

// three waves per SIMD: unconstrained, hipcc spends 172 VGPRs (two waves per SIMD); at <= 168 a third wave fits
// and the TA-bound gather runs 10 % faster (0.316 -> 0.285 ms); four waves need spills and are slower (0.308 ms)
template <int L, int P>
__global__ __launch_bounds__(256, 3) void sca_fused_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lstart, const float* __restrict__ offs, long offs_stride,
    const float* __restrict__ logits, long logits_stride, const float* __restrict__ ref_cam,
    const uint32_t* __restrict__ vis_bits, const int32_t* __restrict__ order,
    float* __restrict__ slots, unsigned long long* __restrict__ stats, int B, int NC, int S, int Z,
    int Nq) {
  constexpr int M = 8, D = 32, LP = L * P;
  constexpr int K = M * LP / 64;  // samples resolved per lane
  static_assert(LP >= 8 && LP <= 32 && (LP & (LP - 1)) == 0, "L*P must be a power of two in [8,32]");
  constexpr int LPp = LP + 1;
  __shared__ __attribute__((aligned(16))) SampleParamB smem[kScaWaves * M * LPp];
  // the camera-independent per-sample terms (softmax weight, normalised offset) wait in LDS, not in registers:
  // live across the gather they push the kernel over the 168 VGPRs of three waves per SIMD (10 scratch spills =
  // 100 MB of extra writes per launch)
  __shared__ __attribute__((aligned(16))) float pre[kScaWaves][3][K][64];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const long wg = (long)blockIdx.x * kScaWaves + wave;
  if (wg >= (long)B * Nq) return;
  const int b = (int)(wg / Nq);
  const int r = (int)(wg - (long)b * Nq);
  const int q = order ? order[r] : r;
  SampleParamB* sp = smem + wave * M * LPp;

  constexpr int row_stride = M * D;
  const uint32_t vis = vis_bits[q];                       // batch 0's mask picks the cameras
  const uint32_t own = vis_bits[(long)b * Nq + q];        // this batch's mask gives the divisor
  const int count = __builtin_popcount(own);

  // ---- camera-independent part: softmax(logits) and offsets / (W_l, H_l) ------------------
  float aw[K], ox[K], oy[K];
  // sample index s = (lane + 64 k) % LP does not depend on k (64 % LP == 0): one level per lane
  static_assert(64 % LP == 0, "the lane's level must not depend on k");
  const int lane_l = (lane % LP) / P;
  const int lvH = (int)shapes[2 * lane_l], lvW = (int)shapes[2 * lane_l + 1], lvS = (int)lstart[lane_l];
  const float* lrow = logits + ((long)b * Nq + q) * logits_stride;
  const float* orow = offs + ((long)b * Nq + q) * offs_stride;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int idx = lane + 64 * k;  // = m*LP + s
    float x = lrow[idx];
    float mx = x;
#pragma unroll
    for (int d = LP / 2; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    const float e = expf(x - mx);
    float sum = e;
#pragma unroll
    for (int d = LP / 2; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
    aw[k] = fdiv(e, sum);                      // (fdiv, not `/`: common.h)
    const float2 o = *reinterpret_cast<const float2*>(orow + 2 * idx);
    ox[k] = fdiv(o.x, (float)lvW);
    oy[k] = fdiv(o.y, (float)lvH);
    pre[wave][0][k][lane] = aw[k];
    pre[wave][1][k][lane] = ox[k];
    pre[wave][2][k][lane] = oy[k];
  }

  const int g = lane >> 3, c4 = lane & 7;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  unsigned n_in = 0, n_rows = 0;

  for (int c = 0; c < NC; ++c) {
    if (!((vis >> c) & 1u)) continue;  // wave-uniform
    const float* rp = ref_cam + (((long)c * B + b) * Nq + q) * Z * 2;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int idx = lane + 64 * k;
      const int m = idx / LP, s = idx % LP;
      const int z = (s % P) % Z;  // point p pairs with z-anchor p % Z (view(.., P//Z, Z, 2))
      const float2 rxy = *reinterpret_cast<const float2*>(rp + 2 * z);
      SampleParamB p;
      const float aw_k = pre[wave][0][k][lane], ox_k = pre[wave][1][k][lane], oy_k = pre[wave][2][k][lane];
      n_in += bilinear_setup_b(rxy.x + ox_k, rxy.y + oy_k, aw_k, lvH, lvW, lvS,
                               (unsigned)row_stride * 4u, kOobOffset, 1, p);
      sp[m * LPp + s] = p;
    }
    wave_lds_sync();
    // corners outside their map carry an out-of-range byte offset: the buffer load returns 0 without a request
    // (round 1 issued a dummy load of row 0 for them: 9 % of the rows through the texture path, and 0 * Inf)
    const __amdgpu_buffer_rsrc_t rsrc =
        uniform_rsrc(value + ((long)b * NC + c) * S * row_stride, (unsigned)S * row_stride * 4u);
    acc = gather_samples_buf<4>(rsrc, (unsigned)(g * D + c4 * 4) * 4u, sp + g * LPp, LP, acc);
    wave_lds_sync();  // WAR: next camera rewrites the LDS slab
    ++n_rows;
  }

  const float inv = (float)(count > 0 ? count : 1);
  float4 o4 = make_float4(fdiv(acc.x, inv), fdiv(acc.y, inv), fdiv(acc.z, inv), fdiv(acc.w, inv));
  *reinterpret_cast<float4*>(slots + ((long)b * Nq + q) * row_stride + g * D + c4 * 4) = o4;

  if (stats) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) n_in += __shfl_xor(n_in, d);
    if (lane == 0) {
      atomicAdd(&stats[0], (unsigned long long)n_rows);
      atomicAdd(&stats[1], (unsigned long long)n_in);
    }
  }
}

template <int L, int P>
static int launch_sca(const float* value, const int64_t* shapes, const int64_t* lstart,
                      const float* offs, long offs_stride, const float* logits, long logits_stride,
                      const float* ref_cam, const uint32_t* vis_bits, const int32_t* order,
                      float* slots, uint64_t* stats, int B, int NC, int S, int Z, int Nq,
                      hipStream_t st) {
  const long waves = (long)B * Nq;
  const long blocks = (waves + kScaWaves - 1) / kScaWaves;
  hipLaunchKernelGGL((sca_fused_kernel<L, P>), dim3((unsigned)blocks), dim3(256), 0, st, value,
                     shapes, lstart, offs, offs_stride, logits, logits_stride, ref_cam, vis_bits,
                     order, slots, reinterpret_cast<unsigned long long*>(stats), B, NC, S, Z, Nq);
  OCC_CHECK_LAUNCH("sca_fused_forward");
  return OCC_OK;
}

// fp16 value rows (see the file header): 4 lanes x 16 B per head row, two sample halves per head, rolling load window
template <int L, int P, int WPS, int DEPTH, bool Q>
__global__ __launch_bounds__(256, WPS) void sca_fused_h_kernel(
    const void* __restrict__ value_, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lstart, const float* __restrict__ offs, long offs_stride,
    const float* __restrict__ logits, long logits_stride, const float* __restrict__ ref_cam,
    const uint32_t* __restrict__ vis_bits, const int32_t* __restrict__ order,
    float* __restrict__ slots, unsigned long long* __restrict__ stats, int B, int NC, int S, int Z,
    int Nq, const float* __restrict__ value_scale) {
  constexpr int M = 8, D = 32, LP = L * P;
  constexpr int K = M * LP / 64;  // samples resolved per lane
  static_assert(LP >= 8 && LP <= 32 && (LP & (LP - 1)) == 0, "L*P must be a power of two in [8,32]");
  constexpr int LPp = LP + 1;
  __shared__ __attribute__((aligned(16))) SampleParamB smem[kScaWaves * M * LPp];
  // the camera-independent per-sample terms (softmax weight, normalised offset) stay in registers here: the fp16 gather
  // has registers to spare (100 VGPRs), and without the 12 KB `pre` slab of the fp32 kernel four blocks fit a CU's LDS

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const long wg = (long)blockIdx.x * kScaWaves + wave;
  if (wg >= (long)B * Nq) return;
  const int b = (int)(wg / Nq);
  const int r = (int)(wg - (long)b * Nq);
  const int q = order ? order[r] : r;
  SampleParamB* sp = smem + wave * M * LPp;

  constexpr int row_stride = M * D;
  constexpr unsigned EV = 2u;                              // bytes per value element (fp16)
  const char* value = reinterpret_cast<const char*>(value_);
  const uint32_t vis = vis_bits[q];                       // batch 0's mask picks the cameras
  const uint32_t own = vis_bits[(long)b * Nq + q];        // this batch's mask gives the divisor
  const int count = __builtin_popcount(own);

  // ---- camera-independent part: softmax(logits) and offsets / (W_l, H_l) ------------------
  // lane = (head m = lane / 8, sample group sK = lane % 8) holds the K CONSECUTIVE samples s = sK K + k of its head: the
  // logits / offsets arrive as one / two vector loads and the softmax is a K-term local reduction + 3 shuffle steps over
  // the head's 8 lanes.  (One sample per (lane, k) with s = lane % LP — the fp32 kernel's map — costs 5 shuffle steps
  // per max and per sum for each of the K registers: 40 cross-lane operations per query instead of 6; the prologue was
  // 0.021 ms of the 0.19 ms launch, profiles/r03_sca_ablation.txt.)
  float aw[K], ox[K], oy[K];
  static_assert(P % K == 0 && K * 8 == LP, "a lane's K samples share one level");
  const int hm = lane >> 3, sK = lane & 7;
  const int lane_l = (sK * K) / P;
  const int lvH = (int)shapes[2 * lane_l], lvW = (int)shapes[2 * lane_l + 1], lvS = (int)lstart[lane_l];
  {
    const float* lp = logits + ((long)b * Nq + q) * logits_stride + hm * LP + sK * K;
    const float* op = offs + ((long)b * Nq + q) * offs_stride + 2 * (hm * LP + sK * K);
    float x[K], o[2 * K];
    if constexpr (K == 4) {
      const float4 t = *reinterpret_cast<const float4*>(lp);
      x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w;
      const float4 u0 = *reinterpret_cast<const float4*>(op), u1 = *reinterpret_cast<const float4*>(op + 4);
      o[0] = u0.x; o[1] = u0.y; o[2] = u0.z; o[3] = u0.w; o[4] = u1.x; o[5] = u1.y; o[6] = u1.z; o[7] = u1.w;
    } else {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        x[k] = lp[k];
        const float2 t = *reinterpret_cast<const float2*>(op + 2 * k);
        o[2 * k] = t.x; o[2 * k + 1] = t.y;
      }
    }
    float mx = x[0];
#pragma unroll
    for (int k = 1; k < K; ++k) mx = fmaxf(mx, x[k]);
#pragma unroll
    for (int d = 4; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) { x[k] = expf(x[k] - mx); sum += x[k]; }
#pragma unroll
    for (int d = 4; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      aw[k] = fdiv(x[k], sum);                 // (fdiv, not `/`: see common.h)
      ox[k] = fdiv(o[2 * k], (float)lvW);
      oy[k] = fdiv(o[2 * k + 1], (float)lvH);
    }
  }

  // gather map: 4 lanes x 8 channels per head row, the two halves of the wave take the two halves of a head's samples
  const int g = (lane >> 2) & 7, c4 = lane & 3, half = lane >> 5;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), acc2 = acc;
  unsigned n_in = 0, n_rows = 0;

  for (int c = 0; c < NC; ++c) {
    if (!((vis >> c) & 1u)) continue;  // wave-uniform
    const float* rp = ref_cam + (((long)c * B + b) * Nq + q) * Z * 2;
    // the K anchor points of this lane are requested together: inside the loop below each one sat behind the branches of
    // the previous sample's set-up — K serial round trips per (query, camera) (round 4 ISA reading)
    float2 rxy_k[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int z = ((sK * K + k) % P) % Z;  // point p pairs with z-anchor p % Z (view(.., P//Z, Z, 2))
      rxy_k[k] = *reinterpret_cast<const float2*>(rp + 2 * z);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int m = hm;
      const float2 rxy = rxy_k[k];
      SampleParamB p;
      const float aw_k = aw[k], ox_k = ox[k], oy_k = oy[k];
      n_in += bilinear_setup_b(rxy.x + ox_k, rxy.y + oy_k, aw_k, lvH, lvW, lvS,
                               (unsigned)row_stride * EV, kOobOffset, 1, p);
      // pixel-pair layout (file header): byte offset pix * 512 -> (pix >> 1) * 1024 + (pix & 1) * 64; the out-of-range
      // marker stays out of range
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) p.o[kk] = (p.o[kk] & ~1023u) | ((p.o[kk] >> 3) & 64u);
      // slot k * 8 + sK, not s: the 8 lanes of a head write consecutive 32-byte entries (conflict-free); the gather
      // sums a head's LP entries, so their order in the slab is free
      sp[m * LPp + k * 8 + sK] = p;
    }
    wave_lds_sync();
    // corners outside their map carry an out-of-range byte offset: the buffer load returns 0 without a request
    // (round 1 issued a dummy load of row 0 for them: 9 % of the rows through the texture path, and 0 * Inf)
    const __amdgpu_buffer_rsrc_t rsrc =
        uniform_rsrc(value + ((long)b * NC + c) * S * row_stride * EV, (unsigned)S * row_stride * EV);
    gather_samples_buf_h<LP / 2, DEPTH, Q>(rsrc, (unsigned)(g * 128 + c4 * 16), sp + g * LPp + half * (LP / 2), acc, acc2);
    wave_lds_sync();  // WAR: next camera rewrites the LDS slab
    ++n_rows;
  }

  // value_scale: the power-of-two range scale s the projection stored the fp16 rows with (value_range.hip); dividing by
  // count * s undoes it exactly
  // (q16 rows: the mantissas additionally carry 2^15 / 2 — stored under s / 2 —, common.h fma8q)
  const float inv = (float)(count > 0 ? count : 1) * (value_scale != nullptr ? *value_scale : 1.f) * (Q ? 16384.f : 1.f);
  // the two sample halves of a head sit 32 lanes apart
  acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32); acc.z += __shfl_xor(acc.z, 32);
  acc.w += __shfl_xor(acc.w, 32);
  acc2.x += __shfl_xor(acc2.x, 32); acc2.y += __shfl_xor(acc2.y, 32); acc2.z += __shfl_xor(acc2.z, 32);
  acc2.w += __shfl_xor(acc2.w, 32);
  if (half == 0) {
    float* dst = slots + ((long)b * Nq + q) * row_stride + g * D + c4 * 8;
    *reinterpret_cast<float4*>(dst) = make_float4(fdiv(acc.x, inv), fdiv(acc.y, inv), fdiv(acc.z, inv), fdiv(acc.w, inv));
    *reinterpret_cast<float4*>(dst + 4) = make_float4(fdiv(acc2.x, inv), fdiv(acc2.y, inv), fdiv(acc2.z, inv), fdiv(acc2.w, inv));
  }

  if (stats) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) n_in += __shfl_xor(n_in, d);
    if (lane == 0) {
      atomicAdd(&stats[0], (unsigned long long)n_rows);
      atomicAdd(&stats[1], (unsigned long long)n_in);
    }
  }
}

// ---- head-major variant (round 6) --------------------------------------------------------------------------------------------
// One wave = 8 consecutive BEV queries (of the tile order) x ONE head; the head is blockIdx & 7, i.e. the XCD the block is
// dispatched to (block b runs on XCD b % 8): every CU — and every XCD's L2 — sees the value rows of a single head, and the 8
// neighbouring queries of a wave sample neighbouring pixels of the same plane, so their corner rows share L1 lines (at the
// coarse FPN levels almost all of them).  The query-major kernel above gives a wave 8 heads = 8 disjoint planes: no line is
// shared inside a wave.  Same arithmetic per sample, same slab / window machinery; what changes is who holds what:
//   prologue / set-up lane = (query slot j = lane >> 3, sample group sK = lane & 7) — the K consecutive samples of head m;
//   gather lane = (query slot g = (lane >> 2) & 7, 16-byte piece c4 = lane & 3, sample half = lane >> 5);
//   cameras: the loop runs over the UNION of the 8 queries' visible cameras, a query that does not see the camera resolves
//   dead samples (weight 0, out-of-range offsets: no memory request).
template <int L, int P, int WPS, int DEPTH, bool Q, int WB = kScaWaves>
__global__ __launch_bounds__(64 * WB, WPS * 4 / WB) void sca_fused_hm_kernel(
    const void* __restrict__ value_, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lstart, const float* __restrict__ offs, long offs_stride,
    const float* __restrict__ logits, long logits_stride, const float* __restrict__ ref_cam,
    const uint32_t* __restrict__ vis_bits, const int32_t* __restrict__ order,
    float* __restrict__ slots, unsigned long long* __restrict__ stats, int B, int NC, int S, int Z,
    int Nq, int tiles_per_b, const float* __restrict__ value_scale) {
  constexpr int M = 8, D = 32, LP = L * P;
  constexpr int K = M * LP / 64;  // samples resolved per lane
  static_assert(LP >= 8 && LP <= 32 && (LP & (LP - 1)) == 0, "L*P must be a power of two in [8,32]");
  static_assert(P % K == 0 && K * 8 == LP, "a lane's K samples share one level");
  constexpr int LPp = LP + 1;
  __shared__ __attribute__((aligned(16))) SampleParamB smem[WB * 8 * LPp];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int m = (int)(blockIdx.x & 7u);                    // the block's head = its XCD
  const int t = (int)(blockIdx.x >> 3);
  const int b = t / tiles_per_b;
  const int r0 = ((t - b * tiles_per_b) * WB + wave) * 8;     // the wave's first query (position in `order`)
  if (r0 >= Nq) return;
  SampleParamB* sp = smem + wave * 8 * LPp;

  constexpr int row_stride = M * D;
  constexpr unsigned EV = 2u;                              // bytes per value element (fp16 / q16)
  const char* value = reinterpret_cast<const char*>(value_);

  // ---- set-up mapping: lane = (query slot j, sample group sK) -----------------------------------------------------------
  const int j = lane >> 3, sK = lane & 7;
  const bool jvalid = r0 + j < Nq;
  const int rj = jvalid ? r0 + j : Nq - 1;
  const int q = order ? order[rj] : rj;
  const uint32_t vis = jvalid ? vis_bits[q] : 0u;          // batch 0's mask picks the cameras (the reference's quirk)
  uint32_t uni = vis;                                      // the union over the wave's 8 queries: the camera loop's trip list
  uni |= (uint32_t)__shfl_xor((int)uni, 8);
  uni |= (uint32_t)__shfl_xor((int)uni, 16);
  uni |= (uint32_t)__shfl_xor((int)uni, 32);
  uni = __builtin_amdgcn_readfirstlane(uni);

  float aw[K], ox[K], oy[K];
  // the lane's level: the L map shapes arrive as scalar loads (wave-uniform addresses) and are selected per lane — indexed by
  // the lane's level they were three vector loads per wave, and the texture path charges a wave instruction ~21 clocks whatever
  // its lanes fetch (round 6, second half: the non-gather loads were 15 % of the launch's wave loads)
  const int lane_l = (sK * K) / P;
  int lvH = (int)shapes[0], lvW = (int)shapes[1], lvS = (int)lstart[0];
#pragma unroll
  for (int l = 1; l < L; ++l) {
    const int hl = (int)shapes[2 * l], wl = (int)shapes[2 * l + 1], sl = (int)lstart[l];
    lvH = lane_l == l ? hl : lvH; lvW = lane_l == l ? wl : lvW; lvS = lane_l == l ? sl : lvS;
  }
  {
    const float* lp = logits + ((long)b * Nq + q) * logits_stride + m * LP + sK * K;
    const float* op = offs + ((long)b * Nq + q) * offs_stride + 2 * (m * LP + sK * K);
    float x[K], o[2 * K];
    if constexpr (K == 4) {
      const float4 tt = *reinterpret_cast<const float4*>(lp);
      x[0] = tt.x; x[1] = tt.y; x[2] = tt.z; x[3] = tt.w;
      const float4 u0 = *reinterpret_cast<const float4*>(op), u1 = *reinterpret_cast<const float4*>(op + 4);
      o[0] = u0.x; o[1] = u0.y; o[2] = u0.z; o[3] = u0.w; o[4] = u1.x; o[5] = u1.y; o[6] = u1.z; o[7] = u1.w;
    } else {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        x[k] = lp[k];
        const float2 tt = *reinterpret_cast<const float2*>(op + 2 * k);
        o[2 * k] = tt.x; o[2 * k + 1] = tt.y;
      }
    }
    float mx = x[0];
#pragma unroll
    for (int k = 1; k < K; ++k) mx = fmaxf(mx, x[k]);
#pragma unroll
    for (int d = 4; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) { x[k] = expf(x[k] - mx); sum += x[k]; }
#pragma unroll
    for (int d = 4; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      aw[k] = fdiv(x[k], sum);
      ox[k] = fdiv(o[2 * k], (float)lvW);
      oy[k] = fdiv(o[2 * k + 1], (float)lvH);
    }
  }

  // ---- gather mapping: lane = (query slot g, 16-byte piece c4, sample half) ------------------------------------------------
  const int g = (lane >> 2) & 7, c4 = lane & 3, half = lane >> 5;
  const bool gvalid = r0 + g < Nq;
  const int qg = __shfl(q, g * 8);                                     // slot g's query: held by the set-up lanes g * 8 ..
  // this batch's mask gives the divisor (batch 0: the word the set-up lanes already hold)
  const uint32_t own = b == 0 ? (uint32_t)__shfl((int)vis, g * 8) : (gvalid ? vis_bits[(long)b * Nq + qg] : 0u);
  const int count = __builtin_popcount(own);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), acc2 = acc;
  unsigned n_in = 0, n_rows = 0;

  for (int c = 0; c < NC; ++c) {
    if (!((uni >> c) & 1u)) continue;  // wave-uniform
    const int live = (int)((vis >> c) & 1u);
    const float* rp = ref_cam + (((long)c * B + b) * Nq + q) * Z * 2;
    float2 rxy_k[K];
    if (K == 4 && P % 4 == 0 && Z % 4 == 0) {
      // the lane's four anchors are consecutive (z0 .. z0 + 3, z0 a multiple of 4): two 16-byte loads instead of four 8-byte ones
      const int z0 = ((sK * K) % P) % Z;
      const float4 t0 = *reinterpret_cast<const float4*>(rp + 2 * z0), t1 = *reinterpret_cast<const float4*>(rp + 2 * z0 + 4);
      rxy_k[0] = make_float2(t0.x, t0.y); rxy_k[1] = make_float2(t0.z, t0.w);
      rxy_k[2 % K] = make_float2(t1.x, t1.y); rxy_k[3 % K] = make_float2(t1.z, t1.w);
    } else {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int z = ((sK * K + k) % P) % Z;  // point p pairs with z-anchor p % Z (view(.., P//Z, Z, 2))
        rxy_k[k] = *reinterpret_cast<const float2*>(rp + 2 * z);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      SampleParamB p;
      n_in += bilinear_setup_b(rxy_k[k].x + ox[k], rxy_k[k].y + oy[k], aw[k], lvH, lvW, lvS,
                               (unsigned)row_stride * EV, kOobOffset, live, p);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) p.o[kk] = (p.o[kk] & ~1023u) | ((p.o[kk] >> 3) & 64u);     // pixel-pair layout
      sp[j * LPp + k * 8 + sK] = p;
    }
    wave_lds_sync();
    const __amdgpu_buffer_rsrc_t rsrc =
        uniform_rsrc(value + ((long)b * NC + c) * S * row_stride * EV, (unsigned)S * row_stride * EV);
    gather_samples_buf_h<LP / 2, DEPTH, Q>(rsrc, (unsigned)(m * 128 + c4 * 16), sp + g * LPp + half * (LP / 2), acc, acc2);
    wave_lds_sync();  // WAR: next camera rewrites the LDS slab
    n_rows += (unsigned)(live & (sK == 0) & (m == 0));
  }

  const float inv = (float)(count > 0 ? count : 1) * (value_scale != nullptr ? *value_scale : 1.f) * (Q ? 16384.f : 1.f);
  acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32); acc.z += __shfl_xor(acc.z, 32);
  acc.w += __shfl_xor(acc.w, 32);
  acc2.x += __shfl_xor(acc2.x, 32); acc2.y += __shfl_xor(acc2.y, 32); acc2.z += __shfl_xor(acc2.z, 32);
  acc2.w += __shfl_xor(acc2.w, 32);
  if (half == 0 && gvalid) {
    float* dst = slots + ((long)b * Nq + qg) * row_stride + m * D + c4 * 8;
    *reinterpret_cast<float4*>(dst) = make_float4(fdiv(acc.x, inv), fdiv(acc.y, inv), fdiv(acc.z, inv), fdiv(acc.w, inv));
    *reinterpret_cast<float4*>(dst + 4) = make_float4(fdiv(acc2.x, inv), fdiv(acc2.y, inv), fdiv(acc2.z, inv), fdiv(acc2.w, inv));
  }

  if (stats) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { n_in += __shfl_xor(n_in, d); n_rows += __shfl_xor(n_rows, d); }
    if (lane == 0) {
      if (n_rows) atomicAdd(&stats[0], (unsigned long long)n_rows);
      atomicAdd(&stats[1], (unsigned long long)n_in);
    }
  }
}

template <int L, int P, bool Q, int WPS = 4, int DEPTH = 2, int WB = kScaWaves>
static int launch_sca_hm(const void* value, const int64_t* shapes, const int64_t* lstart,
                         const float* offs, long offs_stride, const float* logits, long logits_stride,
                         const float* ref_cam, const uint32_t* vis_bits, const int32_t* order,
                         float* slots, uint64_t* stats, int B, int NC, int S, int Z, int Nq,
                         hipStream_t st, const float* value_scale) {
  const int tiles_per_b = (Nq + WB * 8 - 1) / (WB * 8);
  const long blocks = (long)B * tiles_per_b * 8;
  hipLaunchKernelGGL((sca_fused_hm_kernel<L, P, WPS, DEPTH, Q, WB>), dim3((unsigned)blocks), dim3(64 * WB), 0, st, value,
                     shapes, lstart, offs, offs_stride, logits, logits_stride, ref_cam, vis_bits,
                     order, slots, reinterpret_cast<unsigned long long*>(stats), B, NC, S, Z, Nq, tiles_per_b, value_scale);
  OCC_CHECK_LAUNCH(Q ? "sca_fused_forward_q16v (head-major)" : "sca_fused_forward_f16v (head-major)");
  return OCC_OK;
}

template <int L, int P, bool Q, int WPS = 4, int DEPTH = 2>
static int launch_sca_h(const void* value, const int64_t* shapes, const int64_t* lstart,
                        const float* offs, long offs_stride, const float* logits, long logits_stride,
                        const float* ref_cam, const uint32_t* vis_bits, const int32_t* order,
                        float* slots, uint64_t* stats, int B, int NC, int S, int Z, int Nq,
                        hipStream_t st, const float* value_scale) {
  const long waves = (long)B * Nq;
  const long blocks = (waves + kScaWaves - 1) / kScaWaves;
  hipLaunchKernelGGL((sca_fused_h_kernel<L, P, WPS, DEPTH, Q>), dim3((unsigned)blocks), dim3(256), 0, st, value,
                     shapes, lstart, offs, offs_stride, logits, logits_stride, ref_cam, vis_bits,
                     order, slots, reinterpret_cast<unsigned long long*>(stats), B, NC, S, Z, Nq, value_scale);
  OCC_CHECK_LAUNCH(Q ? "sca_fused_forward_q16v" : "sca_fused_forward_f16v");
  return OCC_OK;
}

}  // namespace occ

namespace occ {
static int sca_dispatch(const void* value, int rowfmt /* 0 f32, 1 f16, 2 q16 */, const int64_t* spatial_shapes,
                        const int64_t* level_start_index, const float* offs, int64_t offs_stride,
                        const float* logits, int64_t logits_stride, const float* ref_cam,
                        const uint32_t* vis_bits, const int32_t* order, float* slots, uint64_t* stats, int B,
                        int NC, int S, int M, int D, int L, int P, int Z, int Nq, void* stream,
                        const float* value_scale = nullptr) {
  OCC_CHECK_ARG(value && spatial_shapes && level_start_index && offs && logits && ref_cam &&
                    vis_bits && slots,
                "sca_fused_forward: null pointer argument");
  OCC_CHECK_ARG(B > 0 && NC > 0 && NC <= 32 && S > 0 && Nq > 0 && Z > 0 && L > 0 && P > 0,
                "sca_fused_forward: bad dimension (B=%d NC=%d S=%d Nq=%d Z=%d L=%d P=%d)", B, NC, S,
                Nq, Z, L, P);
  OCC_CHECK_ARG(P % Z == 0, "sca_fused_forward: num_points(%d) must be a multiple of Z(%d)", P, Z);
  const bool halfv = rowfmt != 0;
  OCC_CHECK_ARG(!halfv || S % 2 == 0, "sca_fused_forward: 16-bit value maps are stored in pixel pairs: S(%d) must be even", S);
  OCC_CHECK_ARG(offs_stride >= (int64_t)M * L * P * 2 && logits_stride >= (int64_t)M * L * P,
                "sca_fused_forward: row strides smaller than a row");
  OCC_CHECK_ARG((long)S * M * D * 4 < (long)kOobOffset, "sca_fused_forward: value batch entry too large");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (M != 8 || D != 32) {
    set_error("sca_fused_forward: no fused kernel for M=%d D=%d", M, D);
    return OCC_E_UNSUPPORTED;
  }
  // (fp16 rows: 4 waves per SIMD with a 2-sample rolling window is the default; 3/2, 6/1, 5/2, 3/4 and 8/1 were measured
  // through a development switch, since removed: profiles/r03_sca_probe_fp16_rows.txt)
  // The 16-bit-row gathers run on the head-major kernel (default since round 6: 0.196 against 0.218 ms per launch, same box,
  // profiles/r06_c9_sca_head_major_sweep.txt); OCC_SCA_HEAD_MAJOR=0 selects the query-major kernel (read per call: tests
  // switch it inside one process).  (waves / SIMD, window, waves / block) = (4, 2, 4) measured best of (3, 4, 4) 0.2047,
  // (4, 2, 8) 0.2051, (4, 1, 4) 0.2044 ms.
  const char* hm_env = getenv("OCC_SCA_HEAD_MAJOR");
  const bool head_major = !(hm_env != nullptr && hm_env[0] == '0');
#define OCC_SCA_CASE(LL, PP)                                                                       \
  if (L == LL && P == PP) {                                                                        \
    if (rowfmt == 2 && head_major)                                                                 \
      return launch_sca_hm<LL, PP, true>(value, spatial_shapes, level_start_index, offs, (long)offs_stride, logits, \
                                         (long)logits_stride, ref_cam, vis_bits, order, slots, stats, B, NC, S, Z, Nq, st, \
                                         value_scale);                                                           \
    if (rowfmt == 1 && head_major)                                                                 \
      return launch_sca_hm<LL, PP, false>(value, spatial_shapes, level_start_index, offs, (long)offs_stride, logits, \
                                          (long)logits_stride, ref_cam, vis_bits, order, slots, stats, B, NC, S, Z, Nq, st, \
                                          value_scale);                                                          \
    if (rowfmt == 2)                                                                               \
      return launch_sca_h<LL, PP, true>(value, spatial_shapes, level_start_index, offs, (long)offs_stride, logits, \
                                        (long)logits_stride, ref_cam, vis_bits, order, slots, stats, B, NC, S, Z, Nq, st, \
                                        value_scale);                                                            \
    if (rowfmt == 1)                                                                               \
      return launch_sca_h<LL, PP, false>(value, spatial_shapes, level_start_index, offs, (long)offs_stride, logits, \
                                         (long)logits_stride, ref_cam, vis_bits, order, slots, stats, B, NC, S, Z, Nq, st, \
                                         value_scale);                                                           \
    return launch_sca<LL, PP>(reinterpret_cast<const float*>(value), spatial_shapes, level_start_index, offs,    \
                              (long)offs_stride, logits, (long)logits_stride, ref_cam, vis_bits, order, slots,   \
                              stats, B, NC, S, Z, Nq, st);                                                       \
  }
  OCC_SCA_CASE(4, 8)
  OCC_SCA_CASE(4, 4)
  OCC_SCA_CASE(2, 8)
  OCC_SCA_CASE(1, 8)
#undef OCC_SCA_CASE
  set_error("sca_fused_forward: no fused kernel for L=%d P=%d", L, P);
  return OCC_E_UNSUPPORTED;
}
}  // namespace occ

extern "C" int occ_sca_fused_forward_f32(const float* value, const int64_t* spatial_shapes,
                                         const int64_t* level_start_index, const float* offs,
                                         int64_t offs_stride, const float* logits,
                                         int64_t logits_stride, const float* ref_cam,
                                         const uint32_t* vis_bits, const int32_t* order,
                                         float* slots, uint64_t* stats, int B, int NC, int S, int M,
                                         int D, int L, int P, int Z, int Nq, void* stream) {
  return occ::sca_dispatch(value, 0, spatial_shapes, level_start_index, offs, offs_stride, logits, logits_stride,
                           ref_cam, vis_bits, order, slots, stats, B, NC, S, M, D, L, P, Z, Nq, stream);
}

extern "C" int occ_sca_fused_forward_f16v(const void* value_f16, const int64_t* spatial_shapes,
                                          const int64_t* level_start_index, const float* offs,
                                          int64_t offs_stride, const float* logits,
                                          int64_t logits_stride, const float* ref_cam,
                                          const uint32_t* vis_bits, const int32_t* order,
                                          float* slots, uint64_t* stats, int B, int NC, int S, int M,
                                          int D, int L, int P, int Z, int Nq, const float* value_scale,
                                          void* stream) {
  return occ::sca_dispatch(value_f16, 1, spatial_shapes, level_start_index, offs, offs_stride, logits,
                           logits_stride, ref_cam, vis_bits, order, slots, stats, B, NC, S, M, D, L, P, Z, Nq, stream,
                           value_scale);
}

// q16 value rows (block floating point, common.h fma8q): same layout, geometry and arguments as the fp16-row entry; value_q16 as
// written by occ_value_proj_bf16_planes(out_format = 2) / occ_sca_rows_encode_q16, value_scale the plane's range scale.
extern "C" int occ_sca_fused_forward_q16v(const void* value_q16, const int64_t* spatial_shapes,
                                          const int64_t* level_start_index, const float* offs,
                                          int64_t offs_stride, const float* logits,
                                          int64_t logits_stride, const float* ref_cam,
                                          const uint32_t* vis_bits, const int32_t* order,
                                          float* slots, uint64_t* stats, int B, int NC, int S, int M,
                                          int D, int L, int P, int Z, int Nq, const float* value_scale,
                                          void* stream) {
  return occ::sca_dispatch(value_q16, 2, spatial_shapes, level_start_index, offs, offs_stride, logits,
                           logits_stride, ref_cam, vis_bits, order, slots, stats, B, NC, S, M, D, L, P, Z, Nq, stream,
                           value_scale);
}
