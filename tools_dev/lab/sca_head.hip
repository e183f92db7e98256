// Fused spatial cross-attention gather for gfx950, HEAD-MAJOR decomposition (round 2; the query-major
// sca_fused.hip kernel stays selectable as variant 0).
//
// Same contract as sca_fused.hip — one launch per encoder layer replaces, of the reference's
// projects/mmdet3d_plugin/bevformer/modules/spatial_cross_attention.py:
//   :136-153  per-camera visible-query index lists + rebatch gather (nonzero() host sync, padding)
//   :338-373  softmax over L*P logits, offsets/(W_l,H_l), + z-anchor reference point (anchor p % Z)
//   :386-396  ms_deform_attn_forward on (bs*6, max_len) padded rows
//   :165-173  scatter-add into slots, count of visible cameras, divide
// What changed, and why (profiles/r01_pmc_derived_hotpath_v9.txt: the query-major kernel is bound by the
// texture-addresser / vector-L1 path — 5.8 GB of 128-byte rows per launch through 64 B/clk/CU, L1 misses
// refilled at ~17 B/clk/CU — not by HBM):
//   * a BLOCK works on ONE attention head: blockIdx.x % 8 = head, and the hardware places block b on XCD b % 8,
//     so every XCD's L2 (and every CU's L1) only ever sees ONE head's 1/8 slice of the value maps
//     (23.6 of 189 MB; the two coarse levels of all cameras: 1.4 MB — L2 resident);
//   * a wave owns 8 neighbouring BEV queries ("octet", one row of an 8x8 tile of the BEV plane) of that head:
//     8 lanes x 16 bytes = one 128-byte value row per query, the 8 rows of one load instruction belong to
//     neighbouring pillars = neighbouring pixels;
//   * out-of-map bilinear corners are never requested: rows are fetched with BUFFER loads whose byte offset is
//     out of range for such corners (the hardware returns 0 without touching the cache; 9 % of the query-major
//     kernel's requests were such dummy rows, and 0 * Inf can no longer poison a border sample);
//   * STAGE: the coarsest level's map of (camera, head) — 15x25 px x 128 B = 48 KB — is staged in LDS once per
//     block and camera and its samples (a quarter of all) are read with ds_read_b128 (256 B/clk/CU) instead of
//     through the texture path; the block walks camera-outer over 128-192 queries so one staging pass serves
//     >= 4 000 samples.
// Arithmetic per sample is unchanged (common.h: mmcv's ms_deformable_im2col), so are the camera order of the
// accumulation and the divide by the visible-camera count; only the f32 summation order inside a query differs
// from the query-major kernel (samples are accumulated level by level here as well).
#include <hip/hip_fp16.h>
#include <type_traits>
#include "common.h"

namespace occ {

// NW waves per block, OPW octets (8 queries) per wave; STAGE: last level through LDS (camera-outer block loop)
template <int L, int P, int NW, int OPW, bool STAGE>
__global__ __launch_bounds__(NW * 64, (NW * 64 >= 512) ? 4 : 3) void sca_head_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lstart, const float* __restrict__ offs, long offs_stride,
    const float* __restrict__ logits, long logits_stride, const float* __restrict__ ref_cam,
    const uint32_t* __restrict__ vis_bits, const int32_t* __restrict__ order,
    float* __restrict__ slots, unsigned long long* __restrict__ stats, int B, int NC, int S, int Z,
    int Nq, int stage_pix) {
  constexpr int M = 8, D = 32, LP = L * P;
  constexpr int NCH = LP / 8;                     // 8-sample chunks per (query, head): one sample per lane of a group
  static_assert(P == 4 || P == 8, "a chunk of 8 samples spans at most two levels");
  static_assert(LP % 8 == 0 && LP <= 32, "L*P must be a multiple of 8, at most 32");
  constexpr unsigned ROW_B = M * D * 4;           // bytes between two pixels of a value map (all heads)
  constexpr int GRP_B = 9 * 32;                   // LDS bytes per query group: 8 SampleParamB + 32 pad (banks)
  constexpr int PAR_B = 8 * GRP_B;                // per wave
  extern __shared__ __attribute__((aligned(16))) char lds[];
  // [stage: stage_pix * 128 B + one zero row][params: NW * PAR_B][16 B: block camera mask]
  const int stage_bytes = STAGE ? (stage_pix + 1) * 128 : 0;
  char* par = lds + stage_bytes + (threadIdx.x >> 6) * PAR_B;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qi = lane >> 3, li = lane & 7;
  const int m = blockIdx.x & 7;                   // head (= XCD under round-robin block placement)
  const int chunk = blockIdx.x >> 3;
  const int b = blockIdx.y;

  int Hs[L], Ws[L], Ss[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    Hs[l] = (int)shapes[2 * l]; Ws[l] = (int)shapes[2 * l + 1]; Ss[l] = (int)lstart[l];
  }

  // this lane's query in each of the wave's octets
  int qid[OPW];
  uint32_t visq[OPW], cnt[OPW];
  uint32_t wave_vis = 0;
#pragma unroll
  for (int o = 0; o < OPW; ++o) {
    const long r = ((long)chunk * NW + wave) * (OPW * 8) + o * 8 + qi;
    const bool ok = r < Nq;
    const int q = ok ? (order ? order[r] : (int)r) : 0;
    qid[o] = q;
    visq[o] = ok ? vis_bits[q] : 0u;                              // batch 0's mask picks the cameras
    cnt[o] = ok ? (uint32_t)__builtin_popcount(vis_bits[(long)b * Nq + q]) : 0u;   // own mask: the divisor
    wave_vis |= visq[o];
  }
#pragma unroll
  for (int d = 32; d >= 8; d >>= 1) wave_vis |= __shfl_xor(wave_vis, d);

  uint32_t cams = wave_vis;                       // cameras this wave (STAGE: this block) has to visit
  if (STAGE) {
    // (no static __shared__: it would shift the dynamic region off its 16-byte alignment)
    uint32_t& block_vis = *reinterpret_cast<uint32_t*>(lds + stage_bytes + NW * PAR_B);
    if (threadIdx.x == 0) block_vis = 0;
    // zero row for out-of-map corners of the staged level
    if (threadIdx.x < 32) reinterpret_cast<float*>(lds + stage_pix * 128)[threadIdx.x] = 0.f;
    __syncthreads();
    if (lane == 0 && wave_vis) atomicOr(&block_vis, wave_vis);
    __syncthreads();
    cams = block_vis;
    cams = __builtin_amdgcn_readfirstlane(cams);
  }

  float4 acc[OPW];
#pragma unroll
  for (int o = 0; o < OPW; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
  unsigned n_in = 0, n_rows = 0;

  for (int c = 0; c < NC; ++c) {
    if (!((cams >> c) & 1u)) continue;            // block-uniform (STAGE) / wave-uniform
    const float* vmap = value + ((long)b * NC + c) * S * (M * D) + m * D;
    const __amdgpu_buffer_rsrc_t rsrc = uniform_rsrc(vmap, (unsigned)S * ROW_B - (unsigned)(m * D * 4));
    if (STAGE) {
      __syncthreads();                            // every wave is done with the previous camera's map
      const unsigned pix0 = (unsigned)Ss[L - 1];
      for (int i = threadIdx.x; i < stage_pix * 8; i += NW * 64) {
        const float4 v = buf_load16(rsrc, (pix0 + (unsigned)(i >> 3)) * ROW_B + (unsigned)(i & 7) * 16u);
        *reinterpret_cast<float4*>(lds + i * 16) = v;
      }
      __syncthreads();
    }
    // runtime octet loop (unrolled it doubles the live state and spills): the per-octet values are picked with
    // selects out of their (compile-time indexed) register arrays
#pragma unroll 1
    for (int o = 0; o < OPW; ++o) {
      uint32_t vq = visq[0];
      int q = qid[0];
      float4 a = acc[0];
#pragma unroll
      for (int t = 1; t < OPW; ++t)
        if (o == t) { vq = visq[t]; q = qid[t]; a = acc[t]; }
      const bool live = (vq >> c) & 1u;
      if (__builtin_amdgcn_ballot_w64(live) == 0) continue;   // no query of this octet sees camera c
      const float* lrow = logits + ((long)b * Nq + q) * logits_stride + m * LP;
      const float* orow = offs + ((long)b * Nq + q) * offs_stride + (long)m * LP * 2;
      const float* rp = ref_cam + (((long)c * B + b) * Nq + q) * Z * 2;
      // softmax over the LP samples of (query, head): lane li of the group holds samples li + 8*j.  Only the max
      // and the sum are kept; the logits are re-read (L1) chunk by chunk — a runtime chunk loop, NOT unrolled:
      // unrolled, LLVM hoists every chunk's camera-independent terms out of the camera loop and needs 240 VGPRs
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < NCH; ++j) mx = fmaxf(mx, lrow[8 * j + li]);
      mx = fmaxf(mx, __shfl_xor(mx, 1)); mx = fmaxf(mx, __shfl_xor(mx, 2)); mx = fmaxf(mx, __shfl_xor(mx, 4));
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < NCH; ++j) sum += expf(lrow[8 * j + li] - mx);
      sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);
      if (m == 0 && li == 0 && live) ++n_rows;
      const char* gp = par + qi * GRP_B;

      // one 8-sample chunk.  LAST = the chunk that holds the staged level (compile-time j): only there the LDS
      // path exists — a runtime "this half comes from LDS" branch makes LLVM speculate the ds_reads next to the
      // buffer loads and doubles the data registers (190 VGPRs instead of ~110)
      auto do_chunk = [&](const int j, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        // ---- resolve this lane's sample of chunk j --------------------------------------------------
        const int s = 8 * j + li;
        const int l = s / P;                                       // P = 8: the chunk's level, wave-uniform
        int H = Hs[0], W = Ws[0], st = Ss[0];
#pragma unroll
        for (int t = 1; t < L; ++t)
          if (l == t) { H = Hs[t]; W = Ws[t]; st = Ss[t]; }
        const int z = (s % P) % Z;                                 // point p pairs with z-anchor p % Z
        const float2 of = *reinterpret_cast<const float2*>(orow + 2 * s);
        const float2 rxy = *reinterpret_cast<const float2*>(rp + 2 * z);
        const float aw = expf(lrow[s] - mx) / sum;
        // the staged level is addressed in LDS bytes (128 per pixel, dead corners -> the zero row)
        const bool staged = STAGE && LAST && l == L - 1;
        SampleParamB sp;
        n_in += bilinear_setup_b(rxy.x + of.x / (float)W, rxy.y + of.y / (float)H, aw, H, W, staged ? 0 : st,
                                 staged ? 128u : ROW_B, staged ? (unsigned)stage_pix * 128u : kOobOffset, live,
                                 sp);
        *reinterpret_cast<SampleParamB*>(par + qi * GRP_B + li * 32) = sp;
        wave_lds_sync();
        // ---- gather: group qi walks its 8 samples, 4 at a time (16 rows in flight per lane).  Offsets are read
        // right before their loads and the weights again right before their FMAs: only the 64 data registers
        // stay live across the memory wait
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float4 v[4][4];
          constexpr int last_first = 8 * (NCH - 1);
          const bool from_lds = STAGE && LAST && (last_first + 4 * h) / P == L - 1;   // folds: LAST => j = NCH-1
          if (from_lds) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const occ_u32x4 o4 = *reinterpret_cast<const occ_u32x4*>(gp + (h * 4 + u) * 32 + 16);
#pragma unroll
              for (int k = 0; k < 4; ++k) v[u][k] = *reinterpret_cast<const float4*>(lds + o4[k] + li * 16);
            }
          } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const occ_u32x4 o4 = *reinterpret_cast<const occ_u32x4*>(gp + (h * 4 + u) * 32 + 16);
#pragma unroll
              for (int k = 0; k < 4; ++k) v[u][k] = buf_load16(rsrc, o4[k] + (unsigned)li * 16u);
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float4 w4 = *reinterpret_cast<const float4*>(gp + (h * 4 + u) * 32);
            fma4(a, w4.x, v[u][0]); fma4(a, w4.y, v[u][1]);
            fma4(a, w4.z, v[u][2]); fma4(a, w4.w, v[u][3]);
          }
        }
        wave_lds_sync();                                           // WAR: the next chunk rewrites the slab
      };
      constexpr int NLOOP = STAGE ? NCH - 1 : NCH;
#pragma unroll 1
      for (int j = 0; j < NLOOP; ++j) do_chunk(j, std::false_type{});
      if (STAGE) do_chunk(NCH - 1, std::true_type{});
#pragma unroll
      for (int t = 0; t < OPW; ++t)
        if (o == t) acc[t] = a;
    }
  }

#pragma unroll
  for (int o = 0; o < OPW; ++o) {
    const long r = ((long)chunk * NW + wave) * (OPW * 8) + o * 8 + qi;
    if (r < Nq) {
      const float inv = (float)(cnt[o] > 0 ? cnt[o] : 1u);
      const float4 a = acc[o];
      *reinterpret_cast<float4*>(slots + ((long)b * Nq + qid[o]) * (M * D) + m * D + li * 4) =
          make_float4(a.x / inv, a.y / inv, a.z / inv, a.w / inv);
    }
  }
  if (stats) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      n_in += __shfl_xor(n_in, d);
      n_rows += __shfl_xor(n_rows, d);
    }
    if (lane == 0) {
      if (n_rows) atomicAdd(&stats[0], (unsigned long long)n_rows);
      if (n_in) atomicAdd(&stats[1], (unsigned long long)n_in);
    }
  }
}

// ---- opt-in: fp16 value maps (SURVEY.md §8d's e_v = 2 variant) ---------------------------------------------------
// The exact kernels above are bound by the bytes they pull through the texture-addresser / L1 path (64 B/clk/CU);
// with the projected value stored as fp16 a bilinear corner of a head is one 64-byte row: 4 lanes x 16 bytes.  The
// 8 lanes of a query group split into two half-groups that take samples 0-3 / 4-7 of a chunk at the same time —
// 16 row loads per lane and chunk instead of 32 — each lane accumulating 8 channels in fp32 (v_fma_mix); the two
// halves are added with one shuffle at the end.  Sampling arithmetic (locations, weights, softmax) stays fp32; the
// only change to the result is the rounding of the value elements to 11 significant bits (measured end to end in
// tests/test_gpu_modules.py::test_sca_fp16_values_*), so this path is NOT the default.
template <int L, int P>
__global__ __launch_bounds__(256, 4) void sca_head_h_kernel(
    const __half* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lstart,
    const float* __restrict__ offs, long offs_stride, const float* __restrict__ logits, long logits_stride,
    const float* __restrict__ ref_cam, const uint32_t* __restrict__ vis_bits, const int32_t* __restrict__ order,
    float* __restrict__ slots, unsigned long long* __restrict__ stats, int B, int NC, int S, int Z, int Nq) {
  constexpr int M = 8, D = 32, LP = L * P, NCH = LP / 8, NW = 4;
  static_assert(P == 4 || P == 8, "a chunk of 8 samples spans at most two levels");
  constexpr unsigned ROW_B = M * D * 2;
  constexpr int GRP_B = 9 * 32, PAR_B = 8 * GRP_B;
  __shared__ __attribute__((aligned(16))) char par_all[NW * PAR_B];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  char* par = par_all + wave * PAR_B;
  const int qi = lane >> 3, li = lane & 7, hg = li >> 2, c8 = li & 3;
  const int m = blockIdx.x & 7, chunk = blockIdx.x >> 3, b = blockIdx.y;
  int Hs[L], Ws[L], Ss[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    Hs[l] = (int)shapes[2 * l]; Ws[l] = (int)shapes[2 * l + 1]; Ss[l] = (int)lstart[l];
  }
  const long r = ((long)chunk * NW + wave) * 8 + qi;
  const bool ok = r < Nq;
  const int q = ok ? (order ? order[r] : (int)r) : 0;
  const uint32_t vq = ok ? vis_bits[q] : 0u;
  const uint32_t cnt = ok ? (uint32_t)__builtin_popcount(vis_bits[(long)b * Nq + q]) : 0u;
  uint32_t cams = vq;
#pragma unroll
  for (int d = 32; d >= 8; d >>= 1) cams |= __shfl_xor(cams, d);
  cams = __builtin_amdgcn_readfirstlane(cams);
  const float* lrow = logits + ((long)b * Nq + q) * logits_stride + m * LP;
  const float* orow = offs + ((long)b * Nq + q) * offs_stride + (long)m * LP * 2;
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < NCH; ++j) mx = fmaxf(mx, lrow[8 * j + li]);
  mx = fmaxf(mx, __shfl_xor(mx, 1)); mx = fmaxf(mx, __shfl_xor(mx, 2)); mx = fmaxf(mx, __shfl_xor(mx, 4));
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NCH; ++j) sum += expf(lrow[8 * j + li] - mx);
  sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);

  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  unsigned n_in = 0, n_rows = 0;
  const char* gp = par + qi * GRP_B + hg * 4 * 32;        // this half-group's four samples of a chunk
  for (int c = 0; c < NC; ++c) {
    if (!((cams >> c) & 1u)) continue;                    // wave-uniform
    const bool live = (vq >> c) & 1u;
    const __half* vmap = value + ((long)b * NC + c) * S * (M * D) + m * D;
    const __amdgpu_buffer_rsrc_t rsrc = uniform_rsrc(vmap, (unsigned)S * ROW_B - (unsigned)(m * D * 2));
    const float* rp = ref_cam + (((long)c * B + b) * Nq + q) * Z * 2;
    if (m == 0 && li == 0 && live) ++n_rows;
#pragma unroll 1
    for (int j = 0; j < NCH; ++j) {
      const int s = 8 * j + li;
      const int l = s / P;
      int H = Hs[0], W = Ws[0], st = Ss[0];
#pragma unroll
      for (int t = 1; t < L; ++t)
        if (l == t) { H = Hs[t]; W = Ws[t]; st = Ss[t]; }
      const int z = (s % P) % Z;
      const float2 of = *reinterpret_cast<const float2*>(orow + 2 * s);
      const float2 rxy = *reinterpret_cast<const float2*>(rp + 2 * z);
      const float aw = expf(lrow[s] - mx) / sum;
      SampleParamB sp;
      n_in += bilinear_setup_b(rxy.x + of.x / (float)W, rxy.y + of.y / (float)H, aw, H, W, st, ROW_B, kOobOffset,
                               live, sp);
      *reinterpret_cast<SampleParamB*>(par + qi * GRP_B + li * 32) = sp;
      wave_lds_sync();
      // (bit_cast, not an assignment: the builtin returns a GCC vector, and hipcc converts that to an ext_vector
      // by splatting element 0 — the loads shrink to one dword each, silently)
      typedef _Float16 occ_h8 __attribute__((ext_vector_type(8)));
      occ_h8 v[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const occ_u32x4 o4 = *reinterpret_cast<const occ_u32x4*>(gp + u * 32 + 16);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          v[u][k] = __builtin_bit_cast(
              occ_h8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(o4[k] + (unsigned)c8 * 16u), 0, 0));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 w4 = *reinterpret_cast<const float4*>(gp + u * 32);
        const float ww[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] = fmaf(ww[k], (float)v[u][k][e], acc[e]);
      }
      wave_lds_sync();
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] += __shfl_xor(acc[i], 4);     // the two half-groups' samples
  if (ok && hg == 0) {
    const float inv = (float)(cnt > 0 ? cnt : 1u);
    float* dst = slots + ((long)b * Nq + q) * (M * D) + m * D + c8 * 8;
    *reinterpret_cast<float4*>(dst) = make_float4(acc[0] / inv, acc[1] / inv, acc[2] / inv, acc[3] / inv);
    *reinterpret_cast<float4*>(dst + 4) = make_float4(acc[4] / inv, acc[5] / inv, acc[6] / inv, acc[7] / inv);
  }
  if (stats) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      n_in += __shfl_xor(n_in, d);
      n_rows += __shfl_xor(n_rows, d);
    }
    if (lane == 0) {
      if (n_rows) atomicAdd(&stats[0], (unsigned long long)n_rows);
      if (n_in) atomicAdd(&stats[1], (unsigned long long)n_in);
    }
  }
}

template <int L, int P, int NW, int OPW, bool STAGE>
static int launch_sca_head(const float* value, const int64_t* shapes, const int64_t* lstart, const float* offs,
                           long offs_stride, const float* logits, long logits_stride, const float* ref_cam,
                           const uint32_t* vis_bits, const int32_t* order, float* slots, uint64_t* stats, int B,
                           int NC, int S, int Z, int Nq, int stage_pix, hipStream_t st) {
  constexpr int QPB = NW * OPW * 8;                                // queries per block
  const int nchunks = (Nq + QPB - 1) / QPB;
  const size_t lds = (STAGE ? (size_t)(stage_pix + 1) * 128 : 0) + (size_t)NW * 8 * 9 * 32 + 16;
  auto kern = sca_head_kernel<L, P, NW, OPW, STAGE>;
  if (lds > 48 * 1024) {
    static bool attr_done = false;                                 // per instantiation
    if (!attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024);
      attr_done = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nchunks * 8, (unsigned)B), dim3(NW * 64), lds, st, value, shapes, lstart,
                     offs, offs_stride, logits, logits_stride, ref_cam, vis_bits, order, slots,
                     reinterpret_cast<unsigned long long*>(stats), B, NC, S, Z, Nq, stage_pix);
  OCC_CHECK_LAUNCH("sca_head_forward");
  return OCC_OK;
}

}  // namespace occ

// variant: 1 = head-major, everything through buffer loads (4 waves x 1 octet);
//          2 = head-major + coarsest level staged in LDS (8 waves x 2 octets; needs stage_pix*128 <= 64 KB);
//          3 = as 2 with 6 waves x 2 octets (3 waves per SIMD)
extern "C" int occ_sca_head_forward_f32(const float* value, const int64_t* spatial_shapes,
                                        const int64_t* level_start_index, const float* offs,
                                        int64_t offs_stride, const float* logits, int64_t logits_stride,
                                        const float* ref_cam, const uint32_t* vis_bits, const int32_t* order,
                                        float* slots, uint64_t* stats, int B, int NC, int S, int M, int D, int L,
                                        int P, int Z, int Nq, int stage_pix, int variant, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(value && spatial_shapes && level_start_index && offs && logits && ref_cam && vis_bits && slots,
                "sca_head_forward: null pointer argument");
  OCC_CHECK_ARG(B > 0 && B < 65536 && NC > 0 && NC <= 32 && S > 0 && Nq > 0 && Z > 0 && L > 0 && P > 0,
                "sca_head_forward: bad dimension (B=%d NC=%d S=%d Nq=%d Z=%d L=%d P=%d)", B, NC, S, Nq, Z, L, P);
  OCC_CHECK_ARG(P % Z == 0, "sca_head_forward: num_points(%d) must be a multiple of Z(%d)", P, Z);
  OCC_CHECK_ARG(offs_stride >= (int64_t)M * L * P * 2 && logits_stride >= (int64_t)M * L * P,
                "sca_head_forward: row strides smaller than a row");
  OCC_CHECK_ARG((long)S * M * D * 4 < (long)kOobOffset, "sca_head_forward: value batch entry too large");
  OCC_CHECK_ARG(variant >= 1 && variant <= 3, "sca_head_forward: unknown variant %d", variant);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (M != 8 || D != 32) {
    set_error("sca_head_forward: no fused kernel for M=%d D=%d", M, D);
    return OCC_E_UNSUPPORTED;
  }
  if (variant >= 2 && (stage_pix <= 0 || (long)stage_pix * 128 > 64 * 1024)) variant = 1;   // nothing to stage
#define OCC_SCAH_ARGS value, spatial_shapes, level_start_index, offs, (long)offs_stride, logits, (long)logits_stride, \
                      ref_cam, vis_bits, order, slots, stats, B, NC, S, Z, Nq
  // the staged kernels are instantiated for the base configuration's (4 levels, 8 points) only: with fewer chunks
  // per query LLVM unrolls the chunk loop again and spills; other shapes take variant 1
  if (L == 4 && P == 8) {
    if (variant == 2) return launch_sca_head<4, 8, 8, 2, true>(OCC_SCAH_ARGS, stage_pix, st);
    if (variant == 3) return launch_sca_head<4, 8, 6, 2, true>(OCC_SCAH_ARGS, stage_pix, st);
    return launch_sca_head<4, 8, 4, 1, false>(OCC_SCAH_ARGS, 0, st);
  }
  if (L == 4 && P == 4) return launch_sca_head<4, 4, 4, 1, false>(OCC_SCAH_ARGS, 0, st);
  if (L == 2 && P == 8) return launch_sca_head<2, 8, 4, 1, false>(OCC_SCAH_ARGS, 0, st);
  if (L == 1 && P == 8) return launch_sca_head<1, 8, 4, 1, false>(OCC_SCAH_ARGS, 0, st);
#undef OCC_SCAH_ARGS
  set_error("sca_head_forward: no fused kernel for L=%d P=%d", L, P);
  return OCC_E_UNSUPPORTED;
}

// fp16 value maps (opt-in, see sca_head_h_kernel): value (B*NC, S, M, D) __half; everything else as above.
extern "C" int occ_sca_head_forward_f16v(const void* value_f16, const int64_t* spatial_shapes,
                                         const int64_t* level_start_index, const float* offs, int64_t offs_stride,
                                         const float* logits, int64_t logits_stride, const float* ref_cam,
                                         const uint32_t* vis_bits, const int32_t* order, float* slots,
                                         uint64_t* stats, int B, int NC, int S, int M, int D, int L, int P, int Z,
                                         int Nq, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(value_f16 && spatial_shapes && level_start_index && offs && logits && ref_cam && vis_bits && slots,
                "sca_head_forward_f16v: null pointer argument");
  OCC_CHECK_ARG(B > 0 && B < 65536 && NC > 0 && NC <= 32 && S > 0 && Nq > 0 && Z > 0 && L > 0 && P > 0,
                "sca_head_forward_f16v: bad dimension");
  OCC_CHECK_ARG(P % Z == 0, "sca_head_forward_f16v: num_points(%d) must be a multiple of Z(%d)", P, Z);
  OCC_CHECK_ARG(offs_stride >= (int64_t)M * L * P * 2 && logits_stride >= (int64_t)M * L * P,
                "sca_head_forward_f16v: row strides smaller than a row");
  OCC_CHECK_ARG((long)S * M * D * 2 < (long)kOobOffset, "sca_head_forward_f16v: value batch entry too large");
  if (M != 8 || D != 32) {
    set_error("sca_head_forward_f16v: no fused kernel for M=%d D=%d", M, D);
    return OCC_E_UNSUPPORTED;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)((Nq + 31) / 32) * 8, (unsigned)B);
#define OCC_SCAHH(LL, PP)                                                                                       \
  if (L == LL && P == PP) {                                                                                     \
    hipLaunchKernelGGL((sca_head_h_kernel<LL, PP>), grid, dim3(256), 0, st,                                     \
                       reinterpret_cast<const __half*>(value_f16), spatial_shapes, level_start_index, offs,     \
                       (long)offs_stride, logits, (long)logits_stride, ref_cam, vis_bits, order, slots,         \
                       reinterpret_cast<unsigned long long*>(stats), B, NC, S, Z, Nq);                          \
    OCC_CHECK_LAUNCH("sca_head_forward_f16v");                                                                  \
    return OCC_OK;                                                                                              \
  }
  OCC_SCAHH(4, 8)
  OCC_SCAHH(4, 4)
  OCC_SCAHH(2, 8)
  OCC_SCAHH(1, 8)
#undef OCC_SCAHH
  set_error("sca_head_forward_f16v: no fused kernel for L=%d P=%d", L, P);
  return OCC_E_UNSUPPORTED;
}
