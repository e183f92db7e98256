// Fused temporal self-attention gather for gfx950 (single BEV level, 2-deep queue).
//
// Replaces (reference: projects/mmdet3d_plugin/bevformer/modules/temporal_self_attention.py):
//   :206-222  view/softmax of the offsets/weights Linear outputs and the (bs*2) permutes
//   :224-228  sampling_locations = ref_2d + offsets / (W, H)
//   :240-253  ms_deform_attn_forward on (bs*2, Nq) rows
//   :255-262  permute + mean over the two queue entries
// One 64-lane wave owns one BEV query: lane = m*8 + t*4 + p resolves exactly one sample
// (head m, queue entry t, point p); softmax is a 4-lane shuffle; every 8-lane group then gathers
// its head's 8 samples (32 x 16-byte loads per lane) from the two queue entries' value maps and
// writes the mean.  When there is no history BEV the reference stacks the current BEV twice:
// pass value_bt_stride = 0 and the two entries alias one projected buffer.
#include <cstdlib>
#include "common.h"

namespace occ {

constexpr int kTsaWaves = 4;

__global__ __launch_bounds__(256) void tsa_fused_kernel(
    const float* __restrict__ value, long value_bt_stride, const float* __restrict__ offs,
    long offs_stride, const float* __restrict__ logits, long logits_stride,
    const float* __restrict__ ref_2d, const int32_t* __restrict__ order, float* __restrict__ out,
    int B, int Nq, int bev_h, int bev_w) {
  constexpr int M = 8, D = 32, P = 4, NS = 2 * P;  // samples per head
  constexpr int NSp = NS + 1;
  __shared__ __attribute__((aligned(16))) SampleParamB smem[kTsaWaves * M * NSp];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const long wg = (long)blockIdx.x * kTsaWaves + wave;
  if (wg >= (long)B * Nq) return;
  const int b = (int)(wg / Nq);
  const int r = (int)(wg - (long)b * Nq);
  const int q = order ? order[r] : r;
  SampleParamB* sp = smem + wave * M * NSp;
  constexpr int row_stride = M * D;

  // lane = m*8 + t*4 + p : exactly the memory order of both Linear outputs
  const int m = lane >> 3, t = (lane >> 2) & 1;
  const float x = logits[((long)b * Nq + q) * logits_stride + lane];
  float mx = fmaxf(x, __shfl_xor(x, 1));
  mx = fmaxf(mx, __shfl_xor(mx, 2));
  const float e = expf(x - mx);
  float sum = e + __shfl_xor(e, 1);
  sum += __shfl_xor(sum, 2);
  const float aw = fdiv(e, sum);                 // (fdiv, not `/`: see common.h)
  float2 o = *reinterpret_cast<const float2*>(offs + ((long)b * Nq + q) * offs_stride + 2 * lane);
  o.x = fdiv(o.x, (float)bev_w);
  o.y = fdiv(o.y, (float)bev_h);
  const float2 rf = *reinterpret_cast<const float2*>(ref_2d + (((long)b * 2 + t) * Nq + q) * 2);
  // corners outside the BEV map carry an out-of-range byte offset: the buffer load returns 0 without a request (no
  // dummy load of row 0, no 0 * Inf)
  SampleParamB p;
  bilinear_setup_b(rf.x + o.x, rf.y + o.y, aw, bev_h, bev_w, 0,
                   (unsigned)row_stride * 4u, kOobOffset, true, p);
  sp[m * NSp + (lane & 7)] = p;
  wave_lds_sync();

  const int g = lane >> 3, c4 = lane & 7;
  const unsigned map_bytes = (unsigned)bev_h * (unsigned)bev_w * (unsigned)row_stride * 4u;
  const __amdgpu_buffer_rsrc_t r0 = uniform_rsrc(value + ((long)b * 2 + 0) * value_bt_stride, map_bytes);
  const __amdgpu_buffer_rsrc_t r1 = uniform_rsrc(value + ((long)b * 2 + 1) * value_bt_stride, map_bytes);
  const unsigned lane_off = (unsigned)(g * D + c4 * 4) * 4u;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
  a0 = gather_samples_buf<4>(r0, lane_off, sp + g * NSp, P, a0);
  a1 = gather_samples_buf<4>(r1, lane_off, sp + g * NSp + P, P, a1);
  float4 o4 = make_float4((a0.x + a1.x) * 0.5f, (a0.y + a1.y) * 0.5f, (a0.z + a1.z) * 0.5f,
                          (a0.w + a1.w) * 0.5f);
  *reinterpret_cast<float4*>(out + ((long)b * Nq + q) * row_stride + g * D + c4 * 4) = o4;
}


// ---- tile kernel (round 5, OPT-IN: OCC_TSA_TILE=1): the value rows staged through LDS ----------------------------------------
// The TSA gather is LOCAL: a query samples the BEV map a few pixels around its own position (the reference initialises the
// offsets to k * (cos, sin) steps, k = 1..4 pixels, temporal_self_attention.py:117-133), and 64 samples x 4 corners per query
// re-read every 128-byte head row of the neighbourhood ~32 times.  The wave-per-query kernel above fetches each of those 256
// rows per query through the texture path (1.3 GB per launch at 200 x 200: TA busy 0.84, 9 % of the wave cycles executing,
// profiles/r04_final4_pmc_derived.txt).  Here a block owns an 8 x 8 tile of queries and, head by head, stages the head's rows
// of the tile's window — the tile plus kTsaHalo pixels around it (+ 1 for the right / lower bilinear corner) — into LDS
// with dense 16-byte loads (41 KB per head instead of 256 KB of gathered rows), then gathers from LDS: lane = (query j of
// 8, 16-byte piece c of 8), a query's 8 samples of the head accumulated in registers, no cross-lane reduction.  Corners
// outside the MAP read a zero row (the reference's zero padding: never 0 * Inf); a corner inside the map but outside the
// window (a learned offset beyond the halo) is fetched from global memory by the same lanes, so the result never depends
// on the window size.  Same sampling arithmetic as above (bilinear_setup_b); the two queue entries are staged one after the
// other when they are different maps (history BEV).
constexpr int kTsaTile = 8, kTsaHalo = 5, kTsaWin = kTsaTile + 2 * kTsaHalo;          // 18 x 18 pixels
constexpr int kTsaRowB = 128, kTsaWinBytes = kTsaWin * kTsaWin * kTsaRowB;              // 41 472 B per head
constexpr int kTsaSlab = 8 * 9;                                                        // 8 queries x (8 samples + 1 pad)
constexpr unsigned kTsaGlobalBit = 0x80000000u;                                        // o[k]: global byte offset, not LDS

__global__ __launch_bounds__(256, 3) void tsa_tile_kernel(
    const float* __restrict__ value, long value_bt_stride, const float* __restrict__ offs, long offs_stride,
    const float* __restrict__ logits, long logits_stride, const float* __restrict__ ref_2d, float* __restrict__ out,
    int Nq, int bev_h, int bev_w, int tiles_x, int tiles_per_map, int tiles_per_xcd) {
  constexpr int M = 8, D = 32, P = 4, row_stride = M * D;
  __shared__ __attribute__((aligned(16))) char win[kTsaWinBytes + kTsaRowB];            // + the zero row
  __shared__ __attribute__((aligned(16))) SampleParamB slab[kTsaWaves * kTsaSlab];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // blockIdx -> (batch, tile): consecutive blocks go to consecutive XCDs; each XCD gets a contiguous strip of tiles, so
  // that neighbouring tiles (which share halo rows) share an L2
  const int bid = (int)blockIdx.x;
  const int b = bid / (8 * tiles_per_xcd);
  const int rb = bid - b * 8 * tiles_per_xcd;
  const int tile = (rb & 7) * tiles_per_xcd + (rb >> 3);
  if (tile >= tiles_per_map) return;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int wy0 = ty * kTsaTile - kTsaHalo, wx0 = tx * kTsaTile - kTsaHalo;
  if (tid < kTsaRowB / 4) reinterpret_cast<float*>(win + kTsaWinBytes)[tid] = 0.f;
  const bool shared = value_bt_stride == 0;
  const int T = shared ? 1 : 2;                     // window stages per head: one per DIFFERENT queue entry
  const unsigned map_bytes = (unsigned)bev_h * (unsigned)bev_w * (unsigned)row_stride * 4u;

  // the wave's queries: tile rows 2 * wave + pass, one query per 8-lane group; as set-up lane = (query j, sample s = t' * 4 + p)
  const int j = lane >> 3, c = lane & 7, s_mine = lane & 7;
  const int qx = tx * kTsaTile + j;
  SampleParamB* sp = slab + wave * kTsaSlab;
  bool live[2];
  long qrow[2];
  float2 rf[2];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int qy = ty * kTsaTile + 2 * wave + pass;
    live[pass] = qy < bev_h && qx < bev_w;
    qrow[pass] = (long)b * Nq + (live[pass] ? qy * bev_w + qx : 0);
    rf[pass] = live[pass] ? *reinterpret_cast<const float2*>(ref_2d + (((long)b * 2 + (s_mine >> 2)) * Nq + (qrow[pass] - (long)b * Nq)) * 2)
                          : make_float2(0.f, 0.f);
  }

  // Software pipeline: the window rows of stage st + 1 and the query-side operands of head h + 1 are REQUESTED before the
  // gather of stage st and consumed after it — the first cut issued them where it needed them and spent nine tenths of a
  // block's time waiting for memory (stage, barrier, set-up loads, gather: 99 us per launch against the wave-per-query
  // kernel's 68, profiles/r05_c3_hot_kernel_trace_stats.txt).
  constexpr int kPieces = kTsaWin * kTsaWin * 8, kIter = (kPieces + 255) / 256;
  float4 wv[kIter];
  float lg[2];
  float2 of[2];
#define OCC_TSA_WINDOW(ST)                                                                                          \
  {                                                                                                                  \
    const int h_ = (ST) / T, t_ = (ST) - h_ * T;                                                                     \
    const __amdgpu_buffer_rsrc_t rs_ = uniform_rsrc(value + ((long)b * 2 + t_) * value_bt_stride, map_bytes);        \
    _Pragma("unroll") for (int it = 0; it < kIter; ++it) {                                                           \
      const int i = tid + it * 256;                                                                                  \
      const int pix = i >> 3, wy = pix / kTsaWin, wx = pix - wy * kTsaWin;                                           \
      const int y = wy0 + wy, x = wx0 + wx;                                                                          \
      const bool in = i < kPieces && (unsigned)y < (unsigned)bev_h && (unsigned)x < (unsigned)bev_w;                 \
      const unsigned goff = in ? (unsigned)(y * bev_w + x) * (unsigned)(row_stride * 4) +                            \
                                     (unsigned)(h_ * kTsaRowB + (i & 7) * 16)                                        \
                               : kOobOffset;                                                                         \
      wv[it] = buf_load16(rs_, goff);                                                                                \
    }                                                                                                                \
  }
#define OCC_TSA_PARAMS(H)                                                                                           \
  _Pragma("unroll") for (int pass = 0; pass < 2; ++pass) {                                                           \
    lg[pass] = live[pass] ? logits[qrow[pass] * logits_stride + (H) * 8 + s_mine] : 0.f;                             \
    of[pass] = live[pass] ? *reinterpret_cast<const float2*>(offs + qrow[pass] * offs_stride + 2 * ((H) * 8 + s_mine)) \
                          : make_float2(0.f, 0.f);                                                                   \
  }
  OCC_TSA_WINDOW(0)
  OCC_TSA_PARAMS(0)
  float4 acc[2];
  float lgc[2];
  float2 ofc[2];
  for (int st = 0; st < M * T; ++st) {
    const int h = st / T, t = st - h * T;
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
      const int i = tid + it * 256;
      if (i < kPieces) *reinterpret_cast<float4*>(win + i * 16) = wv[it];
    }
    block_lds_sync();
    if (t == 0) {
      acc[0] = acc[1] = make_float4(0.f, 0.f, 0.f, 0.f);
      lgc[0] = lg[0]; lgc[1] = lg[1]; ofc[0] = of[0]; ofc[1] = of[1];
      if (h + 1 < M) OCC_TSA_PARAMS(h + 1)
    }
    if (st + 1 < M * T) OCC_TSA_WINDOW(st + 1)
    const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(value + ((long)b * 2 + t) * value_bt_stride, map_bytes);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      bool far;
      {
        const float xl = lgc[pass];
        float mx = fmaxf(xl, __shfl_xor(xl, 1));
        mx = fmaxf(mx, __shfl_xor(mx, 2));
        const float e = expf(xl - mx);
        float sum = e + __shfl_xor(e, 1);
        sum += __shfl_xor(sum, 2);
        const float aw = fdiv(e, sum);
        SampleParamB p;
        // offsets in PIXELS first (pix_bytes = 1), then window-relative LDS bytes or a flagged global byte offset
        bilinear_setup_b(rf[pass].x + fdiv(ofc[pass].x, (float)bev_w), rf[pass].y + fdiv(ofc[pass].y, (float)bev_h), aw, bev_h, bev_w,
                         0, 1u, 0xffffffffu, live[pass], p);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (p.o[k] == 0xffffffffu) {
            p.o[k] = (unsigned)kTsaWinBytes;                         // the zero row
          } else {
            const int py = (int)p.o[k] / bev_w, px = (int)p.o[k] - py * bev_w;
            const int wy = py - wy0, wx = px - wx0;
            p.o[k] = ((unsigned)wy < (unsigned)kTsaWin && (unsigned)wx < (unsigned)kTsaWin)
                         ? (unsigned)(wy * kTsaWin + wx) * (unsigned)kTsaRowB
                         : (kTsaGlobalBit | (p.o[k] * (unsigned)(row_stride * 4) + (unsigned)(h * kTsaRowB)));
          }
        }
        sp[j * 9 + s_mine] = p;
        far = ((p.o[0] | p.o[1] | p.o[2] | p.o[3]) & kTsaGlobalBit) != 0;
      }
      wave_lds_sync();
      const int s_lo = shared ? 0 : t * P, s_hi = shared ? 2 * P : (t + 1) * P;
      const SampleParamB* mine = sp + j * 9;
      float4 a = acc[pass];
      // wave-uniform choice: the pure-LDS loop carries no vector-memory instruction, so nothing in it waits for the window
      // rows that are in flight for the next stage (with the fallback load in the same loop hipcc put `s_waitcnt vmcnt(0)`
      // behind every corner and the prefetch was drained before the first sample)
      if (__builtin_amdgcn_ballot_w64(far) == 0) {
        for (int s = s_lo; s < s_hi; ++s) {
          const occ_u32x4 o = *reinterpret_cast<const occ_u32x4*>(mine[s].o);
          const float4 w = *reinterpret_cast<const float4*>(mine[s].w);
          float4 r[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) r[k] = *reinterpret_cast<const float4*>(win + o[k] + c * 16);
          fma4(a, w.x, r[0]); fma4(a, w.y, r[1]); fma4(a, w.z, r[2]); fma4(a, w.w, r[3]);
        }
      } else {
        for (int s = s_lo; s < s_hi; ++s) {
          const occ_u32x4 o = *reinterpret_cast<const occ_u32x4*>(mine[s].o);
          const float4 w = *reinterpret_cast<const float4*>(mine[s].w);
          float4 r[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (o[k] & kTsaGlobalBit) r[k] = buf_load16(rs, (o[k] & ~kTsaGlobalBit) + (unsigned)(c * 16));   // beyond the halo
            else r[k] = *reinterpret_cast<const float4*>(win + o[k] + c * 16);
          }
          fma4(a, w.x, r[0]); fma4(a, w.y, r[1]); fma4(a, w.z, r[2]); fma4(a, w.w, r[3]);
        }
      }
      acc[pass] = a;
      wave_lds_sync();                                               // WAR: the next pass rewrites the slab
    }
    block_lds_sync();                                                // WAR: the next stage rewrites the window
    if (t == T - 1) {
#pragma unroll
      for (int pass = 0; pass < 2; ++pass)
        if (live[pass]) {
          const float4 a = acc[pass];
          *reinterpret_cast<float4*>(out + qrow[pass] * row_stride + h * D + c * 4) =
              make_float4(a.x * 0.5f, a.y * 0.5f, a.z * 0.5f, a.w * 0.5f);
        }
    }
  }
#undef OCC_TSA_WINDOW
#undef OCC_TSA_PARAMS
}

}  // namespace occ

extern "C" int occ_tsa_fused_forward_f32(const float* value, int64_t value_bt_stride,
                                         const float* offs, int64_t offs_stride,
                                         const float* logits, int64_t logits_stride,
                                         const float* ref_2d, const int32_t* order, float* out,
                                         int B, int Nq, int bev_h, int bev_w, int M, int D, int P,
                                         void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(value && offs && logits && ref_2d && out, "tsa_fused_forward: null pointer argument");
  OCC_CHECK_ARG(B > 0 && Nq > 0 && bev_h > 0 && bev_w > 0, "tsa_fused_forward: bad dimension");
  OCC_CHECK_ARG((long)bev_h * bev_w * M * D * 4 < (long)occ::kOobOffset, "tsa_fused_forward: BEV map too large");
  OCC_CHECK_ARG(value_bt_stride >= 0, "tsa_fused_forward: negative value stride");
  OCC_CHECK_ARG(offs_stride >= (int64_t)M * 2 * P * 2 && logits_stride >= (int64_t)M * 2 * P,
                "tsa_fused_forward: row strides smaller than a row");
  OCC_CHECK_ARG((long)bev_h * bev_w * M * D < (1L << 31), "tsa_fused_forward: value map too large");
  if (M != 8 || D != 32 || P != 4) {
    set_error("tsa_fused_forward: no fused kernel for M=%d D=%d P=%d", M, D, P);
    return OCC_E_UNSUPPORTED;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // OCC_TSA_TILE=1: the tile kernel (value rows staged through LDS) when the queries are the whole BEV map, query q at
  // pixel (q / bev_w, q % bev_w) — `order` is a locality hint of the wave-per-query kernel and plays no role there.
  // Measured (round 5, same box, hot-path step, both kernels alone on the stream): 59.5 us per launch against the
  // wave-per-query kernel's 53-54 us (profiles/r05_c11_hot_kernel_trace_stats.txt, r05_c11_tsa_ab.txt) — correct
  // (tests/test_gpu_msda.py), not faster: off by default.
  const char* tile_env = getenv("OCC_TSA_TILE");           // read per call: the tests switch it inside one process
  const bool tile_on = tile_env && tile_env[0] == '1';
  if (tile_on && (long)Nq == (long)bev_h * bev_w) {
    const int tiles_x = (bev_w + kTsaTile - 1) / kTsaTile, tiles_y = (bev_h + kTsaTile - 1) / kTsaTile;
    const int tiles = tiles_x * tiles_y, per_xcd = (tiles + 7) / 8;
    hipLaunchKernelGGL(tsa_tile_kernel, dim3((unsigned)((long)B * 8 * per_xcd)), dim3(256), 0, st, value,
                       (long)value_bt_stride, offs, (long)offs_stride, logits, (long)logits_stride, ref_2d, out, Nq,
                       bev_h, bev_w, tiles_x, tiles, per_xcd);
    OCC_CHECK_LAUNCH("tsa_fused_forward");
    return OCC_OK;
  }
  const long waves = (long)B * Nq;
  const long blocks = (waves + kTsaWaves - 1) / kTsaWaves;
  hipLaunchKernelGGL(tsa_fused_kernel, dim3((unsigned)blocks), dim3(256), 0, st, value,
                     (long)value_bt_stride, offs, (long)offs_stride, logits, (long)logits_stride,
                     ref_2d, order, out, B, Nq, bev_h, bev_w);
  OCC_CHECK_LAUNCH("tsa_fused_forward");
  return OCC_OK;
}
