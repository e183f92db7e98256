// Test infrastructure (tests/test_gpu_hazard_repro.py), not product code: the self-contained reproducer of the co-scheduling
// hazard of round 5 as a standing test.  A TSA-shaped gather (one wave per BEV query, 8 heads x 8 samples, the per-sample
// terms handed from the resolving lane to the gathering lanes through LDS) on a value map of ONES: softmax weights and
// bilinear weights each sum to one, so every interior query's output must be 1 whatever the offsets and logits.
//   SHIPPED = true : the sampling set-up is occ::bilinear_setup_b of occnet_amd/csrc/common.h, included as it ships (lane
//                    predicates as 0 / 1 VGPR integers), quotients by occ::fdiv — what the library's gathers compile;
//   SHIPPED = false: the set-up the library had through round 5 (plain `bool` conditions: hipcc keeps them as lane masks in
//                    SGPR pairs, v_cmp -> s_and_b64 / s_and_saveexec_b64) and IEEE `/` — the CONTROL: wrong weights in lanes
//                    48-63 in 149 of 150 repetitions next to the library's value projection on another stream (round 5).
// Counts the output words that are off by more than 1e-3, by 16-lane quarter of the wave.
#include "../../occnet_amd/csrc/common.h"

namespace {

__device__ __forceinline__ int legacy_bilinear_setup_b(float loc_x, float loc_y, float attn, int H, int W, int lvl_pix0,
                                                       unsigned pix_bytes, unsigned dead, bool live, occ::SampleParamB& sp) {
  sp.w[0] = sp.w[1] = sp.w[2] = sp.w[3] = 0.f;
  sp.o[0] = sp.o[1] = sp.o[2] = sp.o[3] = dead;
  const float h_im = loc_y * (float)H - 0.5f;
  const float w_im = loc_x * (float)W - 0.5f;
  int n_in = 0;
  if (live && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
    const float hf = floorf(h_im), wf = floorf(w_im);
    const int h_low = (int)hf, w_low = (int)wf;
    const int h_high = h_low + 1, w_high = w_low + 1;
    const float lh = h_im - hf, lw = w_im - wf;
    const float hh = 1.f - lh, hw = 1.f - lw;
    const bool t = h_low >= 0, b = h_high <= H - 1, l = w_low >= 0, r = w_high <= W - 1;
    const int base = lvl_pix0 + h_low * W + w_low;
    if (t && l) { sp.w[0] = hh * hw * attn; sp.o[0] = (unsigned)base * pix_bytes; ++n_in; }
    if (t && r) { sp.w[1] = hh * lw * attn; sp.o[1] = (unsigned)(base + 1) * pix_bytes; ++n_in; }
    if (b && l) { sp.w[2] = lh * hw * attn; sp.o[2] = (unsigned)(base + W) * pix_bytes; ++n_in; }
    if (b && r) { sp.w[3] = lh * lw * attn; sp.o[3] = (unsigned)(base + W + 1) * pix_bytes; ++n_in; }
  }
  return n_in;
}

template <bool SHIPPED>
__global__ __launch_bounds__(256) void tsa_victim(const float* __restrict__ value, const float* __restrict__ offs,
                                                  const float* __restrict__ logits, unsigned long long* errors, int bev_h,
                                                  int bev_w) {
  constexpr int M = 8, D = 32, P = 4, NS = 2 * P, NSp = NS + 1;
  __shared__ __attribute__((aligned(16))) occ::SampleParamB smem[4 * M * NSp];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Nq = bev_h * bev_w;
  const long q = (long)blockIdx.x * 4 + wave;
  if (q >= Nq) return;
  occ::SampleParamB* sp = smem + wave * M * NSp;
  constexpr int row_stride = M * D;
  const int m = lane >> 3;
  const float x = logits[q * 64 + lane];
  float mx = fmaxf(x, __shfl_xor(x, 1));
  mx = fmaxf(mx, __shfl_xor(mx, 2));
  const float e = expf(x - mx);
  float sum = e + __shfl_xor(e, 1);
  sum += __shfl_xor(sum, 2);
  const float aw = SHIPPED ? occ::fdiv(e, sum) : e / sum;
  const float2 o = *reinterpret_cast<const float2*>(offs + q * 128 + 2 * lane);
  const int qy = (int)(q / bev_w), qx = (int)(q - (long)qy * bev_w);
  const float2 rf = SHIPPED ? make_float2(occ::fdiv((float)qx + 0.5f, (float)bev_w), occ::fdiv((float)qy + 0.5f, (float)bev_h))
                            : make_float2(((float)qx + 0.5f) / (float)bev_w, ((float)qy + 0.5f) / (float)bev_h);
  const float ox = SHIPPED ? occ::fdiv(o.x, (float)bev_w) : o.x / (float)bev_w;
  const float oy = SHIPPED ? occ::fdiv(o.y, (float)bev_h) : o.y / (float)bev_h;
  occ::SampleParamB p;
  if (SHIPPED) occ::bilinear_setup_b(rf.x + ox, rf.y + oy, aw, bev_h, bev_w, 0, (unsigned)row_stride * 4u, occ::kOobOffset, 1, p);
  else legacy_bilinear_setup_b(rf.x + ox, rf.y + oy, aw, bev_h, bev_w, 0, (unsigned)row_stride * 4u, occ::kOobOffset, true, p);
  sp[m * NSp + (lane & 7)] = p;
  occ::wave_lds_sync();
  const int g = lane >> 3, c4 = lane & 7;
  const unsigned map_bytes = (unsigned)bev_h * (unsigned)bev_w * (unsigned)row_stride * 4u;
  const __amdgpu_buffer_rsrc_t r0 = occ::uniform_rsrc(value, map_bytes);
  const unsigned lane_off = (unsigned)(g * D + c4 * 4) * 4u;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
  a0 = occ::gather_samples_buf<4>(r0, lane_off, sp + g * NSp, P, a0);
  a1 = occ::gather_samples_buf<4>(r0, lane_off, sp + g * NSp + P, P, a1);
  const float4 o4 = make_float4((a0.x + a1.x) * 0.5f, (a0.y + a1.y) * 0.5f, (a0.z + a1.z) * 0.5f, (a0.w + a1.w) * 0.5f);
  const bool interior = qy >= 16 && qy < bev_h - 16 && qx >= 16 && qx < bev_w - 16;     // offsets stay inside the map
  unsigned bad = 0;
  if (interior)
    bad = (fabsf(o4.x - 1.f) > 1e-3f) + (fabsf(o4.y - 1.f) > 1e-3f) + (fabsf(o4.z - 1.f) > 1e-3f) + (fabsf(o4.w - 1.f) > 1e-3f);
  if (bad) {
    atomicAdd(errors, (unsigned long long)bad);
    atomicAdd(errors + 1 + (lane >> 4), 1ull);          // which 16-lane quarter of the wave
  }
}

}  // namespace

// errors: 5 x uint64 (device): [0] wrong words, [1..4] wrong lanes by 16-lane quarter.  Accumulates; the caller zeroes.
extern "C" int hz_tsa_victim(const float* value, const float* offs, const float* logits, unsigned long long* errors, int bev_h,
                             int bev_w, int shipped, void* stream) {
  const int blocks = (bev_h * bev_w + 3) / 4;
  if (shipped) hipLaunchKernelGGL(tsa_victim<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, value, offs, logits, errors, bev_h, bev_w);
  else hipLaunchKernelGGL(tsa_victim<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, value, offs, logits, errors, bev_h, bev_w);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
