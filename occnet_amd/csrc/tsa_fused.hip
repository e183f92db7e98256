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
// (Round 6 measured what 16-bit value rows would buy here — a 4-lanes-per-row variant of this kernel on garbage data, timing only:
// 53.4 -> 41.3 us per launch with the q16 decode, profiles/r06_c12_tsa_16bit_rows_timing.txt — against the q16 encoder it would need
// in the chain kernels' tail epilogues (program B sits at 256 VGPRs): not built.)
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
                   (unsigned)row_stride * 4u, kOobOffset, 1, p);
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
  const long waves = (long)B * Nq;
  const long blocks = (waves + kTsaWaves - 1) / kTsaWaves;
  hipLaunchKernelGGL(tsa_fused_kernel, dim3((unsigned)blocks), dim3(256), 0, st, value,
                     (long)value_bt_stride, offs, (long)offs_stride, logits, (long)logits_stride,
                     ref_2d, order, out, B, Nq, bev_h, bev_w);
  OCC_CHECK_LAUNCH("tsa_fused_forward");
  return OCC_OK;
}
