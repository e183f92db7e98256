// Shared device helpers for the gfx950 deformable-gather kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/occnet_amd.h"

namespace occ {

// Error plumbing for the C ABI (thread-local message, negative return codes).
void set_error(const char* fmt, ...);
#define OCC_CHECK_ARG(cond, ...)              \
  do {                                        \
    if (!(cond)) {                            \
      ::occ::set_error(__VA_ARGS__);          \
      return OCC_E_INVALID;                   \
    }                                         \
  } while (0)
#define OCC_CHECK_LAUNCH(what)                                                  \
  do {                                                                          \
    hipError_t e__ = hipGetLastError();                                         \
    if (e__ != hipSuccess) {                                                    \
      ::occ::set_error("%s: launch failed: %s", what, hipGetErrorString(e__));  \
      return OCC_E_LAUNCH;                                                      \
    }                                                                           \
  } while (0)

// One bilinear sample, pre-resolved: four corner weights (already multiplied by the attention
// weight; 0 for corners outside the map) and four element offsets into the value tensor of one
// batch entry (0 for corners outside the map, so the load stays legal and contributes 0*v).
struct __attribute__((aligned(16))) SampleParam {
  float w[4];
  int o[4];
};

// A lane predicate as a 0 / 1 integer in a VGPR.  The sampling set-up of every gather / scatter kernel of this library keeps
// its conditions in this form: v_cmp -> v_cndmask at once, conditions combined by VALU integer ANDs, selects on those
// integers.  hipcc compiles the plain `bool` form into lane masks in SGPR pairs combined on the scalar unit (v_cmp ->
// s_and_b64 / s_and_saveexec_b64), and a self-contained copy of the TSA gather with THAT set-up returns wrong weights in
// lanes 48-63 in 149 of 150 runs while a wave of an MFMA kernel is resident on another hardware queue; with this form
// 0 of 150, outputs bit-identical (tools_dev/lab/hazard/README.md; standing test tests/test_gpu_hazard_repro.py).  The empty
// asm keeps the compiler from folding the integer back into a scalar lane mask.
__device__ __forceinline__ int lane_flag(bool c) {
  int f = c ? 1 : 0;
  asm volatile("" : "+v"(f));
  return f;
}

// Arithmetic of mmcv's ms_deformable_im2col:
//   h_im = loc_y*H - 0.5, w_im = loc_x*W - 0.5, admitted iff -1 < h_im < H and -1 < w_im < W,
//   corners (h_low,w_low) (h_low,w_high) (h_high,w_low) (h_high,w_high) each read only if inside.
// c[k] = 1 iff the sample is admitted (and `live`) and corner k lies inside the map; lh / lw / hh / hw are the fractional
// weights of an admitted sample and garbage (possibly NaN) otherwise: always select on adm / c[k], never multiply by them.
struct BilinearTerms {
  float lh, lw, hh, hw;
  int h_low, w_low;
  int adm;
  int c[4];
};
__device__ __forceinline__ BilinearTerms bilinear_terms(float loc_x, float loc_y, int H, int W, int live) {
  BilinearTerms t;
  const float h_im = loc_y * (float)H - 0.5f;
  const float w_im = loc_x * (float)W - 0.5f;
  t.adm = live & lane_flag(h_im > -1.f) & lane_flag(w_im > -1.f) & lane_flag(h_im < (float)H) & lane_flag(w_im < (float)W);
  const float hf = floorf(h_im), wf = floorf(w_im);
  t.h_low = (int)hf;
  t.w_low = (int)wf;
  t.lh = h_im - hf;
  t.lw = w_im - wf;
  t.hh = 1.f - t.lh;
  t.hw = 1.f - t.lw;
  const int top = lane_flag(t.h_low >= 0) & t.adm, bot = lane_flag(t.h_low + 1 <= H - 1) & t.adm;
  const int lft = lane_flag(t.w_low >= 0), rgt = lane_flag(t.w_low + 1 <= W - 1);
  t.c[0] = top & lft; t.c[1] = top & rgt; t.c[2] = bot & lft; t.c[3] = bot & rgt;
  return t;
}

// One bilinear sample resolved for the gather: weights already multiplied by the attention weight, element offsets of the
// corner rows (row_stride = floats between two consecutive keys = M*D); 0 / 0 for corners outside the map, so the load
// stays legal and contributes 0*v.  Returns #corners inside the map.
__device__ __forceinline__ int bilinear_setup(float loc_x, float loc_y, float attn, int H, int W,
                                              int lvl_start, int row_stride, int live, SampleParam& sp) {
  const BilinearTerms t = bilinear_terms(loc_x, loc_y, H, W, live);
  const int base = lvl_start + t.h_low * W + t.w_low;
  sp.w[0] = t.c[0] ? t.hh * t.hw * attn : 0.f; sp.o[0] = t.c[0] ? base * row_stride : 0;
  sp.w[1] = t.c[1] ? t.hh * t.lw * attn : 0.f; sp.o[1] = t.c[1] ? (base + 1) * row_stride : 0;
  sp.w[2] = t.c[2] ? t.lh * t.hw * attn : 0.f; sp.o[2] = t.c[2] ? (base + W) * row_stride : 0;
  sp.w[3] = t.c[3] ? t.lh * t.lw * attn : 0.f; sp.o[3] = t.c[3] ? (base + W + 1) * row_stride : 0;
  return t.c[0] + t.c[1] + t.c[2] + t.c[3];
}

// f32 -> bf16, round to nearest even, two values per instruction: gfx950's v_cvt_pk_bf16_f32 (one VALU op
// instead of ~7 per value for the integer restatement of RNE; the epilogues of the bf16 kernels are
// VALU-issue-bound otherwise).  Result: lo in bits 0..15, hi in bits 16..31.
__device__ __forceinline__ unsigned pack_bf16x2_rne(float lo, float hi) {
  typedef float occ_f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 occ_bf16x2 __attribute__((ext_vector_type(2)));
  const occ_f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, occ_bf16x2));
}
__device__ __forceinline__ unsigned short bf16_rne(float f) {
  return (unsigned short)(pack_bf16x2_rne(f, 0.f) & 0xffffu);
}

// fp32 quotient WITHOUT the IEEE division expansion.  hipcc turns `a / d` into v_div_scale_f32 (x2, one writes VCC) /
// v_rcp_f32 / v_fma chain / v_div_fmas_f32 (reads VCC implicitly) / v_div_fixup_f32: 12 instructions and a VCC round trip
// where 5 do.  Reciprocal + one Newton step + one residual correction: faithfully rounded (<= 1 ulp; equal to the correctly
// rounded quotient in all 4 194 304 cases of tests/test_gpu_value_range.py) for normal, non-zero d with |d| < 2^126 (beyond
// it v_rcp_f32 flushes to zero) — every divisor in the gather kernels is a positive map size, a softmax sum >= 1 or a
// camera count times a range scale <= 2^100.  A non-finite numerator (an overflowed accumulator) or an overflowing quotient
// gives a * r — Inf / NaN like the IEEE quotient — instead of the NaN the residual step would make of Inf - Inf.
// (History: tools_dev/lab/hazard/README.md — with `/` the gathers of round 5 met the co-scheduling hazard more often.)
__device__ __forceinline__ float fdiv(float a, float d) {
  float r = __builtin_amdgcn_rcpf(d);
  r = fmaf(fmaf(-d, r, 1.f), r, r);
  const float q = a * r;
  const float c = fmaf(fmaf(-q, d, a), r, q);
  return fabsf(q) < __builtin_huge_valf() ? c : q;
}

// Orders this wave's LDS writes before its later LDS reads (cross-lane hand-off inside ONE wave:
// the hardware executes a wave's DS instructions in order, the fences only pin the compiler).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// acc += sum over NS samples of sum_k w[k] * value[o[k] .. o[k]+3]; vb already points at this
// lane's 4 channels (batch base + head*D + 4*(lane&7)).  UNROLL samples (= 4*UNROLL independent
// 16-byte loads) are kept in flight per lane.
template <int UNROLL>
__device__ __forceinline__ float4 gather_samples(const float* __restrict__ vb,
                                                 const SampleParam* sp, int ns, float4 acc) {
  int s = 0;
  for (; s + UNROLL <= ns; s += UNROLL) {
    float4 w[UNROLL];
    int4 o[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      w[u] = *reinterpret_cast<const float4*>(sp[s + u].w);
      o[u] = *reinterpret_cast<const int4*>(sp[s + u].o);
    }
    float4 v[UNROLL][4];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      v[u][0] = *reinterpret_cast<const float4*>(vb + o[u].x);
      v[u][1] = *reinterpret_cast<const float4*>(vb + o[u].y);
      v[u][2] = *reinterpret_cast<const float4*>(vb + o[u].z);
      v[u][3] = *reinterpret_cast<const float4*>(vb + o[u].w);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const float ww[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc.x = fmaf(ww[k], v[u][k].x, acc.x);
        acc.y = fmaf(ww[k], v[u][k].y, acc.y);
        acc.z = fmaf(ww[k], v[u][k].z, acc.z);
        acc.w = fmaf(ww[k], v[u][k].w, acc.w);
      }
    }
  }
  for (; s < ns; ++s) {
    const float4 w = *reinterpret_cast<const float4*>(sp[s].w);
    const int4 o = *reinterpret_cast<const int4*>(sp[s].o);
    const float4 v0 = *reinterpret_cast<const float4*>(vb + o.x);
    const float4 v1 = *reinterpret_cast<const float4*>(vb + o.y);
    const float4 v2 = *reinterpret_cast<const float4*>(vb + o.z);
    const float4 v3 = *reinterpret_cast<const float4*>(vb + o.w);
    acc.x = fmaf(w.x, v0.x, acc.x); acc.y = fmaf(w.x, v0.y, acc.y);
    acc.z = fmaf(w.x, v0.z, acc.z); acc.w = fmaf(w.x, v0.w, acc.w);
    acc.x = fmaf(w.y, v1.x, acc.x); acc.y = fmaf(w.y, v1.y, acc.y);
    acc.z = fmaf(w.y, v1.z, acc.z); acc.w = fmaf(w.y, v1.w, acc.w);
    acc.x = fmaf(w.z, v2.x, acc.x); acc.y = fmaf(w.z, v2.y, acc.y);
    acc.z = fmaf(w.z, v2.z, acc.z); acc.w = fmaf(w.z, v2.w, acc.w);
    acc.x = fmaf(w.w, v3.x, acc.x); acc.y = fmaf(w.w, v3.y, acc.y);
    acc.z = fmaf(w.w, v3.z, acc.z); acc.w = fmaf(w.w, v3.w, acc.w);
  }
  return acc;
}

constexpr unsigned kOobOffset = 0x7fffff00u;   // byte offset no value map reaches: the buffer load returns 0

struct __attribute__((aligned(16))) SampleParamB {   // like SampleParam, offsets in BYTES (global) or LDS bytes
  float w[4];
  unsigned o[4];
};

// bilinear_setup with byte offsets: corner k of pixel (h, w) -> (lvl_pix0 + h*W + w) * pix_bytes; corners outside the map
// (and every corner of a sample that fails the admission test, or when !live) get `dead` (weight 0).  Returns the number of
// corners inside the map.
__device__ __forceinline__ int bilinear_setup_b(float loc_x, float loc_y, float attn, int H, int W, int lvl_pix0,
                                                unsigned pix_bytes, unsigned dead, int live, SampleParamB& sp) {
  const BilinearTerms t = bilinear_terms(loc_x, loc_y, H, W, live);
  const int base = lvl_pix0 + t.h_low * W + t.w_low;
  sp.w[0] = t.c[0] ? t.hh * t.hw * attn : 0.f; sp.o[0] = t.c[0] ? (unsigned)base * pix_bytes : dead;
  sp.w[1] = t.c[1] ? t.hh * t.lw * attn : 0.f; sp.o[1] = t.c[1] ? (unsigned)(base + 1) * pix_bytes : dead;
  sp.w[2] = t.c[2] ? t.lh * t.hw * attn : 0.f; sp.o[2] = t.c[2] ? (unsigned)(base + W) * pix_bytes : dead;
  sp.w[3] = t.c[3] ? t.lh * t.lw * attn : 0.f; sp.o[3] = t.c[3] ? (unsigned)(base + W + 1) * pix_bytes : dead;
  return t.c[0] + t.c[1] + t.c[2] + t.c[3];
}

typedef unsigned occ_u32x4 __attribute__((ext_vector_type(4)));

// Buffer descriptor over `bytes` bytes at `base`, built from PROVABLY wave-uniform words: anything derived from
// threadIdx — even the wave id — is divergent to hipcc, which then wraps every buffer load in a waterfall loop
// (v_readfirstlane x4, compare, s_and_saveexec, loop) that serialises the loads.  `base` must really be the same
// for all lanes of the wave.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  void* p = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// Block barrier that orders LDS traffic only.  __syncthreads() is a full fence — hipcc puts `s_waitcnt vmcnt(0)` in front of
// the s_barrier, which drains every register ring of global loads that is meant to run ACROSS the barrier (and waits for the
// acknowledgement of every earlier store).  Use only where no global memory is shared between the threads of the block.
__device__ __forceinline__ void block_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float4 buf_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0));
}

__device__ __forceinline__ void fma4(float4& acc, float w, const float4& v) {
  acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y);
  acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
}

// gather_samples with BUFFER loads: offsets in bytes relative to the descriptor's base (this lane's 16 bytes of a
// row at + lane_off), out-of-map corners carry kOobOffset -> the hardware returns 0 and requests nothing.
template <int UNROLL>
__device__ __forceinline__ float4 gather_samples_buf(__amdgpu_buffer_rsrc_t rsrc, unsigned lane_off,
                                                     const SampleParamB* sp, int ns, float4 acc) {
  int s = 0;
  for (; s + UNROLL <= ns; s += UNROLL) {
    float4 v[UNROLL][4];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const occ_u32x4 o = *reinterpret_cast<const occ_u32x4*>(sp[s + u].o);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[u][k] = buf_load16(rsrc, o[k] + lane_off);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const float4 w = *reinterpret_cast<const float4*>(sp[s + u].w);
      fma4(acc, w.x, v[u][0]); fma4(acc, w.y, v[u][1]); fma4(acc, w.z, v[u][2]); fma4(acc, w.w, v[u][3]);
    }
  }
  for (; s < ns; ++s) {
    const occ_u32x4 o = *reinterpret_cast<const occ_u32x4*>(sp[s].o);
    const float4 w = *reinterpret_cast<const float4*>(sp[s].w);
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = buf_load16(rsrc, o[k] + lane_off);
    fma4(acc, w.x, v[0]); fma4(acc, w.y, v[1]); fma4(acc, w.z, v[2]); fma4(acc, w.w, v[3]);
  }
  return acc;
}

// acc += w * (fp16 in the low / high half of `packed`): v_fma_mix_f32 widens the fp16 operand inside the FMA.  Written as
// asm because hipcc only selects the mixed-precision FMA when f32 denormals are flushed (its default keeps them) and
// otherwise emits a v_cvt_f32_f16 per value: twice the VALU work of the fp32 gather.  Pure VALU, register operands only:
// the compiler still places the s_waitcnt for the buffer load that produced `packed`.
__device__ __forceinline__ void fma_mix_lo(float& acc, float w, unsigned packed) {
  asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(acc) : "v"(w), "v"(packed));
}
__device__ __forceinline__ void fma_mix_hi(float& acc, float w, unsigned packed) {
  asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(acc) : "v"(w), "v"(packed));
}

// acc (channels 0-3 of this lane's 8) / acc2 (channels 4-7) += w * (eight fp16 values of one 16-byte load)
__device__ __forceinline__ void fma8h(float4& acc, float4& acc2, float w, const float4& raw) {
  const occ_u32x4 h = __builtin_bit_cast(occ_u32x4, raw);
  fma_mix_lo(acc.x, w, h[0]); fma_mix_hi(acc.y, w, h[0]); fma_mix_lo(acc.z, w, h[1]); fma_mix_hi(acc.w, w, h[1]);
  fma_mix_lo(acc2.x, w, h[2]); fma_mix_hi(acc2.y, w, h[2]); fma_mix_lo(acc2.z, w, h[3]); fma_mix_hi(acc2.w, w, h[3]);
}

// ---- q16 value rows (round 6): block floating point, 16 bits per element like the fp16 rows ---------------------------------
// A 16-byte piece = 8 channels of one head of one pixel = 8 two's-complement int16 mantissas q_j sharing ONE 4-bit exponent E:
//     value_j * s = q_j * 2^(E - 15)          (s = the plane's power-of-two range scale, |value * s| <= 2^15: value_range.hip)
// E is the binary exponent of the piece's largest |value * s| (clamped to 0 .. 15), so the largest element uses all 15
// magnitude bits: its rounding error is 2^-16 relative against fp16's 2^-12 — and EVERY element of the piece is rounded to that
// same absolute step, which is what the weighted sums downstream see (their error is set by the large terms).  (The stored
// rows carry s / 2, not s: a binade of headroom for the encoder, see below.)  E lives in
// the two low bits of elements 0 and 1 (E & 3, E >> 2): those two carry 14-bit mantissas, chosen by the encoder as the
// nearest value with the forced low bits, so the decoder uses all eight int16 as they are.  Decode cost per 16-byte load:
// 3 VALU for E, one v_ldexp for w * 2^E, 8 v_cvt_f32_i32 (SDWA word select, sign-extending) and 4 v_pk_fma_f32 — against
// 8 v_fma_mix for fp16 rows.  The gather's result carries the factor 2^15 * s / 2, divided out with the camera count.
__device__ __forceinline__ void fma8q(float4& acc, float4& acc2, float w, const float4& raw) {
  typedef float occ_f32x2 __attribute__((ext_vector_type(2)));
  const occ_u32x4 h = __builtin_bit_cast(occ_u32x4, raw);
  const unsigned e = (h[0] & 3u) | ((h[0] >> 14) & 12u);
  const float ws = ldexpf(w, (int)e);
  const occ_f32x2 ww = {ws, ws};
  occ_f32x2 a0 = {acc.x, acc.y}, a1 = {acc.z, acc.w}, a2 = {acc2.x, acc2.y}, a3 = {acc2.z, acc2.w};
  const occ_f32x2 v0 = {(float)(short)(h[0] & 0xffffu), (float)((int)h[0] >> 16)};
  const occ_f32x2 v1 = {(float)(short)(h[1] & 0xffffu), (float)((int)h[1] >> 16)};
  const occ_f32x2 v2 = {(float)(short)(h[2] & 0xffffu), (float)((int)h[2] >> 16)};
  const occ_f32x2 v3 = {(float)(short)(h[3] & 0xffffu), (float)((int)h[3] >> 16)};
  a0 = __builtin_elementwise_fma(v0, ww, a0);
  a1 = __builtin_elementwise_fma(v1, ww, a1);
  a2 = __builtin_elementwise_fma(v2, ww, a2);
  a3 = __builtin_elementwise_fma(v3, ww, a3);
  acc = make_float4(a0[0], a0[1], a1[0], a1[1]);
  acc2 = make_float4(a2[0], a2[1], a3[0], a3[1]);
}

// q16 ENCODER (the value projection's epilogue and occ_sca_rows_encode_q16; restated in numpy by tests/q16_ref.py).  The rows
// are stored under s' = s / 2 (s = the plane's range scale, |value * s| <= 2^15: one binade of headroom, so no mantissa can
// reach +-2^15 and the encoder needs no clamp); eosc = log2(s').  For a GROUP of channels sharing one exponent (a 16-byte
// piece of 8, or a whole 64-byte head row of 32: a coarser group is a valid, slightly less precise encoding — the decoder
// reads E per piece either way) with largest magnitude m:
//     E = clip(frexp_exp(m * (1 + 2^-12)) + eosc, 0, 15)   (m = 0: E = 0),        f = 2^(eosc + 15 - E),
//     elements 2 .. 7 of a piece:  q = rne(v * f);      elements 0, 1:  q = 4 * rne(v * f / 4 - r / 4) + r,  r = E & 3, E >> 2
// rne by the magic-number add (x + 1.5 * 2^23 leaves rne(x) in the low mantissa bits: the low 16 bits ARE the two's
// complement mantissa); the 2^-12 margin keeps |v * f| <= 32 760, so the tagged elements stay inside int16 as well.
// Non-finite inputs have no q16 image: they encode to unspecified FINITE mantissas (the fp16 rows clamp them).
constexpr float kQ16Magic = 12582912.f;
__device__ __forceinline__ int q16_group_exponent(float m, int eosc) {
  const int x = __builtin_amdgcn_frexp_expf(m * 1.000244140625f) + eosc;
  const int E = x < 0 ? 0 : x > 15 ? 15 : x;
  return m > 0.f ? E : 0;
}
__device__ __forceinline__ unsigned q16_rne_bits(float y) { return __float_as_uint(y + kQ16Magic); }
// the mantissa words of one piece's elements (j, j + 1): regular, or tagged with (r0, r1) when sh == 2 (sh == 0: r = 0, fS = f)
__device__ __forceinline__ unsigned q16_pack_lo16(unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x05040100u); }
__device__ __forceinline__ unsigned q16_pair(float v0, float v1, float f) {
  return q16_pack_lo16(q16_rne_bits(v0 * f), q16_rne_bits(v1 * f));
}
__device__ __forceinline__ unsigned q16_pair_tagged(float v0, float v1, float fS, float c0, float c1, int r0, int r1, int sh) {
  const unsigned q0 = (q16_rne_bits(fmaf(v0, fS, c0)) << sh) + (unsigned)r0;
  const unsigned q1 = (q16_rne_bits(fmaf(v1, fS, c1)) << sh) + (unsigned)r1;
  return q16_pack_lo16(q0, q1);
}

template <bool Q>
__device__ __forceinline__ void fma8x(float4& acc, float4& acc2, float w, const float4& raw) {
  if (Q) fma8q(acc, acc2, w, raw); else fma8h(acc, acc2, w, raw);
}

// Gather over 16-bit value rows (fp16, or q16 when Q): a lane's 16 bytes are 8 channels (4 lanes per 64-byte head row).  ROLLING window over NS
// (compile-time) samples: the 4 corner loads of sample j + DEPTH are requested right after sample j is consumed, so
// DEPTH * 4 loads stay in flight instead of batches that drain to zero.  The window loop is NOT unrolled (fully unrolled,
// hipcc hoists the parameter reads of later samples and spills); the weights are read when the sample is consumed (their
// LDS latency hides under the wait for the rows; held from issue to consumption they cost 16 more live registers).
template <int NS, int DEPTH_ = 4, bool Q = false>
__device__ __forceinline__ void gather_samples_buf_h(__amdgpu_buffer_rsrc_t rsrc, unsigned lane_off,
                                                     const SampleParamB* sp, float4& acc, float4& acc2) {
  constexpr int DEPTH = NS < DEPTH_ ? NS : DEPTH_;
  static_assert(NS % DEPTH == 0, "sample count must be a multiple of the window");
  float4 v[DEPTH][4];
#pragma unroll
  for (int u = 0; u < DEPTH; ++u) {
    const occ_u32x4 o = *reinterpret_cast<const occ_u32x4*>(sp[u].o);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[u][k] = buf_load16(rsrc, o[k] + lane_off);
  }
#pragma unroll 1
  for (int j0 = 0; j0 + DEPTH < NS; j0 += DEPTH) {
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      const float4 w = *reinterpret_cast<const float4*>(sp[j0 + u].w);
      const occ_u32x4 o = *reinterpret_cast<const occ_u32x4*>(sp[j0 + DEPTH + u].o);
      fma8x<Q>(acc, acc2, w.x, v[u][0]); fma8x<Q>(acc, acc2, w.y, v[u][1]);
      fma8x<Q>(acc, acc2, w.z, v[u][2]); fma8x<Q>(acc, acc2, w.w, v[u][3]);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[u][k] = buf_load16(rsrc, o[k] + lane_off);
    }
  }
#pragma unroll
  for (int u = 0; u < DEPTH; ++u) {
    const float4 w = *reinterpret_cast<const float4*>(sp[NS - DEPTH + u].w);
    fma8x<Q>(acc, acc2, w.x, v[u][0]); fma8x<Q>(acc, acc2, w.y, v[u][1]);
    fma8x<Q>(acc, acc2, w.z, v[u][2]); fma8x<Q>(acc, acc2, w.w, v[u][3]);
  }
}

}  // namespace occ
