// Backbone 3x3 / pad-1 convolutions, stride 1 or 2 (NHWC bf16) as an implicit GEMM on the gfx950 bf16 matrix
// cores with bias + ReLU fused: out = relu?( conv3x3(x, W) + bias ), bf16 in / f32 accumulate / bf16 out.
//
// NOT part of the hand-written hot path: it replaces MIOpen's igemm kernel plus its
// zero-fill / cast helpers plus the elementwise tail for the stride-1 3x3 convolutions of the reference's
// ResNet-50 (mmdet Bottleneck conv2, style='pytorch', incl. the stride-2 first blocks of layer2..4) and the
// FPN output / extra-level convolutions, when Cout % 128 == 0 and Cin % 32 == 0; the 7x7 stem stays on MIOpen
// (the 64-channel layer1 blocks have their own whole-bottleneck kernel).
//
// GEMM view: M = output pixels, N = Cout, K = 9 taps x Cin.  Block = 4 waves x (8 x 16 output pixels = four
// 32-pixel MFMA row tiles of two image rows each) x 128*NT output channels; waves split N.  Per chunk of 32
// input channels the (8+2) x (16+2) pixel halo is staged ONCE into LDS (zero-filled outside the image =
// the convolution's padding; double buffered; 80-byte pixel slots, 1536-byte halo rows: conflict-free
// ds_read_b128) and reused by the 9 taps with compile-time offsets; the weights are pre-packed in MFMA
// B-fragment order ([chunk][tap][k-step][Cout/32][lane][8]) and every wave streams the operands of its own
// 32*NT columns global -> registers through a ring of 6 k-steps — they never touch LDS;
// the chunk order is rotated per block (L2 channel hot-spotting, see conv1x1_nhwc_bf16.hip).
// v_mfma_f32_32x32x16_bf16, 2 k-steps per (chunk, tap).
#include "common.h"

namespace occ {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kC3TW = 16;                             // output tile width (one MFMA row tile = 2 x 16 pixels)
constexpr int kC3PX = 80;                             // bytes per halo pixel slot (32 bf16 + 16 pad)
// Tile geometry per stride S: output tile TH x 16, input halo HH x HW.  Stride 2 keeps the halo columns
// DE-INTERLEAVED in LDS (17 even columns, then 17 odd-column slots): for a fixed tap the 16 output pixels of a
// row then read 16 CONSECUTIVE 80-byte slots, conflict-free like stride 1 (a 160-byte lane stride is not).
template <int S> struct C3Geom {
  static constexpr int TH = S == 1 ? 8 : 4, RT = TH / 2;
  static constexpr int HH = (TH - 1) * S + 3, HW = (kC3TW - 1) * S + 3;     // 10 x 18  /  9 x 33
  static constexpr int ROW = S == 1 ? 1536 : 34 * kC3PX;                    // bytes per halo row
  static constexpr int ITEMS = HH * HW * 4, NR = (ITEMS + 255) / 256;       // 16-byte staging items / roles
  __device__ static constexpr int slot(int hx) { return S == 1 ? hx : (hx & 1) * 17 + (hx >> 1); }
};

__device__ __forceinline__ unsigned short c3_f32_to_bf16(float f) { return bf16_rne(f); }

typedef unsigned short c3_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned c3_absmax2(unsigned m, unsigned x) {   // v_and + v_pk_max_u16
  const c3_u16x2 a = __builtin_bit_cast(c3_u16x2, m), b = __builtin_bit_cast(c3_u16x2, x & 0x7fff7fffu);
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(a, b));
}

// torch weight (Cout, Cin, 3, 3) f32 -> bf16 in MFMA B-fragment order
// packed[chunk = ci/32][tap = ky*3+kx][ks = 0,1][Cout/32][lane][8]:  element j of lane `lane` is
// w[co = nt*32 + (lane & 31)][ci = chunk*32 + ks*16 + (lane >> 5)*8 + j][tap]
__global__ void conv3x3_pack_weight_kernel(const float* __restrict__ w, unsigned short* __restrict__ packed,
                                           int Cout, int Cin) {
  const long n = (long)Cout * Cin * 9;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
  long r = idx >> 9;
  const int nt32 = Cout / 32;
  const int nt = (int)(r % nt32); r /= nt32;
  const int ks = (int)(r & 1); r >>= 1;
  const int tap = (int)(r % 9);
  const int chunk = (int)(r / 9);
  const int co = nt * 32 + (lane & 31), ci = chunk * 32 + ks * 16 + (lane >> 5) * 8 + j;
  packed[idx] = c3_f32_to_bf16(w[((long)co * Cin + ci) * 9 + tap]);
}

template <int NT, int S, int PF>
__global__ __launch_bounds__(256, 3) void conv3x3_nhwc_bf16_kernel(
    const uint4* __restrict__ x, const uint4* __restrict__ wp, const float* __restrict__ bias,
    unsigned short* __restrict__ out, int H, int W, int Ho, int Wo, int Cin, int Cout, int tiles_x,
    int tiles_y, int relu, unsigned* __restrict__ amax8) {
  using G = C3Geom<S>;
  constexpr int RT = G::RT, BN = 128 * NT, WR = 32 * NT, OLD = BN + 4;
  constexpr int kC3ROW = G::ROW, kC3TH = G::TH;
  constexpr int HALO_BYTES = G::HH * kC3ROW;                 // 15 360 (stride 1) / 24 480 (stride 2)
  constexpr int STAGE_BYTES = 2 * HALO_BYTES, OUT_BYTES = 32 * OLD * 4;   // LDS carries only the halo
  __shared__ __attribute__((aligned(16))) char lds[STAGE_BYTES > OUT_BYTES ? STAGE_BYTES : OUT_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vi = lane & 31, kb = lane >> 5;
  int bid = blockIdx.x;
  const int tx_i = bid % tiles_x; bid /= tiles_x;
  const int ty_i = bid % tiles_y;
  const int img = bid / tiles_y;
  const int y0 = ty_i * kC3TH, x0 = tx_i * kC3TW;
  const int n0 = blockIdx.y * BN, nw0 = n0 + wave * WR;
  const int CQ = Cin / 8, NCH = Cin / 32;

  f32x16 acc[RT][NT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rt][t][r] = 0.f;

  // halo staging roles: HH x HW pixels x 4 pieces of 16 B over 256 threads (3 per thread at stride 1, 5 at
  // stride 2; clamped + zero-masked, unconditional loads)
  // (named scalars: small per-thread arrays written in a loop end up in scratch with hipcc / ROCm 7.2)
  constexpr int NR = G::NR;
  static_assert(NR <= 5, "halo staging register budget");
  long hofs0, hofs1, hofs2, hofs3 = 0, hofs4 = 0;
  int hdst0, hdst1, hdst2, hdst3 = 0, hdst4 = 0;
  bool hin0, hin1, hin2, hin3 = false, hin4 = false, hlive0, hlive1, hlive2, hlive3 = false, hlive4 = false;
#define OCC_C3_HALO_ROLE(K, OFS, DST, IN, LIVE)                                                   \
  {                                                                                               \
    const int idx = tid + 256 * (K);                                                              \
    LIVE = idx < G::ITEMS;                                                                        \
    const int p = LIVE ? idx >> 2 : 0, piece = idx & 3;                                           \
    const int hy = p / G::HW, hx = p % G::HW;                                                     \
    const int iy = y0 * S - 1 + hy, ix = x0 * S - 1 + hx;                                         \
    IN = LIVE && iy >= 0 && iy < H && ix >= 0 && ix < W;                                          \
    const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);                           \
    OFS = (((long)img * H + cy) * W + cx) * CQ + piece;                                           \
    DST = hy * kC3ROW + G::slot(hx) * kC3PX + piece * 16;                                         \
  }
  OCC_C3_HALO_ROLE(0, hofs0, hdst0, hin0, hlive0)
  OCC_C3_HALO_ROLE(1, hofs1, hdst1, hin1, hlive1)
  OCC_C3_HALO_ROLE(2, hofs2, hdst2, hin2, hlive2)
  if (NR > 3) OCC_C3_HALO_ROLE(3, hofs3, hdst3, hin3, hlive3)
  if (NR > 4) OCC_C3_HALO_ROLE(4, hofs4, hdst4, hin4, hlive4)
#undef OCC_C3_HALO_ROLE
  // this wave's 32-column tiles in the packed weight: uint4 index of (tile, lane) inside one (chunk, tap, ks)
  const int NT32 = Cout / 32;
  static_assert(NT <= 2, "weight ring register budget");
  const long wl0 = (long)((nw0 / 32) < NT32 ? nw0 / 32 : NT32 - 1) * 64 + lane;
  const long wl1 = (long)((nw0 / 32 + NT - 1) < NT32 ? nw0 / 32 + NT - 1 : NT32 - 1) * 64 + lane;
  // A fragment base offsets (bytes) of this lane's pixel in each row tile, tap (0,0), k-step 0
  int abase[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
    abase[rt] = (2 * rt + (vi >> 4)) * S * kC3ROW + (vi & 15) * kC3PX + kb * 16;

  uint4 vh0, vh1, vh2, vh3, vh4;
  const unsigned hm0 = hin0 ? 0xffffffffu : 0u, hm1 = hin1 ? 0xffffffffu : 0u, hm2 = hin2 ? 0xffffffffu : 0u;
  const unsigned hm3 = hin3 ? 0xffffffffu : 0u, hm4 = hin4 ? 0xffffffffu : 0u;
#define OCC_C3_ISSUE_HALO(CH)                                                                     \
  {                                                                                               \
    const long cq = (long)(CH) * 4;                                                               \
    vh0 = x[hofs0 + cq]; vh1 = x[hofs1 + cq]; vh2 = x[hofs2 + cq];                                \
    if (NR > 3) vh3 = x[hofs3 + cq];                                                              \
    if (NR > 4) vh4 = x[hofs4 + cq];                                                              \
  }
  // Weights: MFMA-fragment-ordered, global -> registers, a ring of PF k-steps in flight (k-step = (tap, ks),
  // 18 per chunk; the ring runs on across chunk boundaries).  Each wave owns its columns, so there is nothing
  // to share through LDS; the scheduling barrier per k-step keeps hipcc from sinking the prefetch.
  static_assert(36 % PF == 0 && PF <= 18, "ring slot pattern repeats every two chunks");
  uint4 wr[PF][NT];
#define OCC_C3_W(CHK, T18, T) wp[(((long)(CHK) * 18 + (T18)) * NT32) * 64 + ((T) == 0 ? wl0 : wl1)]

  const int rot = (int)((blockIdx.x * 5u + blockIdx.y * 3u) % (unsigned)NCH);
#define OCC_C3_CH(CI) ((((CI) < NCH ? (CI) : NCH - 1) + rot) % NCH)
  OCC_C3_ISSUE_HALO(OCC_C3_CH(0))
#pragma unroll
  for (int s = 0; s < PF; ++s)
#pragma unroll
    for (int t = 0; t < NT; ++t) wr[s][t] = OCC_C3_W(OCC_C3_CH(0), s, t);

  // one chunk of 32 input channels: halo registers -> LDS buffer PAR, barrier, next chunk's halo requested,
  // 18 k-steps straight out of the halo with compile-time tap offsets
#define OCC_C3_CHUNK(PAR, CI)                                                                     \
  {                                                                                               \
    char* sH = lds + (PAR) * HALO_BYTES;                                                          \
    /* out-of-image pixels are zero (the convolution's padding): AND with an all-ones / all-zeros mask */ \
    if (hlive0) *reinterpret_cast<uint4*>(sH + hdst0) = make_uint4(vh0.x & hm0, vh0.y & hm0, vh0.z & hm0, vh0.w & hm0); \
    if (hlive1) *reinterpret_cast<uint4*>(sH + hdst1) = make_uint4(vh1.x & hm1, vh1.y & hm1, vh1.z & hm1, vh1.w & hm1); \
    if (hlive2) *reinterpret_cast<uint4*>(sH + hdst2) = make_uint4(vh2.x & hm2, vh2.y & hm2, vh2.z & hm2, vh2.w & hm2); \
    if (NR > 3 && hlive3) *reinterpret_cast<uint4*>(sH + hdst3) = make_uint4(vh3.x & hm3, vh3.y & hm3, vh3.z & hm3, vh3.w & hm3); \
    if (NR > 4 && hlive4) *reinterpret_cast<uint4*>(sH + hdst4) = make_uint4(vh4.x & hm4, vh4.y & hm4, vh4.z & hm4, vh4.w & hm4); \
    __syncthreads();   /* halo chunk visible; the other halo buffer is free for the next chunk */ \
    OCC_C3_ISSUE_HALO(OCC_C3_CH((CI) + 1))                                                        \
    const int ch_cur = OCC_C3_CH(CI), ch_nxt = OCC_C3_CH((CI) + 1);                               \
    bf16x8 af[RT], an[RT];                                                                        \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                             \
      af[rt] = *reinterpret_cast<const bf16x8*>(sH + abase[rt]);                                  \
    _Pragma("unroll") for (int s = 0; s < 18; ++s) {                                              \
      constexpr int dummy_ = 0; (void)dummy_;                                                     \
      const int slot = ((PAR) * 18 + s) % PF;                                                     \
      bf16x8 wf[NT];                                                                              \
      _Pragma("unroll") for (int t = 0; t < NT; ++t) wf[t] = __builtin_bit_cast(bf16x8, wr[slot][t]); \
      {                                                                                           \
        const int sn = s + PF;                                                                    \
        _Pragma("unroll") for (int t = 0; t < NT; ++t)                                            \
          wr[slot][t] = OCC_C3_W(sn < 18 ? ch_cur : ch_nxt, sn % 18, t);                          \
      }                                                                                           \
      if (s + 1 < 18) {                                                                           \
        const int tap = (s + 1) >> 1, ks = (s + 1) & 1;                                           \
        const int toff = (tap / 3) * kC3ROW + G::slot(tap % 3) * kC3PX + ks * 32;                 \
        _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                         \
          an[rt] = *reinterpret_cast<const bf16x8*>(sH + abase[rt] + toff);                       \
      }                                                                                           \
      _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                           \
        _Pragma("unroll") for (int t = 0; t < NT; ++t)                                            \
          acc[rt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt], wf[t], acc[rt][t], 0, 0, 0); \
      __builtin_amdgcn_sched_barrier(0);                                                          \
      if (s + 1 < 18) { _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) af[rt] = an[rt]; }      \
    }                                                                                             \
  }
  for (int ch = 0; ch < NCH; ch += 2) {
    OCC_C3_CHUNK(0, ch)
    if (ch + 1 < NCH) OCC_C3_CHUNK(1, ch + 1)
  }
#undef OCC_C3_CHUNK
#undef OCC_C3_ISSUE_HALO
#undef OCC_C3_W
#undef OCC_C3_CH

  // ---- epilogue, one 32-pixel row tile (two image rows) at a time through an LDS transpose -------------
  const int c = lane * 4;
  const bool col_live = c < BN && n0 + c < Cout;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col_live) bv = *reinterpret_cast<const float4*>(bias + n0 + c);
  float* sO = reinterpret_cast<float*>(lds);
  unsigned amax2 = 0;                                        // two 16-bit maxima of the sign-stripped bf16 patterns stored
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
      const int row = wave * 8 + rr;                       // pixel inside the row tile
      const int oy = y0 + 2 * rt + (row >> 4), ox = x0 + (row & 15);
      if (oy < Ho && ox < Wo && col_live) {
        float4 v = *reinterpret_cast<const float4*>(sO + row * OLD + c);
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        if (relu) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        const uint2 o = make_uint2(pack_bf16x2_rne(v.x, v.y), pack_bf16x2_rne(v.z, v.w));
        *reinterpret_cast<uint2*>(out + (((long)img * Ho + oy) * Wo + ox) * Cout + n0 + c) = o;
        amax2 = c3_absmax2(c3_absmax2(amax2, o.x), o.y);
      }
    }
  }
  // max|out| of what this launch stored, for the consumer's fp16 range scale (value_range.hip): one atomic per wave into
  // one of 8 words (Inf / NaN patterns order above every finite one and reach the consumer as such)
  if (amax8 != nullptr) {
    unsigned m16 = (amax2 & 0xffffu) > (amax2 >> 16) ? (amax2 & 0xffffu) : (amax2 >> 16);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const unsigned o = (unsigned)__shfl_xor((int)m16, d);
      m16 = o > m16 ? o : m16;
    }
    if (lane == 0 && m16 != 0u) atomicMax(amax8 + ((blockIdx.x + wave) & 7u), m16);
  }
}

}  // namespace occ

extern "C" int occ_conv3x3_pack_weight_bf16(const float* weight, void* packed, int Cout, int Cin,
                                            void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(weight && packed, "conv3x3_pack_weight_bf16: null pointer argument");
  if (Cin % 32 || Cout % 128) {
    set_error("conv3x3_pack_weight_bf16: no kernel for Cin=%d Cout=%d (need Cin %% 32 == 0, Cout %% 128 == 0)",
              Cin, Cout);
    return OCC_E_UNSUPPORTED;
  }
  const long n = (long)Cout * Cin * 9;
  hipLaunchKernelGGL(conv3x3_pack_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), weight,
                     reinterpret_cast<unsigned short*>(packed), Cout, Cin);
  OCC_CHECK_LAUNCH("conv3x3_pack_weight_bf16");
  return OCC_OK;
}

static int conv3x3_nhwc_bf16_launch(const void* x, const void* weight_packed, const float* bias, void* out,
                                    int batch, int H, int W, int Cin, int Cout, int stride, int relu,
                                    uint32_t* amax8, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(x && weight_packed && bias && out, "conv3x3_nhwc_bf16: null pointer argument");
  OCC_CHECK_ARG(batch > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv3x3_nhwc_bf16: bad dimension");
  if (Cin % 32 || Cout % 128 || (stride != 1 && stride != 2)) {
    set_error("conv3x3_nhwc_bf16: no kernel for Cin=%d Cout=%d stride=%d (need Cin %% 32 == 0, Cout %% 128 == 0, "
              "stride 1 or 2)", Cin, Cout, stride);
    return OCC_E_UNSUPPORTED;
  }
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;     // floor((H + 2 - 3) / stride) + 1
  const int TH = stride == 1 ? C3Geom<1>::TH : C3Geom<2>::TH;
  const int tiles_x = (Wo + kC3TW - 1) / kC3TW, tiles_y = (Ho + TH - 1) / TH;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const unsigned gx = (unsigned)((long)batch * tiles_x * tiles_y);
#define OCC_C3_LAUNCH(NTT, BNN, SS, PFF)                                                            \
  hipLaunchKernelGGL((conv3x3_nhwc_bf16_kernel<NTT, SS, PFF>), dim3(gx, (unsigned)(Cout / BNN)), dim3(256), 0, st, \
                     reinterpret_cast<const uint4*>(x), reinterpret_cast<const uint4*>(weight_packed), \
                     bias, reinterpret_cast<unsigned short*>(out), H, W, Ho, Wo, Cin, Cout, tiles_x, tiles_y, relu, amax8)
  // ring of 6 k-steps + 3 waves per SIMD measured faster than 12 k-steps + 2 waves on every ResNet-50 shape
  if (stride == 2) OCC_C3_LAUNCH(1, 128, 2, 6); else OCC_C3_LAUNCH(1, 128, 1, 6);
#undef OCC_C3_LAUNCH
  OCC_CHECK_LAUNCH("conv3x3_nhwc_bf16");
  return OCC_OK;
}

extern "C" int occ_conv3x3_nhwc_bf16(const void* x, const void* weight_packed, const float* bias, void* out,
                                     int batch, int H, int W, int Cin, int Cout, int stride, int relu,
                                     void* stream) {
  return conv3x3_nhwc_bf16_launch(x, weight_packed, bias, out, batch, H, W, Cin, Cout, stride, relu, nullptr, stream);
}

// The same convolution; additionally folds max|out| (sign-stripped bf16 patterns of every element it stores) into amax8[0..8)
// with atomic maxima — the words ACCUMULATE across launches (the caller zeroes them once in front of the FPN's output
// convolutions) and feed occ_value_range_scale_from_amax: the consumer's range pass over the maps is not needed.
extern "C" int occ_conv3x3_nhwc_bf16_amax(const void* x, const void* weight_packed, const float* bias, void* out,
                                          int batch, int H, int W, int Cin, int Cout, int stride, int relu,
                                          uint32_t* amax8, void* stream) {
  OCC_CHECK_ARG(amax8, "conv3x3_nhwc_bf16_amax: null amax8");
  return conv3x3_nhwc_bf16_launch(x, weight_packed, bias, out, batch, H, W, Cin, Cout, stride, relu, amax8, stream);
}
