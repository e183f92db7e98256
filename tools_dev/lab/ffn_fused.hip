// The encoder's feed-forward block as ONE kernel on the gfx950 bf16 matrix cores (bf16x3 arithmetic):
//
//     out = LayerNorm( x + W2 . relu(W1 . x + b1) + b2 )          x: (M, 256), hidden 512
//
// = mmcv FFN (Linear -> ReLU -> Linear, + identity) followed by the layer's last
// `nn.LayerNorm` — reference: projects/mmdet3d_plugin/bevformer/modules/encoder.py:377-404 (operation_order
// (..., 'ffn', 'norm')), custom_base_transformer_layer.py:74-99.  Round 1 ran this as two Linear launches with an
// 82 MB hidden tensor in between (41 in + 82 out, then 82 + 41 in + 41 out = 287 MB of HBM traffic; both launches
// are bound by that traffic: r02 trace, 47 + 54 us at ~3 TB/s).  Here the hidden activations never leave the
// registers (41 MB in + 41 residual re-read through L2 + 41 out).
//
// Everything is computed TRANSPOSED — features x rows — so that the D registers of one GEMM are a legal B operand
// of the next without any cross-lane traffic (the occ_heads kernel's trick, on the bf16 32x32x16 MFMA):
//   H^T (32 hidden x 32 rows) = W1[tile h] (32 x 256) . X^T      A = weights (lane m = hidden unit), B = X^T
//   Y^T (256 x 32 rows)      += W2[:, tile h] (256 x 32) . relu(H^T)
// A lane's D registers of the H^T tile hold hidden units 8*(i/4) + 4*(lane/32) + i%4 of activation row lane%32;
// the B operand of a 16-k step wants 8 k-values per lane: registers 0..7 / 8..15 ARE the two k-steps when the k
// order inside each 16-group is permuted to  k-slot j of lane-half g  <->  unit 8*(j/4) + 4g + j%4 — and W2 is
// packed with that permutation (occ_ffn_pack_weights), so the contraction is unchanged.
// bf16x3: every f32 operand is hi + lo bf16, a.b ~= al.bh + ah.bl + ah.bh in f32 accumulation (linear_bf16x3.hip).
//
// Decomposition: block = 4 waves = 128 rows, ONE wave per SIMD (the wave keeps its 32 x 256 input tile as hi/lo
// fragments, 128 VGPRs, and the 256 x 32 output accumulators, 128 more); the weights stream through a two-slot LDS
// ring in 32 KB chunks shared by the four waves (3-slot ring, fetched two chunks ahead) — chunk 2h = W1 tile h (16 k-steps), chunk 2h+1 = the W2 column slice
// of tile h (8 feature tiles x 2 k-steps): 48 MFMAs per wave and chunk, one barrier per chunk, the next chunk's
// global loads in flight under the MFMAs.  Epilogue: + b2 + x (re-read in the D layout: 16-byte pieces, L2
// resident) -> two-pass LayerNorm per row (128 features per lane + one cross-half shuffle) -> 16-byte stores.
#include "common.h"

namespace occ {

typedef float ffn_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 ffn_bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kFfnC = 256, kFfnHid = 512, kFfnChunk = 32 * 1024;   // bytes per weight chunk
constexpr int kFfnChunks = 2 * (kFfnHid / 32);                     // 32

__device__ __forceinline__ void ffn_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  hi = pack_bf16x2_rne(x0, x1);
  lo = pack_bf16x2_rne(x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u));
}

// W1 (512, 256), W2 (256, 512) f32 -> the chunk stream the kernel copies verbatim into LDS (1 MB of bf16):
//   chunk 2h  : [k-step s: 16][plane hi/lo][lane][8]   A fragments of W1 rows 32h..32h+31, natural k order
//   chunk 2h+1: [feature tile t: 8][k-step ks: 2][plane][lane][8]   A fragments of W2 rows 32t.., k = hidden unit
//               32h + 16 ks + 8*(j/4) + 4*(lane/32) + j%4   (the permuted order the H^T registers come in)
__global__ void ffn_pack_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                unsigned short* __restrict__ packed) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one (hi, lo) pair per thread
  if (idx >= (long)kFfnChunks * 8192) return;                        // 8192 weights per chunk
  const int chunk = (int)(idx >> 13), e = (int)(idx & 8191);
  const int j = e & 7, lane = (e >> 3) & 63, frag = e >> 9;          // frag: 0..15
  const int m = lane & 31, g = lane >> 5, h = chunk >> 1;
  float w;
  if ((chunk & 1) == 0) {
    w = w1[(long)(32 * h + m) * kFfnC + 16 * frag + 8 * g + j];
  } else {
    const int t = frag >> 1, ks = frag & 1;
    w = w2[(long)(32 * t + m) * kFfnHid + 32 * h + 16 * ks + 8 * (j >> 2) + 4 * g + (j & 3)];
  }
  const unsigned short hi = bf16_rne(w);
  const unsigned short lo = bf16_rne(w - __uint_as_float((unsigned)hi << 16));
  unsigned short* dst = packed + (long)chunk * (kFfnChunk / 2) + ((long)(frag * 2) * 64 + lane) * 8 + j;
  dst[0] = hi;
  dst[64 * 8] = lo;
}

__global__ __launch_bounds__(256, 1) void ffn_fused_kernel(
    const float* __restrict__ x, long ldx, const uint4* __restrict__ wstream, const float* __restrict__ b1,
    const float* __restrict__ b2, const float* __restrict__ ln_g, const float* __restrict__ ln_b, float ln_eps,
    float* __restrict__ out, long ldo, int M) {
  extern __shared__ __attribute__((aligned(16))) char ring[];      // 3 slots x 32 KB: chunk c lives in slot c % 3
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, g = lane >> 5;
  long row = (long)blockIdx.x * 128 + wave * 32 + n;
  const bool row_live = row < M;
  if (!row_live) row = (long)M - 1;
  const float* xr = x + row * ldx;

  // Weight chunks are fetched TWO chunks ahead into two alternating register sets (a chunk's 48 MFMAs last ~0.7 us,
  // less than a loaded L2 round trip: with one chunk of lead the wave sat in s_waitcnt for half of every chunk) and
  // written to their ring slot one chunk ahead of use.
  uint4 va0, va1, va2, va3, va4, va5, va6, va7, vb0, vb1, vb2, vb3, vb4, vb5, vb6, vb7;
#define OCC_FFN_ISSUE(C, S)                                                                  \
  {                                                                                          \
    const uint4* src_ = wstream + (long)(C) * (kFfnChunk / 16) + tid;                        \
    v##S##0 = src_[0];    v##S##1 = src_[256];  v##S##2 = src_[512];  v##S##3 = src_[768];   \
    v##S##4 = src_[1024]; v##S##5 = src_[1280]; v##S##6 = src_[1536]; v##S##7 = src_[1792]; \
  }
#define OCC_FFN_COMMIT(SLOT, S)                                                                   \
  {                                                                                               \
    uint4* dst_ = reinterpret_cast<uint4*>(ring + (SLOT) * kFfnChunk) + tid;                      \
    dst_[0] = v##S##0;    dst_[256] = v##S##1;  dst_[512] = v##S##2;  dst_[768] = v##S##3;        \
    dst_[1024] = v##S##4; dst_[1280] = v##S##5; dst_[1536] = v##S##6; dst_[1792] = v##S##7;       \
  }
  OCC_FFN_ISSUE(0, a)
  OCC_FFN_ISSUE(1, b)
  // this wave's input tile: B fragments of X^T, hi + lo
  uint4 xh[16], xl[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const float4 a = *reinterpret_cast<const float4*>(xr + 16 * s + 8 * g);
    const float4 b = *reinterpret_cast<const float4*>(xr + 16 * s + 8 * g + 4);
    ffn_split2(a.x, a.y, xh[s].x, xl[s].x); ffn_split2(a.z, a.w, xh[s].y, xl[s].y);
    ffn_split2(b.x, b.y, xh[s].z, xl[s].z); ffn_split2(b.z, b.w, xh[s].w, xl[s].w);
  }
  OCC_FFN_COMMIT(0, a)
  __syncthreads();

  ffn_f32x16 y[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) y[t][i] = 0.f;

  int slot = 0;                                   // ring slot of chunk 2h
#pragma unroll 1
  for (int h = 0; h < kFfnHid / 32; ++h) {
    const int slot1 = slot == 2 ? 0 : slot + 1, slot2 = slot1 == 2 ? 0 : slot1 + 1;     // chunks 2h+1, 2h+2
    // ================= chunk 2h: H^T tile = W1[tile h] . X^T ====================================================
    if (h + 1 < kFfnHid / 32) OCC_FFN_ISSUE(2 * h + 2, a)
    __builtin_amdgcn_sched_barrier(0);
    // three accumulators, one per bf16x3 term: back-to-back MFMAs on ONE accumulator wait out the 64-cycle result
    // latency (issue is 32), and with a single wave on the SIMD nothing else fills the hole
    ffn_f32x16 hacc, hacc1, hacc2;
#pragma unroll
    for (int i = 0; i < 16; ++i) hacc[i] = hacc1[i] = hacc2[i] = 0.f;
    {
      // one wave per SIMD: nobody else hides the LDS latency, so the fragments of step s+1 are requested before
      // the MFMAs of step s (left to itself hipcc waits lgkmcnt(0) between every read and its MFMA)
      const char* sW = ring + slot * kFfnChunk + lane * 16;
      ffn_bf16x8 wh = *reinterpret_cast<const ffn_bf16x8*>(sW);
      ffn_bf16x8 wl = *reinterpret_cast<const ffn_bf16x8*>(sW + 1024);
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        ffn_bf16x8 whn = wh, wln = wl;
        if (s + 1 < 16) {
          whn = *reinterpret_cast<const ffn_bf16x8*>(sW + (s + 1) * 2048);
          wln = *reinterpret_cast<const ffn_bf16x8*>(sW + (s + 1) * 2048 + 1024);
        }
        hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, __builtin_bit_cast(ffn_bf16x8, xh[s]), hacc, 0, 0, 0);
        hacc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, __builtin_bit_cast(ffn_bf16x8, xl[s]), hacc1, 0, 0, 0);
        hacc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, __builtin_bit_cast(ffn_bf16x8, xh[s]), hacc2, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        wh = whn; wl = wln;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) hacc[i] = (hacc[i] + hacc1[i]) + hacc2[i];      // small terms first
    }
    __builtin_amdgcn_sched_barrier(0);
    OCC_FFN_COMMIT(slot1, b)                      // chunk 2h+1, fetched during the previous chunk
    // bias + ReLU, then the two k-steps of the next contraction straight out of the D registers
    uint4 hh0, hl0, hh1, hl1;
    {
      float v[16];
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const float4 bb = *reinterpret_cast<const float4*>(b1 + 32 * h + 8 * q4 + 4 * g);
        v[4 * q4 + 0] = fmaxf(hacc[4 * q4 + 0] + bb.x, 0.f);
        v[4 * q4 + 1] = fmaxf(hacc[4 * q4 + 1] + bb.y, 0.f);
        v[4 * q4 + 2] = fmaxf(hacc[4 * q4 + 2] + bb.z, 0.f);
        v[4 * q4 + 3] = fmaxf(hacc[4 * q4 + 3] + bb.w, 0.f);
      }
      ffn_split2(v[0], v[1], hh0.x, hl0.x);   ffn_split2(v[2], v[3], hh0.y, hl0.y);
      ffn_split2(v[4], v[5], hh0.z, hl0.z);   ffn_split2(v[6], v[7], hh0.w, hl0.w);
      ffn_split2(v[8], v[9], hh1.x, hl1.x);   ffn_split2(v[10], v[11], hh1.y, hl1.y);
      ffn_split2(v[12], v[13], hh1.z, hl1.z); ffn_split2(v[14], v[15], hh1.w, hl1.w);
    }
    __syncthreads();
    // ================= chunk 2h+1: Y^T += W2[:, tile h] . relu(H^T) ===============================================
    if (h + 1 < kFfnHid / 32) OCC_FFN_ISSUE(2 * h + 3, b)
    __builtin_amdgcn_sched_barrier(0);
    {
      // two feature tiles at a time (fragments 2t+ks at (2t+ks) * 2048, lo plane +1024), their MFMAs interleaved so
      // that consecutive instructions never share an accumulator; the next pair's fragments are requested first
      const char* sW = ring + slot1 * kFfnChunk + lane * 16;
      const ffn_bf16x8 bh0 = __builtin_bit_cast(ffn_bf16x8, hh0), bl0 = __builtin_bit_cast(ffn_bf16x8, hl0);
      const ffn_bf16x8 bh1 = __builtin_bit_cast(ffn_bf16x8, hh1), bl1 = __builtin_bit_cast(ffn_bf16x8, hl1);
      ffn_bf16x8 w[8], wn[8];       // [tile a/b][ks][hi, lo]
#pragma unroll
      for (int e = 0; e < 8; ++e) w[e] = *reinterpret_cast<const ffn_bf16x8*>(sW + e * 1024);
#pragma unroll
      for (int tp = 0; tp < 4; ++tp) {
        if (tp + 1 < 4) {
#pragma unroll
          for (int e = 0; e < 8; ++e) wn[e] = *reinterpret_cast<const ffn_bf16x8*>(sW + ((tp + 1) * 8 + e) * 1024);
        }
        ffn_f32x16& ya = y[2 * tp];
        ffn_f32x16& yb = y[2 * tp + 1];
        // tile a: w[0..3] = (ks0 hi, ks0 lo, ks1 hi, ks1 lo); tile b: w[4..7]
        ya = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], bh0, ya, 0, 0, 0);
        yb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[5], bh0, yb, 0, 0, 0);
        ya = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], bl0, ya, 0, 0, 0);
        yb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[4], bl0, yb, 0, 0, 0);
        ya = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[3], bh1, ya, 0, 0, 0);
        yb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[7], bh1, yb, 0, 0, 0);
        ya = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2], bl1, ya, 0, 0, 0);
        yb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[6], bl1, yb, 0, 0, 0);
        ya = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], bh0, ya, 0, 0, 0);
        yb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[4], bh0, yb, 0, 0, 0);
        ya = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2], bh1, ya, 0, 0, 0);
        yb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[6], bh1, yb, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = wn[e];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (h + 1 < kFfnHid / 32) OCC_FFN_COMMIT(slot2, a)            // chunk 2h+2
    __syncthreads();
    slot = slot2;
  }
#undef OCC_FFN_ISSUE
#undef OCC_FFN_COMMIT

  // ---- epilogue: y[t][i] = feature f = 32 t + 8 (i/4) + 4 g + i%4 of this lane's row ------------------------------
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int f = 32 * t + 8 * q4 + 4 * g;
      const float4 bb = *reinterpret_cast<const float4*>(b2 + f);
      const float4 rr = *reinterpret_cast<const float4*>(xr + f);          // identity = the block's input
      y[t][4 * q4 + 0] += bb.x + rr.x; y[t][4 * q4 + 1] += bb.y + rr.y;
      y[t][4 * q4 + 2] += bb.z + rr.z; y[t][4 * q4 + 3] += bb.w + rr.w;
      sum += (y[t][4 * q4 + 0] + y[t][4 * q4 + 1]) + (y[t][4 * q4 + 2] + y[t][4 * q4 + 3]);
    }
  float rstd = 1.f, mean = 0.f;
  if (ln_g != nullptr) {
    sum += __shfl_xor(sum, 32);
    mean = sum * (1.f / kFfnC);
    float var = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float d = y[t][i] - mean;
        var = fmaf(d, d, var);
      }
    var += __shfl_xor(var, 32);
    rstd = rsqrtf(var * (1.f / kFfnC) + ln_eps);
  }
  float* orow = out + row * ldo;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int f = 32 * t + 8 * q4 + 4 * g;
      float4 v = make_float4(y[t][4 * q4 + 0], y[t][4 * q4 + 1], y[t][4 * q4 + 2], y[t][4 * q4 + 3]);
      if (ln_g != nullptr) {
        const float4 gg = *reinterpret_cast<const float4*>(ln_g + f);
        const float4 be = *reinterpret_cast<const float4*>(ln_b + f);
        v.x = (v.x - mean) * rstd * gg.x + be.x; v.y = (v.y - mean) * rstd * gg.y + be.y;
        v.z = (v.z - mean) * rstd * gg.z + be.z; v.w = (v.w - mean) * rstd * gg.w + be.w;
      }
      if (row_live) *reinterpret_cast<float4*>(orow + f) = v;
    }
}

}  // namespace occ

extern "C" int occ_ffn_pack_weights_bf16x3(const float* w1, const float* w2, void* packed, int C, int hidden,
                                           void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(w1 && w2 && packed, "ffn_pack_weights: null pointer argument");
  if (C != kFfnC || hidden != kFfnHid) {
    set_error("ffn_pack_weights: fused FFN kernel exists for embed_dims=256, feedforward_channels=512 (got %d, %d)",
              C, hidden);
    return OCC_E_UNSUPPORTED;
  }
  const long n = (long)kFfnChunks * 8192;
  hipLaunchKernelGGL(ffn_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), w1, w2, reinterpret_cast<unsigned short*>(packed));
  OCC_CHECK_LAUNCH("ffn_pack_weights");
  return OCC_OK;
}

// out = LayerNorm(x + W2 . relu(W1 . x + b1) + b2); x (M, 256) rows of stride ldx floats, packed from
// occ_ffn_pack_weights_bf16x3 (1 MB), ln_gamma/ln_beta may both be NULL (no LayerNorm).
extern "C" int occ_ffn_fused_bf16x3_f32(const float* x, int64_t ldx, const void* packed, const float* b1,
                                        const float* b2, const float* ln_gamma, const float* ln_beta,
                                        float ln_eps, float* out, int64_t ldo, int M, int C, int hidden,
                                        void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(x && packed && b1 && b2 && out, "ffn_fused: null pointer argument");
  OCC_CHECK_ARG(M > 0, "ffn_fused: bad row count %d", M);
  OCC_CHECK_ARG((ln_gamma == nullptr) == (ln_beta == nullptr), "ffn_fused: ln_gamma and ln_beta go together");
  if (C != kFfnC || hidden != kFfnHid || ldx % 4 || ldo % 4 || ldx < C || ldo < C) {
    set_error("ffn_fused: no kernel for C=%d hidden=%d ldx=%ld ldo=%ld", C, hidden, (long)ldx, (long)ldo);
    return OCC_E_UNSUPPORTED;
  }
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_fused_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 3 * kFfnChunk);
    attr_done = true;
  }
  hipLaunchKernelGGL(ffn_fused_kernel, dim3((unsigned)((M + 127) / 128)), dim3(256), 3 * kFfnChunk,
                     reinterpret_cast<hipStream_t>(stream), x, (long)ldx, reinterpret_cast<const uint4*>(packed), b1,
                     b2, ln_gamma, ln_beta, ln_eps, out, (long)ldo, M);
  OCC_CHECK_LAUNCH("ffn_fused");
  return OCC_OK;
}
