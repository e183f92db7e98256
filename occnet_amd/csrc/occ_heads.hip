// Semantics + flow heads of the occupancy decoder as ONE kernel on the gfx950 f32 matrix cores.
//
// Replaces (reference: projects/mmdet3d_plugin/bevformer/modules/transformer_occ.py):
//   :132-141  predicter      = Linear(32,64) -> Softplus -> Linear(64,num_classes)
//             flow_predicter = Linear(32,64) -> ReLU     -> Linear(64,2)
//   :318-319  both applied to every voxel feature (bs, W, H, Z, 32)
// Five torch launches per MLP (2 GEMMs, activation, 2 bias adds) and a 164 MB hidden tensor become one
// pass: the hidden layer never leaves registers.
//
// Everything is computed TRANSPOSED so that no cross-lane shuffle is needed between the two layers:
//   H^T (128 x 32 voxels) = W1cat (128 x 32) . X^T      A = W1cat rows, B = X^T   (4 tiles of 32 rows)
//   O^T ( 32 x 32 voxels) = W2cat ( 32 x 128) . act(H^T) A = W2cat,     B = act(H^T)
// With v_mfma_f32_32x32x2_f32 a lane's D registers of H^T tile a hold hidden units
// u = 32a + (r&3) + 8(r>>2) + 4(lane>>5) of voxel (lane&31) — exactly a legal B operand (k = lane>>5
// picks between two hidden units, j = lane&31 is the voxel) when the k-pairs of the second
// contraction are enumerated as (a, r).  W1cat = [predicter.0 ; flow_predicter.0], W2cat is block
// diagonal: rows [0,ncls) read hidden [0,64) from predicter.2, rows ncls, ncls+1 read hidden [64,128)
// from flow_predicter.2.  Biases enter as one extra k-pair with B = 1.
// A wave keeps its W1 fragments in registers, W2cat fragments live in LDS (shared by the block), and
// loops over 32-voxel tiles; outputs go through a small LDS transpose so the global stores are
// contiguous (32 x ncls floats / 64 floats per tile).
#include "common.h"

namespace occ {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float softplus_f32(float x) {
  // torch.nn.Softplus(beta=1, threshold=20): x > 20 ? x : log1p(exp(x)), evaluated in the overflow-free
  // form max(x,0) + log(1 + exp(-|x|)) on the hardware exp/log units (absolute error ~1e-7), branch-free: above the
  // threshold 1 + exp(-x) rounds to 1 and the sum is x itself (see cvh_softplus in conv3d_mfma.hip)
  const float e = __builtin_amdgcn_exp2f(fabsf(x) * -1.44269504088896341f);
  return fmaxf(x, 0.f) + __builtin_amdgcn_logf(1.f + e) * 0.693147180559945309f;
}

constexpr int kHeadWaves = 4;

__global__ __launch_bounds__(256) void occ_heads_kernel(
    const float* __restrict__ feat, const float* __restrict__ w1o, const float* __restrict__ b1o,
    const float* __restrict__ w2o, const float* __restrict__ b2o, const float* __restrict__ w1f,
    const float* __restrict__ b1f, const float* __restrict__ w2f, const float* __restrict__ b2f,
    float* __restrict__ occ, float* __restrict__ flow, long long* __restrict__ occ_cls, long n_rows, int ncls) {
  constexpr int C = 32, HID = 64;
  __shared__ float w2s[65 * 64];                    // [step (a*16+r), bias step 64][lane]
  __shared__ float osm[kHeadWaves][32 * 33];        // per-wave output transpose
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vi = lane & 31, kh = lane >> 5;

  // ---- W2cat fragments -> LDS: A operand of layer 2, lane (i = output row, kh) at step (a, r) --------
  for (int e = tid; e < 65 * 64; e += 256) {
    const int step = e >> 6, l = e & 63, o = l & 31, k = l >> 5;
    float v = 0.f;
    if (step < 64) {
      const int a = step >> 4, r = step & 15;
      const int u = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * k;   // hidden unit in [0,128)
      if (o < ncls) { if (u < HID) v = w2o[o * HID + u]; }
      else if (o < ncls + 2) { if (u >= HID) v = w2f[(o - ncls) * HID + (u - HID)]; }
    } else if (k == 0) {                                       // bias k-pair (B = 1 on k = 0)
      if (o < ncls) v = b2o[o];
      else if (o < ncls + 2) v = b2f[o - ncls];
    }
    w2s[e] = v;
  }
  // ---- W1cat fragments -> registers: A operand of layer 1, lane (i = hidden row, kh), step s ------
  float w1r[4][16], b1r[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int u = 32 * a + vi;
    const float* src = (u < HID ? w1o + u * C : w1f + (u - HID) * C) + kh * 16;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const float4 w4 = *reinterpret_cast<const float4*>(src + s4 * 4);
      w1r[a][s4 * 4 + 0] = w4.x; w1r[a][s4 * 4 + 1] = w4.y;
      w1r[a][s4 * 4 + 2] = w4.z; w1r[a][s4 * 4 + 3] = w4.w;
    }
    b1r[a] = kh == 0 ? (u < HID ? b1o[u] : b1f[u - HID]) : 0.f;
  }
  __syncthreads();
  const float one = kh == 0 ? 1.f : 0.f;
  float* sm = osm[wave];

  const long n_tiles = (n_rows + 31) / 32;
  for (long tile = (long)blockIdx.x * kHeadWaves + wave; tile < n_tiles;
       tile += (long)gridDim.x * kHeadWaves) {
    const long row0 = tile * 32;
    // X^T fragment: lane (j = voxel, kh) holds channels kh*16 .. kh*16+15 of its voxel
    float xr[16];
    const long row = row0 + vi;
    if (row < n_rows) {
      const float* src = feat + row * C + kh * 16;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const float4 x4 = *reinterpret_cast<const float4*>(src + s4 * 4);
        xr[s4 * 4 + 0] = x4.x; xr[s4 * 4 + 1] = x4.y; xr[s4 * 4 + 2] = x4.z; xr[s4 * 4 + 3] = x4.w;
      }
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) xr[s] = 0.f;
    }
    f32x16 h[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
      for (int r = 0; r < 16; ++r) h[a][r] = 0.f;
      h[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1r[a], one, h[a], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int a = 0; a < 4; ++a)
        h[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1r[a][s], xr[s], h[a], 0, 0, 0);
    // activations: hidden [0,64) Softplus (predicter), [64,128) ReLU (flow_predicter)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      h[0][r] = softplus_f32(h[0][r]);
      h[1][r] = softplus_f32(h[1][r]);
      h[2][r] = fmaxf(h[2][r], 0.f);
      h[3][r] = fmaxf(h[3][r], 0.f);
    }
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    o = __builtin_amdgcn_mfma_f32_32x32x2f32(w2s[64 * 64 + lane], one, o, 0, 0, 0);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(w2s[(a * 16 + r) * 64 + lane], h[a][r], o, 0, 0, 0);
    // O^T (row = output channel, col = voxel) -> sm[voxel][channel]
#pragma unroll
    for (int r = 0; r < 16; ++r) sm[vi * 33 + (r & 3) + 8 * (r >> 2) + 4 * kh] = o[r];
    wave_lds_sync();
    const long valid = n_rows - row0 < 32 ? n_rows - row0 : 32;
    for (int e = lane; e < valid * ncls; e += 64) occ[row0 * ncls + e] = sm[(e / ncls) * 33 + e % ncls];
    if (lane < valid * 2) flow[row0 * 2 + lane] = sm[(lane >> 1) * 33 + ncls + (lane & 1)];
    // decode (reference bevformer_occ_head.py:210-212: softmax(-1).argmax(-1)): softmax is monotonic, so the class
    // is the argmax of the logits, first index on ties as torch.argmax resolves them
    if (occ_cls != nullptr && lane < valid) {
      float best = sm[lane * 33];
      int arg = 0;
      bool nan = best != best;
      for (int ch = 1; ch < ncls; ++ch) {
        const float x = sm[lane * 33 + ch];
        nan |= x != x;
        if (x > best) { best = x; arg = ch; }
      }
      // a NaN logit makes the reference's whole softmax row NaN, whose argmax torch resolves to index 0
      occ_cls[row0 + lane] = nan ? 0 : arg;
    }
    wave_lds_sync();
  }
}

// ---- bf16x3 variant (default) ------------------------------------------------------------------------------------
// The same two MLPs on the bf16 matrix cores with hi/lo-split operands (a.b ~= al.bh + ah.bl + ah.bh, f32
// accumulation, product error <= 2^-16 — linear_bf16x3.hip's arithmetic): 48 v_mfma_f32_32x32x16_bf16 per 32-voxel tile
// instead of 129 v_mfma_f32_32x32x2_f32 at half the issue rate, 5.4x less matrix-pipe time; the kernel drops from
// matrix-bound (MfmaUtil 0.53, 0.135 ms) towards the 131 MB it has to move.  Same transposed formulation: the D
// registers of H^T (hidden x voxels) are the B operand of the second contraction — for the bf16 instruction a lane's
// registers 0..7 / 8..15 of tile a are the two 16-k steps when k-slot j of lane-half g stands for hidden unit
// 32a + 16ks + 8*(j/4) + 4g + j%4, and the W2cat fragments are laid out in LDS with that order.
typedef __bf16 hx_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void hx_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  hi = pack_bf16x2_rne(x0, x1);
  lo = pack_bf16x2_rne(x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u));
}

__global__ __launch_bounds__(256, 2) void occ_heads_x3_kernel(
    const float* __restrict__ feat, const float* __restrict__ w1o, const float* __restrict__ b1o,
    const float* __restrict__ w2o, const float* __restrict__ b2o, const float* __restrict__ w1f,
    const float* __restrict__ b1f, const float* __restrict__ w2f, const float* __restrict__ b2f,
    float* __restrict__ occ, float* __restrict__ flow, long long* __restrict__ occ_cls, long n_rows, int ncls) {
  constexpr int C = 32, HID = 64;
  // W1cat fragments [(a*2 + s)*2 + plane][lane][8 bf16], W2cat fragments [(kk*2 + plane)][lane][8], biases, per-wave
  // output transpose
  __shared__ __attribute__((aligned(16))) unsigned short w1s[16 * 512];
  __shared__ __attribute__((aligned(16))) unsigned short w2s[16 * 512];
  __shared__ __attribute__((aligned(16))) float b1s[128];
  __shared__ __attribute__((aligned(16))) float b2s[32];
  __shared__ float osm[kHeadWaves][32 * 33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vi = lane & 31, g = lane >> 5;

  for (int e = tid; e < 4096; e += 256) {                    // one (hi, lo) pair per element
    const int j = e & 7, l = (e >> 3) & 63, f = e >> 9;      // f: 0..7
    const int m = l & 31, gg = l >> 5;
    {   // W1cat: f = a*2 + s
      const int a = f >> 1, sk = f & 1, u = 32 * a + m, k = 16 * sk + 8 * gg + j;
      const float w = u < HID ? w1o[u * C + k] : w1f[(u - HID) * C + k];
      const unsigned short hi = bf16_rne(w), lo = bf16_rne(w - __uint_as_float((unsigned)hi << 16));
      w1s[((f * 2 + 0) * 64 + l) * 8 + j] = hi;
      w1s[((f * 2 + 1) * 64 + l) * 8 + j] = lo;
    }
    {   // W2cat: f = kk = 2a + ks; output row m, hidden unit u
      const int a = f >> 1, ks = f & 1, u = 32 * a + 16 * ks + 8 * (j >> 2) + 4 * gg + (j & 3);
      float w = 0.f;
      if (m < ncls) { if (u < HID) w = w2o[m * HID + u]; }
      else if (m < ncls + 2) { if (u >= HID) w = w2f[(m - ncls) * HID + (u - HID)]; }
      const unsigned short hi = bf16_rne(w), lo = bf16_rne(w - __uint_as_float((unsigned)hi << 16));
      w2s[((f * 2 + 0) * 64 + l) * 8 + j] = hi;
      w2s[((f * 2 + 1) * 64 + l) * 8 + j] = lo;
    }
  }
  if (tid < 128) b1s[tid] = tid < HID ? b1o[tid] : b1f[tid - HID];
  if (tid < 32) b2s[tid] = tid < ncls ? b2o[tid] : (tid < ncls + 2 ? b2f[tid - ncls] : 0.f);
  __syncthreads();
  float* sm = osm[wave];
  const hx_bf16x8* W1 = reinterpret_cast<const hx_bf16x8*>(w1s) + lane;
  const hx_bf16x8* W2 = reinterpret_cast<const hx_bf16x8*>(w2s) + lane;

  const long n_tiles = (n_rows + 31) / 32;
  for (long tile = (long)blockIdx.x * kHeadWaves + wave; tile < n_tiles; tile += (long)gridDim.x * kHeadWaves) {
    const long row0 = tile * 32;
    // X^T fragments: lane (voxel vi, half g) holds channels 16 s + 8 g .. + 7 of its voxel, hi + lo
    uint4 xh[2], xl[2];
    {
      long row = row0 + vi;
      if (row >= n_rows) row = n_rows - 1;
      const float* src = feat + row * C + 8 * g;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const float4 p = *reinterpret_cast<const float4*>(src + 16 * s);
        const float4 q = *reinterpret_cast<const float4*>(src + 16 * s + 4);
        hx_split2(p.x, p.y, xh[s].x, xl[s].x); hx_split2(p.z, p.w, xh[s].y, xl[s].y);
        hx_split2(q.x, q.y, xh[s].z, xl[s].z); hx_split2(q.z, q.w, xh[s].w, xl[s].w);
      }
    }
    // ---- H^T = W1cat . X^T : 4 hidden tiles, the MFMAs of different tiles interleaved (no back-to-back MFMAs on one
    // accumulator)
    f32x16 h[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) h[a][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const hx_bf16x8 bxh = __builtin_bit_cast(hx_bf16x8, xh[s]), bxl = __builtin_bit_cast(hx_bf16x8, xl[s]);
#pragma unroll
      for (int a = 0; a < 4; ++a)
        h[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W1[((a * 2 + s) * 2 + 1) * 64], bxh, h[a], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 4; ++a)
        h[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W1[((a * 2 + s) * 2 + 0) * 64], bxl, h[a], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 4; ++a)
        h[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W1[((a * 2 + s) * 2 + 0) * 64], bxh, h[a], 0, 0, 0);
    }
    // ---- bias + activation (hidden [0,64): Softplus, [64,128): ReLU), then O^T = W2cat . act(H^T); one accumulator
    // per bf16x3 term
    f32x16 o0, o1, o2;
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = o1[r] = o2[r] = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      float v[16];
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const float4 bb = *reinterpret_cast<const float4*>(b1s + 32 * a + 8 * q4 + 4 * g);
        const float t0 = h[a][4 * q4 + 0] + bb.x, t1 = h[a][4 * q4 + 1] + bb.y;
        const float t2 = h[a][4 * q4 + 2] + bb.z, t3 = h[a][4 * q4 + 3] + bb.w;
        if (a < 2) {
          v[4 * q4 + 0] = softplus_f32(t0); v[4 * q4 + 1] = softplus_f32(t1);
          v[4 * q4 + 2] = softplus_f32(t2); v[4 * q4 + 3] = softplus_f32(t3);
        } else {
          v[4 * q4 + 0] = fmaxf(t0, 0.f); v[4 * q4 + 1] = fmaxf(t1, 0.f);
          v[4 * q4 + 2] = fmaxf(t2, 0.f); v[4 * q4 + 3] = fmaxf(t3, 0.f);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        uint4 hh, hl;
        hx_split2(v[8 * ks + 0], v[8 * ks + 1], hh.x, hl.x); hx_split2(v[8 * ks + 2], v[8 * ks + 3], hh.y, hl.y);
        hx_split2(v[8 * ks + 4], v[8 * ks + 5], hh.z, hl.z); hx_split2(v[8 * ks + 6], v[8 * ks + 7], hh.w, hl.w);
        const int kk = 2 * a + ks;
        const hx_bf16x8 wh = W2[(kk * 2 + 0) * 64], wl = W2[(kk * 2 + 1) * 64];
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, __builtin_bit_cast(hx_bf16x8, hh), o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, __builtin_bit_cast(hx_bf16x8, hl), o1, 0, 0, 0);
        o2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, __builtin_bit_cast(hx_bf16x8, hh), o2, 0, 0, 0);
      }
    }
    // O^T (row = output channel, col = voxel) + bias -> sm[voxel][channel]
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const float4 bb = *reinterpret_cast<const float4*>(b2s + 8 * q4 + 4 * g);
      const float bq[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * q4 + i;
        sm[vi * 33 + 8 * q4 + 4 * g + i] = ((o0[r] + o1[r]) + o2[r]) + bq[i];
      }
    }
    wave_lds_sync();
    const long valid = n_rows - row0 < 32 ? n_rows - row0 : 32;
    for (int e = lane; e < valid * ncls; e += 64) occ[row0 * ncls + e] = sm[(e / ncls) * 33 + e % ncls];
    if (lane < valid * 2) flow[row0 * 2 + lane] = sm[(lane >> 1) * 33 + ncls + (lane & 1)];
    if (occ_cls != nullptr && lane < valid) {     // decode: argmax of the logits, first index on ties (= torch)
      float best = sm[lane * 33];
      int arg = 0;
      bool nan = best != best;
      for (int ch = 1; ch < ncls; ++ch) {
        const float x = sm[lane * 33 + ch];
        nan |= x != x;
        if (x > best) { best = x; arg = ch; }
      }
      // a NaN logit makes the reference's whole softmax row NaN, whose argmax torch resolves to index 0
      occ_cls[row0 + lane] = nan ? 0 : arg;
    }
    wave_lds_sync();
  }
}

}  // namespace occ

extern "C" int occ_occ_heads_decode_f32(const float* feat, const float* w1_occ, const float* b1_occ,
                                        const float* w2_occ, const float* b2_occ, const float* w1_flow,
                                        const float* b1_flow, const float* w2_flow, const float* b2_flow,
                                        float* occ_out, float* flow_out, int64_t* occ_cls_out, int64_t n_rows,
                                        int C, int hidden, int num_classes, int exact_f32, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(feat && w1_occ && b1_occ && w2_occ && b2_occ && w1_flow && b1_flow && w2_flow &&
                    b2_flow && occ_out && flow_out,
                "occ_heads: null pointer argument");
  OCC_CHECK_ARG(n_rows > 0 && num_classes > 0, "occ_heads: bad dimension");
  if (C != 32 || hidden != 64 || num_classes + 2 > 32) {
    set_error("occ_heads: no fused kernel for C=%d hidden=%d num_classes=%d", C, hidden, num_classes);
    return OCC_E_UNSUPPORTED;
  }
  const long n_tiles = (n_rows + 31) / 32;
  long blocks = (n_tiles + kHeadWaves - 1) / kHeadWaves;
  if (blocks > 256 * 4) blocks = 256 * 4;   // persistent waves: W fragments are loaded once per wave
  // exact_f32: v_mfma_f32_32x32x2_f32 kernel; else bf16x3 (hi/lo-split operands on the bf16 MFMA, f32 accumulation)
  if (exact_f32)
    hipLaunchKernelGGL(occ_heads_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), feat, w1_occ, b1_occ, w2_occ, b2_occ,
                       w1_flow, b1_flow, w2_flow, b2_flow, occ_out, flow_out,
                       reinterpret_cast<long long*>(occ_cls_out), (long)n_rows, num_classes);
  else
    hipLaunchKernelGGL(occ_heads_x3_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), feat, w1_occ, b1_occ, w2_occ, b2_occ,
                       w1_flow, b1_flow, w2_flow, b2_flow, occ_out, flow_out,
                       reinterpret_cast<long long*>(occ_cls_out), (long)n_rows, num_classes);
  OCC_CHECK_LAUNCH("occ_heads");
  return OCC_OK;
}

extern "C" int occ_occ_heads_f32(const float* feat, const float* w1_occ, const float* b1_occ,
                                 const float* w2_occ, const float* b2_occ, const float* w1_flow,
                                 const float* b1_flow, const float* w2_flow, const float* b2_flow,
                                 float* occ_out, float* flow_out, int64_t n_rows, int C, int hidden,
                                 int num_classes, void* stream) {
  return occ_occ_heads_decode_f32(feat, w1_occ, b1_occ, w2_occ, b2_occ, w1_flow, b1_flow, w2_flow, b2_flow,
                                  occ_out, flow_out, nullptr, n_rows, C, hidden, num_classes, 1, stream);
}
