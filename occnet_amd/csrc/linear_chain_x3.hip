// Row-local Linear CHAINS of a BEVFormerLayer in one launch each, on the gfx950 bf16 matrix cores ("bf16x3", the
// arithmetic of linear_bf16x3.hip: hi/lo bf16 split of both operands, three MFMAs per product, f32 accumulation).
//
// Every dense op between two gathers of the encoder acts on the same rows (reference: encoder.py:377-404,
// spatial_cross_attention.py:173-175,334-341, temporal_self_attention.py:197-209,266-272, mmcv FFN).  As seven separate
// launches per layer they move 882 MB per layer and run as one partial wave of blocks each, so load, MFMA and store
// phases add up instead of overlapping (profiles/r03_linear_ablation.txt).  Two programs replace six of the seven:
//
//   program A (after the TSA gather):   y = LN(a.Wo^T + bo + res)                            -> x1
//                                       z = y.Wq^T + bq        (sampling_offsets | attention_weights of the SCA, N = 768)
//   program B (after the SCA gather):   x2 = LN(a.Wo^T + bo + res)
//                                       h  = relu(x2.W1^T + b1)                              (512 hidden columns)
//                                       y  = LN(h.W2^T + b2 + x2)                            -> x3, the layer output
//                                       z  = [y.Wsum^T + term | y.Wv^T + bv]                 (optional: the NEXT layer's TSA
//                                            query Linears with the positional term folded in, and its value projection)
//
//   program C (layer 0, no producer):  z  = [a.Wsum^T + term | a.Wv^T + bv]                  (the tail stage alone: the first
//                                            layer's TSA query Linears and value projection straight from the BEV queries)
//
// A block owns 64 rows for ALL columns of every stage.  The stage input is a 64 x 256 tile in LDS as hi and lo bf16
// planes (64 KB, 16-byte pieces XOR-swizzled by row: conflict-free ds_read_b128 of the MFMA fragments); the four waves
// each own 64 of the 256 output columns of a pass (2 x 2 accumulator tiles) and stream their hi/lo weight fragments
// from L2 through a 4-deep register ring that runs ahead ACROSS stage boundaries (the weights of all stages are one
// buffer in consumption order, 16 KB per k-step), so a stage's first MFMA never waits for a weight.  The MFMAs are
// issued transposed (weights as the row operand): a lane ends up with 4 consecutive columns of one row per register
// quad, which is at once (i) the float4 of the row-major global store, (ii) the 8-byte half of a 16-byte LDS piece of
// the next stage's operand tile and (iii) a layout in which bias / residual are accumulator INITIAL VALUES and the
// LayerNorm statistics are a lane-local sum + one shuffle + a 4-wave exchange through 2 KB of LDS.  The LayerNorm'd
// tile never leaves the CU between the Linears; the FFN's hidden activation exists only as accumulator registers and
// as the LDS tile (half of it at a time).  x2, the FFN's residual, is parked in the block's own rows of `y` (written
// and re-read by the same lane, overwritten by x3 at the end).  Two blocks per CU (66 KB of LDS, <= 256 registers):
// one block's epilogues run under the other's MFMAs.
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include "common.h"

namespace occ {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kChRows = 64;                       // rows per block
constexpr int kChPlane = kChRows * 512;           // one bf16 plane of the operand tile: 64 rows x 256 k
constexpr int kChRed = 2 * kChPlane;              // LayerNorm exchange: float[2][4][64]
constexpr int kChPrm = kChRed + 2 * 4 * kChRows * 4;       // per-column parameters: float[kChBiasMax] biases | g1 | b1 | g2 | b2
constexpr bool kChSpread = true;                  // program B: x2 / x3 row stores spread under the next stage's k-loop (false = one
                                                  // burst after the LayerNorm: 127 vs 122-125 us, profiles/r04_c32_chain_probe_no_spills.txt) (false: burst after the LayerNorm; 127 vs 122-125 us, profiles/r04_c32_chain_probe_no_spills.txt)
constexpr int kChBiasMax = 1536;                  // program B with its tail: 256 + 512 + 256 + 2 x 256
constexpr int kChLds = kChPrm + (kChBiasMax + 4 * 256) * 4;       // 77 824 B: two blocks per CU
constexpr int kChStepBytes = 16384;               // weights of one k-step (16 k) of one 256-column pass: 8 tiles x (hi, lo) x 1 KB

struct ChainArgs {
  const float* a; long lda;                       // stage input rows (M, 256)
  const float* res; long ldres;                   // residual of the first LayerNorm (M, 256)
  const uint4* wp; unsigned wbytes;               // chain weights in consumption order
  const float* bias;                              // chain biases, 256 per pass (layout: see the launchers)
  const float* ln1_g; const float* ln1_b; float eps1;
  const float* ln2_g; const float* ln2_b; float eps2;
  float* y; long ldy;                             // LayerNorm output rows (A: x1, B: x3 — and B's x2 parking space)
  int npass;                                      // 256-column passes of the tail stage
  int act;                                        // 1: ReLU on the tail outputs
  const float* term; long ldterm; int term_cols;  // added to tail columns < term_cols (or NULL)
  float* z1; long ldz1; int n1;                   // tail columns [0, n1) -> z1
  float* z2; long ldz2; int off2; int n2;         // tail columns [off2, off2 + n2) -> z2 (column - off2)
  int M;
  int nfull;                                      // blocks with 64-row tiles (the rest: 32-row tiles)
  long long* trace;                               // TRACE builds: 24 wall-clock stamps per wave (development)
};

// Block barrier that orders LDS traffic only.  __syncthreads() is a full fence: hipcc puts `s_waitcnt vmcnt(0)` in front of
// the s_barrier, so every barrier after a row-store phase waited for the stores' L2 acknowledgements and for the weight
// ring's look-ahead (round 4 ISA reading).  Global memory is never shared between the threads of a block here (the one
// re-read of a block's own stores, x2, is by the lane that wrote it and sits behind an explicit vmcnt(0)).
__device__ __forceinline__ void ch_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void ch_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  hi = pack_bf16x2_rne(x0, x1);
  lo = pack_bf16x2_rne(x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u));
}

// (N rows of a (N, K) f32 weight) -> chain order: for every 256-row group g: packed[g][K/16][8 tiles][hi | lo][lane][8 bf16]
// (rows beyond N are zero), i.e. 16 KB per k-step: exactly what one flat ring step of a pass fetches
__global__ void linear_chain_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ packed, long n_elem,
                                         int K, int N) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;     // one (hi, lo) pair per thread
  if (idx >= n_elem) return;
  const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
  const long rest = idx >> 9;                                       // (group, k-step, tile)
  const int tile = (int)(rest & 7);
  const long gk = rest >> 3;
  const int KS = K / 16;
  const int ks = (int)(gk % KS), grp = (int)(gk / KS);
  const int n = grp * 256 + tile * 32 + (lane & 31), k = ks * 16 + (lane >> 5) * 8 + j;
  unsigned short hi = 0, lo = 0;
  if (n < N) {
    const float x = w[(long)n * K + k];
    hi = bf16_rne(x);
    lo = bf16_rne(x - __uint_as_float((unsigned)hi << 16));
  }
  unsigned short* dst = packed + ((rest * 2) * 64 + lane) * 8 + j;
  dst[0] = hi;
  dst[64 * 8] = lo;
}

// ---- building blocks (all indices compile-time: the arrays stay in registers) ------------------------------------------

#define OCC_CH_LOAD(SLOT, STEP)                                                                    \
  {                                                                                                \
    const int so = (STEP) * kChStepBytes;                                                          \
    w[SLOT][0] = __builtin_amdgcn_raw_buffer_load_b128(wr, wv, so, 0);                             \
    w[SLOT][1] = __builtin_amdgcn_raw_buffer_load_b128(wr, wv, so + 1024, 0);   /* (the tile offsets ride in the   */ \
    w[SLOT][2] = __builtin_amdgcn_raw_buffer_load_b128(wr, wv, so + 2048, 0);   /*  scalar offset: one address     */ \
    w[SLOT][3] = __builtin_amdgcn_raw_buffer_load_b128(wr, wv, so + 3072, 0);   /*  VGPR instead of four)          */ \
  }

// operand fragments of (physical) k-step KK — wave-uniform, run time: piece 2 KK + kb of rows vi and (RT == 2) 32 + vi, both planes
// (slot of piece p in row r = p ^ (r & 31))
#define OCC_CH_AFRAG(BUF, KK)                                                                      \
  {                                                                                                \
    /* the address is recomputed at every use from an opaque copy of the base: hipcc otherwise hoists the 16 per-step       \
       addresses out of all seven k-loops of program B and keeps them live for the whole kernel (16 VGPRs, spills) */       \
    unsigned ab_ = abase;                                                                          \
    asm volatile("" : "+v"(ab_));                                                                  \
    const char* ap = tl + (ab_ ^ (unsigned)((KK) * 32));                                           \
    /* lo planes first: the step's first MFMAs (small term wh . al) read them */                   \
    af[BUF][0][1] = *reinterpret_cast<const bf16x8*>(ap + kChPlane);                               \
    if constexpr (RT == 2) af[BUF][1][1] = *reinterpret_cast<const bf16x8*>(ap + kChPlane + 32 * 512);     \
    af[BUF][0][0] = *reinterpret_cast<const bf16x8*>(ap);                                          \
    if constexpr (RT == 2) af[BUF][1][0] = *reinterpret_cast<const bf16x8*>(ap + 32 * 512);        \
  }

// one 256-column pass over the K = 256 tile: 16 k-steps, flat ring steps step0 .. step0 + 15 (requests run 3 ahead).
// Every block walks the 16 k-steps in an order ROTATED by `rot` (a GEMM's k order is free): the blocks of an XCD start
// together and run the same program, so without it all 64 of them request the same 16 KB of weights at the same time —
// 4 096 line requests into the few L2 channels that hold that chunk while the others idle (round 4, call 2: one block
// per CU alone took ~1 000 clocks per k-step, two took twice that: the L2 request rate of a hot channel, not MFMA, not
// L1 bandwidth).  With the rotation the resident blocks are spread over all 16 chunks of a pass at any time.
// ST: one quad of `sacc` (an already finished 64 x 64 register tile: the LayerNorm'd rows) is stored to `sdst` per k-step, so
// that its 16 row stores trickle out under the MFMAs of the NEXT stage instead of leaving as one burst that the wave has to
// sit through (stamped timeline: issuing 16 stores took 5 us, p90 14 us, profiles/r04_c21_chain_trace_waves.txt).
template <int ABL = 0, bool ST = false, int RT>
__device__ __forceinline__ void ch_kloop(f32x16 (&acc)[RT][2], occ_u32x4 (&w)[4][4], const __amdgpu_buffer_rsrc_t wr,
                                         const int wv, const int step0, const int next0, const char* tl, const unsigned abase,
                                         const int rot, const f32x16 (&sacc)[RT][2], const __amdgpu_buffer_rsrc_t sdst,
                                         const unsigned (&soff)[RT]) {   // soff: byte offset of the lane's first quad per row tile
  bf16x8 af[2][RT][2];                              // [buffer][row tile][plane hi, lo]
  // the pinned group sequence must see the loop's own instructions only: the DS reads of an accumulator initialisation in
  // front of it (or the LayerNorm exchange behind it) in the same scheduling region are matched into the DS groups and
  // slide every fragment read two steps late; at the loop's end the last ring requests sank to their uses
  if (ABL == 0) __builtin_amdgcn_sched_barrier(0);
  OCC_CH_AFRAG(0, rot & 15)
  // the group sequence below is matched to instructions in program order: without this leading group the four reads
  // above fill step 0's fragment groups and EVERY step's reads slide one step late — issued right before their use
  // (that is what the first cuts of this kernel did: rocprofv3 round 4 call 2, 56 % of the wave cycles issue-stalled)
  if (ABL == 0) __builtin_amdgcn_sched_group_barrier(0x100, 2 * RT, 0);
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    if (!(ABL & 4) || ks == 0) OCC_CH_AFRAG((ks + 1) & 1, (ks + 1 + rot) & 15)
    // logical step ks + 3 of this pass, or step ks + 3 - 16 of the k loop that follows (first step next0: the ring runs
    // across stage — and, in the persistent kernel, tile — boundaries)
    if (!(ABL & 1)) OCC_CH_LOAD((ks + 3) & 3, (ks + 3 < 16 ? step0 : next0) + ((ks + 3 + rot) & 15))
    if constexpr (ST) {
      if (ks < 8 * RT) {                            // quad (rt, t, q) = (ks >> 3, (ks >> 2) & 1, ks & 3)
        const int rt = ks >> 3, t = (ks >> 2) & 1, q = ks & 3;
        const float4 v = make_float4(sacc[rt][t][4 * q + 0], sacc[rt][t][4 * q + 1], sacc[rt][t][4 * q + 2], sacc[rt][t][4 * q + 3]);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(occ_u32x4, v), sdst,
                                               (int)(soff[rt] + (unsigned)(32 * t + 8 * q) * 4u), 0, 0);
      }
    }
    // D[column][row] (weights as the row operand); small terms first, term-major over the four accumulators
    if (!(ABL & 2)) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        acc[rt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[ks & 3][2 * t]), af[ks & 1][rt][1],
                                                             acc[rt][t], 0, 0, 0);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        acc[rt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[ks & 3][2 * t + 1]), af[ks & 1][rt][0],
                                                             acc[rt][t], 0, 0, 0);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        acc[rt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[ks & 3][2 * t]), af[ks & 1][rt][0],
                                                             acc[rt][t], 0, 0, 0);
    } else {
      // ablation: keep the operands alive without the matrix pipe
#pragma unroll
      for (int t = 0; t < 4; ++t) asm volatile("" :: "v"(w[ks & 3][t]));
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) { asm volatile("" :: "v"(af[ks & 1][rt][0])); asm volatile("" :: "v"(af[ks & 1][rt][1])); }
    }
    if (ABL == 0) {
    // pin the software pipeline (hipcc otherwise sinks every ring request down to its use: load, vmcnt(0), MFMA).  The
    // operand fragments of step s + 1 are requested at the TOP of step s, lo planes first: the next step opens with the
    // MFMAs that read them
    if constexpr (RT == 2) {
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // operand fragments of step s + 1 (lo planes)
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
    if constexpr (ST) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);   // the step's row store
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // (hi planes)
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);   // ring requests of step s + 3
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
    } else {                                             // 32-row tiles: six MFMAs per step
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    if constexpr (ST) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    }
    }
  }
  if (ABL == 0) __builtin_amdgcn_sched_barrier(0);
}

// Per-column parameters (biases, LayerNorm gamma / beta) are staged ONCE per block in LDS (`prm`): as global loads in
// front of every pass they sat behind the previous pass's row stores in the in-order vmcnt queue — every pass opened
// with a full store round trip.  Register 4 q + i of tile (rt, t) = row rt * 32 + vi, column c0 + 32 t + 8 q + 4 kb + i.
template <int RT>
__device__ __forceinline__ void ch_set_bias(f32x16 (&acc)[RT][2], const float* prm_bias, int kb) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 b = *reinterpret_cast<const float4*>(prm_bias + 32 * t + 8 * q + 4 * kb);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        acc[rt][t][4 * q + 0] = b.x; acc[rt][t][4 * q + 1] = b.y; acc[rt][t][4 * q + 2] = b.z; acc[rt][t][4 * q + 3] = b.w;
      }
    }
}

template <int RT>
__device__ __forceinline__ void ch_add_bias(f32x16 (&acc)[RT][2], const float* prm_bias, int kb) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 b = *reinterpret_cast<const float4*>(prm_bias + 32 * t + 8 * q + 4 * kb);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        acc[rt][t][4 * q + 0] += b.x; acc[rt][t][4 * q + 1] += b.y; acc[rt][t][4 * q + 2] += b.z; acc[rt][t][4 * q + 3] += b.w;
      }
    }
}

// Row I/O goes through buffer instructions: an SGPR resource per matrix (base, M x ld x 4 bytes), a 32-bit byte offset per
// lane.  Rows beyond M fall outside the resource — loads return 0, stores are dropped by the bounds check — so there are no
// clamped indices, no `live` predicates and no 64-bit address arithmetic in the vector registers (program B spilled for them).
// accumulators <- src[row][column]: 16 quad loads, all requested before the first use
template <int RT>
__device__ __forceinline__ void ch_load_rows(f32x16 (&acc)[RT][2], const __amdgpu_buffer_rsrc_t src, const unsigned ldb,
                                             const unsigned (&row)[RT], int c0, int kb) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const unsigned ro = row[rt] * ldb + (unsigned)(c0 + 4 * kb) * 4u;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = buf_load16(src, ro + (unsigned)(32 * t + 8 * q) * 4u);
        acc[rt][t][4 * q + 0] = v.x; acc[rt][t][4 * q + 1] = v.y; acc[rt][t][4 * q + 2] = v.z; acc[rt][t][4 * q + 3] = v.w;
      }
  }
}

// register quads -> the operand tile (hi / lo planes): piece 8 wave + 4 t + q of row rt * 32 + vi, half kb
template <int RT>
__device__ __forceinline__ void ch_to_tile(const f32x16 (&acc)[RT][2], char* tl, int wave, int vi, int kb) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        unsigned h01, h23, l01, l23;
        ch_split2(acc[rt][t][4 * q + 0], acc[rt][t][4 * q + 1], h01, l01);
        ch_split2(acc[rt][t][4 * q + 2], acc[rt][t][4 * q + 3], h23, l23);
        char* p = tl + (rt * 32 + vi) * 512 + (((8 * wave + 4 * t + q) ^ vi) & 31) * 16 + kb * 8;
        *reinterpret_cast<uint2*>(p) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(p + kChPlane) = make_uint2(l01, l23);
      }
}

// LayerNorm over the 256 columns of every row, in place in the accumulators (two passes: mean, then squared deviations)
template <int RT>
__device__ __forceinline__ void ch_layernorm(f32x16 (&acc)[RT][2], float* red, const float* __restrict__ g,
                                             const float* __restrict__ b, float eps, int wave, int vi, int kb) {
  float s[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) v += acc[rt][t][r];
    v += __shfl_xor(v, 32);
    s[rt] = v;
    if (kb == 0) red[wave * kChRows + rt * 32 + vi] = v;
  }
  ch_sync();          // also: every wave is past its k loop, the operand tile may be rewritten after this point
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int r = rt * 32 + vi;
    const float mean = ((red[r] + red[kChRows + r]) + (red[2 * kChRows + r] + red[3 * kChRows + r])) * (1.f / 256.f);
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const float d = acc[rt][t][k] - mean;
        acc[rt][t][k] = d;
        v = fmaf(d, d, v);
      }
    v += __shfl_xor(v, 32);
    if (kb == 0) red[4 * kChRows + wave * kChRows + r] = v;
  }
  ch_sync();
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int r = rt * 32 + vi;
    const float* rr = red + 4 * kChRows;
    s[rt] = rsqrtf(((rr[r] + rr[kChRows + r]) + (rr[2 * kChRows + r] + rr[3 * kChRows + r])) * (1.f / 256.f) + eps);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = wave * 64 + 32 * t + 8 * q + 4 * kb;
      const float4 gv = *reinterpret_cast<const float4*>(g + c);
      const float4 bv = *reinterpret_cast<const float4*>(b + c);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        acc[rt][t][4 * q + 0] = fmaf(acc[rt][t][4 * q + 0] * s[rt], gv.x, bv.x);
        acc[rt][t][4 * q + 1] = fmaf(acc[rt][t][4 * q + 1] * s[rt], gv.y, bv.y);
        acc[rt][t][4 * q + 2] = fmaf(acc[rt][t][4 * q + 2] * s[rt], gv.z, bv.z);
        acc[rt][t][4 * q + 3] = fmaf(acc[rt][t][4 * q + 3] * s[rt], gv.w, bv.w);
      }
    }
}

// register quads -> row-major rows of `dst` (column c0 + 32 t + 8 q + 4 kb of row row[rt]); rows beyond M are dropped by the
// resource's bounds check.  One uniform branch per column tile.
template <int RT>
__device__ __forceinline__ void ch_store(const f32x16 (&acc)[RT][2], const __amdgpu_buffer_rsrc_t dst, const unsigned ldb,
                                         const unsigned (&row)[RT], int c0, int kb, bool t0_on, bool t1_on) {
  // c0 = column of `dst` that receives the wave's first column
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const unsigned ro = row[rt] * ldb + (unsigned)(c0 + 4 * kb) * 4u;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (!(t == 0 ? t0_on : t1_on)) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = make_float4(acc[rt][t][4 * q + 0], acc[rt][t][4 * q + 1], acc[rt][t][4 * q + 2], acc[rt][t][4 * q + 3]);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(occ_u32x4, v), dst, (int)(ro + (unsigned)(32 * t + 8 * q) * 4u), 0, 0);
      }
    }
  }
}

template <int RT>
__device__ __forceinline__ void ch_relu(f32x16 (&acc)[RT][2]) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rt][t][r] = fmaxf(acc[rt][t][r], 0.f);
}

// PROG 0: program A, PROG 1: program B, PROG 2: program C (file header)
#define OCC_CH_STAMP(I)                                                                            \
  if constexpr (TRACE) {                                                                           \
    if ((threadIdx.x & 63) == 0) p.trace[((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 24 + (I)] = wall_clock64();   \
  }

// One tile of RT x 32 rows starting at row m0 through the whole program.
template <int PROG, int ABL, bool TRACE, int RT>
__device__ __forceinline__ void chain_tile(const ChainArgs& p, char* tl, const long m0) {
  float* red = reinterpret_cast<float*>(tl + kChRed);
  float* prm = reinterpret_cast<float*>(tl + kChPrm);
  const float* prm_ln = prm + kChBiasMax;           // g1 | b1 | g2 | b2
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vi = lane & 31, kb = lane >> 5;
  const int M = p.M;
  OCC_CH_STAMP(0)

  // (De-phasing the two blocks of a CU — the one in the upper half of the CU's LDS, HW_REG_LDS_ALLOC base != 0, starting
  // 1.7 - 14 us late — was measured and changes nothing: profiles/r04_c16_stagger.txt.  The kernel is bound by the row
  // traffic itself, which is write-dominated; stock copy kernels with this read : write mix reach 3.5 TB/s on this part.)
  // ---- everything the first stage needs is requested before the first wait (round 4 ISA reading: hipcc had sunk the
  // tile's 16 row loads to their uses, 3-4 in flight, and put every residual load behind its own branch: five to six
  // serial HBM round trips per tile instead of one) -----------------------------------------------------------------------
  // per-column parameters (L2 hits): thread t fetches bias quads t and t + 256 and one quad of a LayerNorm vector
  const int nb4 = ((PROG == 0 ? 256 : PROG == 1 ? 1024 : 0) + 256 * p.npass) / 4;
  const float4 pb0 = reinterpret_cast<const float4*>(p.bias)[tid < nb4 ? tid : nb4 - 1];
  const float4 pb1 = reinterpret_cast<const float4*>(p.bias)[tid + 256 < nb4 ? tid + 256 : nb4 - 1];
  const float* lnv = wave == 0 ? p.ln1_g : wave == 1 ? p.ln1_b : wave == 2 ? p.ln2_g : p.ln2_b;
  if (PROG == 0 && wave >= 2) lnv = p.ln1_g;        // program A has one LayerNorm: waves 2, 3 fetch (and drop) a duplicate
  if (PROG == 2) lnv = p.bias;                      // program C has none
  const float4 pl = reinterpret_cast<const float4*>(lnv)[lane];

  // weight ring: slot (step & 3) = {hi tile 0, lo tile 0, hi tile 1, lo tile 1} of the wave's 64 columns of flat step
  occ_u32x4 w[4][4];
  const __amdgpu_buffer_rsrc_t wr = uniform_rsrc(p.wp, p.wbytes);
  const int wv = (wave * 4096) + lane * 16;
  // k-step rotation of this block: blocks b, b + 8, b + 16, .. share an XCD
  const int rot = (int)(blockIdx.x >> 3) & 15;
  OCC_CH_LOAD(0, rot)
  OCC_CH_LOAD(1, (rot + 1) & 15)
  OCC_CH_LOAD(2, (rot + 2) & 15)

  // this lane's rows, and the matrices as buffer resources of M rows (see ch_load_rows)
  unsigned row[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) row[rt] = (unsigned)m0 + (unsigned)(rt * 32 + vi);
  const unsigned lda_b = (unsigned)p.lda * 4u, ldres_b = (unsigned)p.ldres * 4u, ldy_b = (unsigned)p.ldy * 4u;
  const unsigned ldt_b = (unsigned)p.ldterm * 4u, ldz1_b = (unsigned)p.ldz1 * 4u, ldz2_b = (unsigned)p.ldz2 * 4u;
  const __amdgpu_buffer_rsrc_t ra = uniform_rsrc(p.a, (unsigned)M * lda_b);
  const __amdgpu_buffer_rsrc_t rres = uniform_rsrc(p.res, (unsigned)M * ldres_b);
  const __amdgpu_buffer_rsrc_t ry = uniform_rsrc(p.y, (unsigned)M * ldy_b);
  const __amdgpu_buffer_rsrc_t rterm = uniform_rsrc(p.term, (unsigned)M * ldt_b);
  const __amdgpu_buffer_rsrc_t rz1 = uniform_rsrc(p.z1, (unsigned)M * ldz1_b);
  const __amdgpu_buffer_rsrc_t rz2 = uniform_rsrc(p.z2, (unsigned)M * ldz2_b);

  // stage input: RT x 32 rows x 256 f32 (a wave instruction = one whole row, 1 KB), and the residual rows straight into the
  // accumulators
  f32x16 acc[RT][2];
  float4 v[8 * RT];
#pragma unroll
  for (int j = 0; j < 8 * RT; ++j)                   // (the row offset stays in the VGPR offset: the bounds check covers it)
    v[j] = buf_load16(ra, ((unsigned)m0 + (unsigned)(j * 4 + wave)) * lda_b + (unsigned)lane * 16u);
  if constexpr (PROG != 2) ch_load_rows(acc, rres, ldres_b, row, wave * 64, kb);
  __builtin_amdgcn_sched_barrier(0);                // nothing below may be hoisted between the requests above

  reinterpret_cast<float4*>(prm)[tid] = pb0;
  if (tid + 256 < nb4) reinterpret_cast<float4*>(prm)[tid + 256] = pb1;
  if (PROG == 1 || (PROG == 0 && wave < 2)) reinterpret_cast<float4*>(prm + kChBiasMax)[tid] = pl;
#pragma unroll
  for (int j = 0; j < 8 * RT; ++j) {                // -> hi / lo planes
    const int row = j * 4 + wave;
    unsigned h01, h23, l01, l23;
    ch_split2(v[j].x, v[j].y, h01, l01);
    ch_split2(v[j].z, v[j].w, h23, l23);
    char* q = tl + row * 512 + (((lane >> 1) ^ row) & 31) * 16 + (lane & 1) * 8;
    *reinterpret_cast<uint2*>(q) = make_uint2(h01, h23);
    *reinterpret_cast<uint2*>(q + kChPlane) = make_uint2(l01, l23);
  }
  const unsigned abase = (unsigned)(vi * 512 + ((kb ^ vi) & 31) * 16);
  ch_sync();
  OCC_CH_STAMP(1)                                   // first-stage rows landed, tile built
  int step = 0;
  int bias_off = 0;
  int ps0 = 0;                                      // first tail pass of the generic loop below
  if constexpr (PROG != 2) {
  // ---- stage 1: output_proj + bias + residual -> LayerNorm -------------------------------------------------------------
    ch_add_bias(acc, prm + wave * 64, kb);
    ch_kloop<ABL>(acc, w, wr, wv, 0, (0) + 16, tl, abase, rot, acc, wr, row);
    OCC_CH_STAMP(2)
    ch_layernorm(acc, red, prm_ln, prm_ln + 256, p.eps1, wave, vi, kb);
    if constexpr (PROG == 0) { OCC_CH_STAMP(4) }
    if constexpr (PROG == 0) ch_store(acc, ry, ldy_b, row, wave * 64, kb, true, true);      // x1 (B: x2 leaves under S2a)
    if constexpr (PROG == 0) { OCC_CH_STAMP(5) }
    ch_to_tile(acc, tl, wave, vi, kb);
    if constexpr (PROG == 0) { OCC_CH_STAMP(6) }
    ch_sync();
    OCC_CH_STAMP(3)                                   // LayerNorm, row stores issued, tile rebuilt
  step = 16;
  bias_off = 256;
  }

  if constexpr (PROG == 1) {
    // ---- FFN: both hidden halves from the x2 tile (registers), then the second Linear over the two K halves ------------
    f32x16 ha[RT][2], hb[RT][2];
    unsigned yoff[RT];                              // this lane's first quad of its rows of y
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) yoff[rt] = row[rt] * ldy_b + (unsigned)(wave * 64 + 4 * kb) * 4u;
    ch_set_bias(ha, prm + 256 + wave * 64, kb);
    // x2 (still in `acc`) is parked in the block's rows of y one quad per k-step, under the first hidden half's MFMAs
    if constexpr (!kChSpread) ch_store(acc, ry, ldy_b, row, wave * 64, kb, true, true);
    ch_kloop<ABL, kChSpread>(ha, w, wr, wv, 16, (16) + 16, tl, abase, rot, acc, ry, yoff);
    ch_relu(ha);
    OCC_CH_STAMP(4)
    ch_set_bias(hb, prm + 512 + wave * 64, kb);
    ch_kloop<ABL>(hb, w, wr, wv, 32, (32) + 16, tl, abase, rot, hb, wr, row);
    ch_relu(hb);
    OCC_CH_STAMP(5)
    ch_sync();                                // every wave has read the x2 tile for the last time
    ch_to_tile(ha, tl, wave, vi, kb);
    ch_sync();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // the x2 stores (two k loops ago) have landed
    ch_load_rows(acc, ry, ldy_b, row, wave * 64, kb);                    // x2 (this lane's own stores)
    ch_add_bias(acc, prm + 768 + wave * 64, kb);                         // + b2
    OCC_CH_STAMP(6)                                 // ha in the tile, x2 reloaded
    ch_kloop<ABL>(acc, w, wr, wv, 48, (48) + 16, tl, abase, rot, acc, wr, row);
    OCC_CH_STAMP(7)
    ch_sync();
    ch_to_tile(hb, tl, wave, vi, kb);
    ch_sync();
    OCC_CH_STAMP(8)
    ch_kloop<ABL>(acc, w, wr, wv, 64, (64) + 16, tl, abase, rot, acc, wr, row);
    OCC_CH_STAMP(9)
    ch_layernorm(acc, red, prm_ln + 512, prm_ln + 768, p.eps2, wave, vi, kb);
    OCC_CH_STAMP(16)
    step = 80;
    bias_off = 1024;
    if (p.npass > 0) {
      ch_to_tile(acc, tl, wave, vi, kb);
      OCC_CH_STAMP(17)
      ch_sync();
      OCC_CH_STAMP(10)                              // LayerNorm 2, tile rebuilt
      // tail pass 0 accumulates in `ha` (free by now) while x3 — still in `acc` — leaves one quad per k-step
      const int c0 = wave * 64;
      const float* pbias = prm + bias_off + c0;
      if (p.term != nullptr && c0 < p.term_cols) {
        ch_load_rows(ha, rterm, ldt_b, row, c0, kb);
        ch_add_bias(ha, pbias, kb);
      } else {
        ch_set_bias(ha, pbias, kb);
      }
      if constexpr (!kChSpread) ch_store(acc, ry, ldy_b, row, wave * 64, kb, true, true);
      ch_kloop<ABL, kChSpread>(ha, w, wr, wv, step, step + 16, tl, abase, rot, acc, ry, yoff);
      OCC_CH_STAMP(11)
      if (p.act) ch_relu(ha);
      const int ca = c0, cb = c0 + 32;
      const bool a1 = ca < p.n1, b1 = cb < p.n1;
      const bool a2 = ca >= p.off2 && ca < p.off2 + p.n2, b2 = cb >= p.off2 && cb < p.off2 + p.n2;
      if (a1 || b1) ch_store(ha, rz1, ldz1_b, row, c0, kb, a1, b1);
      if (a2 || b2) ch_store(ha, rz2, ldz2_b, row, c0 - p.off2, kb, a2, b2);
      ps0 = 1;
    } else {
      ch_store(acc, ry, ldy_b, row, wave * 64, kb, true, true);           // x3
      OCC_CH_STAMP(10)
    }
  }

  // ---- tail stage: npass passes of 256 columns over the LayerNorm'd tile -----------------------------------------------
#pragma unroll 1
  for (int ps = ps0; ps < p.npass; ++ps) {
    const int c0 = ps * 256 + wave * 64;            // the wave's first tail column of this pass
    const float* pbias = prm + bias_off + c0;
    if (p.term != nullptr && c0 < p.term_cols) {    // term_cols is a multiple of 64: whole waves
      ch_load_rows(acc, rterm, ldt_b, row, c0, kb);
      ch_add_bias(acc, pbias, kb);
    } else {
      ch_set_bias(acc, pbias, kb);
    }
    ch_kloop<ABL>(acc, w, wr, wv, step + ps * 16, (step + ps * 16) + 16, tl, abase, rot, acc, wr, row);
    OCC_CH_STAMP(11 + ps)                           // (passes 0 .. 3)
    if (p.act) ch_relu(acc);
    // the wave's two 32-column tiles go to z1 (columns < n1) or z2 (columns in [off2, off2 + n2)) or nowhere (padding)
    const int ca = c0, cb = c0 + 32;
    const bool a1 = ca < p.n1, b1 = cb < p.n1;
    const bool a2 = ca >= p.off2 && ca < p.off2 + p.n2, b2 = cb >= p.off2 && cb < p.off2 + p.n2;
    if (a1 || b1) ch_store(acc, rz1, ldz1_b, row, c0, kb, a1, b1);
    if (a2 || b2) ch_store(acc, rz2, ldz2_b, row, c0 - p.off2, kb, a2, b2);
  }
  OCC_CH_STAMP(15)
}
#undef OCC_CH_STAMP

// Blocks [0, nfull) own 64-row tiles; the rows behind them are cut into 32-row tiles (blocks nfull ..).  The launcher uses
// the second kind for the tiles that would otherwise form a thin last round: 40 000 rows are 625 tiles on 512 resident
// slots, and the 113 blocks of the second round ran one per CU — at the single-block rate, with 143 CUs idle — for as
// long as the whole first round (stamped timeline: profiles/r04_c18_chain_trace.txt).  As 226 half tiles that round
// occupies every CU and each block carries half the rows.
template <int PROG, int ABL = 0, bool TRACE = false>
__global__ __launch_bounds__(256, 2) void linear_chain_x3_kernel(const ChainArgs p) {
  extern __shared__ __attribute__((aligned(16))) char tl[];
  const int b = (int)blockIdx.x;
  if (b < p.nfull)
    chain_tile<PROG, ABL, TRACE, 2>(p, tl, (long)b * kChRows);
  else
    chain_tile<PROG, ABL, TRACE, 1>(p, tl, (long)p.nfull * kChRows + (long)(b - p.nfull) * 32);
}


#undef OCC_CH_LOAD
#undef OCC_CH_AFRAG

}  // namespace occ

extern "C" int64_t occ_linear_chain_packed_bytes(int N, int K) {
  if (N <= 0 || K <= 0) return 0;
  return (int64_t)((N + 255) / 256) * 256 * K * 4;
}

extern "C" int occ_linear_chain_pack_bf16x3(const float* weight, void* packed, int N, int K, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(weight && packed, "linear_chain_pack_bf16x3: null pointer argument");
  OCC_CHECK_ARG(N > 0 && K > 0, "linear_chain_pack_bf16x3: bad dimension");
  if (K % 16) {
    set_error("linear_chain_pack_bf16x3: K=%d is not a multiple of 16", K);
    return OCC_E_UNSUPPORTED;
  }
  const long n = (long)((N + 255) / 256) * 256 * K;       // (hi, lo) pairs incl. the zero rows
  hipLaunchKernelGGL(linear_chain_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), weight, reinterpret_cast<unsigned short*>(packed), n, K, N);
  OCC_CHECK_LAUNCH("linear_chain_pack_bf16x3");
  return OCC_OK;
}

namespace {
template <int PROG>
int chain_launch(const occ::ChainArgs& args_in, hipStream_t st, const char* what) {
  using namespace occ;
  // OCC_CHAIN_ABLATE (program A, timing only — results wrong by construction): 1 no weight loads in the k loops,
  // 2 no MFMAs, 4 no operand-fragment reads.  (A switch that skipped / streamed the row stores produced
  // profiles/r04_c9_stores.txt and was removed: its run-time test put three branches around every store.)
  static const int abl = [] { const char* e = getenv("OCC_CHAIN_ABLATE"); return e ? atoi(e) : 0; }();
  ChainArgs args = args_in;
  {
    // 32-bit byte offsets inside the kernel (buffer instructions): every matrix, padded by one tile of rows, stays below 4 GB
    const long lds[6] = {args.lda, args.ldres, args.ldy, args.ldterm, args.ldz1, args.ldz2};
    for (long ld : lds)
      if (((long)args.M + kChRows) * ld * 4 >= (1L << 32)) {
        set_error("%s: M=%d rows of %ld floats exceed the 4 GB a buffer resource addresses", what, args.M, ld);
        return OCC_E_UNSUPPORTED;
      }
  }
  // tile split (kernel comment): the last, partial round of 64-row tiles becomes 32-row tiles when it would fill less
  // than half of the resident slots.  OCC_CHAIN_HALF_TILES=0 keeps 64-row tiles throughout (development).
  static const int slots = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return 2 * n;
  }();
  static const bool half_tiles = [] { const char* e = getenv("OCC_CHAIN_HALF_TILES"); return !(e && e[0] == '0'); }();
  const int n64 = (args.M + kChRows - 1) / kChRows;
  const int rem = n64 % slots;
  args.nfull = (half_tiles && n64 > slots && rem > 0 && 2 * rem <= slots) ? n64 - rem : n64;
  const long rows_left = (long)args.M - (long)args.nfull * kChRows;          // rows behind the 64-row tiles
  const int ntiles = args.nfull + (rows_left > 0 ? (int)((rows_left + 31) / 32) : 0);
  void (*kern)(const ChainArgs) = linear_chain_x3_kernel<PROG, 0>;
  if (PROG == 0 && abl == 1) kern = linear_chain_x3_kernel<0, 1>;
  if (PROG == 0 && abl == 2) kern = linear_chain_x3_kernel<0, 2>;
  if (PROG == 0 && abl == 4) kern = linear_chain_x3_kernel<0, 4>;
  if (PROG == 0 && abl == 3) kern = linear_chain_x3_kernel<0, 3>;
  // OCC_CHAIN_TRACE=<file prefix> (development, tools_dev/chain_probe.py): every launch runs the stamped build, waits for
  // it and appends its 24 wall-clock stamps per wave (100 MHz) to <prefix>.<A|B>.bin
  const char* trace_to = getenv("OCC_CHAIN_TRACE");
  if (trace_to && *trace_to) {
    kern = linear_chain_x3_kernel<PROG, 0, true>;
    if (hipMalloc(reinterpret_cast<void**>(&args.trace), (size_t)ntiles * 96 * sizeof(long long)) != hipSuccess) {
      set_error("%s: trace buffer allocation failed", what);
      return OCC_E_LAUNCH;
    }
    (void)hipMemsetAsync(args.trace, 0, (size_t)ntiles * 96 * sizeof(long long), st);
  }
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           kChLds);
  if (e != hipSuccess) {
    set_error("%s: hipFuncSetAttribute failed: %s", what, hipGetErrorString(e));
    return OCC_E_LAUNCH;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)ntiles), dim3(256), kChLds, st, args);
  OCC_CHECK_LAUNCH(what);
  if (args.trace) {
    std::vector<long long> host((size_t)ntiles * 96);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(host.data(), args.trace, host.size() * sizeof(long long), hipMemcpyDeviceToHost);
    (void)hipFree(args.trace);
    const std::string path = std::string(trace_to) + (PROG == 0 ? ".A.bin" : ".B.bin");
    if (FILE* f = fopen(path.c_str(), "ab")) {
      fwrite(host.data(), sizeof(long long), host.size(), f);
      fclose(f);
    }
  }
  return OCC_OK;
}
}  // namespace

extern "C" int occ_linear_ln_chain_bf16x3_f32(const float* a, int64_t lda, const float* residual, int64_t ldres,
                                              const void* w_chain, const float* bias_chain, const float* ln_gamma,
                                              const float* ln_beta, float ln_eps, float* y, int64_t ldy, float* z,
                                              int64_t ldz, int n2, int act2, int M, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(a && residual && w_chain && bias_chain && ln_gamma && ln_beta && y && z,
                "linear_ln_chain: null pointer argument");
  OCC_CHECK_ARG(M > 0 && n2 > 0 && (act2 == 0 || act2 == 1), "linear_ln_chain: bad dimension (M=%d n2=%d act2=%d)", M, n2, act2);
  OCC_CHECK_ARG(lda >= 256 && ldres >= 256 && ldy >= 256 && ldz >= n2, "linear_ln_chain: leading dimension smaller than the row");
  if (n2 % 32 || lda % 4 || ldres % 4 || ldy % 4 || ldz % 4) {
    set_error("linear_ln_chain: n2=%d must be a multiple of 32 and all rows 16-byte aligned", n2);
    return OCC_E_UNSUPPORTED;
  }
  ChainArgs g = {};
  g.a = a; g.lda = lda; g.res = residual; g.ldres = ldres;
  g.npass = (n2 + 255) / 256;
  if (256 + 256 * g.npass > kChBiasMax) {
    set_error("linear_ln_chain: n2=%d exceeds the %d tail columns the kernel stages parameters for", n2, kChBiasMax - 256);
    return OCC_E_UNSUPPORTED;
  }
  g.wp = reinterpret_cast<const uint4*>(w_chain);
  g.wbytes = (unsigned)(16 + 16 * g.npass) * (unsigned)kChStepBytes;
  g.bias = bias_chain;
  g.ln1_g = ln_gamma; g.ln1_b = ln_beta; g.eps1 = ln_eps;
  g.y = y; g.ldy = ldy;
  g.act = act2;
  g.z1 = z; g.ldz1 = ldz; g.n1 = n2;
  g.z2 = nullptr; g.ldz2 = 0; g.off2 = 0; g.n2 = 0;
  g.M = M;
  return chain_launch<0>(g, reinterpret_cast<hipStream_t>(stream), "linear_ln_chain");
}

extern "C" int occ_encoder_ffn_chain_bf16x3_f32(const float* a, int64_t lda, const float* residual, int64_t ldres,
                                                const void* w_chain, const float* bias_chain, const float* ln1_gamma,
                                                const float* ln1_beta, float ln1_eps, const float* ln2_gamma,
                                                const float* ln2_beta, float ln2_eps, float* y, int64_t ldy,
                                                const float* q_term, int64_t ldq_term, float* zq, int64_t ldzq, int nq,
                                                float* zv, int64_t ldzv, int M, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(a && residual && w_chain && bias_chain && ln1_gamma && ln1_beta && ln2_gamma && ln2_beta && y,
                "encoder_ffn_chain: null pointer argument");
  OCC_CHECK_ARG(M > 0, "encoder_ffn_chain: bad dimension (M=%d)", M);
  const bool tail = zq != nullptr || zv != nullptr;
  OCC_CHECK_ARG(!tail || (zq && zv && nq > 0 && nq <= 256), "encoder_ffn_chain: the tail needs zq, zv and 0 < nq <= 256");
  OCC_CHECK_ARG(lda >= 256 && ldres >= 256 && ldy >= 256 && (!tail || (ldzq >= nq && ldzv >= 256 && (!q_term || ldq_term >= nq))),
                "encoder_ffn_chain: leading dimension smaller than the row");
  if (lda % 4 || ldres % 4 || ldy % 4 || (tail && (nq % 64 || ldzq % 4 || ldzv % 4 || ldq_term % 4))) {
    set_error("encoder_ffn_chain: nq=%d must be a multiple of 64 and all rows 16-byte aligned", nq);
    return OCC_E_UNSUPPORTED;
  }
  ChainArgs g = {};
  g.a = a; g.lda = lda; g.res = residual; g.ldres = ldres;
  g.npass = tail ? 2 : 0;
  g.wp = reinterpret_cast<const uint4*>(w_chain);
  g.wbytes = (unsigned)(80 + 16 * g.npass) * (unsigned)kChStepBytes;
  g.bias = bias_chain;
  g.ln1_g = ln1_gamma; g.ln1_b = ln1_beta; g.eps1 = ln1_eps;
  g.ln2_g = ln2_gamma; g.ln2_b = ln2_beta; g.eps2 = ln2_eps;
  g.y = y; g.ldy = ldy;
  g.act = 0;
  g.term = tail ? q_term : nullptr; g.ldterm = ldq_term; g.term_cols = nq;
  g.z1 = zq; g.ldz1 = ldzq; g.n1 = tail ? nq : 0;
  g.z2 = zv; g.ldz2 = ldzv; g.off2 = 256; g.n2 = tail ? 256 : 0;
  g.M = M;
  return chain_launch<1>(g, reinterpret_cast<hipStream_t>(stream), "encoder_ffn_chain");
}

extern "C" int occ_linear_pair_chain_bf16x3_f32(const float* a, int64_t lda, const void* w_chain, const float* bias_chain,
                                                const float* q_term, int64_t ldq_term, float* zq, int64_t ldzq, int nq,
                                                float* zv, int64_t ldzv, int M, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(a && w_chain && bias_chain && zq && zv, "linear_pair_chain: null pointer argument");
  OCC_CHECK_ARG(M > 0 && nq > 0 && nq <= 256, "linear_pair_chain: bad dimension (M=%d nq=%d)", M, nq);
  OCC_CHECK_ARG(lda >= 256 && ldzq >= nq && ldzv >= 256 && (!q_term || ldq_term >= nq),
                "linear_pair_chain: leading dimension smaller than the row");
  if (lda % 4 || nq % 64 || ldzq % 4 || ldzv % 4 || ldq_term % 4) {
    set_error("linear_pair_chain: nq=%d must be a multiple of 64 and all rows 16-byte aligned", nq);
    return OCC_E_UNSUPPORTED;
  }
  ChainArgs g = {};
  g.a = a; g.lda = lda;
  g.npass = 2;
  g.wp = reinterpret_cast<const uint4*>(w_chain);
  g.wbytes = (unsigned)(16 * g.npass) * (unsigned)kChStepBytes;
  g.bias = bias_chain;
  g.act = 0;
  g.term = q_term; g.ldterm = ldq_term; g.term_cols = nq;
  g.z1 = zq; g.ldz1 = ldzq; g.n1 = nq;
  g.z2 = zv; g.ldz2 = ldzv; g.off2 = 256; g.n2 = 256;
  g.M = M;
  return chain_launch<2>(g, reinterpret_cast<hipStream_t>(stream), "linear_pair_chain");
}
