// LAB (not built into libocc_amd.so): one-launch FFN on the weight-stationary tile machinery of csrc/linear_ws.hip.
// Measured on MI355X (profiles/r03_linear_probe_ws_v2.txt): 121 us against 103 us for the two Linear launches it would
// replace.  Why it cannot win in this form: W1 + W2 are 1 MB of hi/lo bf16 fragments; a CU's LDS holds the bf16 planes of
// only 64 rows next to the hidden tile, so the whole megabyte is re-streamed from L2 for every 64 rows — 768 MB per
// launch through L2 -> TA (all 256 CUs asking for the same lines in lockstep), more than the 164 MB of hidden-activation
// HBM traffic the fusion saves.  Kept with its numbers; to build it, append this file's body to csrc/linear_ws.hip
// (it uses that file's helpers) and declare occ_ffn_ws_bf16x3_f32.
// ------------------------------------------------------------------------------------------------------------------
// The encoder's feed-forward block + the LayerNorm that follows it as ONE launch on the same tile machinery:
//     out = LayerNorm( x + W2 . relu(W1 . x + b1) + b2 )          (C = 256, hidden = 512)
// (mmcv FFN + norm, encoder.py:377-404, custom_base_transformer_layer.py:74-99).  Round 2's fused FFN kept the
// hidden activations in registers at one wave per SIMD and lost to two launches; here the 64 x 512 hidden tile goes
// through LDS in two halves and never reaches HBM (2 x 82 MB per layer).  Per 64-row tile, for hidden half p = 0, 1:
//   GEMM1 (transposed: D = W1 . X^T, so a lane's four consecutive D registers are four consecutive hidden units of one
//   row -> bias, ReLU, hi/lo split, 8-byte LDS stores straight into the A-operand planes of GEMM2), then
//   GEMM2 accumulates out += H_p . W2[:, half p]^T.
// The weights (1 MB of hi/lo fragments, too many for the register file) are a STREAM of 64 k-steps per tile — [W1 half
// 0 | W2 half 0 | W1 half 1 | W2 half 1] x 16 — that runs through an 8-step register ring: the fragments of step s + 8
// are requested from L2 right after the MFMAs of step s have consumed their slot, across phase and tile boundaries.
// Two 67.5 KB LDS regions swap roles every tile: {X planes | H planes + epilogue tile}; the next tile's fp32 rows
// arrive by LDS-DMA in the X region as soon as the second GEMM1 pass has read it.
struct FfnRing { uint4 h[8], l[8]; };

// fragments of stream step s (0..63 within a tile; the stream is periodic) of this wave's 32 columns: buffer loads with
// a wave-uniform scalar offset per step and ONE vector offset (lane * 16) for the whole stream — as flat loads hipcc
// built a 64-bit vector address per step and spilled them
template <int S>
__device__ __forceinline__ void ffn_ring_load(FfnRing& ring, __amdgpu_buffer_rsrc_t r1, __amdgpu_buffer_rsrc_t r2,
                                              int wave_off1, int wave_off2, int lane16) {
  constexpr int s = S & 63, ph = s >> 4, ks = s & 15, p = ph >> 1;
  if ((ph & 1) == 0) {       // W1 (hidden 512, C 256): 16 column tiles of 32, 16 k-steps; uint4 index ks*2048 + (p*8+wave)*128
    const int so = (ks * 2048 + p * 8 * 128) * 16 + wave_off1;
    ring.h[S & 7] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r1, lane16, so, 0));
    ring.l[S & 7] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r1, lane16 + 1024, so, 0));
  } else {                   // W2 (C 256, hidden 512): 8 column tiles, 32 k-steps; uint4 index (p*16+ks)*1024 + wave*128
    const int so = (p * 16 + ks) * 1024 * 16 + wave_off2;
    ring.h[S & 7] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r2, lane16, so, 0));
    ring.l[S & 7] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r2, lane16 + 1024, so, 0));
  }
}

// one phase = 16 stream steps over the 64 rows of the planes at `pa` (both 32-row halves per step: independent
// accumulators), A fragments read one step ahead
template <int PH, bool TRANSPOSED>
__device__ __forceinline__ void ffn_phase(const char* pa, FfnRing& ring, __amdgpu_buffer_rsrc_t r1,
                                          __amdgpu_buffer_rsrc_t r2, int wo1, int wo2, int lane16, f32x16& a0,
                                          f32x16& a1) {
  bf16x8 fh0[2], fl0[2], fh1[2], fl1[2];
  fh0[0] = *reinterpret_cast<const bf16x8*>(pa);
  fl0[0] = *reinterpret_cast<const bf16x8*>(pa + kWsPlane);
  fh1[0] = *reinterpret_cast<const bf16x8*>(pa + 32 * kWsPitch);
  fl1[0] = *reinterpret_cast<const bf16x8*>(pa + 32 * kWsPitch + kWsPlane);
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    constexpr int dummy = 0;
    (void)dummy;
    if (ks + 1 < 16) {
      const int n = (ks + 1) & 1;
      fh0[n] = *reinterpret_cast<const bf16x8*>(pa + (ks + 1) * 32);
      fl0[n] = *reinterpret_cast<const bf16x8*>(pa + kWsPlane + (ks + 1) * 32);
      fh1[n] = *reinterpret_cast<const bf16x8*>(pa + 32 * kWsPitch + (ks + 1) * 32);
      fl1[n] = *reinterpret_cast<const bf16x8*>(pa + 32 * kWsPitch + kWsPlane + (ks + 1) * 32);
    }
    const int c = ks & 1, slot = ks & 7;          // (PH*16 + ks) & 7 == ks & 7
    const bf16x8 bh = __builtin_bit_cast(bf16x8, ring.h[slot]), bl = __builtin_bit_cast(bf16x8, ring.l[slot]);
    if (TRANSPOSED) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, fl0[c], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, fl1[c], a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, fh0[c], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, fh1[c], a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, fh0[c], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, fh1[c], a1, 0, 0, 0);
    } else {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl0[c], bh, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl1[c], bh, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh0[c], bl, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh1[c], bl, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh0[c], bh, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh1[c], bh, a1, 0, 0, 0);
    }
    // the slot is free: request the fragments of stream step s + 8 (next phase / next tile included)
    switch (ks) {
#define OCC_FFN_NEXT(KS) case KS: ffn_ring_load<PH * 16 + KS + 8>(ring, r1, r2, wo1, wo2, lane16); break;
      OCC_FFN_NEXT(0) OCC_FFN_NEXT(1) OCC_FFN_NEXT(2) OCC_FFN_NEXT(3) OCC_FFN_NEXT(4) OCC_FFN_NEXT(5)
      OCC_FFN_NEXT(6) OCC_FFN_NEXT(7) OCC_FFN_NEXT(8) OCC_FFN_NEXT(9) OCC_FFN_NEXT(10) OCC_FFN_NEXT(11)
      OCC_FFN_NEXT(12) OCC_FFN_NEXT(13) OCC_FFN_NEXT(14) OCC_FFN_NEXT(15)
#undef OCC_FFN_NEXT
    }
    // pin the software pipeline: hipcc otherwise sinks every ring request down to its use (load, vmcnt(0), MFMA) — and a
    // full sched_barrier here made it spill 290 registers.  Groups in program order per step:
    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);   // the step's six MFMAs,
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);   // then the two ring requests for step s + 8,
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // then the four fragment reads of step s + 1
  }
}

// GEMM1 result of hidden half P -> bias, ReLU, hi/lo split -> H planes (A operand of GEMM2)
__device__ __forceinline__ void ffn_write_hidden(char* HP, const f32x16& h0, const f32x16& h1, const float* __restrict__ b1,
                                                 int p, int wave, int vi, int kb) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {    // register group g of a lane = hidden units wave*32 + 8g + 4kb .. +3 of row rt*32 + vi
    const float4 bb = *reinterpret_cast<const float4*>(b1 + p * 256 + wave * 32 + 8 * g + 4 * kb);
    unsigned h01, h23, l01, l23;
    char* dst = HP + vi * kWsPitch + (wave * 32 + 8 * g + 4 * kb) * 2;
    ws_split2(fmaxf(h0[4 * g] + bb.x, 0.f), fmaxf(h0[4 * g + 1] + bb.y, 0.f), h01, l01);
    ws_split2(fmaxf(h0[4 * g + 2] + bb.z, 0.f), fmaxf(h0[4 * g + 3] + bb.w, 0.f), h23, l23);
    *reinterpret_cast<uint2*>(dst) = make_uint2(h01, h23);
    *reinterpret_cast<uint2*>(dst + kWsPlane) = make_uint2(l01, l23);
    ws_split2(fmaxf(h1[4 * g] + bb.x, 0.f), fmaxf(h1[4 * g + 1] + bb.y, 0.f), h01, l01);
    ws_split2(fmaxf(h1[4 * g + 2] + bb.z, 0.f), fmaxf(h1[4 * g + 3] + bb.w, 0.f), h23, l23);
    *reinterpret_cast<uint2*>(dst + 32 * kWsPitch) = make_uint2(h01, h23);
    *reinterpret_cast<uint2*>(dst + 32 * kWsPitch + kWsPlane) = make_uint2(l01, l23);
  }
}

__global__ __launch_bounds__(512, 2) void ffn_ws_kernel(
    const float* __restrict__ x, long ldx, const uint4* __restrict__ w1p, const float* __restrict__ b1,
    const uint4* __restrict__ w2p, const float* __restrict__ b2, const float* __restrict__ ln_g,
    const float* __restrict__ ln_b, float ln_eps, float* __restrict__ out, long ldo, int M, int rows_per_block) {
  __shared__ __attribute__((aligned(16))) char lds[4 * kWsPlane];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int vi = lane & 31, kb = lane >> 5;
  const long r_begin = (long)blockIdx.x * rows_per_block;
  long r_end = r_begin + rows_per_block;
  if (r_end > M) r_end = M;
  if (r_begin >= r_end) return;
  const int ntiles = (int)((r_end - r_begin + kWsTile - 1) / kWsTile);
  const int c = lane * 4;
  const float4 b2v = *reinterpret_cast<const float4*>(b2 + c);
  float4 gv = make_float4(1.f, 1.f, 1.f, 1.f), bev = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ln_g) {
    gv = *reinterpret_cast<const float4*>(ln_g + c);
    bev = *reinterpret_cast<const float4*>(ln_b + c);
  }

  ws_dma_tile(x, ldx, r_begin, r_end - 1, lds + 2 * kWsPlane, wave, lane);     // raw tile 0 -> region 1
  const __amdgpu_buffer_rsrc_t r1 = uniform_rsrc(w1p, 512u * 256u * 4u), r2 = uniform_rsrc(w2p, 512u * 256u * 4u);
  const int wo1 = wave * 128 * 16, wo2 = wave * 128 * 16, lane16 = lane * 16;
  FfnRing ring;
  ffn_ring_load<0>(ring, r1, r2, wo1, wo2, lane16); ffn_ring_load<1>(ring, r1, r2, wo1, wo2, lane16);
  ffn_ring_load<2>(ring, r1, r2, wo1, wo2, lane16); ffn_ring_load<3>(ring, r1, r2, wo1, wo2, lane16);
  ffn_ring_load<4>(ring, r1, r2, wo1, wo2, lane16); ffn_ring_load<5>(ring, r1, r2, wo1, wo2, lane16);
  ffn_ring_load<6>(ring, r1, r2, wo1, wo2, lane16); ffn_ring_load<7>(ring, r1, r2, wo1, wo2, lane16);
  __syncthreads();
  ws_split_tile(lds + 2 * kWsPlane, lds, tid);                                 // -> X planes in region 0
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    char* XP = lds + (t & 1) * 2 * kWsPlane;
    char* HP = lds + ((t + 1) & 1) * 2 * kWsPlane;
    const long row0 = r_begin + (long)t * kWsTile;
    const char* xa = XP + vi * kWsPitch + kb * 16;
    const char* ha = HP + vi * kWsPitch + kb * 16;
    f32x16 o0, o1, h0, h1;                           // o: GEMM2 accumulators, 64 rows x this wave's 32 output columns
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; h0[r] = 0.f; h1[r] = 0.f; }
    ffn_phase<0, true>(xa, ring, r1, r2, wo1, wo2, lane16, h0, h1);                // GEMM1, hidden half 0
    ffn_write_hidden(HP, h0, h1, b1, 0, wave, vi, kb);
    __syncthreads();                                                           // H half 0 complete
    ffn_phase<1, false>(ha, ring, r1, r2, wo1, wo2, lane16, o0, o1);               // GEMM2, k half 0
#pragma unroll
    for (int r = 0; r < 16; ++r) { h0[r] = 0.f; h1[r] = 0.f; }
    ffn_phase<2, true>(xa, ring, r1, r2, wo1, wo2, lane16, h0, h1);                // GEMM1, hidden half 1 (X planes: no hazard)
    __syncthreads();                                                           // every wave is done reading H half 0 and X
    ffn_write_hidden(HP, h0, h1, b1, 1, wave, vi, kb);
    if (t + 1 < ntiles) ws_dma_tile(x, ldx, row0 + kWsTile, r_end - 1, XP, wave, lane);   // X region is dead: next raw tile
    __syncthreads();                                                           // H half 1 complete
    ffn_phase<3, false>(ha, ring, r1, r2, wo1, wo2, lane16, o0, o1);               // GEMM2, k half 1
    __syncthreads();                                                           // every wave is done reading H half 1
    // ---- epilogue through the H region: + b2 + x, LayerNorm, store
    float4 rres[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {               // the residual rows (= x itself), L2-warm: this tile came in by DMA
      long m = row0 + wave * 8 + rr;
      if (m > r_end - 1) m = r_end - 1;
      rres[rr] = *reinterpret_cast<const float4*>(x + m * ldx + c);
    }
    float* sO = reinterpret_cast<float*>(HP);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * kb;
      sO[row * kWsOLd + wave * 32 + vi] = o0[r];
      sO[(row + 32) * kWsOLd + wave * 32 + vi] = o1[r];
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int row = wave * 8 + rr;
      const long m = row0 + row;
      if (m >= r_end) break;                   // wave-uniform
      float4 v = *reinterpret_cast<const float4*>(sO + row * kWsOLd + c);
      v.x += b2v.x + rres[rr].x; v.y += b2v.y + rres[rr].y; v.z += b2v.z + rres[rr].z; v.w += b2v.w + rres[rr].w;
      if (ln_g) {
        const float mean = ws_wave_sum((v.x + v.y) + (v.z + v.w)) * (1.f / 256.f);
        const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
        const float var = ws_wave_sum((dx * dx + dy * dy) + (dz * dz + dw * dw)) * (1.f / 256.f);
        const float rstd = rsqrtf(var + ln_eps);
        v.x = dx * rstd * gv.x + bev.x; v.y = dy * rstd * gv.y + bev.y;
        v.z = dz * rstd * gv.z + bev.z; v.w = dw * rstd * gv.w + bev.w;
      }
      *reinterpret_cast<float4*>(out + m * ldo + c) = v;
    }
    if (t + 1 < ntiles) {
      __syncthreads();                 // epilogue tile consumed; the next raw tile (in XP) landed
      ws_split_tile(XP, HP, tid);      // -> X planes of tile t+1 in this tile's H region (roles swap)
      __syncthreads();
    }
  }
}

}  // namespace occ

extern "C" int occ_ffn_ws_bf16x3_f32(const float* x, int64_t ldx, const void* w1_packed, const float* b1,
                                     const void* w2_packed, const float* b2, const float* ln_gamma,
                                     const float* ln_beta, float ln_eps, float* out, int64_t ldo, int M, int C,
                                     int hidden, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(x && w1_packed && b1 && w2_packed && b2 && out, "ffn_ws: null pointer argument");
  OCC_CHECK_ARG(M > 0, "ffn_ws: bad dimension (M=%d)", M);
  OCC_CHECK_ARG((ln_gamma == nullptr) == (ln_beta == nullptr), "ffn_ws: ln_gamma and ln_beta go together");
  OCC_CHECK_ARG(ldx >= C && ldo >= C, "ffn_ws: leading dimension smaller than the row");
  if (C != 256 || hidden != 512 || ldx % 4 || ldo % 4 || (reinterpret_cast<uintptr_t>(x) & 15)) {
    set_error("ffn_ws: no kernel for C=%d hidden=%d (need C == 256, hidden == 512, 16-byte aligned rows)", C, hidden);
    return OCC_E_UNSUPPORTED;
  }
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  int nrb = cus;
  const int max_rb = (M + kWsTile - 1) / kWsTile;
  if (nrb > max_rb) nrb = max_rb;
  const int rows_per_block = (M + nrb - 1) / nrb;
  nrb = (M + rows_per_block - 1) / rows_per_block;
  hipLaunchKernelGGL(ffn_ws_kernel, dim3((unsigned)nrb), dim3(512), 0, reinterpret_cast<hipStream_t>(stream), x,
                     (long)ldx, reinterpret_cast<const uint4*>(w1_packed), b1,
                     reinterpret_cast<const uint4*>(w2_packed), b2, ln_gamma, ln_beta, ln_eps, out, (long)ldo, M,
                     rows_per_block);
  OCC_CHECK_LAUNCH("ffn_ws");
  return OCC_OK;
}
