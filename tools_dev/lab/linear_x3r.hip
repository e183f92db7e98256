// REJECTED (round 3) — kept for re-measurement; not built into libocc_amd.so.  See tools_dev/lab/README.md.
// Activation-resident bf16x3 Linear: the value_proj_resident_kernel design applied to the encoder's f32 Linears.
// To revive: paste the kernel into occnet_amd/csrc/linear_bf16x3.hip (inside namespace occ, after linear_bf16x3_kernel)
// and the dispatch block into occ_linear_bf16x3_f32 before the tiled launch.
//
// ======================================= kernel =======================================================================

// ---- activation-resident variant --------------------------------------------------------------------------------------
// The tiled kernel above meets a block barrier every 16 k (12 MFMAs per wave between barriers, two or three waves per
// SIMD): PMC shows its waves 45 % waiting, MfmaUtil 0.30-0.39 (profiles/r03_linear_pmc_x3_vs_ws.txt).  Same idea as
// value_proj_resident_kernel: a block owns 64 rows and keeps their hi/lo bf16 planes for 256 k in LDS (2 x 32 KB, split
// once, 16-byte pieces XOR-swizzled by row: conflict-free A-fragment reads); after ONE barrier the four waves run free —
// wave w walks the column passes (256 columns per pass, its 64 as 2 x 2 accumulator tiles), the hi/lo weight fragments
// stream from L2 through a 4-deep register ring that runs ahead across k-steps, K phases and passes.  K = 512 (FFN2, the
// two-segment TSA query Linear) is two phases over the same LDS (the tile is reloaded between them: 3 barriers instead
// of 32).  MFMAs are transposed (weights = row operand): a lane holds 4 consecutive columns of one row per register
// quad, the accumulators start at the bias.  Epilogues: per wave through a 2.5 KB scratch (bias, ReLU, residual; any
// number of passes), or — LayerNorm, N <= 256 — the block's 64 x N f32 tile goes row-major into the planes' space and
// every wave normalises 16 full rows (two-pass, wave reductions) as in the tiled kernel.
constexpr int kXrRows = 64, kXrPhaseK = 256, kXrPlane = kXrRows * kXrPhaseK * 2;
constexpr int kXrPitch = 80, kXrScratch = 32 * kXrPitch, kXrLnPitch = 260;
constexpr int kXrLdsBytes = 2 * kXrPlane + 4 * kXrScratch;
static_assert(kXrRows * kXrLnPitch * 4 <= kXrLdsBytes, "LayerNorm staging fits the block's LDS");

template <bool LN>
__global__ __launch_bounds__(256, 2) void linear_x3r_kernel(
    const float* __restrict__ a1, long lda1, int K1, const float* __restrict__ a2,
    const float* __restrict__ a2add, long lda2, int K2, const uint4* __restrict__ wp,
    const float* __restrict__ bias, int act, const float* __restrict__ residual, long ldres,
    const float* __restrict__ ln_g, const float* __restrict__ ln_b, float ln_eps,
    float* __restrict__ out, long ldo, int M, int N) {
  extern __shared__ __attribute__((aligned(16))) char xl[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vi = lane & 31, kb = lane >> 5;
  const int m0 = (int)blockIdx.x * kXrRows;
  const int NT32 = N / 32, NP = (N + 255) / 256, NPH = (K1 + K2) / kXrPhaseK;

  // weight ring: slot (step & 3) = {hi tile 0, lo tile 0, hi tile 1, lo tile 1}; a "segment" = 16 k-steps of one
  // (pass, phase); buffer loads: one lane-offset VGPR, the step's offset in an SGPR, plane / tile in the immediate
  occ_u32x4 w[4][4];
  const __amdgpu_buffer_rsrc_t wr = uniform_rsrc(wp, (unsigned)((K1 + K2) / 16) * (unsigned)NT32 * 2048u);
  const int wv = (wave * 2 * 128 + lane) * 16;
  const int kstep_bytes = NT32 * 2048;
#define OCC_XR_LOAD(SLOT, SEGBASE, KS)                                                             \
  {                                                                                               \
    const int so = (SEGBASE) + (KS) * kstep_bytes;                                                \
    w[SLOT][0] = __builtin_amdgcn_raw_buffer_load_b128(wr, wv, so, 0);                            \
    w[SLOT][1] = __builtin_amdgcn_raw_buffer_load_b128(wr, wv + 1024, so, 0);                     \
    w[SLOT][2] = __builtin_amdgcn_raw_buffer_load_b128(wr, wv + 2048, so, 0);                     \
    w[SLOT][3] = __builtin_amdgcn_raw_buffer_load_b128(wr, wv + 3072, so, 0);                     \
  }
  // the 64 x 256 activation tile of K phase ph: wave w splits rows 16 w .. 16 w + 15 (a row = one coalesced 1 KB load);
  // LDS slot s of row r holds the row's 16-byte piece (8 bf16) s ^ (r & 31), hi plane then lo plane
  auto load_tile = [&](int ph) {
    const int k0 = ph * kXrPhaseK;
    const bool seg2 = k0 >= K1;
    const float* __restrict__ src = (seg2 ? a2 + (k0 - K1) : a1 + k0) + lane * 4;
    const long lda = seg2 ? lda2 : lda1;
    const float* __restrict__ add = seg2 && a2add ? a2add + (k0 - K1) + lane * 4 : nullptr;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int m = m0 + wave * 16 + half * 8 + j;
        if (m >= M) m = M - 1;
        v[j] = *reinterpret_cast<const float4*>(src + (long)m * lda);
        if (add) {
          const float4 d = *reinterpret_cast<const float4*>(add + (long)m * lda);
          v[j].x += d.x; v[j].y += d.y; v[j].z += d.z; v[j].w += d.w;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int row = wave * 16 + half * 8 + j;
        unsigned h01, h23, l01, l23;
        x3_split2(v[j].x, v[j].y, h01, l01);
        x3_split2(v[j].z, v[j].w, h23, l23);
        char* d = xl + row * 512 + (((lane >> 1) ^ (row & 31)) * 16) + (lane & 1) * 8;
        *reinterpret_cast<uint2*>(d) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(d + kXrPlane) = make_uint2(l01, l23);
      }
    }
  };

  // A fragments (hi, lo) of k-step ks: double buffered, read one step ahead; the slot of piece 2 ks + kb in row vi is
  // (2 ks) ^ ((kb ^ vi) & 31): one XOR per step on an address the compiler cannot see through (it would otherwise keep
  // 16 addresses in registers)
  bf16x8 ah[2][2], al[2][2];
  unsigned abase = (unsigned)(vi * 512 + ((kb ^ vi) & 31) * 16);
#define OCC_XR_AFRAG(BUF, KS)                                                                      \
  {                                                                                               \
    asm volatile("" : "+v"(abase));                                                               \
    const char* ap = xl + (abase ^ (unsigned)((KS) * 32));                                        \
    _Pragma("unroll") for (int rt = 0; rt < 2; ++rt) {                                            \
      ah[BUF][rt] = *reinterpret_cast<const bf16x8*>(ap + rt * (32 * 512));                       \
      al[BUF][rt] = *reinterpret_cast<const bf16x8*>(ap + rt * (32 * 512) + kXrPlane);            \
    }                                                                                             \
  }
  char* scratch = xl + 2 * kXrPlane + wave * kXrScratch;
  const int nseg = NP * NPH;
  bool active = wave * 64 < N;                      // pass 0; N % 64 == 0: a wave has both of its tiles or none
  if (active) {
    OCC_XR_LOAD(0, 0, 0)
    OCC_XR_LOAD(1, 0, 1)
    OCC_XR_LOAD(2, 0, 2)
  }
#pragma unroll 1
  for (int pass = 0; pass < NP; ++pass) {
    const int nw = pass * 256 + wave * 64;           // the wave's first column of this pass
    active = nw < N;
    // D[column][row]: lane (vi, kb) holds row rt * 32 + vi, register 4 q + i = column 8 q + 4 kb + i of tile t;
    // the accumulators start at the bias
    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias && active) c = *reinterpret_cast<const float4*>(bias + nw + t * 32 + 8 * q + 4 * kb);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          acc[rt][t][4 * q + 0] = c.x; acc[rt][t][4 * q + 1] = c.y;
          acc[rt][t][4 * q + 2] = c.z; acc[rt][t][4 * q + 3] = c.w;
        }
      }
#pragma unroll 1
    for (int ph = 0; ph < NPH; ++ph) {
      if (pass == 0 || NPH > 1) {                    // (re)load the tile; all waves are past their reads of the old one
        if (pass + ph > 0) __syncthreads();
        load_tile(ph);
        __syncthreads();
        OCC_XR_AFRAG(0, 0)
      }
      if (active) {
        const int seg = pass * NPH + ph;
        // segment base offsets in the packed weight: k-steps 16 ph .., column tiles 8 pass ..
        const int cur = ph * 16 * kstep_bytes + pass * (8 * 2048);
        int nph = ph + 1, npass = pass;
        if (nph == NPH) { nph = 0; ++npass; }
        // the ring runs into the next segment (last one: a harmless re-read); a wave idle in the next pass skips it
        const int nxt = (seg + 1 < nseg && npass * 256 + wave * 64 < N) ? nph * 16 * kstep_bytes + npass * (8 * 2048) : cur;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          if (ks + 3 < 16) OCC_XR_LOAD((ks + 3) & 3, cur, ks + 3)
          else OCC_XR_LOAD((ks + 3) & 3, nxt, ks + 3 - 16)
          OCC_XR_AFRAG((ks + 1) & 1, (ks + 1) & 15)
          // small terms first; 4 accumulators between two MFMAs on one accumulator cover the result latency
#pragma unroll
          for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int t = 0; t < 2; ++t)
              acc[rt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[ks & 3][2 * t]), al[ks & 1][rt],
                                                                   acc[rt][t], 0, 0, 0);
#pragma unroll
          for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int t = 0; t < 2; ++t)
              acc[rt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[ks & 3][2 * t + 1]), ah[ks & 1][rt],
                                                                   acc[rt][t], 0, 0, 0);
#pragma unroll
          for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int t = 0; t < 2; ++t)
              acc[rt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[ks & 3][2 * t]), ah[ks & 1][rt],
                                                                   acc[rt][t], 0, 0, 0);
          // pin the software pipeline (hipcc otherwise sinks every ring request down to its use)
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);   // ring requests of step s + 3
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // A fragments of step s + 1
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
      }
    }

    if (!LN) {
      // ---- per-wave epilogue: ReLU, 16 columns of a 32-row tile at a time through the scratch, + residual, stores -----
      if (active) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
              // the residual pieces of this round are requested before the LDS round trip
              float4 res[2];
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                int m = m0 + rt * 32 + (lane >> 2) + 16 * j;
                if (m >= M) m = M - 1;
                res[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (residual) res[j] = *reinterpret_cast<const float4*>(residual + (long)m * ldres + nw + t * 32 + rd * 16 + (lane & 3) * 4);
              }
#pragma unroll
              for (int qq = 0; qq < 2; ++qq) {
                const int q = 2 * rd + qq;
                float4 v = make_float4(acc[rt][t][4 * q + 0], acc[rt][t][4 * q + 1], acc[rt][t][4 * q + 2], acc[rt][t][4 * q + 3]);
                if (act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<float4*>(scratch + vi * kXrPitch + 32 * qq + 16 * kb) = v;
              }
              wave_lds_sync();
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                const int row = (lane >> 2) + 16 * j, piece = lane & 3;
                const int m = m0 + rt * 32 + row;
                float4 v = *reinterpret_cast<const float4*>(scratch + row * kXrPitch + piece * 16);
                v.x += res[j].x; v.y += res[j].y; v.z += res[j].z; v.w += res[j].w;
                if (m < M) *reinterpret_cast<float4*>(out + (long)m * ldo + nw + t * 32 + rd * 16 + piece * 4) = v;
              }
              wave_lds_sync();
            }
      }
    } else {
      // ---- LayerNorm epilogue (one pass): 64 x N tile row-major into the planes' space, 16 full rows per wave ----------
      __syncthreads();                               // every wave is past its last fragment read
      float* sO = reinterpret_cast<float*>(xl);
      if (active) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float4 v = make_float4(acc[rt][t][4 * q + 0], acc[rt][t][4 * q + 1], acc[rt][t][4 * q + 2], acc[rt][t][4 * q + 3]);
              if (act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
              *reinterpret_cast<float4*>(sO + (rt * 32 + vi) * kXrLnPitch + wave * 64 + t * 32 + 8 * q + 4 * kb) = v;
            }
      }
      __syncthreads();
      const int c = lane * 4;
      const bool col_live = c < N;
      float4 gv = make_float4(0.f, 0.f, 0.f, 0.f), bev = gv;
      if (col_live) {
        gv = *reinterpret_cast<const float4*>(ln_g + c);
        bev = *reinterpret_cast<const float4*>(ln_b + c);
      }
      const float inv_n = 1.f / (float)N;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        float4 rres[8];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          int m = m0 + wave * 16 + half * 8 + rr;
          if (m >= M) m = M - 1;
          rres[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (residual != nullptr && col_live) rres[rr] = *reinterpret_cast<const float4*>(residual + (long)m * ldres + c);
        }
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          const int row = wave * 16 + half * 8 + rr;
          const int m = m0 + row;
          if (m >= M) break;                         // wave-uniform
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (col_live) {
            v = *reinterpret_cast<const float4*>(sO + row * kXrLnPitch + c);
            v.x += rres[rr].x; v.y += rres[rr].y; v.z += rres[rr].z; v.w += rres[rr].w;
          }
          const float mean = x3_wave_sum(col_live ? (v.x + v.y) + (v.z + v.w) : 0.f) * inv_n;
          const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
          const float var = x3_wave_sum(col_live ? (dx * dx + dy * dy) + (dz * dz + dw * dw) : 0.f) * inv_n;
          const float rstd = rsqrtf(var + ln_eps);
          v.x = dx * rstd * gv.x + bev.x; v.y = dy * rstd * gv.y + bev.y;
          v.z = dz * rstd * gv.z + bev.z; v.w = dw * rstd * gv.w + bev.w;
          if (col_live) *reinterpret_cast<float4*>(out + (long)m * ldo + c) = v;
        }
      }
    }
  }
#undef OCC_XR_AFRAG
#undef OCC_XR_LOAD
}


// ======================================= dispatch (inside occ_linear_bf16x3_f32) ======================================
#if 0
  // the activation-resident kernel: K in 256-k phases that do not straddle the two segments, whole 64-column wave
  // slices, K = 512 only with one column pass, M >= 8192 rows (below that neither kernel fills the chip; the tiled one
  // has more, smaller blocks).  OCC_LINEAR_RESIDENT=0 (development switch) keeps the tiled kernel.
  static const bool resident_on = [] { const char* e = getenv("OCC_LINEAR_RESIDENT"); return !(e && e[0] == '0'); }();
  const int Kt = K1 + K2;
  if (resident_on && K1 % kXrPhaseK == 0 && K2 % kXrPhaseK == 0 && (Kt == 256 || (Kt == 512 && N <= 256)) &&
      N % 64 == 0 && M >= 8192 && (long)M * 4 < (1L << 31)) {
    auto launch = [&](auto kern) -> hipError_t {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         kXrLdsBytes);
      if (e == hipSuccess)
        hipLaunchKernelGGL(kern, dim3((unsigned)((M + kXrRows - 1) / kXrRows)), dim3(256), kXrLdsBytes, st, a1, (long)lda1,
                           K1, a2, a2_add, (long)lda2, K2, wp, bias, act, residual, (long)ldres, ln_gamma, ln_beta, ln_eps,
                           out, (long)ldo, M, N);
      return e;
    };
    const hipError_t e = ln_gamma ? launch(linear_x3r_kernel<true>) : launch(linear_x3r_kernel<false>);
    if (e != hipSuccess) {
      set_error("linear_bf16x3: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return OCC_E_LAUNCH;
    }
    OCC_CHECK_LAUNCH("linear_bf16x3");
    return OCC_OK;
  }
#endif
