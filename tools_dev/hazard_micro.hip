// Development (round 5): minimal VICTIM kernels for the co-scheduling hazard — which hardware path hands
// a gather kernel wrong data while the library's value-projection kernel runs next to it on another stream?
//   lds_victim : every wave writes a known 32-byte entry per lane into its LDS slab, hands it over with wave_lds_sync()
//                (no waitcnt: the idiom of the gather kernels) and reads its 8-lane group's entries back as broadcast
//                ds_read_b128 — every mismatch is counted;
//   ta_victim  : 8 lanes x 16 B per 128-byte row of a table with known contents, random row per 8-lane group, through
//                BUFFER loads with an SGPR descriptor (the gathers' access shape) — every wrong dword is counted;
//   gl_victim  : one dword per lane at consecutive addresses of a known table through plain global loads (the gathers'
//                logits / offsets reads).
// Not part of libocc_amd.so.  build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools_dev/bin/libhazard_micro.so tools_dev/hazard_micro.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../occnet_amd/csrc/common.h"

namespace {
__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

__global__ __launch_bounds__(256) void fill_kernel(uint32_t* t, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) t[i] = mix((uint32_t)i * 2654435761u + 12345u);
}

__global__ __launch_bounds__(256) void lds_victim(unsigned long long* errors, int iters, int real_wait) {
  __shared__ __attribute__((aligned(16))) occ::SampleParamB slab[4 * 8 * 9];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  occ::SampleParamB* sp = slab + wave * 72;
  const uint32_t id = (blockIdx.x * 4 + wave) * 64 + lane;
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    occ::SampleParamB p;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      p.w[k] = __uint_as_float(mix(id * 8 + k + it * 977u) & 0x3fffffffu);
      p.o[k] = mix(id * 8 + 4 + k + it * 977u);
    }
    sp[(lane >> 3) * 9 + (lane & 7)] = p;
    if (real_wait) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    occ::wave_lds_sync();
    const int g = lane >> 3;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const occ::occ_u32x4 o = *reinterpret_cast<const occ::occ_u32x4*>(sp[g * 9 + s].o);
      const float4 w = *reinterpret_cast<const float4*>(sp[g * 9 + s].w);
      const uint32_t src = (id & ~63u) + g * 8 + s;           // the lane that wrote entry (g, s)
      const float wf[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        bad += __float_as_uint(wf[k]) != (mix(src * 8 + k + it * 977u) & 0x3fffffffu);
        bad += o[k] != mix(src * 8 + 4 + k + it * 977u);
      }
    }
    occ::wave_lds_sync();
  }
  if (bad) atomicAdd(errors, (unsigned long long)bad);
}

__global__ __launch_bounds__(256) void ta_victim(const uint32_t* __restrict__ table, uint32_t n_rows, unsigned long long* errors,
                                                 int iters) {
  const int lane = threadIdx.x & 63, c = lane & 7;
  const uint32_t wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t rs = occ::uniform_rsrc(table, n_rows * 128u);
  unsigned bad = 0;
  for (int it = 0; it < iters; it += 4) {
    float4 v[4];
    uint32_t row[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      row[u] = mix((wid * 8 + (lane >> 3)) * 7919u + (it + u) * 104729u) % n_rows;
      v[u] = occ::buf_load16(rs, row[u] * 128u + c * 16u);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t e0 = row[u] * 32u + c * 4u;
      const float vf[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) bad += __float_as_uint(vf[k]) != mix((e0 + k) * 2654435761u + 12345u);
    }
  }
  if (bad) atomicAdd(errors, (unsigned long long)bad);
}

__global__ __launch_bounds__(256) void gl_victim(const uint32_t* __restrict__ table, uint32_t n, unsigned long long* errors, int iters) {
  const uint32_t wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    const uint32_t base = (mix(wid * 31u + it * 7919u) % (n / 64 - 1)) * 64;
    const uint32_t x = table[base + lane];
    const uint2 y = *reinterpret_cast<const uint2*>(table + ((base + 2 * lane) & ~1u));
    bad += x != mix((base + lane) * 2654435761u + 12345u);
    bad += y.x != mix((((base + 2 * lane) & ~1u)) * 2654435761u + 12345u);
    bad += y.y != mix((((base + 2 * lane) & ~1u) + 1) * 2654435761u + 12345u);
  }
  if (bad) atomicAdd(errors, (unsigned long long)bad);
}
// ds_bpermute (what hipcc makes of __shfl_xor: the gathers' softmax reductions) against the same exchange by DPP quad_perm
template <bool DPP>
__global__ __launch_bounds__(256) void xlane_victim(unsigned long long* errors, int iters) {
  __shared__ float pad[2304];                      // the gathers' LDS footprint (the slab), so that co-residency matches
  const int lane = threadIdx.x & 63;
  const uint32_t id = blockIdx.x * 256 + threadIdx.x;
  if (threadIdx.x == 0) pad[0] = 0.f;
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    const uint32_t mine = mix(id * 31u + it * 7919u);
    uint32_t x1, x2;
    if (DPP) {
      x1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0xb1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
      x2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0x4e, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
    } else {
      x1 = (uint32_t)__shfl_xor((int)mine, 1);
      x2 = (uint32_t)__shfl_xor((int)mine, 2);
    }
    bad += x1 != mix((id ^ 1u) * 31u + it * 7919u);
    bad += x2 != mix((id ^ 2u) * 31u + it * 7919u);
    if (!DPP) {
      const uint32_t x4 = (uint32_t)__shfl_xor((int)mine, 4), x32 = (uint32_t)__shfl_xor((int)mine, 32);
      bad += x4 != mix((id ^ 4u) * 31u + it * 7919u);
      bad += x32 != mix((id ^ 32u) * 31u + it * 7919u);
    }
  }
  if (bad) atomicAdd(errors, (unsigned long long)bad);
  if (lane == 999) pad[lane] = 1.f;
}
// producer -> consumer across a kernel boundary of ONE stream: the producer writes table[i] = f(i, epoch) (16-byte buffer
// stores like the chain kernels' row stores, or plain global stores), the NEXT kernel reads every word back (the gathers'
// dword / 16-byte loads) and counts words that are not this epoch's — a stale or not-yet-visible line shows up as last
// epoch's value
template <bool BUF>
__global__ __launch_bounds__(256) void produce_kernel(uint32_t* __restrict__ table, uint32_t n16, uint32_t epoch) {
  const __amdgpu_buffer_rsrc_t rs = occ::uniform_rsrc(table, n16 * 16u);
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) {
    occ::occ_u32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = mix((i * 4 + k) * 2654435761u + epoch * 40503u);
    if (BUF) __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(i * 16u), 0, 0);
    else *reinterpret_cast<occ::occ_u32x4*>(table + (size_t)i * 4) = v;
  }
}
template <bool BUF>
__global__ __launch_bounds__(256) void consume_kernel(const uint32_t* __restrict__ table, uint32_t n16, uint32_t epoch,
                                                      unsigned long long* errors) {
  const __amdgpu_buffer_rsrc_t rs = occ::uniform_rsrc(table, n16 * 16u);
  unsigned bad = 0, stale = 0;
  // a different thread -> element map than the producer's (another CU / XCD reads what one wrote)
  for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n16; j += gridDim.x * 256) {
    const uint32_t i = (j * 2654435761u) % n16;
    uint32_t w[4];
    if (BUF) {
      const float4 v = occ::buf_load16(rs, i * 16u);
      w[0] = __float_as_uint(v.x); w[1] = __float_as_uint(v.y); w[2] = __float_as_uint(v.z); w[3] = __float_as_uint(v.w);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) w[k] = table[(size_t)i * 4 + k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      bad += w[k] != mix((i * 4 + k) * 2654435761u + epoch * 40503u);
      stale += w[k] == mix((i * 4 + k) * 2654435761u + (epoch - 1) * 40503u);
    }
  }
  if (bad) { atomicAdd(errors, (unsigned long long)bad); atomicAdd(errors + 1, (unsigned long long)stale); }
}
// pure register arithmetic with results known in advance: which VALU instruction class hands a wave a wrong result while
// another kernel's MFMAs run on the same SIMDs?  KIND 0: packed fp32 FMA (v_pk_fma_f32: what hipcc makes of the gathers'
// float4 accumulation), 1: scalar v_fma_f32 (inline asm), 2: 32-bit integer multiply-add (address arithmetic), 3: v_exp_f32
// / v_rcp_f32, 4: v_fma_mix_f32 (the fp16-row gather's accumulate).  Small integers in float: every result is exact.
template <int KIND>
__global__ __launch_bounds__(256) void valu_victim(unsigned long long* errors, int iters) {
  const uint32_t id = blockIdx.x * 256 + threadIdx.x;
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    const uint32_t h = mix(id * 131u + it * 7919u);
    if (KIND == 0 || KIND == 1) {
      float4 acc = make_float4(1.f, 2.f, 3.f, 4.f);
      const float w = (float)(h & 7u), v = (float)((h >> 3) & 15u);
      float4 vv = make_float4(v, v + 1.f, v + 2.f, v + 3.f);
      asm volatile("" : "+v"(vv.x), "+v"(vv.y), "+v"(vv.z), "+v"(vv.w), "+v"(acc.x), "+v"(acc.y), "+v"(acc.z), "+v"(acc.w));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (KIND == 0) {
          occ::fma4(acc, w, vv);
        } else {
          asm volatile("v_fma_f32 %0, %4, %5, %0\n\tv_fma_f32 %1, %4, %6, %1\n\tv_fma_f32 %2, %4, %7, %2\n\tv_fma_f32 %3, %4, %8, %3"
                       : "+v"(acc.x), "+v"(acc.y), "+v"(acc.z), "+v"(acc.w) : "v"(w), "v"(vv.x), "v"(vv.y), "v"(vv.z), "v"(vv.w));
        }
      }
      bad += acc.x != 1.f + 16.f * w * v;
      bad += acc.y != 2.f + 16.f * w * (v + 1.f);
      bad += acc.z != 3.f + 16.f * w * (v + 2.f);
      bad += acc.w != 4.f + 16.f * w * (v + 3.f);
    } else if (KIND == 2) {
      uint32_t a = h & 0xffffu, b = (h >> 16) | 1u, c = id;
      asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
      uint32_t r = c;
#pragma unroll
      for (int k = 0; k < 16; ++k) r = r * b + a;
      uint32_t e = id, eb = (h >> 16) | 1u, ea = h & 0xffffu;
      asm volatile("" : "+v"(e), "+v"(eb), "+v"(ea));  // (the reference chain runs on the same unit: two independently
      for (int k = 0; k < 16; ++k) e = e * eb + ea;    //  scheduled evaluations are compared)
      bad += r != e;
    } else if (KIND == 3) {
      float x = (float)(h & 7u);
      asm volatile("" : "+v"(x));
      const float e2 = __builtin_amdgcn_exp2f(x);           // exact powers of two
      const float rc = __builtin_amdgcn_rcpf(e2);
      bad += e2 != (float)(1u << (h & 7u));
      bad += rc != 1.f / (float)(1u << (h & 7u));
    } else if (KIND == 6) {
      // the set-up arithmetic of the gathers as the FAILING build of tsa_tile_kernel compiled it: two IEEE divisions whose
      // v_div_fixup results are consumed at once by PACKED fp32 ops (v_pk_add_f32 / v_pk_fma_f32) — loc = ref + o / (W, H),
      // (h_im, w_im) = loc * (H, W) - 0.5.  Operands chosen so that every step is exact.
      float W = (float)(128 + (h & 64u)), Hh = (float)(64 + ((h >> 7) & 64u));          // 128 or 192; 64 or 128
      float kx = (float)((h >> 8) & 31u), ky = (float)((h >> 13) & 31u);
      float rx = 0.25f + 0.25f * (float)((h >> 18) & 1u), ry = 0.125f * (float)(((h >> 19) & 3u) + 1u);
      asm volatile("" : "+v"(W), "+v"(Hh), "+v"(kx), "+v"(ky), "+v"(rx), "+v"(ry));
      const float ox = kx * W * 0.0078125f, oy = ky * Hh * 0.0078125f;     // o = k * size / 128: o / size = k / 128, exact
      const float lx = rx + ox / W, ly = ry + oy / Hh;
      const float w_im = lx * W - 0.5f, h_im = ly * Hh - 0.5f;
      const float ex = (rx + kx * 0.0078125f) * W - 0.5f, ey = (ry + ky * 0.0078125f) * Hh - 0.5f;
      float e1 = ex, e2 = ey;
      asm volatile("" : "+v"(e1), "+v"(e2));
      bad += w_im != e1;
      bad += h_im != e2;
    } else if (KIND == 5) {
      // IEEE fp32 division (v_div_scale / v_rcp / v_div_fmas / v_div_fixup: v_div_fmas reads VCC implicitly), the way the
      // gathers normalise their offsets and softmax weights; a compare-and-select in front leaves a non-trivial VCC
      float k = (float)((h & 63u) + 1u), bq = (float)(((h >> 6) & 1023u) + 1u), a2 = (float)((h >> 16) & 255u);
      asm volatile("" : "+v"(k), "+v"(bq), "+v"(a2));
      const float num = k * bq;                             // exact: < 2^16
      const float sel = a2 > 100.f ? num : num + 0.f;
      const float q1 = sel / bq;
      const float q2 = (a2 + 1.f) / (a2 + 1.f);
      const float q3 = (k * 3.f) / 3.f;
      bad += q1 != k;
      bad += q2 != 1.f;
      bad += q3 != k;
    } else {
      float acc = 1.f;
      const float w = (float)(h & 7u);
      unsigned packed = 0x40003c00u;                        // fp16 (1.0, 2.0)
      asm volatile("" : "+v"(packed), "+v"(acc));
#pragma unroll
      for (int r = 0; r < 8; ++r) { occ::fma_mix_lo(acc, w, packed); occ::fma_mix_hi(acc, w, packed); }
      bad += acc != 1.f + 8.f * w * 3.f;
    }
  }
  if (bad) atomicAdd(errors, (unsigned long long)bad);
}
// The TSA gather's inner structure with known answers: per-sample parameters (4 weights, 4 row offsets) written to the
// per-wave LDS slab by the set-up lane, handed over with wave_lds_sync(), read back as broadcast ds_read_b128 by the 8 lanes
// of a group, 4 buffer loads of 16 bytes per sample into registers, float4 FMAs — LDS returns and vector-memory returns
// interleaved as densely as in the real kernel.  Table word (row, j) = small integer f(row, j), weights 0..3: every sum is
// exact, the expected value is recomputed from the indices alone.  LDSPARAMS = false keeps the parameters in registers
// (each lane recomputes its group's samples): the same memory traffic without the slab.
__device__ __forceinline__ float tab_f(uint32_t row, uint32_t j) { return (float)(mix(row * 32u + j) & 15u); }
__global__ __launch_bounds__(256) void fill_small_kernel(float* t, uint32_t n_rows) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_rows * 32u; i += gridDim.x * 256) t[i] = tab_f(i >> 5, i & 31u);
}
template <bool LDSPARAMS>
__global__ __launch_bounds__(256) void gather_victim(const float* __restrict__ table, uint32_t n_rows, unsigned long long* errors,
                                                     int iters) {
  __shared__ __attribute__((aligned(16))) occ::SampleParamB slab[4 * 8 * 9];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 3, c = lane & 7;
  occ::SampleParamB* sp = slab + wave * 72;
  const uint32_t wid = blockIdx.x * 4 + wave;
  const __amdgpu_buffer_rsrc_t rs = occ::uniform_rsrc(table, n_rows * 128u);
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    auto param = [&](int gg, int ss, occ::SampleParamB& p) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t h = mix(((wid * 8 + gg) * 8 + ss) * 4 + k + it * 1000003u);
        p.w[k] = (float)(h & 3u);
        p.o[k] = ((h >> 2) % n_rows) * 128u;
      }
    };
    if (LDSPARAMS) {
      occ::SampleParamB p;
      param(g, c, p);                       // set-up lane (g, s = c)
      sp[g * 9 + c] = p;
      occ::wave_lds_sync();
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), exp4 = acc;
    for (int s0 = 0; s0 < 8; s0 += 4) {
      float4 v[4][4];
      float4 w[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        occ::SampleParamB p;
        if (LDSPARAMS) {
          const occ::occ_u32x4 o = *reinterpret_cast<const occ::occ_u32x4*>(sp[g * 9 + s0 + u].o);
          w[u] = *reinterpret_cast<const float4*>(sp[g * 9 + s0 + u].w);
          p.o[0] = o[0]; p.o[1] = o[1]; p.o[2] = o[2]; p.o[3] = o[3];
        } else {
          param(g, s0 + u, p);
          w[u] = make_float4(p.w[0], p.w[1], p.w[2], p.w[3]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) v[u][k] = occ::buf_load16(rs, p.o[k] + c * 16u);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        occ::fma4(acc, w[u].x, v[u][0]); occ::fma4(acc, w[u].y, v[u][1]); occ::fma4(acc, w[u].z, v[u][2]); occ::fma4(acc, w[u].w, v[u][3]);
      }
    }
    for (int ss = 0; ss < 8; ++ss) {
      occ::SampleParamB p;
      param(g, ss, p);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t row = p.o[k] / 128u;
        exp4.x += p.w[k] * tab_f(row, c * 4 + 0); exp4.y += p.w[k] * tab_f(row, c * 4 + 1);
        exp4.z += p.w[k] * tab_f(row, c * 4 + 2); exp4.w += p.w[k] * tab_f(row, c * 4 + 3);
      }
    }
    bad += (acc.x != exp4.x) + (acc.y != exp4.y) + (acc.z != exp4.z) + (acc.w != exp4.w);
    if (LDSPARAMS) occ::wave_lds_sync();
  }
  if (bad) atomicAdd(errors, (unsigned long long)bad);
}

// The gathers' OUTPUT side: one wave per query row, the row written as ONE 64-lane x 16-byte store after a short gather-like
// prologue (a few buffer loads, so that the wave's timing resembles the real kernels'); a second kernel (after the first has
// completed) reads every row back.  Rows carry (row, word, epoch): a lost store shows as LAST epoch's value.
__global__ __launch_bounds__(256) void store_victim(const float* __restrict__ table, uint32_t n_tab_rows, uint32_t* __restrict__ out,
                                                    uint32_t n_rows, uint32_t epoch) {
  const int lane = threadIdx.x & 63;
  const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const __amdgpu_buffer_rsrc_t rs = occ::uniform_rsrc(table, n_tab_rows * 128u);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const uint32_t r = mix(row * 8u + u + epoch * 977u) % n_tab_rows;
    const float4 v = occ::buf_load16(rs, r * 128u + (lane & 7) * 16u);
    occ::fma4(acc, 1.f, v);
  }
  const uint32_t salt = (acc.x + acc.y + acc.z + acc.w) < 0.f ? 1u : 0u;      // always 0 (the table is non-negative)
  occ::occ_u32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = mix((row * 256u + lane * 4u + k) * 2654435761u + epoch * 40503u) + salt;
  *reinterpret_cast<occ::occ_u32x4*>(out + (size_t)row * 256 + lane * 4) = o;
}
__global__ __launch_bounds__(256) void store_check(const uint32_t* __restrict__ out, uint32_t n_rows, uint32_t epoch,
                                                   unsigned long long* errors) {
  unsigned bad = 0, stale = 0;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_rows * 256u; i += gridDim.x * 256) {
    const uint32_t w = out[i];
    bad += w != mix(i * 2654435761u + epoch * 40503u);
    stale += w == mix(i * 2654435761u + (epoch - 1) * 40503u);
  }
  if (bad) { atomicAdd(errors, (unsigned long long)bad); atomicAdd(errors + 1, (unsigned long long)stale); }
}

// THE REPRODUCER (round 5): the TSA gather as it was BEFORE occ::fdiv — its softmax weight and its two offset normalisations
// are plain fp32 divisions, which hipcc expands into v_div_scale / v_rcp / v_div_fmas / v_div_fixup — on a value map of ones:
// every interior query's output must be 1 (softmax and bilinear weights sum to one), whatever the offsets and logits.  Counts
// the output words that are off by more than 1e-3.  USE_FDIV = true is the same kernel with the three quotients computed by
// occ::fdiv.  (next to an MFMA-issuing kernel on another stream the `/` build returns wrong weights in
// lanes 48-63.)
// bilinear_setup_b without lane masks in scalar registers: every condition becomes a 0 / 1 VGPR at once (v_cmp + v_cndmask,
// VCC consumed by the very next VALU instruction), conditions are combined with VALU integer ANDs and the selects compare those
// integers — no s_and_b64 / s_and_saveexec_b64 / branch ever reads a mask a v_cmp has just written.
__device__ __forceinline__ void bilinear_setup_vb(float loc_x, float loc_y, float attn, int H, int W, unsigned pix_bytes,
                                                  unsigned dead, occ::SampleParamB& sp) {
  const float h_im = loc_y * (float)H - 0.5f, w_im = loc_x * (float)W - 0.5f;
  auto flag = [](bool c) { int f = c ? 1 : 0; asm volatile("" : "+v"(f)); return f; };
  const int adm = flag(h_im > -1.f) & flag(w_im > -1.f) & flag(h_im < (float)H) & flag(w_im < (float)W);
  const float hf = floorf(h_im), wf = floorf(w_im);
  const int h_low = (int)hf, w_low = (int)wf;
  const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
  const int t = flag(h_low >= 0) & adm, b = flag(h_low + 1 <= H - 1) & adm, l = flag(w_low >= 0), r = flag(w_low + 1 <= W - 1);
  const int base = h_low * W + w_low;
  const int c0 = t & l, c1 = t & r, c2 = b & l, c3 = b & r;
  sp.w[0] = c0 ? hh * hw * attn : 0.f; sp.o[0] = c0 ? (unsigned)base * pix_bytes : dead;
  sp.w[1] = c1 ? hh * lw * attn : 0.f; sp.o[1] = c1 ? (unsigned)(base + 1) * pix_bytes : dead;
  sp.w[2] = c2 ? lh * hw * attn : 0.f; sp.o[2] = c2 ? (unsigned)(base + W) * pix_bytes : dead;
  sp.w[3] = c3 ? lh * lw * attn : 0.f; sp.o[3] = c3 ? (unsigned)(base + W + 1) * pix_bytes : dead;
}

// VAR (ablations of the victim): 64 = bilinear_setup_vb (no scalar lane masks); 1 quotients by occ::fdiv; 2 bare v_exp_f32 instead of expf; 4 no softmax at all (weight 1/4:
// no shuffles, no exponential, no quotient); 8 offsets forced to zero (every sample on the query's own pixel centre);
// 16 s_nop 7 x 2 between the set-up arithmetic and the LDS hand-over; 32 the per-sample terms handed over with a real
// s_waitcnt + block barrier instead of the wave-level fence
template <int VAR>
__global__ __launch_bounds__(256) void tsa_div_victim(const float* __restrict__ value, const float* __restrict__ offs,
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
  constexpr bool USE_FDIV = (VAR & 1) != 0;
  const float e = (VAR & 2) ? __builtin_amdgcn_exp2f((x - mx) * 1.44269504f) : expf(x - mx);
  float sum = e + __shfl_xor(e, 1);
  sum += __shfl_xor(sum, 2);
  float aw = USE_FDIV ? occ::fdiv(e, sum) : e / sum;
  if (VAR & 4) aw = 0.25f;
  float2 o = *reinterpret_cast<const float2*>(offs + q * 128 + 2 * lane);
  if (VAR & 8) o = make_float2(0.f, 0.f);
  const int qy = (int)(q / bev_w), qx = (int)(q - (long)qy * bev_w);
  const float2 rf = USE_FDIV ? make_float2(occ::fdiv((float)qx + 0.5f, (float)bev_w), occ::fdiv((float)qy + 0.5f, (float)bev_h))
                             : make_float2(((float)qx + 0.5f) / (float)bev_w, ((float)qy + 0.5f) / (float)bev_h);
  const float ox = USE_FDIV ? occ::fdiv(o.x, (float)bev_w) : o.x / (float)bev_w;
  const float oy = USE_FDIV ? occ::fdiv(o.y, (float)bev_h) : o.y / (float)bev_h;
  occ::SampleParamB p;
  if (VAR & 64) bilinear_setup_vb(rf.x + ox, rf.y + oy, aw, bev_h, bev_w, (unsigned)row_stride * 4u, occ::kOobOffset, p);
  else occ::bilinear_setup_b(rf.x + ox, rf.y + oy, aw, bev_h, bev_w, 0, (unsigned)row_stride * 4u, occ::kOobOffset, true, p);
  if (VAR & 16) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(p.w[0]), "+v"(p.w[1]), "+v"(p.w[2]), "+v"(p.w[3]));
  sp[m * NSp + (lane & 7)] = p;
  if (VAR & 32) __syncthreads(); else occ::wave_lds_sync();
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

// Synthetic AGGRESSORS (round 5): which ingredient of the library's MFMA kernels disturbs a co-resident gather wave?
//   KIND 0: nothing but matrix-core instructions on registers (v_mfma_f32_32x32x16_bf16, 8 independent accumulators,
//           back to back — the issue pattern of the chain / projection kernels, no memory traffic at all);
//   KIND 1: the same + LDS fragment reads (ds_read_b128 out of a 64 KB dynamic tile) between the MFMAs;
//   KIND 2: the LDS reads alone;  KIND 3: packed-fp32 VALU FMAs alone.
typedef float hz_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 hz_bf16x8 __attribute__((ext_vector_type(8)));
template <int KIND>
__global__ __launch_bounds__(256, 2) void spin_kernel(float* __restrict__ sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) char tile[];
  const int lane = threadIdx.x & 63;
  if (KIND == 1 || KIND == 2)
    for (int i = threadIdx.x; i < 65536 / 16; i += 256) reinterpret_cast<float4*>(tile)[i] = make_float4(1.f, 2.f, 3.f, 4.f);
  __syncthreads();
  hz_f32x16 acc[8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[a][k] = 0.f;
  occ::occ_u32x4 fa = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, fb = fa;     // bf16 1.0 x 8
  float4 v = make_float4(1.f, 1.f, 1.f, 1.f), w = make_float4(0.5f, 0.25f, 0.125f, 0.0625f);
  for (int it = 0; it < iters; ++it) {
    if (KIND == 1 || KIND == 2) {
      const float4 r = *reinterpret_cast<const float4*>(tile + (((lane * 16 + it * 1040) & 65535) & ~15));
      fa[0] ^= __float_as_uint(r.x) & 1u;
      fa[1] ^= __float_as_uint(r.z) & 1u;
      v.x += (r.y + r.w) * 1e-30f;
    }
    if (KIND == 0 || KIND == 1) {
#pragma unroll
      for (int a = 0; a < 8; ++a)
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(hz_bf16x8, fa), __builtin_bit_cast(hz_bf16x8, fb),
                                                         acc[a], 0, 0, 0);
    }
    if (KIND == 3) {
#pragma unroll
      for (int r = 0; r < 16; ++r) occ::fma4(v, 0.999f, w);
    }
  }
  float sum = v.x + v.y + v.z + v.w;
#pragma unroll
  for (int a = 0; a < 8; ++a) sum += acc[a][0] + acc[a][7];
  if (sum == 123.456f) sink[0] = sum;               // never true: keeps the work alive
}
}  // namespace

extern "C" void hz_fill_small(float* t, uint32_t n_rows, void* stream) {
  hipLaunchKernelGGL(fill_small_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, t, n_rows);
}
extern "C" void hz_gather_victim(const float* table, uint32_t n_rows, unsigned long long* errors, int blocks, int iters,
                                 int lds_params, void* stream) {
  if (lds_params) hipLaunchKernelGGL(gather_victim<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, table, n_rows, errors, iters);
  else hipLaunchKernelGGL(gather_victim<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, table, n_rows, errors, iters);
}
extern "C" void hz_store_victim(const float* table, uint32_t n_tab_rows, uint32_t* out, uint32_t n_rows, uint32_t epoch,
                                unsigned long long* errors, int check, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!check) hipLaunchKernelGGL(store_victim, dim3((n_rows + 3) / 4), dim3(256), 0, st, table, n_tab_rows, out, n_rows, epoch);
  else hipLaunchKernelGGL(store_check, dim3(2048), dim3(256), 0, st, out, n_rows, epoch, errors);
}
// The suspected pattern in isolation: a lane mask written by v_cmp into an SGPR pair / VCC and read
// by the SCALAR unit in the very next instruction (s_and_b64, s_and_saveexec_b64), as hipcc compiles the gathers' admission test.
// MODE 0: the masks are ANDed on the SALU and the result selects through v_cndmask; MODE 1: ... and gates a v_mov through
// s_and_saveexec_b64 (the branch form); MODE 2: the same test in C (whatever hipcc makes of it).  The expected bit comes from
// shifts and adds on the integer the operands were made from — no comparison, no mask.
template <int MODE>
__global__ __launch_bounds__(256) void mask_victim(unsigned long long* errors, int iters) {
  const uint32_t id = blockIdx.x * 256 + threadIdx.x;
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    const uint32_t h = mix(id * 131u + it * 7919u);
    const uint32_t xi = h & 15u, yi = (h >> 4) & 15u;               // x = xi - 4 in -4 .. 11, y likewise
    float x = (float)xi - 4.f, y = (float)yi - 4.f, lx = 8.f, ly = 6.f;
    asm volatile("" : "+v"(x), "+v"(y), "+v"(lx), "+v"(ly));
    // x > -1  <=>  xi >= 4;   x < 8  <=>  xi < 12;   y > -1  <=>  yi >= 4;   y < 6  <=>  yi < 10
    const unsigned expect = ((xi + 12u) >> 4) & (1u - ((xi + 4u) >> 4)) & ((yi + 12u) >> 4) & (1u - ((yi + 6u) >> 4));
    unsigned r;
    if (MODE == 0) {
      unsigned long long m;
      asm volatile("v_cmp_lt_f32_e32 vcc, -1.0, %2\n\t"
                   "v_cmp_lt_f32_e64 %0, -1.0, %3\n\t"
                   "s_and_b64 %0, vcc, %0\n\t"
                   "v_cmp_lt_f32_e32 vcc, %2, %4\n\t"
                   "s_and_b64 %0, vcc, %0\n\t"
                   "v_cmp_lt_f32_e32 vcc, %3, %5\n\t"
                   "s_and_b64 %0, vcc, %0\n\t"
                   "v_cndmask_b32_e64 %1, 0, 1, %0"
                   : "=&s"(m), "=&v"(r) : "v"(x), "v"(y), "v"(lx), "v"(ly) : "vcc");
    } else if (MODE == 1) {
      unsigned long long m, sv;
      asm volatile("v_cmp_lt_f32_e32 vcc, -1.0, %3\n\t"
                   "v_cmp_lt_f32_e64 %0, -1.0, %4\n\t"
                   "s_and_b64 %0, vcc, %0\n\t"
                   "v_cmp_lt_f32_e32 vcc, %3, %5\n\t"
                   "s_and_b64 %0, vcc, %0\n\t"
                   "v_cmp_lt_f32_e32 vcc, %4, %6\n\t"
                   "s_and_b64 %0, vcc, %0\n\t"
                   "v_mov_b32 %2, 0\n\t"
                   "s_and_saveexec_b64 %1, %0\n\t"
                   "v_mov_b32 %2, 1\n\t"
                   "s_or_b64 exec, exec, %1"
                   : "=&s"(m), "=&s"(sv), "=&v"(r) : "v"(x), "v"(y), "v"(lx), "v"(ly) : "vcc");
    } else {
      r = 0;
      if (x > -1.f && y > -1.f && x < lx && y < ly) r = 1;
      asm volatile("" : "+v"(r));
    }
    bad += r ^ expect;
  }
  if (bad) {
    atomicAdd(errors, (unsigned long long)bad);
    atomicAdd(errors + 1 + ((threadIdx.x & 63) >> 4), (unsigned long long)bad);
  }
}
extern "C" void hz_mask_victim(unsigned long long* errors, int blocks, int iters, int mode, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (mode) {
    case 0: hipLaunchKernelGGL(mask_victim<0>, dim3(blocks), dim3(256), 0, st, errors, iters); break;
    case 1: hipLaunchKernelGGL(mask_victim<1>, dim3(blocks), dim3(256), 0, st, errors, iters); break;
    default: hipLaunchKernelGGL(mask_victim<2>, dim3(blocks), dim3(256), 0, st, errors, iters); break;
  }
}
extern "C" void hz_tsa_div_victim(const float* value, const float* offs, const float* logits, unsigned long long* errors, int bev_h,
                                  int bev_w, int use_fdiv, void* stream) {
  const int blocks = (bev_h * bev_w + 3) / 4;
#define HZ_V(V) case V: hipLaunchKernelGGL(tsa_div_victim<V>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, value, offs, logits, errors, bev_h, bev_w); break;
  switch (use_fdiv) { HZ_V(0) HZ_V(1) HZ_V(2) HZ_V(4) HZ_V(8) HZ_V(12) HZ_V(16) HZ_V(32) HZ_V(5) HZ_V(64) HZ_V(65) HZ_V(76) default: break; }
#undef HZ_V
}
extern "C" void hz_spin(float* sink, int kind, int blocks, int iters, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int lds = (kind == 1 || kind == 2) ? 65536 : 0;
  switch (kind) {
    case 0: hipLaunchKernelGGL(spin_kernel<0>, dim3(blocks), dim3(256), lds, st, sink, iters); break;
    case 1: (void)hipFuncSetAttribute(reinterpret_cast<const void*>(spin_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
            hipLaunchKernelGGL(spin_kernel<1>, dim3(blocks), dim3(256), lds, st, sink, iters); break;
    case 2: (void)hipFuncSetAttribute(reinterpret_cast<const void*>(spin_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
            hipLaunchKernelGGL(spin_kernel<2>, dim3(blocks), dim3(256), lds, st, sink, iters); break;
    default: hipLaunchKernelGGL(spin_kernel<3>, dim3(blocks), dim3(256), lds, st, sink, iters); break;
  }
}
extern "C" void hz_valu_victim(unsigned long long* errors, int blocks, int iters, int kind, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (kind) {
    case 0: hipLaunchKernelGGL(valu_victim<0>, dim3(blocks), dim3(256), 0, st, errors, iters); break;
    case 1: hipLaunchKernelGGL(valu_victim<1>, dim3(blocks), dim3(256), 0, st, errors, iters); break;
    case 2: hipLaunchKernelGGL(valu_victim<2>, dim3(blocks), dim3(256), 0, st, errors, iters); break;
    case 3: hipLaunchKernelGGL(valu_victim<3>, dim3(blocks), dim3(256), 0, st, errors, iters); break;
    case 5: hipLaunchKernelGGL(valu_victim<5>, dim3(blocks), dim3(256), 0, st, errors, iters); break;
    case 6: hipLaunchKernelGGL(valu_victim<6>, dim3(blocks), dim3(256), 0, st, errors, iters); break;
    default: hipLaunchKernelGGL(valu_victim<4>, dim3(blocks), dim3(256), 0, st, errors, iters); break;
  }
}
extern "C" void hz_produce_consume(uint32_t* table, uint32_t n16, uint32_t epoch, unsigned long long* errors, int buf_store,
                                   int buf_load, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (buf_store) hipLaunchKernelGGL(produce_kernel<true>, dim3(1024), dim3(256), 0, st, table, n16, epoch);
  else hipLaunchKernelGGL(produce_kernel<false>, dim3(1024), dim3(256), 0, st, table, n16, epoch);
  if (buf_load) hipLaunchKernelGGL(consume_kernel<true>, dim3(2048), dim3(256), 0, st, table, n16, epoch, errors);
  else hipLaunchKernelGGL(consume_kernel<false>, dim3(2048), dim3(256), 0, st, table, n16, epoch, errors);
}
extern "C" void hz_xlane_victim(unsigned long long* errors, int blocks, int iters, int dpp, void* stream) {
  if (dpp) hipLaunchKernelGGL(xlane_victim<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, errors, iters);
  else hipLaunchKernelGGL(xlane_victim<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, errors, iters);
}
extern "C" void hz_fill(uint32_t* t, long n, void* stream) {
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, t, n);
}
extern "C" void hz_lds_victim(unsigned long long* errors, int blocks, int iters, int real_wait, void* stream) {
  hipLaunchKernelGGL(lds_victim, dim3(blocks), dim3(256), 0, (hipStream_t)stream, errors, iters, real_wait);
}
extern "C" void hz_ta_victim(const uint32_t* table, uint32_t n_rows, unsigned long long* errors, int blocks, int iters, void* stream) {
  hipLaunchKernelGGL(ta_victim, dim3(blocks), dim3(256), 0, (hipStream_t)stream, table, n_rows, errors, iters);
}
extern "C" void hz_gl_victim(const uint32_t* table, uint32_t n, unsigned long long* errors, int blocks, int iters, void* stream) {
  hipLaunchKernelGGL(gl_victim, dim3(blocks), dim3(256), 0, (hipStream_t)stream, table, n, errors, iters);
}
