// Multi-scale deformable attention backward for gfx950 — the drop-in for mmcv's
// `ms_deform_attn_backward` (call sites: projects/mmdet3d_plugin/bevformer/modules/
// multi_scale_deformable_attn_function.py:74-84,150-160).  Arithmetic of mmcv's
// ms_deformable_col2im, per sample (b,q,m,l,p) with top = grad_out[b,q,m,:]:
//   grad_attn      = sum_c top_c * bilinear_c
//   grad_loc_x     = W * sum_c top_c * attn * (hh*(v2-v1) + lh*(v4-v3))      (only in-range corners)
//   grad_loc_y     = H * sum_c top_c * attn * (hw*(v3-v1) + lw*(v4-v2))
//   grad_value[k] += top_c * attn * w_k                                     (atomic, in-range corners)
// The three grad tensors arrive pre-zeroed (the caller's contract); grad_value is accumulated, the other two
// are overwritten (each sample has exactly one writer), as mmcv does.
//
// grad_value without floating-point atomics (D == 32, default).  gfx950 retires float atomic adds at ~82 G/s
// device-wide, in global memory and in LDS alike, whatever the access pattern (tools_dev/bwd_probe.py): the
// 1.8 G dword atomics of one SCA launch cost 21 ms while everything else in the kernel costs 0.6 ms.  So the
// scatter becomes a counting sort + owner-computes gather:
//   1. the per-sample kernel (atomics compiled out) writes grad_loc / grad_attn and flags the items whose output
//      gradient is non-zero;
//   2. every value map is cut into BINS of 32 consecutive pixels; a sample contributes one "row item" per bilinear
//      row (left + right corner weights, already multiplied by the attention weight) to the bin of its left
//      pixel (two items when the pair straddles a bin edge).  COUNT -> exclusive SCAN -> FILL writes the items
//      (12 bytes: query, pixel in bin, two weights) bin by bin.  Both passes run as 1024-thread blocks that own 1024
//      consecutive (query, point) samples of one (batch, head, level) and keep that level's bins as a dense
//      histogram in LDS: ONE global integer atomic per block and touched bin (msda_bwd_bin_block_kernel).  The
//      round-1 form (one atomic per wave and bin, msda_bwd_bin_kernel, OCC_MSDA_BWD_BIN=wave) sent thousands of
//      device-scope atomics to the same counter of a coarse-level bin; those retire one round trip at a time and
//      were 20 of the backward's 34 ms per training step;
//   3. the SCAN also writes the replay WORK LIST: a non-empty bin = ceil(count / 2048) entries;
//   4. REPLAY: one block per work entry: 32 pixels x 32 channels accumulators in LDS, private per half-wave (lane =
//      channel, the items dealt round-robin to the 8 half-waves), plain read-add-write — no atomics, no conflicts;
//      item records and output-gradient rows are prefetched two / one step ahead; the 8 copies are summed and added
//      to grad_value (plain read-add-write when the block is the bin's only owner, float atomics for the pieces
//      of a split bin: a few dozen blocks at most).
//
// Decomposition (D == 32): 8 lanes x 4 channels per (b,q,m) item, 8 items per wave — the forward's
// layout, so a corner is one 128-byte row per group; the channel sums are 3-step DPP/shuffle
// reductions inside the 8-lane group.  Other D: one thread per (item, channel), mmcv's own shape,
// with the channel sums reduced through LDS.
#include <cstdlib>
#include <string>
#include "common.h"

namespace occ {

__device__ __forceinline__ float group8_sum(float v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  return v;
}

template <bool VALUE_ATOMICS>
__global__ __launch_bounds__(256) void msda_bwd_d32_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lstart, const float* __restrict__ loc,
    const float* __restrict__ attn, const float* __restrict__ grad_out,
    float* __restrict__ grad_value, float* __restrict__ grad_loc, float* __restrict__ grad_attn,
    unsigned char* __restrict__ nzflag, int S, int M, int L, int Lq, int P, long n_items) {
  constexpr int D = 32;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long item = gid >> 3;
  const int c4 = (int)(gid & 7);
  if (item >= n_items) return;   // whole 8-lane groups leave together; shuffles stay inside a group
  const int m = (int)(item % M);
  const long b = item / ((long)M * Lq);
  const long row_stride = (long)M * D;
  const long voff = b * (long)S * row_stride + (long)m * D + c4 * 4;
  const float* vb = value + voff;
  float* gvb = grad_value + voff;
  const float4 top = *reinterpret_cast<const float4*>(grad_out + item * D + c4 * 4);
  // an item whose 32 output gradients are all zero (the padded rebatch rows of SpatialCrossAttention's
  // autograd path, a quarter of all rows) contributes nothing: the caller pre-zeroed the three grad tensors
  const bool nz = top.x != 0.f || top.y != 0.f || top.z != 0.f || top.w != 0.f;
  const unsigned long long any = __ballot(nz);
  if (((any >> (threadIdx.x & 56)) & 0xffull) == 0ull) return;
  if (!VALUE_ATOMICS && c4 == 0) nzflag[item] = 1;     // the binning kernels skip the other items
  const int LP = L * P;
  if (!VALUE_ATOMICS) {
    // four samples per step, their 16 corner rows requested together: every load is unconditional (a corner outside
    // the map reads row 0 of the batch entry and is replaced by 0 afterwards), so nothing separates the loads of
    // different samples.  One sample per step left 4 rows in flight per lane: 0.9 ms per SCA launch, latency-bound.
    constexpr int U = 4;
    for (int s0 = 0; s0 < LP; s0 += U) {
      float4 v[U][4];
      float lh_[U], lw_[U], a_[U], Hf[U], Wf[U];
      int ok[U][4];                            // 0 / 1 lane flags in VGPRs (common.h: lane_flag)
      bool live[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int s = s0 + u < LP ? s0 + u : LP - 1;
        live[u] = s0 + u < LP;
        const int l = s / P;
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const long st = lstart[l];
        const long si = item * LP + s;
        const float2 xy = *reinterpret_cast<const float2*>(loc + si * 2);
        a_[u] = attn[si];
        Hf[u] = (float)H; Wf[u] = (float)W;
        const BilinearTerms t = bilinear_terms(xy.x, xy.y, H, W, lane_flag(live[u]));
        lh_[u] = t.adm ? t.lh : 0.f;          // a non-finite location must give 0, not 0 * NaN
        lw_[u] = t.adm ? t.lw : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) ok[u][k] = t.c[k];
        const long base = (st + (long)t.h_low * W + t.w_low) * row_stride;
        v[u][0] = *reinterpret_cast<const float4*>(vb + (ok[u][0] ? base : 0));
        v[u][1] = *reinterpret_cast<const float4*>(vb + (ok[u][1] ? base + row_stride : 0));
        v[u][2] = *reinterpret_cast<const float4*>(vb + (ok[u][2] ? base + (long)W * row_stride : 0));
        v[u][3] = *reinterpret_cast<const float4*>(vb + (ok[u][3] ? base + (long)(W + 1) * row_stride : 0));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 v1 = ok[u][0] ? v[u][0] : z4, v2 = ok[u][1] ? v[u][1] : z4;
        const float4 v3 = ok[u][2] ? v[u][2] : z4, v4 = ok[u][3] ? v[u][3] : z4;
        const float lh = lh_[u], lw = lw_[u], hh = 1.f - lh, hw = 1.f - lw, a = a_[u];
        const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
        const float tx[4] = {top.x, top.y, top.z, top.w};
        const float a1[4] = {v1.x, v1.y, v1.z, v1.w}, a2[4] = {v2.x, v2.y, v2.z, v2.w};
        const float a3[4] = {v3.x, v3.y, v3.z, v3.w}, a4[4] = {v4.x, v4.y, v4.z, v4.w};
        float g_attn = 0.f, g_x = 0.f, g_y = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float ta = tx[k] * a;
          g_attn += tx[k] * (w1 * a1[k] + w2 * a2[k] + w3 * a3[k] + w4 * a4[k]);
          g_x += ta * (hh * (a2[k] - a1[k]) + lh * (a4[k] - a3[k]));
          g_y += ta * (hw * (a3[k] - a1[k]) + lw * (a4[k] - a2[k]));
        }
        g_x *= Wf[u];
        g_y *= Hf[u];
        g_attn = group8_sum(g_attn);
        g_x = group8_sum(g_x);
        g_y = group8_sum(g_y);
        if (c4 == 0 && live[u]) {
          const long si = item * LP + s0 + u;
          grad_attn[si] = g_attn;
          grad_loc[si * 2] = g_x;
          grad_loc[si * 2 + 1] = g_y;
        }
      }
    }
    return;
  }
  for (int s = 0; s < LP; ++s) {
    const int l = s / P;
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const long st = lstart[l];
    const long si = item * LP + s;
    const float2 xy = *reinterpret_cast<const float2*>(loc + si * 2);
    const float a = attn[si];
    float g_attn = 0.f, g_x = 0.f, g_y = 0.f;
    const BilinearTerms t = bilinear_terms(xy.x, xy.y, H, W, 1);
    if (t.adm) {   // group-uniform
      const float lh = t.lh, lw = t.lw, hh = t.hh, hw = t.hw;
      const long base = (st + (long)t.h_low * W + t.w_low) * row_stride;
      const long o1 = base, o2 = base + row_stride, o3 = base + (long)W * row_stride, o4 = base + (long)(W + 1) * row_stride;
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      // unconditional loads (a corner outside the map reads row 0 of the batch entry and is replaced by 0)
      const float4 r1 = *reinterpret_cast<const float4*>(vb + (t.c[0] ? o1 : 0));
      const float4 r2 = *reinterpret_cast<const float4*>(vb + (t.c[1] ? o2 : 0));
      const float4 r3 = *reinterpret_cast<const float4*>(vb + (t.c[2] ? o3 : 0));
      const float4 r4 = *reinterpret_cast<const float4*>(vb + (t.c[3] ? o4 : 0));
      const float4 v1 = t.c[0] ? r1 : z4, v2 = t.c[1] ? r2 : z4, v3 = t.c[2] ? r3 : z4, v4 = t.c[3] ? r4 : z4;
      const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
      const float tx[4] = {top.x, top.y, top.z, top.w};
      const float a1[4] = {v1.x, v1.y, v1.z, v1.w}, a2[4] = {v2.x, v2.y, v2.z, v2.w};
      const float a3[4] = {v3.x, v3.y, v3.z, v3.w}, a4[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float ta = tx[k] * a;
        g_attn += tx[k] * (w1 * a1[k] + w2 * a2[k] + w3 * a3[k] + w4 * a4[k]);
        g_x += ta * (hh * (a2[k] - a1[k]) + lh * (a4[k] - a3[k]));
        g_y += ta * (hw * (a3[k] - a1[k]) + lw * (a4[k] - a2[k]));
        if (VALUE_ATOMICS) {
          if (t.c[0]) unsafeAtomicAdd(gvb + o1 + k, ta * w1);
          if (t.c[1]) unsafeAtomicAdd(gvb + o2 + k, ta * w2);
          if (t.c[2]) unsafeAtomicAdd(gvb + o3 + k, ta * w3);
          if (t.c[3]) unsafeAtomicAdd(gvb + o4 + k, ta * w4);
        }
      }
      g_x *= (float)W;
      g_y *= (float)H;
    }
    g_attn = group8_sum(g_attn);
    g_x = group8_sum(g_x);
    g_y = group8_sum(g_y);
    if (c4 == 0) {
      grad_attn[si] = g_attn;
      grad_loc[si * 2] = g_x;
      grad_loc[si * 2 + 1] = g_y;
    }
  }
}

// ---- counting sort of the bilinear row items into 32-pixel bins + owner-computes replay (file header) ------
constexpr int kBinPix = 32;

struct BwdItem { int qpl; float w0, w1; };            // (query << 5 | pixel in bin), left / right corner weight

// counter[gbin] += 1 for every lane with `valid`, aggregated inside the wave: lanes that target the same bin are
// served by ONE atomic of their leader (up to 4 rounds = 4 distinct bins, the rest individually).  The coarse FPN
// levels put the samples of thousands of queries into a few dozen bins: one atomic per lane made those counters
// the bottleneck of the whole backward (count + fill: 12.7 of 16.4 ms).  Returns the lane's slot (FILL only).
template <bool FILL>
__device__ __forceinline__ int bwd_agg_add(int* __restrict__ counter, bool valid, int gbin, int lane) {
  int slot = 0;
  unsigned long long rem = __ballot(valid);
#pragma unroll 1
  for (int round = 0; round < 4 && rem; ++round) {
    const int leader = __builtin_ctzll(rem);
    const int b0 = __builtin_amdgcn_readlane(gbin, leader);
    const unsigned long long same = __ballot(valid && gbin == b0) & rem;
    int base = 0;
    if (lane == leader) {
      if (FILL) base = atomicAdd(counter + b0, (int)__builtin_popcountll(same));
      else atomicAdd(counter + b0, (int)__builtin_popcountll(same));
    }
    if (FILL) {
      base = __builtin_amdgcn_readlane(base, leader);
      if ((same >> lane) & 1ull) slot = base + (int)__builtin_popcountll(same & ((1ull << lane) - 1ull));
    }
    rem &= ~same;
  }
  if ((rem >> lane) & 1ull) {
    if (FILL) slot = atomicAdd(counter + gbin, 1);
    else atomicAdd(counter + gbin, 1);
  }
  return slot;
}

// FILL == false: count the items per bin; FILL == true: write them at cursor[bin]++.  One thread = one sample,
// ordered (b, m, l, q, p) so that a wave holds 64 consecutive (query, point) pairs of ONE (batch, head, level):
// neighbouring queries sample neighbouring pixels, i.e. mostly the same few bins.
template <bool FILL>
__global__ __launch_bounds__(256) void msda_bwd_bin_kernel(
    const int64_t* __restrict__ shapes, const float* __restrict__ loc, const float* __restrict__ attn,
    const unsigned char* __restrict__ nzflag, int* __restrict__ counter, BwdItem* __restrict__ items, int M,
    int L, int Lq, int P, int bins_per_bm, long n_samples) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const long qp_n = (long)Lq * P;
  const bool in_grid = gid < n_samples;
  const long g = in_grid ? gid : n_samples - 1;
  const long qp = g % qp_n;
  long rest = g / qp_n;
  const int l = (int)(rest % L); rest /= L;
  const int m = (int)(rest % M);
  const long b = rest / M;
  const int q = (int)(qp / P), p = (int)(qp - (long)q * P);
  const long item = (b * Lq + q) * M + m;             // (b, q, m)
  const long si = (item * L + l) * P + p;
  int binoff = 0, H = 0, W = 0;
  for (int i = 0; i <= l; ++i) {
    H = (int)shapes[2 * i]; W = (int)shapes[2 * i + 1];
    if (i < l) binoff += (H * W + kBinPix - 1) / kBinPix;
  }
  const unsigned char fl = nzflag[item];
  const float2 xy = *reinterpret_cast<const float2*>(loc + si * 2);
  const float a = attn[si];
  const BilinearTerms t = bilinear_terms(xy.x, xy.y, H, W, lane_flag(in_grid) & lane_flag(fl != 0));
  const int ok = t.adm;
  const int h_low = t.h_low, w_low = t.w_low;
  const float lh = t.lh, lw = t.lw, hh = t.hh, hw = t.hw;
  const int l_ok = lane_flag(w_low >= 0), r_ok = lane_flag(w_low + 1 <= W - 1);
  const int both = l_ok & r_ok;
  const int gbase = (int)((b * M + m) * (long)bins_per_bm) + binoff;
#pragma unroll
  for (int r = 0; r < 2; ++r) {                       // top row, bottom row (wave-uniform control flow)
    const int hy = h_low + r;
    const int valid = ok & lane_flag(hy >= 0) & lane_flag(hy <= H - 1);
    const float wy = (r ? lh : hh) * a;
    const float wl = l_ok ? wy * hw : 0.f, wr = r_ok ? wy * lw : 0.f;
    // left pixel (or the right one when the left is outside the map) decides the bin
    const int pl = hy * W + (l_ok ? w_low : w_low + 1);
    const int bin = valid ? pl / kBinPix : 0, in = pl - bin * kBinPix;
    const int split = valid & both & lane_flag(in == kBinPix - 1);   // the pair straddles a bin edge: two single items
    const float i0 = l_ok ? wl : wr, i1 = (both & (split ^ 1)) ? wr : 0.f;
    const int slot = bwd_agg_add<FILL>(counter, valid != 0, gbase + bin, lane);
    if (FILL && valid) items[slot] = BwdItem{(q << 5) | in, i0, i1};
    if (split) {                                      // 1 in 32: plain atomics
      if (FILL) {
        const int slot2 = atomicAdd(counter + gbase + bin + 1, 1);
        items[slot2] = BwdItem{(q << 5) | 0, wr, 0.f};
      } else {
        atomicAdd(counter + gbase + bin + 1, 1);
      }
    }
  }
}

// Block-aggregated form of the two binning passes (default).  PMC on the training step (profiles/
// r02_train_pmc_msda_bwd.txt) showed the wave-aggregated kernels above resident at < 2 waves per SIMD with 63-88 % of
// their wave-cycles waiting: an SCA launch sends 8.7 M device-scope integer atomics, thousands of them to the SAME
// counter of a coarse-level bin (32 pixels of the 15 x 25 map collect ~20 000 row items), and same-address
// device-scope atomics retire one round trip at a time.  Here one 1024-thread block owns 1024 consecutive
// (query, point) samples of ONE (batch, head, level); the level's bins (<= a few hundred) are a dense histogram in
// LDS (ds_add is cheap and returns the sample's slot inside the block), and the block issues ONE global atomic per
// bin it touched: ~30 per 1024 samples instead of ~600, and a hot counter sees one atomic per block (77 per SCA
// launch instead of ~3 700).
constexpr int kBinBlockThreads = 1024;

template <bool FILL>
__global__ __launch_bounds__(kBinBlockThreads) void msda_bwd_bin_block_kernel(
    const int64_t* __restrict__ shapes, const float* __restrict__ loc, const float* __restrict__ attn,
    const unsigned char* __restrict__ nzflag, int* __restrict__ counter, BwdItem* __restrict__ items, int M,
    int L, int Lq, int P, int bins_per_bm, int blocks_per_bml) {
  extern __shared__ int bin_lds[];                    // [nb] block histogram, then (FILL) [nb] global bases
  const int tid = threadIdx.x;
  const int chunk = (int)(blockIdx.x % (unsigned)blocks_per_bml);
  long rest = blockIdx.x / (unsigned)blocks_per_bml;
  const int l = (int)(rest % L); rest /= L;
  const int m = (int)(rest % M);
  const long b = rest / M;
  int binoff = 0, H = 0, W = 0;
  for (int i = 0; i <= l; ++i) {
    H = (int)shapes[2 * i]; W = (int)shapes[2 * i + 1];
    if (i < l) binoff += (H * W + kBinPix - 1) / kBinPix;
  }
  const int nb = (H * W + kBinPix - 1) / kBinPix;
  int* hist = bin_lds;
  int* gbase_of = bin_lds + nb;
  for (int i = tid; i < nb; i += kBinBlockThreads) hist[i] = 0;
  __syncthreads();

  const long qp_n = (long)Lq * P;
  const long qp_raw = (long)chunk * kBinBlockThreads + tid;
  const bool in_grid = qp_raw < qp_n;
  const long qp = in_grid ? qp_raw : qp_n - 1;
  const int q = (int)(qp / P), p = (int)(qp - (long)q * P);
  const long item = (b * Lq + q) * M + m;             // (b, q, m)
  const long si = (item * L + l) * P + p;
  const unsigned char fl = nzflag[item];
  const float2 xy = *reinterpret_cast<const float2*>(loc + si * 2);
  const float a = attn[si];
  const BilinearTerms t = bilinear_terms(xy.x, xy.y, H, W, lane_flag(in_grid) & lane_flag(fl != 0));
  const int ok = t.adm;
  const int h_low = t.h_low, w_low = t.w_low;
  const float lh = t.lh, lw = t.lw, hh = t.hh, hw = t.hw;
  const int l_ok = lane_flag(w_low >= 0), r_ok = lane_flag(w_low + 1 <= W - 1);
  const int both = l_ok & r_ok;

  // same row items as msda_bwd_bin_kernel: per bilinear row one item at the left pixel's bin (two single items when
  // the pair straddles a bin edge)
  int valid[2], split[2];                              // 0 / 1 lane flags (common.h: lane_flag)
  int bin[2], in[2], slot[2], slot2[2];
  float i0[2], i1[2], wr_[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int hy = h_low + r;
    valid[r] = ok & lane_flag(hy >= 0) & lane_flag(hy <= H - 1);
    const float wy = (r ? lh : hh) * a;
    const float wl = l_ok ? wy * hw : 0.f, wr = r_ok ? wy * lw : 0.f;
    const int pl = hy * W + (l_ok ? w_low : w_low + 1);
    bin[r] = valid[r] ? pl / kBinPix : 0;
    in[r] = pl - bin[r] * kBinPix;
    split[r] = valid[r] & both & lane_flag(in[r] == kBinPix - 1);
    i0[r] = l_ok ? wl : wr;
    i1[r] = (both & (split[r] ^ 1)) ? wr : 0.f;
    wr_[r] = wr;
    slot[r] = slot2[r] = 0;
    if (valid[r]) slot[r] = atomicAdd(&hist[bin[r]], 1);            // LDS
    if (split[r]) slot2[r] = atomicAdd(&hist[bin[r] + 1], 1);
  }
  __syncthreads();
  int* gcnt = counter + (long)(b * M + m) * bins_per_bm + binoff;
  for (int i = tid; i < nb; i += kBinBlockThreads) {
    const int c = hist[i];
    if (c > 0) {
      if (FILL) gbase_of[i] = atomicAdd(gcnt + i, c);               // global: one per (block, touched bin)
      else atomicAdd(gcnt + i, c);
    }
  }
  if (FILL) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (valid[r]) items[gbase_of[bin[r]] + slot[r]] = BwdItem{(q << 5) | in[r], i0[r], i1[r]};
      if (split[r]) items[gbase_of[bin[r] + 1] + slot2[r]] = BwdItem{(q << 5) | 0, wr_[r], 0.f};
    }
  }
}

// exclusive prefix sum of counts[0..n) into counts (in place) and cursor; counts[n] = total.  One block.
// The same pass writes the REPLAY WORK LIST: every non-empty bin becomes ceil(count / kReplayCap) entries
// {bin, first item, last item + 1, split?} (second prefix sum over the entry counts), work_count[0] = entries.
// A bin of the coarsest FPN level collects ~50 000 row items; replayed by ONE block (latency-bound: ~0.2 us per
// item and half-wave) those few blocks were the whole kernel (PMC: 0.6 waves per SIMD resident, 2.3 ms per SCA
// launch); split into 2048-item pieces they spread over the chip, and empty bins cost no block at all.
constexpr int kReplayCap = 2048;

// Three launches (round 6; one 1 024-thread block walking its bins in a thread-strided order took 90 us per call): (1) every block
// scans its 1 024 consecutive bins — item count, work entries, split flag — and leaves its totals; (2) one block scans the
// <= 1 024 block totals; (3) every block repeats its local scan on top of its offset and writes counts / cursor / work list.
struct ScanTriple { int items, work, split; };

__device__ __forceinline__ ScanTriple bwd_block_scan3(ScanTriple v, int tid, int (*sh)[1024], ScanTriple& total) {
  // inclusive Hillis-Steele over the block's 1 024 triples; returns the EXCLUSIVE prefix of this thread, `total` = block sum
  sh[0][tid] = v.items; sh[1][tid] = v.work; sh[2][tid] = v.split;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int a0 = tid >= d ? sh[0][tid - d] : 0, a1 = tid >= d ? sh[1][tid - d] : 0, a2 = tid >= d ? sh[2][tid - d] : 0;
    __syncthreads();
    sh[0][tid] += a0; sh[1][tid] += a1; sh[2][tid] += a2;
    __syncthreads();
  }
  ScanTriple ex;
  ex.items = sh[0][tid] - v.items; ex.work = sh[1][tid] - v.work; ex.split = sh[2][tid] - v.split;
  total.items = sh[0][1023]; total.work = sh[1][1023]; total.split = sh[2][1023];
  __syncthreads();
  return ex;
}

__device__ __forceinline__ ScanTriple bwd_bin_triple(int c, int cap) {
  ScanTriple v;
  v.items = c;
  v.work = c > 0 ? (c - 1) / cap + 1 : 0;
  v.split = c > cap ? 1 : 0;
  return v;
}

__global__ __launch_bounds__(1024) void msda_bwd_scan_totals_kernel(const int* __restrict__ counts, int n, int cap,
                                                                    int* __restrict__ aux) {
  __shared__ int sh[3][1024];
  const int tid = threadIdx.x, i = (int)blockIdx.x * 1024 + tid;
  ScanTriple total;
  (void)bwd_block_scan3(bwd_bin_triple(i < n ? counts[i] : 0, cap), tid, sh, total);
  if (tid == 0) { aux[3 * blockIdx.x] = total.items; aux[3 * blockIdx.x + 1] = total.work; aux[3 * blockIdx.x + 2] = total.split; }
}

// aux[3 b ..] block totals -> exclusive block offsets in place; counts[n] = all items, work_count[0] = entries, [2] = split bins
__global__ __launch_bounds__(1024) void msda_bwd_scan_offsets_kernel(int* __restrict__ aux, int nblocks, int* __restrict__ counts,
                                                                     int n, int* __restrict__ work_count) {
  __shared__ int sh[3][1024];
  const int tid = threadIdx.x;
  ScanTriple v;
  v.items = tid < nblocks ? aux[3 * tid] : 0; v.work = tid < nblocks ? aux[3 * tid + 1] : 0; v.split = tid < nblocks ? aux[3 * tid + 2] : 0;
  ScanTriple total;
  const ScanTriple ex = bwd_block_scan3(v, tid, sh, total);
  if (tid < nblocks) { aux[3 * tid] = ex.items; aux[3 * tid + 1] = ex.work; aux[3 * tid + 2] = ex.split; }
  if (tid == 0) { counts[n] = total.items; work_count[0] = total.work; work_count[2] = total.split; }
}

// entry = {bin, first item, last item + 1, tag}: tag = slot + 1 of a SPLIT bin (more than one entry; the deterministic replay
// meets its pieces in the slot's scratch tile and needs their number: split_meta[2 slot + 1]; NULL in the default mode), else 0
__global__ __launch_bounds__(1024) void msda_bwd_scan_kernel(int* __restrict__ counts, int* __restrict__ cursor,
                                                             int n, int4* __restrict__ work, const int* __restrict__ aux,
                                                             int cap, int* __restrict__ split_meta) {
  __shared__ int sh[3][1024];
  const int tid = threadIdx.x, i = (int)blockIdx.x * 1024 + tid;
  const int c = i < n ? counts[i] : 0;
  const ScanTriple v = bwd_bin_triple(c, cap);
  ScanTriple total;
  const ScanTriple ex = bwd_block_scan3(v, tid, sh, total);
  if (i >= n) return;
  const int run = aux[3 * blockIdx.x] + ex.items, wrun = aux[3 * blockIdx.x + 1] + ex.work, srun = aux[3 * blockIdx.x + 2] + ex.split;
  counts[i] = run;
  cursor[i] = run;
  const int nw = v.work, tag = nw > 1 ? srun + 1 : 0;
  for (int k = 0; k < nw; ++k)
    work[wrun + k] = make_int4(i, run + k * cap, k + 1 < nw ? run + (k + 1) * cap : run + c, tag);
  if (nw > 1 && split_meta != nullptr) split_meta[2 * srun + 1] = nw;
}

// one block = one work-list entry (a bin, or a 2048-item piece of a hot bin): its items are dealt to the block's 8 half-waves, each with
// a private 32 pixels x 32 channels accumulator in LDS (lane = channel; plain read-add-write, no conflicts); the 8
// copies are summed and added to grad_value by the block, the bin's only writer.  (One WAVE per bin left the hot
// bins of the coarse FPN levels — thousands of items — on a single wave: 24 ms of a training step.)
__global__ __launch_bounds__(256) void msda_bwd_replay_kernel(
    const int64_t* __restrict__ shapes, const int64_t* __restrict__ lstart, const int4* __restrict__ work,
    const int* __restrict__ work_count, const BwdItem* __restrict__ items, const float* __restrict__ grad_out,
    float* __restrict__ grad_value, int S, int M, int L, int Lq, int bins_per_bm) {
  constexpr int D = 32;
  __shared__ float acc_s[8][(kBinPix + 1) * D];        // per half-wave; +1 pixel: the w1 lane of pixel 31
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, ch = lane & 31;
  if ((int)blockIdx.x >= work_count[0]) return;        // the grid is an upper bound of the work list
  const int4 wk = work[blockIdx.x];
  const long bin_g = wk.x;
  const int beg = wk.y, end = wk.z;
  const bool shared_bin = wk.w != 0;                   // other blocks add into the same 32 pixels
  const int hw = wave * 2 + half;
  float* acc = acc_s[hw];
#pragma unroll
  for (int i = 0; i <= kBinPix; ++i) acc[i * D + ch] = 0.f;
  const long bm = bin_g / bins_per_bm;
  int bl = (int)(bin_g - bm * bins_per_bm);
  const int m = (int)(bm % M);
  const long b = bm / M;
  int l = 0, HW = 0;
  for (; l < L; ++l) {
    HW = (int)shapes[2 * l] * (int)shapes[2 * l + 1];
    const int nb = (HW + kBinPix - 1) / kBinPix;
    if (bl < nb) break;
    bl -= nb;
  }
  if (l >= L) return;                                  // slack bins of the upper bound (never filled)
  const float* go = grad_out + (b * Lq * (long)M + m) * D + ch;        // + q * M * D
  // 8 items per half-wave per step, software-pipelined over three steps: the item records of step i+2 and the
  // output-gradient rows of step i+1 are in flight while step i is added into LDS (two dependent global loads per
  // item made every step cost both latencies: ~1.5 us per 64 items and block)
  int qA[8], qB[8], qC[8];
  float w0A[8], w1A[8], w0B[8], w1B[8], w0C[8], w1C[8], gA[8], gB[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int i0 = beg + hw + 8 * u, i1 = i0 + 64;
    const BwdItem a = items[i0 < end ? i0 : beg];
    const BwdItem c = items[i1 < end ? i1 : beg];
    qA[u] = i0 < end ? a.qpl : -1; w0A[u] = a.w0; w1A[u] = a.w1;
    qB[u] = i1 < end ? c.qpl : -1; w0B[u] = c.w0; w1B[u] = c.w1;
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) gA[u] = go[(long)(qA[u] < 0 ? 0 : qA[u] >> 5) * M * D];
  for (int base = beg; base < end; base += 64) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {                      // item records two steps ahead
      const int idx = base + 128 + hw + 8 * u;
      const BwdItem it = items[idx < end ? idx : beg];
      qC[u] = idx < end ? it.qpl : -1; w0C[u] = it.w0; w1C[u] = it.w1;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) gB[u] = go[(long)(qB[u] < 0 ? 0 : qB[u] >> 5) * M * D];   // rows one step ahead
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (qA[u] >= 0) {
        const int pl = qA[u] & 31;
        acc[pl * D + ch] += gA[u] * w0A[u];
        acc[(pl + 1) * D + ch] += gA[u] * w1A[u];      // w1 == 0 for single items (pixel 32 is a dummy row)
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      qA[u] = qB[u]; w0A[u] = w0B[u]; w1A[u] = w1B[u]; gA[u] = gB[u];
      qB[u] = qC[u]; w0B[u] = w0C[u]; w1B[u] = w1C[u];
    }
  }
  __syncthreads();
  // the 8 partial sums -> grad_value
  const int p0 = bl * kBinPix, np = min(kBinPix, HW - p0);
  const long st = lstart[l];
  for (int i = tid; i < np * 8; i += 256) {
    const int px = i >> 3, c4 = i & 7;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float4 u = *reinterpret_cast<const float4*>(acc_s[k] + px * D + c4 * 4);
      sum.x += u.x; sum.y += u.y; sum.z += u.z; sum.w += u.w;
    }
    float* dstf = grad_value + ((b * S + st + p0 + px) * M + m) * D + c4 * 4;
    if (shared_bin) {                                  // a split bin: a few dozen blocks per bin at most
      unsafeAtomicAdd(dstf + 0, sum.x);
      unsafeAtomicAdd(dstf + 1, sum.y);
      unsafeAtomicAdd(dstf + 2, sum.z);
      unsafeAtomicAdd(dstf + 3, sum.w);
    } else {
      float4* dst = reinterpret_cast<float4*>(dstf);
      float4 o = *dst;
      o.x += sum.x; o.y += sum.y; o.z += sum.z; o.w += sum.w;
      *dst = o;
    }
  }
}

// ---- deterministic mode (OCC_MSDA_BWD_DETERMINISTIC=1) ---------------------------------------------------------------
// The default replay is not bit-reproducible: the position of an item inside its bin follows the integer-atomic slot
// order of the fill pass, so the fp32 summation order changes from run to run, and the pieces of a split bin are
// combined with float atomics.  Here the sums are made ORDER-INDEPENDENT instead of the order fixed: every
// contribution g * w (one fp32 product, the same in every run) is converted to 64-bit fixed point — scaled by a power
// of two taken from max |grad_output| of the launch, 40 fraction bits below it, 22 bits of headroom for the item count
// of a bin — and accumulated with integer adds, which commute; bins are not split (one block owns a bin), the total is
// converted back once.  Resolution 2^-40 of the largest output gradient: finer than the fp32 accumulation it replaces.
// Slower (a hot coarse-level bin is replayed by one block): an opt-in for reproducible training runs, the contract of
// the reference's caller-zeroed plain accumulation (multi_scale_deformable_attn_function.py:146-163).
__global__ void msda_bwd_maxabs_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ out) {
  float m = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = fabsf(x[i]);
    if (v < 3.0e38f) m = fmaxf(m, v);                  // Inf / NaN gradients do not set the scale
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));      // non-negative floats order like their bits
}

// scratch of the split bins: slot s owns kDetSlotWords 64-bit words = 32 x 32 sums + 32 "non-finite" flags (low halves)
constexpr int kDetSlotWords = kBinPix * 32 + kBinPix;

// zeroes the scratch tiles and arrival counters of the split bins that exist (the grid is the upper bound of their number)
__global__ __launch_bounds__(256) void msda_bwd_det_zero_kernel(const int* __restrict__ work_count, int* __restrict__ split_meta,
                                                                unsigned long long* __restrict__ tiles) {
  const int slot = (int)blockIdx.x;
  if (slot >= work_count[2]) return;
  if (threadIdx.x == 0) split_meta[2 * slot] = 0;
  for (int i = threadIdx.x; i < kDetSlotWords; i += 256) tiles[(long)slot * kDetSlotWords + i] = 0ull;
}

// Round 6: the loop of the default replay (item records two steps, gradient rows one step ahead) with 64-bit fixed-point sums:
// ONE tile per block in LDS, accumulated with LDS integer atomics (ds_add_u64: fire and forget, any number of half-waves on
// one pixel), and hot bins SPLIT like the default mode's — the pieces of split bin `slot` add their tiles into the slot's
// scratch with 64-bit global atomics (integer addition commutes: no order reaches the result) and the piece that arrives
// last converts the total.  (The first version of this mode replayed a hot coarse-level bin in one block, one dependent
// load pair per item: 78.8 against 50.1 ms per training step.)
__global__ __launch_bounds__(256) void msda_bwd_replay_det_kernel(
    const int64_t* __restrict__ shapes, const int64_t* __restrict__ lstart, const int4* __restrict__ work,
    const int* __restrict__ work_count, const BwdItem* __restrict__ items, const float* __restrict__ grad_out,
    float* __restrict__ grad_value, int S, int M, int L, int Lq, int bins_per_bm, int* __restrict__ split_meta,
    unsigned long long* __restrict__ tiles) {
  constexpr int D = 32;
  __shared__ unsigned long long acc_d[(kBinPix + 1) * D];
  // a non-finite contribution (Inf / NaN in grad_output) has no fixed-point image: the pixels it touches are written as
  // NaN, like the float path would leave them, instead of the finite garbage of a saturated conversion
  __shared__ int bad_px[kBinPix + 1];
  __shared__ int last_piece;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, ch = lane & 31;
  if ((int)blockIdx.x >= work_count[0]) return;
  const int4 wk = work[blockIdx.x];
  const long bin_g = wk.x;
  const int beg = wk.y, end = wk.z;
  const int slot = wk.w - 1;                           // >= 0: one piece of a split bin
  const int hw = wave * 2 + half;
  for (int i = tid; i < (kBinPix + 1) * D; i += 256) acc_d[i] = 0ull;
  if (tid <= kBinPix) bad_px[tid] = 0;
  __syncthreads();
  const long bm = bin_g / bins_per_bm;
  int bl = (int)(bin_g - bm * bins_per_bm);
  const int m = (int)(bm % M);
  const long b = bm / M;
  int l = 0, HW = 0;
  for (; l < L; ++l) {
    HW = (int)shapes[2 * l] * (int)shapes[2 * l + 1];
    const int nb = (HW + kBinPix - 1) / kBinPix;
    if (bl < nb) break;
    bl -= nb;
  }
  if (l >= L) return;
  const float mx = __uint_as_float((unsigned)work_count[1]);           // max |grad_output| of the launch
  int e = 0;
  if (mx > 0.f) (void)frexpf(mx, &e);                                  // mx < 2^e
  const int sh = 40 - e;                                               // products scaled by 2^sh: |.| < 2^40, exact in fp32
  const double inv = ldexp(1.0, e - 40);
  const float* go = grad_out + (b * Lq * (long)M + m) * D + ch;        // + q * M * D
  int qA[8], qB[8], qC[8];
  float w0A[8], w1A[8], w0B[8], w1B[8], w0C[8], w1C[8], gA[8], gB[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int i0 = beg + hw + 8 * u, i1 = i0 + 64;
    const BwdItem a = items[i0 < end ? i0 : beg];
    const BwdItem c = items[i1 < end ? i1 : beg];
    qA[u] = i0 < end ? a.qpl : -1; w0A[u] = a.w0; w1A[u] = a.w1;
    qB[u] = i1 < end ? c.qpl : -1; w0B[u] = c.w0; w1B[u] = c.w1;
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) gA[u] = go[(long)(qA[u] < 0 ? 0 : qA[u] >> 5) * M * D];
  for (int base = beg; base < end; base += 64) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {                      // item records two steps ahead
      const int idx = base + 128 + hw + 8 * u;
      const BwdItem it = items[idx < end ? idx : beg];
      qC[u] = idx < end ? it.qpl : -1; w0C[u] = it.w0; w1C[u] = it.w1;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) gB[u] = go[(long)(qB[u] < 0 ? 0 : qB[u] >> 5) * M * D];   // rows one step ahead
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (qA[u] >= 0) {
        const int pl = qA[u] & 31;
        const float c0 = gA[u] * w0A[u], c1 = gA[u] * w1A[u];
        if (!(fabsf(c0) < 3.0e38f)) bad_px[pl] = 1;
        else atomicAdd(&acc_d[pl * D + ch], (unsigned long long)__float2ll_rn(ldexpf(c0, sh)));
        if (!(fabsf(c1) < 3.0e38f)) bad_px[pl + 1] = 1;
        else atomicAdd(&acc_d[(pl + 1) * D + ch], (unsigned long long)__float2ll_rn(ldexpf(c1, sh)));
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      qA[u] = qB[u]; w0A[u] = w0B[u]; w1A[u] = w1B[u]; gA[u] = gB[u];
      qB[u] = qC[u]; w0B[u] = w0C[u]; w1B[u] = w1C[u];
    }
  }
  __syncthreads();
  const int p0 = bl * kBinPix, np = min(kBinPix, HW - p0);
  const long st = lstart[l];
  if (slot < 0) {                                      // the bin's only writer
    for (int i = tid; i < np * D; i += 256) {
      const int px = i >> 5, c = i & 31;
      float* dst = grad_value + ((b * S + st + p0 + px) * M + m) * D + c;
      if (bad_px[px]) *dst = __builtin_nanf("");
      else *dst += (float)((double)(long long)acc_d[px * D + c] * inv);
    }
    return;
  }
  unsigned long long* tile = tiles + (long)slot * kDetSlotWords;
  for (int i = tid; i < np * D; i += 256) {
    const unsigned long long v = acc_d[i];
    if (v != 0ull) atomicAdd(&tile[i], v);
  }
  if (tid < np && bad_px[tid]) atomicAdd(&tile[kBinPix * D + tid], 1ull);
  __threadfence();
  __syncthreads();
  if (tid == 0) last_piece = atomicAdd(&split_meta[2 * slot], 1) == split_meta[2 * slot + 1] - 1;
  __syncthreads();
  if (!last_piece) return;
  __threadfence();
  for (int i = tid; i < np * D; i += 256) {            // the totals, read where the atomics were executed
    const int px = i >> 5, c = i & 31;
    const unsigned long long bad = atomicAdd(&tile[kBinPix * D + px], 0ull);
    const long long sum = (long long)atomicAdd(&tile[i], 0ull);
    float* dst = grad_value + ((b * S + st + p0 + px) * M + m) * D + c;
    if (bad) *dst = __builtin_nanf("");
    else *dst += (float)((double)sum * inv);
  }
}

// Generic D: one thread per (item, channel); reductions over the item's D threads through LDS.
__global__ __launch_bounds__(256) void msda_bwd_generic_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lstart, const float* __restrict__ loc,
    const float* __restrict__ attn, const float* __restrict__ grad_out,
    float* __restrict__ grad_value, float* __restrict__ grad_loc, float* __restrict__ grad_attn,
    int S, int M, int D, int L, int Lq, int P, long n_items) {
  // block = (256 / D) items x D channels (D divides 256) or one item per block with D <= 256 threads
  extern __shared__ float red[];   // 3 * blockDim.x
  const int per_block = blockDim.x / D;
  const int li = threadIdx.x / D, c = threadIdx.x % D;
  const long item = (long)blockIdx.x * per_block + li;
  const bool live = item < n_items && li < per_block;
  const int LP = L * P;
  const long row_stride = (long)M * D;
  long voff = 0;
  float top = 0.f;
  if (live) {
    const int m = (int)(item % M);
    const long b = item / ((long)M * Lq);
    voff = b * (long)S * row_stride + (long)m * D + c;
    top = grad_out[item * D + c];
  }
  for (int s = 0; s < LP; ++s) {
    float g_attn = 0.f, g_x = 0.f, g_y = 0.f;
    if (live) {
      const int l = s / P;
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const long st = lstart[l];
      const long si = item * LP + s;
      const float loc_w = loc[si * 2], loc_h = loc[si * 2 + 1], a = attn[si];
      const BilinearTerms t = bilinear_terms(loc_w, loc_h, H, W, 1);
      if (t.adm) {
        const float lh = t.lh, lw = t.lw, hh = t.hh, hw = t.hw;
        const long base = voff + (st + (long)t.h_low * W + t.w_low) * row_stride;
        const long o1 = base, o2 = base + row_stride, o3 = base + (long)W * row_stride, o4 = base + (long)(W + 1) * row_stride;
        const float ta = top * a;
        // unconditional loads (a corner outside the map reads element 0 and is replaced by 0), conditional atomics
        const float r1 = value[t.c[0] ? o1 : 0], r2 = value[t.c[1] ? o2 : 0], r3 = value[t.c[2] ? o3 : 0], r4 = value[t.c[3] ? o4 : 0];
        const float v1 = t.c[0] ? r1 : 0.f, v2 = t.c[1] ? r2 : 0.f, v3 = t.c[2] ? r3 : 0.f, v4 = t.c[3] ? r4 : 0.f;
        if (t.c[0]) unsafeAtomicAdd(grad_value + o1, ta * hh * hw);
        if (t.c[1]) unsafeAtomicAdd(grad_value + o2, ta * hh * lw);
        if (t.c[2]) unsafeAtomicAdd(grad_value + o3, ta * lh * hw);
        if (t.c[3]) unsafeAtomicAdd(grad_value + o4, ta * lh * lw);
        g_attn = top * (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4);
        g_x = ta * (hh * (v2 - v1) + lh * (v4 - v3)) * (float)W;
        g_y = ta * (hw * (v3 - v1) + lw * (v4 - v2)) * (float)H;
      }
    }
    red[threadIdx.x] = g_attn;
    red[blockDim.x + threadIdx.x] = g_x;
    red[2 * blockDim.x + threadIdx.x] = g_y;
    __syncthreads();
    if (live && c == 0) {
      float sa = 0.f, sx = 0.f, sy = 0.f;
      for (int k = 0; k < D; ++k) {
        sa += red[li * D + k];
        sx += red[blockDim.x + li * D + k];
        sy += red[2 * blockDim.x + li * D + k];
      }
      const long si = item * LP + s;
      grad_attn[si] = sa;
      grad_loc[si * 2] = sx;
      grad_loc[si * 2 + 1] = sy;
    }
    __syncthreads();
  }
}

}  // namespace occ

namespace occ {
struct BwdWsLayout { size_t off_cnt, off_cur, off_work, off_items, off_meta, off_tiles, off_aux, bytes; long n_bins, n_samples, max_items, work_cap, max_split; int bins_per_bm; bool ok; };
static BwdWsLayout bwd_ws_layout(int B, int S, int M, int L, int Lq, int P) {
  BwdWsLayout w;
  const long n_items = (long)B * Lq * M;
  w.bins_per_bm = S / kBinPix + L + 1;                                // >= sum_l ceil(H_l*W_l / 32)
  w.n_bins = (long)B * M * w.bins_per_bm;
  w.n_samples = n_items * L * P;
  w.max_items = 4 * w.n_samples;                                      // 2 rows x (1 or 2 items)
  w.off_cnt = ((size_t)n_items + 255) & ~(size_t)255;
  w.off_cur = w.off_cnt + (((size_t)(w.n_bins + 1) * 4 + 255) & ~(size_t)255);
  w.off_work = w.off_cur + (((size_t)(w.n_bins + 1) * 4 + 255) & ~(size_t)255);
  w.work_cap = w.n_bins + w.max_items / kReplayCap + 1;               // >= sum_bins ceil(count / kReplayCap)
  w.off_items = w.off_work + (((size_t)(w.work_cap + 1) * 16 + 255) & ~(size_t)255);   // [0] = entry count
  // deterministic mode: a split bin holds more than kReplayCap items, so there are at most max_items / kReplayCap of them;
  // per slot an arrival counter + piece count and a tile of kDetSlotWords 64-bit sums
  w.max_split = w.max_items / kReplayCap + 1;
  w.off_meta = (w.off_items + (size_t)w.max_items * sizeof(BwdItem) + 255) & ~(size_t)255;
  w.off_tiles = (w.off_meta + (size_t)w.max_split * 8 + 255) & ~(size_t)255;
  w.off_aux = (w.off_tiles + (size_t)w.max_split * kDetSlotWords * 8 + 255) & ~(size_t)255;      // 3 ints per 1 024-bin scan block
  w.bytes = w.off_aux + 3 * 1024 * sizeof(int);
  w.ok = w.n_bins <= 1024L * 1024 && Lq < (1 << 26) && w.n_bins < (1L << 30) && w.max_items < (1L << 31) && w.work_cap < (1L << 31);
  return w;
}
}  // namespace occ

// Bytes of scratch the atomic-free grad_value path (D == 32) needs for these shapes; 0 = that path does not apply
// (other D, or index ranges beyond its 32-bit counters) and occ_ms_deform_attn_backward_ws_f32 ignores `workspace`.
extern "C" int64_t occ_ms_deform_attn_backward_workspace_bytes(int B, int S, int M, int D, int L, int Lq, int P) {
  using namespace occ;
  if (D != 32 || B <= 0 || S <= 0 || M <= 0 || L <= 0 || Lq <= 0 || P <= 0) return 0;
  const BwdWsLayout w = bwd_ws_layout(B, S, M, L, Lq, P);
  return w.ok ? (int64_t)w.bytes : 0;
}

// ms_deform_attn_backward with CALLER-PROVIDED scratch (`workspace`, at least ..._workspace_bytes() bytes, 256-byte
// aligned, need not be initialised): the Python operator module passes a tensor from torch's caching allocator, so
// the ~1.0-1.2 GB per SCA call at the base config neither bypass nor compete with that pool.  workspace == NULL:
// the library allocates with hipMallocAsync; if that fails it falls back to the float-atomic kernel (~4x slower) and
// says so ONCE on stderr.  grad_value's summation order inside a 32-pixel bin follows integer-atomic slot order:
// the last bits of grad_value are not reproducible run to run (as with mmcv's atomicAdd) — unless
// OCC_MSDA_BWD_DETERMINISTIC=1 (D == 32): order-independent fixed-point accumulation, bit-identical runs.
extern "C" int occ_ms_deform_attn_backward_ws_f32(
    const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
    const float* sampling_loc, const float* attn_weight, const float* grad_output, float* grad_value,
    float* grad_sampling_loc, float* grad_attn_weight, int B, int S, int M, int D, int L, int Lq,
    int P, int im2col_step, void* workspace, int64_t workspace_bytes, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(value && spatial_shapes && level_start_index && sampling_loc && attn_weight &&
                    grad_output && grad_value && grad_sampling_loc && grad_attn_weight,
                "ms_deform_attn_backward: null pointer argument");
  OCC_CHECK_ARG(B > 0 && S > 0 && M > 0 && D > 0 && L > 0 && Lq > 0 && P > 0,
                "ms_deform_attn_backward: non-positive dimension");
  OCC_CHECK_ARG(im2col_step > 0, "ms_deform_attn_backward: im2col_step must be positive");
  const int step = B < im2col_step ? B : im2col_step;
  OCC_CHECK_ARG(B % step == 0, "ms_deform_attn_backward: batch(%d) must divide im2col_step(%d)", B,
                step);
  OCC_CHECK_ARG(D <= 256, "ms_deform_attn_backward: channels per head (%d) above 256", D);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long n_items = (long)B * Lq * M;
  if (D == 32) {
    const long threads = n_items * 8;
    const dim3 grid1((unsigned)((threads + 255) / 256));
    // atomic-free grad_value (file header): per-sample gradients + flags, count, scan, fill, replay
    static const bool use_atomics = getenv("OCC_MSDA_BWD_ATOMICS") != nullptr;
    // read per call (tests switch it inside one process): bit-reproducible grad_value, see msda_bwd_replay_det_kernel
    const char* det_env = getenv("OCC_MSDA_BWD_DETERMINISTIC");
    const bool deterministic = det_env != nullptr && det_env[0] == '1';
    const BwdWsLayout w = bwd_ws_layout(B, S, M, L, Lq, P);
    char* ws = nullptr;
    bool own = false;
    hipError_t e = hipErrorUnknown;
    if (!use_atomics && w.ok) {
      if (workspace != nullptr) {
        OCC_CHECK_ARG(workspace_bytes >= (int64_t)w.bytes && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0,
                      "ms_deform_attn_backward: workspace too small (%ld < %ld bytes) or not 256-byte aligned",
                      (long)workspace_bytes, (long)w.bytes);
        ws = reinterpret_cast<char*>(workspace);
        e = hipSuccess;
      } else {
        e = hipMallocAsync(reinterpret_cast<void**>(&ws), w.bytes, st);
        own = e == hipSuccess;
      }
    }
    if (e == hipSuccess) e = hipMemsetAsync(ws, 0, w.off_cur, st);      // flags + counts
    if (e == hipSuccess) {
      unsigned char* flags = reinterpret_cast<unsigned char*>(ws);
      int* counts = reinterpret_cast<int*>(ws + w.off_cnt);
      int* cursor = reinterpret_cast<int*>(ws + w.off_cur);
      BwdItem* items = reinterpret_cast<BwdItem*>(ws + w.off_items);
      int* work_count = reinterpret_cast<int*>(ws + w.off_work);
      int4* work = reinterpret_cast<int4*>(ws + w.off_work + 16);
      const dim3 grid_s((unsigned)((w.n_samples + 255) / 256));
      hipLaunchKernelGGL(msda_bwd_d32_kernel<false>, grid1, dim3(256), 0, st, value, spatial_shapes,
                         level_start_index, sampling_loc, attn_weight, grad_output, grad_value,
                         grad_sampling_loc, grad_attn_weight, flags, S, M, L, Lq, P, n_items);
      // binning passes: block-aggregated (default) or the wave-aggregated kernels (OCC_MSDA_BWD_BIN=wave, or a
      // level with more bins than the LDS histogram holds)
      static const bool wave_bins = getenv("OCC_MSDA_BWD_BIN") != nullptr &&
                                    std::string(getenv("OCC_MSDA_BWD_BIN")) == "wave";
      const long qp_n = (long)Lq * P;
      const long blocks_per_bml = (qp_n + kBinBlockThreads - 1) / kBinBlockThreads;
      const long grid_b = (long)B * M * L * blocks_per_bml;
      const size_t lds_b = (size_t)(S / kBinPix + 2) * 2 * sizeof(int);        // >= 2 * bins of the largest level
      const bool block_bins = !wave_bins && lds_b <= 48 * 1024 && grid_b < (1L << 31);
      if (block_bins) {
        hipLaunchKernelGGL(msda_bwd_bin_block_kernel<false>, dim3((unsigned)grid_b), dim3(kBinBlockThreads), lds_b,
                           st, spatial_shapes, sampling_loc, attn_weight, flags, counts, items, M, L, Lq, P,
                           w.bins_per_bm, (int)blocks_per_bml);
      } else {
        hipLaunchKernelGGL(msda_bwd_bin_kernel<false>, grid_s, dim3(256), 0, st, spatial_shapes, sampling_loc,
                           attn_weight, flags, counts, items, M, L, Lq, P, w.bins_per_bm, w.n_samples);
      }
      int* split_meta = reinterpret_cast<int*>(ws + w.off_meta);
      unsigned long long* tiles = reinterpret_cast<unsigned long long*>(ws + w.off_tiles);
const int scan_blocks = (int)((w.n_bins + 1023) / 1024);
      int* scan_aux = reinterpret_cast<int*>(ws + w.off_aux);
      hipLaunchKernelGGL(msda_bwd_scan_totals_kernel, dim3((unsigned)scan_blocks), dim3(1024), 0, st, counts, (int)w.n_bins,
                         kReplayCap, scan_aux);
      hipLaunchKernelGGL(msda_bwd_scan_offsets_kernel, dim3(1), dim3(1024), 0, st, scan_aux, scan_blocks, counts, (int)w.n_bins,
                         work_count);
      hipLaunchKernelGGL(msda_bwd_scan_kernel, dim3((unsigned)scan_blocks), dim3(1024), 0, st, counts, cursor, (int)w.n_bins, work,
                         scan_aux, kReplayCap, deterministic ? split_meta : nullptr);
      if (block_bins) {
        hipLaunchKernelGGL(msda_bwd_bin_block_kernel<true>, dim3((unsigned)grid_b), dim3(kBinBlockThreads), lds_b,
                           st, spatial_shapes, sampling_loc, attn_weight, flags, cursor, items, M, L, Lq, P,
                           w.bins_per_bm, (int)blocks_per_bml);
      } else {
        hipLaunchKernelGGL(msda_bwd_bin_kernel<true>, grid_s, dim3(256), 0, st, spatial_shapes, sampling_loc,
                           attn_weight, flags, cursor, items, M, L, Lq, P, w.bins_per_bm, w.n_samples);
      }
      if (deterministic) {
        const hipError_t ed = hipMemsetAsync(work_count + 1, 0, sizeof(int), st);
        if (ed != hipSuccess) {
          if (own) (void)hipFreeAsync(ws, st);
          set_error("ms_deform_attn_backward: deterministic mode set-up failed: %s", hipGetErrorString(ed));
          return OCC_E_LAUNCH;
        }
        hipLaunchKernelGGL(msda_bwd_maxabs_kernel, dim3(1024), dim3(256), 0, st, grad_output, n_items * 32,
                           reinterpret_cast<unsigned*>(work_count + 1));
        hipLaunchKernelGGL(msda_bwd_det_zero_kernel, dim3((unsigned)w.max_split), dim3(256), 0, st, work_count, split_meta,
                           tiles);
        hipLaunchKernelGGL(msda_bwd_replay_det_kernel, dim3((unsigned)w.work_cap), dim3(256), 0, st,
                           spatial_shapes, level_start_index, work, work_count, items, grad_output, grad_value, S, M,
                           L, Lq, w.bins_per_bm, split_meta, tiles);
      } else {
        hipLaunchKernelGGL(msda_bwd_replay_kernel, dim3((unsigned)w.work_cap), dim3(256), 0, st,
                           spatial_shapes, level_start_index, work, work_count, items, grad_output, grad_value, S, M,
                           L, Lq, w.bins_per_bm);
      }
      if (own) (void)hipFreeAsync(ws, st);
    } else {
      if (own && ws) (void)hipFreeAsync(ws, st);
      (void)hipGetLastError();
      if (deterministic) {
        set_error("ms_deform_attn_backward: OCC_MSDA_BWD_DETERMINISTIC=1 needs the binned path (%.2f GB of scratch, "
                  "OCC_MSDA_BWD_ATOMICS unset)", (double)w.bytes / 1e9);
        return OCC_E_UNSUPPORTED;
      }
      if (!use_atomics) {
        static bool warned = false;
        if (!warned) {
          warned = true;
          fprintf(stderr, "occnet_amd: ms_deform_attn_backward: no %.2f GB of scratch for the atomic-free grad_value "
                          "path (or shapes beyond its index range) - falling back to float atomics (~4x slower)\n",
                  (double)w.bytes / 1e9);
        }
      }
      hipLaunchKernelGGL(msda_bwd_d32_kernel<true>, grid1, dim3(256), 0, st, value, spatial_shapes,
                         level_start_index, sampling_loc, attn_weight, grad_output, grad_value,
                         grad_sampling_loc, grad_attn_weight, nullptr, S, M, L, Lq, P, n_items);
    }
  } else {
    {
      const char* det_env = getenv("OCC_MSDA_BWD_DETERMINISTIC");
      if (det_env != nullptr && det_env[0] == '1') {     // the generic kernel sums with float atomics: say so, do not pretend
        set_error("ms_deform_attn_backward: OCC_MSDA_BWD_DETERMINISTIC=1 exists for D = 32 only (D = %d)", D);
        return OCC_E_UNSUPPORTED;
      }
    }
    const int per_block = 256 / D > 0 ? 256 / D : 1;
    const int threads = per_block * D;
    const long blocks = (n_items + per_block - 1) / per_block;
    hipLaunchKernelGGL(msda_bwd_generic_kernel, dim3((unsigned)blocks), dim3(threads),
                       3 * threads * sizeof(float), st, value, spatial_shapes, level_start_index,
                       sampling_loc, attn_weight, grad_output, grad_value, grad_sampling_loc,
                       grad_attn_weight, S, M, D, L, Lq, P, n_items);
  }
  OCC_CHECK_LAUNCH("ms_deform_attn_backward");
  return OCC_OK;
}

// mmcv's exact argument list (no workspace argument): the library allocates the scratch itself.
extern "C" int occ_ms_deform_attn_backward_f32(
    const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
    const float* sampling_loc, const float* attn_weight, const float* grad_output, float* grad_value,
    float* grad_sampling_loc, float* grad_attn_weight, int B, int S, int M, int D, int L, int Lq,
    int P, int im2col_step, void* stream) {
  return occ_ms_deform_attn_backward_ws_f32(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                            grad_output, grad_value, grad_sampling_loc, grad_attn_weight, B, S, M,
                                            D, L, Lq, P, im2col_step, nullptr, 0, stream);
}
