// Range-safe fp16 value rows for the SCA gather (the reference keeps these rows in fp32:
// `spatial_cross_attention.py:75,387-390`, @force_fp32).
//
// The projected value maps are stored as fp16 (sca_fused.hip): 11 significant bits, largest finite value 65 504.
// A plane whose values pass that limit would be clamped silently (round 4 warned about |v| = 1.8e4 on the benchmarked
// maps: 3.6x below the limit).  Here every plane gets a POWER-OF-TWO scale s, chosen per call and on the device from
// an a-priori bound of the plane's values, so that no finite input can saturate:
//
//     |v[n]| = |sum_k x[k] W[n][k] + gbias[n]|  <=  max|x| * max_n sum_k |W[n][k]|  +  max|gbias|  =: bound
//     s = 2^(15 - e),  bound = m * 2^e with m in [0.5, 1)      =>      |v * s| <= 2^15 = half of the fp16 limit
//
// max|x| comes from the PRODUCER of the maps when it is this library's backbone plan (round 6: the FPN output
// convolutions fold the sign-stripped bf16 patterns of what they store into 8 device words — conv3x3_nhwc_bf16.hip —
// and occ_value_range_scale_from_amax derives the scales in one 64-thread launch), or is measured here for foreign maps
// (one pass over the bf16 feature maps, 16-bit integer maxima of the sign-stripped patterns: two elements per VALU
// op, HBM/MALL-bound: 52 us for the 95 MB of the base maps).  The two weight-side terms are constants of the weight
// state and are read from DEVICE memory (round 6: no host round trip, graph-safe, follow the live weights).  The projection multiplies its fp32 accumulators by s before the fp16 conversion, the gather divides
// its fp32 result by count * s: both are exact (power of two), so the only effect of the scale is WHERE the fp16
// exponent window sits — fp16's relative precision is the same anywhere in its normal range (2^-14 .. 2^16), and with
// the bound at 2^15 values down to 2^-29 of the bound are still normal numbers.  Inf / NaN in the maps give s = 1 (such
// rows are Inf / NaN in the fp32 path as well).
//
// One kernel: the blocks fold their maxima into work[0] (atomic max), take a ticket from work[1], and the last block to
// finish derives the scales.  The launcher zeroes both words in front of the kernel (a 8-byte memset node: a stale
// ticket word — an aborted launch, a foreign writer — can then not leave scale_out unwritten).
#include "common.h"

namespace occ {

constexpr int kVrMaxSeg = 8, kVrMaxPlanes = 8, kVrUnroll = 12;   // 12 x 2048 x 256 pieces = 100 MB per round: the base maps in ONE round
struct VrSegments {
  const uint4* a[kVrMaxSeg];
  int first[kVrMaxSeg + 1];           // cumulative 16-byte pieces: segment s holds pieces [first[s], first[s + 1])
  int lda8[kVrMaxSeg];
  int n;
};
constexpr int kVrAmaxWords = 8;       // producer-side maxima are sharded over 8 words (block id & 7): less same-address traffic

typedef unsigned short vr_u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned vr_absmax2(unsigned m, unsigned x) {   // v_and + v_pk_max_u16
  const vr_u16x2 a = __builtin_bit_cast(vr_u16x2, m), b = __builtin_bit_cast(vr_u16x2, x & 0x7fff7fffu);
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(a, b));
}
__device__ __forceinline__ unsigned vr_absmax8(unsigned m, const uint4 v) {
  return vr_absmax2(vr_absmax2(vr_absmax2(vr_absmax2(m, v.x), v.y), v.z), v.w);
}

// s = 2^(15 - e) for bound = m 2^e; 1 when the bound is zero, Inf or NaN
__device__ __forceinline__ float vr_scale_of(float bound) {
  if (!(bound > 0.f) || !(bound < __builtin_huge_valf())) return 1.f;
  int e;
  (void)frexpf(bound, &e);
  int k = 15 - e;
  k = k < -100 ? -100 : k > 100 ? 100 : k;       // s and 1/s stay normal fp32 numbers
  return ldexpf(1.f, k);
}

// scale_out[0..P) = scales, [P] = max|x|, [P+1..2P] = bounds (diagnostics)
__device__ __forceinline__ void vr_write_scales(unsigned bits, int n_planes, const float* __restrict__ row_l1,
                                                const float* __restrict__ bias_max, float* __restrict__ scale_out) {
  const float amax = __uint_as_float(bits << 16);        // bf16 pattern -> f32 (Inf / NaN patterns stay what they are)
  scale_out[n_planes] = amax;
  for (int p = 0; p < n_planes; ++p) {
    // 2^-8 of slack on the weight term: the kernel's hi/lo bf16 weight split and fp32 accumulation are exact to
    // 2^-17 of |x|.|w|, far inside it (and the target leaves another factor of two below the fp16 limit)
    const float bound = fabsf(row_l1[p]) * amax * 1.00390625f + fabsf(bias_max[p]);     // (|.|: a negative term is a caller bug)
    scale_out[p] = vr_scale_of(bound);
    scale_out[n_planes + 1 + p] = bound;
  }
}

// the maxima arrive from the producer of the maps (8 words of sign-stripped bf16 patterns): one thread
__global__ void value_range_from_amax_kernel(const unsigned* __restrict__ amax8, int n_planes,
                                             const float* __restrict__ row_l1, const float* __restrict__ bias_max,
                                             float* __restrict__ scale_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  unsigned b = 0;
  for (int i = 0; i < kVrAmaxWords; ++i) b = amax8[i] > b ? amax8[i] : b;
  vr_write_scales(b & 0xffffu, n_planes, row_l1, bias_max, scale_out);
}

template <bool CONTIG>
__global__ __launch_bounds__(256) void value_range_scale_kernel(VrSegments seg, int K8, int n_planes,
                                                                const float* __restrict__ row_l1,
                                                                const float* __restrict__ bias_max,
                                                                float* __restrict__ scale_out,
                                                                unsigned* __restrict__ work) {
  __shared__ unsigned smax[4];
  const int stride = (int)gridDim.x * 256;
  const int total = seg.first[seg.n];            // < 2^31 - kVrUnroll * stride (the launcher checks): 32-bit index arithmetic
  unsigned m = 0;
  // ONE index space over all segments, kVrUnroll independent 16-byte loads per thread and round: the first cut walked the
  // segments one after the other with a one-load-at-a-time tail per segment — six serial memory round trips per thread,
  // 85-141 us for the 95 MB of the base maps (profiles/r05_c2_hot_kernel_trace_stats.txt)
  for (int base = (int)blockIdx.x * 256 + (int)threadIdx.x; base < total; base += stride * kVrUnroll) {
    uint4 v[kVrUnroll];
#pragma unroll
    for (int k = 0; k < kVrUnroll; ++k) {
      // branch-free: a piece beyond the index space re-reads the last piece (same maximum) instead of being skipped —
      // with a branch per load hipcc waits for every load before it issues the next one
      int g = base + k * stride;
      g = g < total ? g : total - 1;
      // the segment of g by a select chain over CONSTANT table indices (a per-lane index into the by-value table would be
      // fetched with vector loads from the kernel-argument segment: two more dependent round trips per piece)
      const uint4* ap = seg.a[0];
      int first = 0, lda8 = seg.lda8[0];
#pragma unroll
      for (int i = 1; i < kVrMaxSeg; ++i) {
        const bool hit = i < seg.n && g >= seg.first[i];
        ap = hit ? seg.a[i] : ap;
        first = hit ? seg.first[i] : first;
        if (!CONTIG) lda8 = hit ? seg.lda8[i] : lda8;
      }
      const int local = g - first;
      long off = local;
      if (!CONTIG) {                  // strided rows: (row, piece) from the flat index
        const int r = local / K8;
        off = (long)r * lda8 + (local - r * K8);
      }
      v[k] = ap[off];
    }
#pragma unroll
    for (int k = 0; k < kVrUnroll; ++k) m = vr_absmax8(m, v[k]);
  }
  unsigned m16 = (m & 0xffffu) > (m >> 16) ? (m & 0xffffu) : (m >> 16);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned o = (unsigned)__shfl_xor((int)m16, d);
    m16 = o > m16 ? o : m16;
  }
  if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = m16;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned b = smax[0];
    for (int w = 1; w < 4; ++w) b = smax[w] > b ? smax[w] : b;
    // No fences: both words are only ever touched by device-scope atomics (performed at the coherence point, not in an XCD's
    // L2), and the ticket is requested only after the maximum's RETURN value has arrived — i.e. after that atomic has been
    // performed.  The ticket's operand is made to depend on that return value (an asm the compiler cannot see through, with a
    // memory clobber), so neither the compiler nor the memory pipeline can put the ticket in front of the maximum.  (The
    // first cut put a __threadfence() around the ticket: an L2 write-back + invalidate per block, 2 048 of them as the
    // launch drains: 85-114 us for a 95 MB stream.)
    const unsigned old = atomicMax(&work[0], b);
    unsigned one = 1u;
    asm volatile("" : "+v"(one) : "v"(old) : "memory");
    const unsigned ticket = atomicAdd(&work[1], one);
    if (ticket == gridDim.x - 1) {
      const unsigned bits = atomicMax(&work[0], 0u);         // read at the coherence point
      vr_write_scales(bits, n_planes, row_l1, bias_max, scale_out);
    }
  }
}

}  // namespace occ

extern "C" int occ_value_range_scale_bf16(int n_segments, const void* const* a, const int64_t* lda,
                                          const int64_t* rows, int K, int n_planes, const float* row_l1,
                                          const float* bias_max, float* scale_out, uint32_t* work, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(a && lda && rows && row_l1 && bias_max && scale_out && work,
                "value_range_scale: null pointer argument");
  OCC_CHECK_ARG(n_segments > 0 && n_segments <= kVrMaxSeg, "value_range_scale: 1..%d segments", kVrMaxSeg);
  OCC_CHECK_ARG(n_planes > 0 && n_planes <= kVrMaxPlanes, "value_range_scale: 1..%d planes", kVrMaxPlanes);
  OCC_CHECK_ARG(K > 0, "value_range_scale: bad K");
  if (K % 8) {
    set_error("value_range_scale: K=%d is not a multiple of 8 (16-byte pieces)", K);
    return OCC_E_UNSUPPORTED;
  }
  VrSegments seg;
  long pieces = 0;
  for (int i = 0; i < kVrMaxSeg; ++i) {
    const int j = i < n_segments ? i : n_segments - 1;
    OCC_CHECK_ARG(a[j] && rows[j] > 0 && lda[j] >= K, "value_range_scale: bad segment %d", j);
    if (lda[j] % 8) {
      set_error("value_range_scale: segment %d row stride %ld is not 16-byte aligned", j, (long)lda[j]);
      return OCC_E_UNSUPPORTED;
    }
    seg.a[i] = reinterpret_cast<const uint4*>(a[j]);
    OCC_CHECK_ARG(lda[j] / 8 < (1L << 31) && pieces + rows[j] * (K / 8) < (1L << 31) - (long)kVrUnroll * 2048 * 256,
                  "value_range_scale: more than 2^31 16-byte pieces");
    seg.lda8[i] = (int)(lda[j] / 8);
    seg.first[i] = (int)pieces;
    if (i < n_segments) pieces += rows[j] * (K / 8);
  }
  for (int i = n_segments; i <= kVrMaxSeg; ++i) seg.first[i] = (int)pieces;
  seg.n = n_segments;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (hipMemsetAsync(work, 0, 2 * sizeof(uint32_t), st) != hipSuccess) {
    (void)hipGetLastError();
    set_error("value_range_scale: could not zero the work words");
    return OCC_E_LAUNCH;
  }
  // kVrUnroll 16-byte loads per thread and round; no more blocks than one round needs, at most 8 per CU
  long blocks = (pieces + kVrUnroll * 256 - 1) / (kVrUnroll * 256);
  blocks = blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks;
  bool contig = true;
  for (int i = 0; i < n_segments; ++i) contig = contig && lda[i] == K;
  if (contig)
    hipLaunchKernelGGL(value_range_scale_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, seg, K / 8, n_planes,
                       row_l1, bias_max, scale_out, work);
  else
    hipLaunchKernelGGL(value_range_scale_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, seg, K / 8, n_planes,
                       row_l1, bias_max, scale_out, work);
  OCC_CHECK_LAUNCH("value_range_scale");
  return OCC_OK;
}

// The same scales from maxima the PRODUCER of the maps accumulated (amax8: 8 words, each the largest sign-stripped bf16
// pattern a shard of the producer's blocks stored — occ_conv3x3_nhwc_bf16_amax): no pass over the maps.
extern "C" int occ_value_range_scale_from_amax(const uint32_t* amax8, int n_planes, const float* row_l1,
                                               const float* bias_max, float* scale_out, void* stream) {
  using namespace occ;
  OCC_CHECK_ARG(amax8 && row_l1 && bias_max && scale_out, "value_range_scale_from_amax: null pointer argument");
  OCC_CHECK_ARG(n_planes > 0 && n_planes <= kVrMaxPlanes, "value_range_scale_from_amax: 1..%d planes", kVrMaxPlanes);
  hipLaunchKernelGGL(value_range_from_amax_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), amax8,
                     n_planes, row_l1, bias_max, scale_out);
  OCC_CHECK_LAUNCH("value_range_scale_from_amax");
  return OCC_OK;
}
