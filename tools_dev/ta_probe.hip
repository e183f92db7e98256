// Row-gather rate probe for gfx950 (development tool, not part of libocc_amd.so).
// Question it answers: how many 128-byte rows per clock can one CU pull through the texture-addresser / L1 path when
// every 8-lane group of a wave fetches a different row (the access shape of the SCA / TSA gathers), as a function of
// where the rows live (L1, L2, MALL, HBM)?  The SCA gather's `frac_l1` is quoted against 64 B/clk/CU; this measures
// what that path actually delivers for the gather's access shape.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools_dev/bin/ta_probe tools_dev/ta_probe.hip
//   run:   tools_dev/bin/ta_probe            (prints one line per (pattern, table size))
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// PATTERN 0: 8 lanes x 16 B per 128-byte row, random row per 8-lane group (fp32 value rows)
// PATTERN 1: 8 lanes x  8 B per  64-byte row, random row per 8-lane group (fp16 value rows)
// PATTERN 2: 64 lanes x 16 B contiguous (8 consecutive rows), random start per wave (coalesced reference)
// PATTERN 3: like 0, but the 8 groups of a wave read 8 CONSECUTIVE rows of a random start (one head-major octet)
// PATTERN 4: like 0 with LOCAL randomness: row = base(wave, step) + small random offset (window of 64 rows = 8 KB)
template <int PATTERN>
__global__ __launch_bounds__(256) void gather_kernel(const char* __restrict__ table, uint32_t n_rows, int iters,
                                                     float* __restrict__ out) {
  extern __shared__ float occupancy_limiter[];          // dynamic LDS sized by the host to cap the resident blocks
  if (iters < 0) occupancy_limiter[threadIdx.x] = 0.f;
  constexpr int INFLIGHT = 16;
  const uint32_t lane = threadIdx.x & 63, grp = lane >> 3, sub = lane & 7;
  const uint32_t wave_id = blockIdx.x * 4 + (threadIdx.x >> 6);
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  uint32_t seed = mix(wave_id * 2654435761u + 12345u);
  for (int it = 0; it < iters; ++it) {
    f4 v[INFLIGHT];
#pragma unroll
    for (int k = 0; k < INFLIGHT; ++k) {
      const uint32_t step = (uint32_t)it * INFLIGHT + k;
      uint32_t row;
      if (PATTERN == 0 || PATTERN == 1) row = mix(seed + step * 8u + grp) % n_rows;
      else if (PATTERN == 2 || PATTERN == 3) row = (mix(seed + step) % (n_rows - 8)) + grp;
      else { const uint32_t base = mix(seed + (step >> 2)) % (n_rows - 64); row = base + (mix(seed ^ (step * 8u + grp)) & 63u); }
      if (PATTERN == 1) {
        const f2 t = *reinterpret_cast<const f2*>(table + (size_t)row * 64 + sub * 8);
        v[k] = f4{t.x, t.y, 0.f, 0.f};
      } else {
        v[k] = *reinterpret_cast<const f4*>(table + (size_t)row * 128 + sub * 16);
      }
    }
#pragma unroll
    for (int k = 0; k < INFLIGHT; ++k) acc += v[k];
  }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int PATTERN>
static void run(const char* name, const char* table, size_t table_bytes, float* out, int blocks, int iters, double clk_ghz, int cus, int lds) {
  const int row_bytes = PATTERN == 1 ? 64 : 128;
  const uint32_t n_rows = (uint32_t)(table_bytes / row_bytes);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(gather_kernel<PATTERN>, dim3(blocks), dim3(256), lds, 0, table, n_rows, iters, out);
  CK(hipDeviceSynchronize());
  const int reps = 5; float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(gather_kernel<PATTERN>, dim3(blocks), dim3(256), lds, 0, table, n_rows, iters, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const double rows = (double)blocks * 4 * 8 * 16.0 * iters;          // waves x groups x in-flight x iters
  const double sec = best * 1e-3;
  printf("{\"pattern\": \"%s\", \"table_bytes\": %zu, \"row_bytes\": %d, \"ms\": %.4f, \"rows_per_s\": %.4g, \"GBps\": %.1f, "
         "\"rows_per_clk_per_cu\": %.4f, \"bytes_per_clk_per_cu\": %.2f}\n",
         name, table_bytes, row_bytes, best, rows / sec, rows * row_bytes / sec / 1e9,
         rows / sec / (clk_ghz * 1e9) / cus, rows * row_bytes / sec / (clk_ghz * 1e9) / cus);
  fflush(stdout);
}

int main(int argc, char** argv) {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount; const double ghz = p.clockRate / 1e6;
  printf("# %s, %d CUs, %.2f GHz; blocks of 4 waves, 16 loads in flight per lane\n", p.name, cus, ghz);
  const size_t max_bytes = (size_t)1 << 30;
  char* table; CK(hipMalloc(&table, max_bytes)); CK(hipMemset(table, 0, max_bytes));
  const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 3;            // resident waves per SIMD to emulate (3 = the SCA kernel)
  const int blocks = cus * waves_per_simd * 8;                         // 8 rounds of resident blocks
  const int lds = (160 * 1024) / waves_per_simd - 1024;                // one block = one wave per SIMD; LDS caps blocks/CU
  printf("# %d resident waves per SIMD (dynamic LDS %d bytes per block)\n", waves_per_simd, lds);
  float* out; CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  const size_t sizes[] = {(size_t)16 << 10, (size_t)256 << 10, (size_t)2 << 20, (size_t)24 << 20, (size_t)190 << 20, (size_t)1 << 30};
  for (size_t sz : sizes) {
    const int iters = 64;
    run<0>("rows128_random", table, sz, out, blocks, iters, ghz, cus, lds);
    run<1>("rows64_random_b64", table, sz, out, blocks, iters, ghz, cus, lds);
    run<2>("coalesced_1KB", table, sz, out, blocks, iters, ghz, cus, lds);
    run<3>("rows128_8consecutive", table, sz, out, blocks, iters, ghz, cus, lds);
    run<4>("rows128_local_window", table, sz, out, blocks, iters, ghz, cus, lds);
  }
  return 0;
}
