// Row-store patterns of the chain kernels in isolation (round 4): 625 blocks x 256 threads write a (40 000, 768) f32 matrix,
// 64 rows x 256 columns per block and pass, three passes.
//   quad : the MFMA accumulator layout — a store instruction covers 32 rows x 32 contiguous bytes (lane pair kb = 0, 1)
//   seg64: 16 rows x 64 contiguous bytes per instruction (what a lane exchange vi <-> vi ^ 16 would give)
//   row  : one whole 1 KB row segment per instruction (what an LDS transpose would give)
// hipcc --offload-arch=gfx950 -O3 -o /tmp/sp tools_dev/lab/store_pattern_probe.hip && /tmp/sp
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* z, int ld, int M) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, vi = lane & 31, kb = lane >> 5;
  const long m0 = (long)blockIdx.x * 64;
  const float4 v = make_float4((float)tid, 1.f, 2.f, 3.f);
  for (int ps = 0; ps < 3; ++ps) {
    if (MODE == 0) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const long r = m0 + rt * 32 + vi;
            if (r < M) *reinterpret_cast<float4*>(z + r * ld + ps * 256 + wave * 64 + 32 * t + 8 * q + 4 * kb) = v;
          }
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {                 // 16 rows x 64 B per instruction: lane = (row r16 = lane >> 2, piece = lane & 3)
        const long r = m0 + (i & 3) * 16 + (lane >> 2);
        if (r < M) *reinterpret_cast<float4*>(z + r * ld + ps * 256 + wave * 64 + (i >> 2) * 16 + (lane & 3) * 4) = v;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const long r = m0 + j * 4 + wave;
        if (r < M) *reinterpret_cast<float4*>(z + r * ld + ps * 256 + lane * 4) = v;
      }
    }
  }
}
template <int MODE> float run(float* z, int M) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int nb = (M + 63) / 64;
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), 0, 0, z, 768, M);
  hipEventRecord(a);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), 0, 0, z, 768, M);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 20 * 1e3f;
}
int main() {
  const int M = 40000; float* z; hipMalloc(&z, (size_t)M * 768 * 4);
  for (int rep = 0; rep < 2; ++rep) {
    const float t0 = run<0>(z, M), t1 = run<1>(z, M), t2 = run<2>(z, M);
    const double gb = (double)M * 768 * 4 / 1e3;
    printf("123 MB written:  quad %6.1f us (%5.2f TB/s)   seg64 %6.1f us (%5.2f TB/s)   row %6.1f us (%5.2f TB/s)\n", t0, gb / t0 / 1e6 * 1e0,
           t1, gb / t1 / 1e6, t2, gb / t2 / 1e6);
  }
  return 0;
}
