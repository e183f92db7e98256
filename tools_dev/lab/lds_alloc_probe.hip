// What does HW_REG_LDS_ALLOC hold for the blocks of a launch with two 78 KB blocks per CU?  (round 4: the chain kernels
// use "LDS base != 0" to tell the two co-resident blocks of a CU apart.)  hipcc --offload-arch=gfx950 -o /tmp/p this && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256, 2) void probe(unsigned* out) {
  extern __shared__ char lds[];
  unsigned la, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(la));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  lds[threadIdx.x] = (char)la;
  __syncthreads();
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = la; out[2 * blockIdx.x + 1] = hw; }
  // stay resident long enough that the whole first round overlaps
  for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(127);
  if (lds[threadIdx.x ^ 1] == 77) out[0] = 0;
}
int main() {
  const int n = 640;
  unsigned* d; hipMalloc(&d, n * 8);
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 77824);
  hipLaunchKernelGGL(probe, dim3(n), dim3(256), 77824, 0, d);
  unsigned h[2 * n]; hipMemcpy(h, d, n * 8, hipMemcpyDeviceToHost);
  int nz = 0;
  for (int i = 0; i < n; ++i) nz += (h[2 * i] & 0xff) != 0;
  printf("blocks with LDS base != 0: %d of %d (first 512: ", nz, n);
  int nz1 = 0; for (int i = 0; i < 512; ++i) nz1 += (h[2 * i] & 0xff) != 0;
  printf("%d)\n", nz1);
  for (int i = 0; i < 24; ++i) printf("block %3d: LDS_ALLOC %08x HW_ID %08x\n", i, h[2 * i], h[2 * i + 1]);
  for (int i = 508; i < 520; ++i) printf("block %3d: LDS_ALLOC %08x HW_ID %08x\n", i, h[2 * i], h[2 * i + 1]);
  return 0;
}
