// Rate of random byte stores and random 2-byte loads as a function of the footprint they fall
// into (L2-resident ... far larger than L2 and the 256 MB Infinity Cache).
// Build: hipcc --offload-arch=gfx950 -O3 tools/random_access_ubench.hip -o tools/_build/random_access_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void __launch_bounds__(256) k_store(unsigned char *buf, unsigned long long mask, unsigned iters) {
  unsigned long long x = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1;
  for (unsigned i = 0; i < iters; i++) { x = x * 6364136223846793005ull + 1442695040888963407ull; buf[(x >> 20) & mask] = (unsigned char)i; }
}
__global__ void __launch_bounds__(256) k_load(const unsigned short *buf, unsigned long long mask, unsigned iters, unsigned *out) {
  unsigned long long x = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1;
  unsigned acc = 0;
  for (unsigned i = 0; i < iters; i += 4) {  // four independent loads in flight per lane
    unsigned long long a[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { x = x * 6364136223846793005ull + 1442695040888963407ull; a[j] = (x >> 20) & mask; }
#pragma unroll
    for (int j = 0; j < 4; j++) acc += buf[a[j]];
  }
  if (acc == 0x12345u) out[0] = acc;
}

int main() {
  const size_t cap = (size_t)1 << 30;
  unsigned char *buf; unsigned *out;
  CK(hipMalloc(&buf, cap)); CK(hipMalloc(&out, 4)); CK(hipMemset(buf, 1, cap));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned grid = 4096, iters = 512;
  const double n = (double)grid * 256 * iters;
  for (size_t bytes : {(size_t)1 << 20, (size_t)4 << 20, (size_t)32 << 20, (size_t)256 << 20, (size_t)1 << 30}) {
    float ms_s = 0, ms_l = 0;
    for (int rep = 0; rep < 2; rep++) {
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_store, dim3(grid), dim3(256), 0, 0, buf, (unsigned long long)bytes - 1, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_s, e0, e1));
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_load, dim3(grid), dim3(256), 0, 0, (const unsigned short *)buf, (unsigned long long)bytes / 2 - 1, iters, out);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_l, e0, e1));
    }
    printf("footprint %5zu MB: random byte stores %6.1f G/s   random 2-byte loads %6.1f G/s\n", bytes >> 20, n / ms_s / 1e6, n / ms_l / 1e6);
  }
  return 0;
}
