// Calibration: what do plain streaming stores cost on this GPU?  16-byte, 2-byte and 1-byte per
// lane stores of the same buffer, and a copy (read + write), all coalesced.
// Build: hipcc --offload-arch=gfx950 -O3 tools/write_bw_ubench.hip -o tools/_build/write_bw_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <typename T>
__global__ void __launch_bounds__(256) k_fill(T *p, size_t n, T v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void __launch_bounds__(256) k_copy(const uint4 *a, uint4 *b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void __launch_bounds__(256) k_read(const uint4 *a, size_t n, unsigned *out) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = a[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const size_t bytes = (size_t)2 << 30;
  uint8_t *a, *b; unsigned *o;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 4));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](const char *name, double gb, auto launch) {
    float best = 1e9;
    for (int r = 0; r < 4; r++) {
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-28s %7.3f ms  %7.1f GB/s\n", name, best, gb / (best * 1e-3));
  };
  const double GB = bytes / 1e9;
  for (int grid : {2048, 16384}) {
    printf("grid %d x 256\n", grid);
    time("store 16 B per lane", GB, [&] { hipLaunchKernelGGL(k_fill<uint4>, dim3(grid), dim3(256), 0, 0, (uint4 *)a, bytes / 16, make_uint4(1, 2, 3, 4)); });
    time("store 4 B per lane", GB, [&] { hipLaunchKernelGGL(k_fill<unsigned>, dim3(grid), dim3(256), 0, 0, (unsigned *)a, bytes / 4, 7u); });
    time("store 2 B per lane", GB, [&] { hipLaunchKernelGGL(k_fill<uint16_t>, dim3(grid), dim3(256), 0, 0, (uint16_t *)a, bytes / 2, (uint16_t)7); });
    time("store 1 B per lane", GB / 2, [&] { hipLaunchKernelGGL(k_fill<uint8_t>, dim3(grid), dim3(256), 0, 0, a, bytes / 2, (uint8_t)7); });
    time("read 16 B per lane", GB, [&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, (const uint4 *)a, bytes / 16, o); });
    time("copy 16 B per lane (r+w)", 2 * GB, [&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, (const uint4 *)a, (uint4 *)b, bytes / 16); });
  }
  return 0;
}
