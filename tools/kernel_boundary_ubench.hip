// Does a kernel boundary on ONE stream disturb the L2 contents another stream's kernel lives on?
// Kernel A re-reads a small (L2-resident) buffer with dependent gathers and reports its rate;
// it runs alone, and then while a second stream launches back-to-back tiny kernels (which write
// a little, so every one of them ends with a release and starts with an acquire).
// Build: hipcc --offload-arch=gfx950 -O3 tools/kernel_boundary_ubench.hip -o tools/_build/kernel_boundary_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void __launch_bounds__(256) k_gather(const unsigned *buf, unsigned mask, unsigned iters, unsigned *out) {
  unsigned x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u;
  for (unsigned i = 0; i < iters; i++) x = buf[(x >> 4) & mask] + i;  // dependent random reads inside the buffer
  if (x == 0x12345u) out[0] = x;
}
// read-modify-write of random bytes inside a small buffer: lives on L2 write combining
__global__ void __launch_bounds__(256) k_scatter_bytes(unsigned char *buf, unsigned mask, unsigned iters) {
  unsigned x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u;
  for (unsigned i = 0; i < iters; i++) { x = x * 1664525u + 1013904223u; buf[(x >> 8) & mask] = (unsigned char)i; }
}
__global__ void k_tiny(unsigned *p, unsigned v) { p[threadIdx.x] = v; }

int main() {
  const unsigned words = 1u << 19;  // 2 MB
  unsigned *buf, *out, *tiny; unsigned char *bytes;
  CK(hipMalloc(&buf, words * 4)); CK(hipMalloc(&out, 4)); CK(hipMalloc(&tiny, 4096)); CK(hipMalloc(&bytes, 64u << 20));
  CK(hipMemset(buf, 0x5a, words * 4)); CK(hipMemset(bytes, 0, 64u << 20));
  hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int test = 0; test < 2; test++) {
    for (int with_tiny = 0; with_tiny < 2; with_tiny++) {
      for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0, sa));
        if (test == 0) hipLaunchKernelGGL(k_gather, dim3(2048), dim3(256), 0, sa, buf, words - 1, 4000u, out);
        else hipLaunchKernelGGL(k_scatter_bytes, dim3(2048), dim3(256), 0, sa, bytes, (32u << 20) - 1, 2000u);
        CK(hipEventRecord(e1, sa));
        int launched = 0;
        if (with_tiny)
          while (hipEventQuery(e1) == hipErrorNotReady) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, sb, tiny, (unsigned)launched); launched++; }
        CK(hipEventSynchronize(e1)); CK(hipStreamSynchronize(sb));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 1)
          printf("%-34s %s: %.3f ms  (%d tiny kernels alongside)\n", test == 0 ? "gathers in a 2 MB buffer" : "byte scatter in a 32 MB buffer",
                 with_tiny ? "with tiny kernels on another stream" : "alone", ms, launched);
      }
    }
  }
  return 0;
}
