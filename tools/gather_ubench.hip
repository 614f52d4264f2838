// Micro-benchmark: throughput of 64-lane random u16 LDS gathers (the inner operation of a
// candidate-set walk): WAVES waves per workgroup share one 16 KB next[s][x] table, every lane
// advances M independent states per symbol.  Reports wave-gathers per cycle per CU.
// Two tables: "random" (a random function: the lanes' states coalesce within a few steps, most
// lanes then read the same few dwords and the LDS broadcasts) and "permutation" (every row a
// bijection: 64 lanes keep 64 DISTINCT states, the situation of the state-set walk, where the
// classes a wave carries are distinct by construction).
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_ubench.hip -o tools/_build/gather_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr unsigned LOG = 11, SIZE = 1u << LOG;

template <int M>
__global__ void __launch_bounds__(1024) k(const uint16_t *tab, const uint32_t *sym, unsigned n_words, unsigned *out) {
  __shared__ uint16_t next[4 * SIZE];
  for (unsigned e = threadIdx.x; e < 4 * SIZE; e += blockDim.x) next[e] = tab[e];
  __syncthreads();
  const char *nb = reinterpret_cast<const char *>(next);
  unsigned x[M];
#pragma unroll
  for (int j = 0; j < M; j++) x[j] = ((threadIdx.x * M + j) * 2) & (2 * SIZE - 1);
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t *my = sym + (size_t)(blockIdx.x * (blockDim.x >> 6) + wave) * n_words;
  for (unsigned w = 0; w < n_words; w++) {
    const unsigned word = __builtin_amdgcn_readfirstlane(my[w]);  // uniform: 16 symbols
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const unsigned s = (word >> (2 * i)) & 3u;
#pragma unroll
      for (int j = 0; j < M; j++) x[j] = *reinterpret_cast<const uint16_t *>(nb + ((s << (LOG + 1)) + x[j]));
    }
  }
  unsigned acc = 0;
#pragma unroll
  for (int j = 0; j < M; j++) acc += x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
  const unsigned n_words = 2048;  // 32768 symbols per wave
  const bool perm = argc > 1 && argv[1][0] == 'p';
  std::vector<uint16_t> tab(4 * SIZE);
  srand(1);
  for (auto &v : tab) v = (uint16_t)((rand() % SIZE) * 2);
  if (perm)
    for (unsigned r = 0; r < 4; r++) {
      std::vector<uint16_t> p(SIZE);
      for (unsigned i = 0; i < SIZE; i++) p[i] = (uint16_t)(i * 2);
      for (unsigned i = SIZE - 1; i > 0; i--) std::swap(p[i], p[rand() % (i + 1)]);
      for (unsigned i = 0; i < SIZE; i++) tab[r * SIZE + i] = p[i];
    }
  printf("table: %s\n", perm ? "permutation rows (lanes keep distinct states)" : "random function (states coalesce)");
  const unsigned max_waves = 256 * 32 * 2;
  std::vector<uint32_t> sym((size_t)max_waves * n_words);
  for (auto &v : sym) v = (uint32_t)rand() * 2654435761u;
  uint16_t *dtab; uint32_t *dsym; unsigned *dout;
  CK(hipMalloc(&dtab, tab.size() * 2)); CK(hipMalloc(&dsym, sym.size() * 4)); CK(hipMalloc(&dout, (size_t)max_waves * 64 * 4));
  CK(hipMemcpy(dtab, tab.data(), tab.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsym, sym.data(), sym.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int waves : {4, 8, 16}) {
    for (int wg_per_cu : {1, 2}) {
      if (waves * wg_per_cu > 32) continue;
      for (int m : {1, 2, 3, 4, 8}) {
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
          CK(hipEventRecord(a));
          const dim3 g(256 * wg_per_cu), t(64 * waves);
          if (m == 1) hipLaunchKernelGGL(k<1>, g, t, 0, 0, dtab, dsym, n_words, dout);
          if (m == 2) hipLaunchKernelGGL(k<2>, g, t, 0, 0, dtab, dsym, n_words, dout);
          if (m == 3) hipLaunchKernelGGL(k<3>, g, t, 0, 0, dtab, dsym, n_words, dout);
          if (m == 4) hipLaunchKernelGGL(k<4>, g, t, 0, 0, dtab, dsym, n_words, dout);
          if (m == 8) hipLaunchKernelGGL(k<8>, g, t, 0, 0, dtab, dsym, n_words, dout);
          CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
          CK(hipEventElapsedTime(&ms, a, b));
        }
        const double gathers_per_cu = (double)waves * wg_per_cu * n_words * 16 * m;
        const double cycles = ms * 1e-3 * 2.4e9;
        printf("waves/WG %2d WG/CU %d M %d: %.3f ms  %.2f cycles per wave-gather per CU  (%.1f symbol-waves/us/CU)\n", waves,
               wg_per_cu, m, ms, cycles / gathers_per_cu, (double)waves * wg_per_cu * n_words * 16 / (ms * 1e3));
      }
    }
  }
  return 0;
}
