// Micro-benchmark of the serial tANS state chain: how many ns (and shader cycles) does one
// dependent table step cost on an MI355X, for the addressing variants the sequence chain
// kernel could use?  Build: hipcc --offload-arch=gfx950 -O3 tools/chain_ubench.hip -o /tmp/chain_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr unsigned LOG = 11, SIZE = 1u << LOG;

// variant 0: next[s][xi] u16 pre-scaled, symbols from LDS, 1 add + 1 ds_read_u16 per step
template <int VAR>
__global__ void __launch_bounds__(64) k(const uint16_t *tab, const uint8_t *sym, unsigned n, unsigned *out,
                                        unsigned long long *cyc) {
  __shared__ uint16_t next[4 * SIZE];
  __shared__ uint32_t next32[VAR == 2 ? 4 * SIZE : 1];
  __shared__ uint8_t sbuf[4096];
  for (unsigned e = threadIdx.x; e < 4 * SIZE; e += 64) { next[e] = tab[e]; if (VAR == 2) next32[e] = tab[e] * 2; }
  __syncthreads();
  unsigned xo = 0;
  unsigned long long t_acc = 0, r_acc = 0;
  for (unsigned c0 = 0; c0 < n; c0 += 4096) {
    for (unsigned i = threadIdx.x; i < 4096; i += 64) sbuf[i] = sym[c0 + i];
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
      const char *nb = reinterpret_cast<const char *>(next);
      const char *nb32 = reinterpret_cast<const char *>(next32);
      for (unsigned g = 0; g < 4096; g += 16) {
        const uint4 sv = *reinterpret_cast<const uint4 *>(&sbuf[g]);
        const unsigned w[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
        for (int j = 0; j < 16; j++) {
          const unsigned s = (w[j >> 2] >> (8 * (j & 3))) & 3u;
          if (VAR == 0) {  // byte-offset table, add + ds_read_u16
            xo = *reinterpret_cast<const uint16_t *>(nb + ((s << (LOG + 1)) + xo));
          } else if (VAR == 1) {  // index table: shift + add + ds_read_u16
            xo = next[(s << LOG) + xo];
          } else if (VAR == 2) {  // dword table, ds_read_b32
            xo = *reinterpret_cast<const uint32_t *>(nb32 + ((s << (LOG + 2)) + xo));
          } else if (VAR == 3) {  // two lookups per step through global memory (L1/L2 resident)
            xo = tab[(s << LOG) + (xo >> 1)];
          }
        }
      }
      t_acc += __builtin_amdgcn_s_memtime() - t0;
      r_acc += __builtin_amdgcn_s_memrealtime() - r0;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[blockIdx.x] = xo; cyc[2 * blockIdx.x] = t_acc; cyc[2 * blockIdx.x + 1] = r_acc; }
}

// variant 4: two symbols per step through a 64 KB table (16 << LOG u16 entries), as k_chain_seq2
template <int PACK>
__global__ void __launch_bounds__(64) k2(const uint16_t *tab, const uint8_t *sym, unsigned n, unsigned *out,
                                         unsigned long long *cyc, unsigned log) {
  extern __shared__ uint32_t lds[];
  __shared__ uint4 sbuf[4096 / 16];
  __shared__ uint16_t statebuf[2048];
  uint16_t *t2 = reinterpret_cast<uint16_t *>(lds);
  for (unsigned e = threadIdx.x; e < 16 * SIZE; e += 64) t2[e] = tab[e & (4 * SIZE - 1)];
  __syncthreads();
  unsigned xo = 0;
  unsigned long long t_acc = 0, r_acc = 0;
  const char *tbase = reinterpret_cast<const char *>(t2);
  for (unsigned c0 = 0; c0 < n; c0 += 4096) {
    for (unsigned i = threadIdx.x; i < 256; i += 64) sbuf[i] = reinterpret_cast<const uint4 *>(sym + c0)[i];
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
      uint4 *state4 = reinterpret_cast<uint4 *>(statebuf);
      uint4 sv = sbuf[0];
      for (unsigned g = 0; g < 256; g++) {
        const uint4 sv_next = sbuf[g + 1 < 256 ? g + 1 : g];
        const unsigned w[4] = {sv.x, sv.y, sv.z, sv.w};
        unsigned xs[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const unsigned half = w[j >> 1] >> (16 * (j & 1));
          const unsigned pc = (half & 3u) | ((half >> 6) & 0xCu);
          xs[j] = xo;
          xo = *reinterpret_cast<const uint16_t *>(tbase + ((pc << (log + 1)) + xo));
        }
        if (PACK) state4[g] = make_uint4(xs[0] | (xs[1] << 16), xs[2] | (xs[3] << 16), xs[4] | (xs[5] << 16), xs[6] | (xs[7] << 16));
        sv = sv_next;
      }
      t_acc += __builtin_amdgcn_s_memtime() - t0;
      r_acc += __builtin_amdgcn_s_memrealtime() - r0;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[blockIdx.x] = xo + statebuf[5]; cyc[2 * blockIdx.x] = t_acc; cyc[2 * blockIdx.x + 1] = r_acc; }
}

int main() {
  const unsigned n = 1u << 20;
  std::vector<uint16_t> tab(4 * SIZE), tab1(4 * SIZE);
  srand(1);
  for (unsigned s = 0; s < 4; s++)
    for (unsigned x = 0; x < SIZE; x++) { unsigned v = rand() % SIZE; tab[s * SIZE + x] = (uint16_t)(v * 2); tab1[s * SIZE + x] = (uint16_t)v; }
  std::vector<uint8_t> sym(n + 4096);
  for (auto &b : sym) b = rand() & 3;
  uint16_t *dtab, *dtab1; uint8_t *dsym; unsigned *dout; unsigned long long *dcyc;
  CK(hipMalloc(&dtab, tab.size() * 2)); CK(hipMalloc(&dtab1, tab.size() * 2)); CK(hipMalloc(&dsym, sym.size()));
  CK(hipMalloc(&dout, 4096 * 4)); CK(hipMalloc(&dcyc, 4096 * 16));
  CK(hipMemcpy(dtab, tab.data(), tab.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dtab1, tab1.data(), tab.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsym, sym.data(), sym.size(), hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int grid : {1, 256}) {
    for (int var = 0; var < 4; var++) {
      for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(a));
        if (var == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 0, 0, dtab, dsym, n, dout, dcyc);
        if (var == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 0, 0, dtab1, dsym, n, dout, dcyc);
        if (var == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(64), 0, 0, dtab, dsym, n, dout, dcyc);
        if (var == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(64), 0, 0, dtab, dsym, n, dout, dcyc);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        unsigned long long c[2]; CK(hipMemcpy(c, dcyc, 16, hipMemcpyDeviceToHost));
        if (rep == 1)
          printf("grid %4d var %d: %.2f ms  %.1f ns/step (wall)  %.1f shader-cycles/step  in-loop %.1f ns/step  clock %.2f GHz\n",
                 grid, var, ms, ms * 1e6 / n, (double)c[0] / n, (double)c[1] * 10.0 / n, (double)c[0] / ((double)c[1] * 10.0));
      }
    }
  }
  for (int grid : {1, 256, 512, 768, 1024}) {
    for (int pack = 0; pack < 2; pack++) {
      for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(a));
        if (pack) hipLaunchKernelGGL(k2<1>, dim3(grid), dim3(64), 32u << LOG, 0, dtab, dsym, n, dout, dcyc, LOG);
        else hipLaunchKernelGGL(k2<0>, dim3(grid), dim3(64), 32u << LOG, 0, dtab, dsym, n, dout, dcyc, LOG);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        unsigned long long c[2]; CK(hipMemcpy(c, dcyc, 16, hipMemcpyDeviceToHost));
        if (rep == 1)
          printf("T2 grid %4d pack %d: %.2f ms  %.1f ns/pair-step (wall)  %.1f shader-cycles/pair-step  in-loop %.1f ns\n",
                 grid, pack, ms, ms * 1e6 / (n / 2), (double)c[0] / (n / 2), (double)c[1] * 10.0 / (n / 2));
      }
    }
  }
  return 0;
}
