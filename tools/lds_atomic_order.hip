// Probe: do same-address LDS atomics issued by ONE wave instruction take effect in lane order?
// (ds_add_rtn_u32 returning the old value = rank of the lane among the lanes with the same key.)
// Compared against the ballot-based rank over many random key patterns, with many waves in flight.
// Build: hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_order.hip -o tools/_build/lds_atomic_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ unsigned rng(unsigned &s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

template <int WAVES, bool PACKED>
__global__ void __launch_bounds__(WAVES * 64) probe(unsigned iters, unsigned nkeys, unsigned long long *bad, unsigned seed) {
  __shared__ unsigned cursor[WAVES][PACKED ? 4096 : 8192];  // PACKED: two 16-bit counters per word
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned s = seed * 2654435761u + (blockIdx.x * blockDim.x + threadIdx.x) * 40503u + 1;
  unsigned long long nbad = 0;
  for (unsigned it = 0; it < iters; it++) {
    for (unsigned c = lane; c < (PACKED ? (nkeys + 1) / 2 : nkeys); c += 64) cursor[wave][c] = 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // skewed keys: small ranges make many equal lanes
    const unsigned mode = it & 7;
    unsigned range = mode == 0 ? 1 : mode == 1 ? 2 : mode == 2 ? 4 : mode == 3 ? 16 : mode == 4 ? 64 : mode == 5 ? 97 : nkeys;
    if (range > nkeys) range = nkeys;
    unsigned key = rng(s) % range;
    if (mode == 6) key = (key * 64u) % nkeys;  // same bank, different addresses
    const bool active = (rng(s) & 15) != 0;   // some lanes sit out
    unsigned got = 0;
    if (active) {
      if (PACKED) got = (atomicAdd(&cursor[wave][key >> 1], 1u << (16 * (key & 1))) >> (16 * (key & 1))) & 0xFFFFu;
      else got = atomicAdd(&cursor[wave][key], 1u);
    }
    // reference rank: active lanes below me with the same key
    unsigned ref = 0;
    for (int l = 0; l < 64; l++) {
      const unsigned k2 = __shfl(key, l);
      const int a2 = __shfl((int)active, l);
      if ((unsigned)l < lane && a2 && k2 == key) ref++;
    }
    if (active && got != ref) nbad++;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (nbad) atomicAdd(bad, nbad);
}

int main() {
  unsigned long long *dbad; CK(hipMalloc(&dbad, 8)); CK(hipMemset(dbad, 0, 8));
  for (int rep = 0; rep < 4; rep++) {
    hipLaunchKernelGGL((probe<2, false>), dim3(2048), dim3(128), 0, 0, 2000u, 8192u, dbad, (unsigned)rep);
    hipLaunchKernelGGL((probe<4, true>), dim3(2048), dim3(256), 0, 0, 2000u, 8192u, dbad, (unsigned)rep + 100);
    hipLaunchKernelGGL((probe<4, true>), dim3(2048), dim3(256), 0, 0, 2000u, 256u, dbad, (unsigned)rep + 200);
    CK(hipDeviceSynchronize());
    unsigned long long h; CK(hipMemcpy(&h, dbad, 8, hipMemcpyDeviceToHost));
    printf("rep %d: mismatches so far %llu (of %llu lane-ops)\n", rep, h, (unsigned long long)(rep + 1) * 3ull * 2048 * 2000 * 160);
  }
  return 0;
}
