// Exclusive prefix sums over device arrays (record lengths -> first encode index,
// N counts -> n_pos offsets, per-tile bit counts -> bit offsets).
// Three launches: per-chunk sums, one workgroup scanning the chunk sums, per-chunk
// scan + base.  Chunk = 256 threads x 8 items, coalesced in both passes.
#include "fqgpu_internal.h"

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_CHUNK = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ unsigned long long wave_incl_scan(unsigned long long v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long o = __shfl_up(v, d);
    if (fq_lane() >= (unsigned)d) v += o;
  }
  return v;
}

// exclusive scan of one value per thread over the 256-thread block; returns the
// exclusive prefix, *total receives the block sum
__device__ unsigned long long block_excl_scan(unsigned long long v, unsigned long long *total) {
  __shared__ unsigned long long wsum[SCAN_THREADS / 64];
  const unsigned long long inc = wave_incl_scan(v);
  const unsigned w = threadIdx.x >> 6;
  if (fq_lane() == 63) wsum[w] = inc;
  __syncthreads();
  unsigned long long base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < SCAN_THREADS / 64; i++) {
    if ((unsigned)i < w) base += wsum[i];
    tot += wsum[i];
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ void __launch_bounds__(SCAN_THREADS)
k_chunk_sums(const uint32_t *__restrict__ in, size_t n, unsigned long long *__restrict__ sums) {
  const size_t base = (size_t)blockIdx.x * SCAN_CHUNK;
  unsigned long long v = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    const size_t idx = base + (size_t)i * SCAN_THREADS + threadIdx.x;
    if (idx < n) v += in[idx];
  }
  unsigned long long tot;
  (void)block_excl_scan(v, &tot);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// single workgroup: exclusive scan of the chunk sums in place
__global__ void __launch_bounds__(SCAN_THREADS)
k_scan_sums(unsigned long long *__restrict__ sums, size_t n_chunks) {
  unsigned long long carry = 0;
  for (size_t base = 0; base < n_chunks; base += SCAN_THREADS) {
    const size_t idx = base + threadIdx.x;
    const unsigned long long v = idx < n_chunks ? sums[idx] : 0ull;
    unsigned long long tot;
    const unsigned long long ex = block_excl_scan(v, &tot);
    if (idx < n_chunks) sums[idx] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) sums[n_chunks] = carry;
}

template <class OutT>
__global__ void __launch_bounds__(SCAN_THREADS)
k_chunk_scan(const uint32_t *__restrict__ in, size_t n, const unsigned long long *__restrict__ sums,
             size_t n_chunks, OutT *__restrict__ out) {
  const size_t base = (size_t)blockIdx.x * SCAN_CHUNK + (size_t)threadIdx.x * SCAN_ITEMS;
  uint32_t x[SCAN_ITEMS];
  unsigned long long v = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    x[i] = (base + i < n) ? in[base + i] : 0u;
    v += x[i];
  }
  unsigned long long tot;
  unsigned long long run = sums[blockIdx.x] + block_excl_scan(v, &tot);
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    if (base + i < n) out[base + i] = (OutT)run;
    run += x[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = (OutT)sums[n_chunks];
}

// two arrays of the same length scanned by the same three launches (record lengths and N counts)
__global__ void __launch_bounds__(SCAN_THREADS)
k_chunk_sums2(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, size_t n,
              unsigned long long *__restrict__ sums_a, unsigned long long *__restrict__ sums_b) {
  const size_t base = (size_t)blockIdx.x * SCAN_CHUNK;
  unsigned long long va = 0, vb = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    const size_t idx = base + (size_t)i * SCAN_THREADS + threadIdx.x;
    if (idx < n) { va += a[idx]; vb += b[idx]; }
  }
  unsigned long long ta, tb;
  (void)block_excl_scan(va, &ta);
  (void)block_excl_scan(vb, &tb);
  if (threadIdx.x == 0) { sums_a[blockIdx.x] = ta; sums_b[blockIdx.x] = tb; }
}

__global__ void __launch_bounds__(SCAN_THREADS)
k_scan_sums2(unsigned long long *__restrict__ sums_a, unsigned long long *__restrict__ sums_b, size_t n_chunks) {
  unsigned long long ca = 0, cb = 0;
  for (size_t base = 0; base < n_chunks; base += SCAN_THREADS) {
    const size_t idx = base + threadIdx.x;
    const unsigned long long va = idx < n_chunks ? sums_a[idx] : 0ull, vb = idx < n_chunks ? sums_b[idx] : 0ull;
    unsigned long long ta, tb;
    const unsigned long long ea = block_excl_scan(va, &ta);
    const unsigned long long eb = block_excl_scan(vb, &tb);
    if (idx < n_chunks) { sums_a[idx] = ca + ea; sums_b[idx] = cb + eb; }
    ca += ta; cb += tb;
  }
  if (threadIdx.x == 0) { sums_a[n_chunks] = ca; sums_b[n_chunks] = cb; }
}

__global__ void __launch_bounds__(SCAN_THREADS)
k_chunk_scan2(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, size_t n,
              const unsigned long long *__restrict__ sums_a, const unsigned long long *__restrict__ sums_b,
              size_t n_chunks, uint32_t *__restrict__ out_a, uint32_t *__restrict__ out_b) {
  const size_t base = (size_t)blockIdx.x * SCAN_CHUNK + (size_t)threadIdx.x * SCAN_ITEMS;
  uint32_t xa[SCAN_ITEMS], xb[SCAN_ITEMS];
  unsigned long long va = 0, vb = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    xa[i] = (base + i < n) ? a[base + i] : 0u; va += xa[i];
    xb[i] = (base + i < n) ? b[base + i] : 0u; vb += xb[i];
  }
  unsigned long long ta, tb;
  unsigned long long ra = sums_a[blockIdx.x] + block_excl_scan(va, &ta);
  unsigned long long rb = sums_b[blockIdx.x] + block_excl_scan(vb, &tb);
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    if (base + i < n) { out_a[base + i] = (uint32_t)ra; out_b[base + i] = (uint32_t)rb; }
    ra += xa[i]; rb += xb[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { out_a[n] = (uint32_t)sums_a[n_chunks]; out_b[n] = (uint32_t)sums_b[n_chunks]; }
}

template <class OutT>
int scan_impl(hipStream_t st, const uint32_t *in, size_t n, OutT *out, DevBuf &tmp) {
  const size_t n_chunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
  const size_t nc = n_chunks ? n_chunks : 1;
  int rc = tmp.reserve((nc + 1) * sizeof(unsigned long long));
  if (rc) return rc;
  unsigned long long *sums = tmp.as<unsigned long long>();
  hipLaunchKernelGGL(k_chunk_sums, dim3((unsigned)nc), dim3(SCAN_THREADS), 0, st, in, n, sums);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_THREADS), 0, st, sums, nc);
  hipLaunchKernelGGL(k_chunk_scan<OutT>, dim3((unsigned)nc), dim3(SCAN_THREADS), 0, st, in, n, sums, nc, out);
  FQ_HIP(hipGetLastError());
  return FQGPU_OK;
}

}  // namespace

int fq_scan2_u32_to_u32(hipStream_t st, const uint32_t *a, const uint32_t *b, size_t n, uint32_t *out_a,
                        uint32_t *out_b, DevBuf &tmp) {
  const size_t n_chunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
  const size_t nc = n_chunks ? n_chunks : 1;
  int rc = tmp.reserve(2 * (nc + 1) * sizeof(unsigned long long));
  if (rc) return rc;
  unsigned long long *sa = tmp.as<unsigned long long>(), *sb = sa + nc + 1;
  hipLaunchKernelGGL(k_chunk_sums2, dim3((unsigned)nc), dim3(SCAN_THREADS), 0, st, a, b, n, sa, sb);
  hipLaunchKernelGGL(k_scan_sums2, dim3(1), dim3(SCAN_THREADS), 0, st, sa, sb, nc);
  hipLaunchKernelGGL(k_chunk_scan2, dim3((unsigned)nc), dim3(SCAN_THREADS), 0, st, a, b, n, sa, sb, nc, out_a, out_b);
  FQ_HIP(hipGetLastError());
  return FQGPU_OK;
}

int fq_scan_u32_to_u32(hipStream_t st, const uint32_t *in, size_t n, uint32_t *out, DevBuf &tmp) {
  return scan_impl<uint32_t>(st, in, n, out, tmp);
}
int fq_scan_u32_to_u64(hipStream_t st, const uint32_t *in, size_t n, unsigned long long *out, DevBuf &tmp) {
  return scan_impl<unsigned long long>(st, in, n, out, tmp);
}
