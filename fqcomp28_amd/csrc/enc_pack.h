// Part of encode.hip (included there, inside its anonymous namespace): K6 scan, K6b packing, K7 epilogue.

// Bit offsets of the packing tiles, size/overflow verdict and zeroing of the words shared by two
// tiles, in ONE single-workgroup kernel: the per-tile counts are few (M / 4096) and every extra
// launch on a block's critical path costs its scheduling latency on a busy GPU (measured ~0.8 ms
// per tiny kernel when four blocks are in flight).
// Verdict = BIT_closeCStream: 0 when the write pointer reached dst+cap-8 (zstd bitstream.h).
__global__ void __launch_bounds__(1024)
k_bitscan(const uint32_t *__restrict__ tile_bits, unsigned n_ptiles, unsigned long long *__restrict__ tile_bit_base,
          const uint32_t *__restrict__ log_prefix, unsigned B, unsigned long long cap, uint32_t *__restrict__ out,
          StreamResult *res) {
  __shared__ unsigned long long wsum[16];
  __shared__ unsigned long long s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  const unsigned wave = threadIdx.x >> 6, lane = fq_lane();
  for (unsigned base = 0; base < n_ptiles; base += 1024) {
    const unsigned i = base + threadIdx.x;
    const unsigned long long v = i < n_ptiles ? tile_bits[i] : 0ull;
    unsigned long long inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned long long o = __shfl_up(inc, d);
      if (lane >= (unsigned)d) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned long long off = s_carry;
    for (unsigned w = 0; w < wave; w++) off += wsum[w];
    if (i < n_ptiles) tile_bit_base[i] = off + inc - v;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = off + inc;
    __syncthreads();
  }
  const unsigned long long payload = s_carry;
  const unsigned long long all = payload + log_prefix[B] + 1ull;  // + state flush + end mark
  const bool overflow = cap <= 8ull || (all >> 3) >= cap - 8ull;
  if (threadIdx.x == 0) {
    tile_bit_base[n_ptiles] = payload;
    res->total_bits = payload;
    res->len = (all + 7ull) >> 3;
    res->overflow = overflow ? 1u : 0u;
  }
  if (overflow) return;
  __syncthreads();  // tile_bit_base of this workgroup's own writes
  // words shared by two packing tiles are OR-ed into, so they start from zero
  for (unsigned t = threadIdx.x; t <= n_ptiles; t += 1024) {
    const unsigned long long b = t < n_ptiles ? tile_bit_base[t] : payload;
    out[b >> 5] = 0u;
  }
}

__global__ void __launch_bounds__(PACK_THREADS)
k_pack(const uint16_t *__restrict__ enc16, unsigned n_sym,
       const unsigned long long *__restrict__ tile_bit_base, uint32_t *__restrict__ out,
       const StreamResult *res) {
  __shared__ uint32_t words[PACK_TILE * 12 / 32 + 4];
  __shared__ unsigned wsum[PACK_THREADS / 64];
  if (res->overflow) return;
  constexpr unsigned NW = PACK_TILE * 12 / 32 + 4;
  for (unsigned i = threadIdx.x; i < NW; i += PACK_THREADS) words[i] = 0;
  const unsigned ptile = fq_xcd_tile(blockIdx.x, gridDim.x);
  const unsigned long long b0 = tile_bit_base[ptile], b1 = tile_bit_base[ptile + 1];
  const unsigned e0 = ptile * PACK_TILE + threadIdx.x * PACK_PER_THREAD;
  unsigned v[PACK_PER_THREAD];
  unsigned bits = 0;
  {
    const uint4 *i4 = reinterpret_cast<const uint4 *>(enc16 + e0);
    const uint4 a = e0 < n_sym ? i4[0] : make_uint4(0, 0, 0, 0), b = e0 < n_sym ? i4[1] : make_uint4(0, 0, 0, 0);
    const unsigned w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (unsigned i = 0; i < PACK_PER_THREAD; i++) {
      v[i] = e0 + i < n_sym ? (w[i >> 1] >> (16 * (i & 1))) & 0xFFFFu : 0u;
      bits += v[i] >> 12;
    }
  }
  // exclusive scan of the per-thread bit counts over the workgroup
  unsigned inc = bits;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned o = __shfl_up(inc, d);
    if (fq_lane() >= (unsigned)d) inc += o;
  }
  if (fq_lane() == 63) wsum[threadIdx.x >> 6] = inc;
  __syncthreads();
  unsigned off = inc - bits;
  for (unsigned w = 0; w < (threadIdx.x >> 6); w++) off += wsum[w];
  // bit position relative to the first 32-bit word this tile touches
  off += (unsigned)(b0 & 31ull);
  unsigned long long acc = 0;
  unsigned nacc = off & 31u, w = off >> 5;
#pragma unroll
  for (unsigned i = 0; i < PACK_PER_THREAD; i++) {
    const unsigned nb = v[i] >> 12;
    acc |= (unsigned long long)(v[i] & 0xFFFu) << nacc;
    nacc += nb;
    if (nacc >= 32) {
      atomicOr(&words[w], (uint32_t)acc);
      acc >>= 32; nacc -= 32; w++;
    }
  }
  if (nacc) atomicOr(&words[w], (uint32_t)acc);
  __syncthreads();
  if (b1 == b0) return;
  const unsigned long long gw0 = b0 >> 5;
  const unsigned nw = (unsigned)(((b1 + 31ull) >> 5) - gw0);
  const bool tail_shared = (b1 & 31ull) != 0;
  for (unsigned i = threadIdx.x; i < nw; i += PACK_THREADS) {
    if (i == 0 || (tail_shared && i == nw - 1)) atomicOr(&out[gw0 + i], words[i]);
    else out[gw0 + i] = words[i];
  }
}

// ------------------------------------------------------------------ K7: state flush + end mark
// FSE_Encoder::endChunk (src/fse_common.hpp:86-90): states of context 0..B-1, log bits each,
// then one '1' bit.  A context never used in the block still holds its initial state 2^log.
template <class M>
__global__ void __launch_bounds__(256)
k_epilogue(const uint32_t *__restrict__ arrays, const uint16_t *__restrict__ final_state,
           const uint32_t *__restrict__ logs, const uint32_t *__restrict__ log_prefix,
           uint32_t *__restrict__ out, const StreamResult *res, const uint4 *__restrict__ edges, unsigned n_tiles) {
  constexpr unsigned B = M::B;
  constexpr unsigned NW = (B * 12 + 1 + 31) / 32 + 2;
  __shared__ uint32_t words[NW];
  if (res->overflow) return;
  const unsigned long long p0 = res->total_bits;
  const unsigned sh = (unsigned)(p0 & 31ull);
  // Tile-sorted path (edges != nullptr): the stream was never zeroed.  The words two tiles share -- or the last tile and
  // the state flush -- were left out by k_tile_gather_pack; every tile's share of them is in edges[].  Zero each of them
  // once, then OR the shares in (one workgroup: the two passes are separated by its barrier).
  if (edges != nullptr) {
    for (unsigned t = threadIdx.x; t < n_tiles; t += blockDim.x) {
      const uint4 e = edges[t];
      if (e.x != 0xFFFFFFFFu) out[e.x] = 0u;
      if (e.z != 0xFFFFFFFFu) out[e.z] = 0u;
    }
    if (threadIdx.x == 0 && sh == 0u) out[p0 >> 5] = 0u;  // (the flush starts a word of its own: nobody has listed it)
    __threadfence();
    __syncthreads();
    for (unsigned t = threadIdx.x; t < n_tiles; t += blockDim.x) {
      const uint4 e = edges[t];
      if (e.x != 0xFFFFFFFFu && e.y) atomicOr(&out[e.x], e.y);
      if (e.z != 0xFFFFFFFFu && e.w) atomicOr(&out[e.z], e.w);
    }
  }
  const uint32_t *ctx_count = arrays;
  for (unsigned i = threadIdx.x; i < NW; i += blockDim.x) words[i] = 0;
  __syncthreads();
  for (unsigned c = threadIdx.x; c <= B; c += blockDim.x) {
    unsigned val, nb;
    if (c < B) {
      const unsigned n = ctx_count[c];
      nb = logs[c];
      val = n ? ((unsigned)final_state[c] & ((1u << nb) - 1u)) : 0u;
    } else { val = 1u; nb = 1u; }  // end mark
    const unsigned off = sh + log_prefix[c];  // log_prefix[B] = sum of logs
    const unsigned long long field = (unsigned long long)val << (off & 31u);
    atomicOr(&words[off >> 5], (uint32_t)field);
    if ((off & 31u) + nb > 32u) atomicOr(&words[(off >> 5) + 1], (uint32_t)(field >> 32));
  }
  __syncthreads();
  const unsigned long long gw0 = p0 >> 5;
  const unsigned nw = (sh + log_prefix[B] + 1u + 31u) >> 5;
  for (unsigned i = threadIdx.x; i < nw; i += blockDim.x) {
    if (i == 0) atomicOr(&out[gw0], words[0]);
    else out[gw0 + i] = words[i];
  }
}
