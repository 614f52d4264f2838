// Part of encode.hip (included there, inside its anonymous namespace): K6a: gather of (nb, bits) into encode order by slot (quality stream).

// ------------------------------------------------------------------ K6: bit offsets and packing
// Gathers every symbol's packed (nb, bits) back into encode order ONCE: enc16[e] is written
// coalesced (it reuses the key buffer, dead after K3) so that the packing pass is a linear read.
__global__ void __launch_bounds__(PACK_THREADS)
k_bitcount(const uint32_t *__restrict__ slot_of, const uint16_t *__restrict__ out16, unsigned n_sym,
           uint32_t *__restrict__ tile_bits, uint16_t *__restrict__ enc16) {
  __shared__ unsigned wsum[PACK_THREADS / 64];
  const unsigned ptile = fq_xcd_tile(blockIdx.x, gridDim.x);
  const unsigned e0 = ptile * PACK_TILE + threadIdx.x * PACK_PER_THREAD;
  unsigned bits = 0;
  unsigned v[PACK_PER_THREAD];
  // slot_of / enc16 are padded past n_sym: whole 16-symbol groups can be moved unconditionally
  const uint4 *sl4 = reinterpret_cast<const uint4 *>(slot_of + e0);
  unsigned sl[PACK_PER_THREAD];
#pragma unroll
  for (unsigned i = 0; i < PACK_PER_THREAD / 4; i++) {
    const uint4 t = e0 < n_sym ? sl4[i] : make_uint4(0, 0, 0, 0);
    sl[4 * i] = t.x; sl[4 * i + 1] = t.y; sl[4 * i + 2] = t.z; sl[4 * i + 3] = t.w;
  }
#pragma unroll
  for (unsigned i = 0; i < PACK_PER_THREAD; i++) {
    v[i] = e0 + i < n_sym ? (unsigned)out16[sl[i]] : 0u;
    bits += v[i] >> 12;
  }
  if (e0 < n_sym) {
    uint4 *o4 = reinterpret_cast<uint4 *>(enc16 + e0);
    o4[0] = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
    o4[1] = make_uint4(v[8] | (v[9] << 16), v[10] | (v[11] << 16), v[12] | (v[13] << 16), v[14] | (v[15] << 16));
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) bits += __shfl_xor(bits, d);
  if (fq_lane() == 0) wsum[threadIdx.x >> 6] = bits;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned tot = 0;
    for (unsigned i = 0; i < PACK_THREADS / 64; i++) tot += wsum[i];
    tile_bits[ptile] = tot;
  }
}
