// Part of encode.hip (included there, inside its anonymous namespace): sequence stream: batch-sorted partition (K3) and run-based gather (K6a).

// ---- sequence stream: batch-sorted partition ---------------------------------------------
// Random byte stores and random 2-byte gathers run at ~90 G accesses/s on the whole chip
// whatever their size (tools/kernel_boundary_ubench.hip) -- one L2 transaction each -- and the
// partition (symbol scatter) and the gather into encode order did 2 x 120 M of them per block and
// stream.  With 256 contexts a batch of a few thousand symbols holds a RUN of every context, so
// the sequence stream sorts every batch by context in LDS:
//   K3  k_scatter_seq   writes the batch's symbols as 256 short contiguous runs (consecutive lanes =
//                       consecutive bytes: a handful of transactions per wave store), the position
//                       of every symbol inside its batch-sorted order (lpos16, coalesced) and the
//                       batch descriptor (global start slot and batch-local offset of every run)
//   K6a k_bitcount_seq  reads the batch's 256 runs of (nb, bits) with consecutive lanes on
//                       consecutive slots into LDS and picks every symbol's value there by lpos16
// slot_of (4 bytes per symbol) is not needed for this stream.
constexpr unsigned SEQ_BATCH = PACK_TILE;  // one partition batch = one packing tile (two were measured: no gain)

struct SeqBatchDesc {
  uint32_t *start;  // [batches][256] global slot of the first symbol of context c in this batch
  uint16_t *pre;    // [batches][256] symbols of contexts < c in this batch
};

template <bool ORDERED>
__global__ void __launch_bounds__(64)
k_scatter_seq(const uint16_t *__restrict__ ckey, unsigned n_sym, unsigned T,
              const uint32_t *__restrict__ tile_base, uint8_t *__restrict__ sorted_sym,
              uint16_t *__restrict__ lpos16, SeqBatchDesc bd, int dbg_no_sym) {
  constexpr unsigned B = SeqModel::B, BATCH = SEQ_BATCH;
  __shared__ uint32_t cursor32[B / 2];  // 16-bit ranks inside the tile, two per word
  __shared__ uint32_t base[B];
  __shared__ uint16_t cb[B], pre[B];
  __shared__ uint4 kbatch4[BATCH / 8], rbatch4[BATCH / 8];
  __shared__ uint8_t ssym[BATCH], sctx[BATCH];
  uint16_t *kbatch = reinterpret_cast<uint16_t *>(kbatch4), *rbatch = reinterpret_cast<uint16_t *>(rbatch4);
  uint16_t *cursor = reinterpret_cast<uint16_t *>(cursor32);
  const unsigned tile = fq_xcd_tile(blockIdx.x, gridDim.x), lane = threadIdx.x;
  const uint32_t *tb_row = tile_base + (size_t)tile * B;
  const unsigned e0 = tile * T;
  const unsigned e1 = min(e0 + T, n_sym);
  for (unsigned c = lane; c < B / 2; c += 64) cursor32[c] = 0;
  for (unsigned c = lane; c < B; c += 64) base[c] = tb_row[c];
  fq_lds_wave_sync();
  for (unsigned b0 = e0; b0 < e1; b0 += BATCH) {
    const unsigned nb = min(BATCH, e1 - b0), gb = b0 / BATCH;
    const uint4 *gk = reinterpret_cast<const uint4 *>(ckey + b0);
#pragma unroll
    for (unsigned i = 0; i < BATCH / 8 / 64; i++) kbatch4[i * 64 + lane] = gk[i * 64 + lane];
    // ranks of the contexts at the start of the batch (lane l owns contexts 4 l .. 4 l + 3)
    const uint2 snap = reinterpret_cast<const uint2 *>(cursor32)[lane];
    reinterpret_cast<uint2 *>(cb)[lane] = snap;
    fq_lds_wave_sync();
    if (ORDERED) {
      for (unsigned cbk = 0; cbk < nb; cbk += 64) {  // no global memory operation in here
        const unsigned i = cbk + lane;
        if (i < nb) {
          const unsigned ctx = (unsigned)kbatch[i] & 0xFFu;
          rbatch[i] = (uint16_t)(atomicAdd(&cursor32[ctx >> 1], 1u << (16 * (ctx & 1u))) >> (16 * (ctx & 1u)));
        }
      }
      fq_lds_wave_sync();
    } else {
      for (unsigned cbk = 0; cbk < nb; cbk += 64) {
        const unsigned i = cbk + lane;
        const bool valid = i < nb;
        const unsigned ctx = (unsigned)kbatch[i] & 0xFFu;
        const unsigned long long grp = fq_match_any<SeqModel::KEYBITS>(ctx, valid);
        const unsigned rank = fq_mbcnt(grp);
        const unsigned cur = cursor[ctx];
        fq_lds_wave_sync();
        if (valid) {
          if (rank == 0) cursor[ctx] = (uint16_t)(cur + (unsigned)__popcll(grp));
          rbatch[i] = (uint16_t)(cur + rank);
        }
        fq_lds_wave_sync();
      }
    }
    // run lengths of this batch -> offsets of the runs inside the batch-sorted order
    {
      const uint2 now = reinterpret_cast<const uint2 *>(cursor32)[lane];
      const unsigned n0 = (now.x & 0xFFFFu) - (snap.x & 0xFFFFu), n1 = (now.x >> 16) - (snap.x >> 16),
                     n2 = (now.y & 0xFFFFu) - (snap.y & 0xFFFFu), n3 = (now.y >> 16) - (snap.y >> 16);
      unsigned incl = n0 + n1 + n2 + n3;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(incl, d);
        if (lane >= (unsigned)d) incl += o;
      }
      const unsigned p0 = incl - (n0 + n1 + n2 + n3), p1 = p0 + n0, p2 = p1 + n1, p3 = p2 + n2;
      reinterpret_cast<uint2 *>(pre)[lane] = make_uint2(p0 | (p1 << 16), p2 | (p3 << 16));
      // descriptor of the batch (read back by k_bitcount_seq)
      reinterpret_cast<uint2 *>(bd.pre + (size_t)gb * B)[lane] = make_uint2(p0 | (p1 << 16), p2 | (p3 << 16));
      reinterpret_cast<uint4 *>(bd.start + (size_t)gb * B)[lane] =
          make_uint4(base[4 * lane] + (snap.x & 0xFFFFu), base[4 * lane + 1] + (snap.x >> 16),
                     base[4 * lane + 2] + (snap.y & 0xFFFFu), base[4 * lane + 3] + (snap.y >> 16));
    }
    fq_lds_wave_sync();
    for (unsigned i = lane; i < nb; i += 64) {  // encode order -> batch-sorted order, in LDS
      const unsigned key = kbatch[i], c = key & 0xFFu;
      const unsigned lp = (unsigned)pre[c] + (unsigned)rbatch[i] - (unsigned)cb[c];
      ssym[lp] = (uint8_t)(key >> 8);
      sctx[lp] = (uint8_t)c;
      rbatch[i] = (uint16_t)lp;
    }
    fq_lds_wave_sync();
    // the batch's stores, back to back: positions coalesced, symbols as 256 contiguous runs
    if (nb == BATCH) {
      uint4 *gl = reinterpret_cast<uint4 *>(lpos16 + b0);
#pragma unroll
      for (unsigned i = 0; i < BATCH / 8 / 64; i++) gl[i * 64 + lane] = rbatch4[i * 64 + lane];
    } else {
      for (unsigned i = lane; i < nb; i += 64) lpos16[b0 + i] = rbatch[i];
    }
    if (!dbg_no_sym)
      for (unsigned p = lane; p < nb; p += 64) {
        const unsigned c = sctx[p];
        sorted_sym[base[c] + (unsigned)cb[c] + (p - (unsigned)pre[c])] = ssym[p];
      }
    fq_lds_wave_sync();
  }
}

// K6a for the sequence stream: one workgroup per batch (= packing tile)
__global__ void __launch_bounds__(PACK_THREADS)
k_bitcount_seq(const uint16_t *__restrict__ lpos16, SeqBatchDesc bd, const uint16_t *__restrict__ out16,
               unsigned n_sym, uint32_t *__restrict__ tile_bits, uint16_t *__restrict__ enc16) {
  constexpr unsigned B = SeqModel::B;
  __shared__ uint32_t start[B];
  __shared__ uint16_t pre[B + 2];
  __shared__ uint16_t vals[SEQ_BATCH];
  __shared__ unsigned wsum[PACK_THREADS / 64];
  const unsigned gb = fq_xcd_tile(blockIdx.x, gridDim.x);  // batch
  const unsigned b0 = gb * SEQ_BATCH, nb = min((unsigned)SEQ_BATCH, n_sym - b0);
  static_assert(PACK_THREADS == B, "one thread per context loads the batch descriptor");
  start[threadIdx.x] = bd.start[(size_t)gb * B + threadIdx.x];
  pre[threadIdx.x] = bd.pre[(size_t)gb * B + threadIdx.x];
  __syncthreads();
  // the batch's runs of (nb, bits): consecutive threads on consecutive slots of a run
  for (unsigned p = threadIdx.x; p < nb; p += PACK_THREADS) {
    unsigned lo = 0, hi = B - 1;  // last context c with pre[c] <= p (empty contexts share their successor's offset)
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const unsigned mid = (lo + hi + 1) >> 1;
      if ((unsigned)pre[mid] <= p) lo = mid; else hi = mid - 1;
    }
    vals[p] = out16[start[lo] + (p - (unsigned)pre[lo])];
  }
  __syncthreads();
  for (unsigned pt = 0; pt < SEQ_BATCH / PACK_TILE; pt++) {  // the packing tiles of the batch
    const unsigned t0 = b0 + pt * PACK_TILE;
    if (t0 >= n_sym) break;
    const unsigned e0 = t0 + threadIdx.x * PACK_PER_THREAD;
    unsigned bits = 0;
    unsigned v[PACK_PER_THREAD];
    {
      const uint4 *l4 = reinterpret_cast<const uint4 *>(lpos16 + e0);
#pragma unroll
      for (unsigned i = 0; i < PACK_PER_THREAD / 8; i++) {
        const uint4 t = e0 < n_sym ? l4[i] : make_uint4(0, 0, 0, 0);
        const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (unsigned j = 0; j < 8; j++) {
          const unsigned lp = (w[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
          v[8 * i + j] = e0 + 8 * i + j < n_sym ? (unsigned)vals[lp] : 0u;
          bits += v[8 * i + j] >> 12;
        }
      }
    }
    if (e0 < n_sym) {
      uint4 *o4 = reinterpret_cast<uint4 *>(enc16 + e0);
      o4[0] = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
      o4[1] = make_uint4(v[8] | (v[9] << 16), v[10] | (v[11] << 16), v[12] | (v[13] << 16), v[14] | (v[15] << 16));
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) bits += __shfl_xor(bits, d);
    __syncthreads();  // wsum of the previous tile has been read
    if (fq_lane() == 0) wsum[threadIdx.x >> 6] = bits;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned tot = 0;
      for (unsigned i = 0; i < PACK_THREADS / 64; i++) tot += wsum[i];
      tile_bits[t0 / PACK_TILE] = tot;
    }
  }
}
