// Part of encode.hip (included there, inside its anonymous namespace): K3: stable partition by context (generic kernel: quality stream, ballot fallback).

// ------------------------------------------------------------------ K3: stable partition by context
// One wave per tile walks its symbols in encode order, 64 at a time; lanes with equal
// context are ranked by lane order (ballot match), the group leader advances the context's
// cursor in LDS.  Stability is what makes every context's run = its chain.
// The ranking loop is a latency chain through LDS (cursor read -> leader write), so it must not
// contain global memory operations: vmcnt retires in order and hipcc drains it at the loop
// back-edge, which put one full HBM round trip into every 64-symbol iteration (measured
// 1.4-2 us).  Keys therefore arrive in batches of SC_BATCH through LDS (one bulk load, many
// 16-byte requests in flight), the slots of a batch are collected in LDS, and the stores of
// the batch (coalesced slot_of, scattered sorted_sym) are issued back to back afterwards.
// (Staging the tile's partition in LDS to write whole runs was measured SLOWER: the 32 KB buffer
// costs two thirds of the occupancy and the loop is VALU-bound on the ballot match, not on stores.)
constexpr unsigned SC_BATCH = 4096;      // quality: 36 KB of LDS per wave
constexpr unsigned SC_BATCH_SEQ = 8192;  // sequence: 32 bytes per context and batch


// ORDERED: the rank comes from one LDS atomic per lane instead of the ballot match.  Same-address
// LDS atomics of ONE wave instruction take effect in lane order on gfx950 -- measured
// (tools/lds_atomic_order.hip: 0 mismatches in 7.9e9 lane-ops, packed and plain counters), not
// documented, so every handle re-verifies it at creation (fq_probe_lds_atomic_order) and falls
// back to the ballot kernel otherwise.  REL packs two 16-bit cursors per word: a cursor reaches
// 65536 only with the last symbol of a tile that holds nothing but that context.
template <class M, bool ORDERED>
__global__ void __launch_bounds__(64)
k_scatter(const uint16_t *__restrict__ ckey, const uint8_t *__restrict__ csym, unsigned n_sym, unsigned T,
          const uint32_t *__restrict__ tile_base, uint8_t *__restrict__ sorted_sym,
          uint32_t *__restrict__ slot_of, int dbg_no_sym) {
  constexpr unsigned B = M::B;
  constexpr bool QUAL = M::STREAM == 1;
  constexpr unsigned BATCH = QUAL ? SC_BATCH : SC_BATCH_SEQ;
  // 16-bit cursors = rank inside the tile (a tile has at most 65536 symbols), two per word; the
  // tile's base is added from the tile_base row (sequence: LDS copy; quality: L2-resident row)
  __shared__ uint32_t cursor32[B / 2];
  __shared__ uint32_t base[QUAL ? 1 : B];
  __shared__ uint4 kbatch4[BATCH / 8], rbatch4[BATCH / 8], sbatch4[QUAL ? BATCH / 16 : 1];
  uint16_t *kbatch = reinterpret_cast<uint16_t *>(kbatch4), *rbatch = reinterpret_cast<uint16_t *>(rbatch4);
  uint16_t *cursor = reinterpret_cast<uint16_t *>(cursor32);
  const uint8_t *sbatch = reinterpret_cast<const uint8_t *>(sbatch4);
  const unsigned tile = fq_xcd_tile(blockIdx.x, gridDim.x), lane = threadIdx.x;
  const uint32_t *tb_row = tile_base + (size_t)tile * B;
  const unsigned e0 = tile * T;
  const unsigned e1 = min(e0 + T, n_sym);
  for (unsigned c = lane; c < B / 2; c += 64) cursor32[c] = 0;
  if (!QUAL) for (unsigned c = lane; c < B; c += 64) base[c] = tb_row[c];
  fq_lds_wave_sync();
  for (unsigned b0 = e0; b0 < e1; b0 += BATCH) {
    const unsigned nb = min(BATCH, e1 - b0);
    // bulk load of the batch's keys (b0 is a multiple of 16 symbols; the arrays are padded)
    const uint4 *gk = reinterpret_cast<const uint4 *>(ckey + b0);
#pragma unroll
    for (unsigned i = 0; i < BATCH / 8 / 64; i++) kbatch4[i * 64 + lane] = gk[i * 64 + lane];
    if (QUAL) {
      const uint4 *gs = reinterpret_cast<const uint4 *>(csym + b0);
#pragma unroll
      for (unsigned i = 0; i < BATCH / 16 / 64; i++) sbatch4[i * 64 + lane] = gs[i * 64 + lane];
    }
    fq_lds_wave_sync();
    if (ORDERED) {
      for (unsigned cb = 0; cb < nb; cb += 64) {  // no global memory operation in here
        const unsigned i = cb + lane;
        if (i < nb) {
          const unsigned ctx = QUAL ? (unsigned)kbatch[i] : (unsigned)kbatch[i] & 0xFFu;
          rbatch[i] = (uint16_t)(atomicAdd(&cursor32[ctx >> 1], 1u << (16 * (ctx & 1u))) >> (16 * (ctx & 1u)));
        }
      }
      fq_lds_wave_sync();
    } else {
      for (unsigned cb = 0; cb < nb; cb += 64) {  // no global memory operation in here
        const unsigned i = cb + lane;
        const bool valid = i < nb;
        const unsigned ctx = QUAL ? (unsigned)kbatch[i] : (unsigned)kbatch[i] & 0xFFu;
        const unsigned long long grp = fq_match_any<M::KEYBITS>(ctx, valid);
        const unsigned rank = fq_mbcnt(grp);
        const unsigned cur = cursor[ctx];
        fq_lds_wave_sync();  // every lane has read its cursor before any leader advances it
        if (valid) {
          if (rank == 0) cursor[ctx] = (uint16_t)(cur + (unsigned)__popcll(grp));
          rbatch[i] = (uint16_t)(cur + rank);
        }
        fq_lds_wave_sync();
      }
    }
    // the batch's stores, back to back: slots coalesced, symbols scattered
    if (nb == BATCH) {
      // all gathers of the tile_base row first (one wait), then the stores: a load between
      // two stores would wait for the older store (vmcnt retires in order)
      unsigned slots[BATCH / 64];
#pragma unroll
      for (unsigned j = 0; j < BATCH / 64; j++) {
        const unsigned key = kbatch[j * 64 + lane];
        slots[j] = (QUAL ? tb_row[key] : base[key & 0xFFu]) + rbatch[j * 64 + lane];
      }
#pragma unroll
      for (unsigned j = 0; j < BATCH / 64; j++) slot_of[b0 + j * 64 + lane] = slots[j];
      if (!dbg_no_sym) {
#pragma unroll
        for (unsigned j = 0; j < BATCH / 64; j++)
          sorted_sym[slots[j]] = QUAL ? sbatch[j * 64 + lane] : (uint8_t)(kbatch[j * 64 + lane] >> 8);
      }
    } else {
      for (unsigned i = lane; i < nb; i += 64) {
        const unsigned key = kbatch[i];
        const unsigned slot = (QUAL ? tb_row[key] : base[key & 0xFFu]) + rbatch[i];
        slot_of[b0 + i] = slot;
        sorted_sym[slot] = QUAL ? sbatch[i] : (uint8_t)(key >> 8);
      }
    }
    fq_lds_wave_sync();
  }
}
