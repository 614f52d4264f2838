// Part of encode.hip (included there, inside its anonymous namespace): tile-sorted partition (K3) and
// fused gather + bit packing (K6) -- the path taken when same-address LDS atomics are lane-ordered.
//
// Why: the partition's scattered byte stores and the un-permute's scattered 2-byte gathers cost one
// L2 transaction per symbol each and moved 4 + 4 bytes of slot numbers per
// symbol besides.  Here a tile of 32 K symbols is sorted by context INSIDE LDS (stable counting sort:
// the per-tile histogram K1 left gives every context's offset, one lane-ordered LDS atomic per
// symbol gives its rank), so that
//   * the symbols leave as one contiguous RUN per (tile, context) -- consecutive lanes store
//     consecutive bytes -- and only the position of every symbol inside its sorted tile (lpos16,
//     2 bytes, coalesced) is kept: no 4-byte slot_of;
//   * K6 reads the same runs of (nb, bits) back with consecutive lanes on consecutive slots into
//     LDS, picks every symbol's value there by lpos16, and packs the bits in the same kernel: the
//     (nb, bits) in encode order (enc16: 2 bytes written and read per symbol) never exist in
//     memory, and the bit offset of a tile comes from a decoupled look-back over the tiles' bit
//     totals instead of three more launches (bit counts, scan, pack).
// Stable = every context's run keeps encode order = its tANS state chain; the streams are
// bit-identical to the slot-based path (kept as the fallback when the probe of
// fq_probe_lds_atomic_order fails).
constexpr unsigned TS_TILE = 32768;    // symbols per tile (lpos16 and the 16-bit cursors hold 0 .. 32768)
constexpr unsigned TS_BATCH = 4096;    // symbols ranked between two workgroup barriers (K3: 72 KB of LDS, two workgroups per CU)
constexpr unsigned TS_THREADS = 512;   // K3
constexpr unsigned TS_WAVES = TS_THREADS / 64;
#ifndef FQ_K6_THREADS
#define FQ_K6_THREADS 512
#endif
constexpr unsigned TS_GP_THREADS = FQ_K6_THREADS;  // K6: 78 KB of LDS, two workgroups per CU = 16 waves (256 threads: 8 waves per CU, 2.6 % slower on the step)
constexpr unsigned TS_GP_PPT = 4096 / TS_GP_THREADS;  // symbols a thread packs per round: 16 or 8
constexpr unsigned TS_SUB = TS_GP_THREADS * TS_GP_PPT;  // symbols packed per round of K6 (4096)
static_assert(TS_GP_PPT == 16 || TS_GP_PPT == 8, "K6 packs 16 or 8 symbols per thread and round");
static_assert(TS_TILE % TS_BATCH == 0 && TS_TILE % TS_SUB == 0 && TS_SUB % PACK_TILE == 0, "tile geometry");

// phase timing of the two kernels (experiments build only): g_ts_prof[8 * kernel + phase] += wall clock ticks of workgroup thread 0
#ifdef FQGPU_EXPERIMENTS
#define TS_PROF_DECL unsigned long long ts_t_ = wall_clock64();
#define TS_PROF(slot) do { if (threadIdx.x == 0) { const unsigned long long n_ = wall_clock64(); atomicAdd(&g_ts_prof[slot], n_ - ts_t_); ts_t_ = n_; } } while (0)
#else
#define TS_PROF_DECL
#define TS_PROF(slot) do { } while (0)
#endif

// run r of a tile (runs are listed in context order = order of their local start):
//   x = global slot of its first symbol, y = local start | length << 16
template <class M> constexpr unsigned ts_run_stride() { return M::B < TS_TILE ? M::B : TS_TILE; }

// "which run holds local position p": bit p of bm is set where a run starts
struct TsRunMap {
  uint32_t bm[TS_TILE / 32];
  uint16_t wpre[TS_TILE / 32];  // run starts in the words before this one
};
__device__ __forceinline__ unsigned ts_run_of(const TsRunMap &m, unsigned p) {
  const unsigned w = p >> 5;
  return (unsigned)m.wpre[w] + __popc(m.bm[w] & (0xFFFFFFFFu >> (31u - (p & 31u)))) - 1u;
}

// exclusive scan of one value per thread over the workgroup of NT threads; *total = sum (all threads call)
template <unsigned NT>
__device__ __forceinline__ unsigned ts_block_scan(unsigned v, unsigned *wsum, unsigned *total) {
  unsigned inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned o = __shfl_up(inc, d);
    if (fq_lane() >= (unsigned)d) inc += o;
  }
  const unsigned w = threadIdx.x >> 6;
  __syncthreads();  // wsum of an earlier call has been read
  if (fq_lane() == 63) wsum[w] = inc;
  __syncthreads();
  unsigned base = 0, tot = 0;
#pragma unroll
  for (unsigned i = 0; i < NT / 64; i++) {
    const unsigned s = wsum[i];
    if (i < w) base += s;
    tot += s;
  }
  *total = tot;
  return base + inc - v;
}

// run-start bitmap -> wpre (all threads; bm complete and visible)
template <unsigned NT>
__device__ __forceinline__ void ts_build_wpre(TsRunMap &m, unsigned *wsum) {
  constexpr unsigned WPT = TS_TILE / 32 / NT;  // bitmap words per thread: 2 or 4
  unsigned c[WPT], sum = 0;
#pragma unroll
  for (unsigned k = 0; k < WPT; k++) { c[k] = __popc(m.bm[WPT * threadIdx.x + k]); sum += c[k]; }
  unsigned tot;
  unsigned ex = ts_block_scan<NT>(sum, wsum, &tot);
#pragma unroll
  for (unsigned k = 0; k < WPT; k++) { m.wpre[WPT * threadIdx.x + k] = (uint16_t)ex; ex += c[k]; }
  __syncthreads();
}
template <unsigned NT>
__device__ __forceinline__ void ts_clear_bitmap(TsRunMap &m) {
#pragma unroll
  for (unsigned k = 0; k < TS_TILE / 32 / NT; k++) m.bm[TS_TILE / 32 / NT * threadIdx.x + k] = 0;
}

// ------------------------------------------------------------------ K3: stable partition, one workgroup per tile
// Wave 0 ranks (key read, one lane-ordered atomic on the context's cursor, symbol written to its sorted
// place in LDS, position stored to lpos16 -- stores only, nothing in the loop ever waits for global
// memory); the other seven waves bring in the next batch of keys.  Then all waves write
// the tile's runs: position-major, so that consecutive lanes store consecutive bytes.
// DERIVED: the keys are the contexts alone, as the fused K1 (k_tile_hist2) leaves them -- two bytes per quality symbol, ONE per
// base --, and the symbol at encode index e is taken from the context at e - 1 (its low six bits / its top two: the position
// in front of p + 1 is p); the first symbol of every record in encode order has no such neighbour and is patched in from
// first_sym[record] once the tile is ranked.  Not DERIVED (keys of k_tile_hist<M>): u16 keys, sequence ctx | sym << 8,
// quality symbols in csym.
template <class M, bool DERIVED>
__global__ void __launch_bounds__(TS_THREADS)
k_tile_partition(const void *__restrict__ ckey_v, const uint8_t *__restrict__ csym, unsigned n_sym,
                 const uint16_t *__restrict__ tile_hist, const uint32_t *__restrict__ tile_base,
                 uint8_t *__restrict__ sorted_sym, uint16_t *__restrict__ lpos16, uint2 *__restrict__ runs,
                 uint32_t *__restrict__ run_count, unsigned long long *__restrict__ k6_status, unsigned *__restrict__ k6_counter,
                 const uint32_t *__restrict__ rec_start, unsigned R, const uint8_t *__restrict__ first_sym) {
  constexpr unsigned B = M::B;
  constexpr bool QUAL = M::STREAM == 1;
  constexpr bool K8 = DERIVED && !QUAL;  // one byte per key in memory
  static_assert(DERIVED || QUAL || true, "");
  const uint16_t *ckey = reinterpret_cast<const uint16_t *>(ckey_v);
  const uint8_t *ckey8 = reinterpret_cast<const uint8_t *>(ckey_v);
  // K6's look-back starts from clean words: this kernel runs in front of it on the same stream and has a workgroup per tile
  // (a memset per stream and block less)
  if (k6_status != nullptr && threadIdx.x == 0) {
    k6_status[blockIdx.x] = 0ull;
    if (blockIdx.x == 0) *k6_counter = 0u;
  }
  constexpr unsigned NCHUNK = B / 64;  // 64 contexts per chunk: 128 (quality) / 4 (sequence)
  __shared__ uint32_t cursor32[B / 2];  // 16-bit cursors (local positions), two per word
  __shared__ __attribute__((aligned(16))) uint8_t lsym[TS_TILE + 64];  // the tile's symbols in sorted order (+ a dump for idle lanes)
  // two batch buffers: batch j is ranked while the other seven waves (a) finish batch j - 1 -- the ranking wave left
  // every symbol's POSITION in the slot its key came from; they put the symbols to their places in lsym and store the
  // positions (lpos16: 16 bytes per thread) --, (b) write batch j + 1 into the buffer that has just become free,
  // from REGISTERS they filled one batch earlier, and (c) request batch j + 2 into those registers: the loads stay
  // in flight across the barrier (it waits for LDS only), so nobody ever sits at a barrier waiting for global
  // memory.  The ranking wave -- ONE wave has to do it: the rank comes from lane-ordered atomics in program order --
  // is left with three LDS operations per 64 symbols (key, atomic, position); round 3's also read the symbol,
  // placed it and stored the position to global memory: half of this kernel's time was that one wave's loop.
  // The run map lives in the same bytes: it is built when the last batch has been ranked.
  constexpr unsigned NBUF = 2;
  constexpr bool PLACE = true;   // the loaders place the symbols (a ranking wave that placed the sequence stream's itself -- no symbol buffers, three workgroups per CU -- measured the same)
  constexpr unsigned KB_BYTES = NBUF * (TS_BATCH / 8) * 16, SB_BYTES = PLACE ? NBUF * (TS_BATCH / 8) * 8 : 16;
  __shared__ __attribute__((aligned(16))) uint8_t stage_raw[(KB_BYTES + SB_BYTES) > sizeof(TsRunMap) ? (KB_BYTES + SB_BYTES) : sizeof(TsRunMap)];
  uint4 (*kb4)[TS_BATCH / 8] = reinterpret_cast<uint4 (*)[TS_BATCH / 8]>(stage_raw);
  uint2 (*sb8)[PLACE ? TS_BATCH / 8 : 1] = reinterpret_cast<uint2 (*)[PLACE ? TS_BATCH / 8 : 1]>(stage_raw + KB_BYTES);  // the symbols of key piece p: eight bytes
  TsRunMap &rm = *reinterpret_cast<TsRunMap *>(stage_raw);
  // behind the run map, once the batch buffers are dead: the first GD_CAP runs' offsets "global slot - local position"
  constexpr unsigned GD_CAP = (sizeof(stage_raw) - sizeof(TsRunMap)) / 4;
  uint32_t *gd = reinterpret_cast<uint32_t *>(stage_raw + sizeof(TsRunMap));
  __shared__ unsigned wsum[TS_WAVES], s_cnt[NCHUNK < 2 ? 2 : NCHUNK], s_nruns, s_max;
  __shared__ uint32_t dummy[64];  // where the lanes of the combining ranker that are no run heads send their (empty) atomics
  uint16_t *cur16 = reinterpret_cast<uint16_t *>(cursor32);
  const unsigned tile = fq_xcd_tile(blockIdx.x, gridDim.x), tid = threadIdx.x, wave = tid >> 6, lane = fq_lane();
  const unsigned e0 = tile * TS_TILE, nt = min(TS_TILE, n_sym - e0);
  const uint16_t *hrow = tile_hist + (size_t)tile * B;
  const uint32_t *tb_row = tile_base + (size_t)tile * B;
  TS_PROF_DECL
  constexpr unsigned PS = QUAL ? 0 : 8; (void)PS;

  // ---- local start of every context = exclusive scan of the tile's histogram row
  {
    uint32_t *h32 = reinterpret_cast<uint32_t *>(lsym);  // B * 4 <= TS_TILE bytes
    static_assert(B * 4 <= TS_TILE, "histogram row fits the symbol staging area");
    for (unsigned c = tid; c < B; c += TS_THREADS) h32[c] = hrow[c];
    if (tid == 0) s_max = 0;
    if (tid < 64) dummy[tid] = 0;
    __syncthreads();
    constexpr unsigned CPT = B >= TS_THREADS ? B / TS_THREADS : 1;  // contexts per thread
    const unsigned c0 = tid * CPT;
    unsigned sum = 0, mx = 0;
    if (c0 < B)
#pragma unroll
      for (unsigned k = 0; k < CPT; k++) { sum += h32[c0 + k]; mx = max(mx, h32[c0 + k]); }
    if (mx * 8u >= nt) atomicMax(&s_max, mx);  // (rare: only a context that holds an eighth of the tile reports)
    unsigned tot;
    unsigned run = ts_block_scan<TS_THREADS>(sum, wsum, &tot);
    if (c0 < B) {
#pragma unroll
      for (unsigned k = 0; k < CPT; k++) {
        const unsigned n = h32[c0 + k];
        cur16[c0 + k] = (uint16_t)run;  // (own contexts only: no two threads share a word when CPT is even; CPT = 1: 16-bit stores)
        run += n;
      }
    }
    __syncthreads();  // h32 is dead: lsym may be written
  }

  TS_PROF(PS + 0);
  // ---- ranking, batch by batch
  const unsigned nbatch = (nt + TS_BATCH - 1) / TS_BATCH;
  // the loading waves (448 threads): thread mt owns the 16-byte key pieces mt and mt + 448 (eight symbols each) of
  // every batch and the eight symbol bytes that go with each -- it loads them, deposits them, and later finishes
  // exactly those symbols, so no two threads ever touch one piece between two barriers
  static_assert(TS_BATCH / 8 == 512, "piece ownership below assumes a 4096-symbol batch");
  constexpr unsigned NL = TS_THREADS - 64;
  static_assert(2 * NL >= TS_BATCH / 8, "at most two pieces per loading thread");
  const unsigned mt = tid - 64;  // loader thread number (valid for tid >= 64)
  const bool two = mt + NL < TS_BATCH / 8;
  uint4 rk0 = make_uint4(0, 0, 0, 0), rk1 = rk0;
  uint2 rs0 = make_uint2(0, 0), rs1 = rs0;
  unsigned rp0 = 0, rp1 = 0;  // DERIVED: the dword in front of the piece (its top key is the one at the piece's first index - 1)
  auto syms_of = [](const uint4 k) {  // not DERIVED, sequence stream: the symbol is bits 9:8 of the key
    auto two_of = [](unsigned w) { return ((w >> 8) & 0xFFu) | ((w >> 24) << 8); };
    return make_uint2(two_of(k.x) | (two_of(k.y) << 16), two_of(k.z) | (two_of(k.w) << 16));
  };
  // DERIVED: the eight symbols of a piece from the piece's keys and the key in front of it
  auto derive = [](const uint4 k, unsigned prev) {
    if (K8) {  // k.x, k.y: eight one-byte contexts; symbol = the context's bits 7:6
      const unsigned t = (k.x >> 6) & 0x03030303u, u = (k.y >> 6) & 0x03030303u;
      return make_uint2((prev >> 30) | (t << 8), (t >> 24) | (u << 8));
    }
    // eight 16-bit contexts; symbol = the context's bits 5:0
    return make_uint2(((prev >> 16) & 63u) | ((k.x & 63u) << 8) | (((k.x >> 16) & 63u) << 16) | ((k.y & 63u) << 24),
                      ((k.y >> 16) & 63u) | ((k.z & 63u) << 8) | (((k.z >> 16) & 63u) << 16) | ((k.w & 63u) << 24));
  };
  auto keys16_of = [](const uint4 k) {  // the piece as eight 16-bit keys for the batch buffer
    if (K8) return make_uint4(__builtin_amdgcn_perm(0u, k.x, 0x0C010C00u), __builtin_amdgcn_perm(0u, k.x, 0x0C030C02u),
                              __builtin_amdgcn_perm(0u, k.y, 0x0C010C00u), __builtin_amdgcn_perm(0u, k.y, 0x0C030C02u));
    return k;
  };
  // piece `piece` of the batch that starts at encode index eb: its keys (K8: in .x, .y) and the dword in front of them
  auto load_piece = [&](unsigned eb, unsigned piece, uint4 &k, unsigned &prev, uint2 &sy) {
    if (K8) {
      const uint2 v = reinterpret_cast<const uint2 *>(ckey8 + eb)[piece];
      k = make_uint4(v.x, v.y, 0u, 0u);
      prev = eb + 8u * piece ? reinterpret_cast<const uint32_t *>(ckey8 + eb)[2 * (int)piece - 1] : 0u;
    } else {
      k = reinterpret_cast<const uint4 *>(ckey + eb)[piece];  // 16-byte aligned; arrays are padded by a batch
      if (DERIVED) prev = eb + 8u * piece ? reinterpret_cast<const uint32_t *>(ckey + eb)[4 * (int)piece - 1] : 0u;
      else if (QUAL) sy = reinterpret_cast<const uint2 *>(csym + eb)[piece];
    }
  };
  auto request = [&](unsigned j) {  // batch j -> registers (loaders)
    load_piece(e0 + j * TS_BATCH, mt, rk0, rp0, rs0);
    if (two) load_piece(e0 + j * TS_BATCH, mt + NL, rk1, rp1, rs1);
  };
  auto syms_for = [&](const uint4 k, unsigned prev, const uint2 sy) { return DERIVED ? derive(k, prev) : QUAL ? sy : syms_of(k); };
  auto deposit = [&](unsigned j) {  // registers -> LDS buffer of batch j (loaders)
    kb4[j % NBUF][mt] = keys16_of(rk0);
    sb8[j % NBUF][mt] = syms_for(rk0, rp0, rs0);
    if (two) { kb4[j % NBUF][mt + NL] = keys16_of(rk1); sb8[j % NBUF][mt + NL] = syms_for(rk1, rp1, rs1); }
  };
  // batch j has been ranked: its buffer holds positions where the keys were.  Symbols to their places, positions out.
  auto finish_piece = [&](unsigned j, unsigned piece) {
    const unsigned i0 = piece * 8u, nb = min(TS_BATCH, nt - j * TS_BATCH);
    if (i0 >= nb) return;
    const uint4 p4 = kb4[j % NBUF][piece];
    const uint2 s2 = PLACE ? sb8[j % NBUF][piece] : make_uint2(0, 0);
    const unsigned pw[4] = {p4.x, p4.y, p4.z, p4.w};
    uint16_t *gpos = lpos16 + e0 + j * TS_BATCH + i0;
    if (i0 + 8u <= nb) {
      *reinterpret_cast<uint4 *>(gpos) = p4;
      if (PLACE)
#pragma unroll
        for (unsigned k = 0; k < 8; k++)
          lsym[(pw[k >> 1] >> (16u * (k & 1u))) & 0xFFFFu] = (uint8_t)(((k < 4 ? s2.x : s2.y) >> (8u * (k & 3u))) & 0xFFu);
    } else {
#pragma unroll
      for (unsigned k = 0; k < 8; k++)
        if (i0 + k < nb) {
          const unsigned pos = (pw[k >> 1] >> (16u * (k & 1u))) & 0xFFFFu;
          gpos[k] = (uint16_t)pos;
          if (PLACE) lsym[pos] = (uint8_t)(((k < 4 ? s2.x : s2.y) >> (8u * (k & 3u))) & 0xFFu);
        }
    }
  };
  auto finish = [&](unsigned j) {
    finish_piece(j, mt);
    if (two) finish_piece(j, mt + NL);
  };
  // Barrier of the batch loop: LDS traffic only.  __syncthreads() also drains vmcnt, i.e. the loaders would wait at
  // every barrier for the batch they have just requested and for their position stores (fire and forget).
  auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  {  // batch 0 by everybody, straight to LDS
    for (unsigned i = tid; i < TS_BATCH / 8; i += TS_THREADS) {
      uint4 k;
      unsigned prev = 0;
      uint2 sy = make_uint2(0, 0);
      load_piece(e0, i, k, prev, sy);
      kb4[0][i] = keys16_of(k);
      sb8[0][i] = syms_for(k, prev, sy);
    }
  }
  if (wave != 0 && nbatch > 1) request(1);
  __syncthreads();
  // One context with an eighth of the tile or more: its symbols meet on ONE cursor, and same-address
  // LDS atomics of an instruction are served one lane after the other (binned qualities: the partition
  // of a tile took twice as long, one quality everywhere: sixteen times).  Such data comes in RUNS --
  // neighbouring symbols share the context -- so the combining ranker lets the first lane of every run
  // add the run's length and hands the result down the run.  More instructions per symbol, hence
  // only for such tiles; the ranks are the same numbers either way.
  const bool combine = s_max * 8u >= nt;
  const unsigned long long lanes_le = (2ull << lane) - 1ull;
  for (unsigned j = 0; j <= nbatch; j++) {  // (one period more than there are batches: the last batch is finished in it)
    if (wave == 0 && j < nbatch && combine) {
      const uint16_t *kb = reinterpret_cast<const uint16_t *>(kb4[j % NBUF]);
      const uint8_t *sbb = reinterpret_cast<const uint8_t *>(sb8[j % NBUF]);
      uint16_t *gpos = lpos16 + e0 + j * TS_BATCH;
      const unsigned nb = fq_uniform(min(TS_BATCH, nt - j * TS_BATCH));  // (wave-uniform by construction; said so, the loop below runs on a scalar counter: tests/test_build_invariants.py)
      constexpr unsigned G = 4;  // (four iterations in flight: eight cost 22 more VGPRs for the whole kernel, which every tile pays)
      for (unsigned cb = 0; cb < nb; cb += 64 * G) {  // wave-uniform trip count, branch-free, as below
        unsigned key[G], pos[G], head_of[G], sy[G];
#pragma unroll
        for (unsigned g = 0; g < G; g++) {
          const unsigned i = cb + 64 * g + lane;
          key[g] = kb[i];
          sy[g] = (unsigned)sbb[i];
        }
#pragma unroll
        for (unsigned g = 0; g < G; g++) {
          const bool on = cb + 64 * g + lane < nb;
          const unsigned c = on ? (QUAL ? key[g] : key[g] & 0xFFu) : 0xFFFFFFFFu;  // lanes beyond the batch: a run of their own
          const unsigned left = (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFEu, (int)c, 0x138, 0xF, 0xF, false);  // lane - 1's context (wave_shr:1)
          const unsigned long long heads = __ballot(c != left);  // (lane 0 is one)
          const unsigned long long le = heads & lanes_le, gt = heads & ~lanes_le;
          const unsigned h = 63u - (unsigned)__clzll((long long)le);
          const unsigned nx = gt ? (unsigned)__ffsll((long long)gt) - 1u : 64u;
          const bool act = on && h == lane;
          const unsigned sh = 16u * (c & 1u);
          uint32_t *a = act ? &cursor32[c >> 1] : &dummy[lane];
          pos[g] = (atomicAdd(a, act ? (nx - lane) << sh : 0u) >> sh) & 0xFFFFu;
          head_of[g] = h;
        }
        // (this ranker finishes its symbols itself: its loop is short, and on such tiles -- long runs of one context --
        // the loaders' share of the work would be the longer one: constant data 4 %, real reads 2 % slower when split)
#pragma unroll
        for (unsigned g = 0; g < G; g++) {
          const unsigned i = cb + 64 * g + lane;
          const unsigned p = (unsigned)__shfl((int)pos[g], (int)head_of[g]) + (lane - head_of[g]);
          gpos[i] = (uint16_t)p;  // (beyond the batch: garbage that lands behind the tile's part of lpos16)
          lsym[i < nb ? p : TS_TILE + lane] = (uint8_t)sy[g];
        }
      }
    } else if (wave == 0 && j < nbatch) {
      uint16_t *kb = reinterpret_cast<uint16_t *>(kb4[j % NBUF]);
      const unsigned nb = fq_uniform(min(TS_BATCH, nt - j * TS_BATCH));  // (see above)
      // Wave-uniform trip count with the bound checked inside: with a per-lane trip count the
      // compiler's unrolling lets low lanes run ahead of high lanes by a whole group of iterations,
      // and the rank is only right if iteration k of every lane precedes iteration k + 1 of any lane.
      // Software-pipelined by hand, G iterations at a time: all key reads, then all atomics
      // (the LDS executes them in program order), then all position writes -- three LDS round trips per group
      // instead of three per iteration: the loop is a pure latency chain and ONE wave ranks a tile.
      // Branch-free: a lane beyond the batch adds 0 to cursor word 0 (its position is never used) -- behind a
      // branch hipcc waits for every atomic's return value before it issues the next one (s_waitcnt lgkmcnt(0)
      // in each arm), which is the whole latency again.
      constexpr unsigned G = 8;
#ifdef FQGPU_EXPERIMENTS
      const unsigned long long tr0 = wall_clock64();
#endif
      for (unsigned cb = 0; cb < nb; cb += 64 * G) {  // nothing in here touches global memory
        unsigned key[G], pos[G];
#pragma unroll
        for (unsigned g = 0; g < G; g++) key[g] = kb[cb + 64 * g + lane];  // (< TS_BATCH: the buffers hold a whole batch)
#pragma unroll
        for (unsigned g = 0; g < G; g++) {
          const bool on = cb + 64 * g + lane < nb;
          const unsigned c = on ? (QUAL ? key[g] : key[g] & 0xFFu) : 0u;
          pos[g] = (atomicAdd(&cursor32[c >> 1], on ? 1u << (16u * (c & 1u)) : 0u) >> (16u * (c & 1u))) & 0xFFFFu;
        }
#pragma unroll
        for (unsigned g = 0; g < G; g++) {
          const unsigned i = cb + 64 * g + lane;
          kb[i] = (uint16_t)pos[g];
        }
      }
#ifdef FQGPU_EXPERIMENTS
      if (lane == 0) atomicAdd(&g_ts_prof[PS + 4], wall_clock64() - tr0);
#endif
    } else if (wave != 0) {
      if (j >= 1 && !combine) finish(j - 1);         // ranked in the period before
      if (j >= 1 && j + 1 < nbatch) deposit(j + 1);  // requested one period ago, into the buffer finish() has just emptied
      if (j == 0 && nbatch > 1) deposit(1);
      if (j + 2 < nbatch) request(j + 2);
    }
    lds_barrier();
  }
  TS_PROF(PS + 1);

  // ---- the tile's runs, in context order: cur16[c] is now the END of context c's run
  __syncthreads();  // (also drains the position stores) the batch buffers are dead: the run map takes their place
  if (DERIVED) {
    // the first symbol of every record that starts in this tile: its position comes back from lpos16 (this workgroup's own
    // stores, complete behind the barrier above; read past the L1), its symbol from K1's side table
    unsigned lo = 0, hi = R;  // first record with rec_start >= e0
    while (lo < hi) {
      const unsigned mid = (lo + hi) >> 1;
      if (rec_start[mid] < e0) lo = mid + 1; else hi = mid;
    }
    for (unsigned r = lo + tid; r < R; r += TS_THREADS) {
      const unsigned e = rec_start[r];
      if (e >= e0 + nt) break;
      const unsigned w = __hip_atomic_load(reinterpret_cast<const uint32_t *>(lpos16) + (e >> 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      lsym[(w >> (16u * (e & 1u))) & 0xFFFFu] = first_sym[r];
    }
    __syncthreads();
  }
  ts_clear_bitmap<TS_THREADS>(rm);
  unsigned long long my_mask[(NCHUNK + TS_WAVES - 1) / TS_WAVES];
  unsigned my_beg[(NCHUNK + TS_WAVES - 1) / TS_WAVES], my_len[(NCHUNK + TS_WAVES - 1) / TS_WAVES];
#pragma unroll
  for (unsigned k = 0; k < (NCHUNK + TS_WAVES - 1) / TS_WAVES; k++) {
    const unsigned chunk = wave + k * TS_WAVES;
    my_mask[k] = 0; my_beg[k] = 0; my_len[k] = 0;
    if (chunk < NCHUNK) {
      const unsigned c = chunk * 64 + lane;
      const unsigned end = cur16[c], beg = c ? (unsigned)cur16[c - 1] : 0u;
      my_beg[k] = beg; my_len[k] = end - beg;
      my_mask[k] = __ballot(end > beg);
      if (lane == 0) s_cnt[chunk] = (unsigned)__popcll(my_mask[k]);
    }
  }
  __syncthreads();
  if (wave == 0) {  // exclusive scan of the chunk counts (at most 128 chunks: two per lane)
    const unsigned a = 2 * lane < NCHUNK ? s_cnt[2 * lane] : 0u, b = 2 * lane + 1 < NCHUNK ? s_cnt[2 * lane + 1] : 0u;
    unsigned inc = a + b;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned o = __shfl_up(inc, d);
      if (lane >= (unsigned)d) inc += o;
    }
    if (2 * lane < NCHUNK) s_cnt[2 * lane] = inc - a - b;
    if (2 * lane + 1 < NCHUNK) s_cnt[2 * lane + 1] = inc - b;
    if (lane == 63) s_nruns = inc;
  }
  __syncthreads();
  uint2 *rlist = runs + (size_t)tile * ts_run_stride<M>();
#pragma unroll
  for (unsigned k = 0; k < (NCHUNK + TS_WAVES - 1) / TS_WAVES; k++) {
    const unsigned chunk = wave + k * TS_WAVES;
    if (chunk < NCHUNK && my_len[k]) {
      const unsigned c = chunk * 64 + lane;
      const unsigned slot = s_cnt[chunk] + fq_mbcnt(my_mask[k]);
      const unsigned slot_base = tb_row[c];
      rlist[slot] = make_uint2(slot_base, my_beg[k] | (my_len[k] << 16));
      if (slot < GD_CAP) gd[slot] = slot_base - my_beg[k];  // "global slot of local position p" = p + gd[run of p]
      atomicOr(&rm.bm[my_beg[k] >> 5], 1u << (my_beg[k] & 31u));
    }
  }
  __syncthreads();  // run list (global, this workgroup's own stores) and bitmap complete
  ts_build_wpre<TS_THREADS>(rm, wsum);
  if (tid == 0) run_count[tile] = s_nruns;
  TS_PROF(PS + 2);
  // The sorted tile goes out in pieces of 16 positions per thread.  A piece that lies inside ONE run -- nearly all of them
  // for the sequence stream (runs of 128), most for the quality stream -- is one 16-byte LDS read, one run lookup and one
  // 16-byte store (round 3: a run lookup = two LDS reads and a global read, and a byte store, per POSITION); a piece with
  // run boundaries inside walks its bytes, stepping to the next run where the boundary map says so.  +1.3 % on the step.
  // (The same for K6's gather of the (nb, bits) LOST 11 %: there a piece with boundaries walks sixteen dependent global
  // loads where the position-major loop keeps sixteen independent ones in flight per thread.)
  for (unsigned p0 = tid * 16u; p0 < nt; p0 += TS_THREADS * 16u) {
    const unsigned wd = p0 >> 5, shb = p0 & 31u;  // (p0 is a multiple of 16: the piece is one half of a map word)
    const unsigned mw = rm.bm[wd];
    const unsigned starts = (mw >> shb) & 0xFFFFu;   // run starts at positions p0 .. p0 + 15
    unsigned r = (unsigned)rm.wpre[wd] + __popc(mw & ((1u << shb) - 1u)) + (starts & 1u) - 1u;  // run of position p0
    const uint4 v = *reinterpret_cast<const uint4 *>(lsym + p0);
    const unsigned n = min(16u, nt - p0);
    auto delta_of = [&](unsigned run) {
      if (run < GD_CAP) return gd[run];
      const uint2 e = rlist[run];
      return e.x - (e.y & 0xFFFFu);
    };
    unsigned delta = delta_of(r);
    if ((starts & 0xFFFEu) == 0u && n == 16u) {
      *reinterpret_cast<FqBytes16 *>(sorted_sym + p0 + delta) = FqBytes16{{v.x, v.y, v.z, v.w}};  // (16 bytes at any address)
    } else {
      const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (unsigned i = 0; i < 16; i++) {
        if (i && ((starts >> i) & 1u)) { r++; delta = delta_of(r); }
        if (i < n) sorted_sym[p0 + i + delta] = (uint8_t)(w4[i >> 2] >> (8u * (i & 3u)));
      }
    }
  }
  __syncthreads();
  TS_PROF(PS + 3);
}

// ------------------------------------------------------------------ K6: gather + bit offsets + packing, fused
// Tile bit totals are chained by a decoupled look-back: status[t] = flag << 62 | bits, flag 1 =
// the tile's own total, 2 = inclusive total of tiles 0 .. t; one naturally aligned 8-byte word per
// tile, written and read with relaxed agent-scope atomics (value and flag travel together, so no
// fence is needed).  Tiles are handed out by an atomic counter: a tile only ever waits for tiles
// that were handed out before it, i.e. that are already running.
constexpr unsigned long long TS_FLAG_AGG = 1ull << 62, TS_FLAG_INCL = 2ull << 62, TS_VAL_MASK = (1ull << 62) - 1ull;

template <class M>
__global__ void __launch_bounds__(TS_GP_THREADS)
k_tile_gather_pack(const uint16_t *__restrict__ lpos16, const uint2 *__restrict__ runs,
                   const uint32_t *__restrict__ run_count, const uint16_t *__restrict__ out16, unsigned n_sym,
                   unsigned n_tiles, unsigned long long *__restrict__ status, unsigned *__restrict__ tile_counter,
                   const uint32_t *__restrict__ log_prefix, unsigned long long cap, uint32_t *__restrict__ out,
                   StreamResult *res, unsigned long long *__restrict__ tile_bit_base, uint4 *__restrict__ edges) {
  constexpr unsigned NW = TS_SUB * 12 / 32 + 4;
  __shared__ uint16_t vals[TS_TILE];  // (nb, bits) of the tile in sorted order
  __shared__ uint32_t words[NW];
  __shared__ TsRunMap rm;
  __shared__ unsigned wsum[TS_GP_THREADS / 64], s_tile;
  __shared__ unsigned long long s_base;
  const unsigned tid = threadIdx.x, wave = tid >> 6, lane = fq_lane();
  if (tid == 0) {
    s_tile = atomicAdd(tile_counter, 1u);
    if (s_tile < n_tiles) edges[s_tile] = make_uint4(0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u);  // no shared words yet
  }
  ts_clear_bitmap<TS_GP_THREADS>(rm);
  __syncthreads();
  const unsigned tile = s_tile;
  if (tile >= n_tiles) return;  // (uniform; the grid is exactly n_tiles)
  TS_PROF_DECL
  constexpr unsigned PS = M::STREAM == 1 ? 16 : 24; (void)PS;
  const unsigned e0 = tile * TS_TILE, nt = min(TS_TILE, n_sym - e0);
  const uint2 *rlist = runs + (size_t)tile * ts_run_stride<M>();
  const unsigned nr = run_count[tile];
  // "global slot of local position p" = p + gd[run of p]; the deltas of the first NW runs sit in LDS
  // (the packing buffer is idle until the tile's values are in), later runs are read from the list
  uint32_t *gd = words;
  for (unsigned r = tid; r < nr; r += TS_GP_THREADS) {
    const uint2 e = rlist[r];
    const unsigned beg = e.y & 0xFFFFu;
    if (r < NW) gd[r] = e.x - beg;
    atomicOr(&rm.bm[beg >> 5], 1u << (beg & 31u));
  }
  __syncthreads();
  ts_build_wpre<TS_GP_THREADS>(rm, wsum);
  TS_PROF(PS + 0);
  // ---- the runs of (nb, bits): consecutive lanes on consecutive slots
  unsigned bits = 0;
  if (nr <= NW) {  // (uniform) every lookup stays in LDS: the loads of sixteen positions per thread are in flight together
#pragma unroll 16
    for (unsigned p = tid; p < nt; p += TS_GP_THREADS) {
      const unsigned v = out16[p + gd[ts_run_of(rm, p)]];
      vals[p] = (uint16_t)v;
      bits += v >> 12;
    }
  } else {
#pragma unroll 8
    for (unsigned p = tid; p < nt; p += TS_GP_THREADS) {
      const uint2 r = rlist[ts_run_of(rm, p)];
      const unsigned v = out16[r.x + (p - (r.y & 0xFFFFu))];
      vals[p] = (uint16_t)v;
      bits += v >> 12;
    }
  }
  unsigned tile_bits;
  (void)ts_block_scan<TS_GP_THREADS>(bits, wsum, &tile_bits);  // (also the barrier behind the stores to vals)
  TS_PROF(PS + 1);
  // ---- bit offset of the tile
  if (wave == 0) {
    if (lane == 0)
      __hip_atomic_store(&status[tile], (tile == 0 ? TS_FLAG_INCL : TS_FLAG_AGG) | (unsigned long long)tile_bits, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long excl = 0;
    if (tile > 0) {
      int first = (int)tile - 1;  // lane l looks at tile first - l
      for (;;) {
        const int idx = first - (int)lane;
        unsigned long long st = TS_FLAG_AGG;  // (tiles in front of tile 0: empty aggregates)
        if (idx >= 0) {
          do {
            st = __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((st >> 62) == 0) __builtin_amdgcn_s_sleep(1);
          } while ((st >> 62) == 0);
        }
        const unsigned long long incl_mask = __ballot((st >> 62) == 2ull);
        const unsigned stop = incl_mask ? (unsigned)__ffsll((long long)incl_mask) - 1u : 64u;  // nearest inclusive total
        unsigned long long v = lane <= stop ? (st & TS_VAL_MASK) : 0ull;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
        excl += v;
        if (incl_mask || first < 64) break;
        first -= 64;
      }
      if (lane == 0)
        __hip_atomic_store(&status[tile], TS_FLAG_INCL | (excl + tile_bits), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) {
      s_base = excl;
      tile_bit_base[tile] = excl;
      if (tile == n_tiles - 1) {  // size and verdict = BIT_closeCStream: 0 when the write pointer reached dst+cap-8
        const unsigned long long payload = excl + tile_bits, all = payload + log_prefix[M::B] + 1ull;
        tile_bit_base[n_tiles] = payload;
        res->total_bits = payload;
        res->len = (all + 7ull) >> 3;
        if (cap <= 8ull || (all >> 3) >= cap - 8ull) atomicOr(&res->overflow, 1u);
      }
    }
  }
  __syncthreads();
  TS_PROF(PS + 2);
  unsigned long long cursor = s_base;  // bit offset of the next sub-tile (uniform)
  // Nobody zeroes the stream (round 3: a memset of its whole capacity per stream and block, 138 MB).  A word that this
  // tile alone fills is stored; a word it shares -- with the tile in front (its first word, when the tile does not start
  // on a word boundary), with the tile or the state flush behind (its last word) -- is NOT written here: the tile's share
  // of it goes to edges[tile] = {first word's index, bits, last word's index, bits}, and k_epilogue, which runs behind this
  // kernel, zeroes every such word once and ORs the shares in.  Between two rounds of one tile the open word travels in `carry`.
  const unsigned first_word = (unsigned)(s_base >> 5);
  const bool first_shared = (s_base & 31ull) != 0;
  unsigned carry = 0;  // (uniform) this tile's bits in the word that holds bit `cursor`, when cursor is not on a word boundary
  // ---- packing, TS_SUB symbols per round: thread t owns 16 consecutive symbols
  // (the positions of the next round are requested before the current one is packed)
  constexpr unsigned PPT = TS_GP_PPT, Q4 = PPT / 8;  // 16-byte pieces of positions per thread and round
  const uint4 *lp4 = reinterpret_cast<const uint4 *>(lpos16 + e0 + tid * PPT);
  uint4 nx[Q4];
#pragma unroll
  for (unsigned q = 0; q < Q4; q++) nx[q] = tid * PPT < nt ? lp4[q] : make_uint4(0, 0, 0, 0);
  for (unsigned s0 = 0; s0 < nt; s0 += TS_SUB) {
    for (unsigned i = tid; i < NW; i += TS_GP_THREADS) words[i] = i == 0 ? carry : 0u;
    const unsigned el = s0 + tid * PPT;  // local encode index of the thread's first symbol
    unsigned v[PPT];
    unsigned tb = 0;
    {
      uint4 cur4[Q4];
#pragma unroll
      for (unsigned q = 0; q < Q4; q++) cur4[q] = nx[q];
      if (el + TS_SUB < nt) {
        const uint4 *n4 = lp4 + (s0 + TS_SUB) / 8;
#pragma unroll
        for (unsigned q = 0; q < Q4; q++) nx[q] = n4[q];
      }
#pragma unroll
      for (unsigned i = 0; i < PPT; i++) {
        const uint4 c4 = cur4[i >> 3];
        const unsigned w2 = ((i >> 1) & 3u) == 0 ? c4.x : ((i >> 1) & 3u) == 1 ? c4.y : ((i >> 1) & 3u) == 2 ? c4.z : c4.w;
        const unsigned lp = (w2 >> (16 * (i & 1))) & 0xFFFFu;
        v[i] = el + i < nt ? (unsigned)vals[lp] : 0u;
        tb += v[i] >> 12;
      }
    }
    unsigned sub_bits;
    unsigned off = ts_block_scan<TS_GP_THREADS>(tb, wsum, &sub_bits);  // (its barriers also order the zeroing of words)
    const unsigned long long b0 = cursor, b1 = cursor + sub_bits;
    off += (unsigned)(b0 & 31ull);
    unsigned long long acc = 0;
    unsigned nacc = off & 31u, w = off >> 5;
#pragma unroll
    for (unsigned i = 0; i < PPT; i++) {
      const unsigned nb = v[i] >> 12;
      acc |= (unsigned long long)(v[i] & 0xFFFu) << nacc;
      nacc += nb;
      if (nacc >= 32) {
        if ((uint32_t)acc) atomicOr(&words[w], (uint32_t)acc);  // (OR-ing nothing is left out: on data that codes in a fraction of a bit per
        acc >>= 32; nacc -= 32; w++;                            //  symbol all threads of a wave would meet on one word)
      }
    }
    if ((uint32_t)acc) atomicOr(&words[w], (uint32_t)acc);
    __syncthreads();
    {
      const unsigned long long gw0 = b0 >> 5;
      const unsigned nw = (unsigned)(((b1 + 31ull) >> 5) - gw0);   // words that hold bits of this tile up to b1 (word 0: the carry's)
      const bool tail_open = (b1 & 31ull) != 0;
      const unsigned n_final = tail_open ? nw - 1u : nw;           // ... and are complete as far as this tile goes
      const bool fits = (gw0 + nw) * 4ull <= cap + 32ull;  // (the buffer has 64 spare bytes; an overflowing stream is discarded)
      const unsigned next_carry = tail_open ? words[nw - 1u] : 0u;
      if (fits) {
        for (unsigned i = tid; i < n_final; i += TS_GP_THREADS) {
          const unsigned gw = (unsigned)gw0 + i;
          if (gw == first_word && first_shared) { edges[tile].x = gw; edges[tile].y = words[i]; }  // (one thread, once per tile)
          else out[gw] = words[i];
        }
      } else if (tid == 0) atomicOr(&res->overflow, 1u);
      carry = next_carry;
    }
    cursor = b1;
    __syncthreads();  // words are rewritten by the next round
  }
  if (tid == 0 && (cursor & 31ull) != 0 && (cursor >> 5) * 4ull + 4ull <= cap + 32ull) {  // the open last word: shared with whatever follows
    const unsigned gw = (unsigned)(cursor >> 5);
    if (gw == first_word && first_shared) { edges[tile].x = gw; edges[tile].y = carry; }  // the whole tile inside one word
    else { edges[tile].z = gw; edges[tile].w = carry; }
  }
  TS_PROF(PS + 3);
}
