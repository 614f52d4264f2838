// Bit-exact parallel encoder of one (block, stream) -- replaces the per-record
// loop of CompressionWorkspace::encodeChunk (reference src/workspace.cpp:25-31)
// with its SequenceEncoder::encodeRecord (src/fse_sequence.cpp:53-112) /
// QualityEncoder::encodeRecord (src/fse_quality.cpp:5-53) bodies and
// FSE_Encoder::startChunk/endChunk (src/fse_common.hpp:77-90).
//
// The reference interleaves one tANS state per context into ONE bitstream:
//   for every symbol e in encode order (records in file order, positions L-1..0):
//       emit low nb(e) bits of state[ctx(e)];  state[ctx(e)] = T[ctx(e)][sym(e)](state)
//   then flush all states (ctx ascending), one '1' end-mark bit, zero pad.
// ctx(e)/sym(e) depend on the INPUT only, each context's state chain only on the
// symbols of that context.  Hence the decomposition:
//   K1 tile_hist / K2 layout : counting sort keys = context, per-tile histograms + scan
//   K3 scatter               : stable partition of the symbols by context (rank inside a
//                              (tile, context) = one lane-ordered LDS atomic per symbol); the
//                              sequence stream sorts every 4 K batch by context in LDS and
//                              writes runs instead of scattered bytes
//   K4 chains                : a tANS encoder state never forgets its history completely, but
//                              the SET of states it can be in collapses fast.  A chain is cut
//                              into segments of 4096 symbols; the state is known without its
//                              past right after a symbol whose normalised count is 1 or -1 (one
//                              table cell: "reset"), and for a segment without such a symbol
//                              one wave computes the segment's FUNCTION entry state -> exit
//                              state for every possible entry state over collapsing state sets.
//                              Entry states follow by applying the functions along the chain;
//                              then one lane per segment emits (nb, bits).  No serial chain,
//                              no speculation.
//   K6 bitcount/scan/pack    : per-symbol (nb,bits) brought back into encode order (quality:
//                              gather by slot; sequence: runs of the batch read into LDS),
//                              exclusive scan of nb = bit offsets, bit packing through LDS
//   K7 epilogue              : state flush + end mark + size/overflow
#include "fqgpu_internal.h"

#include <cstdlib>
#include <type_traits>

namespace {

constexpr unsigned TILE_SEQ = 32768;   // symbols per partition tile (one wave each in K3)
constexpr unsigned TILE_QUAL = 65536;
constexpr unsigned GROUP_TILES = 64;   // tiles per group in the two-level tile scan
constexpr unsigned PACK_TILE = 4096;   // symbols per bit-packing tile
constexpr unsigned PACK_THREADS = 256;
constexpr unsigned PACK_PER_THREAD = PACK_TILE / PACK_THREADS;  // 16
constexpr unsigned CTX_PAD = 16;       // every context's sorted run starts 16-aligned

template <class M> constexpr unsigned tile_size() { return M::STREAM == 0 ? TILE_SEQ : TILE_QUAL; }

// ------------------------------------------------------------------ record-level kernels

// readlens + N count per record (replaceAndEncodeNs, src/fse_sequence.cpp:35-51, first half).
// One wave per record, lanes stride the bases.
__global__ void __launch_bounds__(256)
k_readlens_ncount(const uint8_t *__restrict__ raw, const fqgpu_rec *__restrict__ recs, unsigned R,
                  uint16_t *__restrict__ readlens, uint16_t *__restrict__ n_count,
                  uint32_t *__restrict__ n_cnt32, uint32_t *__restrict__ lens32) {
  const unsigned waves = (gridDim.x * blockDim.x) >> 6;
  const unsigned lane = fq_lane();
  for (unsigned r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < R; r += waves) {
    const fqgpu_rec rec = recs[r];
    const uint8_t *s = raw + rec.seq_off;
    unsigned cnt = 0;
    for (unsigned base = 0; base < rec.len; base += 64) {
      const unsigned i = base + lane;
      const bool isn = i < rec.len && s[i] == 'N';
      cnt += (unsigned)__popcll(__ballot(isn));
    }
    if (lane == 0) {
      readlens[r] = (uint16_t)rec.len;
      n_count[r] = (uint16_t)cnt;
      n_cnt32[r] = cnt;
      lens32[r] = rec.len;
    }
  }
}

// N position deltas (second half of replaceAndEncodeNs) + optional N -> A write-back
__global__ void __launch_bounds__(256)
k_npos(uint8_t *__restrict__ raw, const fqgpu_rec *__restrict__ recs, unsigned R,
       const uint32_t *__restrict__ n_off, uint16_t *__restrict__ n_pos, int write_back) {
  const unsigned waves = (gridDim.x * blockDim.x) >> 6;
  const unsigned lane = fq_lane();
  for (unsigned r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < R; r += waves) {
    const unsigned first = n_off[r];
    if (n_off[r + 1] == first) continue;  // no N in this record
    const fqgpu_rec rec = recs[r];
    uint8_t *s = raw + rec.seq_off;
    unsigned done = 0, prev = 0;  // N's written so far, position of the last one (0 before any)
    for (unsigned base = 0; base < rec.len; base += 64) {
      const unsigned i = base + lane;
      const bool isn = i < rec.len && s[i] == 'N';
      const unsigned long long m = __ballot(isn);
      if (isn) {
        const unsigned long long below = m & ((1ull << lane) - 1ull);
        const unsigned p = below ? base + (63u - (unsigned)__clzll(below)) : prev;
        n_pos[first + done + (unsigned)__popcll(below)] = (uint16_t)(i - p);
        if (write_back) s[i] = 'A';
      }
      if (m) prev = base + (63u - (unsigned)__clzll(m));
      done += (unsigned)__popcll(m);
    }
  }
}

__global__ void k_store_npos_len(const uint32_t *__restrict__ n_off, unsigned R, BlockResult *res) {
  res->n_pos_len = n_off[R];
}

// ------------------------------------------------------------------ walking symbols in encode order
// Encode order = records in file order, positions L-1 .. 0 inside a record
// (src/fse_sequence.cpp:76-77,101; src/fse_quality.cpp:7,19).  A wave walks a range of encode
// indices 64 at a time; the record of every lane is found by stepping through the (few)
// records a chunk touches with wave-uniform loads instead of a per-lane binary search.
// The records a wave is walking through are cached 64 at a time in LDS (one coalesced load per
// 64 records instead of a dependent global round trip in front of every 64-symbol chunk).
struct RecCache {
  uint32_t start[65];   // rec_start of records r0 .. r0 + 64
  fqgpu_rec rec[64];
};

constexpr int K1_DEPTH = 4;  // chunk buffers of K1's software pipeline

struct SymbolWalker {
  const fqgpu_rec *__restrict__ recs;
  const uint32_t *__restrict__ rec_start;
  unsigned r;   // record holding the first symbol of the next chunk (wave-uniform)
  unsigned R;   // number of records
  RecCache *cache;
  unsigned r0;  // first cached record

  // window of 64 records starting at `first`; returns the first encode index it does NOT cover
  __device__ __forceinline__ unsigned refill(unsigned first) {
    const unsigned lane = fq_lane();
    r0 = first;
    fq_lds_wave_sync();  // nobody still reads the old window
    cache->start[lane] = first + lane <= R ? rec_start[first + lane] : 0xFFFFFFFFu;
    if (lane == 0) cache->start[64] = first + 64 <= R ? rec_start[first + 64] : 0xFFFFFFFFu;
    if (first + lane < R) cache->rec[lane] = recs[first + lane];
    fq_lds_wave_sync();
    return __builtin_amdgcn_readfirstlane(cache->start[64]);
  }

  // lanes with valid == true get their record and position.  The chunk [eb, eb + 64) must lie
  // inside the cached window (no global memory operation in here).
  __device__ __forceinline__ void locate(unsigned eb, unsigned e_end, unsigned e, bool valid,
                                         fqgpu_rec &rec, unsigned &p) {
    const unsigned chunk_end = min(eb + 64u, e_end);
    unsigned rr = r;
    rec.seq_off = rec.qual_off = rec.len = 0;
    p = 0;
    for (;;) {
      const unsigned k = rr - r0;
      const unsigned rs = __builtin_amdgcn_readfirstlane(cache->start[k]),
                     rn = __builtin_amdgcn_readfirstlane(cache->start[k + 1]);
      if (valid && e >= rs && e < rn) { rec = cache->rec[k]; p = rec.len - 1u - (e - rs); }
      if (rn > chunk_end) break;            // record rr continues into the next chunk
      rr++;
      if (rn == chunk_end) break;           // next chunk starts exactly at record rr
    }
    r = rr;
  }
};

// ------------------------------------------------------------------ K1: per-tile context histogram
// Also leaves the key of every symbol in encode order, so that the partition pass is a plain
// prefetchable linear scan: ckey[e] = ctx | sym << 8 (sequence: 10 bits) or ctx (quality: 13
// bits, the symbol goes to csym[e]).  Two or three bytes per symbol instead of four: the key
// stores alone were 2.7 of the 21 ms step (tools/traffic_experiment.py).
template <class M>
__global__ void __launch_bounds__(256)
k_tile_hist(const uint8_t *__restrict__ raw, const fqgpu_rec *__restrict__ recs,
            const uint32_t *__restrict__ rec_start, unsigned R, unsigned n_sym, unsigned T,
            uint32_t *__restrict__ tile_hist, uint16_t *__restrict__ ckey, uint8_t *__restrict__ csym,
            StreamResult *res, int dbg) {
  __shared__ uint32_t hist[M::B];
  __shared__ RecCache rcache[4];  // one per wave
  const unsigned tile = blockIdx.x;
  const unsigned e0 = tile * T;
  const unsigned e1 = min(e0 + T, n_sym);
  for (unsigned c = threadIdx.x; c < (unsigned)M::B; c += blockDim.x) hist[c] = 0;
  __syncthreads();
  // every wave takes a contiguous share of the tile (multiple of 64 symbols)
  const unsigned wave = threadIdx.x >> 6, lane = fq_lane();
  const unsigned per = (((e1 - e0) + blockDim.x - 1u) / blockDim.x) * 64u;
  const unsigned wb = min(e0 + wave * per, e1), we = min(wb + per, e1);
  bool bad = false;
  if (wb < we) {
    SymbolWalker w{recs, rec_start, fq_locate(rec_start, 0, R - 1, wb), R, &rcache[wave], 0};
    // Software pipeline over a ring of K1_DEPTH chunk buffers: the bytes of chunk i + K1_DEPTH - 1
    // are requested before chunk i is hashed, so a load has K1_DEPTH - 1 chunks of work to land.
    // (A two-stage version with "cur = nxt" at the end of the iteration made the register copy
    // wait for the load it had just issued: the full global latency in every iteration.)  The
    // pipeline runs over the chunks that lie inside one 64-record window of the LDS cache, so that
    // its body contains no global memory operation besides the byte loads and the key stores and
    // the compiler can keep several chunks' loads outstanding.
    SymBytes buf[K1_DEPTH];
    unsigned bp[K1_DEPTH];
    unsigned lim = 0;  // end (encode index) of the chunks of the current window
    auto fetch = [&](int slot, unsigned eb2) {
      const unsigned e2 = eb2 + lane;
      const bool v2 = e2 < lim;
      fqgpu_rec rec;
      unsigned p;
      w.locate(eb2, lim, e2, v2, rec, p);
      buf[slot] = fq_load_sym_bytes<M>(raw, rec, p, v2);
      bp[slot] = p;
    };
    auto consume = [&](int slot, unsigned eb2) {
      const unsigned e = eb2 + lane;
      if (e < lim) {
        unsigned ctx, sym;
        fq_ctx_from_bytes<M>(buf[slot], bp[slot], ctx, sym);
        bad |= sym >= (unsigned)M::A;
        if (!(dbg & 2)) {
          if (M::STREAM == 0) ckey[e] = (uint16_t)(ctx | ((sym & 3u) << 8));
          else { ckey[e] = (uint16_t)ctx; csym[e] = (uint8_t)(sym & 63u); }
        }
        if (!(dbg & 1)) atomicAdd(&hist[ctx], 1u);
      }
    };
    for (unsigned eb = wb; eb < we;) {
      const unsigned covered = w.refill(w.r);  // records w.r .. w.r + 63
      // whole chunks inside the window (the wave's last chunk may be short)
      lim = covered >= we ? we : wb + ((covered - wb) & ~63u);
#pragma unroll
      for (int d = 0; d < K1_DEPTH - 1; d++)
        if (eb + 64u * d < lim) fetch(d, eb + 64u * d);
      // steady state: straight-line fetch / consume (no branch the load counters could get lost in)
      for (; eb + 64u * (2 * K1_DEPTH - 2) < lim; eb += 64u * K1_DEPTH) {
#pragma unroll
        for (int d = 0; d < K1_DEPTH; d++) {
          fetch((d + K1_DEPTH - 1) % K1_DEPTH, eb + 64u * (d + K1_DEPTH - 1));
          consume(d, eb + 64u * d);
        }
      }
      for (; eb < lim; eb += 64u * K1_DEPTH) {  // drain
#pragma unroll
        for (int d = 0; d < K1_DEPTH; d++) {
          const unsigned cur = eb + 64u * d;
          if (cur < lim) {
            const unsigned nxt = cur + 64u * (K1_DEPTH - 1);
            if (nxt < lim) fetch((d + K1_DEPTH - 1) % K1_DEPTH, nxt);
            consume(d, cur);
          }
        }
      }
      eb = lim;
    }
  }
  if (bad) atomicOr(&res->bad_symbol, 1u);
  __syncthreads();
  if (dbg & 1) return;  // timing experiment: the previous encode's histogram stays
  for (unsigned c = threadIdx.x; c < (unsigned)M::B; c += blockDim.x)
    tile_hist[(size_t)tile * M::B + c] = hist[c];
}

// ------------------------------------------------------------------ K2: layout of the sorted arrays
// group_sum[g][c] = sum of tile_hist over the tiles of group g
__global__ void __launch_bounds__(256)
k_group_sum(const uint32_t *__restrict__ tile_hist, unsigned n_tiles, unsigned B,
            uint32_t *__restrict__ group_sum) {
  const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned g = blockIdx.y;
  if (c >= B) return;
  const unsigned t0 = g * GROUP_TILES, t1 = min(t0 + GROUP_TILES, n_tiles);
  uint32_t acc = 0;
  for (unsigned t = t0; t < t1; t++) acc += tile_hist[(size_t)t * B + c];
  group_sum[(size_t)g * B + c] = acc;
}

// One workgroup: per-context totals, exclusive scan over groups (in place), then the
// context layout: padded start of every context's run, segment and work-item prefix sums.
// arrays: ctx_count[B] | ctx_start[B+1] | seg_base[B+1] | item_base[B+1]
__global__ void __launch_bounds__(1024)
k_ctx_layout(uint32_t *__restrict__ group_sum, unsigned n_groups, unsigned B, unsigned S,
             uint32_t *__restrict__ arrays) {
  __shared__ unsigned part[3][1024];
  uint32_t *ctx_count = arrays, *ctx_start = arrays + B, *seg_base = ctx_start + B + 1,
           *item_base = seg_base + B + 1;
  for (unsigned c = threadIdx.x; c < B; c += blockDim.x) {
    uint32_t acc = 0;
    for (unsigned g = 0; g < n_groups; g++) {
      const uint32_t v = group_sum[(size_t)g * B + c];
      group_sum[(size_t)g * B + c] = acc;
      acc += v;
    }
    ctx_count[c] = acc;
  }
  __syncthreads();
  // blocked scan: thread t owns contexts [t*per, (t+1)*per)
  const unsigned per = (B + blockDim.x - 1) / blockDim.x;
  const unsigned c0 = threadIdx.x * per, c1 = min(c0 + per, B);
  unsigned a0 = 0, a1 = 0, a2 = 0;
  for (unsigned c = c0; c < c1; c++) {
    const unsigned n = ctx_count[c];
    const unsigned nseg = (n + S - 1) / S;
    a0 += (n + CTX_PAD - 1) & ~(CTX_PAD - 1);
    a1 += nseg;
    a2 += (nseg + 63) >> 6;
  }
  part[0][threadIdx.x] = a0; part[1][threadIdx.x] = a1; part[2][threadIdx.x] = a2;
  __syncthreads();
  if (threadIdx.x < 3) {  // three short serial scans over 1024 partials
    unsigned run = 0;
    for (unsigned i = 0; i < blockDim.x; i++) {
      const unsigned v = part[threadIdx.x][i];
      part[threadIdx.x][i] = run;
      run += v;
    }
  }
  __syncthreads();
  a0 = part[0][threadIdx.x]; a1 = part[1][threadIdx.x]; a2 = part[2][threadIdx.x];
  for (unsigned c = c0; c < c1; c++) {
    const unsigned n = ctx_count[c];
    const unsigned nseg = (n + S - 1) / S;
    ctx_start[c] = a0; seg_base[c] = a1; item_base[c] = a2;
    a0 += (n + CTX_PAD - 1) & ~(CTX_PAD - 1);
    a1 += nseg;
    a2 += (nseg + 63) >> 6;
  }
  if (c1 == B && c0 < B) { ctx_start[B] = a0; seg_base[B] = a1; item_base[B] = a2; }
}

// tile_base[t][c] = ctx_start[c] + (symbols of context c in tiles before t)
__global__ void __launch_bounds__(256)
k_tile_base(const uint32_t *__restrict__ tile_hist, const uint32_t *__restrict__ group_sum,
            const uint32_t *__restrict__ ctx_start, unsigned n_tiles, unsigned B,
            uint32_t *__restrict__ tile_base) {
  const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned g = blockIdx.y;
  if (c >= B) return;
  const unsigned t0 = g * GROUP_TILES, t1 = min(t0 + GROUP_TILES, n_tiles);
  uint32_t acc = ctx_start[c] + group_sum[(size_t)g * B + c];
  for (unsigned t = t0; t < t1; t++) {
    tile_base[(size_t)t * B + c] = acc;
    acc += tile_hist[(size_t)t * B + c];
  }
}

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

// ------------------------------------------------------------------ K4: state chains
struct LdsCTable {
  const uint16_t *state_table;
  const uint32_t *tt;  // {deltaFindState, deltaNbBits} pairs
  unsigned log;
};

// copies one context's CTable (zstd word layout) into LDS; all 64 lanes of the wave
template <class M>
__device__ __forceinline__ LdsCTable stage_ctable(uint32_t *lds, const uint32_t *__restrict__ tbl) {
  const unsigned log = tbl[0] & 0xFFFFu;
  const unsigned words = 1u + (1u << (log - 1)) + 2u * M::A;
  for (unsigned i = fq_lane(); i < words; i += 64) lds[i] = tbl[i];
  __syncthreads();
  LdsCTable t;
  t.log = log;
  t.state_table = reinterpret_cast<const uint16_t *>(lds) + 2;
  t.tt = lds + 1 + (1u << (log - 1));
  return t;
}

// FSE_encodeSymbol (zstd fse.h) on the LDS copy: returns the packed (nb << 12 | bits)
__device__ __forceinline__ unsigned chain_step(const LdsCTable &t, unsigned &x, unsigned sym) {
  const int dfs = (int)t.tt[2 * sym];
  const unsigned dnb = t.tt[2 * sym + 1];
  const unsigned nb = (x + dnb) >> 16;
  const unsigned out = (nb << 12) | (x & ((1u << nb) - 1u));
  x = t.state_table[(int)(x >> nb) + dfs];
  return out;
}

// the one cell of a symbol with normalised count 1 or -1: stateTable[1 + deltaFindState]
__device__ __forceinline__ unsigned reset_state(const LdsCTable &t, unsigned sym) {
  return t.state_table[1 + (int)t.tt[2 * sym]];
}

// ---- sequence chains -------------------------------------------------------------------
// Sequence contexts have no single-state symbols, so a chain cannot be cut "for free", and a
// tANS encoder state never forgets its history.  But it forgets MOST of it: pushed through the
// same symbols, the 2^log possible states collapse onto a small set (about 150 survivors
// after 128 symbols for exactly uniform counts, a few dozen otherwise), because every
// transition x -> stateTable[(x >> nb) + delta] merges the states that share x >> nb.  The
// chain of a context is therefore cut into segments of S symbols and coded in three exact steps:
//  (A) k_seq_setfunc: one wave per segment computes F: entry state -> exit state for EVERY
//      possible entry state.  It starts with all 2^log states spread over the lanes, and at a
//      few points (after 4, 16, 48, 128, 512, 2048, ... symbols) replaces the states it carries by
//      the distinct ones ("classes"), remembering which class every entry state fell into.
//      After the first hundred symbols a step costs 1-3 LDS gathers per wave for 64 lanes.
//  (B) k_seq_resolve: entry state of every segment, x <- F_k[x] segment after segment.
//  (C) k_seq_emit: every lane walks ONE segment from its now-known entry state and writes the
//      packed (nb, bits) of every symbol; 64 segments of a context per wave.
// All three read the context's one-symbol transition table next[s][x] from LDS (tables.hip
// builds it once per handle).  Exact by construction: no speculation, nothing to verify.
constexpr unsigned SETS_WAVES = 8;          // segments (waves) per workgroup in step A, one-symbol table
constexpr unsigned SETS_WAVES2 = 16;        // ... with the 64 KB two-symbol table (one workgroup per CU)
constexpr unsigned SETS_ROUNDS = 4;         // a workgroup owns WAVES * SETS_ROUNDS segments, handed out to its waves one by one
constexpr unsigned SETS_MAX_CLASSES = 512;  // above this a segment keeps carrying every state
constexpr unsigned SETS_BLOCK = 1024;       // symbols per 16-byte-per-lane load; S is a multiple

struct SetsWaveLds {
  uint32_t bm[128];                  // bitmap over the states (size <= 4096)
  uint16_t wpre[128];                // set bits before every bitmap word
  uint16_t list[SETS_MAX_CLASSES];   // class -> state, as (state - size) * 2
  uint16_t tmp[SETS_MAX_CLASSES];    // old class -> new class during a merge
  uint16_t m[SETS_MAX_CLASSES];      // first-level class -> current class
};

// plan[]: fitem_base[B+1] (step A workgroups before every context) | fseg_base[B+1] (functions
// before every context) | seg_base[B+1] (segments) | eitem_base[B+1] (step C waves)
constexpr unsigned SEGPLAN_WORDS = 4 * (SeqModel::B + 1) + 4;  // + the work counter of step A

__global__ void __launch_bounds__(256)
k_seq_segplan(const uint32_t *__restrict__ arrays, unsigned S, unsigned wpg, uint32_t *__restrict__ plan) {
  constexpr unsigned B = SeqModel::B;
  __shared__ unsigned s_nseg[B];
  const unsigned c = threadIdx.x;
  const unsigned n = arrays[c];
  s_nseg[c] = (n + S - 1) / S;
  __syncthreads();
  unsigned fi = 0, fs = 0, sg = 0, ei = 0;
  for (unsigned o = 0; o < c; o++) {
    const unsigned ns = s_nseg[o], nf = ns ? ns - 1 : 0;
    fi += (nf + wpg - 1) / wpg; fs += nf; sg += ns; ei += (ns + 63) / 64;
  }
  uint32_t *fitem = plan, *fseg = plan + (B + 1), *seg = plan + 2 * (B + 1), *eitem = plan + 3 * (B + 1);
  fitem[c] = fi; fseg[c] = fs; seg[c] = sg; eitem[c] = ei;
  if (c == 0) plan[4 * (B + 1)] = 0;  // step A's work counter
  if (c == B - 1) {
    const unsigned ns = s_nseg[c], nf = ns ? ns - 1 : 0;
    fitem[B] = fi + (nf + wpg - 1) / wpg; fseg[B] = fs + nf; seg[B] = sg + ns; eitem[B] = ei + (ns + 63) / 64;
  }
}

// last context c with base[c] <= item (base is an exclusive prefix with B + 1 entries)
__device__ __forceinline__ unsigned seq_item_ctx(const uint32_t *__restrict__ base, unsigned item) {
  unsigned lo = 0, hi = SeqModel::B - 1;
  while (lo < hi) {
    const unsigned mid = lo + ((hi - lo + 1) >> 1);
    if (base[mid] <= item) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__device__ __forceinline__ unsigned sets_incl_scan(unsigned v) {
  const unsigned lane = fq_lane();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned o = __shfl_up(v, d);
    if (lane >= (unsigned)d) v += o;
  }
  return v;
}

// The segment's symbols travel 1024 at a time: lane l holds symbols 16 l .. 16 l + 15 of the
// block, a word of four symbols is fetched with v_readlane (w: uniform word index in the segment).
__device__ __forceinline__ unsigned sets_word(const uint4 cur, unsigned w) {
  const unsigned g = (w >> 2) & 63u, q = w & 3u;
  const unsigned a = __builtin_amdgcn_readlane(cur.x, g), b = __builtin_amdgcn_readlane(cur.y, g),
                 c = __builtin_amdgcn_readlane(cur.z, g), d = __builtin_amdgcn_readlane(cur.w, g);
  return q == 0 ? a : q == 1 ? b : q == 2 ? c : d;
}

// number of distinct states marked in L.bm; fills L.wpre
__device__ __forceinline__ unsigned sets_count(SetsWaveLds &L, unsigned nw) {
  const unsigned lane = fq_lane();
  const unsigned c0 = lane < nw ? __popc(L.bm[lane]) : 0u, c1 = lane + 64 < nw ? __popc(L.bm[lane + 64]) : 0u;
  const unsigned p0 = sets_incl_scan(c0), t0 = __builtin_amdgcn_readlane(p0, 63);
  const unsigned p1 = sets_incl_scan(c1), t1 = __builtin_amdgcn_readlane(p1, 63);
  L.wpre[lane] = (uint16_t)(p0 - c0);
  L.wpre[lane + 64] = (uint16_t)(t0 + p1 - c1);
  fq_lds_wave_sync();
  return t0 + t1;
}
__device__ __forceinline__ unsigned sets_rank(const SetsWaveLds &L, unsigned xi) {
  return (unsigned)L.wpre[xi >> 5] + __popc(L.bm[xi >> 5] & ((1u << (xi & 31u)) - 1u));
}
__device__ __forceinline__ void sets_clear(SetsWaveLds &L) {
  L.bm[fq_lane()] = 0; L.bm[fq_lane() + 64] = 0;
  fq_lds_wave_sync();
}

// byte offsets of the table rows the four symbols of a word select: one row per symbol, or
// (TWO) one row of the two-symbol table per symbol pair
template <bool TWO>
__device__ __forceinline__ void sets_rows(unsigned word, unsigned log, unsigned (&row)[TWO ? 2 : 4]) {
  if (TWO) {
    row[0] = ((word & 3u) | ((word >> 6) & 0xCu)) << (log + 1);
    row[1] = (((word >> 16) & 3u) | ((word >> 22) & 0xCu)) << (log + 1);
  } else {
#pragma unroll
    for (int i = 0; i < 4; i++) row[i] = ((word >> (8 * i)) & 3u) << (log + 1);
  }
}

// two-symbol table: the lane's 16 symbols as eight 16-bit row offsets, two per dword
__device__ __forceinline__ uint4 sets_pack_rows(const uint4 cur, unsigned log) {
  auto pk = [&](unsigned w) {
    return (((w & 3u) | ((w >> 6) & 0xCu)) << (log + 1)) | (((((w >> 16) & 3u) | ((w >> 22) & 0xCu)) << (log + 1)) << 16);
  };
  return make_uint4(pk(cur.x), pk(cur.y), pk(cur.z), pk(cur.w));
}

// n classes (states in L.list) walked through words [w0, w1) of the segment, M per lane
template <int M, bool TWO>
__device__ __forceinline__ void sets_walk(SetsWaveLds &L, unsigned n, const char *tbase, unsigned log,
                                          const uint4 cur, const uint4 rows, unsigned w0, unsigned w1) {
  const unsigned lane = fq_lane();
  unsigned y[M];
#pragma unroll
  for (int j = 0; j < M; j++) {
    const unsigned i = lane + 64u * j;
    y[j] = L.list[i < n ? i : n - 1];  // spare slots shadow the last class
  }
  auto step_word = [&](unsigned word) {
    unsigned row[TWO ? 2 : 4];
    sets_rows<TWO>(word, log, row);
#pragma unroll
    for (int i = 0; i < (TWO ? 2 : 4); i++) {
#pragma unroll
      for (int j = 0; j < M; j++) y[j] = *reinterpret_cast<const uint16_t *>(tbase + (row[i] + y[j]));
    }
  };
  if ((w0 | w1) & 3u) {  // only the first two ranges of a segment: [0, 1) and [1, 4)
    for (unsigned w = w0; w < w1; w++) step_word(sets_word(cur, w));
  } else if (TWO) {  // whole groups of 16 symbols = eight prepared row offsets of lane g
    for (unsigned g = w0 >> 2; g < (w1 >> 2); g++) {
      const unsigned gi = g & 63u;
      const unsigned r[4] = {(unsigned)__builtin_amdgcn_readlane(rows.x, gi), (unsigned)__builtin_amdgcn_readlane(rows.y, gi),
                             (unsigned)__builtin_amdgcn_readlane(rows.z, gi), (unsigned)__builtin_amdgcn_readlane(rows.w, gi)};
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const unsigned row = (i & 1) ? r[i >> 1] >> 16 : r[i >> 1] & 0xFFFFu;
#pragma unroll
        for (int j = 0; j < M; j++) y[j] = *reinterpret_cast<const uint16_t *>(tbase + (row + y[j]));
      }
    }
  } else {
    for (unsigned g = w0 >> 2; g < (w1 >> 2); g++) {
      const unsigned gi = g & 63u;
      const unsigned a = __builtin_amdgcn_readlane(cur.x, gi), b = __builtin_amdgcn_readlane(cur.y, gi),
                     c = __builtin_amdgcn_readlane(cur.z, gi), d = __builtin_amdgcn_readlane(cur.w, gi);
      step_word(a); step_word(b); step_word(c); step_word(d);
    }
  }
#pragma unroll
  for (int j = 0; j < M; j++) {
    const unsigned i = lane + 64u * j;
    if (i < n) L.list[i] = (uint16_t)y[j];
  }
  fq_lds_wave_sync();
}

// bitmap index of a carried state: XO = (state - size) * 2 (sequence kernels), else the state itself
template <bool XO>
__device__ __forceinline__ unsigned sets_idx(unsigned v, unsigned size) { return XO ? v >> 1 : v - size; }

// merge of equal states among the n classes of L.list; returns the new class count.  Skipped
// (list untouched) when it would not lower the number of gathers per step.
template <bool XO>
__device__ __forceinline__ unsigned sets_merge(SetsWaveLds &L, unsigned n, unsigned n1, unsigned nw, unsigned size) {
  const unsigned lane = fq_lane();
  sets_clear(L);
  for (unsigned i = lane; i < n; i += 64) {
    const unsigned xi = sets_idx<XO>(L.list[i], size);
    atomicOr(&L.bm[xi >> 5], 1u << (xi & 31u));
  }
  fq_lds_wave_sync();
  const unsigned nn = sets_count(L, nw);
  if ((nn + 63) / 64 >= (n + 63) / 64) return n;
  unsigned st[SETS_MAX_CLASSES / 64];
#pragma unroll
  for (unsigned j = 0; j < SETS_MAX_CLASSES / 64; j++) {
    const unsigned i = lane + 64u * j;
    st[j] = i < n ? (unsigned)L.list[i] : 0u;
    if (i < n) L.tmp[i] = (uint16_t)sets_rank(L, sets_idx<XO>(st[j], size));
  }
  fq_lds_wave_sync();
#pragma unroll
  for (unsigned j = 0; j < SETS_MAX_CLASSES / 64; j++) {
    const unsigned i = lane + 64u * j;
    if (i < n) L.list[L.tmp[i]] = (uint16_t)st[j];  // equal states write the same value
  }
  for (unsigned i = lane; i < n1; i += 64) L.m[i] = L.tmp[L.m[i]];
  fq_lds_wave_sync();
  return nn;
}

// Step A.  PER0 = states per lane at the start: 32 covers log <= 11, 64 covers log 12.
// TWO: two symbols per gather through the context's 64 KB two-symbol table (log <= 11).
template <unsigned PER0, bool TWO>
__global__ void __launch_bounds__((TWO ? SETS_WAVES2 : SETS_WAVES) * 64)
k_seq_setfunc(const uint8_t *__restrict__ sorted_sym, const uint32_t *__restrict__ arrays,
              const uint32_t *__restrict__ plan, const uint32_t *__restrict__ logs,
              const uint16_t *__restrict__ next, unsigned next_stride, unsigned S, unsigned fstride,
              uint16_t *__restrict__ fbuf, unsigned *__restrict__ work_counter) {
  constexpr unsigned WAVES = TWO ? SETS_WAVES2 : SETS_WAVES;
  extern __shared__ uint32_t lds[];  // next[4][size] (TWO: next2[16][size]) of this context
  __shared__ SetsWaveLds wl[WAVES];
  __shared__ unsigned s_next, s_item;
  constexpr unsigned B = SeqModel::B;
  const uint32_t *fitem = plan, *fseg = plan + (B + 1);
  const unsigned wave = threadIdx.x >> 6, lane = fq_lane();
  SetsWaveLds &L = wl[wave];
  const char *tbase = reinterpret_cast<const char *>(lds);
  const unsigned nblk = S / SETS_BLOCK, w_end = S / 4;
  const unsigned n_items = fitem[B];
  unsigned loaded = 0xFFFFFFFFu;  // context whose table is in LDS
  // Persistent workgroups (one per CU, 125 KB of LDS with the two-symbol table): items are
  // (context, group of WAVES * SETS_ROUNDS segments), taken from a global counter, so a
  // workgroup that has found a CU keeps it until the work is gone.
  for (;;) {
    __syncthreads();  // every wave is done with the previous item's table and queue
    if (threadIdx.x == 0) { s_item = atomicAdd(work_counter, 1u); s_next = WAVES; }
    __syncthreads();
    const unsigned item = s_item;
    if (item >= n_items) break;
    const unsigned c = seq_item_ctx(fitem, item);
    const unsigned log = logs[c], size = 1u << log;
    if (c != loaded) {  // (4 or 16) * size u16 entries, a multiple of 16 bytes
      const uint4 *src = reinterpret_cast<const uint4 *>(next + (size_t)c * next_stride);
      uint4 *dst = reinterpret_cast<uint4 *>(lds);
      for (unsigned e = threadIdx.x; e < (TWO ? 2 * size : size / 2); e += WAVES * 64) dst[e] = src[e];
      loaded = c;
      __syncthreads();
    }
    const unsigned nf = fseg[c + 1] - fseg[c];
    // the item's segments [k0, k_end) of the chain go to whichever wave is free
    const unsigned k0 = (item - fitem[c]) * (WAVES * SETS_ROUNDS), k_end = min(k0 + WAVES * SETS_ROUNDS, nf);
    const unsigned per = max(size >> 6, 1u), nw = max(size >> 5, 1u);
    for (unsigned k = k0 + wave; k < k_end;) {
      const uint4 *gseg = reinterpret_cast<const uint4 *>(sorted_sym + arrays[B + c] + (size_t)k * S);

      // level 0: every state; lane l carries states l, l + 64, ...
      unsigned x0[PER0];
#pragma unroll
      for (unsigned j = 0; j < PER0; j++) x0[j] = ((lane + 64u * j) & (size - 1)) * 2u;
      unsigned level = 0, n = size, n1 = 0;
      unsigned w = 0, stop = 1;  // merge points after 4, 16, 48, 128, 512, 2048, 8192, ... symbols
      uint4 cur = gseg[lane];
      for (unsigned blk = 0; blk < nblk; blk++) {
        const uint4 nxt = blk + 1 < nblk ? gseg[(size_t)(blk + 1) * 64 + lane] : cur;  // lands while cur is walked
        const unsigned wb_end = (blk + 1) * (SETS_BLOCK / 4);
        const uint4 rows = TWO ? sets_pack_rows(cur, log) : cur;
        while (w < wb_end) {
          const unsigned w1 = min(stop, wb_end);
          if (level == 0) {
            for (; w < w1; w++) {
              unsigned row[TWO ? 2 : 4];
              sets_rows<TWO>(sets_word(cur, w), log, row);
#pragma unroll
              for (int i = 0; i < (TWO ? 2 : 4); i++) {
#pragma unroll
                for (unsigned j = 0; j < PER0; j++)
                  if (j < per) x0[j] = *reinterpret_cast<const uint16_t *>(tbase + (row[i] + x0[j]));
              }
            }
          } else {
            switch ((n + 63) / 64) {
              case 1: sets_walk<1, TWO>(L, n, tbase, log, cur, rows, w, w1); break;
              case 2: sets_walk<2, TWO>(L, n, tbase, log, cur, rows, w, w1); break;
              case 3: sets_walk<3, TWO>(L, n, tbase, log, cur, rows, w, w1); break;
              case 4: sets_walk<4, TWO>(L, n, tbase, log, cur, rows, w, w1); break;
              case 5: sets_walk<5, TWO>(L, n, tbase, log, cur, rows, w, w1); break;
              case 6: sets_walk<6, TWO>(L, n, tbase, log, cur, rows, w, w1); break;
              case 7: sets_walk<7, TWO>(L, n, tbase, log, cur, rows, w, w1); break;
              default: sets_walk<8, TWO>(L, n, tbase, log, cur, rows, w, w1); break;
            }
            w = w1;
          }
          if (w != stop || w >= w_end) continue;
          stop = stop == 1 ? 4 : stop == 4 ? 12 : stop == 12 ? 32 : stop * 4;
          if (level == 0) {
            sets_clear(L);
#pragma unroll
            for (unsigned j = 0; j < PER0; j++)
              if (j < per) { const unsigned xi = x0[j] >> 1; atomicOr(&L.bm[xi >> 5], 1u << (xi & 31u)); }
            fq_lds_wave_sync();
            const unsigned nn = sets_count(L, nw);
            if (nn <= SETS_MAX_CLASSES) {  // from here on only the distinct states are carried
#pragma unroll
              for (unsigned j = 0; j < PER0; j++)
                if (j < per) {
                  const unsigned r = sets_rank(L, x0[j] >> 1);
                  L.list[r] = (uint16_t)x0[j];
                  x0[j] = r;  // class of entry state lane + 64 j
                }
              for (unsigned i = lane; i < nn; i += 64) L.m[i] = (uint16_t)i;
              fq_lds_wave_sync();
              level = 1; n = n1 = nn;
            }
          } else if (n > 64) {
            n = sets_merge<true>(L, n, n1, nw, size);
          }
        }
        cur = nxt;
      }
      // F[entry] = exit, both as (state - size) * 2
      uint16_t *f = fbuf + (size_t)(fseg[c] + k) * fstride;
#pragma unroll
      for (unsigned j = 0; j < PER0; j++) {
        const unsigned xi = lane + 64u * j;
        if (j < per && xi < size) f[xi] = level == 0 ? (uint16_t)x0[j] : L.list[L.m[x0[j]]];
      }
      unsigned nk = 0;
      if (lane == 0) nk = atomicAdd(&s_next, 1u);
      k = k0 + (unsigned)__builtin_amdgcn_readfirstlane(nk);
    }
  }
}

// Step B: entry state of every segment of every chain
__global__ void __launch_bounds__(256)
k_seq_resolve(const uint32_t *__restrict__ plan, const uint16_t *__restrict__ fbuf, unsigned fstride,
              uint16_t *__restrict__ entry) {
  constexpr unsigned B = SeqModel::B;
  const uint32_t *fseg = plan + (B + 1), *seg = plan + 2 * (B + 1);
  const unsigned c = threadIdx.x;
  const unsigned ns = seg[c + 1] - seg[c];
  unsigned xo = 0;  // FSE_initCState: state = size
  for (unsigned k = 0; k < ns; k++) {
    entry[seg[c] + k] = (uint16_t)xo;
    if (k + 1 < ns) xo = fbuf[(size_t)(fseg[c] + k) * fstride + (xo >> 1)];
  }
}

// Step C: one lane per segment, 64 segments of one context per wave
__global__ void __launch_bounds__(64)
k_seq_emit(const uint8_t *__restrict__ sorted_sym, uint16_t *__restrict__ out16,
           const uint32_t *__restrict__ arrays, const uint32_t *__restrict__ plan,
           const uint32_t *__restrict__ ct, const uint32_t *__restrict__ ct_off,
           const uint16_t *__restrict__ next1, unsigned next_stride, unsigned S,
           const uint16_t *__restrict__ entry, uint16_t *__restrict__ final_state, StreamResult *res) {
  extern __shared__ uint32_t lds[];
  constexpr unsigned B = SeqModel::B;
  const uint32_t *seg = plan + 2 * (B + 1), *eitem = plan + 3 * (B + 1);
  if (blockIdx.x >= eitem[B]) return;  // the grid is an upper bound
  const unsigned c = seq_item_ctx(eitem, blockIdx.x);
  const uint32_t *tbl = ct + ct_off[c];
  const unsigned log = tbl[0] & 0xFFFFu, size = 1u << log;
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(next1 + (size_t)c * next_stride);
    uint4 *dst = reinterpret_cast<uint4 *>(lds);
    for (unsigned e = threadIdx.x; e < size / 2; e += 64) dst[e] = src[e];
  }
  const uint32_t *tt = tbl + 1 + (size >> 1);
  unsigned dnb[4];
#pragma unroll
  for (int s = 0; s < 4; s++) dnb[s] = tt[2 * s + 1];
  fq_lds_wave_sync();
  const unsigned n = arrays[c], ns = seg[c + 1] - seg[c];
  const unsigned k = (blockIdx.x - eitem[c]) * 64 + fq_lane();
  if (k >= ns) return;
  const char *tbase = reinterpret_cast<const char *>(lds);
  const size_t run0 = (size_t)arrays[B + c] + (size_t)k * S;  // 16-byte aligned
  const unsigned len = min(S, n - k * S);
  const uint4 *gsym = reinterpret_cast<const uint4 *>(sorted_sym + run0);
  uint4 *gout = reinterpret_cast<uint4 *>(out16 + run0);
  unsigned xo = entry[seg[c] + k];
  const unsigned groups = (len + 15) >> 4;  // the run is padded to 16: the pad is walked and never read back
  uint4 sv = gsym[0];
  for (unsigned g = 0; g < groups; g++) {
    const uint4 sv_next = gsym[g + 1 < groups ? g + 1 : g];
    const unsigned wds[4] = {sv.x, sv.y, sv.z, sv.w};
    unsigned o[8];
    const unsigned live = min(16u, len - g * 16);
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const unsigned s = (wds[j >> 2] >> (8 * (j & 3))) & 3u;
      const unsigned x = size + (xo >> 1);
      const unsigned nb = (x + (s == 0 ? dnb[0] : s == 1 ? dnb[1] : s == 2 ? dnb[2] : dnb[3])) >> 16;
      const unsigned v = (nb << 12) | (x & ((1u << nb) - 1u));
      if (j & 1) o[j >> 1] |= v << 16; else o[j >> 1] = v;
      const unsigned nx = *reinterpret_cast<const uint16_t *>(tbase + ((s << (log + 1)) + xo));
      if ((unsigned)j < live) xo = nx;  // the state stops at the end of the chain
    }
    gout[2 * g] = make_uint4(o[0], o[1], o[2], o[3]);
    gout[2 * g + 1] = make_uint4(o[4], o[5], o[6], o[7]);
    sv = sv_next;
  }
  if (k == ns - 1) final_state[c] = (uint16_t)(size + (xo >> 1));
  if (fq_lane() == 0) atomicMax(&res->refixed, len);
}

// ---- generic chains: segments, single-state symbols and segment functions ---------------
// Works for any table set; used for the quality stream (and for the sequence stream with
// FQGPU_CHAIN_SEQ_GENERIC).  A symbol with normalised count 1 or -1 owns ONE table cell: every
// state emits `log` bits and lands on the same state ("reset" symbol), so the state after it is
// known without knowing anything before it.  The chain of a context is cut into segments of S
// symbols; a segment that contains a reset symbol is TRANSPARENT, one that does not is OPAQUE.
//  k_seg_scan    first reset symbol of every segment (one wave per segment, stops at the first
//                hit); lists the opaque segments
//  k_seg_walk<1> one lane per transparent segment: from behind its first reset symbol to the end
//                of the segment; its final state is the entry state of the next segment
//  k_seg_setfunc one wave per opaque segment: F: entry state -> exit state over collapsing state
//                sets, as k_seq_setfunc but stepping through the CTable (symbolTT + stateTable)
//  k_seg_resolve entry states behind opaque segments: x <- F[x] along every run of them
//  k_seg_walk<2> one lane per segment: the head of a transparent segment (up to and including
//                its first reset symbol) or a whole opaque segment, from the entry state
// Every lane walks at most S symbols, whatever the data: a context without reset symbols (binned
// or constant qualities) costs state-set work instead of one endless serial chain.
constexpr unsigned SEG_NONE = 0xFFFFFFFFu;

// segment table of one stream (all arrays indexed by the global segment number)
struct SegArrays {
  uint32_t *first_reset;  // offset of the first reset symbol inside the segment, or SEG_NONE
  uint32_t *fidx;         // function slot of an opaque segment
  uint32_t *olist;        // opaque segments that have a successor, in no particular order
  uint32_t *n_opaque;     // length of olist
  uint16_t *entry_state;  // state in front of the first symbol of every segment
};

template <class M>
__device__ __forceinline__ unsigned seg_ctx_of(const uint32_t *__restrict__ base, unsigned v) {
  unsigned lo = 0, hi = M::B - 1;  // last context c with base[c] <= v
  while (lo < hi) {
    const unsigned mid = lo + ((hi - lo + 1) >> 1);
    if (base[mid] <= v) lo = mid; else hi = mid - 1;
  }
  return lo;
}

template <class M>
__global__ void __launch_bounds__(256)
k_seg_scan(const uint8_t *__restrict__ sorted_sym, const uint32_t *__restrict__ arrays,
           const unsigned long long *__restrict__ reset_mask, const uint32_t *__restrict__ logs, unsigned S,
           SegArrays sa) {
  constexpr unsigned B = M::B;
  const uint32_t *ctx_count = arrays, *ctx_start = arrays + B, *seg_base = ctx_start + B + 1;
  const unsigned seg = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (seg >= seg_base[B]) return;  // the grid is an upper bound
  const unsigned lane = fq_lane();
  const unsigned c = seg_ctx_of<M>(seg_base, seg), k = seg - seg_base[c];
  const unsigned n = ctx_count[c], begin = k * S, end = min(n, begin + S);
  const unsigned long long mask = reset_mask[c];
  const uint8_t *sym = sorted_sym + ctx_start[c];
  unsigned found = SEG_NONE;
  if (mask != 0ull) {
    for (unsigned b0 = begin; b0 < end; b0 += 1024) {
      const unsigned p = b0 + 16 * lane;
      unsigned hit = 16;
      if (p < end) {  // the run is padded to 16 bytes: whole-group loads stay inside it
        const uint4 v = *reinterpret_cast<const uint4 *>(sym + p);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 15; j >= 0; j--) {
          const unsigned s = (w[j >> 2] >> (8 * (j & 3))) & (unsigned)(M::A - 1);
          if (p + j < end && ((mask >> s) & 1ull)) hit = (unsigned)j;
        }
      }
      const unsigned long long any = __ballot(hit < 16);
      if (any) {
        const unsigned l0 = (unsigned)__ffsll((long long)any) - 1u;
        found = (b0 - begin) + 16 * l0 + (unsigned)__shfl((int)hit, (int)l0);
        break;
      }
    }
  }
  if (lane == 0) {
    sa.first_reset[seg] = found;
    if (k == 0) sa.entry_state[seg] = (uint16_t)(1u << logs[c]);  // FSE_initCState
    const unsigned nseg = seg_base[c + 1] - seg_base[c];
    unsigned slot = SEG_NONE;
    if (found == SEG_NONE && k + 1 < nseg) {
      slot = atomicAdd(sa.n_opaque, 1u);
      sa.olist[slot] = seg | (k == 0 ? 0x80000000u : 0u);
    }
    sa.fidx[seg] = slot;
  }
}

// symbols [i, end) of a context's run walked from state x: packed (nb, bits) into out, 16
// symbols per 16-byte load and two 16-byte stores per aligned group; returns the final state
template <class M>
__device__ __forceinline__ unsigned seg_walk_range(const LdsCTable &t, const uint8_t *__restrict__ sym,
                                                   uint16_t *__restrict__ out, unsigned i, unsigned end, unsigned x) {
  while (i < end && (i & 15u)) {
    out[i] = (uint16_t)chain_step(t, x, sym[i] & (unsigned)(M::A - 1));
    i++;
  }
  if (i + 16 <= end) {
    const uint4 *sym16 = reinterpret_cast<const uint4 *>(sym);
    uint4 *out16v = reinterpret_cast<uint4 *>(out);
    uint4 cur = sym16[i >> 4];
    while (i + 16 <= end) {
      const uint4 nxt = i + 32 <= end ? sym16[(i >> 4) + 1] : cur;
      const unsigned w[4] = {cur.x, cur.y, cur.z, cur.w};
      unsigned o[8];
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const unsigned v = chain_step(t, x, (w[j >> 2] >> (8 * (j & 3))) & (unsigned)(M::A - 1));
        if (j & 1) o[j >> 1] |= v << 16; else o[j >> 1] = v;
      }
      out16v[i >> 3] = make_uint4(o[0], o[1], o[2], o[3]);
      out16v[(i >> 3) + 1] = make_uint4(o[4], o[5], o[6], o[7]);
      i += 16;
      cur = nxt;
    }
  }
  while (i < end) {
    out[i] = (uint16_t)chain_step(t, x, sym[i] & (unsigned)(M::A - 1));
    i++;
  }
  return x;
}

// PASS 1: lane = transparent segment, from behind its first reset symbol to its end.
// PASS 2: lane = segment, its head up to and including the first reset symbol (transparent) or
//         all of it (opaque), from the resolved entry state.
template <class M, int PASS>
__global__ void __launch_bounds__(64)
k_seg_walk(const uint8_t *__restrict__ sorted_sym, uint16_t *__restrict__ out16,
           const uint32_t *__restrict__ arrays, const uint32_t *__restrict__ ct,
           const uint32_t *__restrict__ ct_off, uint16_t *__restrict__ final_state, unsigned S,
           SegArrays sa, StreamResult *res) {
  extern __shared__ uint32_t lds[];
  constexpr unsigned B = M::B;
  const uint32_t *ctx_count = arrays, *ctx_start = arrays + B, *seg_base = ctx_start + B + 1,
                 *item_base = seg_base + B + 1;
  const unsigned item = blockIdx.x;
  if (item >= item_base[B]) return;  // the grid is an upper bound
  const unsigned c = seg_ctx_of<M>(item_base, item);
  const unsigned n = ctx_count[c];
  const unsigned nseg = seg_base[c + 1] - seg_base[c];
  const unsigned k = (item - item_base[c]) * 64 + fq_lane();
  const LdsCTable t = stage_ctable<M>(lds, ct + ct_off[c]);
  if (k >= nseg) return;
  const unsigned seg = seg_base[c] + k;
  const uint8_t *sym = sorted_sym + ctx_start[c];
  uint16_t *out = out16 + ctx_start[c];
  const unsigned begin = k * S, end = min(n, begin + S);
  const unsigned fr = sa.first_reset[seg];
  unsigned x, i0, i1;
  if (PASS == 1) {
    if (fr == SEG_NONE) return;
    i0 = begin + fr + 1; i1 = end;
    x = reset_state(t, sym[begin + fr] & (unsigned)(M::A - 1));
  } else {
    i0 = begin; i1 = fr == SEG_NONE ? end : begin + fr + 1;
    x = sa.entry_state[seg];
  }
  x = seg_walk_range<M>(t, sym, out, i0, i1, x);
  if (PASS == 1 && k + 1 < nseg) sa.entry_state[seg + 1] = (uint16_t)x;
  if (k == nseg - 1 && (PASS == 1 || fr == SEG_NONE)) final_state[c] = (uint16_t)x;
  if (PASS == 2 && fq_lane() == 0) atomicMax(&res->refixed, min(S, n));
}

// n classes (states in L.list) stepped through words [w0, w1) of the segment with the CTable
template <class M, int MM>
__device__ __forceinline__ void seg_sets_walk(SetsWaveLds &L, unsigned n, const LdsCTable &t, const uint4 cur,
                                              unsigned w0, unsigned w1) {
  const unsigned lane = fq_lane();
  unsigned y[MM];
#pragma unroll
  for (int j = 0; j < MM; j++) {
    const unsigned i = lane + 64u * j;
    y[j] = L.list[i < n ? i : n - 1];
  }
  for (unsigned w = w0; w < w1; w++) {
    const unsigned word = sets_word(cur, w);
    int dfs[4];
    unsigned dnb[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {  // uniform addresses: LDS broadcasts, ahead of the dependent chain
      const unsigned s = (word >> (8 * i)) & (unsigned)(M::A - 1);
      dfs[i] = (int)t.tt[2 * s];
      dnb[i] = t.tt[2 * s + 1];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
      for (int j = 0; j < MM; j++) {
        const unsigned nb = (y[j] + dnb[i]) >> 16;
        y[j] = t.state_table[(int)(y[j] >> nb) + dfs[i]];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < MM; j++) {
    const unsigned i = lane + 64u * j;
    if (i < n) L.list[i] = (uint16_t)y[j];
  }
  fq_lds_wave_sync();
}

// F of one opaque segment; one wave per workgroup, its own copy of the context's CTable
template <class M, unsigned PER0>
__global__ void __launch_bounds__(64)
k_seg_setfunc(const uint8_t *__restrict__ sorted_sym, const uint32_t *__restrict__ arrays,
              const uint32_t *__restrict__ ct, const uint32_t *__restrict__ ct_off, unsigned S,
              unsigned fstride, SegArrays sa, uint16_t *__restrict__ fbuf) {
  extern __shared__ uint32_t lds[];
  __shared__ SetsWaveLds L;
  constexpr unsigned B = M::B;
  if (blockIdx.x >= *sa.n_opaque) return;  // the grid is an upper bound
  const uint32_t *ctx_start = arrays + B, *seg_base = ctx_start + B + 1;
  const unsigned seg = sa.olist[blockIdx.x] & 0x7FFFFFFFu;
  const unsigned c = seg_ctx_of<M>(seg_base, seg), k = seg - seg_base[c];
  const LdsCTable t = stage_ctable<M>(lds, ct + ct_off[c]);
  const unsigned log = t.log, size = 1u << log, lane = fq_lane();
  const unsigned per = max(size >> 6, 1u), nw = max(size >> 5, 1u);
  const uint4 *gseg = reinterpret_cast<const uint4 *>(sorted_sym + ctx_start[c] + (size_t)k * S);
  const unsigned nblk = S / SETS_BLOCK, w_end = S / 4;

  unsigned x0[PER0];  // level 0: every state; lane l carries states size + l, size + l + 64, ...
#pragma unroll
  for (unsigned j = 0; j < PER0; j++) x0[j] = size + ((lane + 64u * j) & (size - 1));
  unsigned level = 0, n = size, n1 = 0;
  unsigned w = 0, stop = 1;
  uint4 cur = gseg[lane];
  for (unsigned blk = 0; blk < nblk; blk++) {
    const uint4 nxt = blk + 1 < nblk ? gseg[(size_t)(blk + 1) * 64 + lane] : cur;
    const unsigned wb_end = (blk + 1) * (SETS_BLOCK / 4);
    while (w < wb_end) {
      const unsigned w1 = min(stop, wb_end);
      if (level == 0) {
        for (; w < w1; w++) {
          const unsigned word = sets_word(cur, w);
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const unsigned s = (word >> (8 * i)) & (unsigned)(M::A - 1);
            const int dfs = (int)t.tt[2 * s];
            const unsigned dnb = t.tt[2 * s + 1];
#pragma unroll
            for (unsigned j = 0; j < PER0; j++)
              if (j < per) { const unsigned nb = (x0[j] + dnb) >> 16; x0[j] = t.state_table[(int)(x0[j] >> nb) + dfs]; }
          }
        }
      } else {
        switch ((n + 63) / 64) {
          case 1: seg_sets_walk<M, 1>(L, n, t, cur, w, w1); break;
          case 2: seg_sets_walk<M, 2>(L, n, t, cur, w, w1); break;
          case 3: seg_sets_walk<M, 3>(L, n, t, cur, w, w1); break;
          case 4: seg_sets_walk<M, 4>(L, n, t, cur, w, w1); break;
          case 5: seg_sets_walk<M, 5>(L, n, t, cur, w, w1); break;
          case 6: seg_sets_walk<M, 6>(L, n, t, cur, w, w1); break;
          case 7: seg_sets_walk<M, 7>(L, n, t, cur, w, w1); break;
          default: seg_sets_walk<M, 8>(L, n, t, cur, w, w1); break;
        }
        w = w1;
      }
      if (w != stop || w >= w_end) continue;
      stop = stop == 1 ? 4 : stop == 4 ? 12 : stop == 12 ? 32 : stop * 4;
      if (level == 0) {
        sets_clear(L);
#pragma unroll
        for (unsigned j = 0; j < PER0; j++)
          if (j < per) { const unsigned xi = x0[j] - size; atomicOr(&L.bm[xi >> 5], 1u << (xi & 31u)); }
        fq_lds_wave_sync();
        const unsigned nn = sets_count(L, nw);
        if (nn <= SETS_MAX_CLASSES) {
#pragma unroll
          for (unsigned j = 0; j < PER0; j++)
            if (j < per) {
              const unsigned r = sets_rank(L, x0[j] - size);
              L.list[r] = (uint16_t)x0[j];
              x0[j] = r;
            }
          for (unsigned i = lane; i < nn; i += 64) L.m[i] = (uint16_t)i;
          fq_lds_wave_sync();
          level = 1; n = n1 = nn;
        }
      } else if (n > 64) {
        n = sets_merge<false>(L, n, n1, nw, size);
      }
    }
    cur = nxt;
  }
  uint16_t *f = fbuf + (size_t)blockIdx.x * fstride;  // F[entry - size] = exit
#pragma unroll
  for (unsigned j = 0; j < PER0; j++) {
    const unsigned xi = lane + 64u * j;
    if (j < per && xi < size) f[xi] = level == 0 ? (uint16_t)x0[j] : L.list[L.m[x0[j]]];
  }
}

// Entry states behind opaque segments.  Every other entry state is already there: k_seg_scan
// stored the initial state of every chain, k_seg_walk<1> the state behind every transparent
// segment.  A run of opaque segments is a chain x <- F_k[x] of dependent loads (0.5 us each;
// 29 K of them in a row for a block of constant qualities), so it is resolved in three levels
// over the walk kernels' items (64 consecutive segments of a chain):
//  k_seg_compose   one wave per item with an opaque segment: G = composition of the item's
//                  segment functions (a transparent segment contributes a constant), for every
//                  possible entry state of the item
//  k_seg_resolve2  one thread per context: entry state of every item, x <- G_item[x]
//  k_seg_resolve3  one lane per such item: entry state of every segment inside the item
struct ItemArrays {
  uint16_t *g;           // [items][fstride] composed function (items flagged in has_g only)
  uint32_t *has_g;       // [items]
  uint16_t *item_entry;  // [items]
};

template <class M, unsigned PER0>
__global__ void __launch_bounds__(64)
k_seg_compose(const uint32_t *__restrict__ arrays, const uint32_t *__restrict__ logs,
              const uint16_t *__restrict__ fbuf, unsigned fstride, SegArrays sa, ItemArrays ia) {
  constexpr unsigned B = M::B;
  const uint32_t *seg_base = arrays + B + (B + 1), *item_base = seg_base + B + 1;
  const unsigned item = blockIdx.x, lane = fq_lane();
  if (item >= item_base[B]) return;  // the grid is an upper bound
  const unsigned c = seg_ctx_of<M>(item_base, item);
  const unsigned nseg = seg_base[c + 1] - seg_base[c];
  const unsigned k0 = (item - item_base[c]) * 64, n_here = min(64u, nseg - k0);
  const unsigned seg0 = seg_base[c] + k0;
  // lane t looks at segment t of the item: function slot (SEG_NONE: transparent or last of the chain)
  const unsigned slot = lane < n_here ? sa.fidx[seg0 + lane] : SEG_NONE;
  const unsigned long long opaque = __ballot(slot != SEG_NONE);
  if (lane == 0) ia.has_g[item] = opaque != 0ull;
  if (!opaque) return;
  const unsigned exit_state = lane < n_here && slot == SEG_NONE && k0 + lane + 1 < nseg ? sa.entry_state[seg0 + lane + 1] : 0u;
  const unsigned size = 1u << logs[c], per = max(size >> 6, 1u);
  unsigned x[PER0];
#pragma unroll
  for (unsigned j = 0; j < PER0; j++) x[j] = size + ((lane + 64u * j) & (size - 1));
  for (unsigned t = 0; t < n_here; t++) {
    const unsigned sl = (unsigned)__shfl((int)slot, (int)t);
    if (sl == SEG_NONE) {
      if (k0 + t + 1 >= nseg) break;  // last segment of the chain: nothing follows
      const unsigned e = (unsigned)__shfl((int)exit_state, (int)t);
#pragma unroll
      for (unsigned j = 0; j < PER0; j++) x[j] = e;
    } else {
      const uint16_t *f = fbuf + (size_t)sl * fstride;
#pragma unroll
      for (unsigned j = 0; j < PER0; j++)
        if (j < per) x[j] = f[x[j] - size];
    }
  }
  uint16_t *g = ia.g + (size_t)item * fstride;
#pragma unroll
  for (unsigned j = 0; j < PER0; j++) {
    const unsigned xi = lane + 64u * j;
    if (j < per && xi < size) g[xi] = (uint16_t)x[j];
  }
}

template <class M>
__global__ void __launch_bounds__(256)
k_seg_resolve2(const uint32_t *__restrict__ arrays, const uint32_t *__restrict__ logs, unsigned fstride,
               SegArrays sa, ItemArrays ia) {
  constexpr unsigned B = M::B;
  const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= B) return;
  const uint32_t *seg_base = arrays + B + (B + 1), *item_base = seg_base + B + 1;
  const unsigned i0 = item_base[c], ni = item_base[c + 1] - i0;
  const unsigned size = 1u << logs[c];
  unsigned x = size;  // FSE_initCState
  for (unsigned i = 0; i < ni; i++) {
    ia.item_entry[i0 + i] = (uint16_t)x;
    if (i + 1 == ni) break;
    // an item without opaque segments ends behind a transparent one: k_seg_walk<1> left that state
    x = ia.has_g[i0 + i] ? ia.g[(size_t)(i0 + i) * fstride + (x - size)]
                         : sa.entry_state[seg_base[c] + (i + 1) * 64];
  }
}

template <class M>
__global__ void __launch_bounds__(256)
k_seg_resolve3(const uint32_t *__restrict__ arrays, const uint32_t *__restrict__ logs,
               const uint16_t *__restrict__ fbuf, unsigned fstride, SegArrays sa, ItemArrays ia) {
  constexpr unsigned B = M::B;
  const uint32_t *seg_base = arrays + B + (B + 1), *item_base = seg_base + B + 1;
  const unsigned item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= item_base[B] || !ia.has_g[item]) return;
  const unsigned c = seg_ctx_of<M>(item_base, item);
  const unsigned nseg = seg_base[c + 1] - seg_base[c];
  const unsigned k0 = (item - item_base[c]) * 64, n_here = min(64u, nseg - k0);
  const unsigned seg0 = seg_base[c] + k0;
  const unsigned size = 1u << logs[c];
  unsigned x = ia.item_entry[item];
  for (unsigned t = 0; t < n_here; t++) {
    sa.entry_state[seg0 + t] = (uint16_t)x;
    if (k0 + t + 1 >= nseg) break;
    const unsigned sl = sa.fidx[seg0 + t];
    x = sl == SEG_NONE ? (unsigned)sa.entry_state[seg0 + t + 1] : (unsigned)fbuf[(size_t)sl * fstride + (x - size)];
  }
  if (k0 + n_here < nseg) sa.entry_state[seg0 + n_here] = (uint16_t)x;  // first segment of the next item
}

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

// ---- decode index (extension, include/fqgpu.h FQGPU_F_DECODE_INDEX) ------------------------
// Snapshot k sits at encode index e = k * stride (a multiple of both partition tile sizes).
// k_index_meta: bit position (= bit offset of packing tile e / 4096), the bytes in front of
// symbol e - 1 in its record.  k_index_states: the state of every context at that point = the
// state in front of the context's first symbol at or behind e, found by walking from the entry
// state of the segment that holds it (at most one segment); one lane per (context, snapshot),
// the context's CTable in LDS.
template <class M>
__global__ void __launch_bounds__(256)
k_index_meta(const uint8_t *__restrict__ raw, const fqgpu_rec *__restrict__ recs,
             const uint32_t *__restrict__ rec_start, unsigned R, unsigned n_sym, unsigned stride,
             const unsigned long long *__restrict__ tile_bit_base, uint8_t *__restrict__ index) {
  const unsigned n_snap = n_sym ? (n_sym - 1) / stride : 0u;
  const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;  // 0: header, 1 .. n_snap: snapshots
  if (k == 0) {
    FqIndexHeader h;
    h.magic = FQ_INDEX_MAGIC; h.stream = M::STREAM; h.stride = stride; h.n_snap = n_snap;
    h.n_sym = n_sym; h.reserved = 0;
    *reinterpret_cast<FqIndexHeader *>(index) = h;
    return;
  }
  if (k > n_snap) return;
  const unsigned e = k * stride;
  uint8_t *snap = index + sizeof(FqIndexHeader) + (size_t)(k - 1) * (FQ_INDEX_SNAP_HEAD + 2 * (size_t)M::B);
  *reinterpret_cast<unsigned long long *>(snap) = tile_bit_base[e / PACK_TILE];
  // symbol e - 1: record r, position p (encode order walks a record from its last position)
  const unsigned r = fq_locate(rec_start, 0, R - 1, e - 1);
  const fqgpu_rec rec = recs[r];
  const unsigned p = rec.len - 1u - (e - 1u - rec_start[r]);
  const uint8_t *line = raw + (M::STREAM == 0 ? rec.seq_off : rec.qual_off);
  unsigned packed = 0;
  for (unsigned i = 0; i < 4; i++) packed |= (p >= i + 1 ? (unsigned)line[p - 1 - i] : 0xFFu) << (8 * i);
  reinterpret_cast<uint32_t *>(snap)[2] = packed;
  reinterpret_cast<uint32_t *>(snap)[3] = 0;
}

template <class M>
__global__ void __launch_bounds__(64)
k_index_states(const uint8_t *__restrict__ sorted_sym, const uint32_t *__restrict__ arrays,
               const uint32_t *__restrict__ tile_base, unsigned T, unsigned n_sym, unsigned stride,
               const uint32_t *__restrict__ seg_prefix, const uint16_t *__restrict__ entry, int entry_is_xo,
               unsigned S, const uint32_t *__restrict__ ct, const uint32_t *__restrict__ ct_off,
               const uint16_t *__restrict__ final_state, uint8_t *__restrict__ index) {
  extern __shared__ uint32_t lds[];
  constexpr unsigned B = M::B;
  const unsigned c = blockIdx.x;
  const LdsCTable t = stage_ctable<M>(lds, ct + ct_off[c]);
  const unsigned n_snap = n_sym ? (n_sym - 1) / stride : 0u;
  const unsigned k = blockIdx.y * 64 + fq_lane() + 1;
  if (k > n_snap) return;
  const unsigned size = 1u << t.log;
  const unsigned n = arrays[c], run0 = arrays[B + c];
  const unsigned rel = tile_base[(size_t)((k * stride) / T) * B + c] - run0;  // symbols of c in front of e
  unsigned x = size;  // a context without symbols keeps its initial state
  if (n) {
    if (rel >= n) {
      x = final_state[c];
    } else {
      const unsigned seg = rel / S;
      const unsigned ev = entry[seg_prefix[c] + seg];
      x = entry_is_xo ? size + (ev >> 1) : ev;
      const uint8_t *sym = sorted_sym + run0;
      for (unsigned i = seg * S; i < rel; i++) (void)chain_step(t, x, sym[i] & (unsigned)(M::A - 1));
    }
  }
  uint16_t *st = reinterpret_cast<uint16_t *>(index + sizeof(FqIndexHeader) + (size_t)(k - 1) * (FQ_INDEX_SNAP_HEAD + 2 * (size_t)B) +
                                              FQ_INDEX_SNAP_HEAD);
  st[c] = (uint16_t)(x - size);
}

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
           uint32_t *__restrict__ out, const StreamResult *res) {
  constexpr unsigned B = M::B;
  constexpr unsigned NW = (B * 12 + 1 + 31) / 32 + 2;
  __shared__ uint32_t words[NW];
  if (res->overflow) return;
  const uint32_t *ctx_count = arrays;
  for (unsigned i = threadIdx.x; i < NW; i += blockDim.x) words[i] = 0;
  __syncthreads();
  const unsigned long long p0 = res->total_bits;
  const unsigned sh = (unsigned)(p0 & 31ull);
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

// ------------------------------------------------------------------ host orchestration
#define FQ_SPAN_BEGIN(name) fq_timer_span_begin(ctx, name, st)
#define FQ_SPAN_END() fq_timer_span_end(ctx, st)

// timing experiments only (tools/traffic_experiment.py): FQGPU_DEBUG_SKIP = bit mask of kernel groups
// that are NOT launched; their outputs keep the values of the previous encode of the same lane
// FQGPU_DEBUG_NO_SYM_STORE: 1 = both streams, 2 = sequence only, 3 = quality only
static int fq_debug_no_sym(int stream) {
  const char *e = getenv("FQGPU_DEBUG_NO_SYM_STORE");
  const int v = e ? atoi(e) : 0;
  return v == 1 || (v == 2 && stream == 0) || (v == 3 && stream == 1);
}
static unsigned fq_debug_skip() {
  const char *e = getenv("FQGPU_DEBUG_SKIP");
  return e ? (unsigned)strtoul(e, nullptr, 0) : 0u;
}

template <class M>
int encode_stream(fqgpu_ctx *ctx, EncLane &lane, hipStream_t st, fqgpu_dblock *b,
                  const uint32_t *rec_start, uint8_t *out_dev, size_t cap, unsigned flags) {
  EncScratch &sc = lane.enc[M::STREAM];
  const DevTables &tab = ctx->tab[M::STREAM];
  constexpr unsigned B = M::B;
  const unsigned n_sym = (unsigned)b->n_bases;
  const unsigned R = (unsigned)b->n_recs;
  const unsigned T = tile_size<M>();
  // segment length of the generic chain kernels: whole 1024-symbol blocks
  const unsigned S = (unsigned)min(((size_t)ctx->seg_len + SETS_BLOCK - 1) / SETS_BLOCK * SETS_BLOCK, (size_t)1 << 30);
  const unsigned n_tiles = (n_sym + T - 1) / T;
  const unsigned n_groups = (n_tiles + GROUP_TILES - 1) / GROUP_TILES;
  const unsigned n_ptiles = (n_sym + PACK_TILE - 1) / PACK_TILE;
  const size_t padded = (size_t)n_sym + (size_t)CTX_PAD * B + 64;
  const unsigned max_items = (unsigned)((size_t)n_sym / ((size_t)S * 64) + B + 1);
  StreamResult *res = &b->result->s[M::STREAM];
  const bool serial_seq = M::STREAM == 0 && !ctx->seq_generic;

  int rc;
  unsigned dbg_mask = fq_debug_skip();
  if ((dbg_mask & 256u) && M::STREAM == 1) dbg_mask = 0;  // 256: sequence stream only
  if ((dbg_mask & 512u) && M::STREAM == 0) dbg_mask = 0;  // 512: quality stream only
  bool dbg_off = false;
  // keys: ckey u16 | csym u8 (quality), later enc16 u16 over both;  slot_of: u32 -- padded by one batch
  const size_t n_pad = ((size_t)n_sym + SC_BATCH_SEQ + 15) & ~(size_t)15;  // keeps every sub-array 16-byte aligned
  static_assert(TILE_SEQ % SEQ_BATCH == 0 && SEQ_BATCH % PACK_TILE == 0 && TILE_QUAL % PACK_TILE == 0, "a packing tile lies inside one partition tile");
  if ((rc = sc.slot_of.reserve(n_pad * 4))) return rc;
  if ((rc = sc.keys.reserve(n_pad * 3))) return rc;
  if ((rc = sc.sorted_sym.reserve(padded))) return rc;
  if ((rc = sc.out16.reserve(padded * 2))) return rc;
  if ((rc = sc.tile_hist.reserve((size_t)n_tiles * B * 4))) return rc;
  if ((rc = sc.tile_base.reserve((size_t)n_tiles * B * 4))) return rc;
  if ((rc = sc.group_sum.reserve((size_t)n_groups * B * 4))) return rc;
  if ((rc = sc.ctx_arrays.reserve((size_t)(4 * B + 3) * 4))) return rc;
  if ((rc = sc.seg_state.reserve((size_t)B * 2 + (size_t)B * 8))) return rc;
  // sequence chains: segment length of the candidate-set kernels
  unsigned seq_S = ctx->seq_segment ? ctx->seq_segment : 4096u;
  seq_S = (unsigned)min(((size_t)seq_S + SETS_BLOCK - 1) / SETS_BLOCK * SETS_BLOCK, (size_t)1 << 30);
  const unsigned seq_max_segs = n_sym / seq_S + B + 1;
  const unsigned seq_fstride = 1u << tab.max_log;
  // generic chain kernels (quality stream; sequence stream with FQGPU_CHAIN_SEQ_GENERIC)
  const unsigned gen_max_segs = n_sym / S + B + 1;
  const unsigned gen_fstride = 1u << tab.max_log;
  if (!serial_seq) {
    if ((rc = sc.seg_arrays.reserve((size_t)gen_max_segs * 16 + 64 + (size_t)max_items * (2 * gen_fstride + 8) + 64))) return rc;
    if ((rc = sc.seq_fbuf.reserve(((size_t)n_sym / S + 2) * gen_fstride * 2 + 64))) return rc;
  }
  if (serial_seq) {
    if ((rc = sc.seq_plan.reserve((size_t)SEGPLAN_WORDS * 4 + (size_t)seq_max_segs * 2 + 64))) return rc;
    if ((rc = sc.seq_fbuf.reserve((size_t)seq_max_segs * seq_fstride * 2 + 64))) return rc;
  }
  if (serial_seq && (rc = sc.seq_bdesc.reserve((size_t)(n_ptiles + 1) * SeqModel::B * 6 + 64))) return rc;
  if ((rc = sc.tile_bits.reserve((size_t)n_ptiles * 4))) return rc;
  if ((rc = sc.tile_bit_base.reserve((size_t)(n_ptiles + 1) * 8))) return rc;

  uint16_t *ckey = sc.keys.as<uint16_t>();
  uint8_t *csym = reinterpret_cast<uint8_t *>(ckey + n_pad);
  uint16_t *enc16 = ckey;  // the keys are dead after K3
  uint32_t *arrays = sc.ctx_arrays.as<uint32_t>();
  uint16_t *final_state = sc.seg_state.as<uint16_t>();
  const unsigned lds_ct = (1u + (1u << (tab.max_log - 1)) + 2u * M::A) * 4u;
  const char *pfx = M::STREAM ? "qual." : "seq.";
  (void)pfx;

  FQ_SPAN_BEGIN(M::STREAM ? "qual.tile_hist" : "seq.tile_hist");  dbg_off = (dbg_mask & 1u) != 0;
  if (!dbg_off) hipLaunchKernelGGL(k_tile_hist<M>, dim3(n_tiles), dim3(256), 0, st, b->raw, b->recs,
                     rec_start, R, n_sym, T, sc.tile_hist.as<uint32_t>(), ckey, csym, res,
                     getenv("FQGPU_DEBUG_K1") ? atoi(getenv("FQGPU_DEBUG_K1")) : 0);
  FQ_SPAN_END();
  FQ_SPAN_BEGIN(M::STREAM ? "qual.layout" : "seq.layout");  dbg_off = (dbg_mask & 2u) != 0;
  if (!dbg_off) hipLaunchKernelGGL(k_group_sum, dim3((B + 255) / 256, n_groups), dim3(256), 0, st,
                     sc.tile_hist.as<uint32_t>(), n_tiles, B, sc.group_sum.as<uint32_t>());
  if (!dbg_off) hipLaunchKernelGGL(k_ctx_layout, dim3(1), dim3(1024), 0, st, sc.group_sum.as<uint32_t>(), n_groups,
                     B, S, arrays);
  if (!dbg_off) hipLaunchKernelGGL(k_tile_base, dim3((B + 255) / 256, n_groups), dim3(256), 0, st,
                     sc.tile_hist.as<uint32_t>(), sc.group_sum.as<uint32_t>(), arrays + B, n_tiles, B,
                     sc.tile_base.as<uint32_t>());
  FQ_SPAN_END();
  FQ_SPAN_BEGIN(M::STREAM ? "qual.scatter" : "seq.scatter");  dbg_off = (dbg_mask & 4u) != 0;
  SeqBatchDesc bd;
  bd.start = sc.seq_bdesc.as<uint32_t>();
  bd.pre = reinterpret_cast<uint16_t *>(bd.start + (size_t)(n_ptiles + 1) * SeqModel::B);
  uint16_t *lpos16 = reinterpret_cast<uint16_t *>(sc.slot_of.as<uint32_t>());  // the sequence stream has no slot_of
  if (!dbg_off && serial_seq) {
    if (ctx->lds_atomics_ordered)
      hipLaunchKernelGGL(k_scatter_seq<true>, dim3(n_tiles), dim3(64), 0, st, ckey, n_sym, T, sc.tile_base.as<uint32_t>(),
                         sc.sorted_sym.as<uint8_t>(), lpos16, bd, fq_debug_no_sym(0));
    else
      hipLaunchKernelGGL(k_scatter_seq<false>, dim3(n_tiles), dim3(64), 0, st, ckey, n_sym, T, sc.tile_base.as<uint32_t>(),
                         sc.sorted_sym.as<uint8_t>(), lpos16, bd, fq_debug_no_sym(0));
  } else if (!dbg_off) {
    if (ctx->lds_atomics_ordered)
      hipLaunchKernelGGL((k_scatter<M, true>), dim3(n_tiles), dim3(64), 0, st, ckey, csym, n_sym, T,
                         sc.tile_base.as<uint32_t>(), sc.sorted_sym.as<uint8_t>(), sc.slot_of.as<uint32_t>(),
                         fq_debug_no_sym(M::STREAM));
    else
      hipLaunchKernelGGL((k_scatter<M, false>), dim3(n_tiles), dim3(64), 0, st, ckey, csym, n_sym, T,
                         sc.tile_base.as<uint32_t>(), sc.sorted_sym.as<uint8_t>(), sc.slot_of.as<uint32_t>(),
                         fq_debug_no_sym(M::STREAM));
  }
  FQ_SPAN_END();
  FQ_SPAN_BEGIN(M::STREAM ? "qual.scan" : (serial_seq ? "seq.plan" : "seq.scan"));  dbg_off = (dbg_mask & 8u) != 0;
  if (serial_seq) {
    uint32_t *plan = sc.seq_plan.as<uint32_t>();
    uint16_t *entry = reinterpret_cast<uint16_t *>(plan + SEGPLAN_WORDS);
    uint16_t *fbuf = sc.seq_fbuf.as<uint16_t>();
    const unsigned next_stride = 4u << tab.max_log;
    const bool two = tab.next2 != nullptr;  // two-symbol tables exist up to log 11
    const unsigned wpg = two ? SETS_WAVES2 : SETS_WAVES;
    const unsigned max_fitems = seq_max_segs / (wpg * SETS_ROUNDS) + B + 1, max_eitems = seq_max_segs / 64 + B + 1;
    static const bool dbg_skip = getenv("FQGPU_DEBUG_SKIP_SEQ_CHAIN") != nullptr;  // timing experiment only: wrong output
    if (!dbg_off) hipLaunchKernelGGL(k_seq_segplan, dim3(1), dim3(256), 0, st, arrays, seq_S, wpg * SETS_ROUNDS, plan);
    FQ_SPAN_END();
    FQ_SPAN_BEGIN("seq.setfunc");
    if (!dbg_skip) {
      if (dbg_off) {
      } else if (two)
        hipLaunchKernelGGL((k_seq_setfunc<32, true>), dim3(min(max_fitems, ctx->n_cus)), dim3(SETS_WAVES2 * 64),
                           32u << tab.max_log, st, sc.sorted_sym.as<uint8_t>(), arrays, plan, tab.logs, tab.next2,
                           4 * next_stride, seq_S, seq_fstride, fbuf, plan + 4 * (B + 1));
      else
        hipLaunchKernelGGL((k_seq_setfunc<64, false>), dim3(min(max_fitems, 2 * ctx->n_cus)), dim3(SETS_WAVES * 64),
                           8u << tab.max_log, st, sc.sorted_sym.as<uint8_t>(), arrays, plan, tab.logs, tab.next1,
                           next_stride, seq_S, seq_fstride, fbuf, plan + 4 * (B + 1));
      FQ_SPAN_END();
      FQ_SPAN_BEGIN("seq.resolve");  dbg_off = (dbg_mask & 8u) != 0;
      if (!dbg_off) hipLaunchKernelGGL(k_seq_resolve, dim3(1), dim3(256), 0, st, plan, fbuf, seq_fstride, entry);
      FQ_SPAN_END();
      FQ_SPAN_BEGIN("seq.chains");  dbg_off = (dbg_mask & 8u) != 0;
      if (!dbg_off) hipLaunchKernelGGL(k_seq_emit, dim3(max_eitems), dim3(64), 8u << tab.max_log, st, sc.sorted_sym.as<uint8_t>(),
                         sc.out16.as<uint16_t>(), arrays, plan, tab.ct, tab.ct_off, tab.next1, next_stride, seq_S,
                         entry, final_state, res);
    }
  } else {
    SegArrays sa;
    sa.first_reset = sc.seg_arrays.as<uint32_t>();
    sa.fidx = sa.first_reset + gen_max_segs;
    sa.olist = sa.fidx + gen_max_segs;
    sa.n_opaque = sa.olist + gen_max_segs;
    sa.entry_state = reinterpret_cast<uint16_t *>(sa.n_opaque + 4);
    uint16_t *fbuf = sc.seq_fbuf.as<uint16_t>();
    FQ_HIP(hipMemsetAsync(sa.n_opaque, 0, 4, st));
    if (!dbg_off) hipLaunchKernelGGL(k_seg_scan<M>, dim3((gen_max_segs + 3) / 4), dim3(256), 0, st, sc.sorted_sym.as<uint8_t>(),
                       arrays, tab.reset_mask, tab.logs, S, sa);
    FQ_SPAN_END();
    FQ_SPAN_BEGIN(M::STREAM ? "qual.walk1" : "seq.walk1");  dbg_off = (dbg_mask & 8u) != 0;
    if (!dbg_off) hipLaunchKernelGGL((k_seg_walk<M, 1>), dim3(max_items), dim3(64), lds_ct, st, sc.sorted_sym.as<uint8_t>(),
                       sc.out16.as<uint16_t>(), arrays, tab.ct, tab.ct_off, final_state, S, sa, res);
    FQ_SPAN_END();
    FQ_SPAN_BEGIN(M::STREAM ? "qual.setfunc" : "seq.setfunc");  dbg_off = (dbg_mask & 8u) != 0;
    if (dbg_off) {
    } else if (tab.max_log <= 11)
      hipLaunchKernelGGL((k_seg_setfunc<M, 32>), dim3(n_sym / S + 1), dim3(64), lds_ct, st, sc.sorted_sym.as<uint8_t>(),
                         arrays, tab.ct, tab.ct_off, S, gen_fstride, sa, fbuf);
    else
      hipLaunchKernelGGL((k_seg_setfunc<M, 64>), dim3(n_sym / S + 1), dim3(64), lds_ct, st, sc.sorted_sym.as<uint8_t>(),
                         arrays, tab.ct, tab.ct_off, S, gen_fstride, sa, fbuf);
    FQ_SPAN_END();
    FQ_SPAN_BEGIN(M::STREAM ? "qual.resolve" : "seq.resolve");  dbg_off = (dbg_mask & 8u) != 0;
    if (!dbg_off) {
      ItemArrays ia;
      ia.has_g = reinterpret_cast<uint32_t *>(sc.seg_arrays.as<uint8_t>() + (((size_t)gen_max_segs * 16 + 64 + 15) & ~(size_t)15));
      ia.item_entry = reinterpret_cast<uint16_t *>(ia.has_g + max_items);
      ia.g = ia.item_entry + ((max_items + 7) & ~7u);
      if (tab.max_log <= 11)
        hipLaunchKernelGGL((k_seg_compose<M, 32>), dim3(max_items), dim3(64), 0, st, arrays, tab.logs, fbuf, gen_fstride, sa, ia);
      else
        hipLaunchKernelGGL((k_seg_compose<M, 64>), dim3(max_items), dim3(64), 0, st, arrays, tab.logs, fbuf, gen_fstride, sa, ia);
      hipLaunchKernelGGL(k_seg_resolve2<M>, dim3((B + 255) / 256), dim3(256), 0, st, arrays, tab.logs, gen_fstride, sa, ia);
      hipLaunchKernelGGL(k_seg_resolve3<M>, dim3((max_items + 255) / 256), dim3(256), 0, st, arrays, tab.logs, fbuf,
                         gen_fstride, sa, ia);
    }
    FQ_SPAN_END();
    FQ_SPAN_BEGIN(M::STREAM ? "qual.walk2" : "seq.walk2");  dbg_off = (dbg_mask & 8u) != 0;
    if (!dbg_off) hipLaunchKernelGGL((k_seg_walk<M, 2>), dim3(max_items), dim3(64), lds_ct, st, sc.sorted_sym.as<uint8_t>(),
                       sc.out16.as<uint16_t>(), arrays, tab.ct, tab.ct_off, final_state, S, sa, res);
  }
  FQ_SPAN_END();
  FQ_SPAN_BEGIN(M::STREAM ? "qual.bitcount" : "seq.bitcount");  dbg_off = (dbg_mask & 16u) != 0;
  if (!dbg_off && serial_seq)
    hipLaunchKernelGGL(k_bitcount_seq, dim3((n_sym + SEQ_BATCH - 1) / SEQ_BATCH), dim3(PACK_THREADS), 0, st, lpos16, bd, sc.out16.as<uint16_t>(), n_sym,
                       sc.tile_bits.as<uint32_t>(), enc16);
  else if (!dbg_off)
    hipLaunchKernelGGL(k_bitcount, dim3(n_ptiles), dim3(PACK_THREADS), 0, st, sc.slot_of.as<uint32_t>(),
                       sc.out16.as<uint16_t>(), n_sym, sc.tile_bits.as<uint32_t>(), enc16);
  FQ_SPAN_END();
  FQ_SPAN_BEGIN(M::STREAM ? "qual.bitscan" : "seq.bitscan");  dbg_off = (dbg_mask & 32u) != 0;
  if (!dbg_off) hipLaunchKernelGGL(k_bitscan, dim3(1), dim3(1024), 0, st, sc.tile_bits.as<uint32_t>(), n_ptiles,
                     sc.tile_bit_base.as<unsigned long long>(), tab.log_prefix, B, (unsigned long long)cap,
                     reinterpret_cast<uint32_t *>(out_dev), res);
  FQ_SPAN_END();
  FQ_SPAN_BEGIN(M::STREAM ? "qual.pack" : "seq.pack");  dbg_off = (dbg_mask & 64u) != 0;
  if (!dbg_off) hipLaunchKernelGGL(k_pack, dim3(n_ptiles), dim3(PACK_THREADS), 0, st, enc16, n_sym,
                     sc.tile_bit_base.as<unsigned long long>(), reinterpret_cast<uint32_t *>(out_dev), res);
  FQ_SPAN_END();
  FQ_SPAN_BEGIN(M::STREAM ? "qual.epilogue" : "seq.epilogue");  dbg_off = (dbg_mask & 128u) != 0;
  if (!dbg_off) hipLaunchKernelGGL(k_epilogue<M>, dim3(1), dim3(256), 0, st, arrays, final_state, tab.logs,
                     tab.log_prefix, reinterpret_cast<uint32_t *>(out_dev), res);
  FQ_SPAN_END();
  b->index_bytes[M::STREAM] = 0;
  if ((flags & FQGPU_F_DECODE_INDEX) && !dbg_mask) {
    const unsigned stride = ctx->index_stride;
    const unsigned n_snap = n_sym ? (n_sym - 1) / stride : 0u;
    const size_t bytes = sizeof(FqIndexHeader) + (size_t)n_snap * fq_index_snap_bytes(B);
    if (bytes > b->index_cap[M::STREAM]) {
      if (b->index[M::STREAM]) FQ_HIP(hipFree(b->index[M::STREAM]));
      b->index[M::STREAM] = fq_dev_alloc<uint8_t>(bytes + 64);
      b->index_cap[M::STREAM] = b->index[M::STREAM] ? bytes : 0;
      if (!b->index[M::STREAM]) return FQGPU_E_NOMEM;
    }
    FQ_SPAN_BEGIN(M::STREAM ? "qual.index" : "seq.index");
    hipLaunchKernelGGL(k_index_meta<M>, dim3(n_snap / 256 + 1), dim3(256), 0, st, b->raw, b->recs, rec_start, R, n_sym,
                       stride, sc.tile_bit_base.as<unsigned long long>(), b->index[M::STREAM]);
    if (n_snap) {
      const uint32_t *seg_prefix = serial_seq ? sc.seq_plan.as<uint32_t>() + 2 * (B + 1) : arrays + B + (B + 1);
      const uint16_t *entry = serial_seq ? reinterpret_cast<const uint16_t *>(sc.seq_plan.as<uint32_t>() + SEGPLAN_WORDS)
                                         : reinterpret_cast<const uint16_t *>(sc.seg_arrays.as<uint32_t>() + 3 * (size_t)gen_max_segs + 4);
      hipLaunchKernelGGL(k_index_states<M>, dim3(B, (n_snap + 63) / 64), dim3(64), lds_ct, st, sc.sorted_sym.as<uint8_t>(),
                         arrays, sc.tile_base.as<uint32_t>(), T, n_sym, stride, seg_prefix, entry, serial_seq ? 1 : 0,
                         serial_seq ? seq_S : S, tab.ct, tab.ct_off, final_state, b->index[M::STREAM]);
    }
    FQ_SPAN_END();
    b->index_bytes[M::STREAM] = bytes;
  }
  FQ_HIP(hipGetLastError());
  return FQGPU_OK;
}

}  // namespace

namespace {
// Verifies on the device that same-address LDS atomics of one wave instruction are applied in
// lane order (what k_scatter<ORDERED> relies on): random keys, some lanes idle, plain and packed
// counters, compared with the rank counted by shuffles.
__global__ void __launch_bounds__(256)
k_probe_lds_atomic_order(unsigned iters, unsigned *__restrict__ bad) {
  __shared__ unsigned cnt[4][1024];
  const unsigned wave = threadIdx.x >> 6, lane = fq_lane();
  unsigned s = (blockIdx.x * blockDim.x + threadIdx.x) * 40503u + 12345u;
  unsigned nbad = 0;
  for (unsigned it = 0; it < iters; it++) {
    for (unsigned c = lane; c < 1024; c += 64) cnt[wave][c] = 0;
    fq_lds_wave_sync();
    s ^= s << 13; s ^= s >> 17; s ^= s << 5;
    const unsigned range = 1u << ((it % 6) * 2);  // 1, 4, 16, ... 1024 distinct keys
    const unsigned key = (s >> 8) % range;
    const bool active = (s & 15u) != 0;
    const bool packed = (it & 1u) != 0;
    unsigned got = 0;
    if (active) {
      if (packed) got = (atomicAdd(&cnt[wave][key >> 1], 1u << (16 * (key & 1u))) >> (16 * (key & 1u))) & 0xFFFFu;
      else got = atomicAdd(&cnt[wave][key], 1u);
    }
    unsigned ref = 0;
    for (int l = 0; l < 64; l++) {
      const unsigned k2 = (unsigned)__shfl((int)key, l);
      const int a2 = __shfl((int)active, l);
      if ((unsigned)l < lane && a2 && k2 == key) ref++;
    }
    if (active && got != ref) nbad++;
    fq_lds_wave_sync();
  }
  if (nbad) atomicAdd(bad, nbad);
}

}  // namespace

int fq_probe_lds_atomic_order(hipStream_t st, bool *ordered) {
  unsigned *bad = fq_dev_alloc<unsigned>(1);
  if (!bad) return FQGPU_E_NOMEM;
  unsigned h = 1;
  hipError_t e = hipMemsetAsync(bad, 0, 4, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_probe_lds_atomic_order, dim3(512), dim3(256), 0, st, 96u, bad);
    e = hipMemcpyAsync(&h, bad, 4, hipMemcpyDeviceToHost, st);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(bad);
  if (e != hipSuccess) return FQGPU_E_HIP;
  *ordered = h == 0;
  return FQGPU_OK;
}

// One block = one encode lane: two HIP streams (sequence pipeline, quality pipeline) forked
// after the record-level kernels and joined before the N-position pass.  Blocks handed to
// different lanes overlap on the device: the serial sequence chains of one block hide behind
// the bandwidth-bound passes of the others.
int fq_encode_launch(fqgpu_ctx *ctx, fqgpu_dblock *b, unsigned flags) {
  const unsigned R = (unsigned)b->n_recs;
  if (R == 0 || b->n_bases == 0) return FQGPU_E_ARG;
  EncLane *lp = fq_next_lane(ctx);
  if (!lp) return FQGPU_E_NOMEM;
  EncLane &lane = *lp;
  hipStream_t st = lane.st_seq;
  int rc;
  if ((rc = lane.rec_start.reserve((size_t)(R + 1) * 4))) return rc;
  if ((rc = lane.n_cnt32.reserve((size_t)R * 4 * 2))) return rc;  // n_cnt32 | lens32
  if ((rc = lane.n_off.reserve((size_t)(R + 1) * 4))) return rc;
  uint32_t *n_cnt32 = lane.n_cnt32.as<uint32_t>();
  uint32_t *lens32 = n_cnt32 + R;
  uint32_t *rec_start = lane.rec_start.as<uint32_t>();

  FQ_HIP(hipMemsetAsync(b->result, 0, sizeof(BlockResult), st));
  const unsigned rec_blocks = (unsigned)min((size_t)(R + 3) / 4, (size_t)8192);
  FQ_SPAN_BEGIN("records");
  hipLaunchKernelGGL(k_readlens_ncount, dim3(rec_blocks), dim3(256), 0, st, b->raw, b->recs, R,
                     b->readlens, b->n_count, n_cnt32, lens32);
  if ((rc = fq_scan2_u32_to_u32(st, lens32, n_cnt32, R, rec_start, lane.n_off.as<uint32_t>(), lane.scan_tmp))) return rc;
  hipLaunchKernelGGL(k_store_npos_len, dim3(1), dim3(1), 0, st, lane.n_off.as<uint32_t>(), R, b->result);
  FQ_SPAN_END();

  FQ_HIP(hipEventRecord(lane.ev_fork, lane.st_seq));
  FQ_HIP(hipStreamWaitEvent(lane.st_qual, lane.ev_fork, 0));
  // timing experiments only (wrong output): one stream at a time
  static const bool dbg_no_qual = getenv("FQGPU_DEBUG_SKIP_QUAL") != nullptr, dbg_no_seq = getenv("FQGPU_DEBUG_SKIP_SEQ") != nullptr;
  if (!dbg_no_qual && (rc = encode_stream<QualModel>(ctx, lane, lane.st_qual, b, rec_start, b->qual, b->qual_cap, flags))) return rc;
  if (!dbg_no_seq && (rc = encode_stream<SeqModel>(ctx, lane, lane.st_seq, b, rec_start, b->seq, b->seq_cap, flags))) return rc;
  FQ_HIP(hipEventRecord(lane.ev_join, lane.st_qual));
  FQ_HIP(hipStreamWaitEvent(lane.st_seq, lane.ev_join, 0));

  // after both streams: the optional in-place N -> A must not race with their reads of raw
  FQ_SPAN_BEGIN("npos");
  hipLaunchKernelGGL(k_npos, dim3(rec_blocks), dim3(256), 0, st, b->raw, b->recs, R,
                     lane.n_off.as<uint32_t>(), b->n_pos, (flags & FQGPU_F_WRITE_BACK_N) ? 1 : 0);
  FQ_SPAN_END();
  FQ_HIP(hipGetLastError());
  return FQGPU_OK;
}
