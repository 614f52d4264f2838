// Part of encode.hip (included there, inside its anonymous namespace): record-level kernels, the encode-order symbol walker, K1 (tile histograms + keys), K2 (layout).

// ------------------------------------------------------------------ record-level kernels

// readlens + N count per record (replaceAndEncodeNs, src/fse_sequence.cpp:35-51, first half).
// Sixteen lanes per record, 16 bases per lane and load: four records per wave and round, one
// round trip for a read of up to 256 bases.
struct __attribute__((packed)) FqBytes16 { uint32_t w[4]; };  // 16 bytes at any address
__global__ void __launch_bounds__(256)
k_readlens_ncount(const uint8_t *__restrict__ raw, const fqgpu_rec *__restrict__ recs, unsigned R,
                  uint16_t *__restrict__ readlens, uint16_t *__restrict__ n_count,
                  uint32_t *__restrict__ n_cnt32, uint32_t *__restrict__ lens32, BlockResult *res) {
  if (res != nullptr && blockIdx.x == 0 && threadIdx.x < sizeof(BlockResult) / 4) reinterpret_cast<uint32_t *>(res)[threadIdx.x] = 0u;  // the block's result starts clean (first kernel of the encode: no memset in front of it)
  const unsigned waves = (gridDim.x * blockDim.x) >> 6;
  const unsigned lane = fq_lane(), sub = lane >> 4, l = lane & 15u;
  for (unsigned r0 = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 4; r0 < R; r0 += waves * 4) {
    const unsigned r = r0 + sub;
    const bool on = r < R;
    const fqgpu_rec rec = recs[on ? r : R - 1];
    const unsigned len = on ? rec.len : 0u;
    const uint8_t *s = raw + rec.seq_off;
    unsigned cnt = 0;
    for (unsigned off = l * 16; off < len; off += 256) {  // (the block has 64 spare bytes behind raw)
      const FqBytes16 v = *reinterpret_cast<const FqBytes16 *>(s + off);
      const unsigned live = min(16u, len - off);
#pragma unroll
      for (unsigned i = 0; i < 16; i++)
        cnt += (i < live && ((v.w[i >> 2] >> (8 * (i & 3))) & 0xFFu) == 'N') ? 1u : 0u;
    }
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);  // stays inside the 16 lanes
    if (l == 0 && on) {
      readlens[r] = (uint16_t)len;
      n_count[r] = (uint16_t)cnt;
      n_cnt32[r] = cnt;
      lens32[r] = len;
    }
  }
}

// Tile-sorted path: the read lengths alone (record table only: the raw block is not touched); the N's are
// counted by K1 (k_tile_hist2), which sees every base anyway, into n_cnt32 -- zeroed here.
__global__ void __launch_bounds__(256)
k_readlens(const fqgpu_rec *__restrict__ recs, unsigned R, uint16_t *__restrict__ readlens,
           uint32_t *__restrict__ n_cnt32, uint32_t *__restrict__ lens32, BlockResult *res) {
  if (res != nullptr && blockIdx.x == 0 && threadIdx.x < sizeof(BlockResult) / 4) reinterpret_cast<uint32_t *>(res)[threadIdx.x] = 0u;  // the block's result starts clean (first kernel of the encode: no memset in front of it)
  for (unsigned r = blockIdx.x * blockDim.x + threadIdx.x; r < R; r += gridDim.x * blockDim.x) {
    const unsigned len = recs[r].len;
    readlens[r] = (uint16_t)len;
    lens32[r] = len;
    n_cnt32[r] = 0;
  }
}
// ------------------------------------------------------------------ record-level scans in ONE launch each
// Round 3: k_readlens + three scan kernels in front of K1 and k_ncount16 + three scan kernels + k_store_npos_len in front
// of k_npos -- nine launches of a few microseconds of work each on every block's critical path.  Here one kernel per
// scan: a workgroup takes 2048 records (by ticket, so that it only ever waits for workgroups that already run), scans
// them, publishes its total in status[chunk] and finds its base by a decoupled look-back, as K6 does for the tiles' bit
// counts: value, flag and EPOCH travel in one 8-byte word (relaxed agent-scope atomics, no fence).  The epoch -- a launch
// counter the host keeps per lane -- makes words of earlier launches read as "not there yet", so nothing is zeroed in
// between; tickets are counted on from launch to launch (the host passes the count before this launch).
//   MODE 0  lengths from the record table: readlens (u16), rec_start (u32 [R + 1], exclusive), n_cnt32 zeroed for K1,
//           the block's result zeroed (first kernel of an encode)
//   MODE 1  N counts left by K1: n_count (u16), n_off (u32 [R + 1]), result->n_pos_len
constexpr unsigned RSCAN_THREADS = 256, RSCAN_PER = 8, RSCAN_CHUNK = RSCAN_THREADS * RSCAN_PER;
// exclusive scan of one value per thread over a workgroup of RSCAN_THREADS; *total = sum (all threads call)
__device__ __forceinline__ unsigned ts_block_scan_r(unsigned v, unsigned *wsum, unsigned *total) {
  unsigned inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned o = __shfl_up(inc, d);
    if (fq_lane() >= (unsigned)d) inc += o;
  }
  const unsigned w = threadIdx.x >> 6;
  if (fq_lane() == 63) wsum[w] = inc;
  __syncthreads();
  unsigned base = 0, tot = 0;
#pragma unroll
  for (unsigned i = 0; i < 4; i++) {
    const unsigned s = wsum[i];
    if (i < w) base += s;
    tot += s;
  }
  *total = tot;
  return base + inc - v;
}


constexpr unsigned long long RSCAN_FLAG_AGG = 1ull << 38, RSCAN_FLAG_INCL = 2ull << 38, RSCAN_VAL_MASK = (1ull << 38) - 1ull;

template <int MODE>
__global__ void __launch_bounds__(RSCAN_THREADS)
k_record_scan(const fqgpu_rec *__restrict__ recs, uint32_t *__restrict__ n_cnt32, unsigned R, uint16_t *__restrict__ out16,
              uint32_t *__restrict__ out_prefix, unsigned long long *__restrict__ status, unsigned *__restrict__ ticket,
              unsigned ticket_base, unsigned epoch, BlockResult *res) {
  __shared__ unsigned wsum[RSCAN_THREADS / 64], s_chunk;
  __shared__ unsigned long long s_base;
  const unsigned tid = threadIdx.x, lane = fq_lane(), wave = tid >> 6;
  if (tid == 0) s_chunk = atomicAdd(ticket, 1u) - ticket_base;
  __syncthreads();
  const unsigned chunk = s_chunk, n_chunks = gridDim.x;
  if (MODE == 0 && chunk == 0 && tid < sizeof(BlockResult) / 4) reinterpret_cast<uint32_t *>(res)[tid] = 0u;  // the block's result starts clean
  const unsigned r0 = chunk * RSCAN_CHUNK + tid * RSCAN_PER;
  unsigned v[RSCAN_PER], sum = 0;
#pragma unroll
  for (unsigned i = 0; i < RSCAN_PER; i++) {
    const unsigned r = r0 + i;
    v[i] = r < R ? (MODE == 0 ? recs[r].len : n_cnt32[r]) : 0u;
    sum += v[i];
  }
  unsigned tot;
  unsigned off = ts_block_scan_r(sum, wsum, &tot);
  if (wave == 0) {
    const unsigned long long ep = (unsigned long long)epoch << 40;
    if (lane == 0)
      __hip_atomic_store(&status[chunk], ep | (chunk == 0 ? RSCAN_FLAG_INCL : RSCAN_FLAG_AGG) | (unsigned long long)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long excl = 0;
    if (chunk > 0) {
      int first = (int)chunk - 1;  // lane l looks at chunk first - l
      for (;;) {
        const int idx = first - (int)lane;
        unsigned long long st = ep | RSCAN_FLAG_AGG;  // (in front of chunk 0: empty aggregates)
        if (idx >= 0) {
          for (;;) {
            st = __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((st >> 40) == (unsigned long long)epoch && ((st >> 38) & 3ull) != 0ull) break;
            __builtin_amdgcn_s_sleep(1);
          }
        }
        const unsigned long long incl_mask = __ballot(((st >> 38) & 3ull) == 2ull);
        const unsigned stop = incl_mask ? (unsigned)__ffsll((long long)incl_mask) - 1u : 64u;  // nearest inclusive total
        unsigned long long x = lane <= stop ? (st & RSCAN_VAL_MASK) : 0ull;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) x += __shfl_xor(x, d);
        excl += x;
        if (incl_mask || first < 64) break;
        first -= 64;
      }
      if (lane == 0)
        __hip_atomic_store(&status[chunk], ep | RSCAN_FLAG_INCL | (excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) s_base = excl;
  }
  __syncthreads();
  off += (unsigned)s_base;
#pragma unroll
  for (unsigned i = 0; i < RSCAN_PER; i++) {
    const unsigned r = r0 + i;
    if (r < R) {
      out_prefix[r] = off;
      out16[r] = (uint16_t)v[i];
      if (MODE == 0) n_cnt32[r] = 0u;
    }
    off += v[i];
  }
  if (chunk == n_chunks - 1 && tid == RSCAN_THREADS - 1) {  // (the last thread of the last chunk holds the grand total)
    out_prefix[R] = off;
    if (MODE == 1) res->n_pos_len = off;
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
    unsigned ridx;
    locate(eb, e_end, e, valid, rec, p, ridx);
  }
  // ... and the record's number
  __device__ __forceinline__ void locate(unsigned eb, unsigned e_end, unsigned e, bool valid,
                                         fqgpu_rec &rec, unsigned &p, unsigned &ridx) {
    const unsigned chunk_end = min(eb + 64u, e_end);
    unsigned rr = r;
    rec.seq_off = rec.qual_off = rec.len = 0;
    p = 0;
    ridx = 0;
    for (;;) {
      const unsigned k = rr - r0;
      const unsigned rs = __builtin_amdgcn_readfirstlane(cache->start[k]),
                     rn = __builtin_amdgcn_readfirstlane(cache->start[k + 1]);
      if (valid && e >= rs && e < rn) { rec = cache->rec[k]; p = rec.len - 1u - (e - rs); ridx = rr; }
      if (rn > chunk_end) break;            // record rr continues into the next chunk
      rr++;
      if (rn == chunk_end) break;           // next chunk starts exactly at record rr
    }
    r = rr;
  }
};

__device__ __forceinline__ unsigned sets_incl_scan_k1(unsigned v) {  // inclusive scan over the wave
  const unsigned lane = fq_lane();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned o = __shfl_up(v, d);
    if (lane >= (unsigned)d) v += o;
  }
  return v;
}

// ------------------------------------------------------------------ K1: per-tile context histogram
// Also leaves the key of every symbol in encode order, so that the partition pass is a plain
// prefetchable linear scan: ckey[e] = ctx | sym << 8 (sequence: 10 bits) or ctx (quality: 13
// bits, the symbol goes to csym[e]).  Two or three bytes per symbol instead of four: the key
// stores alone were 2.7 of the 21 ms step (tools/traffic_experiment.py).
// HT: type of a histogram row's entries -- uint16_t on the tile-sorted path (a tile has at most 32768 symbols), else uint32_t
template <class M, class HT>
__global__ void __launch_bounds__(256)
k_tile_hist(const uint8_t *__restrict__ raw, const fqgpu_rec *__restrict__ recs,
            const uint32_t *__restrict__ rec_start, unsigned R, unsigned n_sym, unsigned T,
            HT *__restrict__ tile_hist, uint16_t *__restrict__ ckey, uint8_t *__restrict__ csym,
            StreamResult *res, int dbg) {
  // Quality stream: 16-bit counters, two per word -- 16 KB of LDS instead of 32, so that a CU takes
  // seven of these workgroups, or four and still has room for another kernel's.  A tile has up to
  // 65536 symbols: a counter overflows only if ALL of them share one context (constant Phred 0:
  // every context is calcContext(0, 0, 0)); then the counter of the tile's first symbol reads 0,
  // which no other tile can produce, and the histogram is rebuilt from that (see the write-out).
  constexpr bool PACKED = M::B > 1024;
  constexpr unsigned HWORDS = PACKED ? M::B / 2 : M::B;
  __shared__ uint32_t hist[HWORDS];
  __shared__ unsigned s_first_ctx;
  __shared__ RecCache rcache[4];  // one per wave
  __shared__ uint8_t code_lut[256], sym_lut[256];  // fq_base_code / fq_base_sym of every byte value (sequence stream)
  code_lut[threadIdx.x & 255u] = (uint8_t)fq_base_code(threadIdx.x & 255u);
  sym_lut[threadIdx.x & 255u] = (uint8_t)fq_base_sym(threadIdx.x & 255u);
  const unsigned tile = blockIdx.x;
  const unsigned e0 = tile * T;
  const unsigned e1 = min(e0 + T, n_sym);
  for (unsigned c = threadIdx.x; c < HWORDS; c += blockDim.x) hist[c] = 0;
  if (threadIdx.x == 0) s_first_ctx = 0;
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
        fq_ctx_from_bytes<M>(buf[slot], bp[slot], ctx, sym, code_lut, sym_lut);
        bad |= sym >= (unsigned)M::A;
        if (!(dbg & 2)) {
          if (M::STREAM == 0) ckey[e] = (uint16_t)(ctx | ((sym & 3u) << 8));
          else { if (!(dbg & 8)) ckey[e] = (uint16_t)ctx; if (!(dbg & 4)) csym[e] = (uint8_t)(sym & 63u); }
        }
        if (!(dbg & 1)) {
          if (PACKED) {
            atomicAdd(&hist[ctx >> 1], 1u << (16u * (ctx & 1u)));
            if (e == e0) s_first_ctx = ctx;
          } else {
            atomicAdd(&hist[ctx], 1u);
          }
        }
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
  if (PACKED) {
    const unsigned c0 = s_first_ctx;
    const bool wrapped = e1 > e0 && ((hist[c0 >> 1] >> (16u * (c0 & 1u))) & 0xFFFFu) == 0u;  // 65536 symbols, all in c0
    for (unsigned c = threadIdx.x; c < (unsigned)M::B; c += blockDim.x)
      tile_hist[(size_t)tile * M::B + c] = (HT)(wrapped ? (c == c0 ? e1 - e0 : 0u) : (hist[c >> 1] >> (16u * (c & 1u))) & 0xFFFFu);
  } else {
    for (unsigned c = threadIdx.x; c < (unsigned)M::B; c += blockDim.x)
      tile_hist[(size_t)tile * M::B + c] = (HT)hist[c];
  }
}

// K1 for BOTH streams in one pass (tile-sorted path: the two streams share the tile geometry), FOUR consecutive
// positions of one read per lane.  The encoder is bound by the instructions it issues and by LDS cycles (DESIGN.md
// section 8), and round 3's K1 spent 93 vector instructions and 11 LDS operations per 64 symbols on finding every
// symbol's record, loading a window per symbol and stream and turning five bytes into codes through an LDS table.
// Here a lane owns a QUAD: encode indices rs + 4 j .. rs + 4 j + 3 of ONE record (rs = the record's first encode index),
// i.e. positions p_hi = L - 1 - 4 j down to p_hi - 3 (encode order runs backwards through a read,
// src/fse_sequence.cpp:76-77,101; src/fse_quality.cpp:7,19).  Quads never straddle two reads, so one record search,
// ONE 8-byte window per stream (positions p_hi - 7 .. p_hi) and one round of byte-parallel arithmetic serve four
// symbols:
//   sequence  the eight bases become eight 2-bit codes at once -- ((w >> 1) ^ (w >> 2)) & 0x03.. is 0 1 2 3 for A C G T
//             and 0 for N, which the coder codes as A (src/fse_sequence.cpp:44) -- packed into a 16-bit string H, oldest
//             position lowest; the key of the symbol at window byte k is (H >> 2 (k - 4)) & 0x3FF: context (the four
//             bases in front of it, nearest in bits 7:6: src/fse_sequence.h:22-24) | symbol << 8 in ONE bit-field
//             extract.  Bases in front of the read: the same string with 0xD7 (virtual T,C,C,T) shifted in.
//             A byte that is no base at all is found by mapping the codes back to letters (v_perm) and comparing.
//   quality   33 comes off all eight bytes at once; calcContext (src/fse_quality.h:40-44) per symbol from byte fields.
// The last quad of a read may be short (L mod 4), and a quad at the edge of the wave's range belongs to two waves:
// every symbol is masked by its encode index.  Full quads leave their keys with one 8-byte (+ one 4-byte) store.
struct RecCache4 {
  uint32_t start[65];   // rec_start of records r0 .. r0 + 64
  uint32_t qpre[65];    // quads of this wave's range in the cached records before record k
  uint32_t jlo[64];     // first quad of record k inside the wave's range
  fqgpu_rec rec[64];
};
struct __attribute__((packed)) FqU64x1 { unsigned long long a; };  // eight bytes at any address

#ifndef FQ_K1Q_DEPTH
#define FQ_K1Q_DEPTH 3
#endif
constexpr int K1Q_DEPTH = FQ_K1Q_DEPTH;  // quad chunks (256 symbols) whose window loads are in flight

__global__ void __launch_bounds__(256)
k_tile_hist2(const uint8_t *__restrict__ raw, const fqgpu_rec *__restrict__ recs,
             const uint32_t *__restrict__ rec_start, unsigned R, unsigned n_sym, unsigned T,
             uint16_t *__restrict__ tile_hist_seq, uint8_t *__restrict__ ckey_seq,
             uint16_t *__restrict__ tile_hist_qual, uint16_t *__restrict__ ckey_qual, uint8_t *__restrict__ first_seq,
             uint8_t *__restrict__ first_qual, uint32_t *__restrict__ n_cnt32, BlockResult *res) {
  constexpr unsigned BS = SeqModel::B, BQ = QualModel::B;
  __shared__ uint32_t hist_s[BS];
  __shared__ uint32_t hist_q[BQ / 2];  // 16-bit counters, two per word (T <= 32768: they cannot wrap)
  __shared__ RecCache4 rcache[4];      // one per wave
  const unsigned tile = blockIdx.x;
  const unsigned e0 = tile * T;
  const unsigned e1 = min(e0 + T, n_sym);
  for (unsigned c = threadIdx.x; c < BS; c += blockDim.x) hist_s[c] = 0;
  for (unsigned c = threadIdx.x; c < BQ / 2; c += blockDim.x) hist_q[c] = 0;
  __syncthreads();
  const unsigned wave = threadIdx.x >> 6, lane = fq_lane();
  const unsigned per = (((e1 - e0) + blockDim.x - 1u) / blockDim.x) * 64u;
  const unsigned wb = min(e0 + wave * per, e1), we = min(wb + per, e1);
  bool bad_s = false, bad_q = false;
  if (wb < we) {
    RecCache4 &rc = rcache[wave];
    unsigned r0 = fq_locate(rec_start, 0, R - 1, wb);
    for (;;) {
      // ---- window of 64 records: starts, the quads of every record that fall into [wb, we), their prefix sums
      fq_lds_wave_sync();  // nobody still reads the old window
      rc.start[lane] = r0 + lane <= R ? rec_start[r0 + lane] : 0xFFFFFFFFu;
      if (lane == 0) rc.start[64] = r0 + 64 <= R ? rec_start[r0 + 64] : 0xFFFFFFFFu;
      if (r0 + lane < R) rc.rec[lane] = recs[r0 + lane];
      fq_lds_wave_sync();
      unsigned nq = 0, jl = 0;
      {
        const unsigned rs = rc.start[lane], rn = rc.start[lane + 1];
        if (r0 + lane < R) {
          const unsigned lo = max(rs, wb), hi = min(rn, we);
          if (hi > lo) { jl = (lo - rs) >> 2; nq = ((hi - rs + 3u) >> 2) - jl; }
        }
      }
      const unsigned qincl = sets_incl_scan_k1(nq);
      rc.qpre[lane] = qincl - nq;
      rc.jlo[lane] = jl;
      if (lane == 63) rc.qpre[64] = qincl;
      fq_lds_wave_sync();
      const unsigned Q = fq_uniform(rc.qpre[64]);
      const unsigned covered = fq_uniform(rc.start[64]);
      const unsigned nch = (Q + 63u) >> 6;

      // ---- software pipeline over the chunks of 64 quads
      unsigned long long ws[K1Q_DEPTH], wq[K1Q_DEPTH];  // windows, already shifted: byte k = position p_hi - 7 + k
      unsigned ef[K1Q_DEPTH], vm[K1Q_DEPTH], mm[K1Q_DEPTH], rr[K1Q_DEPTH];  // first encode index, valid mask (4 bits; bit 4: the quad opens its record), missing bytes, record
      unsigned kc = 0;  // (uniform) cached record holding the first quad of the next chunk to be fetched
      auto fetch = [&](int slot, unsigned c) {
        const unsigned g0 = c << 6, g = g0 + lane;
        const bool on = g < Q;
        unsigned k = 0, kk = kc, qn;
        for (;;) {
          const unsigned qs = fq_uniform(rc.qpre[kk]);
          qn = fq_uniform(rc.qpre[kk + 1]);
          if (on && g >= qs && g < qn) k = kk;
          if (qn >= g0 + 64u || kk == 63u) break;
          kk++;
        }
        kc = qn > g0 + 64u ? kk : min(kk + 1u, 63u);
        const fqgpu_rec rec = rc.rec[k];
        const unsigned rs = rc.start[k], j = rc.jlo[k] + (g - rc.qpre[k]);
        const unsigned L = on ? rec.len : 4u, j4 = on ? 4u * j : 0u;
        const unsigned p_hi = L - 1u - j4;            // the quad's first symbol in encode order
        const unsigned m = p_hi < 7u ? 7u - p_hi : 0u;  // window bytes in front of the read
        const unsigned first = p_hi < 7u ? 0u : p_hi - 7u;
        const unsigned long long a = reinterpret_cast<const FqU64x1 *>(raw + (on ? rec.seq_off : 0u) + first)->a;
        const unsigned long long b = reinterpret_cast<const FqU64x1 *>(raw + (on ? rec.qual_off : 0u) + first)->a;
        const unsigned e_first = rs + j4;
        unsigned valid = 0;
#pragma unroll
        for (unsigned i = 0; i < 4; i++) {
          const unsigned e = e_first + i;
          valid |= (on && i <= p_hi && e >= wb && e < we) ? 1u << i : 0u;
        }
        valid |= (j4 == 0u && (valid & 1u)) ? 16u : 0u;  // the record's first symbol in encode order is this quad's first
        ws[slot] = a; wq[slot] = b; ef[slot] = e_first; vm[slot] = valid; mm[slot] = m; rr[slot] = r0 + k;
      };
      auto consume = [&](int slot) {
        const unsigned opens = vm[slot] >> 4, valid = vm[slot] & 15u, sh = 8u * mm[slot], e_first = ef[slot];
        if (!valid) return;
        // -------- sequence
        const unsigned long long w = ws[slot] << sh;   // byte k = position p_hi - 7 + k (0 in front of the read)
        const unsigned lo = (unsigned)w, hi = (unsigned)(w >> 32);
        const unsigned clo = ((lo >> 1) ^ (lo >> 2)) & 0x03030303u, chi = ((hi >> 1) ^ (hi >> 2)) & 0x03030303u;
        auto pack4 = [](unsigned c) { c |= c >> 6; c |= c >> 12; return c & 0xFFu; };  // four byte codes -> 8 bits
        unsigned H = pack4(clo) | (pack4(chi) << 8);
        H |= 0xD700u >> (16u - 2u * mm[slot]);          // virtual T,C,C,T in front of the read (mm = 0: nothing)
        // bytes that are not A C G T: N (counted, coded as A) or no base at all
        const unsigned diff = (__builtin_amdgcn_perm(0u, 0x54474341u, chi) ^ hi);
        unsigned k16[4];
#pragma unroll
        for (unsigned i = 0; i < 4; i++) k16[i] = (H >> (2u * (3u - i))) & 0x3FFu;
        if (diff) {
#pragma unroll
          for (unsigned i = 0; i < 4; i++) {
            const unsigned d = (diff >> (8u * (3u - i))) & 0xFFu;
            if (d && ((valid >> i) & 1u)) {
              if (((hi >> (8u * (3u - i))) & 0xFFu) == 'N') atomicAdd(&n_cnt32[rr[slot]], 1u);  // (rare) replaceAndEncodeNs's n_count, src/fse_sequence.cpp:35-51
              else bad_s = true;
            }
          }
        }
        // -------- quality
        const unsigned long long q33 = (wq[slot] - 0x2121212121212121ull) << sh;  // (a byte below 33 borrows from its UPPER neighbour only: a later position, and then the block is refused anyway)
        const unsigned ql = (unsigned)q33, qh = (unsigned)(q33 >> 32);
        // byte fields 1 .. 7: b[t] = quality at position p_hi - 7 + t
        const unsigned b1 = (ql >> 8) & 0xFFu, b2 = (ql >> 16) & 0xFFu, b3 = ql >> 24, b4 = qh & 0xFFu, b5 = (qh >> 8) & 0xFFu,
                       b6 = (qh >> 16) & 0xFFu, b7 = qh >> 24;
        const unsigned qs_[4] = {b7, b6, b5, b4};   // symbol of quad entry i
        const unsigned qa[4] = {b6, b5, b4, b3};    // position - 1
        const unsigned qb[4] = {b5, b4, b3, b2};    // position - 2
        const unsigned qc[4] = {b4, b3, b2, b1};    // position - 3
        unsigned cq[4];
#pragma unroll
        for (unsigned i = 0; i < 4; i++) {
          cq[i] = ((((qb[i] > qc[i] ? qb[i] : qc[i]) << 6) + qa[i]) & 0xFFFu) | ((qb[i] == qc[i]) ? 0x1000u : 0u);
          bad_q |= ((valid >> i) & 1u) && qs_[i] >= (unsigned)QualModel::A;
        }
        // -------- keys out, histograms.  The keys are the CONTEXTS alone (one byte / two bytes per symbol): the symbol at
        // encode index e is part of the context at e - 1 -- the base in bits 7:6, the quality in bits 5:0 (the position in front
        // of p + 1 is p) -- and K3 takes it from there; only a record's first symbol in encode order has no such neighbour
        // and is left in first_seq / first_qual [record].  240 MB of symbols and key bytes less written here and read by K3.
        if (opens) { first_seq[rr[slot]] = (uint8_t)(k16[0] >> 8); first_qual[rr[slot]] = (uint8_t)(qs_[0] & 63u); }
        if (valid == 0xFu) {
          struct __attribute__((packed)) P8 { uint32_t a, b; };
          struct __attribute__((packed)) P4 { uint32_t a; };
          *reinterpret_cast<P4 *>(ckey_seq + e_first) = P4{(k16[0] & 0xFFu) | ((k16[1] & 0xFFu) << 8) | ((k16[2] & 0xFFu) << 16) | (k16[3] << 24)};
          *reinterpret_cast<P8 *>(ckey_qual + e_first) = P8{cq[0] | (cq[1] << 16), cq[2] | (cq[3] << 16)};
#pragma unroll
          for (unsigned i = 0; i < 4; i++) {
            atomicAdd(&hist_s[k16[i] & 0xFFu], 1u);
            atomicAdd(&hist_q[cq[i] >> 1], 1u << (16u * (cq[i] & 1u)));
          }
        } else {
#pragma unroll
          for (unsigned i = 0; i < 4; i++)
            if ((valid >> i) & 1u) {
              ckey_seq[e_first + i] = (uint8_t)k16[i];
              ckey_qual[e_first + i] = (uint16_t)cq[i];
              atomicAdd(&hist_s[k16[i] & 0xFFu], 1u);
              atomicAdd(&hist_q[cq[i] >> 1], 1u << (16u * (cq[i] & 1u)));
            }
        }
      };
#pragma unroll
      for (int d = 0; d < K1Q_DEPTH - 1; d++)
        if ((unsigned)d < nch) fetch(d, (unsigned)d);
      unsigned c = 0;
      for (; c + 2u * (K1Q_DEPTH - 1) < nch; c += K1Q_DEPTH) {  // steady state: straight-line fetch / consume
#pragma unroll
        for (int d = 0; d < K1Q_DEPTH; d++) {
          fetch((d + K1Q_DEPTH - 1) % K1Q_DEPTH, c + d + (K1Q_DEPTH - 1));
          consume(d);
        }
      }
      for (; c < nch; c += K1Q_DEPTH) {  // drain
#pragma unroll
        for (int d = 0; d < K1Q_DEPTH; d++) {
          const unsigned cur = c + d;
          if (cur < nch) {
            if (cur + (K1Q_DEPTH - 1) < nch) fetch((d + K1Q_DEPTH - 1) % K1Q_DEPTH, cur + (K1Q_DEPTH - 1));
            consume(d);
          }
        }
      }
      if (covered >= we || r0 + 64u >= R) break;
      r0 += 64u;
    }
  }
  if (bad_s) atomicOr(&res->s[0].bad_symbol, 1u);
  if (bad_q) atomicOr(&res->s[1].bad_symbol, 1u);
  __syncthreads();
  // histogram rows as 16-bit entries (a tile has at most 32768 symbols): the quality stream's rows are 3662 x 8192 entries
  // per 256 MiB block -- 60 MB written here and read twice by K2 and once by K3 instead of 120; the packed LDS words go out as they are
  for (unsigned c = threadIdx.x; c < BS; c += blockDim.x) tile_hist_seq[(size_t)tile * BS + c] = (uint16_t)hist_s[c];
  uint32_t *rowq = reinterpret_cast<uint32_t *>(tile_hist_qual + (size_t)tile * BQ);
  for (unsigned c = threadIdx.x; c < BQ / 2; c += blockDim.x) rowq[c] = hist_q[c];
}

// ------------------------------------------------------------------ K2: layout of the sorted arrays
// group_sum[g][c] = sum of tile_hist over the tiles of group g
template <class HT>
__global__ void __launch_bounds__(256)
k_group_sum(const HT *__restrict__ tile_hist, unsigned n_tiles, unsigned B,
            uint32_t *__restrict__ group_sum) {
  const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned g = blockIdx.y;
  if (c >= B) return;
  const unsigned t0 = g * GROUP_TILES, t1 = min(t0 + GROUP_TILES, n_tiles);
  uint32_t acc = 0;
  for (unsigned t = t0; t < t1; t++) acc += tile_hist[(size_t)t * B + c];
  group_sum[(size_t)g * B + c] = acc;
}

// Per-context totals and the exclusive scan over groups (in place): one thread per context over
// the whole grid, eight groups' loads in flight at a time.  (Inside the single workgroup of
// k_ctx_layout this loop was 8 contexts x n_groups dependent load/store pairs per thread: 0.17 ms
// of a 0.45 ms layout step for the quality stream's 8192 contexts and 58 groups.)
__global__ void __launch_bounds__(256)
k_group_prefix(uint32_t *__restrict__ group_sum, unsigned n_groups, unsigned B, uint32_t *__restrict__ ctx_count) {
  const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= B) return;
  uint32_t acc = 0;
  for (unsigned g0 = 0; g0 < n_groups; g0 += 8) {
    uint32_t v[8];
#pragma unroll
    for (unsigned k = 0; k < 8; k++) v[k] = g0 + k < n_groups ? group_sum[(size_t)(g0 + k) * B + c] : 0u;
#pragma unroll
    for (unsigned k = 0; k < 8; k++)
      if (g0 + k < n_groups) { group_sum[(size_t)(g0 + k) * B + c] = acc; acc += v[k]; }
  }
  ctx_count[c] = acc;
}

// One workgroup: the context layout from the per-context totals: padded start of every context's
// run, segment and work-item prefix sums.
// arrays: ctx_count[B] | ctx_start[B+1] | seg_base[B+1] | item_base[B+1]
__global__ void __launch_bounds__(1024)
k_ctx_layout(unsigned B, unsigned S, uint32_t *__restrict__ arrays, uint32_t *__restrict__ fill_none) {
  if (fill_none)  // SegArrays::usym of the generic chain kernels starts from "none" (one memset less on the stream)
    for (unsigned c = threadIdx.x; c < B; c += blockDim.x) fill_none[c] = 0xFFFFFFFFu;
  __shared__ unsigned part[3][1024];
  uint32_t *ctx_count = arrays, *ctx_start = arrays + B, *seg_base = ctx_start + B + 1,
           *item_base = seg_base + B + 1;
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
template <class HT>
__global__ void __launch_bounds__(256)
k_tile_base(const HT *__restrict__ tile_hist, const uint32_t *__restrict__ group_sum,
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
