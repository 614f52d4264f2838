// Block decoder -- replaces the second pass of DecompressionWorkspace::decodeChunk
// (reference src/workspace.cpp:84-87): SequenceDecoder::decodeRecord
// (src/fse_sequence.cpp:114-143) and QualityDecoder::decodeRecord
// (src/fse_quality.cpp:55-67) with FSE_Decoder::startChunk/endChunk
// (src/fse_common.hpp:130-142).
//
// The format gives a decoder nothing to split a stream on: the bit position of a
// symbol depends on every earlier nbBits and its context on the symbols just decoded
// (SURVEY.md 7.3).  Parallelism = (blocks in the batch) x 2 streams: one workgroup per
// (block, stream), one lane walks the chain, the wave only helps where the work is
// data-parallel (loading the 256 / 8192 initial states, whose bit positions are known
// from the table logs).  All blocks of a batch are decoded by ONE launch.
#include "fqgpu_internal.h"

#include <vector>

namespace {

struct DecJob {
  const uint8_t *seq;  unsigned seq_len;
  const uint8_t *qual; unsigned qual_len;
  const fqgpu_rec *recs; unsigned n_recs;
  const uint16_t *n_count;  // the block's own n_recs entries
  const uint16_t *n_pos;  unsigned n_pos_len;
  uint8_t *raw;
  BlockResult *res;
  unsigned rec_base;  // first index of this block in the batch-wide record arrays
  // decode index (extension): both present -> one lane per (stream, stride) instead of one per stream
  const uint8_t *index[2];
  const uint32_t *rec_start;  // [n_recs + 1] encode index of the first symbol of every record
};

struct DecChunk {
  unsigned job, stream, chunk;
};

// Pointers read out of the job descriptors are generic to the compiler, which then emits
// flat_load / flat_store: those also count on lgkmcnt, so every LDS wait of the walking loop
// would wait for the previous output byte to reach memory.  Declare them global.
#define FQ_GLOBAL __attribute__((address_space(1)))
typedef FQ_GLOBAL uint8_t g_u8;
typedef const FQ_GLOBAL uint32_t g_cu32;
typedef const FQ_GLOBAL fqgpu_rec g_crec;

// Per context ONE LDS word: decoder state (12 bits, log <= 12) | the context's DTable position.
// DTables lie back to back, 1 + 2^log words each (log >= 5), so dt_off[c] - c is a multiple of 32
// and (dt_off[c] - c) / 32 < 2^20: word = that << 12 | state; entry = dt[(word >> 12) * 32 + c + 1 + state].
// 32 KB of LDS per quality lane instead of 48 KB (16-bit states + 32-bit offsets): five lanes per CU
// for the strides of the indexed decode, where the number of resident lanes is what counts (the
// one-lane-per-stream kernel keeps the two arrays: one ALU op less on its dependent chain).
__device__ __forceinline__ uint32_t fq_pack_state(uint32_t dt_off_c, unsigned c, unsigned state) {
  return (((dt_off_c - c) >> 5) << 12) | (state & 0xFFFu);
}
__device__ __forceinline__ uint32_t fq_entry_index(uint32_t word, unsigned c) {
  return ((word >> 12) << 5) + c + 1u + (word & 0xFFFu);
}

struct TabView {
  const uint32_t *logs, *log_prefix, *dt, *dt_off;
};

// bits [lo, lo+nb) of the stream, for the data-parallel state load
__device__ __forceinline__ unsigned peek_bits(g_cu32 *w, long long lo, unsigned nb) {
  const unsigned wi = (unsigned)(lo >> 5);
  const unsigned long long v = (unsigned long long)w[wi] | ((unsigned long long)w[wi + 1] << 32);
  return (unsigned)(v >> (unsigned)(lo & 31)) & ((1u << nb) - 1u);
}

// ---- wave-cooperative walk -------------------------------------------------------------------
// The chain of one stream is serial: symbol k's DTable entry is dt[context k][state of that context],
// context k + 1 follows from symbol k.  One lane walking it pays an LDS read plus an L2 (or
// Infinity Cache) read per symbol, back to back.  The other 63 lanes can shorten that: when
// symbols .. k - 1 are known, context k + 1 is one of at most 64 (quality: calcContext(s, q[k-1],
// q[k-2]) for the 64 values s of symbol k; sequence: 4), so lane s fetches the entry that context
// k + 1 would need IF symbol k is s -- one step before symbol k is known -- and symbol k then only
// picks a lane (v_readlane).  Two table reads are in flight at any time instead of one: the walk
// advances two symbols per memory round trip.  A prefetched entry is stale when the context it
// belongs to was updated after the fetch; that is exactly the case context k + 1 == context k
// (runs of one quality value, homopolymers): then the entry is read again, behind the update.
//
// Backward bit reader (BIT_DStream_t, zstd bitstream.h) in functional form: `pos` = number of unread
// bits below the end mark; reading nb bits returns bits [pos-nb, pos) of the little-endian bit array,
// bit pos-1 being the MSB.  The stream travels through a small LDS buffer (all lanes fetch 2 KB at a
// time, once per ~16 K bits), so the 64-bit register window is refilled from LDS: no bit read of the
// walking loop ever touches vmcnt, which the two table reads in flight own.
constexpr unsigned FQ_BITBUF_DW = 512;
struct LdsBits {
  g_cu32 *w;
  uint32_t *buf;        // LDS: stream dwords [buf_lo, buf_lo + FQ_BITBUF_DW)
  unsigned n_dw;        // dwords of the stream (reads beyond are zeros)
  unsigned buf_lo;
  unsigned wdw;         // the window holds stream dwords wdw, wdw + 1
  unsigned avail;       // unread bits of the window: the read position is bit 32 * wdw + avail of the stream
  unsigned underflow;   // a read went below bit 0 (corrupt stream)
  unsigned long long win;
  __device__ __forceinline__ void fill(unsigned top_dw) {  // all lanes; afterwards the buffer ends with dword top_dw
    buf_lo = top_dw >= FQ_BITBUF_DW - 1 ? top_dw - (FQ_BITBUF_DW - 1) : 0u;
    fq_lds_wave_sync();
    for (unsigned k = threadIdx.x; k < FQ_BITBUF_DW; k += 64) buf[k] = buf_lo + k < n_dw ? w[buf_lo + k] : 0u;
    fq_lds_wave_sync();
  }
  __device__ __forceinline__ void load_window() {
    if (wdw < buf_lo) fill(wdw + 1);
    const uint32_t lo = buf[wdw - buf_lo], hi = buf[wdw + 1 - buf_lo];
    // (kept per lane: with the two words forced into scalar registers by v_readfirstlane the walk
    // decoded wrong bits on gfx950 / ROCm 7.2 -- measured, cause not found; the window is 2 VGPRs)
    win = ((unsigned long long)hi << 32) | lo;
  }
  __device__ __forceinline__ void init(g_cu32 *words, long long p, uint32_t *lds, unsigned stream_dwords) {
    w = words; buf = lds; n_dw = stream_dwords; underflow = 0;
    const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)p), phi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
    const long long pu = ((long long)phi << 32) | plo;
    const long long top = (pu + 31) & ~31ll;
    wdw = top >= 64 ? (unsigned)(top >> 5) - 2u : 0u;
    avail = (unsigned)(pu - (long long)wdw * 32);
    buf_lo = 0;
    fill(wdw + 1);
    load_window();
  }
  __device__ __forceinline__ long long pos() const { return underflow ? -1ll : (long long)wdw * 32 + avail; }
  __device__ __forceinline__ unsigned read(unsigned nb) {  // nb <= 12
    if (avail < nb) {
      if (wdw == 0) { underflow = 1; avail = 0; return 0u; }  // corrupt stream: the caller checks pos()
      wdw -= 1; avail += 32;
      load_window();
    }
    avail -= nb;
    return (unsigned)(win >> avail) & ((1u << nb) - 1u);
  }
};

// history of the context model: sequence: the context itself; quality: the last three symbols
template <class M> struct CtxHist;
template <> struct CtxHist<SeqModel> {
  unsigned ctx;
  __device__ __forceinline__ void start() { ctx = 0xD7u; }  // FSE_Sequence::INITIAL_CONTEXT
  __device__ __forceinline__ unsigned cur() const { return ctx; }
  __device__ __forceinline__ unsigned next_if(unsigned s) const { return (ctx >> 2) + ((s & 3u) << 6); }  // addSymUpper
  __device__ __forceinline__ void push(unsigned s) { ctx = (ctx >> 2) + (s << 6); }
};
template <> struct CtxHist<QualModel> {
  unsigned q, q1, q2;  // symbols k-1, k-2, k-3
  __device__ __forceinline__ void start() { q = q1 = q2 = 0; }
  __device__ __forceinline__ unsigned cur() const { return fq_qual_ctx(q, q1, q2); }  // calcContext
  __device__ __forceinline__ unsigned next_if(unsigned s) const { return fq_qual_ctx(s & 63u, q, q1); }
  __device__ __forceinline__ void push(unsigned s) { q2 = q1; q1 = q; q = s; }
};

// entry of context c in its current state, and the context's LDS word it was found through
__device__ __forceinline__ uint32_t fq_entry_of(const uint32_t *pk, const uint32_t *__restrict__ dt, unsigned c, uint32_t &word) {
  word = pk[c];
  return dt[fq_entry_index(word, c)];
}

// Positions [i0, i1) of one read into out[i0 ..], all 64 lanes of the wave; h = history in front of
// position i0.  Everything the walk decides on (entry, symbol, context, bit window) is wave-uniform
// and kept in scalar registers (readfirstlane / readlane), the candidates are per lane.  The loop is
// unrolled by two with the candidate registers swapped by name: a register copy "cand = next" at the
// end of an iteration would wait for the load just issued.  Bytes are collected 64 at a time (lane k
// keeps byte k) and stored as whole lines.
template <class M>
__device__ __forceinline__ void walk_positions(uint32_t *pk, const uint32_t *__restrict__ dt, LdsBits &br, g_u8 *out,
                                               unsigned i0, unsigned i1, CtxHist<M> h) {
  if (i0 >= i1) return;
  const unsigned lane = threadIdx.x;
  unsigned ctx = h.cur();
  uint32_t wv, candw_a, candw_b = 0;
  uint32_t cur = __builtin_amdgcn_readfirstlane(fq_entry_of(pk, dt, ctx, wv));  // entry of position i0
  uint32_t curw = __builtin_amdgcn_readfirstlane(wv);
  uint32_t cand_a = fq_entry_of(pk, dt, h.next_if(lane), candw_a), cand_b = 0;      // position i0 + 1, if symbol i0 is `lane`
  unsigned keep = 0;
  // one position: consumes cur, leaves the candidates of position i + 2 in (cout, coutw), picks position i + 1's entry from (cin, cinw)
  auto step = [&](unsigned i, uint32_t cin, uint32_t cinw, uint32_t &cout, uint32_t &coutw) {
    const unsigned sym = (cur >> 16) & (unsigned)(M::A - 1);
    const unsigned ns = (cur & 0xFFFFu) + br.read(cur >> 24);
    pk[ctx] = (curw & ~0xFFFu) | (ns & 0xFFFu);
    const unsigned byte = M::STREAM == 0 ? (0x54474341u >> (8u * sym)) & 0xFFu : sym + 33u;  // "ACGT"[sym] / Phred + 33
    keep = lane == ((i - i0) & 63u) ? byte : keep;
    if (((i - i0) & 63u) == 63u) out[i - 63u + lane] = (uint8_t)keep;
    const unsigned prev_ctx = ctx;
    h.push(sym);
    ctx = h.cur();
    cout = fq_entry_of(pk, dt, h.next_if(lane), coutw);  // behind the update above, in front of the one of position i + 1
    if (ctx == prev_ctx) {  // (uniform) the prefetched entry is older than the update: read again
      uint32_t w2;
      cur = __builtin_amdgcn_readfirstlane(fq_entry_of(pk, dt, ctx, w2));
      curw = __builtin_amdgcn_readfirstlane(w2);
    } else {
      cur = (uint32_t)__builtin_amdgcn_readlane((int)cin, (int)sym);
      curw = (uint32_t)__builtin_amdgcn_readlane((int)cinw, (int)sym);
    }
  };
  unsigned i = i0;
  for (; i + 2 <= i1; i += 2) {
    step(i, cand_a, candw_a, cand_b, candw_b);
    step(i + 1, cand_b, candw_b, cand_a, candw_a);
  }
  if (i < i1) step(i, cand_a, candw_a, cand_b, candw_b);
  const unsigned n = i1 - i0, tail = n & 63u;
  if (lane < tail) out[i1 - tail + lane] = (uint8_t)keep;
}

template <class M>
__device__ void decode_stream(const DecJob &j, const TabView &tab, uint32_t *pk, uint32_t *bitbuf) {
  constexpr unsigned B = M::B;
  const uint8_t *src = M::STREAM == 0 ? j.seq : j.qual;
  const unsigned len = M::STREAM == 0 ? j.seq_len : j.qual_len;
  StreamResult *res = &j.res->s[M::STREAM];
  const unsigned lane = threadIdx.x;
  g_cu32 *w = (g_cu32 *)reinterpret_cast<const uint32_t *>(src);
  g_crec *recs = (g_crec *)j.recs;
  g_u8 *raw = (g_u8 *)j.raw;

  // BIT_initDStream: the highest set bit of the last byte is the end mark
  const unsigned last = len ? src[len - 1] : 0u;
  if (last == 0) { if (lane == 0) res->corrupt = 1; return; }
  const long long p0 = (long long)(len - 1) * 8 + (31 - __clz((int)last));
  const unsigned sum_logs = tab.log_prefix[B];
  if (p0 < (long long)sum_logs) { if (lane == 0) res->corrupt = 1; return; }
  // FSE_initDState for ctx B-1 .. 0 (src/fse_common.hpp:134-138): ctx c sits at a fixed
  // offset below the end mark, so all of them load in parallel
  for (unsigned c = lane; c < B; c += 64) {
    const unsigned lg = tab.logs[c];
    const long long lo = p0 - (long long)(sum_logs - tab.log_prefix[c]);
    pk[c] = fq_pack_state(tab.dt_off[c], c, peek_bits(w, lo, lg));
  }
  __syncthreads();

  LdsBits br;
  br.init(w, p0 - (long long)sum_logs, bitbuf, (len + 3) / 4);
  const uint32_t *__restrict__ dt = tab.dt;
  fqgpu_rec nxt;
  nxt.seq_off = recs[j.n_recs - 1].seq_off; nxt.qual_off = recs[j.n_recs - 1].qual_off; nxt.len = recs[j.n_recs - 1].len;
  for (unsigned r = j.n_recs; r > 0; r--) {  // records last -> first (src/workspace.cpp:84-87)
    const fqgpu_rec rec = nxt;
    if (r > 1) { nxt.seq_off = recs[r - 2].seq_off; nxt.qual_off = recs[r - 2].qual_off; nxt.len = recs[r - 2].len; }  // lands while this record is walked
    CtxHist<M> h;
    h.start();
    walk_positions<M>(pk, dt, br, raw + (M::STREAM == 0 ? rec.seq_off : rec.qual_off), 0u, rec.len, h);
    if (br.underflow) break;
  }
  // BIT_endOfDStream (src/fse_common.hpp:141): every bit consumed, none invented
  if (lane == 0) {
    if (br.pos() != 0) res->corrupt = 1;
    res->total_bits = (unsigned long long)(p0 - (long long)sum_logs);
  }
}

// One stride of one stream, started from a snapshot of the decode index (or from the end of the
// stream for the last stride): encode indices [e_lo, e_hi) in decoder order, i.e. from the record
// and position of symbol e_hi - 1 towards the front of the block.
template <class M>
__device__ void decode_chunk(const DecJob &j, unsigned chunk, const TabView &tab, uint32_t *pk, uint32_t *bitbuf) {
  constexpr unsigned B = M::B;
  const uint8_t *src = M::STREAM == 0 ? j.seq : j.qual;
  const unsigned len = M::STREAM == 0 ? j.seq_len : j.qual_len;
  StreamResult *res = &j.res->s[M::STREAM];
  const unsigned lane = threadIdx.x;
  g_cu32 *w = (g_cu32 *)reinterpret_cast<const uint32_t *>(src);
  g_crec *recs = (g_crec *)j.recs;
  g_cu32 *rec_start = (g_cu32 *)j.rec_start;
  g_u8 *raw = (g_u8 *)j.raw;
  const FqIndexHeader hdr = *reinterpret_cast<const FqIndexHeader *>(j.index[M::STREAM]);
  const size_t snap_bytes = FQ_INDEX_SNAP_HEAD + 2 * (size_t)B;
  const uint8_t *snaps = j.index[M::STREAM] + sizeof(FqIndexHeader);
  const unsigned n_sym = (unsigned)hdr.n_sym, stride = hdr.stride;
  const unsigned e_lo = chunk * stride, e_hi = min(e_lo + stride, n_sym);
  const bool from_end = chunk == hdr.n_snap;  // the last stride starts at the stream's end mark

  long long pos;
  unsigned prev = 0xFFFFFFFFu;
  if (from_end) {
    const unsigned last = len ? src[len - 1] : 0u;
    if (last == 0) { if (lane == 0) res->corrupt = 1; return; }
    const long long p0 = (long long)(len - 1) * 8 + (31 - __clz((int)last));
    const unsigned sum_logs = tab.log_prefix[B];
    if (p0 < (long long)sum_logs) { if (lane == 0) res->corrupt = 1; return; }
    for (unsigned c = lane; c < B; c += 64) {
      const long long lo = p0 - (long long)(sum_logs - tab.log_prefix[c]);
      pk[c] = fq_pack_state(tab.dt_off[c], c, peek_bits(w, lo, tab.logs[c]));
    }
    pos = p0 - (long long)sum_logs;
    if (lane == 0) res->total_bits = (unsigned long long)pos;
  } else {
    const uint8_t *snap = snaps + (size_t)chunk * snap_bytes;  // snapshot chunk + 1 sits at e_hi
    const uint16_t *st = reinterpret_cast<const uint16_t *>(snap + FQ_INDEX_SNAP_HEAD);
    for (unsigned c = lane; c < B; c += 64) {
      pk[c] = fq_pack_state(tab.dt_off[c], c, (unsigned)st[c] & ((1u << tab.logs[c]) - 1u));  // a damaged index must not leave the table
    }
    pos = (long long)*reinterpret_cast<const unsigned long long *>(snap);
    prev = reinterpret_cast<const uint32_t *>(snap)[2];
    if (pos > (long long)len * 8) { if (lane == 0) res->corrupt = 1; return; }
  }
  __syncthreads();
  // every bit of this stride consumed, none invented: the walk must end where the previous
  // snapshot (or the start of the stream) says
  const long long pos_end = chunk == 0 ? 0ll : (long long)*reinterpret_cast<const unsigned long long *>(snaps + (size_t)(chunk - 1) * snap_bytes);

  LdsBits br;
  br.init(w, pos, bitbuf, (len + 3) / 4);
  const uint32_t *__restrict__ dt = tab.dt;
  unsigned r = fq_locate((const uint32_t *)j.rec_start, 0, j.n_recs - 1, e_hi - 1);  // record of symbol e_hi - 1
  bool first = true;
  for (;;) {
    const unsigned rs = rec_start[r];
    fqgpu_rec rec;
    rec.seq_off = recs[r].seq_off; rec.qual_off = recs[r].qual_off; rec.len = recs[r].len;
    // positions of this record inside [e_lo, e_hi): encode index of position i is rs + len - 1 - i
    const unsigned i0 = first ? rec.len - 1u - (e_hi - 1u - rs) : 0u;
    const unsigned i1 = rs >= e_lo ? rec.len : rec.len - (e_lo - rs);  // one past the last position
    CtxHist<M> h;
    h.start();
    if (first && !from_end) {  // the stride starts inside a read: the model has seen the bytes in front
      if constexpr (M::STREAM == 0) {
        for (int b = 3; b >= 0; b--) {
          const unsigned ch = (prev >> (8 * b)) & 0xFFu;
          if (ch != 0xFFu) h.push(fq_base_code(ch));
        }
      } else {
        const unsigned a = prev & 0xFFu, b = (prev >> 8) & 0xFFu, c = (prev >> 16) & 0xFFu;
        h.q = a != 0xFFu ? (a - 33u) & 63u : 0u;
        h.q1 = b != 0xFFu ? (b - 33u) & 63u : 0u;
        h.q2 = c != 0xFFu ? (c - 33u) & 63u : 0u;
      }
    }
    walk_positions<M>(pk, dt, br, raw + (M::STREAM == 0 ? rec.seq_off : rec.qual_off), i0, i1, h);
    first = false;
    if (br.underflow || rs <= e_lo || r == 0) break;
    r--;
  }
  if (lane == 0 && br.pos() != pos_end) res->corrupt = 1;
}

__global__ void __launch_bounds__(64)
k_decode_chunks(const DecJob *__restrict__ jobs, const DecChunk *__restrict__ chunks, TabView seq_tab, TabView qual_tab) {
  __shared__ uint32_t pk[QualModel::B];
  __shared__ uint32_t bitbuf[FQ_BITBUF_DW];
  const DecChunk ch = chunks[blockIdx.x];
  if (ch.stream) decode_chunk<QualModel>(jobs[ch.job], ch.chunk, qual_tab, pk, bitbuf);
  else decode_chunk<SeqModel>(jobs[ch.job], ch.chunk, seq_tab, pk, bitbuf);
}

// record lengths of one block, for the encode index of the first symbol of every record
__global__ void __launch_bounds__(256)
k_lens_of(const fqgpu_rec *__restrict__ recs, unsigned n, uint32_t *__restrict__ lens32) {
  const unsigned r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) lens32[r] = recs[r].len;
}

// grid = 2 * n_blocks: the quality streams (longer chains, MB-scale DTables) are
// dispatched first, the sequence streams behind them
__global__ void __launch_bounds__(64)
k_decode(const DecJob *__restrict__ jobs, unsigned n_blocks, TabView seq_tab, TabView qual_tab) {
  __shared__ uint32_t pk[QualModel::B];
  __shared__ uint32_t bitbuf[FQ_BITBUF_DW];
  if (blockIdx.x < n_blocks) decode_stream<QualModel>(jobs[blockIdx.x], qual_tab, pk, bitbuf);
  else decode_stream<SeqModel>(jobs[blockIdx.x - n_blocks], seq_tab, pk, bitbuf);
}

// batch-wide record arrays: N counts widened for the scan
__global__ void __launch_bounds__(256)
k_gather_ncount(const DecJob *__restrict__ jobs, uint32_t *__restrict__ cnt32) {
  const DecJob j = jobs[blockIdx.y];
  for (unsigned r = blockIdx.x * blockDim.x + threadIdx.x; r < j.n_recs; r += gridDim.x * blockDim.x)
    cnt32[j.rec_base + r] = j.n_count[r];
}

// N restoration (tail of SequenceDecoder::decodeRecord, src/fse_sequence.cpp:138-142).
// The reference pops counts and deltas from the END of n_count / n_pos while walking the
// records backwards, which equals forward indexing from (n_pos_len - total N of the block).
__global__ void __launch_bounds__(256)
k_npatch(const DecJob *__restrict__ jobs, const uint32_t *__restrict__ off) {
  const DecJob j = jobs[blockIdx.y];
  const unsigned first = off[j.rec_base], total = off[j.rec_base + j.n_recs] - first;
  if (total > j.n_pos_len) {
    if (blockIdx.x == 0 && threadIdx.x == 0) j.res->s[0].corrupt = 1;
    return;
  }
  const unsigned shift = j.n_pos_len - total;
  for (unsigned r = blockIdx.x * blockDim.x + threadIdx.x; r < j.n_recs; r += gridDim.x * blockDim.x) {
    const unsigned cnt = j.n_count[r];
    if (!cnt) continue;
    const fqgpu_rec rec = j.recs[r];
    const uint16_t *d = j.n_pos + shift + (off[j.rec_base + r] - first);
    unsigned at = 0;
    for (unsigned k = 0; k < cnt; k++) {
      at += d[k];
      if (at >= rec.len) { j.res->s[0].corrupt = 1; break; }
      j.raw[rec.seq_off + at] = 'N';
    }
  }
}

// overwrites every sequence / quality byte of the block (decode target, tests, bench)
__global__ void __launch_bounds__(256)
k_wipe(uint8_t *__restrict__ raw, const fqgpu_rec *__restrict__ recs, unsigned R) {
  const unsigned waves = (gridDim.x * blockDim.x) >> 6;
  const unsigned lane = fq_lane();
  for (unsigned r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < R; r += waves) {
    const fqgpu_rec rec = recs[r];
    for (unsigned i = lane; i < rec.len; i += 64) {
      raw[rec.seq_off + i] = '?';
      raw[rec.qual_off + i] = '?';
    }
  }
}

}  // namespace

int fq_wipe_launch(fqgpu_ctx *ctx, fqgpu_dblock *b) {
  const unsigned R = (unsigned)b->n_recs;
  int rcs = fqgpu_sync(ctx);  // the block may still be in an encode lane
  if (rcs) return rcs;
  if (!R) return FQGPU_OK;
  const unsigned blocks = (unsigned)min((size_t)(R + 3) / 4, (size_t)8192);
  hipLaunchKernelGGL(k_wipe, dim3(blocks), dim3(256), 0, ctx->stream, b->raw, b->recs, R);
  FQ_HIP(hipGetLastError());
  return FQGPU_OK;
}

int fq_decode_launch(fqgpu_ctx *ctx, fqgpu_dblock *const *blocks_in, size_t n_blocks) {
  hipStream_t st = ctx->stream;
  if (!n_blocks) return FQGPU_OK;
  int rcs = fqgpu_sync(ctx);  // blocks may still be in an encode lane
  if (rcs) return rcs;
  // blocks with a decode index (both streams, at least one snapshot) go to the chunk kernel, the
  // others to the one-lane-per-stream kernel: plain blocks first in the job array
  auto snaps_of = [](const fqgpu_dblock *b, int stream) -> size_t {
    const size_t sb = fq_index_snap_bytes(stream ? FQGPU_QUAL_MODELS : FQGPU_SEQ_MODELS);
    return b->index_bytes[stream] >= sizeof(FqIndexHeader) ? (b->index_bytes[stream] - sizeof(FqIndexHeader)) / sb : 0;
  };
  std::vector<fqgpu_dblock *> blocks;
  blocks.reserve(n_blocks);
  for (size_t i = 0; i < n_blocks; i++)
    if (!(snaps_of(blocks_in[i], 0) && snaps_of(blocks_in[i], 1))) blocks.push_back(blocks_in[i]);
  const size_t n_plain = blocks.size();
  for (size_t i = 0; i < n_blocks; i++)
    if (snaps_of(blocks_in[i], 0) && snaps_of(blocks_in[i], 1)) blocks.push_back(blocks_in[i]);

  std::vector<DecJob> host(n_blocks);
  std::vector<DecChunk> chunks;
  size_t r_tot = 0, r_max = 0, rs_tot = 0;
  for (size_t i = 0; i < n_blocks; i++) {
    const fqgpu_dblock *b = blocks[i];
    DecJob &j = host[i];
    j.seq = b->seq;   j.seq_len = (unsigned)b->seq_len;
    j.qual = b->qual; j.qual_len = (unsigned)b->qual_len;
    j.recs = b->recs; j.n_recs = (unsigned)b->n_recs;
    j.n_count = b->n_count;
    j.n_pos = b->n_pos; j.n_pos_len = (unsigned)b->n_pos_len;
    j.raw = b->raw;
    j.res = b->result;
    j.rec_base = (unsigned)r_tot;
    j.index[0] = j.index[1] = nullptr;
    j.rec_start = nullptr;
    r_tot += b->n_recs;
    if (b->n_recs > r_max) r_max = b->n_recs;
    if (i >= n_plain) rs_tot += b->n_recs + 1;
  }
  int rc;
  if ((rc = ctx->dec_desc.reserve(n_blocks * sizeof(DecJob)))) return rc;
  if ((rc = ctx->n_cnt32.reserve((r_tot + 1) * 4))) return rc;
  if ((rc = ctx->n_off.reserve((r_tot + 1) * 4))) return rc;
  if (n_plain < n_blocks) {
    if ((rc = ctx->dec_recstart.reserve(rs_tot * 4 + 64))) return rc;
    size_t at = 0;
    for (size_t i = n_plain; i < n_blocks; i++) {
      const fqgpu_dblock *b = blocks[i];
      DecJob &j = host[i];
      j.index[0] = b->index[0]; j.index[1] = b->index[1];
      uint32_t *rs = ctx->dec_recstart.as<uint32_t>() + at;
      j.rec_start = rs;
      at += b->n_recs + 1;
      // lengths -> n_cnt32 (free until the N pass), exclusive scan -> rec_start
      hipLaunchKernelGGL(k_lens_of, dim3((unsigned)((b->n_recs + 255) / 256)), dim3(256), 0, st, b->recs, (unsigned)b->n_recs,
                         ctx->n_cnt32.as<uint32_t>());
      if ((rc = fq_scan_u32_to_u32(st, ctx->n_cnt32.as<uint32_t>(), b->n_recs, rs, ctx->scan_tmp))) return rc;
      for (int s = 1; s >= 0; s--)  // quality strides first: they are the longer ones
        for (size_t k = snaps_of(b, s) + 1; k-- > 0;) chunks.push_back(DecChunk{(unsigned)i, (unsigned)s, (unsigned)k});
    }
    if ((rc = ctx->dec_chunks.reserve(chunks.size() * sizeof(DecChunk)))) return rc;
    FQ_HIP(hipMemcpyAsync(ctx->dec_chunks.p, chunks.data(), chunks.size() * sizeof(DecChunk), hipMemcpyHostToDevice, st));
  }
  hipError_t he = hipMemcpyAsync(ctx->dec_desc.p, host.data(), n_blocks * sizeof(DecJob), hipMemcpyHostToDevice, st);
  if (he == hipSuccess) he = hipStreamSynchronize(st);  // the host vectors die with this call
  if (he != hipSuccess) return fq_hip_error(he, __FILE__, __LINE__);
  const DecJob *jobs = ctx->dec_desc.as<DecJob>();

  for (size_t i = 0; i < n_blocks; i++)
    FQ_HIP(hipMemsetAsync(blocks[i]->result, 0, sizeof(BlockResult), st));
  TabView ts = {ctx->tab[0].logs, ctx->tab[0].log_prefix, ctx->tab[0].dt, ctx->tab[0].dt_off};
  TabView tq = {ctx->tab[1].logs, ctx->tab[1].log_prefix, ctx->tab[1].dt, ctx->tab[1].dt_off};
  fq_timer_span_begin(ctx, "decode", st);
  if (n_plain)
    hipLaunchKernelGGL(k_decode, dim3((unsigned)(2 * n_plain)), dim3(64), 0, st, jobs, (unsigned)n_plain, ts, tq);
  if (!chunks.empty())
    hipLaunchKernelGGL(k_decode_chunks, dim3((unsigned)chunks.size()), dim3(64), 0, st, jobs, ctx->dec_chunks.as<DecChunk>(), ts, tq);
  fq_timer_span_end(ctx, st);
  fq_timer_span_begin(ctx, "npatch", st);
  const unsigned gx = (unsigned)min((r_max + 255) / 256, (size_t)4096);
  hipLaunchKernelGGL(k_gather_ncount, dim3(gx ? gx : 1, (unsigned)n_blocks), dim3(256), 0, st, jobs,
                     ctx->n_cnt32.as<uint32_t>());
  if ((rc = fq_scan_u32_to_u32(st, ctx->n_cnt32.as<uint32_t>(), r_tot, ctx->n_off.as<uint32_t>(),
                               ctx->scan_tmp)))
    return rc;
  hipLaunchKernelGGL(k_npatch, dim3(gx ? gx : 1, (unsigned)n_blocks), dim3(256), 0, st, jobs,
                     ctx->n_off.as<uint32_t>());
  fq_timer_span_end(ctx, st);
  FQ_HIP(hipGetLastError());
  return FQGPU_OK;
}
