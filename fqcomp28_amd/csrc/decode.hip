// Block decoder -- replaces the second pass of DecompressionWorkspace::decodeChunk
// (reference src/workspace.cpp:84-87): SequenceDecoder::decodeRecord
// (src/fse_sequence.cpp:114-143) and QualityDecoder::decodeRecord
// (src/fse_quality.cpp:55-67) with FSE_Decoder::startChunk/endChunk
// (src/fse_common.hpp:130-142).
//
// The format gives a decoder nothing to split a stream on: the bit position of a
// symbol depends on every earlier nbBits and its context on the symbols just decoded
// (SURVEY.md 7.3).  Parallelism = (blocks in the batch) x 2 streams: one 64-lane workgroup per
// (block, stream) -- or per (block, stream, stride) with the decode index --, the wave walks
// the chain with everything it decides on in scalar registers and helps where the work is
// data-parallel (the 256 / 8192 initial entries, whose bit positions are known from the table
// logs; the bit buffer; the stores).  A batch is one launch, or one per stream when it has
// more chains than the chip has places (fq_decode_launch).
#include "fqgpu_internal.h"

#include <type_traits>
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

// Per context TWO LDS words: the DTable entry of the context's CURRENT state and the position of the
// context's table (index of its state-0 entry in dt).  The decoder never needs the state itself
// again once the entry is there: the next state is entry.newState + bits, i.e. the next entry is
// dt[table + that].  DTables lie back to back, 1 + 2^log words each.
struct CtxEntry {
  uint32_t entry, table;
};
typedef __attribute__((address_space(3))) CtxEntry lds_CtxEntry;  // the compiler must see LDS, not a generic pointer (flat_load)
struct TabView {
  const uint32_t *logs, *log_prefix, *dt, *dt_off;
};

// bits [lo, lo+nb) of the stream, for the data-parallel state load
__device__ __forceinline__ unsigned peek_bits(g_cu32 *w, long long lo, unsigned nb) {
  const unsigned wi = (unsigned)(lo >> 5);
  const unsigned long long v = (unsigned long long)w[wi] | ((unsigned long long)w[wi + 1] << 32);
  return (unsigned)(v >> (unsigned)(lo & 31)) & ((1u << nb) - 1u);
}

// ---- the walk -------------------------------------------------------------------------------
// The chain of one stream is serial: symbol k's DTable entry is dt[table of context k + state of
// that context], context k + 1 follows from symbol k.  Looked up when it is needed, that is an L2
// (or Infinity Cache) read per symbol on the dependent chain.  But the entry a context will need
// NEXT is known the moment the context is left: its new state is entry.newState + bits.  So the
// walk keeps, per context, the entry of the current state in LDS (CtxEntry) and refills it when the
// context is used: an LDS-DMA load (global_load_lds_dword, lane 0) of dt[table + new state]
// straight into the context's LDS slot -- no register, nothing waits for it.  The dependent chain
// of a symbol is: entry (scalar) -> symbol -> next context -> ONE LDS read.
//
// A context that comes back before its refill has landed (runs of one quality value,
// homopolymers) must wait for it.  The slot itself says so: before the refill is issued the slot's
// entry is overwritten with FQ_ENTRY_PENDING (no real entry: nbBits <= 12), the DMA replaces it
// with the new entry, and a reader that finds the mark drains the vector-memory counter and reads
// again.  Order: LDS operations of a wave complete in issue order and the mark is waited for
// (lgkmcnt) before the DMA is issued, so the mark can never overwrite the refill; a second refill
// of a context cannot be issued before the first one has been read, i.e. has landed.
//
// One wave per stream, and that wave is bound by the number of instructions it issues per symbol
// (about one per 4-5 cycles, more for a taken branch), not by memory: the loop is written to be
// straight-line -- the bit reader reads its two stream words from the LDS buffer every symbol
// instead of branching on a register window, the DMA is issued under an EXEC mask set in the
// assembly, bytes are collected four per register and sixty-four registers per store.
//
// The refill is issued from inline assembly: the compiler then does not know of an LDS-DMA in
// flight, and does not put a vmcnt(0) in front of every later LDS read.  Its own counted waits
// stay correct (operations it does not know of only make them wait for more).
//
// Backward bit reader (BIT_DStream_t, zstd bitstream.h) in functional form: `pos` = number of unread
// bits below the end mark; reading nb bits returns bits [pos-nb, pos) of the little-endian bit array,
// bit pos-1 being the MSB.  The stream travels through a small LDS buffer (all lanes fetch 2 KB at a
// time, once per ~16 K bits); position = 32 * wdw + avail, 0 <= avail < 32.
constexpr unsigned FQ_BITBUF_DW = 512;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
// The position is (d_lo, rel): the scalar register pair W holds stream dwords d_lo (low half) and
// d_lo + 1, the unread bits end at bit `rel` of W, 32 <= rel < 64 before every read.  A read of nb <= 12
// bits is three scalar instructions (subtract, 64-bit shift, mask); when rel drops below 32 the
// window slides down one dword, which was asked for from the LDS buffer at the previous slide.  (Until
// round 3 every read fetched its two dwords from LDS into vector registers: four more instructions and
// one more LDS operation in front of the entry read of the chain, per symbol.)
struct LdsBits {
  g_cu32 *w;
  lds_u32 *buf;         // LDS: stream dwords [buf_lo, buf_lo + FQ_BITBUF_DW)
  unsigned n_dw;        // dwords of the stream (reads beyond are zeros)
  int buf_lo;
  unsigned long long W; // (wave-uniform: scalar registers)
  int d_lo;             // stream dword in the low half of W; dwords below 0 read as zero (corrupt stream)
  unsigned rel;
  unsigned vnext;       // dword d_lo - 1, asked for (per lane: the same value in every lane)
  unsigned underflow;   // a read went below bit 0 (corrupt stream)
  __device__ __forceinline__ void fill(int top_dw) {  // all lanes; afterwards the buffer ends with dword top_dw
    buf_lo = top_dw >= (int)FQ_BITBUF_DW - 1 ? top_dw - ((int)FQ_BITBUF_DW - 1) : 0;
    fq_lds_wave_sync();
    for (unsigned k = threadIdx.x; k < FQ_BITBUF_DW; k += 64) buf[k] = (unsigned)buf_lo + k < n_dw ? w[(unsigned)buf_lo + k] : 0u;
    fq_lds_wave_sync();
  }
  // dword i of the stream out of the buffer (i < 0: zero); the buffer is moved when i lies below it
  __device__ __forceinline__ unsigned dword(int i) {
    if (i < 0) return 0u;
    if (i < buf_lo) fill(i + 1 < (int)FQ_BITBUF_DW ? (int)FQ_BITBUF_DW - 1 : i + 1);  // (as far up as keeps dword i inside and dword 0 at the bottom when possible)
    return buf[i - buf_lo];
  }
  __device__ __forceinline__ void place(long long pos0) {  // position = absolute bit pos0 >= 0
    const int wdw = (int)(pos0 >> 5);
    fill(wdw + 1);
    const unsigned hi = fq_uniform(dword(wdw)), lo = fq_uniform(dword(wdw - 1));
    W = ((unsigned long long)hi << 32) | lo;
    d_lo = wdw - 1;
    rel = 32u + (unsigned)(pos0 & 31);
    vnext = dword(d_lo - 1);
  }
  __device__ __forceinline__ void init(g_cu32 *words, long long pos0, uint32_t *lds, unsigned stream_dwords) {
    w = words; buf = (lds_u32 *)lds; n_dw = stream_dwords; underflow = 0;
    const unsigned plo = fq_uniform((unsigned)pos0), phi = fq_uniform((unsigned)(pos0 >> 32));
    place(((long long)phi << 32) | plo);
  }
  __device__ __forceinline__ long long pos() const { return underflow ? -1ll : (long long)d_lo * 32 + (long long)rel; }
  // the window moves down one dword (rel < 32): every 32 bits, i.e. every 5 .. 30 symbols
  __device__ __forceinline__ void slide() {
    W = (W << 32) | fq_uniform(vnext);
    d_lo--;
    rel += 32u;
    if (d_lo < -1) underflow = 1;  // (what is read from then on is zeros: arbitrary but inside the tables; the caller checks pos())
    vnext = dword(d_lo - 1);
  }
  // the next nb bits (nb <= 12; uniform)
  // (the window is looked after every SECOND read: 32 <= rel before two reads of at most 12 bits each leaves
  // rel >= 8, and one slide brings it back above 32 -- a compare and a branch less every other symbol)
  __device__ __forceinline__ unsigned read(unsigned nb, bool look_after) {
    rel -= nb;
    unsigned v = (unsigned)(W >> rel) & ((1u << nb) - 1u);
    asm volatile("" : "+s"(v));  // cut out before the slide's branch: sunk behind it, both versions of W stay live (two copies per symbol)
    if (look_after && __builtin_expect(rel < 32u, 0)) slide();
    return v;
  }
};

// History of the context model, in units of the walk's LDS slots (sizeof(CtxEntry) = 8 bytes): symbols
// arrive as symbol * 8 (the entry's symbol field as it is), slot() = 8 * context.
// sequence: the context itself; quality: the last three symbols
template <class M> struct CtxHist;
// next(s8) = slot of the context behind symbol s8, with as little as possible between the symbol and
// the slot (that is the chain): the part of the next context that does not depend on the new symbol
// (`pre`) is worked out one symbol earlier, in push().
template <> struct CtxHist<SeqModel> {
  unsigned c8, pre;
  __device__ __forceinline__ void start() { c8 = 0xD7u * 8u; pre = (c8 >> 2) & ~7u; }  // FSE_Sequence::INITIAL_CONTEXT
  __device__ __forceinline__ unsigned slot() const { return c8; }
  __device__ __forceinline__ unsigned next(unsigned s8) const { return pre + (s8 << 6); }  // addSymUpper
  __device__ __forceinline__ void push(unsigned s8) {
    c8 = pre + (s8 << 6);
    pre = (c8 >> 2) & ~7u;
    asm volatile("" : "+s"(pre));  // worked out HERE, in the shadow of the entry read (volatile asm keeps its place among the walk's other asm), not sunk into the next step's chain
  }
  // one base of the history in front of the first position of a walk (no scalar-register pin: see uniform())
  __device__ __forceinline__ void seed(unsigned s8) { c8 = pre + (s8 << 6); pre = (c8 >> 2) & ~7u; }
  // the walk keeps the history in scalar registers: what was seeded from memory is made provably uniform
  __device__ __forceinline__ void uniform() { c8 = fq_uniform(c8); pre = fq_uniform(pre); }
};
template <> struct CtxHist<QualModel> {
  unsigned q, q1, pre;  // 8 * symbols k-1, k-2; pre = what symbols k-1 and k-2 give the context of symbol k + 1
  __device__ __forceinline__ void start() { q = q1 = 0; pre = cur_slot = 1u << 15; }
  __device__ __forceinline__ unsigned slot() const { return cur_slot; }
  unsigned cur_slot;
  // calcContext (fq_qual_ctx), times 8: (max(q1, q2) << 6) | q | (q1 == q2) << 15 with q = the new symbol
  __device__ __forceinline__ unsigned next(unsigned s8) const { return pre | s8; }
  __device__ __forceinline__ void push(unsigned s8) {
    cur_slot = pre | s8;
    q1 = q; q = s8;
    pre = ((q > q1 ? q : q1) << 6) | ((unsigned)(q == q1) << 15);
    asm volatile("" : "+s"(pre));  // (see above)
  }
  // history in front of the first position of a walk: symbols k-1, k-2, k-3 (8 * symbol each)
  __device__ __forceinline__ void set(unsigned a, unsigned b, unsigned c) {
    q = a; q1 = b;
    cur_slot = ((b > c ? b : c) << 6) | a | ((unsigned)(b == c) << 15);
    pre = ((a > b ? a : b) << 6) | ((unsigned)(a == b) << 15);
  }
  __device__ __forceinline__ void uniform() {
    q = fq_uniform(q); q1 = fq_uniform(q1);
    pre = fq_uniform(pre); cur_slot = fq_uniform(cur_slot);
  }
};

// What a stream's walk carries from read to read: the per-context entries (LDS) and the table.
// Contexts are named by their slot: 8 * context = the byte offset of their CtxEntry.
//
// COMPACT: the entries alone in LDS (4 bytes per context: 32 KB for the quality stream, four
// workgroups per CU instead of two), the context's table offset read from dt_off by a SCALAR load
// one symbol ahead (it waits in the shadow of the next entry's LDS read).  Five more instructions
// per symbol and twice the places: for batches with more chains than places.
constexpr uint32_t FQ_ENTRY_PENDING = 0xFFFFFFFFu;  // (nbBits 15: no entry)
typedef __attribute__((address_space(3))) char lds_char;
template <bool COMPACT>
struct WalkT {
  lds_char *ce;            // CtxEntry[B] (COMPACT: u32[B]) at LDS byte address ce_lds (< 64 KB: the DMA's M0 offset)
  unsigned ce_lds;
  const uint32_t *dt, *dt_off;
  using Slots = typename std::conditional<COMPACT, uint32_t, CtxEntry>::type;
  __device__ __forceinline__ void init(Slots *lds, const uint32_t *tables, const uint32_t *table_offsets) {
    ce = (lds_char *)(__attribute__((address_space(3))) Slots *)lds;
    ce_lds = fq_uniform((unsigned)(size_t)ce);
    dt = tables;
    dt_off = table_offsets;
  }
  // LDS byte offset of a slot
  static __device__ __forceinline__ unsigned lds_of(unsigned slot) { return COMPACT ? slot >> 1 : slot; }
  __device__ __forceinline__ lds_u32 *entry_at(unsigned slot) const { return reinterpret_cast<lds_u32 *>(ce + lds_of(slot)); }
  __device__ __forceinline__ void mark(unsigned slot) const { *entry_at(slot) = FQ_ENTRY_PENDING; }
  // {entry, table byte offset} (one ds_read_b64); COMPACT: {entry, -}
  __device__ __forceinline__ uint2 load(unsigned slot) const {
    if (COMPACT) return make_uint2(*entry_at(slot), 0u);
    const lds_CtxEntry *e = reinterpret_cast<const lds_CtxEntry *>(ce + slot);
    return make_uint2(e->entry, e->table);
  }
  // COMPACT: dt_off[context of slot], asked for now (scalar load: it must not touch vmcnt, which the
  // refills own), valid behind settle()
  __device__ __forceinline__ uint32_t ask_table(unsigned slot) const {
    uint32_t t = 0;
    if (COMPACT) asm volatile("s_load_dword %0, %1, %2" : "=s"(t) : "s"(dt_off), "s"(fq_uniform(slot >> 1)) : "memory");
    return t;
  }
  static __device__ __forceinline__ void settle(uint32_t &t) {
    if (COMPACT) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(t) : : "memory");
  }
  // entry of `slot` <- the dword at byte offset `off` of the tables, asynchronously, behind
  // mark(slot): lane 0 alone issues (EXEC = 1; the walk runs with all 64 lanes); the LDS destination
  // of an LDS-DMA load is M0 + 4 * lane id.
  // lgkmcnt(1): LDS operations complete in issue order and the load() of the next entry was issued
  // behind the mark (the compiler cannot swap the two: they may alias), so at most that load, or
  // one behind it, is still on its way -- the mark has landed.  (COMPACT: the scalar load asked for
  // in between may return out of order: lgkmcnt(0).)
  // M0 is not restored and cannot be declared clobbered (hipcc: "reserved register"): nothing else in these
  // kernels uses it but the walk's v_writelane, which sets it itself -- no other LDS-DMA, no movrel --, which
  // tests/test_build_invariants.py checks on the ISA.
  // EXEC is set back to all ones, not restored: the walk runs with the full wave (decode_stream / decode_chunk
  // check that on entry and refuse to run otherwise).
  __device__ __forceinline__ void refill(unsigned slot, unsigned off) const {
    const unsigned dst = fq_uniform(ce_lds + lds_of(slot));
    if (COMPACT)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 m0, %2\n\ts_mov_b64 exec, 1\n\t"
                   "global_load_lds_dword %0, %1\n\ts_mov_b64 exec, -1"
                   : : "v"(off), "s"(dt), "s"(dst) : "memory");
    else
      asm volatile("s_waitcnt lgkmcnt(1)\n\ts_mov_b32 m0, %2\n\ts_mov_b64 exec, 1\n\t"
                   "global_load_lds_dword %0, %1\n\ts_mov_b64 exec, -1"
                   : : "v"(off), "s"(dt), "s"(dst) : "memory");
  }
  // the table dword at byte offset `off`, now (scalar load; waits for it)
  __device__ __forceinline__ uint32_t fetch(unsigned off) const {
    uint32_t v;
    asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(dt), "s"(off) : "memory");
    return v;
  }
  // entry of the context from what load(slot) returned; a pending entry is waited for
  __device__ __forceinline__ uint32_t take(unsigned slot, uint2 &e) const {
    unsigned unused = 0;
    return take(slot, e, ~0u, unused);
  }
  // ... `left` = the slot the walk has just left: coming straight back to it is counted in `runs`
#ifdef FQGPU_EXPERIMENTS
  mutable unsigned n_pending = 0, n_self = 0;  // slow paths taken, self-following contexts among them
  mutable unsigned long long t_pending = 0;     // shader clocks spent in them
#endif
  // (e.x is kept equal to the entry returned: the refill's address is worked out from the vector register)
  __device__ __forceinline__ uint32_t take(unsigned slot, uint2 &e, unsigned left, unsigned &runs) const {
    uint32_t entry = fq_uniform(e.x);
    if (__builtin_expect(entry == FQ_ENTRY_PENDING, 0)) {
#ifdef FQGPU_EXPERIMENTS
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      n_pending++;
      n_self += slot == left;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      entry = fq_uniform(*entry_at(slot));
      t_pending += __builtin_amdgcn_s_memtime() - t0;
      runs += slot == left;
      e.x = entry;
      return entry;
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      entry = fq_uniform(*entry_at(slot));
      runs += slot == left;
      e.x = entry;
    }
    return entry;
  }
};

// four symbols, one per byte -> their four output bytes
template <class M>
__device__ __forceinline__ unsigned fq_sym_bytes(unsigned acc) {
  return M::STREAM == 0 ? __builtin_amdgcn_perm(0u, 0x54474341u, acc)  // "ACGT"[sym] per byte
                        : acc + 0x21212121u;                             // Phred + 33
}

// Positions [i0, i1) of one read into out[i0 ..], all 64 lanes of the wave; h = history in front of
// position i0.  Everything the walk decides on (entry, symbol, context, bit position) is wave-uniform
// and kept in scalar registers.  Lane k keeps output bytes 4k .. 4k + 3 of the current 256.
template <class M, bool COMPACT>
__device__ __forceinline__ void walk_positions(WalkT<COMPACT> &wk, LdsBits &br, g_u8 *out, unsigned i0, unsigned i1, CtxHist<M> h) {
  i0 = fq_uniform(i0);  // (uniform by construction; said so, so that the trip counts and with them the
  i1 = fq_uniform(i1);  //  walk's whole state stay in scalar registers)
  if (i0 >= i1) return;
  const unsigned lane = threadIdx.x;
  unsigned slot = h.slot();
  uint32_t toff = wk.ask_table(slot);  // (COMPACT) dt_off of the current context
  uint2 e = wk.load(slot);
  uint32_t cur = wk.take(slot, e);
  wk.settle(toff);
  // ebase = new state + table of the entry in hand, from the entry's own vector register: the upper half picked by
  // the operand selector of one add (the scalar shift of `cur` and the shift of the bits become one shift-add with
  // it); first thing behind the next entry's read, well in front of its use
  unsigned ebase = 0;
  auto set_ebase = [&] {
    if (!COMPACT) ebase = (e.x >> 16) + e.y;  // (v_add_u32_sdwa ... src0_sel:WORD_1: the compiler's own choice once the entry counts as a vector value)
  };
  unsigned keep = 0, acc = 0;
  // One symbol; j = its byte in acc.  ORDER IS THE POINT: the chain of a stream is entry -> symbol ->
  // next context -> LDS read of that context's entry, and a lone wave issues one instruction every four
  // cycles or so -- whatever stands between `cur` and the issue of that read delays the next symbol.
  // So the mark of the slot being left (its address sits in a register since the last step) and the
  // read of the next entry go first, pinned by a scheduling barrier; everything else -- this symbol's
  // bits (scalar), the refill's address, the output byte, the half of the NEXT context that does not
  // depend on the next symbol -- runs while that read is on its way.  Entry fields: FQ_DENTRY.
  //
  // RUN = the form of the step that also knows what to do when a context FOLLOWS ITSELF (a run of one
  // quality value, a homopolymer).  Through the slot that costs a wait for this very symbol's refill
  // -- an L2 round trip plus an LDS one per symbol: 160 ns on binned or constant data --, so there
  // the next entry, dt[table + new state], is fetched directly by a scalar load (the tables are
  // read-only) and the slot is left alone, stale, until the context is left: the step that leaves it
  // marks and refills the slot from `cur` as always.  The test and its branch cost every symbol a few
  // cycles and the registers their copies, so the plain form runs until its slow path has met such
  // a context (`runs`), and the RUN form as long as it keeps meeting one: the switch is made between
  // groups of four symbols.
  unsigned runs = 0;   // self-following contexts met in the current group of four (counted twice in the RUN form)
  unsigned tbl_s = 0;  // RUN form: table of the current context (scalar)
  bool tail = false;  // the last one to three symbols of a walk: every read looks after the window
  auto step = [&](unsigned j, auto run_form) {
    constexpr bool RUN = decltype(run_form)::value;
    const unsigned s8 = cur & (unsigned)((M::A - 1) << 3);
    const unsigned nslot = h.next(s8);
    if (RUN && nslot == slot) {
      const unsigned nb = (cur >> 9) & 15u;
      const unsigned bits = br.read(nb, (j & 1u) != 0 || tail);
      h.push(s8);
      acc |= (s8 >> 3) << (8u * j);
      cur = wk.fetch((cur >> 16) + (bits << 2) + tbl_s);
      e.x = cur;
      runs += 2;
      return;
    }
    wk.mark(slot);
    const uint2 ne = wk.load(nslot);
    uint32_t ntoff = wk.ask_table(nslot);
    __builtin_amdgcn_sched_barrier(0);
    set_ebase();
    const unsigned nb = (cur >> 9) & 15u;
    const unsigned bits = br.read(nb, (j & 1u) != 0 || tail);
    h.push(s8);
    // (e.y, the table's byte offset, stays in its vector register; COMPACT: the state-0 entry is word dt_off + 1)
    unsigned off;
    if (COMPACT) off = (cur >> 16) + ((bits + (toff + 1u)) << 2);
    else {
      off = (bits << 2) + ebase;
    }
    acc |= (s8 >> 3) << (8u * j);
    wk.refill(slot, off);
    const unsigned left = slot;
    slot = nslot;
    e = ne;
    cur = wk.take(slot, e, left, runs);  // (one too many at the end of a read: the next read starts from another context)
    wk.settle(ntoff);
    toff = ntoff;
    if (RUN) tbl_s = COMPACT ? (toff + 1u) << 2 : fq_uniform(e.y);  // byte offset of the context's table, for the direct fetch
  };
  g_u8 *o = out + i0;
  const unsigned n = i1 - i0;
  unsigned g = 0;  // symbols done
  auto group_done = [&] {
    const unsigned kslot = (g >> 2) & 63u;
    // lane kslot of keep <- the four bytes (two instructions; compare, move, wait state and select are four)
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(keep) : "s"(fq_uniform(fq_sym_bytes<M>(acc))), "s"(fq_uniform(kslot)));
    if (__builtin_expect(kslot == 63u, 0)) *reinterpret_cast<FQ_GLOBAL uint32_t *>(o + (g - 252u) + 4u * lane) = keep;  // (reads start anywhere: unaligned dwords)
  };
  const unsigned n4 = n & ~3u;
  // (one loop condition: the switch of forms moves the loop's end -- `stop` -- to where the walk stands)
  while (g < n4) {
    for (unsigned stop = n4; g < stop;) {  // the plain form, until a group meets two such contexts
      acc = 0;
      runs = 0;
      step(0, std::false_type()); step(1, std::false_type()); step(2, std::false_type()); step(3, std::false_type());
      group_done();
      g += 4;
      stop = runs >= 2 ? g : n4;
      asm volatile("" : "+s"(stop));  // (kept a number: the compiler would make two lane masks and their conjunction of it again)
    }
    if (g < n4) tbl_s = COMPACT ? (toff + 1u) << 2 : fq_uniform(e.y);
    for (unsigned stop = n4; g < stop;) {  // the form for runs, as long as every group has one
      acc = 0;
      runs = 0;
      step(0, std::true_type()); step(1, std::true_type()); step(2, std::true_type()); step(3, std::true_type());
      group_done();
      g += 4;
      stop = runs >= 2 ? n4 : g;
      asm volatile("" : "+s"(stop));
    }
  }
  const unsigned r = n - g;  // < 4 symbols left
  acc = 0;
  tbl_s = COMPACT ? (toff + 1u) << 2 : fq_uniform(e.y);
  tail = true;
  for (unsigned j = 0; j < r; j++) step(j, std::true_type());
  // a read that ends inside a run leaves the run's context with a stale slot (the self-following path above
  // does not touch it): the entry the context is in goes there now (anywhere else this writes what is there)
  *wk.entry_at(slot) = cur;
  const unsigned full = (g >> 2) & 63u;  // whole dwords not stored yet
  const unsigned base = g - 4u * full;
  if (lane < full) *reinterpret_cast<FQ_GLOBAL uint32_t *>(o + base + 4u * lane) = keep;
  if (lane < r) o[g + lane] = (uint8_t)(fq_sym_bytes<M>(acc) >> (8u * lane));
}

// the entry (and, resident form, the table offset) of context c at the start of a walk
__device__ __forceinline__ void set_slot(CtxEntry *ce, unsigned c, uint32_t table, uint32_t entry) { ce[c].table = table * 4u; ce[c].entry = entry; }
__device__ __forceinline__ void set_slot(uint32_t *ce, unsigned c, uint32_t, uint32_t entry) { ce[c] = entry; }

template <class M, bool COMPACT>
__device__ void decode_stream(const DecJob &j, const TabView &tab, typename WalkT<COMPACT>::Slots *ce, uint32_t *bitbuf) {
  constexpr unsigned B = M::B;
  const uint8_t *src = M::STREAM == 0 ? j.seq : j.qual;
  const unsigned len = M::STREAM == 0 ? j.seq_len : j.qual_len;
  StreamResult *res = &j.res->s[M::STREAM];
  const unsigned lane = threadIdx.x;
  g_cu32 *w = (g_cu32 *)reinterpret_cast<const uint32_t *>(src);
  g_crec *recs = (g_crec *)j.recs;
  g_u8 *raw = (g_u8 *)j.raw;

  if (__ballot(1) != ~0ull) { res->corrupt = 1; return; }  // (the refill's assembly sets EXEC to all ones: a partial wave must not get there)
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
    const uint32_t table = tab.dt_off[c] + 1u;  // behind the table's header word
    set_slot(ce, c, table, tab.dt[table + peek_bits(w, lo, lg)]);
  }
  __syncthreads();
  WalkT<COMPACT> wk;
  wk.init(ce, tab.dt, tab.dt_off);

  LdsBits br;
  br.init(w, p0 - (long long)sum_logs, bitbuf, (len + 3) / 4);
  fqgpu_rec nxt;
  nxt.seq_off = recs[j.n_recs - 1].seq_off; nxt.qual_off = recs[j.n_recs - 1].qual_off; nxt.len = recs[j.n_recs - 1].len;
  for (unsigned r = j.n_recs; r > 0; r--) {  // records last -> first (src/workspace.cpp:84-87)
    const fqgpu_rec rec = nxt;
    if (r > 1) { nxt.seq_off = recs[r - 2].seq_off; nxt.qual_off = recs[r - 2].qual_off; nxt.len = recs[r - 2].len; }  // lands while this record is walked
    CtxHist<M> h;
    h.start();
    walk_positions<M, COMPACT>(wk, br, raw + (M::STREAM == 0 ? rec.seq_off : rec.qual_off), 0u, rec.len, h);
    if (br.underflow) break;
  }
  // BIT_endOfDStream (src/fse_common.hpp:141): every bit consumed, none invented
  if (lane == 0) {
    if (br.pos() != 0) res->corrupt = 1;
    res->total_bits = (unsigned long long)(p0 - (long long)sum_logs);
  }
#ifdef FQGPU_EXPERIMENTS
  // (make experiments: how often the walk finds an entry pending, and what that costs)
  if (lane == 0 && !COMPACT)
    printf("walk stats stream %d: %u records, entries found pending %u (the context following itself: %u), shader clocks in the slow path %llu\n",
           (int)M::STREAM, j.n_recs, wk.n_pending, wk.n_self, wk.t_pending);
#endif
}

// One stride of one stream, started from a snapshot of the decode index (or from the end of the
// stream for the last stride): encode indices [e_lo, e_hi) in decoder order, i.e. from the record
// and position of symbol e_hi - 1 towards the front of the block.
template <class M, bool COMPACT>
__device__ void decode_chunk(const DecJob &j, unsigned chunk, const TabView &tab, typename WalkT<COMPACT>::Slots *ce, uint32_t *bitbuf) {
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
  if (__ballot(1) != ~0ull) { res->corrupt = 1; return; }  // (see decode_stream)

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
      const uint32_t table = tab.dt_off[c] + 1u;
      set_slot(ce, c, table, tab.dt[table + peek_bits(w, lo, tab.logs[c])]);
    }
    pos = p0 - (long long)sum_logs;
    if (lane == 0) res->total_bits = (unsigned long long)pos;
  } else {
    const uint8_t *snap = snaps + (size_t)chunk * snap_bytes;  // snapshot chunk + 1 sits at e_hi
    const uint16_t *st = reinterpret_cast<const uint16_t *>(snap + FQ_INDEX_SNAP_HEAD);
    for (unsigned c = lane; c < B; c += 64) {
      const uint32_t table = tab.dt_off[c] + 1u;
      set_slot(ce, c, table, tab.dt[table + ((unsigned)st[c] & ((1u << tab.logs[c]) - 1u))]);  // a damaged index must not leave the table
    }
    pos = (long long)*reinterpret_cast<const unsigned long long *>(snap);
    prev = fq_uniform(reinterpret_cast<const uint32_t *>(snap)[2]);  // (uniform: the walk's history lives in scalar registers)
    if (pos > (long long)len * 8) { if (lane == 0) res->corrupt = 1; return; }
  }
  __syncthreads();
  // every bit of this stride consumed, none invented: the walk must end where the previous
  // snapshot (or the start of the stream) says
  const long long pos_end = chunk == 0 ? 0ll : (long long)*reinterpret_cast<const unsigned long long *>(snaps + (size_t)(chunk - 1) * snap_bytes);

  LdsBits br;
  br.init(w, pos, bitbuf, (len + 3) / 4);
  WalkT<COMPACT> wk;
  wk.init(ce, tab.dt, tab.dt_off);
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
          if (ch != 0xFFu) h.seed(fq_base_code(ch) * 8u);
        }
      } else {
        const unsigned a = prev & 0xFFu, b = (prev >> 8) & 0xFFu, c = (prev >> 16) & 0xFFu;
        h.set(a != 0xFFu ? ((a - 33u) & 63u) * 8u : 0u, b != 0xFFu ? ((b - 33u) & 63u) * 8u : 0u,
              c != 0xFFu ? ((c - 33u) & 63u) * 8u : 0u);
      }
    }
    h.uniform();
    walk_positions<M, COMPACT>(wk, br, raw + (M::STREAM == 0 ? rec.seq_off : rec.qual_off), i0, i1, h);
    first = false;
    if (br.underflow || rs <= e_lo || r == 0) break;
    r--;
  }
  if (lane == 0 && br.pos() != pos_end) res->corrupt = 1;
}

// one workgroup per stride; M's chunks only: the two streams are launched separately because their
// LDS footprints differ by a factor of 16 (quality: 64 KB of entries, two workgroups per CU; sequence: 2 KB)
// -- in one kernel the sequence strides would take the quality strides' places
template <class M, bool COMPACT>
__global__ void __launch_bounds__(64)
k_decode_chunks(const DecJob *__restrict__ jobs, const DecChunk *__restrict__ chunks, TabView tab) {
  __shared__ typename WalkT<COMPACT>::Slots ce[M::B];
  __shared__ uint32_t bitbuf[FQ_BITBUF_DW];
  const DecChunk ch = chunks[blockIdx.x];
  decode_chunk<M, COMPACT>(jobs[ch.job], ch.chunk, tab, ce, bitbuf);
}

// record lengths of one block, for the encode index of the first symbol of every record
__global__ void __launch_bounds__(256)
k_lens_of(const fqgpu_rec *__restrict__ recs, unsigned n, uint32_t *__restrict__ lens32) {
  const unsigned r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) lens32[r] = recs[r].len;
}

// one workgroup per block: stream M of every block of the batch
template <class M, bool COMPACT>
__global__ void __launch_bounds__(64)
k_decode(const DecJob *__restrict__ jobs, TabView tab) {
  __shared__ typename WalkT<COMPACT>::Slots ce[M::B];
  __shared__ uint32_t bitbuf[FQ_BITBUF_DW];
  decode_stream<M, COMPACT>(jobs[blockIdx.x], tab, ce, bitbuf);
}
// both streams in one launch (grid = 2 * n_blocks, the quality streams first): for batches whose
// chains all find a place at once (two workgroups of 66 KB per CU) -- then the placement of one
// launch, a quality and a sequence chain per CU, is the better one (256 x 1 MiB blocks: 5.0 against 4.6 GB/s)
__global__ void __launch_bounds__(64)
k_decode_both(const DecJob *__restrict__ jobs, unsigned n_blocks, TabView seq_tab, TabView qual_tab) {
  __shared__ CtxEntry ce[QualModel::B];
  __shared__ uint32_t bitbuf[FQ_BITBUF_DW];
  if (blockIdx.x < n_blocks) decode_stream<QualModel, false>(jobs[blockIdx.x], qual_tab, ce, bitbuf);
  else decode_stream<SeqModel, false>(jobs[blockIdx.x - n_blocks], seq_tab, ce, bitbuf);
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
  std::vector<DecChunk> chunks, seq_chunks;  // quality strides | sequence strides
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
  size_t n_qual_chunks = 0;
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
      for (size_t k = snaps_of(b, 1) + 1; k-- > 0;) chunks.push_back(DecChunk{(unsigned)i, 1u, (unsigned)k});
      for (size_t k = snaps_of(b, 0) + 1; k-- > 0;) seq_chunks.push_back(DecChunk{(unsigned)i, 0u, (unsigned)k});
    }
    n_qual_chunks = chunks.size();
    chunks.insert(chunks.end(), seq_chunks.begin(), seq_chunks.end());
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
  // the sequence streams run beside the quality streams on a second stream
  if (!ctx->dec_stream2) {
    FQ_HIP(hipStreamCreateWithFlags(&ctx->dec_stream2, hipStreamNonBlocking));
    FQ_HIP(hipEventCreateWithFlags(&ctx->dec_fork, hipEventDisableTiming));
    FQ_HIP(hipEventCreateWithFlags(&ctx->dec_join, hipEventDisableTiming));
  }
  hipStream_t st2 = ctx->dec_stream2;
  fq_timer_span_begin(ctx, "decode", st);
  FQ_HIP(hipEventRecord(ctx->dec_fork, st));
  FQ_HIP(hipStreamWaitEvent(st2, ctx->dec_fork, 0));
  const DecChunk *dch = ctx->dec_chunks.as<DecChunk>();
  if (n_plain && 2 * n_plain <= 2 * (size_t)ctx->n_cus) {
    hipLaunchKernelGGL(k_decode_both, dim3((unsigned)(2 * n_plain)), dim3(64), 0, st, jobs, (unsigned)n_plain, ts, tq);
  } else if (n_plain) {
    // more quality chains than places for the resident form (two 66 KB workgroups per CU): the compact form
    if (n_plain > 2 * (size_t)ctx->n_cus) hipLaunchKernelGGL((k_decode<QualModel, true>), dim3((unsigned)n_plain), dim3(64), 0, st, jobs, tq);
    else hipLaunchKernelGGL((k_decode<QualModel, false>), dim3((unsigned)n_plain), dim3(64), 0, st, jobs, tq);
    hipLaunchKernelGGL((k_decode<SeqModel, false>), dim3((unsigned)n_plain), dim3(64), 0, st2, jobs, ts);
  }
  if (n_qual_chunks > 2 * (size_t)ctx->n_cus) hipLaunchKernelGGL((k_decode_chunks<QualModel, true>), dim3((unsigned)n_qual_chunks), dim3(64), 0, st, jobs, dch, tq);
  else if (n_qual_chunks) hipLaunchKernelGGL((k_decode_chunks<QualModel, false>), dim3((unsigned)n_qual_chunks), dim3(64), 0, st, jobs, dch, tq);
  if (chunks.size() > n_qual_chunks)
    hipLaunchKernelGGL((k_decode_chunks<SeqModel, false>), dim3((unsigned)(chunks.size() - n_qual_chunks)), dim3(64), 0, st2, jobs, dch + n_qual_chunks, ts);
  FQ_HIP(hipEventRecord(ctx->dec_join, st2));
  FQ_HIP(hipStreamWaitEvent(st, ctx->dec_join, 0));
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
