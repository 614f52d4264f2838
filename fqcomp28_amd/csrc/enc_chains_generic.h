// Part of encode.hip (included there, inside its anonymous namespace): K4: reset-aware segment kernels for any table set (quality stream).

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
// class of a segment (SegArrays::cls): 0 = opaque, 1 .. 64 = anchored (the state behind its anchor
// symbol is one of that many candidates; 1 = reset symbol: transparent), 0xFF = uniform (S times
// one symbol whose count is too large to be an anchor)
constexpr unsigned SEG_CLS_OPAQUE = 0, SEG_CLS_UNIFORM = 0xFF, SEG_MAX_CAND = 64;
constexpr unsigned SEG_SLOT = 16;  // lanes of one candidate walk: four walks share a wave

// segment table of one stream (all arrays indexed by the global segment number)
struct SegArrays {
  uint32_t *anchor;       // offset of the anchor symbol inside the segment (cls 1 .. 64), the symbol (uniform), or SEG_NONE
  uint32_t *usym;         // [B] the symbol whose power table the context owns in this block, or SEG_NONE
  uint16_t *entry_state;  // state in front of the first symbol of every segment
  uint16_t *cand_exit;    // [segment][64] state at the end of the segment for every candidate of its anchor
  uint8_t *cls;           // class of every segment
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

// One wave per segment.  First reset symbol (stops there: class 1); otherwise the first occurrence
// of the symbol with the fewest table cells if that is <= 64 (class = that number: after this
// symbol the state is one of so many known candidates); otherwise uniform or opaque.
template <class M>
__global__ void __launch_bounds__(256)
k_seg_scan(const uint8_t *__restrict__ sorted_sym, const uint32_t *__restrict__ arrays,
           const unsigned long long *__restrict__ reset_mask, const int16_t *__restrict__ norm,
           const uint32_t *__restrict__ logs, unsigned S, SegArrays sa) {
  constexpr unsigned B = M::B, A = M::A;
  __shared__ uint8_t s_cells[4][64];
  const uint32_t *ctx_count = arrays, *ctx_start = arrays + B, *seg_base = ctx_start + B + 1;
  const unsigned wave = threadIdx.x >> 6, seg = blockIdx.x * 4 + wave;
  if (seg >= seg_base[B]) return;  // the grid is an upper bound
  const unsigned lane = fq_lane();
  const unsigned c = seg_ctx_of<M>(seg_base, seg), k = seg - seg_base[c];
  const unsigned n = ctx_count[c], begin = k * S, end = min(n, begin + S);
  const unsigned long long mask = reset_mask[c];
  const uint8_t *sym = sorted_sym + ctx_start[c];
  {  // table cells of every symbol (255: too many for an anchor, or none)
    const int v = lane < A ? (int)norm[(size_t)c * A + lane] : 0;
    s_cells[wave][lane] = (uint8_t)(v == -1 ? 1 : (v >= 1 && v <= (int)SEG_MAX_CAND) ? v : 255);
    fq_lds_wave_sync();
  }
  const uint8_t *cells = s_cells[wave];
  unsigned found = SEG_NONE;             // first reset symbol
  unsigned best = (255u << 20) | 0xFFFFFu;  // cells << 20 | offset of the best anchor so far
  bool uniform = end - begin == S;       // (a short last segment needs no function)
  unsigned s0 = 0;
  for (unsigned b0 = begin; b0 < end; b0 += 1024) {
    const unsigned p = b0 + 16 * lane;
    unsigned hit = 16, mine = 0xFFFFFFFFu;
    bool same = true;
    if (p < end) {  // the run is padded to 16 bytes: whole-group loads stay inside it
      const uint4 v = *reinterpret_cast<const uint4 *>(sym + p);
      const unsigned w[4] = {v.x, v.y, v.z, v.w};
      if (b0 == begin) s0 = (unsigned)__builtin_amdgcn_readfirstlane(v.x) & (A - 1);
#pragma unroll
      for (int j = 15; j >= 0; j--) {
        const unsigned s = (w[j >> 2] >> (8 * (j & 3))) & (A - 1);
        if (p + j < end) {
          if ((mask >> s) & 1ull) hit = (unsigned)j;
          const unsigned off = b0 - begin + 16 * lane + (unsigned)j;  // (anchors are looked for in the first 2^20 symbols)
          if (off < (1u << 20)) mine = min(mine, ((unsigned)cells[s] << 20) | off);
          same = same && s == s0;
        }
      }
    }
    const unsigned long long any = __ballot(hit < 16);
    if (any) {
      const unsigned l0 = (unsigned)__ffsll((long long)any) - 1u;
      found = (b0 - begin) + 16 * l0 + (unsigned)__shfl((int)hit, (int)l0);
      break;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mine = min(mine, (unsigned)__shfl_xor((int)mine, d));
    best = min(best, mine);
    uniform = uniform && __ballot(!same) == 0ull;
  }
  if (lane == 0) {
    unsigned cls = SEG_CLS_OPAQUE, anchor = SEG_NONE;
    if (found != SEG_NONE) { cls = 1; anchor = found; }
    else if ((best >> 20) <= SEG_MAX_CAND) { cls = best >> 20; anchor = best & 0xFFFFFu; }
    else if (uniform) {  // one power table per context and block: the first symbol to ask gets it
      unsigned old = __hip_atomic_load(&sa.usym[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (a chain of uniform segments: one CAS, not thousands on one word)
      if (old == SEG_NONE) old = atomicCAS(&sa.usym[c], SEG_NONE, s0);
      if (old == SEG_NONE || old == s0) { cls = SEG_CLS_UNIFORM; anchor = s0; }
    }
    sa.cls[seg] = (uint8_t)cls;
    sa.anchor[seg] = anchor;
    if (k == 0) sa.entry_state[seg] = (uint16_t)(1u << logs[c]);  // FSE_initCState
  }
}

// Who computes the function of a segment whose successor needs one:
//   opaque                          -> k_seg_setfunc, slot = the segment
//   uniform                         -> the context's power table (slot pbase + context)
//   anchored, 2 .. 64 candidates    -> k_seg_cand<false> walks every candidate to the end of the segment;
//       k_seg_cand<true> walks every state the segment can be entered with up to the anchor -- possible
//       when the predecessor is anchored too (its candidates' exits) or the chain starts here;
//       behind an opaque or uniform segment the entry state can be anything: k_seg_setfunc
//   reset symbol (1 candidate)      -> nothing: k_seg_walk<1> knows the exit state
// seg_wants_*: the same decisions for the kernels that act on them (k: index in the chain).
__device__ __forceinline__ bool seg_is_cand(unsigned cls) { return cls >= 2 && cls <= SEG_MAX_CAND; }
__device__ __forceinline__ bool seg_entry_known(unsigned k, unsigned pred_cls) {  // ... to be one of <= 64 states
  return k == 0 || pred_cls == 1 || seg_is_cand(pred_cls);
}

// function slot of segment k of a chain of nseg (slot = the segment itself; the power tables follow
// behind the last segment's slot), or SEG_NONE: reset symbol inside, or nothing follows
__device__ __forceinline__ unsigned seg_fslot(const SegArrays &sa, unsigned seg, unsigned k, unsigned nseg, unsigned c,
                                              unsigned pbase) {
  if (k + 1 >= nseg) return SEG_NONE;
  const unsigned cls = sa.cls[seg];
  return cls == 1 ? SEG_NONE : cls == SEG_CLS_UNIFORM ? pbase + c : seg;
}
// does k_seg_setfunc compute the function of this segment?
__device__ __forceinline__ bool seg_wants_setfunc(const SegArrays &sa, unsigned seg, unsigned k, unsigned nseg) {
  if (k + 1 >= nseg) return false;
  const unsigned cls = sa.cls[seg];
  return cls == SEG_CLS_OPAQUE || (seg_is_cand(cls) && !seg_entry_known(k, k ? (unsigned)sa.cls[seg - 1] : 1u));
}

// T_s^S for the context's uniform symbol s: the function of a segment that is S times s, by
// square and multiply over the bits of S.  Slot pbase + context of the function buffer.  One wave
// (a role of k_seg_setfunc's launch): the running power in registers, the base in LDS.
template <class M, unsigned PER0>
__device__ __forceinline__ void seg_pow_table(uint32_t *lds, const uint32_t *__restrict__ tbl, unsigned s, unsigned S,
                                              uint16_t *__restrict__ f) {
  const unsigned log = tbl[0] & 0xFFFFu, size = 1u << log, lane = fq_lane(), per = max(size >> 6, 1u);
  const uint16_t *st = reinterpret_cast<const uint16_t *>(tbl) + 2;
  const uint32_t *tt = tbl + 1 + (size >> 1);
  const int dfs = (int)tt[2 * s];
  const unsigned dnb = tt[2 * s + 1];
  uint16_t *base = reinterpret_cast<uint16_t *>(lds);  // 2^log entries: fits the CTable's staging area
  unsigned acc[PER0], tmp[PER0];
#pragma unroll
  for (unsigned j = 0; j < PER0; j++) {
    const unsigned i = (lane + 64u * j) & (size - 1);
    acc[j] = i;
    if (j < per) {
      const unsigned x = size + i, nb = (x + dnb) >> 16;
      base[i] = (uint16_t)(st[(int)(x >> nb) + dfs] - size);
    }
  }
  fq_lds_wave_sync();
  for (unsigned e = S; e; e >>= 1) {
    if (e & 1u) {
#pragma unroll
      for (unsigned j = 0; j < PER0; j++)
        if (j < per) acc[j] = base[acc[j]];
    }
    if (e > 1u) {
#pragma unroll
      for (unsigned j = 0; j < PER0; j++)
        if (j < per) tmp[j] = base[base[(lane + 64u * j) & (size - 1)]];
      fq_lds_wave_sync();  // every lane has read the old base
#pragma unroll
      for (unsigned j = 0; j < PER0; j++)
        if (j < per) base[(lane + 64u * j) & (size - 1)] = (uint16_t)tmp[j];
      fq_lds_wave_sync();
    }
  }
#pragma unroll
  for (unsigned j = 0; j < PER0; j++) {
    const unsigned i = lane + 64u * j;
    if (j < per && i < size) f[i] = (uint16_t)(size + acc[j]);
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
//         all of it (every other class), from the resolved entry state.
template <class M, int PASS>
__device__ __forceinline__ void seg_walk_role(uint32_t *lds, unsigned item, const uint8_t *__restrict__ sorted_sym,
                                              uint16_t *__restrict__ out16, const uint32_t *__restrict__ arrays,
                                              const uint32_t *__restrict__ ct, const uint32_t *__restrict__ ct_off,
                                              uint16_t *__restrict__ final_state, unsigned S, const SegArrays &sa, StreamResult *res) {
  constexpr unsigned B = M::B;
  const uint32_t *ctx_count = arrays, *ctx_start = arrays + B, *seg_base = ctx_start + B + 1,
                 *item_base = seg_base + B + 1;
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
  const bool reset = sa.cls[seg] == 1;  // transparent: the segment holds a reset symbol
  const unsigned fr = sa.anchor[seg];
  unsigned x, i0, i1;
  if (PASS == 1) {
    if (!reset) return;
    i0 = begin + fr + 1; i1 = end;
    x = reset_state(t, sym[begin + fr] & (unsigned)(M::A - 1));
  } else {
    i0 = begin; i1 = reset ? begin + fr + 1 : end;
    x = sa.entry_state[seg];
  }
  x = seg_walk_range<M>(t, sym, out, i0, i1, x);
  if (PASS == 1 && k + 1 < nseg) sa.entry_state[seg + 1] = (uint16_t)x;
  if (k == nseg - 1 && (PASS == 1 || !reset)) final_state[c] = (uint16_t)x;
  if (PASS == 2 && fq_lane() == 0) atomicMax(&res->refixed, min(S, n));
}

template <class M, int PASS>
__global__ void __launch_bounds__(64)
k_seg_walk(const uint8_t *__restrict__ sorted_sym, uint16_t *__restrict__ out16,
           const uint32_t *__restrict__ arrays, const uint32_t *__restrict__ ct,
           const uint32_t *__restrict__ ct_off, uint16_t *__restrict__ final_state, unsigned S,
           SegArrays sa, StreamResult *res) {
  extern __shared__ uint32_t lds[];
  seg_walk_role<M, PASS>(lds, blockIdx.x, sorted_sym, out16, arrays, ct, ct_off, final_state, S, sa, res);
}

// ---- anchored segments without a reset symbol: candidate walks ------------------------------
// A symbol with n table cells leaves the coder in one of n states whatever it was in: cell j of the
// symbol, j = (x >> nb) - n (zstd fse.h: FSE_encodeSymbol lands on stateTable[(x >> nb) +
// deltaFindState], deltaFindState = first cell - n).  n = 1 is the reset symbol above; for
// n <= 64 one lane per candidate walks on from the anchor:
//   tails (a role of k_seg_stage1)  to the end of the segment: cand_exit[segment][j]
//   k_seg_heads  from every state the segment can be ENTERED with -- the predecessor's cand_exit,
//                or the one known entry state behind a reset symbol / at the start of the chain --
//                up to the anchor: which candidate that is, hence F[entry state] = cand_exit[that]
// F has 2^log entries like a function of k_seg_setfunc, but only the (at most 64) entries that can
// occur are written; k_seg_compose / k_seg_resolve2/3 treat it like any other function.  Sixteen
// lanes per walk and four walks per wave: a block of binned qualities costs n / 64 of a gather
// per symbol instead of the 1 .. 32 of the state sets.
// symbols [i, end) of a run walked from state x, nothing written; returns the final state
template <class M>
__device__ __forceinline__ unsigned seg_state_range(const LdsCTable &t, const uint8_t *__restrict__ sym, unsigned i,
                                                    unsigned end, unsigned x) {
  auto step = [&](unsigned s) {
    const int dfs = (int)t.tt[2 * s];
    const unsigned nb = (x + t.tt[2 * s + 1]) >> 16;
    x = t.state_table[(int)(x >> nb) + dfs];
  };
  while (i < end && (i & 15u)) { step(sym[i] & (unsigned)(M::A - 1)); i++; }
  if (i + 16 <= end) {
    const uint4 *sym16 = reinterpret_cast<const uint4 *>(sym);
    uint4 cur = sym16[i >> 4];
    while (i + 16 <= end) {
      const uint4 nxt = i + 32 <= end ? sym16[(i >> 4) + 1] : cur;
      const unsigned w[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
      for (int j = 0; j < 16; j++) step((w[j >> 2] >> (8 * (j & 3))) & (unsigned)(M::A - 1));
      i += 16;
      cur = nxt;
    }
  }
  while (i < end) { step(sym[i] & (unsigned)(M::A - 1)); i++; }
  return x;
}

// Four consecutive segments per wave (sixteen lanes each); they nearly always belong to one chain, so
// the wave stages ONE CTable at a time and serves the slots of that context (4.6 KB of LDS per wave
// instead of four tables: four times the waves per CU, and these walks are latency chains).
template <class M, bool HEADS>
__device__ __forceinline__ void seg_cand_role(uint32_t *lds, unsigned w_first, unsigned w_step, const uint8_t *__restrict__ sorted_sym,
                                              const uint32_t *__restrict__ arrays, const uint32_t *__restrict__ ct,
                                              const uint32_t *__restrict__ ct_off, unsigned S, unsigned fstride, const SegArrays &sa,
                                              uint16_t *__restrict__ fbuf) {
  constexpr unsigned B = M::B, PER = 64 / SEG_SLOT;
  const uint32_t *ctx_count = arrays, *ctx_start = arrays + B, *seg_base = ctx_start + B + 1;
  const unsigned n_segs = seg_base[B];
  const unsigned lane = fq_lane(), slot = lane / SEG_SLOT, j = lane % SEG_SLOT;
  for (unsigned w = w_first; w * PER < n_segs; w += w_step) {
    const unsigned seg = w * PER + slot;
    unsigned cls = 0, c = 0, k = 0, pred = 1;
    bool want = false;
    if (seg < n_segs) {
      cls = sa.cls[seg];
      if (seg_is_cand(cls)) {
        c = seg_ctx_of<M>(seg_base, seg);
        k = seg - seg_base[c];
        pred = k ? (unsigned)sa.cls[seg - 1] : 1u;
        want = k + 1 < seg_base[c + 1] - seg_base[c] && (!HEADS || seg_entry_known(k, pred));
      }
    }
    unsigned long long todo = __ballot(want);
    while (todo) {  // (uniform) one context at a time: usually one round
      const unsigned cc = (unsigned)__shfl((int)c, __ffsll((long long)todo) - 1);
      const bool mine = want && c == cc;
      todo &= ~__ballot(mine);
      fq_lds_wave_sync();  // the previous table is no longer read
      const LdsCTable t = stage_ctable<M>(lds, ct + ct_off[cc]);
      if (!mine) continue;
      const unsigned size = 1u << t.log;
      const unsigned n = ctx_count[c], begin = k * S, end = min(n, begin + S);
      const uint8_t *sym = sorted_sym + ctx_start[c];
      const unsigned r = begin + sa.anchor[seg];
      const unsigned s = sym[r] & (unsigned)(M::A - 1);
      const int dfs = (int)t.tt[2 * s];
      const unsigned n_cand = HEADS ? pred : cls;  // walks of this segment: sixteen per round
      for (unsigned cand = j; cand < n_cand; cand += SEG_SLOT) {
        if (!HEADS) {
          unsigned x = t.state_table[dfs + (int)cls + (int)cand];
          x = seg_state_range<M>(t, sym, r + 1, end, x);
          sa.cand_exit[(size_t)seg * SEG_MAX_CAND + cand] = (uint16_t)x;
        } else {
          const unsigned entry = pred == 1 ? (unsigned)sa.entry_state[seg] : (unsigned)sa.cand_exit[(size_t)(seg - 1) * SEG_MAX_CAND + cand];
          const unsigned x = seg_state_range<M>(t, sym, begin, r, entry);
          const unsigned nb = (x + t.tt[2 * s + 1]) >> 16;
          const unsigned u = (x >> nb) - cls;  // which cell of the anchor symbol
          fbuf[(size_t)seg * fstride + (entry - size)] = sa.cand_exit[(size_t)seg * SEG_MAX_CAND + u];
        }
      }
    }
  }
}

// the walks up to the anchors (behind k_seg_stage1: they start from its results)
template <class M>
__global__ void __launch_bounds__(64)
k_seg_heads(const uint8_t *__restrict__ sorted_sym, const uint32_t *__restrict__ arrays,
            const uint32_t *__restrict__ ct, const uint32_t *__restrict__ ct_off, unsigned S,
            unsigned fstride, SegArrays sa, uint16_t *__restrict__ fbuf) {
  extern __shared__ uint32_t lds[];
  seg_cand_role<M, true>(lds, blockIdx.x, gridDim.x, sorted_sym, arrays, ct, ct_off, S, fstride, sa, fbuf);
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

// F of one segment that needs a full function (seg_wants_setfunc); one wave per workgroup, its own
// copy of the context's CTable.  Workgroup = segment; the workgroups behind the last segment (pbase
// on) build the power tables of the contexts that have uniform segments in this block.
template <class M, unsigned PER0>
__device__ __forceinline__ void seg_setfunc_role(uint32_t *lds, SetsWaveLds &L, unsigned idx, const uint8_t *__restrict__ sorted_sym,
                                                 const uint32_t *__restrict__ arrays, const uint32_t *__restrict__ ct,
                                                 const uint32_t *__restrict__ ct_off, unsigned S, unsigned pbase, unsigned fstride,
                                                 const SegArrays &sa, uint16_t *__restrict__ fbuf) {
  constexpr unsigned B = M::B;
  const uint32_t *ctx_start = arrays + B, *seg_base = ctx_start + B + 1;
  if (idx >= pbase) {  // (uniform)
    const unsigned pc = idx - pbase;
    const unsigned s = sa.usym[pc];
    if (s != SEG_NONE) seg_pow_table<M, PER0>(lds, ct + ct_off[pc], s, S, fbuf + (size_t)(pbase + pc) * fstride);
    return;
  }
  const unsigned seg = idx;
  if (seg >= seg_base[B]) return;  // the grid is an upper bound
  const unsigned c = seg_ctx_of<M>(seg_base, seg), k = seg - seg_base[c];
  if (!seg_wants_setfunc(sa, seg, k, seg_base[c + 1] - seg_base[c])) return;
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
  uint16_t *f = fbuf + (size_t)seg * fstride;  // F[entry - size] = exit
#pragma unroll
  for (unsigned j = 0; j < PER0; j++) {
    const unsigned xi = lane + 64u * j;
    if (j < per && xi < size) f[xi] = level == 0 ? (uint16_t)x0[j] : L.list[L.m[x0[j]]];
  }
}

// Everything that needs the classes of k_seg_scan and nothing else, in ONE launch (a kernel boundary
// on a lane's stream is a wait for room on a busy chip, and these are latency chains that fill a
// fraction of it): workgroups [0, n_walk) are k_seg_walk<1>'s items, the next n_cand the candidate
// walks behind the anchors, the rest k_seg_setfunc's segments and the contexts' power tables.
template <class M, unsigned PER0>
__global__ void __launch_bounds__(64)
k_seg_stage1(const uint8_t *__restrict__ sorted_sym, uint16_t *__restrict__ out16, const uint32_t *__restrict__ arrays,
             const uint32_t *__restrict__ ct, const uint32_t *__restrict__ ct_off, uint16_t *__restrict__ final_state,
             unsigned S, unsigned n_walk, unsigned n_cand, unsigned pbase, unsigned fstride, SegArrays sa,
             uint16_t *__restrict__ fbuf, StreamResult *res) {
  extern __shared__ uint32_t lds[];
  __shared__ SetsWaveLds L;
  if (blockIdx.x < n_walk)
    seg_walk_role<M, 1>(lds, blockIdx.x, sorted_sym, out16, arrays, ct, ct_off, final_state, S, sa, res);
  else if (blockIdx.x < n_walk + n_cand)
    seg_cand_role<M, false>(lds, blockIdx.x - n_walk, n_cand, sorted_sym, arrays, ct, ct_off, S, fstride, sa, fbuf);
  else
    seg_setfunc_role<M, PER0>(lds, L, blockIdx.x - n_walk - n_cand, sorted_sym, arrays, ct, ct_off, S, pbase, fstride, sa, fbuf);
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
              const uint16_t *__restrict__ fbuf, unsigned pbase, unsigned fstride, SegArrays sa, ItemArrays ia) {
  constexpr unsigned B = M::B;
  const uint32_t *seg_base = arrays + B + (B + 1), *item_base = seg_base + B + 1;
  const unsigned item = blockIdx.x, lane = fq_lane();
  if (item >= item_base[B]) return;  // the grid is an upper bound
  const unsigned c = seg_ctx_of<M>(item_base, item);
  const unsigned nseg = seg_base[c + 1] - seg_base[c];
  const unsigned k0 = (item - item_base[c]) * 64, n_here = min(64u, nseg - k0);
  const unsigned seg0 = seg_base[c] + k0;
  // lane t looks at segment t of the item: function slot (SEG_NONE: transparent or last of the chain)
  const unsigned slot = lane < n_here ? seg_fslot(sa, seg0 + lane, k0 + lane, nseg, c, pbase) : SEG_NONE;
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
      // (a function of k_seg_cand is defined on the states that can occur only: whatever else
      // sits in its slot is brought into range and never followed by the resolve kernels)
      const uint16_t *f = fbuf + (size_t)sl * fstride;
#pragma unroll
      for (unsigned j = 0; j < PER0; j++)
        if (j < per) x[j] = size | ((unsigned)f[x[j] - size] & (size - 1));
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
               const uint16_t *__restrict__ fbuf, unsigned pbase, unsigned fstride, SegArrays sa, ItemArrays ia) {
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
    const unsigned sl = seg_fslot(sa, seg0 + t, k0 + t, nseg, c, pbase);
    x = sl == SEG_NONE ? (unsigned)sa.entry_state[seg0 + t + 1] : (unsigned)fbuf[(size_t)sl * fstride + (x - size)];
  }
  if (k0 + n_here < nseg) sa.entry_state[seg0 + n_here] = (uint16_t)x;  // first segment of the next item
}
