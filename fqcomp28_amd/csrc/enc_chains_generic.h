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
