// Part of encode.hip (included there, inside its anonymous namespace): K4: CTable helpers and the sequence chain kernels (segment functions over state sets).

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
//  (A) k_seq_setfunc: a wave computes F: entry state -> exit state for EVERY possible entry
//      state.  It starts with all 2^log states spread over the lanes, and at a few points (after
//      4, 16, 48, 128, 512, 2048, ... symbols) replaces the states it carries by the distinct
//      ones ("classes"), remembering which class every entry state fell into.  After the first
//      hundred symbols a step costs 1-3 LDS gathers per wave for 64 lanes, from ~2048 symbols on
//      one -- so the wave walks on through a GROUP of up to 8 consecutive segments and writes
//      the function "entry state of the group -> state here" at every segment boundary.
//  (B) k_seq_resolve: entry state of every segment; a group's functions all start at the group's
//      entry state (one round of independent loads), x <- F_last[x] from group to group.
//  (C) k_seq_emit: every lane walks ONE segment from its now-known entry state and writes the
//      packed (nb, bits) of every symbol; 64 segments of a context per wave.
// All three read the context's one-symbol transition table next[s][x] from LDS (tables.hip
// builds it once per handle).  Exact by construction: no speculation, nothing to verify.
constexpr unsigned SETS_WAVES = 8;          // segments (waves) per workgroup in step A, one-symbol table
constexpr unsigned SETS_WAVES2 = 16;        // ... with the 64 KB two-symbol table (one workgroup per CU)
constexpr unsigned SETS_ROUNDS = 4;         // a workgroup owns WAVES * max(SETS_ROUNDS / group, 1) groups, handed out to its waves one by one
constexpr unsigned SETS_MAX_CLASSES = 512;  // above this a segment keeps carrying every state
constexpr unsigned SETS_MAX_GROUP = 16;     // segments a wave walks in one go, at most
constexpr unsigned SETS_BLOCK = 1024;       // symbols per 16-byte-per-lane load; S is a multiple

template <unsigned MAXC_, unsigned BMW_>
struct SetsWaveLdsT {
  static constexpr unsigned MAXC = MAXC_, BMW = BMW_;
  uint32_t bm[BMW];       // bitmap over the states (size <= 32 * BMW)
  uint16_t wpre[BMW];     // set bits before every bitmap word
  uint16_t list[MAXC];    // class -> state, as (state - size) * 2
  uint16_t tmp[MAXC];     // old class -> new class during a merge
  uint16_t m[MAXC];       // first-level class -> current class
};
using SetsWaveLds = SetsWaveLdsT<SETS_MAX_CLASSES, 128>;    // any log <= 12
// log <= 11 (the two-symbol kernel): 3456 B per wave, 16 waves + the 64 KB table = 118 KB, which
// leaves 42 KB of a CU's LDS to the kernels of the other lanes -- the quality K1 and K3 need 36 KB
// each and could not run beside a workgroup of this kernel while it took 125 KB
using SetsWaveLds11 = SetsWaveLdsT<SETS_MAX_CLASSES, 64>;

// Segments a k_seq_setfunc wave walks in one go ("group") for a chain with nf functions: as many
// as keep >= gmin groups in the chain (a workgroup's waves all busy), at most qmax.
__device__ __forceinline__ unsigned seq_group_of(unsigned nf, unsigned qmax, unsigned gmin) {
  return min(max((nf + gmin - 1) / gmin, 1u), qmax);
}

// plan[]: fitem_base[B+1] (step A workgroups before every context) | fseg_base[B+1] (functions
// before every context) | seg_base[B+1] (segments) | eitem_base[B+1] (step C waves) |
// citem_base[B+1] (step B items: 64 groups each, for chains of more than SEQ_ITEM_GROUPS groups)
constexpr unsigned SEGPLAN_WORDS = 5 * (SeqModel::B + 1) + 4;  // + the work counter of step A
constexpr unsigned SEQ_ITEM_GROUPS = 64;
// items of a chain with nl groups (0: the chain is short, k_seq_resolve walks its groups itself)
__device__ __forceinline__ unsigned seq_items_of(unsigned nl) { return nl > SEQ_ITEM_GROUPS ? (nl + SEQ_ITEM_GROUPS - 1) / SEQ_ITEM_GROUPS : 0u; }

__global__ void __launch_bounds__(256)
k_seq_segplan(const uint32_t *__restrict__ arrays, unsigned S, unsigned qmax, unsigned gmin, unsigned wpg,
              uint32_t *__restrict__ plan) {
  constexpr unsigned B = SeqModel::B;
  __shared__ unsigned s_nseg[B];
  const unsigned c = threadIdx.x;
  const unsigned n = arrays[c];
  s_nseg[c] = (n + S - 1) / S;
  __syncthreads();
  unsigned fi = 0, fs = 0, sg = 0, ei = 0, ci = 0;
  for (unsigned o = 0; o < c; o++) {
    const unsigned ns = s_nseg[o], nf = ns ? ns - 1 : 0, Q = seq_group_of(nf, qmax, gmin), nl = (nf + Q - 1) / Q;
    fi += (nl + wpg - 1) / wpg; fs += nf; sg += ns; ei += (ns + 63) / 64; ci += seq_items_of(nl);
  }
  uint32_t *fitem = plan, *fseg = plan + (B + 1), *seg = plan + 2 * (B + 1), *eitem = plan + 3 * (B + 1), *citem = plan + 4 * (B + 1);
  fitem[c] = fi; fseg[c] = fs; seg[c] = sg; eitem[c] = ei; citem[c] = ci;
  if (c == 0) plan[5 * (B + 1)] = 0;  // step A's work counter
  if (c == B - 1) {
    const unsigned ns = s_nseg[c], nf = ns ? ns - 1 : 0, Q = seq_group_of(nf, qmax, gmin), nl = (nf + Q - 1) / Q;
    fitem[B] = fi + (nl + wpg - 1) / wpg; fseg[B] = fs + nf; seg[B] = sg + ns; eitem[B] = ei + (ns + 63) / 64; citem[B] = ci + seq_items_of(nl);
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
template <class LT>
__device__ __forceinline__ unsigned sets_count(LT &L, unsigned nw) {
  const unsigned lane = fq_lane();
  const unsigned c0 = lane < nw ? __popc(L.bm[lane]) : 0u;
  const unsigned p0 = sets_incl_scan(c0), t0 = __builtin_amdgcn_readlane(p0, 63);
  L.wpre[lane] = (uint16_t)(p0 - c0);
  unsigned t1 = 0;
  if (LT::BMW > 64) {
    const unsigned c1 = lane + 64 < nw ? __popc(L.bm[lane + 64]) : 0u;
    const unsigned p1 = sets_incl_scan(c1);
    t1 = __builtin_amdgcn_readlane(p1, 63);
    L.wpre[lane + 64] = (uint16_t)(t0 + p1 - c1);
  }
  fq_lds_wave_sync();
  return t0 + t1;
}
template <class LT>
__device__ __forceinline__ unsigned sets_rank(const LT &L, unsigned xi) {
  return (unsigned)L.wpre[xi >> 5] + __popc(L.bm[xi >> 5] & ((1u << (xi & 31u)) - 1u));
}
template <class LT>
__device__ __forceinline__ void sets_clear(LT &L) {
  L.bm[fq_lane()] = 0;
  if (LT::BMW > 64) L.bm[fq_lane() + 64] = 0;
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

// One gather of the two-symbol walk, table at LDS address 0 (k_seq_setfunc has no static LDS: tests/test_build_invariants.py):
// the row offset is one 16-bit half of a scalar register (two rows per register, sets_pack_rows) and is picked by the ADD
// itself (SDWA word select on the scalar operand), the LDS address is the sum -- add, read, wait: three instructions per
// gather where the compiler's version had five (s_and / s_lshr for the half, s_add for the table's base).
typedef __attribute__((address_space(3))) const uint16_t fq_lds_u16;
template <int HALF>
__device__ __forceinline__ unsigned sets_gather2(unsigned rowpair, unsigned y) {
  unsigned addr;
  if (HALF == 0)
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(addr) : "s"(rowpair), "v"(y));
  else
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(addr) : "s"(rowpair), "v"(y));
  return *reinterpret_cast<fq_lds_u16 *>((uintptr_t)addr);
}

// n classes (states in L.list) walked through words [w0, w1) of the segment, M per lane
template <int M, bool TWO, class LT>
__device__ __forceinline__ void sets_walk(LT &L, unsigned n, const char *tbase, unsigned log,
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
  w0 = fq_uniform(w0); w1 = fq_uniform(w1);  // (wave-uniform by construction; said so, the loops below run on scalar counters)
  if ((w0 | w1) & 3u) {  // only the first two ranges of a segment: [0, 1) and [1, 4)
    for (unsigned w = w0; w < w1; w++) step_word(sets_word(cur, w));
  } else if (TWO) {  // whole groups of 16 symbols = eight prepared row offsets of lane g
    for (unsigned g = w0 >> 2; g < (w1 >> 2); g++) {
      const unsigned gi = g & 63u;
      const unsigned r[4] = {(unsigned)__builtin_amdgcn_readlane(rows.x, gi), (unsigned)__builtin_amdgcn_readlane(rows.y, gi),
                             (unsigned)__builtin_amdgcn_readlane(rows.z, gi), (unsigned)__builtin_amdgcn_readlane(rows.w, gi)};
#pragma unroll
      for (int i = 0; i < 4; i++) {
#pragma unroll
        for (int j = 0; j < M; j++) y[j] = sets_gather2<0>(r[i], y[j]);
#pragma unroll
        for (int j = 0; j < M; j++) y[j] = sets_gather2<1>(r[i], y[j]);
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
template <bool XO, class LT>
__device__ __forceinline__ unsigned sets_merge(LT &L, unsigned n, unsigned n1, unsigned nw, unsigned size) {
  const unsigned lane = fq_lane();
  sets_clear(L);
  for (unsigned i = lane; i < n; i += 64) {
    const unsigned xi = sets_idx<XO>(L.list[i], size);
    atomicOr(&L.bm[xi >> 5], 1u << (xi & 31u));
  }
  fq_lds_wave_sync();
  const unsigned nn = sets_count(L, nw);
  if ((nn + 63) / 64 >= (n + 63) / 64) return n;
  unsigned st[LT::MAXC / 64];
#pragma unroll
  for (unsigned j = 0; j < LT::MAXC / 64; j++) {
    const unsigned i = lane + 64u * j;
    st[j] = i < n ? (unsigned)L.list[i] : 0u;
    if (i < n) L.tmp[i] = (uint16_t)sets_rank(L, sets_idx<XO>(st[j], size));
  }
  fq_lds_wave_sync();
#pragma unroll
  for (unsigned j = 0; j < LT::MAXC / 64; j++) {
    const unsigned i = lane + 64u * j;
    if (i < n) L.list[L.tmp[i]] = (uint16_t)st[j];  // equal states write the same value
  }
  for (unsigned i = lane; i < n1; i += 64) L.m[i] = L.tmp[L.m[i]];
  fq_lds_wave_sync();
  return nn;
}

// Step A.  PER0 = states per lane at the start: 32 covers log <= 11, 64 covers log 12.
// TWO: two symbols per gather through the context's 64 KB two-symbol table (log <= 11).
// Registers: the two-symbol kernel is held to 96 VGPRs (five waves per SIMD; it spills 8 more
// registers of its rarely run merge code than at 128).  A workgroup is 4 waves per SIMD: at 128
// VGPRs they filled the SIMD's register file and NOTHING else could run on a CU while this
// kernel held it -- its time simply added to the step.  With 96, waves of the other lanes'
// kernels fit beside it: 66.9 -> 70.8 GB/s (80 VGPRs: 68.8; 64: 65.7).
template <unsigned PER0, bool TWO>
__global__ void __launch_bounds__((TWO ? SETS_WAVES2 : SETS_WAVES) * 64, TWO ? 5 : 1)
k_seq_setfunc(const uint8_t *__restrict__ sorted_sym, const uint32_t *__restrict__ arrays,
              const uint32_t *__restrict__ plan, const uint32_t *__restrict__ logs,
              const uint16_t *__restrict__ next, unsigned next_stride, const uint16_t *__restrict__ pow, unsigned pow_stride,
              unsigned S, unsigned qmax, unsigned gmin,
              unsigned rounds, unsigned fstride, uint16_t *__restrict__ fbuf, unsigned *__restrict__ work_counter, unsigned table_bytes) {
  constexpr unsigned WAVES = TWO ? SETS_WAVES2 : SETS_WAVES;
  // ALL of the kernel's LDS is the dynamic block, the context's table first: next[4][size] (TWO: next2[16][size]) sits at LDS
  // address 0, so a gather's address is row + state with nothing added (with the per-wave buffers as static LDS in front of
  // it the compiler spent one s_add per gather on the table's base); behind it the waves' set buffers and two words
  using LT = typename std::conditional<TWO, SetsWaveLds11, SetsWaveLds>::type;
  extern __shared__ uint32_t lds[];
  LT *wl = reinterpret_cast<LT *>(reinterpret_cast<char *>(lds) + table_bytes);
  unsigned &s_next = *reinterpret_cast<unsigned *>(wl + WAVES), &s_item = *(reinterpret_cast<unsigned *>(wl + WAVES) + 1);
  constexpr unsigned B = SeqModel::B;
  const uint32_t *fitem = plan, *fseg = plan + (B + 1);
  const unsigned wave = threadIdx.x >> 6, lane = fq_lane();
  LT &L = wl[wave];
  const char *tbase = reinterpret_cast<const char *>(lds);
  const unsigned sub_blocks = S / SETS_BLOCK;  // 1024-symbol loads per segment
  const unsigned n_items = fitem[B];
  unsigned loaded = 0xFFFFFFFFu;  // context whose table is in LDS
  // Persistent workgroups (one per CU, 118 KB of LDS with the two-symbol table): items are
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
    const unsigned nf = fseg[c + 1] - fseg[c], Q = seq_group_of(nf, qmax, gmin), nl = (nf + Q - 1) / Q;
    // A wave walks a GROUP of up to Q consecutive segments in one go and writes the function
    // "entry state of the group -> state behind segment j" at every segment boundary: the state
    // sets keep shrinking along the group (one gather per step from ~2048 symbols on), while a
    // fresh start pays ~1500 gathers for its first 2048 symbols.  The item's groups [k0, k_end)
    // of the chain go to whichever wave is free.
    const unsigned k0 = (item - fitem[c]) * (WAVES * rounds), k_end = min(k0 + WAVES * rounds, nl);
    const unsigned per = max(size >> 6, 1u), nw = max(size >> 5, 1u);
    for (unsigned k = k0 + wave; k < k_end;) {
      const unsigned s0 = k * Q, nblk = min(Q, nf - s0) * sub_blocks, w_end = nblk * (SETS_BLOCK / 4);
      const uint4 *gseg = reinterpret_cast<const uint4 *>(sorted_sym + arrays[B + c] + (size_t)s0 * S);

      // level 0: every state; lane l carries states l, l + 64, ...
      unsigned x0[PER0];
#pragma unroll
      for (unsigned j = 0; j < PER0; j++) x0[j] = ((lane + 64u * j) & (size - 1)) * 2u;
      unsigned level = 0, n = size, n1 = 0;
      unsigned w = 0, stop = 1;  // merge points after 4, 16, 48, 128, 512, 2048, 8192, ... symbols
      uint4 cur = gseg[lane];
      auto write_function = [&](unsigned j) {  // F_j[entry of the group] = state here, both as (state - size) * 2
        uint16_t *f = fbuf + (size_t)(fseg[c] + s0 + j) * fstride;
#pragma unroll
        for (unsigned jj = 0; jj < PER0; jj++) {
          const unsigned xi = lane + 64u * jj;
          if (jj < per && xi < size) f[xi] = level == 0 ? (uint16_t)x0[jj] : L.list[L.m[x0[jj]]];
        }
      };
      for (unsigned blk = 0; blk < nblk; blk++) {
        // A segment that is S times ONE symbol (a homopolymer context fed its own base) is a power of
        // that symbol's transition: T_s^S from the handle's table, one lookup per carried state instead
        // of S steps -- such a context is (nearly) a permutation of the states, nothing ever merges.
        if (pow != nullptr && blk % sub_blocks == 0) {
          const unsigned s_first = (unsigned)__builtin_amdgcn_readfirstlane(cur.x) & 3u, pat = s_first * 0x01010101u;
          auto differs = [&](const uint4 v) { return (((v.x ^ pat) | (v.y ^ pat) | (v.z ^ pat) | (v.w ^ pat)) & 0x03030303u) != 0u; };
          bool uni = __ballot(differs(cur)) == 0ull;
          for (unsigned b = 1; uni && b < sub_blocks; b++) uni = __ballot(differs(gseg[(size_t)(blk + b) * 64 + lane])) == 0ull;
          if (uni) {
            const char *P = reinterpret_cast<const char *>(pow + (size_t)c * pow_stride + ((size_t)s_first << log));
            if (level == 0) {
#pragma unroll
              for (unsigned j = 0; j < PER0; j++)
                if (j < per) x0[j] = *reinterpret_cast<const uint16_t *>(P + x0[j]);
            } else {
              for (unsigned i = lane; i < n; i += 64) L.list[i] = *reinterpret_cast<const uint16_t *>(P + L.list[i]);
              fq_lds_wave_sync();
            }
            blk += sub_blocks - 1;
            w = (blk + 1) * (SETS_BLOCK / 4);
            while (stop <= w) stop = stop == 1 ? 4 : stop == 4 ? 12 : stop == 12 ? 32 : stop * 4;  // merge points inside the segment are dropped
            if (blk + 1 < nblk) cur = gseg[(size_t)(blk + 1) * 64 + lane];
            write_function(blk / sub_blocks);
            continue;
          }
        }
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
        if ((blk + 1) % sub_blocks == 0) write_function(blk / sub_blocks);
      }
      unsigned nk = 0;
      if (lane == 0) nk = atomicAdd(&s_next, 1u);
      k = k0 + (unsigned)__builtin_amdgcn_readfirstlane(nk);
    }
  }
}

// Step B: entry state of every segment of every chain.  The functions of a group all start at
// the group's entry state, so a group costs one round of independent 2-byte loads, and the chain
// x <- F_last[x] runs from group to group.  A chain of more than SEQ_ITEM_GROUPS groups (one context
// holding most of a block: 3 600 groups for 256 MiB of poly-A) is resolved in three levels over
// ITEMS of 64 groups, like the quality stream's runs of opaque segments:
//  compose  one wave per item: the item's composed function for every possible entry state
//  resolve  one thread per context: item by item
//  expand   one thread per item: group by group inside the item
__device__ __forceinline__ unsigned seq_resolve_groups(const uint32_t *__restrict__ fseg, const uint32_t *__restrict__ seg,
                                                       const uint16_t *__restrict__ fbuf, unsigned fstride, unsigned c, unsigned Q,
                                                       unsigned nf, unsigned ns, unsigned s_begin, unsigned s_end, unsigned xo,
                                                       uint16_t *__restrict__ entry) {
  for (unsigned s0 = s_begin; s0 < s_end; s0 += Q) {
    entry[seg[c] + s0] = (uint16_t)xo;
    unsigned v[SETS_MAX_GROUP];
#pragma unroll
    for (unsigned j = 0; j < SETS_MAX_GROUP; j++)
      v[j] = j < Q && s0 + j < nf ? (unsigned)fbuf[(size_t)(fseg[c] + s0 + j) * fstride + (xo >> 1)] : 0u;
    unsigned nx = xo;
#pragma unroll
    for (unsigned j = 0; j < SETS_MAX_GROUP; j++)
      if (j < Q && s0 + j < nf) {
        if (j + 1 < Q) entry[seg[c] + s0 + j + 1] = (uint16_t)v[j]; else nx = v[j];
      }
    xo = nx;
  }
  (void)ns;
  return xo;
}

constexpr unsigned SEQ_RESOLVE_THREADS = 1024;

// the composed function of one item (64 groups) for every possible entry state; one wave
template <unsigned PER0>
__device__ __forceinline__ void seq_compose_item(const uint32_t *__restrict__ plan, const uint32_t *__restrict__ logs,
                                                 const uint16_t *__restrict__ fbuf, unsigned fstride, unsigned qmax, unsigned gmin,
                                                 unsigned item, uint16_t *__restrict__ cbuf) {
  constexpr unsigned B = SeqModel::B;
  const uint32_t *fseg = plan + (B + 1), *citem = plan + 4 * (B + 1);
  const unsigned lane = fq_lane();
  const unsigned c = seq_item_ctx(citem, item);
  const unsigned nf = fseg[c + 1] - fseg[c], Q = seq_group_of(nf, qmax, gmin), nl = (nf + Q - 1) / Q;
  const unsigned g0 = (item - citem[c]) * SEQ_ITEM_GROUPS, g1 = min(g0 + SEQ_ITEM_GROUPS, nl);
  const unsigned size = 1u << logs[c], per = max(size >> 6, 1u);
  unsigned x[PER0];  // (state - size) * 2 behind the groups walked so far, for every entry state of the item
#pragma unroll
  for (unsigned j = 0; j < PER0; j++) x[j] = ((lane + 64u * j) & (size - 1)) * 2u;
  for (unsigned g = g0; g < g1; g++) {
    const unsigned last = min(g * Q + Q, nf) - 1;  // the group's last function: entry of the group -> entry of the next
    const uint16_t *f = fbuf + (size_t)(fseg[c] + last) * fstride;
#pragma unroll
    for (unsigned j = 0; j < PER0; j++)
      if (j < per) x[j] = f[x[j] >> 1];
  }
  uint16_t *o = cbuf + (size_t)item * fstride;
#pragma unroll
  for (unsigned j = 0; j < PER0; j++) {
    const unsigned xi = lane + 64u * j;
    if (j < per && xi < size) o[xi] = (uint16_t)x[j];
  }
}

// One workgroup, one launch (every kernel boundary on a lane's stream is a wait for the chip to have
// room again): thread c resolves a short chain group by group; long chains go through the three
// levels, the levels separated by workgroup barriers.
template <unsigned PER0>
__global__ void __launch_bounds__(SEQ_RESOLVE_THREADS)
k_seq_resolve(const uint32_t *__restrict__ plan, const uint32_t *__restrict__ logs, const uint16_t *__restrict__ fbuf,
              unsigned fstride, unsigned qmax, unsigned gmin, uint16_t *__restrict__ cbuf, uint16_t *__restrict__ item_entry,
              uint16_t *__restrict__ entry) {
  constexpr unsigned B = SeqModel::B;
  const uint32_t *fseg = plan + (B + 1), *seg = plan + 2 * (B + 1), *citem = plan + 4 * (B + 1);
  const unsigned c = threadIdx.x;
  unsigned ns = 0, nf = 0, Q = 1, ni = 0;
  if (c < B) {
    ns = seg[c + 1] - seg[c]; nf = ns ? ns - 1 : 0; Q = seq_group_of(nf, qmax, gmin);
    ni = citem[c + 1] - citem[c];
    if (ni == 0) (void)seq_resolve_groups(fseg, seg, fbuf, fstride, c, Q, nf, ns, 0, ns, 0u, entry);  // FSE_initCState: state = size
  }
  const unsigned n_items = citem[B];
  if (n_items == 0) return;  // (uniform) no long chain in this block
  for (unsigned item = threadIdx.x >> 6; item < n_items; item += SEQ_RESOLVE_THREADS / 64)
    seq_compose_item<PER0>(plan, logs, fbuf, fstride, qmax, gmin, item, cbuf);
  __threadfence();
  __syncthreads();
  if (c < B && ni) {
    unsigned xo = 0;
    for (unsigned i = 0; i < ni; i++) {
      item_entry[citem[c] + i] = (uint16_t)xo;
      if (i + 1 < ni) xo = __hip_atomic_load(&cbuf[(size_t)(citem[c] + i) * fstride + (xo >> 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __threadfence();
  __syncthreads();
  for (unsigned item = threadIdx.x; item < n_items; item += SEQ_RESOLVE_THREADS) {
    const unsigned ic = seq_item_ctx(citem, item);
    const unsigned ins = seg[ic + 1] - seg[ic], inf = ins ? ins - 1 : 0, iQ = seq_group_of(inf, qmax, gmin);
    // (the chain's LAST segment has no function of its own and belongs to no group: when the functions fill the
    // last item's 64 groups exactly, s_begin + 64 * iQ stops one segment short of it -- the last item runs to the end)
    const unsigned s_begin = (item - citem[ic]) * SEQ_ITEM_GROUPS * iQ;
    const unsigned s_end = item + 1 == citem[ic + 1] ? ins : min(s_begin + SEQ_ITEM_GROUPS * iQ, ins);
    const unsigned xo = __hip_atomic_load(&item_entry[item], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    (void)seq_resolve_groups(fseg, seg, fbuf, fstride, ic, iQ, inf, ins, s_begin, s_end, xo, entry);
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
