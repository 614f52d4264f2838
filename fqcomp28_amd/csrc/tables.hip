// Dataset analysis and table construction on the device.
//   k_hist_seq / k_hist_qual : FSE_Sequence::calculateFreqTable (reference
//       src/fse_sequence.cpp:145-169) / FSE_Quality::calculateFreqTable
//       (src/fse_quality.cpp:69-97): per-context symbol counts, initialised to 1
//   k_normalize              : makeNormalizedFreqTable (src/fse_common.hpp:179-200) =
//       FSE_optimalTableLog + FSE_normalizeCount(useLowProbCount=1) per context,
//       one wave per context, one lane per symbol
//   k_build_tables           : FSE_buildCTable_wksp + FSE_buildDTable_wksp for every context
//       (FSE_Encoder/FSE_Decoder ctors, src/fse_common.hpp:46-71,107-127), one wave per
//       context: wave prefix sums over the normalised counts give the cumulative table,
//       ballot ranks give the symbol spread and the per-symbol state numbering
// zstd's algorithms are restated from their published behaviour.
#include "fqgpu_internal.h"

namespace {

// ------------------------------------------------------------------ histograms
// One wave per record.  N is skipped WITHOUT advancing the context
// (src/fse_sequence.cpp:156-158): the context of a base is the last four non-N bases.
__global__ void __launch_bounds__(256)
k_hist_seq(const uint8_t *__restrict__ raw, const fqgpu_rec *__restrict__ recs, unsigned R,
           uint32_t *__restrict__ counts, uint32_t *__restrict__ err) {
  __shared__ uint32_t hist[FQGPU_SEQ_MODELS * FQGPU_SEQ_ALPHA];
  for (unsigned i = threadIdx.x; i < FQGPU_SEQ_MODELS * FQGPU_SEQ_ALPHA; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  const unsigned waves = (gridDim.x * blockDim.x) >> 6;
  const unsigned lane = fq_lane();
  bool bad = false;  // a byte that is neither a base nor N (base2bits_arr: UINT_MAX, src/fse_sequence.cpp:6-14)
  for (unsigned r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < R; r += waves) {
    const fqgpu_rec rec = recs[r];
    const uint8_t *s = raw + rec.seq_off;
    unsigned carry = 0xD7u;  // INITIAL_CONTEXT: codes of the 4 bases before, nearest in bits 7:6
    for (unsigned base = 0; base < rec.len; base += 64) {
      const unsigned i = base + lane;
      const unsigned ch = i < rec.len ? s[i] : 'N';
      const bool keep = i < rec.len && ch != 'N';
      const unsigned code = fq_base_code(ch);
      bad |= fq_base_sym(ch) > 3u;
      const unsigned long long km = __ballot(keep);
      unsigned long long m = km & ((1ull << lane) - 1ull);  // kept lanes before me
      const unsigned avail = (unsigned)__popcll(m);
      unsigned ctx = 0;
#pragma unroll
      for (int k = 1; k <= 4; k++) {  // k-th nearest non-N predecessor
        const int src = m ? 63 - __clzll(m) : 0;
        const unsigned from_wave = (unsigned)__shfl((int)code, src);
        const unsigned back = (unsigned)k > avail ? (unsigned)k - avail : 1u;  // 1..4 into the carry
        const unsigned from_carry = (carry >> (2 * (4 - back))) & 3u;
        const unsigned c = (unsigned)k <= avail ? from_wave : from_carry;
        ctx |= c << (2 * (4 - k));
        if (m) m &= ~(1ull << src);
      }
      if (keep) atomicAdd(&hist[ctx * 4 + code], 1u);
      // new carry = context after the last kept base of this chunk
      const unsigned kept = (unsigned)__popcll(km);
      if (kept) {
        const int lastl = 63 - __clzll(km);
        const unsigned ctx_last = (unsigned)__shfl((int)ctx, lastl);
        const unsigned code_last = (unsigned)__shfl((int)code, lastl);
        carry = (ctx_last >> 2) + (code_last << 6);  // addSymUpper
      }
    }
  }
  if (bad) atomicOr(err, 1u);
  __syncthreads();
  for (unsigned i = threadIdx.x; i < FQGPU_SEQ_MODELS * FQGPU_SEQ_ALPHA; i += blockDim.x)
    if (hist[i]) atomicAdd(&counts[i], hist[i]);
}

// One wave per record; lanes that hit the same (context, symbol) cell as the first
// active lane are folded into one atomic (real data has a few very hot cells).
__global__ void __launch_bounds__(256)
k_hist_qual(const uint8_t *__restrict__ raw, const fqgpu_rec *__restrict__ recs, unsigned R,
            uint32_t *__restrict__ counts, uint32_t *__restrict__ err) {
  const unsigned waves = (gridDim.x * blockDim.x) >> 6;
  const unsigned lane = fq_lane();
  bool bad = false;
  for (unsigned r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < R; r += waves) {
    const fqgpu_rec rec = recs[r];
    const uint8_t *qs = raw + rec.qual_off;
    for (unsigned base = 0; base < rec.len; base += 64) {
      const unsigned i = base + lane;
      bool active = i < rec.len;
      unsigned cell = 0;
      if (active) {
        // decoder-side definition (src/fse_quality.cpp:79-93): ctx from the three
        // previous symbols, zeros before the read
        const unsigned q = (unsigned)qs[i] - 33u;
        const unsigned a = i >= 1 ? (unsigned)qs[i - 1] - 33u : 0u;
        const unsigned b = i >= 2 ? (unsigned)qs[i - 2] - 33u : 0u;
        const unsigned c = i >= 3 ? (unsigned)qs[i - 3] - 33u : 0u;
        if (q >= FQGPU_QUAL_ALPHA) { bad = true; active = false; }
        cell = fq_qual_ctx(a & 63u, b & 63u, c & 63u) * FQGPU_QUAL_ALPHA + (q & 63u);
      }
      for (int round = 0; round < 2; round++) {
        const unsigned long long am = __ballot(active);
        if (!am) break;
        const int lead = __ffsll((long long)am) - 1;
        const unsigned lead_cell = (unsigned)__shfl((int)cell, lead);
        const bool same = active && cell == lead_cell;
        const unsigned long long sm = __ballot(same);
        if ((int)lane == lead) atomicAdd(&counts[lead_cell], (unsigned)__popcll(sm));
        if (same) active = false;
      }
      if (active) atomicAdd(&counts[cell], 1u);
    }
  }
  if (bad) atomicOr(err, 1u);
}

__global__ void k_fill_u32(uint32_t *p, size_t n, uint32_t v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = v;
}

// ------------------------------------------------------------------ normalisation
__device__ __forceinline__ unsigned hb32(unsigned v) { return 31u - (unsigned)__clz((int)v); }

// FSE_optimalTableLog(0, total, maxSV): note the u32 wrap when total-1 < 4
__device__ unsigned optimal_table_log(unsigned long long total, unsigned max_sv) {
  const unsigned max_bits_src = hb32((unsigned)(total - 1)) - 2u;
  unsigned t = 11;
  const unsigned by_src = hb32((unsigned)total) + 1, by_sym = hb32(max_sv) + 2;
  const unsigned min_bits = by_src < by_sym ? by_src : by_sym;
  if (max_bits_src < t) t = max_bits_src;
  if (min_bits > t) t = min_bits;
  if (t < 5) t = 5;
  if (t > 12) t = 12;
  return t;
}

// FSE_normalizeM2: rare second-chance path, run by one lane on the LDS copies
__device__ int normalize_m2(short *norm, unsigned t, const uint32_t *count, unsigned long long total,
                            unsigned max_sv) {
  const short UNSET = -2;
  unsigned distributed = 0, to_dist;
  const unsigned low_thr = (unsigned)(total >> t);
  unsigned low_one = (unsigned)((total * 3) >> (t + 1));
  for (unsigned s = 0; s <= max_sv; s++) {
    if (count[s] == 0) { norm[s] = 0; continue; }
    if (count[s] <= low_thr) { norm[s] = -1; distributed++; total -= count[s]; continue; }
    if (count[s] <= low_one) { norm[s] = 1; distributed++; total -= count[s]; continue; }
    norm[s] = UNSET;
  }
  to_dist = (1u << t) - distributed;
  if (to_dist == 0) return 0;
  if ((total / to_dist) > low_one) {
    low_one = (unsigned)((total * 3) / (to_dist * 2));
    for (unsigned s = 0; s <= max_sv; s++)
      if (norm[s] == UNSET && count[s] <= low_one) { norm[s] = 1; distributed++; total -= count[s]; }
    to_dist = (1u << t) - distributed;
  }
  if (distributed == max_sv + 1) {
    unsigned best = 0, best_c = 0;
    for (unsigned s = 0; s <= max_sv; s++)
      if (count[s] > best_c) { best = s; best_c = count[s]; }
    norm[best] += (short)to_dist;
    return 0;
  }
  if (total == 0) {
    for (unsigned s = 0; to_dist > 0; s = (s + 1) % (max_sv + 1))
      if (norm[s] > 0) { to_dist--; norm[s]++; }
    return 0;
  }
  const unsigned long long vsl = 62 - t;
  const unsigned long long mid = (1ull << (vsl - 1)) - 1;
  const unsigned long long rstep = (((1ull << vsl) * to_dist) + mid) / (unsigned)total;
  unsigned long long run = mid;
  for (unsigned s = 0; s <= max_sv; s++) {
    if (norm[s] == UNSET) {
      const unsigned long long end = run + (unsigned long long)count[s] * rstep;
      const unsigned w = (unsigned)(end >> vsl) - (unsigned)(run >> vsl);
      if (w < 1) return -1;
      norm[s] = (short)w;
      run = end;
    }
  }
  return 0;
}

// rest-to-beat thresholds of FSE_normalizeCount (zstd fse_compress.c)
__constant__ unsigned rtb[8] = {0, 473195, 504333, 520860, 550000, 700000, 750000, 830000};

// One wave per context, lane s = symbol s (A <= 64).
template <int A>
__global__ void __launch_bounds__(64)
k_normalize(const uint32_t *__restrict__ counts, int16_t *__restrict__ norm_out,
            uint32_t *__restrict__ logs, uint32_t *__restrict__ max_log, uint32_t *__restrict__ err) {
  __shared__ short s_norm[64];
  __shared__ uint32_t s_cnt[64];
  const unsigned ctx = blockIdx.x, lane = threadIdx.x;
  const bool on = lane < (unsigned)A;
  const uint32_t cnt = on ? counts[(size_t)ctx * A + lane] : 0u;
  unsigned long long total = cnt;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) total += __shfl_xor(total, d);
  if (total == 0 || total > 0xFFFFFFFFull) { if (lane == 0) atomicOr(err, 2u); return; }
  const unsigned t = optimal_table_log(total, A - 1);
  const unsigned long long scale = 62 - t;
  const unsigned long long step = (1ull << 62) / (unsigned)total;
  const unsigned long long vstep = 1ull << (scale - 20);
  const unsigned low_thr = (unsigned)(total >> t);
  int p = 0;
  if (__ballot(on && (unsigned long long)cnt == total)) {  // RLE case: the reference cannot reach it
    if (lane == 0) atomicOr(err, 4u);
    return;
  }
  if (on && cnt != 0) {
    if (cnt <= low_thr) {
      p = -1;
    } else {
      p = (int)(short)(((unsigned long long)cnt * step) >> scale);
      if (p < 8) {
        const unsigned long long to_beat = vstep * rtb[p];
        p += (((unsigned long long)cnt * step) - ((unsigned long long)p << scale)) > to_beat;
      }
    }
  }
  int used = p == -1 ? 1 : p;  // what the symbol takes out of 2^t
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) used += __shfl_xor(used, d);
  const int still = (1 << (int)t) - used;
  // first strictly largest probability (low-prob symbols never qualify: largestP starts at 0)
  int best = p > 0 ? p : 0;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { const int o = __shfl_xor(best, d); best = o > best ? o : best; }
  const unsigned long long at_best = __ballot(on && p > 0 && p == best);
  const unsigned largest = at_best ? (unsigned)__ffsll((long long)at_best) - 1u : 0u;
  const int norm_largest = __shfl(p, (int)largest);
  if (-still >= (norm_largest >> 1)) {
    s_cnt[lane] = cnt;
    __syncthreads();
    if (lane == 0 && normalize_m2(s_norm, t, s_cnt, total, A - 1) != 0) atomicOr(err, 8u);
    __syncthreads();
    p = s_norm[lane];
  } else if (lane == largest) {
    p += still;
  }
  if (on) norm_out[(size_t)ctx * A + lane] = (int16_t)p;
  if (lane == 0) { logs[ctx] = t; atomicMax(max_log, t); }
}

// exclusive prefix of the logs = bit offset of every context in the state flush
__global__ void __launch_bounds__(1024)
k_log_prefix(const uint32_t *__restrict__ logs, unsigned B, uint32_t *__restrict__ prefix,
             uint32_t *__restrict__ ct_off, uint32_t *__restrict__ dt_off, unsigned A,
             uint32_t *__restrict__ totals) {
  // single thread-block, B <= 8192: serial per thread over a blocked range
  __shared__ unsigned part[3][1024];
  const unsigned per = (B + blockDim.x - 1) / blockDim.x;
  const unsigned c0 = threadIdx.x * per, c1 = min(c0 + per, B);
  unsigned a = 0, b = 0, d = 0;
  for (unsigned c = c0; c < c1; c++) {
    const unsigned t = logs[c];
    a += t;
    b += 1u + (1u << (t - 1)) + 2u * A;  // FSE_CTABLE_SIZE_U32
    d += 1u + (1u << t);                 // FSE_DTABLE_SIZE_U32
  }
  part[0][threadIdx.x] = a; part[1][threadIdx.x] = b; part[2][threadIdx.x] = d;
  __syncthreads();
  if (threadIdx.x < 3) {
    unsigned run = 0;
    for (unsigned i = 0; i < blockDim.x; i++) {
      const unsigned v = part[threadIdx.x][i];
      part[threadIdx.x][i] = run;
      run += v;
    }
    totals[threadIdx.x] = run;
  }
  __syncthreads();
  a = part[0][threadIdx.x]; b = part[1][threadIdx.x]; d = part[2][threadIdx.x];
  for (unsigned c = c0; c < c1; c++) {
    const unsigned t = logs[c];
    prefix[c] = a; ct_off[c] = b; dt_off[c] = d;
    a += t; b += 1u + (1u << (t - 1)) + 2u * A; d += 1u + (1u << t);
  }
  if (c1 == B && c0 < B) prefix[B] = a;
}

// ------------------------------------------------------------------ CTable + DTable of one context
// One wave per context.  zstd memory layouts are kept so the tables can be dumped and
// compared word for word with FSE_buildCTable_wksp / FSE_buildDTable_wksp.
template <int A, int ABITS>
__global__ void __launch_bounds__(64)
k_build_tables(const int16_t *__restrict__ norm, const uint32_t *__restrict__ logs,
               const uint32_t *__restrict__ ct_off, const uint32_t *__restrict__ dt_off,
               uint32_t *__restrict__ ct_pool, uint32_t *__restrict__ dt_pool,
               uint32_t *__restrict__ err) {
  __shared__ uint8_t cell[1 << 12];     // symbol of every table position
  __shared__ uint32_t cum_pos[65];      // prefix of the positive counts (spread order)
  __shared__ uint32_t cumul[65];        // prefix with -1 counted as 1 (state numbering)
  __shared__ uint32_t next_rank[64];    // running occurrence count per symbol
  const unsigned ctx = blockIdx.x, lane = threadIdx.x;
  const unsigned t = logs[ctx];
  if (t < 5 || t > 12) { if (lane == 0) atomicOr(err, 16u); return; }
  const unsigned size = 1u << t, mask = size - 1;
  const unsigned step = (size >> 1) + (size >> 3) + 3;
  const bool on = lane < (unsigned)A;
  const int n = on ? (int)norm[(size_t)ctx * A + lane] : 0;
  uint32_t *ct = ct_pool + ct_off[ctx];
  uint32_t *dt = dt_pool + dt_off[ctx];

  // wave prefix sums over the normalised counts
  const unsigned pos_n = n > 0 ? (unsigned)n : 0u;
  const unsigned any_n = n == -1 ? 1u : pos_n;
  unsigned inc_pos = pos_n, inc_any = any_n;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned o1 = __shfl_up(inc_pos, d), o2 = __shfl_up(inc_any, d);
    if (lane >= (unsigned)d) { inc_pos += o1; inc_any += o2; }
  }
  const unsigned sum_any = __shfl(inc_any, 63);
  if (sum_any != size) { if (lane == 0) atomicOr(err, 32u); return; }
  cum_pos[lane + 1] = inc_pos;
  cumul[lane + 1] = inc_any;
  if (lane == 0) { cum_pos[0] = 0; cumul[0] = 0; }
  next_rank[lane] = 0;
  // low-probability symbols take the top cells, in ascending symbol order
  const unsigned long long low_mask = __ballot(n == -1);
  const unsigned n_low = (unsigned)__popcll(low_mask);
  const unsigned high = size - 1 - n_low;  // last cell of the regular spread
  if (n == -1) cell[size - 1 - fq_mbcnt(low_mask)] = (uint8_t)lane;
  __syncthreads();

  // regular spread: the walk j -> (j*step)&mask visits every cell once; the valid cells
  // (<= high) in visiting order receive the symbols in ascending order, norm[s] times each
  unsigned seen = 0;  // valid cells visited so far
  for (unsigned jb = 0; jb < size; jb += 64) {
    const unsigned j = jb + lane;
    const unsigned pos = (j * step) & mask;
    const bool ok = j < size && pos <= high;
    const unsigned long long okm = __ballot(ok);
    if (ok) {
      const unsigned k = seen + fq_mbcnt(okm);
      unsigned lo = 0, hi = A - 1;  // last symbol with cum_pos[s] <= k
      while (lo < hi) {
        const unsigned mid = lo + ((hi - lo + 1) >> 1);
        if (cum_pos[mid] <= k) lo = mid; else hi = mid - 1;
      }
      cell[pos] = (uint8_t)lo;
    }
    seen += (unsigned)__popcll(okm);
  }
  __syncthreads();

  // header words
  if (lane == 0) {
    ct[0] = t | ((unsigned)(A - 1) << 16);
    unsigned fast = 1;  // FSE_DTableHeader.fastMode: cleared by any norm >= size/2
    for (int s = 0; s < A; s++) if ((int)norm[(size_t)ctx * A + s] >= (int)(size >> 1)) fast = 0;
    dt[0] = t | (fast << 16);
  }
  // state numbering: cell u, symbol s, r-th occurrence of s in ascending u
  uint16_t *state_table = reinterpret_cast<uint16_t *>(ct) + 2;
  for (unsigned ub = 0; ub < size; ub += 64) {
    const unsigned u = ub + lane;
    const bool valid = u < size;  // size can be 32
    const unsigned s = valid ? cell[u] : 0u;
    const unsigned long long grp = fq_match_any<ABITS>(s, valid);
    const unsigned r = next_rank[s] + fq_mbcnt(grp);
    __syncthreads();
    if (valid) {
      if (fq_mbcnt(grp) == 0) next_rank[s] = r + (unsigned)__popcll(grp);
      // CTable: stateTable[cumul[s] + r] = size + u
      state_table[cumul[s] + r] = (uint16_t)(size + u);
      // DTable: x = norm'(s) + r; nbBits = t - hb(x); newState = (x << nbBits) - size
      const unsigned ns = cumul[s + 1] - cumul[s];
      const unsigned x = ns + r;
      const unsigned nb = t - hb32(x);
      dt[1 + u] = FQ_DENTRY(((x << nb) - size) & 0x3FFFu, s, nb);
    }
    __syncthreads();
  }
  // symbol transformation table
  if (on) {
    uint32_t *tt = ct + 1 + (size >> 1);
    const unsigned total = cumul[lane];
    if (n == 0) {
      tt[2 * lane + 1] = ((t + 1) << 16) - size;
      tt[2 * lane] = 0;
    } else if (n == -1 || n == 1) {
      tt[2 * lane + 1] = (t << 16) - size;
      tt[2 * lane] = total - 1;
    } else {
      const unsigned max_bits = t - hb32((unsigned)n - 1);
      tt[2 * lane + 1] = (max_bits << 16) - ((unsigned)n << max_bits);
      tt[2 * lane] = total - (unsigned)n;
    }
  }
}

}  // namespace

int fq_build_freq_tables(int device, hipStream_t st, const uint8_t *raw_dev, const fqgpu_rec *recs_dev,
                         size_t n_recs, uint32_t *seq_counts_dev, uint32_t *qual_counts_dev, size_t n_bases, unsigned min_len) {
  (void)device;
  const size_t ns = (size_t)FQGPU_SEQ_MODELS * FQGPU_SEQ_ALPHA, nq = (size_t)FQGPU_QUAL_MODELS * FQGPU_QUAL_ALPHA;
  // counts start at 1 (src/fse_sequence.cpp:148-149, src/fse_quality.cpp:74-75)
  hipLaunchKernelGGL(k_fill_u32, dim3(4), dim3(256), 0, st, seq_counts_dev, ns, 1u);
  hipLaunchKernelGGL(k_fill_u32, dim3(512), dim3(256), 0, st, qual_counts_dev, nq, 1u);
  if (n_recs) {
    const unsigned blocks = (unsigned)min((n_recs + 3) / 4, (size_t)4096);
    uint32_t *err = qual_counts_dev + nq;  // caller provides one extra word
    hipLaunchKernelGGL(k_hist_seq, dim3(blocks), dim3(256), 0, st, raw_dev, recs_dev, (unsigned)n_recs, seq_counts_dev, err);
    // quality counts: through the encoder's sort when the sample is worth it (reads of three symbols or
    // more: what the encoder's record walker is written for); the scattered-atomics kernel otherwise
    if (min_len >= 3 && n_bases >= (1u << 20) && n_bases < 0xFFF00000ull) {
      const int rc = fq_qual_counts_sorted(st, raw_dev, recs_dev, n_recs, n_bases, qual_counts_dev, err);
      if (rc) return rc;
    } else {
      hipLaunchKernelGGL(k_hist_qual, dim3(blocks), dim3(256), 0, st, raw_dev, recs_dev, (unsigned)n_recs,
                         qual_counts_dev, err);
    }
  }
  FQ_HIP(hipGetLastError());
  return FQGPU_OK;
}

int fq_normalize_counts(hipStream_t st, const uint32_t *counts_dev, int n_models, int alpha,
                        int16_t *norm_dev, uint32_t *logs_dev, uint32_t *max_log_dev, uint32_t *err_dev) {
  if (alpha == FQGPU_SEQ_ALPHA)
    hipLaunchKernelGGL(k_normalize<FQGPU_SEQ_ALPHA>, dim3(n_models), dim3(64), 0, st, counts_dev, norm_dev,
                       logs_dev, max_log_dev, err_dev);
  else if (alpha == FQGPU_QUAL_ALPHA)
    hipLaunchKernelGGL(k_normalize<FQGPU_QUAL_ALPHA>, dim3(n_models), dim3(64), 0, st, counts_dev, norm_dev,
                       logs_dev, max_log_dev, err_dev);
  else
    return FQGPU_E_ARG;
  FQ_HIP(hipGetLastError());
  return FQGPU_OK;
}

// One-symbol transition tables of the sequence contexts (FSE_encodeSymbol's state update,
// zstd fse.h, tabulated): next[s][x - size] = (stateTable[(x >> nb) + deltaFindState] - size) * 2,
// nb = (x + deltaNbBits) >> 16.  Stored pre-scaled to the byte offset of the following lookup.
__global__ void __launch_bounds__(256)
k_build_seq_next(const uint32_t *__restrict__ ct, const uint32_t *__restrict__ ct_off, unsigned stride,
                 uint16_t *__restrict__ next1) {
  const uint32_t *tbl = ct + ct_off[blockIdx.x];
  const unsigned log = tbl[0] & 0xFFFFu, size = 1u << log;
  const uint16_t *st = reinterpret_cast<const uint16_t *>(tbl) + 2;
  const uint32_t *tt = tbl + 1 + (size >> 1);
  uint16_t *next = next1 + (size_t)blockIdx.x * stride;
  for (unsigned e = threadIdx.x; e < 4 * size; e += 256) {
    const unsigned s = e >> log, x = size + (e & (size - 1));
    const unsigned nb = (x + tt[2 * s + 1]) >> 16;
    next[e] = (uint16_t)(((unsigned)st[(int)(x >> nb) + (int)tt[2 * s]] - size) * 2u);
  }
}

// two symbols per lookup: next2[s1 | s2 << 2][xi] = next[s2][next[s1][xi]] (same pre-scaling)
__global__ void __launch_bounds__(256)
k_build_seq_next2(const uint32_t *__restrict__ logs, const uint16_t *__restrict__ next1, unsigned stride1,
                  unsigned stride2, uint16_t *__restrict__ next2) {
  const unsigned log = logs[blockIdx.x], size = 1u << log;
  const uint16_t *n1 = next1 + (size_t)blockIdx.x * stride1;
  uint16_t *n2 = next2 + (size_t)blockIdx.x * stride2;
  for (unsigned e = threadIdx.x; e < 16 * size; e += 256) {
    const unsigned xi = e & (size - 1), pc = e >> log, s1 = pc & 3u, s2 = pc >> 2;
    n2[e] = n1[(s2 << log) + ((unsigned)n1[(s1 << log) + xi] >> 1)];
  }
}

// S-fold powers of the one-symbol transitions: pow[s][xi] = next[s] applied S times to xi (same
// pre-scaling): the function of a chain segment that is S times the symbol s (k_seq_setfunc).
// Square and multiply over the bits of S, one workgroup per (context, symbol).
__global__ void __launch_bounds__(256)
k_build_seq_pow(const uint32_t *__restrict__ logs, const uint16_t *__restrict__ next1, unsigned stride, unsigned S,
                uint16_t *__restrict__ pow) {
  __shared__ uint16_t base[1 << 12], acc[1 << 12], tmp[1 << 12];
  const unsigned c = blockIdx.x, s = blockIdx.y;
  const unsigned log = logs[c], size = 1u << log;
  const uint16_t *n1 = next1 + (size_t)c * stride + ((size_t)s << log);
  for (unsigned i = threadIdx.x; i < size; i += blockDim.x) { base[i] = (uint16_t)(n1[i] >> 1); acc[i] = (uint16_t)i; }
  __syncthreads();
  for (unsigned e = S; e; e >>= 1) {
    if (e & 1u) {
      for (unsigned i = threadIdx.x; i < size; i += blockDim.x) tmp[i] = base[acc[i]];
      __syncthreads();
      for (unsigned i = threadIdx.x; i < size; i += blockDim.x) acc[i] = tmp[i];
      __syncthreads();
    }
    if (e > 1u) {
      for (unsigned i = threadIdx.x; i < size; i += blockDim.x) tmp[i] = base[base[i]];
      __syncthreads();
      for (unsigned i = threadIdx.x; i < size; i += blockDim.x) base[i] = tmp[i];
      __syncthreads();
    }
  }
  uint16_t *o = pow + (size_t)c * stride + ((size_t)s << log);
  for (unsigned i = threadIdx.x; i < size; i += blockDim.x) o[i] = (uint16_t)(acc[i] * 2u);
}

// symbols with normalised count 1 or -1 ("reset" symbols of the chain kernels, encode.hip)
__global__ void __launch_bounds__(256)
k_reset_masks(const int16_t *__restrict__ norm, unsigned B, unsigned A, unsigned long long *__restrict__ mask) {
  const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= B) return;
  unsigned long long m = 0;
  for (unsigned s = 0; s < A; s++) {
    const int v = norm[(size_t)c * A + s];
    if (v == 1 || v == -1) m |= 1ull << s;
  }
  mask[c] = m;
}

// Power tables of the sequence contexts for segments of S symbols (kept per handle for a few S:
// the three default segment lengths and the last one set by hand).  Launched on st; the caller
// orders it against the encodes that use it.
int fq_seq_pow_ensure(hipStream_t st, DevTables &t, unsigned n_models, unsigned S) {
  if (!t.next1 || S == 0) return FQGPU_OK;
  for (unsigned i = 0; i < FQ_SEQ_POW_SETS; i++)
    if (t.seq_pow[i] && t.seq_pow_S[i] == S) return FQGPU_OK;
  unsigned slot = FQ_SEQ_POW_SETS - 1;  // the last slot is the one that gets replaced
  for (unsigned i = 0; i < FQ_SEQ_POW_SETS; i++)
    if (!t.seq_pow[i]) { slot = i; break; }
  const unsigned stride = 4u << t.max_log;
  if (!t.seq_pow[slot]) {
    t.seq_pow[slot] = fq_dev_alloc<uint16_t>((size_t)n_models * stride + 64);
    if (!t.seq_pow[slot]) return FQGPU_E_NOMEM;
  }
  t.seq_pow_S[slot] = S;
  hipLaunchKernelGGL(k_build_seq_pow, dim3(n_models, 4), dim3(256), 0, st, t.logs, t.next1, stride, S, t.seq_pow[slot]);
  FQ_HIP(hipGetLastError());
  return FQGPU_OK;
}

// t.norm and t.logs are already on the device; allocates and fills everything else
int fq_build_tables(hipStream_t st, DevTables &t, int n_models, int alpha, uint32_t *err_dev) {
  const unsigned B = (unsigned)n_models;
  t.log_prefix = fq_dev_alloc<uint32_t>(B + 1);
  t.ct_off = fq_dev_alloc<uint32_t>(B);
  t.dt_off = fq_dev_alloc<uint32_t>(B);
  uint32_t *totals = fq_dev_alloc<uint32_t>(4);
  if (!t.log_prefix || !t.ct_off || !t.dt_off || !totals) return FQGPU_E_NOMEM;
  hipLaunchKernelGGL(k_log_prefix, dim3(1), dim3(1024), 0, st, t.logs, B, t.log_prefix, t.ct_off, t.dt_off,
                     (unsigned)alpha, totals);
  uint32_t h_tot[3];
  FQ_HIP(hipMemcpyAsync(h_tot, totals, sizeof(h_tot), hipMemcpyDeviceToHost, st));
  FQ_HIP(hipStreamSynchronize(st));
  (void)hipFree(totals);
  t.ct_words = h_tot[1];
  t.dt_words = h_tot[2];
  t.ct = fq_dev_alloc<uint32_t>(t.ct_words + 16);
  t.dt = fq_dev_alloc<uint32_t>(t.dt_words + 16);
  if (!t.ct || !t.dt) return FQGPU_E_NOMEM;
  FQ_HIP(hipMemsetAsync(t.ct, 0, (t.ct_words + 16) * 4, st));
  FQ_HIP(hipMemsetAsync(t.dt, 0, (t.dt_words + 16) * 4, st));
  if (alpha == FQGPU_SEQ_ALPHA)
    hipLaunchKernelGGL((k_build_tables<FQGPU_SEQ_ALPHA, 2>), dim3(B), dim3(64), 0, st, t.norm, t.logs, t.ct_off,
                       t.dt_off, t.ct, t.dt, err_dev);
  else
    hipLaunchKernelGGL((k_build_tables<FQGPU_QUAL_ALPHA, 6>), dim3(B), dim3(64), 0, st, t.norm, t.logs, t.ct_off,
                       t.dt_off, t.ct, t.dt, err_dev);
  t.reset_mask = fq_dev_alloc<unsigned long long>(B);
  if (!t.reset_mask) return FQGPU_E_NOMEM;
  hipLaunchKernelGGL(k_reset_masks, dim3((B + 255) / 256), dim3(256), 0, st, t.norm, B, (unsigned)alpha, t.reset_mask);
  if (alpha == FQGPU_SEQ_ALPHA) {
    const unsigned stride = 4u << t.max_log;
    t.next1 = fq_dev_alloc<uint16_t>((size_t)B * stride + 64);
    if (!t.next1) return FQGPU_E_NOMEM;
    hipLaunchKernelGGL(k_build_seq_next, dim3(B), dim3(256), 0, st, t.ct, t.ct_off, stride, t.next1);
    if (t.max_log <= 11) {
      t.next2 = fq_dev_alloc<uint16_t>((size_t)B * 4 * stride + 64);
      if (!t.next2) return FQGPU_E_NOMEM;
      hipLaunchKernelGGL(k_build_seq_next2, dim3(B), dim3(256), 0, st, t.logs, t.next1, stride, 4 * stride, t.next2);
    }
    for (unsigned S : {1024u, 2048u, 4096u}) {  // the default segment lengths of the sequence chains (encode.hip: auto_seq_S)
      const int rc = fq_seq_pow_ensure(st, t, B, S);
      if (rc) return rc;
    }
  }
  FQ_HIP(hipGetLastError());
  return FQGPU_OK;
}
