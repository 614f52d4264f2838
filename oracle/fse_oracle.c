/*
 * oracle/fse_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see header).
 *
 * Scalar restatement of the zstd FSE subset used by the reference
 * (src/fse_common.hpp: FSE_optimalTableLog/FSE_normalizeCount :191-194,
 * FSE_buildCTable_wksp :65-68, FSE_buildDTable_wksp :121-124,
 * BIT_initCStream/FSE_initCState :79-82, FSE_flushCState/BIT_closeCStream
 * :87-89, BIT_initDStream/FSE_initDState :132-137, BIT_endOfDStream :141;
 * FSE_encodeSymbol/BIT_flushBitsFast at src/fse_sequence.cpp:88-89,109-110 and
 * src/fse_quality.cpp:25-26; FSE_decodeSymbol/BIT_reloadDStream at
 * src/fse_sequence.cpp:131-132 and src/fse_quality.cpp:59-60).
 * Pinned against libzstd.so.1 (1.4.8) in tests/test_oracle_zstd.py.
 */
#include "fse_oracle.h"

#include <string.h>

static unsigned hb32(uint32_t v) { return 31u - (unsigned)__builtin_clz(v); }

static unsigned min_table_log(size_t src_size, unsigned max_sv) {
  const unsigned by_src = hb32((uint32_t)src_size) + 1;
  const unsigned by_sym = hb32(max_sv) + 2;
  return by_src < by_sym ? by_src : by_sym;
}

unsigned fo_optimal_table_log(unsigned max_log, size_t src_size, unsigned max_sv) {
  /* the subtraction wraps in u32 when src_size-1 < 4: kept on purpose */
  const uint32_t max_bits_src = hb32((uint32_t)(src_size - 1)) - 2u;
  uint32_t t = max_log ? max_log : FO_DEFAULT_TABLELOG;
  const unsigned min_bits = min_table_log(src_size, max_sv);
  if (max_bits_src < t) t = max_bits_src;
  if (min_bits > t) t = min_bits;
  if (t < FO_MIN_TABLELOG) t = FO_MIN_TABLELOG;
  if (t > FO_MAX_TABLELOG) t = FO_MAX_TABLELOG;
  return t;
}

/* second-chance normalisation used when the first pass over-allocates */
static int normalize_m2(int16_t *norm, unsigned t, const uint32_t *count, size_t total,
                        unsigned max_sv, int16_t low_prob) {
  const int16_t UNSET = -2;
  uint32_t distributed = 0, to_dist;
  const uint32_t low_thr = (uint32_t)(total >> t);
  uint32_t low_one = (uint32_t)((total * 3) >> (t + 1));
  unsigned s;

  for (s = 0; s <= max_sv; s++) {
    if (count[s] == 0) { norm[s] = 0; continue; }
    if (count[s] <= low_thr) { norm[s] = low_prob; distributed++; total -= count[s]; continue; }
    if (count[s] <= low_one) { norm[s] = 1; distributed++; total -= count[s]; continue; }
    norm[s] = UNSET;
  }
  to_dist = (1u << t) - distributed;
  if (to_dist == 0) return 0;

  if ((total / to_dist) > low_one) {
    low_one = (uint32_t)((total * 3) / (to_dist * 2));
    for (s = 0; s <= max_sv; s++) {
      if (norm[s] == UNSET && count[s] <= low_one) {
        norm[s] = 1; distributed++; total -= count[s];
      }
    }
    to_dist = (1u << t) - distributed;
  }

  if (distributed == max_sv + 1) {
    uint32_t best = 0, best_c = 0;
    for (s = 0; s <= max_sv; s++)
      if (count[s] > best_c) { best = s; best_c = count[s]; }
    norm[best] += (int16_t)to_dist;
    return 0;
  }

  if (total == 0) {
    for (s = 0; to_dist > 0; s = (s + 1) % (max_sv + 1))
      if (norm[s] > 0) { to_dist--; norm[s]++; }
    return 0;
  }

  {
    const uint64_t vsl = 62 - t;
    const uint64_t mid = (1ull << (vsl - 1)) - 1;
    const uint64_t rstep = (((1ull << vsl) * to_dist) + mid) / (uint32_t)total;
    uint64_t run = mid;
    for (s = 0; s <= max_sv; s++) {
      if (norm[s] == UNSET) {
        const uint64_t end = run + (uint64_t)count[s] * rstep;
        const uint32_t w = (uint32_t)(end >> vsl) - (uint32_t)(run >> vsl);
        if (w < 1) return -1;
        norm[s] = (int16_t)w;
        run = end;
      }
    }
  }
  return 0;
}

int fo_normalize_count(int16_t *norm, unsigned t, const uint32_t *count, size_t total,
                       unsigned max_sv, int use_low_prob) {
  static const uint32_t rtb[8] = {0, 473195, 504333, 520860, 550000, 700000, 750000, 830000};
  if (t == 0) t = FO_DEFAULT_TABLELOG;
  if (t < FO_MIN_TABLELOG) return -1;
  if (t > FO_MAX_TABLELOG) return -2;
  if (t < min_table_log(total, max_sv)) return -1;
  {
    const int16_t low_prob = use_low_prob ? -1 : 1;
    const uint64_t scale = 62 - t;
    const uint64_t step = (1ull << 62) / (uint32_t)total;
    const uint64_t vstep = 1ull << (scale - 20);
    int still = 1 << t;
    unsigned s, largest = 0;
    int16_t largest_p = 0;
    const uint32_t low_thr = (uint32_t)(total >> t);

    for (s = 0; s <= max_sv; s++) {
      if (count[s] == total) return 0; /* rle */
      if (count[s] == 0) { norm[s] = 0; continue; }
      if (count[s] <= low_thr) {
        norm[s] = low_prob;
        still--;
      } else {
        int16_t p = (int16_t)(((uint64_t)count[s] * step) >> scale);
        if (p < 8) {
          const uint64_t to_beat = vstep * rtb[p];
          p += ((uint64_t)count[s] * step) - ((uint64_t)p << scale) > to_beat;
        }
        if (p > largest_p) { largest_p = p; largest = s; }
        norm[s] = p;
        still -= p;
      }
    }
    if (-still >= (norm[largest] >> 1)) {
      if (normalize_m2(norm, t, count, total, max_sv, low_prob) != 0) return -1;
    } else {
      norm[largest] += (int16_t)still;
    }
  }
  return (int)t;
}

size_t fo_ctable_words(unsigned t, unsigned max_sv) {
  return 1 + (t ? ((size_t)1 << (t - 1)) : 1) + ((size_t)max_sv + 1) * 2;
}
size_t fo_dtable_words(unsigned t) { return 1 + ((size_t)1 << t); }

/* Distribute symbols over the 2^t table cells; shared by both table kinds.
 * cell[] gets the symbol of every position; returns 0 on success. */
static int spread_symbols(uint8_t *cell, const int16_t *norm, unsigned max_sv, unsigned t) {
  const uint32_t size = 1u << t, mask = size - 1;
  const uint32_t step = (size >> 1) + (size >> 3) + 3;
  uint32_t high = size - 1, pos = 0;
  unsigned s;
  for (s = 0; s <= max_sv; s++)
    if (norm[s] == -1) cell[high--] = (uint8_t)s;
  for (s = 0; s <= max_sv; s++) {
    int i;
    for (i = 0; i < norm[s]; i++) {
      cell[pos] = (uint8_t)s;
      do { pos = (pos + step) & mask; } while (pos > high);
    }
  }
  return pos == 0 ? 0 : -1;
}

int fo_build_ctable(uint32_t *ct, const int16_t *norm, unsigned max_sv, unsigned t) {
  const uint32_t size = 1u << t;
  uint16_t *hdr = (uint16_t *)ct;
  uint16_t *state_table = hdr + 2;
  uint32_t *tt = ct + 1 + (t ? (size >> 1) : 1);
  uint8_t cell[1u << FO_MAX_TABLELOG];
  uint32_t cumul[258];
  unsigned s;
  uint32_t u;

  if (t > FO_MAX_TABLELOG || max_sv > 255) return -1;
  hdr[0] = (uint16_t)t;
  hdr[1] = (uint16_t)max_sv;

  cumul[0] = 0;
  for (s = 1; s <= max_sv + 1; s++)
    cumul[s] = cumul[s - 1] + (norm[s - 1] == -1 ? 1u : (uint32_t)norm[s - 1]);
  cumul[max_sv + 1] = size + 1;

  if (spread_symbols(cell, norm, max_sv, t) != 0) return -1;

  for (u = 0; u < size; u++) {
    const unsigned sy = cell[u];
    state_table[cumul[sy]++] = (uint16_t)(size + u);
  }

  {
    uint32_t total = 0;
    for (s = 0; s <= max_sv; s++) {
      const int n = norm[s];
      if (n == 0) {
        tt[2 * s + 1] = ((t + 1) << 16) - size; /* deltaFindState left as is */
      } else if (n == -1 || n == 1) {
        tt[2 * s + 1] = (t << 16) - size;
        tt[2 * s] = total - 1;
        total++;
      } else {
        const uint32_t max_bits = t - hb32((uint32_t)n - 1);
        const uint32_t min_state_plus = (uint32_t)n << max_bits;
        tt[2 * s + 1] = (max_bits << 16) - min_state_plus;
        tt[2 * s] = total - (uint32_t)n;
        total += (uint32_t)n;
      }
    }
  }
  return 0;
}

int fo_build_dtable(uint32_t *dt, const int16_t *norm, unsigned max_sv, unsigned t) {
  const uint32_t size = 1u << t;
  uint8_t cell[1u << FO_MAX_TABLELOG];
  uint16_t next[256];
  uint16_t fast = 1;
  const int16_t large = (int16_t)(1 << (t - 1));
  unsigned s;
  uint32_t u;

  if (t > FO_MAX_TABLELOG || max_sv > 255) return -1;
  for (s = 0; s <= max_sv; s++) {
    if (norm[s] == -1) {
      next[s] = 1;
    } else {
      if (norm[s] >= large) fast = 0;
      next[s] = (uint16_t)norm[s];
    }
  }
  dt[0] = (uint32_t)t | ((uint32_t)fast << 16);
  if (spread_symbols(cell, norm, max_sv, t) != 0) return -1;
  for (u = 0; u < size; u++) {
    const unsigned sy = cell[u];
    const uint32_t x = next[sy]++;
    const uint32_t nb = t - hb32(x);
    const uint32_t ns = (x << nb) - size;
    dt[1 + u] = (ns & 0xFFFF) | ((uint32_t)sy << 16) | (nb << 24);
  }
  return 0;
}

/* ---------------- bit writer ---------------- */

int fo_bitw_init(fo_bitw *w, void *dst, size_t cap) {
  w->acc = 0;
  w->nbits = 0;
  w->start = w->ptr = (uint8_t *)dst;
  w->end = w->start + cap - sizeof(uint64_t);
  return cap <= sizeof(uint64_t) ? -1 : 0;
}

static void store_le64(uint8_t *p, uint64_t v) {
  int i;
  for (i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (8 * i));
}

void fo_bitw_flush_fast(fo_bitw *w) {
  const unsigned nbytes = w->nbits >> 3;
  store_le64(w->ptr, w->acc);
  w->ptr += nbytes;
  w->nbits &= 7;
  w->acc = nbytes == 8 ? 0 : w->acc >> (nbytes * 8);
}

void fo_bitw_flush(fo_bitw *w) {
  fo_bitw_flush_fast(w);
  if (w->ptr > w->end) w->ptr = w->end;
}

size_t fo_bitw_close(fo_bitw *w) {
  fo_bitw_add(w, 1, 1);
  fo_bitw_flush(w);
  if (w->ptr >= w->end) return 0;
  return (size_t)(w->ptr - w->start) + (w->nbits > 0);
}

/* ---------------- encoder state ---------------- */

void fo_cstate_init(fo_cstate *s, const uint32_t *ct) {
  const uint16_t *h = (const uint16_t *)ct;
  const unsigned t = h[0];
  s->log = t;
  s->value = 1u << t;
  s->state_table = h + 2;
  s->symbol_tt = ct + 1 + (t ? (1u << (t - 1)) : 1);
}

void fo_cstate_init2(fo_cstate *s, const uint32_t *ct, unsigned sym) {
  fo_cstate_init(s, ct);
  {
    const int32_t dfs = (int32_t)s->symbol_tt[2 * sym];
    const uint32_t dnb = s->symbol_tt[2 * sym + 1];
    const uint32_t nb = (dnb + (1u << 15)) >> 16;
    s->value = (nb << 16) - dnb;
    s->value = s->state_table[(int32_t)(s->value >> nb) + dfs];
  }
}

void fo_cstate_flush(fo_bitw *w, const fo_cstate *s) {
  fo_bitw_add(w, s->value, s->log);
  fo_bitw_flush(w);
}

size_t fo_compress_using_ctable(void *dst, size_t cap, const void *src, size_t n,
                                const uint32_t *ct) {
  const uint8_t *const base = (const uint8_t *)src;
  const uint8_t *ip = base + n;
  fo_bitw w;
  fo_cstate c1, c2;
  if (n <= 2) return 0;
  if (fo_bitw_init(&w, dst, cap) != 0) return 0;
  if (n & 1) {
    fo_cstate_init2(&c1, ct, *--ip);
    fo_cstate_init2(&c2, ct, *--ip);
    fo_encode_symbol(&w, &c1, *--ip);
    fo_bitw_flush_fast(&w);
  } else {
    fo_cstate_init2(&c2, ct, *--ip);
    fo_cstate_init2(&c1, ct, *--ip);
  }
  n -= 2;
  if (n & 2) {
    fo_encode_symbol(&w, &c2, *--ip);
    fo_encode_symbol(&w, &c1, *--ip);
    fo_bitw_flush_fast(&w);
  }
  while (ip > base) {
    fo_encode_symbol(&w, &c2, *--ip);
    fo_encode_symbol(&w, &c1, *--ip);
    fo_bitw_flush_fast(&w);
    fo_encode_symbol(&w, &c2, *--ip);
    fo_encode_symbol(&w, &c1, *--ip);
    fo_bitw_flush_fast(&w);
  }
  fo_cstate_flush(&w, &c2);
  fo_cstate_flush(&w, &c1);
  return fo_bitw_close(&w);
}

/* ---------------- bit reader ---------------- */

int fo_bitr_init(fo_bitr *r, const void *src, size_t len) {
  r->src = (const uint8_t *)src;
  r->len = len;
  r->overrun = 0;
  r->pos = 0;
  if (len == 0) return -1;
  {
    const uint8_t last = r->src[len - 1];
    if (last == 0) return -1; /* end mark missing */
    r->pos = (int64_t)(len - 1) * 8 + hb32(last);
  }
  return 0;
}

uint32_t fo_bitr_read(fo_bitr *r, unsigned nb) {
  const int64_t lo = r->pos - (int64_t)nb;
  uint64_t w = 0;
  r->pos = lo;
  if (nb == 0) return 0;
  if (lo < 0) {
    /* zstd keeps decoding zeros past the start and reports overflow at the end */
    int64_t b;
    uint32_t v = 0;
    r->overrun = 1;
    for (b = lo + (int64_t)nb - 1; b >= lo; b--)
      v = (v << 1) | (b >= 0 ? ((r->src[b >> 3] >> (b & 7)) & 1u) : 0u);
    return v;
  }
  {
    const size_t byte = (size_t)(lo >> 3);
    const size_t avail = r->len - byte;
    const size_t n = avail < 8 ? avail : 8;
    size_t i;
    for (i = 0; i < n; i++) w |= (uint64_t)r->src[byte + i] << (8 * i);
  }
  return (uint32_t)((w >> (lo & 7)) & ((1ull << nb) - 1));
}

void fo_dstate_init(fo_dstate *s, fo_bitr *r, const uint32_t *dt) {
  const unsigned t = dt[0] & 0xFFFF;
  s->state = fo_bitr_read(r, t);
  s->table = dt + 1;
}

size_t fo_decompress_using_dtable(void *dst, size_t n, const void *src, size_t len,
                                  const uint32_t *dt) {
  uint8_t *out = (uint8_t *)dst;
  fo_bitr r;
  fo_dstate st[2];
  size_t i;
  if (n < 2 || fo_bitr_init(&r, src, len) != 0) return 0;
  fo_dstate_init(&st[0], &r, dt);
  fo_dstate_init(&st[1], &r, dt);
  for (i = 0; i + 2 < n; i++) out[i] = (uint8_t)fo_decode_symbol(&st[i & 1], &r);
  /* the first two encoded symbols live in the final states */
  for (; i < n; i++) out[i] = (uint8_t)((st[i & 1].table[st[i & 1].state] >> 16) & 0xFF);
  return fo_bitr_finished(&r) ? n : 0;
}
