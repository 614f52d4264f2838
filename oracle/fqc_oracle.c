/*
 * oracle/fqc_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see header).
 *
 * Scalar restatement of the reference's model layer on top of fse_oracle.c.
 * Every function cites the reference lines it follows.
 */
#include "fqc_oracle.h"
#include "fse_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ---- base <-> 2-bit code (src/sequtils.h:7-37, src/fse_sequence.cpp:6-24) ---- */
static inline unsigned base_code(uint8_t b) {
  switch (b) {
  case 'C': return 1;
  case 'G': return 2;
  case 'T': return 3;
  default: return 0; /* 'A'; 'N' has been replaced by 'A' before coding */
  }
}
/* base2bits_arr (src/fse_sequence.cpp:6-14) holds UINT_MAX for every byte but A, C, G, T: the
 * reference indexes out of its tables on such input (assert / undefined).  The oracle refuses it. */
static inline int is_base(uint8_t b) { return b == 'A' || b == 'C' || b == 'G' || b == 'T'; }
static const char CODE_BASE[4] = {'A', 'C', 'G', 'T'};

/* FSE_Sequence::INITIAL_CONTEXT (src/fse_sequence.h:42-63): the four virtual
 * bases in front of a read, nearest first, are T,C,C,T -> 0xD7 */
#define SEQ_INIT_CTX 0xD7u
static const uint8_t VIRT_CODE[4] = {3, 1, 1, 3}; /* position -1, -2, -3, -4 */

/* context of base p: codes of p-1 (bits 7:6), p-2, p-3, p-4 (bits 1:0);
 * closed form of the two loops of SequenceEncoder::encodeRecord
 * (src/fse_sequence.cpp:56-111) */
static inline unsigned seq_ctx_at(const uint8_t *seq, long p) {
  unsigned ctx = 0;
  int k;
  for (k = 1; k <= 4; k++) {
    const long q = p - k;
    const unsigned c = q >= 0 ? base_code(seq[q]) : VIRT_CODE[-q - 1];
    ctx |= c << (2 * (4 - k));
  }
  return ctx;
}

/* FSE_Quality::calcContext (src/fse_quality.h:40-44) */
static inline unsigned qual_ctx(unsigned q, unsigned q1, unsigned q2) {
  unsigned ctx = ((((q1 > q2) ? q1 : q2) << 6) + q) & 0xFFFu;
  ctx += (unsigned)(q1 == q2) << 12;
  return ctx;
}
/* FSE_Quality::symbolToBits (src/fse_quality.h:54-56) */
static inline unsigned qual_sym(uint8_t c) { return (unsigned)c - 33u; }

/* ------------------------------------------------------------------ */

int fqo_seq_counts(const uint8_t *raw, const fqo_rec *recs, size_t n_recs,
                   uint32_t counts[FQO_SEQ_MODELS][FQO_SEQ_ALPHA]) {
  /* FSE_Sequence::calculateFreqTable, src/fse_sequence.cpp:145-169 */
  size_t r, i;
  for (i = 0; i < FQO_SEQ_MODELS; i++)
    counts[i][0] = counts[i][1] = counts[i][2] = counts[i][3] = 1;
  for (r = 0; r < n_recs; r++) {
    const uint8_t *s = raw + recs[r].seq_off;
    unsigned ctx = SEQ_INIT_CTX;
    for (i = 0; i < recs[r].len; i++) {
      unsigned sym;
      if (s[i] == 'N') continue; /* context is NOT advanced (:156-158) */
      if (!is_base(s[i])) return FQO_E_ARG;
      sym = base_code(s[i]);
      counts[ctx][sym]++;
      ctx = (ctx >> 2) + (sym << 6); /* addSymUpper, src/fse_sequence.h:22-24 */
    }
  }
  return FQO_OK;
}

int fqo_qual_counts(const uint8_t *raw, const fqo_rec *recs, size_t n_recs,
                    uint32_t (*counts)[FQO_QUAL_ALPHA]) {
  /* FSE_Quality::calculateFreqTable, src/fse_quality.cpp:69-97 */
  size_t r, i;
  for (i = 0; i < FQO_QUAL_MODELS; i++) {
    unsigned s;
    for (s = 0; s < FQO_QUAL_ALPHA; s++) counts[i][s] = 1;
  }
  for (r = 0; r < n_recs; r++) {
    const uint8_t *qs = raw + recs[r].qual_off;
    unsigned ctx = qual_ctx(0, 0, 0), q1 = 0, q2 = 0;
    for (i = 0; i < recs[r].len; i++) {
      const unsigned q = qual_sym(qs[i]);
      if (q >= FQO_QUAL_ALPHA) return FQO_E_ARG; /* reference: .at() throws (:88) */
      counts[ctx][q]++;
      ctx = qual_ctx(q, q1, q2);
      q2 = q1;
      q1 = q;
    }
  }
  return FQO_OK;
}

/* makeNormalizedFreqTable, src/fse_common.hpp:179-200 */
static int make_ft(const uint32_t *counts, unsigned n_models, unsigned alpha, int16_t *norm,
                   uint32_t *logs, uint32_t *max_log) {
  unsigned ctx;
  *max_log = 0;
  for (ctx = 0; ctx < n_models; ctx++) {
    const uint32_t *c = counts + (size_t)ctx * alpha;
    size_t total = 0;
    unsigned s;
    int got;
    for (s = 0; s < alpha; s++) total += c[s];
    logs[ctx] = fo_optimal_table_log(0, total, alpha - 1);
    got = fo_normalize_count(norm + (size_t)ctx * alpha, logs[ctx], c, total, alpha - 1, 1);
    if (got != (int)logs[ctx]) return FQO_E_ARG;
    if (logs[ctx] > *max_log) *max_log = logs[ctx];
  }
  return FQO_OK;
}

int fqo_seq_ft_from_counts(const uint32_t counts[FQO_SEQ_MODELS][FQO_SEQ_ALPHA], fqo_seq_ft *ft) {
  memset(ft, 0, sizeof(*ft));
  return make_ft(&counts[0][0], FQO_SEQ_MODELS, FQO_SEQ_ALPHA, &ft->norm[0][0], ft->logs,
                 &ft->max_log);
}
int fqo_qual_ft_from_counts(const uint32_t (*counts)[FQO_QUAL_ALPHA], fqo_qual_ft *ft) {
  memset(ft, 0, sizeof(*ft));
  return make_ft(&counts[0][0], FQO_QUAL_MODELS, FQO_QUAL_ALPHA, &ft->norm[0][0], ft->logs,
                 &ft->max_log);
}

/* Workspace::compressBoundSequence / compressBoundQuality, src/workspace.h:21-35 */
size_t fqo_bound_seq(size_t n) {
  if (n < 1024) return (size_t)1024 * FQO_SEQ_MODELS;
  return n / 4 + 1024;
}
size_t fqo_bound_qual(size_t n) {
  const size_t a = (size_t)1024 * FQO_QUAL_MODELS, b = n * 7 / 8 + 1024;
  return a > b ? a : b;
}

/* ------------------------------------------------------------------ */

struct fqo_ctx {
  uint32_t *seq_ct[FQO_SEQ_MODELS], *seq_dt[FQO_SEQ_MODELS];
  uint32_t *qual_ct[FQO_QUAL_MODELS], *qual_dt[FQO_QUAL_MODELS];
  uint32_t *pool;
  fo_cstate *qual_cs; /* 8192 encoder states */
  fo_dstate *qual_ds;
};

fqo_ctx *fqo_ctx_create(const fqo_seq_ft *sft, const fqo_qual_ft *qft) {
  /* FSE_Encoder / FSE_Decoder ctors, src/fse_common.hpp:46-71, 107-127:
   * tables packed back to back */
  fqo_ctx *c = (fqo_ctx *)calloc(1, sizeof(*c));
  size_t words = 0, off = 0;
  unsigned i;
  if (!c) return NULL;
  for (i = 0; i < FQO_SEQ_MODELS; i++)
    words += fo_ctable_words(sft->logs[i], FQO_SEQ_ALPHA - 1) + fo_dtable_words(sft->logs[i]);
  for (i = 0; i < FQO_QUAL_MODELS; i++)
    words += fo_ctable_words(qft->logs[i], FQO_QUAL_ALPHA - 1) + fo_dtable_words(qft->logs[i]);
  c->pool = (uint32_t *)calloc(words, sizeof(uint32_t));
  c->qual_cs = (fo_cstate *)calloc(FQO_QUAL_MODELS, sizeof(fo_cstate));
  c->qual_ds = (fo_dstate *)calloc(FQO_QUAL_MODELS, sizeof(fo_dstate));
  if (!c->pool || !c->qual_cs || !c->qual_ds) { fqo_ctx_destroy(c); return NULL; }
  for (i = 0; i < FQO_SEQ_MODELS; i++) {
    c->seq_ct[i] = c->pool + off;
    off += fo_ctable_words(sft->logs[i], FQO_SEQ_ALPHA - 1);
    c->seq_dt[i] = c->pool + off;
    off += fo_dtable_words(sft->logs[i]);
    if (fo_build_ctable(c->seq_ct[i], sft->norm[i], FQO_SEQ_ALPHA - 1, sft->logs[i]) ||
        fo_build_dtable(c->seq_dt[i], sft->norm[i], FQO_SEQ_ALPHA - 1, sft->logs[i])) {
      fqo_ctx_destroy(c);
      return NULL;
    }
  }
  for (i = 0; i < FQO_QUAL_MODELS; i++) {
    c->qual_ct[i] = c->pool + off;
    off += fo_ctable_words(qft->logs[i], FQO_QUAL_ALPHA - 1);
    c->qual_dt[i] = c->pool + off;
    off += fo_dtable_words(qft->logs[i]);
    if (fo_build_ctable(c->qual_ct[i], qft->norm[i], FQO_QUAL_ALPHA - 1, qft->logs[i]) ||
        fo_build_dtable(c->qual_dt[i], qft->norm[i], FQO_QUAL_ALPHA - 1, qft->logs[i])) {
      fqo_ctx_destroy(c);
      return NULL;
    }
  }
  return c;
}

void fqo_ctx_destroy(fqo_ctx *c) {
  if (!c) return;
  free(c->pool);
  free(c->qual_cs);
  free(c->qual_ds);
  free(c);
}

int fqo_encode_block(fqo_ctx *c, uint8_t *raw, const fqo_rec *recs, size_t n_recs,
                     uint8_t *seq_out, size_t seq_cap, size_t *seq_len, uint8_t *qual_out,
                     size_t qual_cap, size_t *qual_len, uint16_t *readlens, uint16_t *n_count,
                     uint16_t *n_pos, size_t *n_pos_len) {
  fo_bitw sw, qw;
  fo_cstate seq_cs[FQO_SEQ_MODELS];
  fo_cstate *qual_cs = c->qual_cs;
  size_t r, npos_n = 0;
  unsigned i;

  /* FSE_Encoder::startChunk, src/fse_common.hpp:77-83 */
  if (fo_bitw_init(&sw, seq_out, seq_cap) || fo_bitw_init(&qw, qual_out, qual_cap))
    return FQO_E_ARG;
  for (i = 0; i < FQO_SEQ_MODELS; i++) fo_cstate_init(&seq_cs[i], c->seq_ct[i]);
  for (i = 0; i < FQO_QUAL_MODELS; i++) fo_cstate_init(&qual_cs[i], c->qual_ct[i]);

  /* record loop of encodeChunk, src/workspace.cpp:25-31 */
  for (r = 0; r < n_recs; r++) {
    uint8_t *s = raw + recs[r].seq_off;
    const uint8_t *qs = raw + recs[r].qual_off;
    const long L = (long)recs[r].len;
    long p;
    if (L < 3) return FQO_E_SHORT_READ;
    readlens[r] = (uint16_t)L; /* storeAsBytes(r.length, cbs.readlens) */

    /* replaceAndEncodeNs, src/fse_sequence.cpp:35-51 */
    {
      uint16_t cnt = 0, prev = 0;
      for (p = 0; p < L; p++) {
        if (s[p] == 'N') {
          cnt++;
          n_pos[npos_n++] = (uint16_t)(p - prev);
          s[p] = 'A';
          prev = (uint16_t)p;
        } else if (!is_base(s[p])) {
          return FQO_E_ARG;
        }
      }
      n_count[r] = cnt;
    }

    /* SequenceEncoder::encodeRecord, src/fse_sequence.cpp:53-112: last base first,
     * context rolled with addBaseLower (src/fse_sequence.cpp:26-29) */
    {
      unsigned ctx = seq_ctx_at(s, L - 1);
      for (p = L - 1; p >= 0; p--) {
        const long far = p - 5; /* base entering the context of position p-1 */
        const unsigned in = far >= 0 ? base_code(s[far]) : (far >= -4 ? VIRT_CODE[-far - 1] : 0);
        fo_encode_symbol(&sw, &seq_cs[ctx], base_code(s[p]));
        fo_bitw_flush_fast(&sw);
        if (sw.ptr > sw.end) return FQO_E_OVERFLOW;
        ctx = ((ctx << 2) & 0xFFu) | in;
      }
    }

    /* QualityEncoder::encodeRecord, src/fse_quality.cpp:5-53 (L >= 3):
     * sym/q/q1/q2 shifted exactly like the reference loop (:28-31) */
    {
      unsigned sym = qual_sym(qs[L - 1]), q = qual_sym(qs[L - 2]), q1 = qual_sym(qs[L - 3]);
      unsigned q2 = L >= 4 ? qual_sym(qs[L - 4]) : 0;
      if ((sym | q | q1 | q2) >= FQO_QUAL_ALPHA) return FQO_E_ARG;
      for (p = L - 1; p >= 0; p--) {
        fo_encode_symbol(&qw, &qual_cs[qual_ctx(q, q1, q2)], sym);
        fo_bitw_flush_fast(&qw);
        if (qw.ptr > qw.end) return FQO_E_OVERFLOW;
        sym = q;
        q = q1;
        q1 = q2;
        q2 = p >= 4 ? qual_sym(qs[p - 4]) : 0;
        if (q2 >= FQO_QUAL_ALPHA) return FQO_E_ARG;
      }
    }
  }

  /* FSE_Encoder::endChunk, src/fse_common.hpp:86-90 */
  for (i = 0; i < FQO_SEQ_MODELS; i++) fo_cstate_flush(&sw, &seq_cs[i]);
  *seq_len = fo_bitw_close(&sw);
  for (i = 0; i < FQO_QUAL_MODELS; i++) fo_cstate_flush(&qw, &qual_cs[i]);
  *qual_len = fo_bitw_close(&qw);
  *n_pos_len = npos_n;
  if (*seq_len == 0 || *qual_len == 0) return FQO_E_OVERFLOW;
  return FQO_OK;
}

int fqo_decode_block(fqo_ctx *c, const uint8_t *seq, size_t seq_len, const uint8_t *qual,
                     size_t qual_len, const uint16_t *n_count, size_t n_count_len,
                     const uint16_t *n_pos, size_t n_pos_len, const fqo_rec *recs, size_t n_recs,
                     uint8_t *raw_out) {
  fo_bitr sr, qr;
  fo_dstate seq_ds[FQO_SEQ_MODELS];
  fo_dstate *qual_ds = c->qual_ds;
  size_t r, idx_cnt = n_count_len, idx_pos = n_pos_len;
  int i;

  /* FSE_Decoder::startChunk, src/fse_common.hpp:130-138: states read N-1 .. 0 */
  if (fo_bitr_init(&sr, seq, seq_len) || fo_bitr_init(&qr, qual, qual_len)) return FQO_E_CORRUPT;
  for (i = FQO_SEQ_MODELS - 1; i >= 0; i--) fo_dstate_init(&seq_ds[i], &sr, c->seq_dt[i]);
  for (i = FQO_QUAL_MODELS - 1; i >= 0; i--) fo_dstate_init(&qual_ds[i], &qr, c->qual_dt[i]);

  /* second pass of decodeChunk, src/workspace.cpp:84-87: records last -> first */
  for (r = n_recs; r > 0; r--) {
    const fqo_rec *rec = &recs[r - 1];
    uint8_t *s = raw_out + rec->seq_off;
    uint8_t *qs = raw_out + rec->qual_off;
    uint32_t p;
    uint16_t cnt;
    unsigned ctx;

    /* SequenceDecoder::decodeRecord, src/fse_sequence.cpp:114-143 */
    if (idx_cnt < 1) return FQO_E_CORRUPT;
    cnt = n_count[--idx_cnt];
    if (idx_pos < cnt) return FQO_E_CORRUPT;
    idx_pos -= cnt;

    ctx = SEQ_INIT_CTX;
    for (p = 0; p < rec->len; p++) {
      const unsigned sym = fo_decode_symbol(&seq_ds[ctx], &sr);
      s[p] = (uint8_t)CODE_BASE[sym & 3];
      ctx = (ctx >> 2) + (sym << 6);
    }
    {
      uint32_t at = 0;
      uint16_t k;
      for (k = 0; k < cnt; k++) {
        at += n_pos[idx_pos + k];
        if (at >= rec->len) return FQO_E_CORRUPT;
        s[at] = 'N';
      }
    }

    /* QualityDecoder::decodeRecord, src/fse_quality.cpp:55-67 */
    {
      unsigned q1 = 0, q2 = 0;
      ctx = qual_ctx(0, 0, 0);
      for (p = 0; p < rec->len; p++) {
        const unsigned q = fo_decode_symbol(&qual_ds[ctx], &qr);
        qs[p] = (uint8_t)(q + 33);
        ctx = qual_ctx(q, q1, q2);
        q2 = q1;
        q1 = q;
      }
    }
  }
  /* FSE_Decoder::endChunk, src/fse_common.hpp:141 */
  if (!fo_bitr_finished(&sr) || !fo_bitr_finished(&qr)) return FQO_E_CORRUPT;
  return FQO_OK;
}
