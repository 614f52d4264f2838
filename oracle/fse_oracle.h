/*
 * oracle/fse_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the subset of zstd's FSE/bitstream layer that
 * the reference's hot path calls through zstd's internal headers
 * (reference: src/fse_common.hpp:17-22 includes common/bitstream.h,
 * common/fse.h, compress/hist.h of the un-vendored fork iam28th/zstd@b010526d,
 * cmake/Dependencies.cmake:21-27).  The fork's source is not available here, so
 * the algorithm is restated from zstd's published behaviour
 * and pinned byte-for-byte against the system libzstd.so.1 (1.4.8) exports in
 * tests/test_oracle_zstd.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use anything under oracle/.  The product path (fqcomp28_amd/) never does.
 */
#ifndef FSE_ORACLE_H
#define FSE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FO_MIN_TABLELOG 5
#define FO_DEFAULT_TABLELOG 11
#define FO_MAX_TABLELOG 12

/* FSE_optimalTableLog (fse_common.hpp:191) */
unsigned fo_optimal_table_log(unsigned max_log, size_t src_size, unsigned max_sv);

/* FSE_normalizeCount(..., useLowProbCount) (fse_common.hpp:192-194).
 * Returns the table log, 0 for the RLE case, or a negative value on error. */
int fo_normalize_count(int16_t *norm, unsigned table_log, const uint32_t *count,
                       size_t total, unsigned max_sv, int use_low_prob);

/* zstd memory layouts, in u32 words */
size_t fo_ctable_words(unsigned table_log, unsigned max_sv);
size_t fo_dtable_words(unsigned table_log);

/* FSE_buildCTable_wksp (fse_common.hpp:65-68); 0 on success */
int fo_build_ctable(uint32_t *ct, const int16_t *norm, unsigned max_sv, unsigned table_log);
/* FSE_buildDTable_wksp (fse_common.hpp:121-124); 0 on success */
int fo_build_dtable(uint32_t *dt, const int16_t *norm, unsigned max_sv, unsigned table_log);

/* ---- forward bit writer (BIT_CStream_t) ---- */
typedef struct {
  uint64_t acc;
  unsigned nbits;
  uint8_t *start, *ptr, *end;
} fo_bitw;

int fo_bitw_init(fo_bitw *w, void *dst, size_t cap); /* BIT_initCStream */
static inline void fo_bitw_add(fo_bitw *w, uint64_t value, unsigned nb) {
  /* BIT_addBits: value is masked to nb bits */
  w->acc |= (value & ((nb >= 64) ? ~0ull : ((1ull << nb) - 1))) << w->nbits;
  w->nbits += nb;
}
void fo_bitw_flush_fast(fo_bitw *w); /* BIT_flushBitsFast: no bound check */
void fo_bitw_flush(fo_bitw *w);      /* BIT_flushBits: clamps to end */
size_t fo_bitw_close(fo_bitw *w);    /* BIT_closeCStream: 0 on overflow */

/* ---- encoder state (FSE_CState_t) ---- */
typedef struct {
  uint32_t value;
  const uint16_t *state_table;
  const uint32_t *symbol_tt; /* pairs {deltaFindState, deltaNbBits} */
  unsigned log;
} fo_cstate;

void fo_cstate_init(fo_cstate *s, const uint32_t *ct);                 /* FSE_initCState  */
void fo_cstate_init2(fo_cstate *s, const uint32_t *ct, unsigned sym);  /* FSE_initCState2 */
static inline void fo_encode_symbol(fo_bitw *w, fo_cstate *s, unsigned sym) {
  /* FSE_encodeSymbol */
  const int32_t dfs = (int32_t)s->symbol_tt[2 * sym];
  const uint32_t dnb = s->symbol_tt[2 * sym + 1];
  const uint32_t nb = (s->value + dnb) >> 16;
  fo_bitw_add(w, s->value, nb);
  s->value = s->state_table[(int32_t)(s->value >> nb) + dfs];
}
void fo_cstate_flush(fo_bitw *w, const fo_cstate *s); /* FSE_flushCState */

/* FSE_compress_usingCTable: only used to cross-check the primitives against
 * libzstd (the reference never calls it).  Returns compressed size or 0. */
size_t fo_compress_using_ctable(void *dst, size_t cap, const void *src, size_t n,
                                const uint32_t *ct);

/* ---- backward bit reader (BIT_DStream_t), functional form ----
 * The stream is one little-endian bit array; `pos` is the number of unread
 * bits below the end mark.  Reading nb bits returns bits [pos-nb, pos) with
 * bit pos-1 as the MSB, exactly what BIT_readBits/BIT_reloadDStream deliver. */
typedef struct {
  const uint8_t *src;
  size_t len;
  int64_t pos;  /* may go negative on a corrupt stream */
  int overrun;
} fo_bitr;

int fo_bitr_init(fo_bitr *r, const void *src, size_t len); /* BIT_initDStream */
uint32_t fo_bitr_read(fo_bitr *r, unsigned nb);
static inline int fo_bitr_finished(const fo_bitr *r) { /* BIT_endOfDStream */
  return r->pos == 0 && !r->overrun;
}

typedef struct {
  uint32_t state;
  const uint32_t *table; /* DTable entries (dt + 1) */
} fo_dstate;

void fo_dstate_init(fo_dstate *s, fo_bitr *r, const uint32_t *dt); /* FSE_initDState */
static inline unsigned fo_decode_symbol(fo_dstate *s, fo_bitr *r) { /* FSE_decodeSymbol */
  const uint32_t e = s->table[s->state];
  const unsigned nb = e >> 24;
  const unsigned sym = (e >> 16) & 0xFF;
  s->state = (e & 0xFFFF) + fo_bitr_read(r, nb);
  return sym;
}

/* FSE_decompress_usingDTable equivalent (cross-check only). Returns n decoded or 0. */
size_t fo_decompress_using_dtable(void *dst, size_t n, const void *src, size_t len,
                                  const uint32_t *dt);

#ifdef __cplusplus
}
#endif
#endif
