/*
 * oracle/fqc_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's per-block context-modelled FSE coding of
 * bases and qualities (src/fse_sequence.{h,cpp}, src/fse_quality.{h,cpp},
 * src/fse_common.hpp, src/sequtils.h, and the seq/qual part of
 * src/workspace.cpp:14-88).  It is the checker for the HIP path and the CPU
 * baseline of bench.py ("kind": "port"); nothing under fqcomp28_amd/ may call it.
 *
 * Parity status: PARITY UNPINNED against the reference itself -- the reference cannot be built here (its build fetches four
 * repositories over the network) and holds no golden byte
 * vectors for this path (every test is a round-trip).  The zstd
 * primitives are pinned against libzstd.so.1 1.4.8; the model layer is pinned
 * by the reference's own round-trip tests restated in tests/ and by the
 * surveyor's independent regression values.
 */
#ifndef FQC_ORACLE_H
#define FQC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FQO_SEQ_MODELS 256
#define FQO_SEQ_ALPHA 4
#define FQO_QUAL_MODELS 8192
#define FQO_QUAL_ALPHA 64

/* One parsed record: byte offsets of the sequence and quality lines inside the
 * raw block and their common length (reference FastqRecord, src/defs.h:22-32) */
typedef struct {
  uint32_t seq_off, qual_off, len;
} fqo_rec;

/* FreqTable<N,A> POD exactly as the reference dumps it into the archive
 * (src/fse_common.hpp:147-174, src/prepare.cpp:18-20) */
typedef struct {
  int16_t norm[FQO_SEQ_MODELS][FQO_SEQ_ALPHA];
  uint32_t logs[FQO_SEQ_MODELS];
  uint32_t max_log;
} fqo_seq_ft; /* 3076 bytes */

typedef struct {
  int16_t norm[FQO_QUAL_MODELS][FQO_QUAL_ALPHA];
  uint32_t logs[FQO_QUAL_MODELS];
  uint32_t max_log;
} fqo_qual_ft; /* 1081348 bytes */

enum {
  FQO_OK = 0,
  FQO_E_OVERFLOW = -1,   /* reference: endChunk()==0 */
  FQO_E_SHORT_READ = -2, /* len < 3: reference behaviour undefined */
  FQO_E_CORRUPT = -3,
  FQO_E_ARG = -4
};

/* src/fse_sequence.cpp:145-169 / src/fse_quality.cpp:69-97: raw counts (init 1) */
int fqo_seq_counts(const uint8_t *raw, const fqo_rec *recs, size_t n_recs,
                   uint32_t counts[FQO_SEQ_MODELS][FQO_SEQ_ALPHA]); /* FQO_E_ARG: a byte outside ACGTN */
int fqo_qual_counts(const uint8_t *raw, const fqo_rec *recs, size_t n_recs,
                    uint32_t (*counts)[FQO_QUAL_ALPHA]);
/* src/fse_common.hpp:179-200 */
int fqo_seq_ft_from_counts(const uint32_t counts[FQO_SEQ_MODELS][FQO_SEQ_ALPHA], fqo_seq_ft *ft);
int fqo_qual_ft_from_counts(const uint32_t (*counts)[FQO_QUAL_ALPHA], fqo_qual_ft *ft);

/* src/workspace.h:21-35 */
size_t fqo_bound_seq(size_t total_bases);
size_t fqo_bound_qual(size_t total_bases);

/* Encoder/decoder workspace: all CTables/DTables built once
 * (FSE_Encoder/FSE_Decoder ctors, src/fse_common.hpp:46-71,107-127) */
typedef struct fqo_ctx fqo_ctx;
fqo_ctx *fqo_ctx_create(const fqo_seq_ft *sft, const fqo_qual_ft *qft);
void fqo_ctx_destroy(fqo_ctx *c);

/* seq/qual part of CompressionWorkspace::encodeChunk (src/workspace.cpp:14-45).
 * raw is mutated (N -> A) like the reference does.  n_count gets one u16 per
 * record, n_pos one u16 per N (fresh buffers). */
int fqo_encode_block(fqo_ctx *c, uint8_t *raw, const fqo_rec *recs, size_t n_recs,
                     uint8_t *seq_out, size_t seq_cap, size_t *seq_len,
                     uint8_t *qual_out, size_t qual_cap, size_t *qual_len,
                     uint16_t *readlens, uint16_t *n_count, uint16_t *n_pos, size_t *n_pos_len);

/* seq/qual part of DecompressionWorkspace::decodeChunk (src/workspace.cpp:47-88):
 * fills the sequence and quality line bytes of raw_out at the record offsets. */
int fqo_decode_block(fqo_ctx *c, const uint8_t *seq, size_t seq_len, const uint8_t *qual,
                     size_t qual_len, const uint16_t *n_count, size_t n_count_len,
                     const uint16_t *n_pos, size_t n_pos_len, const fqo_rec *recs, size_t n_recs,
                     uint8_t *raw_out);

#ifdef __cplusplus
}
#endif
#endif
