/*
 * fqgpu.h -- C ABI of the MI355X-native FASTQ block entropy coder.
 *
 * This is the drop-in boundary for ONE hot path of iam28th/fqcomp28: the
 * per-block context-modelled FSE coding of bases and quality scores.  The
 * reference has no FFI layer; the seam is its block codec
 *     CompressionWorkspace::encodeChunk(FastqChunk&, CompressedBuffersDst&)    src/workspace.h:69
 *     DecompressionWorkspace::decodeChunk(FastqChunk&, CompressedBuffersSrc&)  src/workspace.h:112
 *     FSE_{Sequence,Quality}::calculateFreqTable(const FastqChunk&)            src/fse_sequence.h:72, src/fse_quality.h:50
 * and each entry point below names the reference code it replaces.  The C++
 * shim that keeps the reference's Workspace/CompressedBuffers surface on top of
 * this ABI is fqcomp28_amd/csrc/workspace.hpp; INTEGRATION.md shows the patch a
 * maintainer of the reference would apply.
 *
 * Conventions: plain pointers and sizes, little-endian integers, no exceptions
 * across the boundary; every function returns FQGPU_OK (0) or a negative
 * FQGPU_E_* code.  Pointers are HOST memory unless the name ends in _dev.
 * A handle is not re-entrant (like a reference Workspace, src/process.cpp:49-54);
 * different handles are independent and may live on different GPUs.
 * There is NO CPU fallback: every call fails with FQGPU_E_NO_DEVICE without a GPU.
 */
#ifndef FQGPU_H
#define FQGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FQGPU_SEQ_MODELS 256   /* FSE_Sequence::N_MODELS,  src/fse_sequence.h:66 */
#define FQGPU_SEQ_ALPHA 4      /* FSE_Sequence::ALPHABET_SIZE, :36 */
#define FQGPU_QUAL_MODELS 8192 /* FSE_Quality::N_MODELS,   src/fse_quality.h:31 */
#define FQGPU_QUAL_ALPHA 64    /* FSE_Quality::ALPHABET_SIZE, :22 */
#define FQGPU_SEQ_FT_BYTES 3076     /* sizeof(FreqTable<256,4>),   src/fse_common.hpp:147-174 */
#define FQGPU_QUAL_FT_BYTES 1081348 /* sizeof(FreqTable<8192,64>) */

enum {
  FQGPU_OK = 0,
  FQGPU_E_OVERFLOW = -1,   /* stream does not fit the reference capacity rule: endChunk()==0, src/fse_common.hpp:85-90 */
  FQGPU_E_SHORT_READ = -2, /* a read shorter than 3: undefined in the reference (src/fse_quality.cpp:11-12) */
  FQGPU_E_CORRUPT = -3,    /* decode: end mark missing / stream not fully consumed (BIT_endOfDStream, src/fse_common.hpp:141) */
  FQGPU_E_ARG = -4,        /* bad argument, quality above Q63 (src/fse_quality.cpp:88 throws), a sequence byte that is
                            * neither A, C, G, T nor N (base2bits_arr holds UINT_MAX there, src/fse_sequence.cpp:6-14), bad table */
  FQGPU_E_NO_DEVICE = -5,  /* no usable GPU / HIP runtime error: the product path has no CPU fallback */
  FQGPU_E_NOMEM = -6,
  FQGPU_E_HIP = -7,
  FQGPU_E_HEADER = -8      /* a read header the field coder cannot code: a NUMERIC field without digits or outside int32, a
                            * changing STRING field of 255 or more bytes (asserts in the reference: src/headers.cpp:20,83,117) */
};

/* FastqRecord reduced to what this path needs (src/defs.h:22-32): byte offsets
 * of the sequence and quality lines inside the raw block, and their length. */
typedef struct {
  uint32_t seq_off, qual_off, len;
} fqgpu_rec;

typedef struct fqgpu_ctx fqgpu_ctx;

/* ---- lifecycle ------------------------------------------------------- */
int fqgpu_device_count(void);
const char *fqgpu_strerror(int code);
const char *fqgpu_version(void);

/* Workspace::compressBoundSequence / compressBoundQuality (src/workspace.h:21-35) */
size_t fqgpu_bound_seq(size_t total_bases);
size_t fqgpu_bound_qual(size_t total_bases);

/* ---- dataset analysis (replaces FSE_Sequence::calculateFreqTable
 * src/fse_sequence.cpp:145-169, FSE_Quality::calculateFreqTable
 * src/fse_quality.cpp:69-97 and makeNormalizedFreqTable src/fse_common.hpp:179-200).
 * Histograms the parsed sample block on the GPU and normalises every context
 * into the reference's FreqTable POD layout (what DatasetMeta dumps into the
 * archive, src/prepare.cpp:18-20).  seq_counts_out / qual_counts_out are
 * optional (NULL) dumps of the raw u32 counts [256][4] / [8192][64]. */
int fqgpu_freq_tables(int device, const uint8_t *raw, size_t raw_len, const fqgpu_rec *recs,
                      size_t n_recs, void *seq_ft_out, void *qual_ft_out,
                      uint32_t *seq_counts_out, uint32_t *qual_counts_out);
/* Same, from raw counts already on the host (normalisation only, on the GPU). */
int fqgpu_tables_from_counts(int device, const uint32_t *seq_counts, const uint32_t *qual_counts,
                             void *seq_ft_out, void *qual_ft_out);

/* ---- workspace (replaces the SequenceEncoder/QualityEncoder/…Decoder ctors,
 * src/fse_common.hpp:46-71,107-127: 256 + 8192 CTables and DTables are built on
 * the device from the FreqTable PODs).  One per (host thread, GPU). */
int fqgpu_ctx_create(int device, const void *seq_ft, const void *qual_ft, fqgpu_ctx **out);
void fqgpu_ctx_destroy(fqgpu_ctx *ctx);
/* Tuning knobs of the state-chain kernels; results never depend on them.
 * segment: nominal length, in symbols, of the pieces a context's chain is cut into at
 * single-state ("reset") symbols (0 keeps the default).  flags: FQGPU_CHAIN_SEQ_GENERIC
 * runs the sequence stream through the same reset-cut kernel instead of the segment-function
 * kernels (sequence tables have no single-state symbols: every chain is then walked by one lane). */
#define FQGPU_CHAIN_SEQ_GENERIC 1u
int fqgpu_ctx_set_chain_params(fqgpu_ctx *ctx, unsigned segment, unsigned flags);
/* Segment length, in symbols, of the sequence chain kernels: every chain is cut into segments
 * whose exact entry states come from per-segment state functions (0 = default: 4096, 2048 or 1024 by block size; rounded up
 * to a multiple of 1024).  Results never depend on it. */
int fqgpu_ctx_set_seq_segment(fqgpu_ctx *ctx, unsigned symbols);
/* A wave of the segment-function kernel walks up to max_segments consecutive segments in one go
 * (the state sets keep shrinking along the way; 1..16, 0 = default 8), as long as the chain still
 * splits into min_groups such groups (0 = default 16, the waves of a workgroup).  Results never
 * depend on it. */
int fqgpu_ctx_set_seq_group(fqgpu_ctx *ctx, unsigned max_segments, unsigned min_groups);
/* Allocates now what blocks of up to this shape will need (staging block of the host-pointer calls,
 * scratch of every encode lane): a worker calls it while it builds its workspace, so that its
 * first block does not pay for the allocations.  Optional; everything grows on demand. */
int fqgpu_ctx_reserve(fqgpu_ctx *ctx, size_t raw_len, size_t n_recs, size_t n_bases);
/* Number of blocks the handle keeps in flight (encode lanes, 1..8; 0 = the default: four, six for
 * blocks of less than 48 M symbols): each fqgpu_dblock_encode goes to the next lane (own HIP streams
 * and scratch). */
int fqgpu_ctx_set_lanes(fqgpu_ctx *ctx, unsigned lanes);
/* Test hook: copies the device-built tables of one context out in zstd's memory
 * layout (FSE_CTable / FSE_DTable u32 words).  stream: 0 = sequence, 1 = quality. */
int fqgpu_ctx_dump_tables(fqgpu_ctx *ctx, int stream, unsigned model, uint32_t *ctable_out,
                          size_t ctable_cap_words, uint32_t *dtable_out, size_t dtable_cap_words);

/* ---- block encode (replaces the seq/qual part of
 * CompressionWorkspace::encodeChunk, src/workspace.cpp:14-45, i.e.
 * prepareBuffersForEncoding :159-174, startChunk, the per-record
 * replaceAndEncodeNs + SequenceEncoder::encodeRecord + QualityEncoder::encodeRecord
 * loop :25-31, and endChunk).  seq_out/qual_out receive bytes bit-identical to
 * cbs.seq / cbs.qual; readlens_out (n_recs u16) = cbs.readlens; n_count_out
 * (n_recs u16) and n_pos_out (one u16 per N) are what a FRESH CompressedBuffersDst
 * would hold.  seq_cap/qual_cap are the reference capacities (fqgpu_bound_*):
 * exceeding them returns FQGPU_E_OVERFLOW instead of the reference's silent 0.
 * flags: FQGPU_F_WRITE_BACK_N also rewrites N -> A inside `raw` like the reference. */
#define FQGPU_F_WRITE_BACK_N 1u
/* EXTENSION (not part of the reference format): the encode also leaves a
 * decode index per stream -- every `stride` symbols the bit position of the stream, the decoder
 * state of every context and the bytes the context model needs -- so that a decoder can start
 * in the middle of a block.  With both indexes present fqgpu_dblocks_decode runs one lane per
 * (block, stream, stride) instead of one per (block, stream); the streams themselves are the
 * reference's, byte for byte, with or without the index. */
#define FQGPU_F_DECODE_INDEX 2u
int fqgpu_encode_block(fqgpu_ctx *ctx, uint8_t *raw, size_t raw_len, const fqgpu_rec *recs,
                       size_t n_recs, uint8_t *seq_out, size_t seq_cap, size_t *seq_len,
                       uint8_t *qual_out, size_t qual_cap, size_t *qual_len,
                       uint16_t *readlens_out, uint16_t *n_count_out, uint16_t *n_pos_out,
                       size_t n_pos_cap, size_t *n_pos_len, unsigned flags);

/* The same encode in two halves, so that the caller can work while the GPU does (the shim's
 * encodeChunk codes the block's headers in between: src/workspace.cpp:25-31 does both in one
 * per-record loop), and with the record table built on the GPU when the caller has none:
 *   fqgpu_encode_begin    uploads the chunk (recs == NULL: finds its records on the device --
 *                         FastqReader::parseRecords, src/fastq_io.cpp:67-125, which the reference runs
 *                         under the reader mutex, :29-52 -- a trailing partial record is ignored),
 *                         starts the encode and returns; *used_len = bytes up to the last complete record
 *   fqgpu_encode_records  the record table of the block in flight (n_recs entries), without waiting
 *                         for the encode
 *   fqgpu_encode_wait     waits for the encode and reports the sizes of what it produced, so that the
 *                         caller can size its buffers exactly (optional)
 *   fqgpu_encode_end      waits, delivers exactly what fqgpu_encode_block delivers; `raw` may be NULL
 *                         (no N -> A write-back).  Capacities below the stream sizes: FQGPU_E_OVERFLOW.
 *   fqgpu_encode_cancel   drops the block in flight: waits until no copy or kernel of it touches the
 *                         caller's buffers any more (a caller that unwinds between begin and end calls
 *                         this before it lets go of the chunk and the stream buffers)
 * One block in flight per handle (a block begun and never ended is dropped by the next begin); every
 * other call on the handle waits for it. */
int fqgpu_encode_begin(fqgpu_ctx *ctx, const uint8_t *raw, size_t raw_len, const fqgpu_rec *recs, size_t n_recs,
                       unsigned flags, size_t *n_recs_out, size_t *n_bases_out, size_t *used_len);
int fqgpu_encode_records(fqgpu_ctx *ctx, fqgpu_rec *recs_out, size_t cap);
int fqgpu_encode_wait(fqgpu_ctx *ctx, size_t *seq_len, size_t *qual_len, size_t *n_pos_len);
int fqgpu_encode_cancel(fqgpu_ctx *ctx);
int fqgpu_encode_end(fqgpu_ctx *ctx, uint8_t *raw, uint8_t *seq_out, size_t seq_cap, size_t *seq_len,
                     uint8_t *qual_out, size_t qual_cap, size_t *qual_len, uint16_t *readlens_out,
                     uint16_t *n_count_out, uint16_t *n_pos_out, size_t n_pos_cap, size_t *n_pos_len);

/* ---- header fields of the block in flight, coded on the device (replaces the per-record calls of
 * CompressionWorkspace::encodeHeader, src/workspace.cpp:95-126, over FieldStorageDst::storeString /
 * storeNumeric, src/headers.cpp:76-91, 110-120).  The chunk is on the device already and every field is
 * coded against the SAME field of the header in front, which is input: all records at once.
 * Between fqgpu_encode_begin and fqgpu_encode_end / _cancel, in any order with _records / _wait:
 *   fqgpu_encode_headers_begin  field_types[i]: 0 = NUMERIC, 1 = STRING (headers::FieldType, src/headers.h:12);
 *                               separators[i] behind field i (n_fields - 1 of them): HeaderFormatSpeciciation,
 *                               src/headers.h:28-41; first_header: the dataset's first header, '@' included,
 *                               against which the chunk's first header is coded (Workspace::startNewChunk,
 *                               src/workspace.cpp:90-93).  Queues the work and returns.
 *   fqgpu_encode_headers_wait   sizes[i] = FieldStorage sizes of field i; *total_bytes = their sum.
 *                               FQGPU_E_HEADER: *bad_record = the first record whose header cannot be coded
 *   fqgpu_encode_headers_end    out: per field, in order, isDifferentFlag | content | contentLength, the bytes
 *                               the reference's FieldStorageDst holds after the chunk's last header */
#define FQGPU_HDR_MAX_FIELDS 64
typedef struct {
  uint32_t isDifferentFlag, content, contentLength;  /* = FieldStorage::sizes, src/headers.h:60-64 */
} fqgpu_field_sizes;
int fqgpu_encode_headers_begin(fqgpu_ctx *ctx, const uint8_t *field_types, const char *separators, unsigned n_fields,
                               const uint8_t *first_header, size_t first_header_len);
int fqgpu_encode_headers_wait(fqgpu_ctx *ctx, fqgpu_field_sizes *sizes, size_t *total_bytes, size_t *bad_record);
int fqgpu_encode_headers_end(fqgpu_ctx *ctx, uint8_t *out, size_t out_cap);

/* ---- block decode (replaces the second pass of
 * DecompressionWorkspace::decodeChunk, src/workspace.cpp:84-87:
 * SequenceDecoder::decodeRecord src/fse_sequence.cpp:114-143 and
 * QualityDecoder::decodeRecord src/fse_quality.cpp:55-67, records last->first).
 * raw_out is the block skeleton laid out by the first pass (:62-80); only the
 * sequence and quality line bytes are written. */
int fqgpu_decode_block(fqgpu_ctx *ctx, const uint8_t *seq, size_t seq_len, const uint8_t *qual,
                       size_t qual_len, const uint16_t *n_count, size_t n_count_len,
                       const uint16_t *n_pos, size_t n_pos_len, const fqgpu_rec *recs,
                       size_t n_recs, uint8_t *raw_out, size_t raw_len);
/* Extension (nothing in the reference): the same decode with the sidecar the block's encode left when it ran with
 * FQGPU_F_DECODE_INDEX -- fqgpu_encode_index(stream 0 / 1) after fqgpu_encode_wait or _end hands it out, *len alone
 * when out is NULL.  The streams are unchanged; with its index a stream is decoded from every snapshot (one per
 * Mi symbols) at once instead of by one lane from its end.  An index that does not describe the block, or between
 * whose snapshots the strides do not consume exactly the stream's bits, is FQGPU_E_CORRUPT (the states inside a
 * snapshot are taken as they are: store an index under a checksum); length 0 = no index for that stream. */
int fqgpu_encode_index(fqgpu_ctx *ctx, int stream, uint8_t *out, size_t cap, size_t *len);
int fqgpu_decode_block_indexed(fqgpu_ctx *ctx, const uint8_t *seq, size_t seq_len, const uint8_t *qual, size_t qual_len,
                               const uint16_t *n_count, size_t n_count_len, const uint16_t *n_pos, size_t n_pos_len,
                               const fqgpu_rec *recs, size_t n_recs, uint8_t *raw_out, size_t raw_len,
                               const uint8_t *seq_index, size_t seq_index_len, const uint8_t *qual_index, size_t qual_index_len);

/* ---- device-resident block farm ---------------------------------------
 * Blocks stay in HBM: a "dblock" owns device copies of one raw block, its
 * record table and its coded streams.  Used by the pipeline shim to overlap
 * H2D/D2H with coding, by bench.py (timed region starts with inputs resident)
 * and by the many-blocks decode path, where all blocks of a batch are decoded
 * by ONE launch (the format gives a decoder no parallelism inside a stream:
 * SURVEY.md 7.3). */
typedef struct fqgpu_dblock fqgpu_dblock;
int fqgpu_dblock_create(fqgpu_ctx *ctx, const uint8_t *raw, size_t raw_len, const fqgpu_rec *recs,
                        size_t n_recs, fqgpu_dblock **out);
/* Same from an UNPARSED chunk: the record table is built on the GPU (newline scan + 4-line
 * grouping), replacing FastqReader::parseRecords (src/fastq_io.cpp:67-125), which the reference
 * runs serially under the reader mutex (src/fastq_io.cpp:29-52).  A trailing partial record is
 * ignored like the reference's carry-over; malformed input returns FQGPU_E_ARG. */
int fqgpu_dblock_create_from_raw(fqgpu_ctx *ctx, const uint8_t *raw, size_t raw_len, fqgpu_dblock **out);
/* record table / size of a block (any pointer may be NULL; at most cap records are copied) */
int fqgpu_dblock_records(fqgpu_ctx *ctx, const fqgpu_dblock *b, fqgpu_rec *recs_out, size_t cap,
                         size_t *n_recs, size_t *raw_len);
void fqgpu_dblock_destroy(fqgpu_dblock *b);
/* asynchronous on the handle's stream; sizes are valid after fqgpu_sync() */
int fqgpu_dblock_encode(fqgpu_ctx *ctx, fqgpu_dblock *b, unsigned flags);
/* wipes the sequence/quality bytes of the device raw block (decode target) */
int fqgpu_dblock_wipe(fqgpu_ctx *ctx, fqgpu_dblock *b);
/* decodes every block of the batch from its own device-resident streams */
int fqgpu_dblocks_decode(fqgpu_ctx *ctx, fqgpu_dblock *const *blocks, size_t n_blocks);
int fqgpu_sync(fqgpu_ctx *ctx);
/* status/sizes of the last encode/decode of this block.  If that operation is still in flight the
 * call waits for the block's handle first (the lanes run on non-blocking streams), so it never
 * reports stale or zero sizes; fqgpu_dblock_fetch and fqgpu_dblocks_decode do the same. */
int fqgpu_dblock_status(const fqgpu_dblock *b, size_t *seq_len, size_t *qual_len,
                        size_t *n_pos_len, size_t *n_bases);
/* diagnostics of the last encode (after fqgpu_sync): the longest run of symbols one lane
 * had to walk serially, per stream (the latency floor of the chain kernels) */
int fqgpu_dblock_longest_chain(const fqgpu_dblock *b, unsigned *seq_steps, unsigned *qual_steps);
/* diagnostics of the last encode of this block: how many segments of the quality chains
 * (src/fse_quality.cpp:19-52 cut into segments of S symbols) the chain kernels took as
 * counts[0] transparent (a symbol with one table cell inside: the state behind it is known),
 * [1] anchored (a symbol with <= 64 cells: one walk per candidate), [2] uniform (S times one symbol:
 * a power of one transition), [3] opaque (the full entry-state -> exit-state function).  Reads the
 * scratch of the encode lane that coded the block: valid until that lane codes another block. */
int fqgpu_dblock_qual_segment_classes(fqgpu_ctx *ctx, const fqgpu_dblock *b, size_t counts[4]);
/* copies results to the host (synchronous); any pointer may be NULL */
int fqgpu_dblock_fetch(fqgpu_ctx *ctx, const fqgpu_dblock *b, uint8_t *seq_out, uint8_t *qual_out,
                       uint16_t *readlens_out, uint16_t *n_count_out, uint16_t *n_pos_out,
                       uint8_t *raw_out);
/* decode index of one stream (0 = sequence, 1 = quality) of the last encode with
 * FQGPU_F_DECODE_INDEX: size, copy to the host, and the way back for a later decode.  An index
 * is only valid together with the streams it was made for (load it after the streams). */
int fqgpu_ctx_set_index_stride(fqgpu_ctx *ctx, unsigned symbols); /* default 1 Mi, multiple of 64 Ki */
int fqgpu_dblock_index_bytes(const fqgpu_dblock *b, int stream, size_t *bytes);
int fqgpu_dblock_fetch_index(fqgpu_ctx *ctx, const fqgpu_dblock *b, int stream, void *out, size_t cap);
int fqgpu_dblock_load_index(fqgpu_ctx *ctx, fqgpu_dblock *b, int stream, const void *data, size_t len);
/* replaces the block's coded streams with host data (decode of foreign archives) */
int fqgpu_dblock_load_streams(fqgpu_ctx *ctx, fqgpu_dblock *b, const uint8_t *seq, size_t seq_len,
                              const uint8_t *qual, size_t qual_len, const uint16_t *n_count,
                              const uint16_t *n_pos, size_t n_pos_len);

/* Device time per kernel group, measured with HIP events on the streams the kernels are
 * launched on, accumulated from fqgpu_ctx_enable_timing(ctx, 1) until read: kernel_ms =
 * summed duration, kernel_calls = number of launches; total_ms = first start to last end. */
typedef struct {
  float total_ms;
  float kernel_ms[32];
  int kernel_calls[32];
  const char *kernel_name[32];
  int n_kernels;
} fqgpu_timing;
int fqgpu_ctx_enable_timing(fqgpu_ctx *ctx, int on);
int fqgpu_ctx_last_timing(fqgpu_ctx *ctx, fqgpu_timing *out);
/* Restricts the events to the kernel group of that name (NULL or "" = all groups again): two
 * events per launch of that group instead of two per launch of every group -- the events between
 * the kernels of a stream cost about 7 % of a step of 256 MiB blocks. */
int fqgpu_ctx_timing_only(fqgpu_ctx *ctx, const char *name);

/* ---- host helpers of the path's callers (not GPU code) ------------------
 * Minimal 4-line FASTQ parser with the reference's semantics
 * (FastqReader::parseRecords, src/fastq_io.cpp:67-125): returns the number of
 * complete records found (writes at most cap of them) or a negative error. */
long fqgpu_parse_fastq(const uint8_t *raw, size_t len, fqgpu_rec *recs, size_t cap);
/* Deterministic synthetic FASTQ for the BASELINE.json configs:
 * mode 1: 150 bp, N w.p. 0.001, all quals 'I'; mode 2: 150 bp uniform ACGT,
 * Phred ~ round(N(34,5)) clipped to [2,41]; mode 4: length U[50,300], N w.p. 0.01
 * with quality '#'.  Two more modes are NOT BASELINE configs; they probe how the path depends on
 * the data (bench.py: encode_binned_MBps, encode_constant_MBps): mode 3: mode 2's bases with binned
 * qualities -- '#', '-', '8', 'F' at 5/10/15/70 %, the previous position's level kept w.p. 0.85 (no
 * symbol with a single table cell, long runs of one context); mode 5: every base 'A', every quality
 * 'F' (one context per stream); mode 6: mode 2's bases, two quality levels '-' / 'F' i.i.d. at 30/70 %
 * (every quality segment needs its full entry-state -> exit-state function).  Writes whole records
 * only; returns bytes written. */
size_t fqgpu_synth_fastq(uint8_t *dst, size_t cap, int mode, uint64_t seed, uint64_t first_read_id,
                         uint64_t *n_reads_out);


/* Pinned (page-locked) host memory for the buffers that cross PCIe: the shim's FastqChunk::raw_data
 * and CompressedBuffers::seq/qual live in it, so that fqgpu_encode_block / fqgpu_decode_block copy
 * at the full link rate and asynchronously.  Without a usable GPU the memory is ordinary heap
 * memory (host-only tools and tests still run); there is still no compute fallback.
 * fqgpu_host_free keeps pinned blocks in a cache (hipHostFree waits until the device is idle: a
 * worker thread freeing a buffer would wait for every other worker's kernels); fqgpu_host_trim
 * returns the cache to the system (bytes freed).  Limit of the cache: FQGPU_PINNED_CACHE_MB (8192). */
void *fqgpu_host_alloc(size_t bytes);
void fqgpu_host_free(void *p);
size_t fqgpu_host_trim(void);

/* memcompress / memdecompress (src/memcompress.h:5-28) for the misc streams -- readlens, n_count,
 * n_pos and the header field streams, src/workspace.cpp:176-256.  The reference uses libbsc
 * (third party, source absent): the compressed BYTES are this library's own format and out of
 * parity scope; the contract is the reference's: dst holds src_size + 28 bytes
 * (extra_csize_misc, src/workspace.h:18), empty in = empty out, the original size travels in the
 * container.  fqgpu_memdecompress returns dst_size, 0 for an empty input, (size_t)-1 if malformed. */
size_t fqgpu_memcompress_bound(size_t src_size);
size_t fqgpu_memcompress(uint8_t *dst, size_t dst_cap, const uint8_t *src, size_t src_size);
size_t fqgpu_memdecompress(uint8_t *dst, size_t dst_size, const uint8_t *src, size_t src_size);

#ifdef __cplusplus
}
#endif
#endif
