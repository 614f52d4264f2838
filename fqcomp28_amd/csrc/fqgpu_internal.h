// Internal declarations shared by the HIP translation units of libfqgpu.so.
// gfx950 (MI355X, wave64) only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/fqgpu.h"

#define FQ_WAVE 64

// ---------------------------------------------------------------- models
// Context/symbol definition of the two streams (reference a2/a4 in SURVEY.md 8(a)).
struct SeqModel {
  static constexpr int B = FQGPU_SEQ_MODELS;  // contexts
  static constexpr int A = FQGPU_SEQ_ALPHA;   // alphabet
  static constexpr int KEYBITS = 8;
  static constexpr int STREAM = 0;
};
struct QualModel {
  static constexpr int B = FQGPU_QUAL_MODELS;
  static constexpr int A = FQGPU_QUAL_ALPHA;
  static constexpr int KEYBITS = 13;
  static constexpr int STREAM = 1;
};

// Device-side tables of one stream, built once per handle.
// Decoder entry (one per state, behind the table's zstd header word): the fields of zstd's FSE_decode_t
// arranged for the decode walk -- [31:16] newState * 4 (byte offset inside the table), [12:9] nbBits,
// [8:3] symbol (= symbol * 8, the stride of the walk's per-context LDS slots).  fqgpu_ctx_dump_tables
// hands out zstd's layout {u16 newState; u8 symbol; u8 nbBits}.
#define FQ_DENTRY(new_state, sym, nb) ((((uint32_t)(new_state)) << 18) | ((uint32_t)(nb) << 9) | ((uint32_t)(sym) << 3))
#define FQ_DENTRY_TO_ZSTD(w) ((((uint32_t)(w)) >> 18) | (((((uint32_t)(w)) >> 3) & 63u) << 16) | (((((uint32_t)(w)) >> 9) & 15u) << 24))

struct DevTables {
  int16_t *norm = nullptr;      // [B][A]
  uint32_t *logs = nullptr;     // [B]
  uint32_t *log_prefix = nullptr;  // [B+1] exclusive prefix of logs (bit offsets of the state flush)
  uint32_t *ct = nullptr;       // CTable pool, zstd word layout per context
  uint32_t *ct_off = nullptr;   // [B] word offset of each CTable
  uint32_t *dt = nullptr;       // DTable pool, zstd word layout per context
  uint32_t *dt_off = nullptr;   // [B]
  unsigned long long *reset_mask = nullptr;  // [B] bit s set: symbol s has normalised count 1 or -1
  uint16_t *next1 = nullptr;    // sequence stream only: [B][4 << max_log] one-symbol transition tables,
                                // next[s][x - size] = ((state after coding s in state x) - size) * 2
  uint16_t *next2 = nullptr;    // sequence stream, max_log <= 11: [B][16 << max_log] two-symbol tables,
                                // next2[s1 | s2 << 2][x - size] = next[s2][next[s1][x - size]]
  uint16_t *seq_pow[4] = {nullptr, nullptr, nullptr, nullptr};  // sequence stream: next1-shaped tables of S-fold powers,
  unsigned seq_pow_S[4] = {0, 0, 0, 0};                         // pow[s][x - size] = next[s]^S, for the S named here
  uint32_t max_log = 0;
  size_t ct_words = 0, dt_words = 0;
};
#define FQ_SEQ_POW_SETS 4

// Result block of one (block, stream) coding job, in device memory.
struct StreamResult {
  unsigned long long total_bits;  // payload bits (before state flush)
  unsigned long long len;         // bytes of the finished stream
  unsigned int overflow;          // reference capacity rule violated
  unsigned int bad_symbol;        // quality above Q63 seen
  unsigned int refixed;           // segments the fix-up pass had to re-run
  unsigned int corrupt;           // decode: bad end mark / bits left over
};

struct BlockResult {
  StreamResult s[2];
  unsigned long long n_pos_len;  // number of u16 entries in n_pos
};

// ---------------------------------------------------------------- helpers
#define FQ_HIP(call)                                                 \
  do {                                                               \
    hipError_t e_ = (call);                                          \
    if (e_ != hipSuccess) return fq_hip_error(e_, __FILE__, __LINE__); \
  } while (0)

int fq_hip_error(hipError_t e, const char *file, int line);

template <class T> static inline T *fq_dev_alloc(size_t n) {
  void *p = nullptr;
  if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) return nullptr;
  return reinterpret_cast<T *>(p);
}

// Growable device buffer (never shrinks): the per-handle scratch.
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return FQGPU_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = bytes + bytes / 8 + 256;
    if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return FQGPU_E_NOMEM; }
    cap = want;
    return FQGPU_OK;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct KernelTimer;  // api.cpp

// grow-only scratch of the device record parser (parse.hip)
struct ParseScratch {
  DevBuf cnt, base, sum, nl_pos, scan_tmp;
  size_t total_nl = 0;  // newlines counted by fq_parse_count
  void release() { for (DevBuf *b : {&cnt, &base, &sum, &nl_pos, &scan_tmp}) b->release(); total_nl = 0; }
};

// Result words of the device header coder (headers.hip), and its grow-only scratch
struct HdrResult {
  unsigned long long first_error;  // record << 8 | code (1: numeric field, 2: string field too long, 3: the streams would not fit -- a record
                                   // table that is not this chunk's) of the first header that cannot be coded; ~0: none
  unsigned long long total;        // bytes of the output buffer in use
  unsigned long long off[3 * FQGPU_HDR_MAX_FIELDS];  // per field: flags, content, lengths
  uint32_t size[3 * FQGPU_HDR_MAX_FIELDS];
};
struct HdrScratch {
  DevBuf wg_sum, wg_base, out, first, res;
  HdrResult *host_res = nullptr;  // page-locked
  uint8_t *host_first = nullptr;  // page-locked staging of the dataset's first header
  unsigned n_fields = 0;
  size_t bound = 0;
  bool pending = false;           // kernels queued for the block in flight, results not collected yet
  bool collected = false;
  void release() {
    for (DevBuf *b : {&wg_sum, &wg_base, &out, &first, &res}) b->release();
    if (host_res) (void)hipHostFree(host_res);
    if (host_first) (void)hipHostFree(host_first);
    host_res = nullptr; host_first = nullptr; pending = collected = false;
  }
};

// Scratch of the encode pipeline for one stream.
struct EncScratch {
  DevBuf slot_of;     // u32 [M]   sorted position of every symbol (encode order)
  DevBuf keys;        // u32 [M]   ctx | sym << 16 of every symbol (encode order)
  DevBuf sorted_sym;  // u8  [M+pad]
  DevBuf out16;       // u16 [M+pad] (nb<<12 | bits) at sorted position
  DevBuf tile_hist;   // u32 [tiles][B]
  DevBuf tile_base;   // u32 [tiles][B]
  DevBuf group_sum;   // u32 [groups][B]
  DevBuf ctx_arrays;  // ctx_count[B], ctx_start[B+1], seg_base[B+1], item_base[B+1]
  DevBuf seg_state;   // u16 final_state[B]
  DevBuf seg_arrays;  // generic chain kernels: per-segment tables (encode.hip: SegArrays)
  DevBuf seq_bdesc;   // sequence stream: batch descriptors of the batch-sorted partition (encode.hip: SeqBatchDesc)
  DevBuf seq_plan;    // segment plan of the sequence chains (encode.hip: SEGPLAN_WORDS) + entry states
  DevBuf seq_fbuf;    // u16 [segments][1 << max_log] segment functions F: entry state -> exit state
  DevBuf seq_cbuf;    // sequence stream: composed functions and entry states of the items of long chains (k_seq_compose)
  DevBuf tile_bits;   // u32 [ptiles]
  DevBuf tile_bit_base;  // u64 [ptiles+1]
  DevBuf dbg_enc16;   // timing experiments only (FQGPU_DEBUG_NO_ALIAS): enc16 apart from the keys, so that stale keys stay valid
  DevBuf scan_tmp;    // u64 chunk sums for the scans
  DevBuf tile_runs;   // tile-sorted path: uint2 [tiles][min(B, tile)] run list of every tile (encode.hip: enc_tile_sort.h)
  DevBuf tile_sync;   // tile-sorted path: u64 status[tiles] | u32 tile counter, pad | u32 run_count[tiles]
};

// One encode lane: everything a block needs while it is being coded, so that several
// blocks can be in flight on one GPU (two HIP streams: sequence pipeline, quality pipeline).
struct EncLane {
  hipStream_t st_seq = nullptr, st_qual = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  DevBuf rec_start;   // u32 [R+1] first encode index of each record
  DevBuf n_cnt32;     // u32 [R] N count | u32 [R] length
  DevBuf n_off;       // u32 [R+1]
  DevBuf first_sym;   // u8 [R] sequence | u8 [R] quality: every record's first symbol in encode order (fused K1 -> K3)
  DevBuf rscan;       // k_record_scan: u32 ticket, pad | u64 status[chunks] (zero when allocated; epochs and tickets count on)
  unsigned rscan_epoch = 0, rscan_tickets = 0;  // launches so far (24 bits used) / tickets handed out so far
  DevBuf scan_tmp;
  EncScratch enc[2];
};

#define FQ_MAX_LANES 8
#define FQ_RECENT_BLOCKS 8

struct fqgpu_ctx {
  int device = 0;
  hipStream_t stream = nullptr;  // uploads, decode
  hipStream_t dec_stream2 = nullptr;  // decode: the sequence streams, beside the quality streams on `stream`
  hipEvent_t dec_fork = nullptr, dec_join = nullptr;
  DevTables tab[2];
  unsigned seg_len = 0;          // segment of the generic chain kernels (0 = by block size: 1024..4096)
  int seq_generic = 0;           // 1: sequence stream also uses the reset-cut kernel
  unsigned seq_segment = 0;      // segment length of the sequence chain kernels (0 = default)
  unsigned seq_group = 8;        // segments a k_seq_setfunc wave walks in one go, at most (<= SETS_MAX_GROUP)
  unsigned seq_group_min = 16;   // ... as long as a chain keeps this many groups (one per wave of a workgroup)
  bool lds_atomics_ordered = false;  // probed at creation: k_scatter may rank with LDS atomics
  bool tile_sorted = true;           // tile-sorted partition + fused gather/pack (needs lds_atomics_ordered); false: slot-based path
  unsigned index_stride = 1u << 20;  // symbols between the snapshots of a decode index
  unsigned n_cus = 256;          // compute units of the device
  unsigned setfunc_wgs = 0;      // persistent workgroups of k_seq_setfunc (0 = default, see fqgpu_ctx_create)
  unsigned n_lanes = 0, next_lane = 0;  // n_lanes 0 = by block size: fq_lanes_for()
  const void *recent[FQ_RECENT_BLOCKS] = {};  // the blocks coded last (compared, never followed: api.hip fq_next_lane)
  unsigned recent_at = 0;
  EncLane lanes[FQ_MAX_LANES];
  // decode scratch
  DevBuf n_cnt32, n_off, scan_tmp;
  DevBuf dec_desc;    // decode job descriptors
  DevBuf dec_chunks, dec_recstart;  // chunk list and per-block rec_start of the indexed decode
  KernelTimer *timer = nullptr;
  // staging block of the host-pointer calls, kept between calls (grow-only device buffers)
  fqgpu_dblock *hp_block = nullptr;
  size_t hp_raw = 0, hp_recs = 0, hp_seq = 0, hp_qual = 0, hp_side = 0, hp_npos = 0;  // allocated elements
  hipEvent_t hp_ev_h2d = nullptr;     // "the block's inputs have arrived": the lane's streams wait for it
  BlockResult *hp_result = nullptr;   // page-locked landing place of the result block
  ParseScratch hp_parse;              // fqgpu_encode_begin without a record table: the table is built on the device
  bool hp_pending = false;            // a block of fqgpu_encode_begin is in flight (fqgpu_encode_end collects it)
  unsigned hp_flags = 0;
  hipStream_t hp_done = nullptr;      // the stream its last kernel runs on
  size_t hp_used = 0;                 // bytes of the chunk up to its last complete record
  HdrScratch hp_hdr;                  // fqgpu_encode_headers_*: the header fields of the block in flight
};

EncLane *fq_next_lane(fqgpu_ctx *ctx, size_t n_bases, fqgpu_dblock *b = nullptr);  // api.hip: the next lane in turn or the block's own; creates streams on first use
// Blocks a handle keeps in flight when the caller has not said (fqgpu_ctx_set_lanes(ctx, 0), the default): four -- two
// already fill the chip with 256 MiB blocks --, six for blocks of less than 48 M symbols (about 100 MiB), whose chains of
// short kernels leave more gaps to fill: 16 x 64 MiB blocks 70.6 -> 75.6 GB/s (eight: 71.2).
static inline unsigned fq_lanes_for(const fqgpu_ctx *ctx, size_t n_bases) {
  return ctx->n_lanes ? ctx->n_lanes : (n_bases && n_bases < ((size_t)48 << 20) ? 6u : 4u);
}

struct fqgpu_dblock {
  int device = 0;
  fqgpu_ctx *owner = nullptr;  // the handle whose lanes code this block (status / fetch synchronise it)
  uint8_t *raw = nullptr;
  size_t raw_len = 0;
  fqgpu_rec *recs = nullptr;
  size_t n_recs = 0;
  size_t n_bases = 0;
  uint8_t *seq = nullptr;  size_t seq_cap = 0;   // capacities the overflow rule is judged against (reference rule or the caller's)
  uint8_t *qual = nullptr; size_t qual_cap = 0;
  size_t seq_alloc = 0, qual_alloc = 0;          // bytes allocated behind seq / qual (>= cap + 64)
  uint16_t *readlens = nullptr, *n_count = nullptr, *n_pos = nullptr;
  size_t n_pos_cap = 0;
  BlockResult *result = nullptr;  // device
  BlockResult host_result;        // filled by fqgpu_sync-ing calls
  size_t seq_len = 0, qual_len = 0, n_pos_len = 0;  // stream sizes used by decode
  // "the block's last encode is through", recorded behind its last kernel: the next encode of the SAME block -- on whichever lane it
  // lands -- waits for it, so that two encodes of one block never write its streams and result words at the same time (a caller
  // that queues a block again without a sync in between, with more lanes than blocks; recorded on and waited for by streams of
  // one lane it is free)
  hipEvent_t ev_encoded = nullptr;
  int home_lane = -1;  // the lane of the block's last encode (api.hip: fq_next_lane)
  int last_op = 0;  // 1 = encode, 2 = decode: which fields of the result block are meaningful
  bool result_pulled = true;  // host_result / stream sizes reflect the last launched operation
  // decode index (extension): device copy per stream, valid bytes, allocated bytes
  uint8_t *index[2] = {nullptr, nullptr};
  size_t index_bytes[2] = {0, 0}, index_cap[2] = {0, 0};
  // diagnostics (fqgpu_dblock_qual_segment_classes): where the lane that coded this block last left the classes of its
  // quality segments -- lane scratch, valid until that lane codes another block
  const uint8_t *diag_cls = nullptr;
  const uint32_t *diag_n_segs = nullptr;
};

// Decode index of one stream: header, then one snapshot per multiple of `stride` symbols.
struct FqIndexHeader {
  uint32_t magic;    // 'FQIX'
  uint32_t stream;   // 0 = sequence, 1 = quality
  uint32_t stride;   // symbols between snapshots (multiple of 65536)
  uint32_t n_snap;   // snapshots: encode indices stride, 2 stride, ... < n_sym
  uint64_t n_sym;
  uint64_t reserved;
};
// snapshot k (k = 1 .. n_snap) at encode index e = k * stride, FQ_INDEX_SNAP_HEAD + 2 B bytes:
//   u64 bitpos     bits of the stream in front of symbol e
//   u8  prev[4]    bytes at positions p-1 .. p-4 of the record of symbol e-1 (p = its position;
//                  0xFF in front of the read): what the context model has seen when the decoder,
//                  coming from the stream's end, is about to decode symbol e-1
//   u32 reserved
//   u16 state[B]   decoder state (encoder state - table size) of every context at that point
constexpr unsigned FQ_INDEX_MAGIC = 0x58495146u, FQ_INDEX_SNAP_HEAD = 16;
static inline size_t fq_index_snap_bytes(unsigned B) { return FQ_INDEX_SNAP_HEAD + 2 * (size_t)B; }

// ---------------------------------------------------------------- launches (encode.hip / decode.hip / tables.hip)
int fq_build_freq_tables(int device, hipStream_t st, const uint8_t *raw_dev, const fqgpu_rec *recs_dev,
                         size_t n_recs, uint32_t *seq_counts_dev, uint32_t *qual_counts_dev, size_t n_bases, unsigned min_len);
int fq_normalize_counts(hipStream_t st, const uint32_t *counts_dev, int n_models, int alpha,
                        int16_t *norm_dev, uint32_t *logs_dev, uint32_t *max_log_dev, uint32_t *err_dev);
int fq_build_tables(hipStream_t st, DevTables &t, int n_models, int alpha, uint32_t *err_dev);
int fq_seq_pow_ensure(hipStream_t st, DevTables &t, unsigned n_models, unsigned S);

// wait: event the lane's first kernel waits for (inputs arriving on another stream), or nullptr;
// done: receives the stream on which the block's last kernel was launched (copies of the results
// ordered behind the encode go there), may be nullptr
int fq_encode_launch(fqgpu_ctx *ctx, fqgpu_dblock *b, unsigned flags, hipEvent_t wait = nullptr,
                     hipStream_t *done = nullptr, bool reserve_only = false);
int fq_probe_lds_atomic_order(hipStream_t st, bool *ordered);
int fq_decode_launch(fqgpu_ctx *ctx, fqgpu_dblock *const *blocks, size_t n_blocks);
int fq_wipe_launch(fqgpu_ctx *ctx, fqgpu_dblock *b);
int fq_qual_counts_sorted(hipStream_t st, const uint8_t *raw_dev, const fqgpu_rec *recs_dev, size_t n_recs, size_t n_bases,
                          uint32_t *counts_dev, uint32_t *err_dev);
int fq_parse_count(hipStream_t st, const uint8_t *raw_dev, size_t raw_len, ParseScratch &ps, size_t *n_recs);
size_t fq_headers_bound(size_t raw_len, size_t n_recs, size_t n_bases, unsigned n_fields);
int fq_headers_launch(hipStream_t st, const uint8_t *raw_dev, size_t raw_len, const fqgpu_rec *recs_dev, size_t n_recs, size_t n_bases,
                      const uint8_t *field_types, const char *separators, unsigned n_fields, const uint8_t *first_header,
                      size_t first_header_len, HdrScratch &hs);
int fq_parse_records(hipStream_t st, const uint8_t *raw_dev, size_t raw_len, ParseScratch &ps, fqgpu_rec *recs_dev,
                     size_t n_recs, size_t *n_bases, size_t *n_n, size_t *used_len);

// generic exclusive scans (scan.hip): out has n+1 entries, out[n] = total
int fq_scan_u32_to_u32(hipStream_t st, const uint32_t *in, size_t n, uint32_t *out, DevBuf &tmp);
int fq_scan2_u32_to_u32(hipStream_t st, const uint32_t *a, const uint32_t *b, size_t n, uint32_t *out_a,
                        uint32_t *out_b, DevBuf &tmp);
int fq_scan_u32_to_u64(hipStream_t st, const uint32_t *in, size_t n, unsigned long long *out, DevBuf &tmp);

// kernel timing hooks (api.hip): HIP events on the stream the kernels are launched on
void fq_timer_span_begin(fqgpu_ctx *ctx, const char *name, hipStream_t st);
void fq_timer_span_end(fqgpu_ctx *ctx, hipStream_t st);

// ---------------------------------------------------------------- device helpers
#if defined(__HIPCC__)

__device__ __forceinline__ unsigned fq_lane() { return threadIdx.x & 63u; }

// v_readfirstlane as an UNSIGNED value.  The builtin's type is int: `u64 | __builtin_amdgcn_readfirstlane(x)`
// sign-extends, i.e. ORs thirty-two ones into the upper half whenever bit 31 of x is set.  That -- not a
// hardware hazard -- was the "wrong bits" of round 2's scalar bit window in the decode walk (the ISA showed
// s_ashr_i32 hi, lo, 31 in front of the s_or_b64 that slides the window; tests/test_build_invariants.py
// keeps both out of decode.hip).  decode.hip uses this wrapper only.
__device__ __forceinline__ unsigned fq_uniform(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }

// LDS ordering inside ONE wave (single-wave workgroups): the LDS executes a wave's
// instructions in order, so all that is needed is that earlier LDS operations have completed
// and that the compiler does not move LDS accesses across this point.  __syncthreads() would
// also drain vmcnt(0) -- every outstanding global load AND store -- once per loop iteration.
__device__ __forceinline__ void fq_lds_wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (each with its own
// 4 MiB L2), so with the identity mapping neighbouring tiles never share an L2.  This
// bijective remap gives every XCD a contiguous range of tiles, walked in order: tiles that
// run at the same time on one XCD then touch adjacent bytes of every context's run and the
// partial lines of the permutation passes are completed inside L2 instead of in HBM.
__device__ __forceinline__ unsigned fq_xcd_tile(unsigned b, unsigned n) {
  const unsigned q = n >> 3, r = n & 7u, xcd = b & 7u, idx = b >> 3;
  return (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + idx;
}

// number of set bits of m below the calling lane
__device__ __forceinline__ unsigned fq_mbcnt(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// lanes of the wave holding the same key as the caller (among valid lanes).
// The partition loops are VALU-bound on this function (SQ counters: 164 VALU instructions per
// 64-symbol iteration for 13-bit keys), so it is written for instruction count: the per-lane
// mask is kept as two 32-bit halves of MISMATCH bits, two key bits are folded per v_or3.
template <int BITS>
__device__ __forceinline__ unsigned long long fq_match_any(unsigned key, bool valid) {
  const unsigned long long vm = __ballot(valid);
  unsigned mis_lo = 0, mis_hi = 0;  // lanes whose key differs from mine in some bit
#pragma unroll
  for (int b = 0; b + 1 < BITS; b += 2) {
    const int m0 = __builtin_amdgcn_sbfe((int)key, b, 1), m1 = __builtin_amdgcn_sbfe((int)key, b + 1, 1);  // 0 / -1
    const unsigned long long b0 = __ballot(m0 != 0), b1 = __ballot(m1 != 0);
    mis_lo |= ((unsigned)b0 ^ (unsigned)m0) | ((unsigned)b1 ^ (unsigned)m1);
    mis_hi |= ((unsigned)(b0 >> 32) ^ (unsigned)m0) | ((unsigned)(b1 >> 32) ^ (unsigned)m1);
  }
  if (BITS & 1) {
    const int m0 = __builtin_amdgcn_sbfe((int)key, BITS - 1, 1);
    const unsigned long long b0 = __ballot(m0 != 0);
    mis_lo |= (unsigned)b0 ^ (unsigned)m0;
    mis_hi |= (unsigned)(b0 >> 32) ^ (unsigned)m0;
  }
  return vm & ~(((unsigned long long)mis_hi << 32) | mis_lo);
}

// A0 C1 G2 T3, everything else (N, which the coder treats as A: src/fse_sequence.cpp:44) -> 0
// (three independent selects: the nested form compiles to a chain of exec-mask branches, five
// times per symbol in K1)
__device__ __forceinline__ unsigned fq_base_code(unsigned c) {
  return (c == 'C' ? 1u : 0u) | (c == 'G' ? 2u : 0u) | (c == 'T' ? 3u : 0u);
}
// The reference's base2bits_arr (src/fse_sequence.cpp:6-14) maps everything but A, C, G, T to
// UINT_MAX -- lowercase, IUPAC codes, '.', a stray '\r' index out of its tables (assert / UB); N is
// legal because replaceAndEncodeNs (:35-51) has turned it into 'A' before.  A lossless coder must
// not code such a byte as 'A' silently: symbol code as above, 4 for a byte that is not a base.
__device__ __forceinline__ unsigned fq_base_sym(unsigned c) {
  return (c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N') ? fq_base_code(c) : 4u;
}
// the same with bit 6 set for 'N' (K1 of the tile-sorted path counts the N's of every read on the way)
constexpr unsigned FQ_SYM_IS_N = 0x40u, FQ_SYM_BAD_MASK = 0x3Cu;
__device__ __forceinline__ unsigned fq_base_sym_n(unsigned c) { return fq_base_sym(c) | (c == 'N' ? FQ_SYM_IS_N : 0u); }

// FSE_Quality::calcContext (src/fse_quality.h:40-44)
__device__ __forceinline__ unsigned fq_qual_ctx(unsigned q, unsigned q1, unsigned q2) {
  unsigned ctx = ((((q1 > q2) ? q1 : q2) << 6) + q) & 0xFFFu;
  ctx += (unsigned)(q1 == q2) << 12;
  return ctx;
}

// largest r in [lo, hi] with rec_start[r] <= e
__device__ __forceinline__ unsigned fq_locate(const uint32_t *__restrict__ rec_start, unsigned lo,
                                              unsigned hi, unsigned e) {
  while (lo < hi) {
    const unsigned mid = lo + ((hi - lo + 1) >> 1);
    if (rec_start[mid] <= e) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// ---- (context, symbol) of position p of a record, encode-side definition, in two halves so
// that callers can issue the loads of the NEXT chunk before they consume (and store results
// of) the current one: vmcnt retires in order, a store between two loads would serialise them.
//   sequence (SequenceEncoder::encodeRecord src/fse_sequence.cpp:53-112): context = the four
//     bases in front of p, nearest in bits 7:6, virtual T,C,C,T = 0xD7 before the read
//   quality (QualityEncoder::encodeRecord src/fse_quality.cpp:5-53, L >= 3): context of p is
//     calcContext(Q[p-1], Q[p-2], Q[p-3]) with zeros in front of the read
// The symbol at position p of a read and the up to four (three) symbols in front of it, as the
// raw bytes q .. q + 7 (q .. q + 3) of the line with q = max(p, 4) - 4 (max(p, 3) - 3): ONE
// unaligned load per symbol instead of five (four) byte loads with an address computation each --
// K1 is bound by instruction issue.
struct SymBytes {
  unsigned lo, hi;  // bytes q .. q + 3, q + 4 .. q + 7 (hi: sequence stream only)
};
struct __attribute__((packed)) FqU32x2 { uint32_t a, b; };  // eight bytes at any address
struct __attribute__((packed)) FqU32x1 { uint32_t a; };

template <class M>
__device__ __forceinline__ SymBytes fq_load_sym_bytes(const uint8_t *__restrict__ raw, const fqgpu_rec &rec,
                                                      unsigned p, bool valid) {
  // Unconditional loads (an idle lane has rec = {0, 0, 0}, p = 0 and reads the first bytes of the
  // block; a read's last positions read a few bytes of what follows its line: the block has 64
  // spare bytes behind raw): straight-line code lets the compiler keep several chunks' loads in flight.
  (void)valid;
  constexpr unsigned K = M::STREAM == 0 ? 4u : 3u;
  const uint8_t *s = raw + (M::STREAM == 0 ? rec.seq_off : rec.qual_off) + (p > K ? p - K : 0u);
  SymBytes r;
  if (M::STREAM == 0) {
    const FqU32x2 v = *reinterpret_cast<const FqU32x2 *>(s);
    r.lo = v.a; r.hi = v.b;
  } else {
    r.lo = reinterpret_cast<const FqU32x1 *>(s)->a; r.hi = 0;
  }
  return r;
}

// Context and symbol of position p from its window (fq_load_sym_bytes), without a branch or a
// compare: K1 is bound by instruction issue, and "p >= k ? code(byte k) : virtual" compiled to four
// exec-mask branches plus fifteen compare/select pairs per symbol.
//  sequence: the bytes in front of the symbol are shifted to the top of a word (missing ones
//            become 0), every byte goes through a 256-entry code table in LDS (exactly
//            fq_base_code: 0 for anything but C, G, T -- so missing bytes add nothing), and the
//            virtual bases in front of the read are 0xD7 >> 2 p (src/fse_sequence.h:22-24 applied p times)
//  quality:  33 is subtracted from all four bytes at once (a byte < 33 borrows from its upper
//            neighbour, but then the block is refused anyway: that byte is somebody's symbol >= 64)
//            The SYMBOL goes through a second table (sym_lut = fq_base_sym): 4 for a byte that is
//            no base at all, which the caller turns into FQGPU_E_ARG -- same instruction count.
template <class M>
__device__ __forceinline__ void fq_ctx_from_bytes(const SymBytes &r, unsigned p, unsigned &ctx, unsigned &sym,
                                                  const uint8_t *code_lut, const uint8_t *sym_lut) {
  constexpr unsigned K = M::STREAM == 0 ? 4u : 3u;
  const unsigned pq = min(p, K), sh = 8u * pq;  // the symbol is byte pq of the window
  if (M::STREAM == 0) {
    const unsigned long long w = ((unsigned long long)r.hi << 32) | r.lo;
    sym = sym_lut[(unsigned)(w >> sh) & 0xFFu];
    const unsigned prev = (unsigned)((unsigned long long)r.lo << (32u - sh));  // bytes p-1 | p-2 | p-3 | p-4
    ctx = ((unsigned)code_lut[prev >> 24] << 6) | ((unsigned)code_lut[(prev >> 16) & 0xFFu] << 4) |
          ((unsigned)code_lut[(prev >> 8) & 0xFFu] << 2) | (unsigned)code_lut[prev & 0xFFu] | (0xD7u >> (2u * pq));
  } else {
    (void)code_lut; (void)sym_lut;
    const unsigned w33 = r.lo - 0x21212121u;
    sym = (w33 >> sh) & 0xFFu;
    const unsigned prev = (unsigned)((unsigned long long)w33 << (32u - sh));  // q | q1 | q2 | -
    ctx = fq_qual_ctx((prev >> 24) & 63u, (prev >> 16) & 63u, (prev >> 8) & 63u);
  }
}

#endif  // __HIPCC__
