// Block decoder -- replaces the second pass of DecompressionWorkspace::decodeChunk
// (reference src/workspace.cpp:84-87): SequenceDecoder::decodeRecord
// (src/fse_sequence.cpp:114-143) and QualityDecoder::decodeRecord
// (src/fse_quality.cpp:55-67) with FSE_Decoder::startChunk/endChunk
// (src/fse_common.hpp:130-142).
//
// The format gives a decoder nothing to split a stream on: the bit position of a
// symbol depends on every earlier nbBits and its context on the symbols just decoded
// (SURVEY.md 7.3).  Parallelism = (blocks in the batch) x 2 streams: one workgroup per
// (block, stream), one lane walks the chain, the wave only helps where the work is
// data-parallel (loading the 256 / 8192 initial states, whose bit positions are known
// from the table logs).  All blocks of a batch are decoded by ONE launch.
#include "fqgpu_internal.h"

namespace {

struct DecJob {
  const uint8_t *seq;  unsigned seq_len;
  const uint8_t *qual; unsigned qual_len;
  const fqgpu_rec *recs; unsigned n_recs;
  const uint16_t *n_count;  // the block's own n_recs entries
  const uint16_t *n_pos;  unsigned n_pos_len;
  uint8_t *raw;
  BlockResult *res;
  unsigned rec_base;  // first index of this block in the batch-wide record arrays
};

// Pointers read out of the job descriptors are generic to the compiler, which then emits
// flat_load / flat_store: those also count on lgkmcnt, so every LDS wait of the walking loop
// would wait for the previous output byte to reach memory.  Declare them global.
#define FQ_GLOBAL __attribute__((address_space(1)))
typedef FQ_GLOBAL uint8_t g_u8;
typedef const FQ_GLOBAL uint32_t g_cu32;
typedef const FQ_GLOBAL fqgpu_rec g_crec;

struct TabView {
  const uint32_t *logs, *log_prefix, *dt, *dt_off;
};

// Backward bit reader (BIT_DStream_t, zstd bitstream.h) in functional form: `pos` = number
// of unread bits below the end mark; reading nb bits returns bits [pos-nb, pos) of the
// little-endian bit array, bit pos-1 being the MSB.  A 64-bit register window is refilled
// from two aligned dwords only when the read position leaves it.
struct BitReader {
  g_cu32 *w;           // stream as aligned dwords (buffers are padded)
  long long pos;
  long long wbase;     // bit index of window bit 0 (multiple of 32)
  unsigned long long win;
  __device__ __forceinline__ void refill() {
    long long top = (pos + 31) & ~31ll;
    wbase = top >= 64 ? top - 64 : 0;
    const unsigned wi = (unsigned)(wbase >> 5);
    win = (unsigned long long)w[wi] | ((unsigned long long)w[wi + 1] << 32);
  }
  __device__ __forceinline__ unsigned read(unsigned nb) {
    pos -= nb;
    if (pos < wbase) {
      if (pos < 0) return 0u;  // corrupt stream: caller checks pos at the end
      pos += nb; refill(); pos -= nb;
    }
    return (unsigned)(win >> (unsigned)(pos - wbase)) & ((1u << nb) - 1u);
  }
};

// bits [lo, lo+nb) of the stream, for the data-parallel state load
__device__ __forceinline__ unsigned peek_bits(g_cu32 *w, long long lo, unsigned nb) {
  const unsigned wi = (unsigned)(lo >> 5);
  const unsigned long long v = (unsigned long long)w[wi] | ((unsigned long long)w[wi + 1] << 32);
  return (unsigned)(v >> (unsigned)(lo & 31)) & ((1u << nb) - 1u);
}

template <class M>
__device__ void decode_stream(const DecJob &j, const TabView &tab, uint16_t *state, uint32_t *dt_off) {
  constexpr unsigned B = M::B;
  const uint8_t *src = M::STREAM == 0 ? j.seq : j.qual;
  const unsigned len = M::STREAM == 0 ? j.seq_len : j.qual_len;
  StreamResult *res = &j.res->s[M::STREAM];
  const unsigned lane = threadIdx.x;
  g_cu32 *w = (g_cu32 *)reinterpret_cast<const uint32_t *>(src);
  g_crec *recs = (g_crec *)j.recs;
  g_u8 *raw = (g_u8 *)j.raw;

  // BIT_initDStream: the highest set bit of the last byte is the end mark
  const unsigned last = len ? src[len - 1] : 0u;
  if (last == 0) { if (lane == 0) res->corrupt = 1; return; }
  const long long p0 = (long long)(len - 1) * 8 + (31 - __clz((int)last));
  const unsigned sum_logs = tab.log_prefix[B];
  if (p0 < (long long)sum_logs) { if (lane == 0) res->corrupt = 1; return; }
  // FSE_initDState for ctx B-1 .. 0 (src/fse_common.hpp:134-138): ctx c sits at a fixed
  // offset below the end mark, so all of them load in parallel
  for (unsigned c = lane; c < B; c += 64) {
    const unsigned lg = tab.logs[c];
    const long long lo = p0 - (long long)(sum_logs - tab.log_prefix[c]);
    state[c] = (uint16_t)peek_bits(w, lo, lg);
    dt_off[c] = tab.dt_off[c] + 1u;  // skip the DTable header word
  }
  __syncthreads();
  if (lane != 0) return;

  BitReader br;
  br.w = w;
  br.pos = p0 - (long long)sum_logs;
  br.refill();
  const uint32_t *__restrict__ dt = tab.dt;
  for (unsigned r = j.n_recs; r > 0; r--) {  // records last -> first
    fqgpu_rec rec;
    rec.seq_off = recs[r - 1].seq_off; rec.qual_off = recs[r - 1].qual_off; rec.len = recs[r - 1].len;
    if (M::STREAM == 0) {
      g_u8 *out = raw + rec.seq_off;
      unsigned ctx = 0xD7u;  // FSE_Sequence::INITIAL_CONTEXT
      for (unsigned i = 0; i < rec.len; i++) {
        const uint32_t e = dt[dt_off[ctx] + state[ctx]];
        const unsigned sym = (e >> 16) & 3u;
        state[ctx] = (uint16_t)((e & 0xFFFFu) + br.read(e >> 24));
        out[i] = (uint8_t)(0x54474341u >> (8u * sym));  // "ACGT"[sym]
        ctx = (ctx >> 2) + (sym << 6);                   // addSymUpper
      }
    } else {
      g_u8 *out = raw + rec.qual_off;
      unsigned ctx = 1u << 12, q1 = 0, q2 = 0;  // calcContext(0,0,0)
      for (unsigned i = 0; i < rec.len; i++) {
        const uint32_t e = dt[dt_off[ctx] + state[ctx]];
        const unsigned q = (e >> 16) & 63u;
        state[ctx] = (uint16_t)((e & 0xFFFFu) + br.read(e >> 24));
        out[i] = (uint8_t)(q + 33u);
        ctx = fq_qual_ctx(q, q1, q2);
        q2 = q1;
        q1 = q;
      }
    }
    if (br.pos < 0) break;
  }
  // BIT_endOfDStream (src/fse_common.hpp:141): every bit consumed, none invented
  if (br.pos != 0) res->corrupt = 1;
  res->total_bits = (unsigned long long)(p0 - (long long)sum_logs);
}

// grid = 2 * n_blocks: the quality streams (longer chains, MB-scale DTables) are
// dispatched first, the sequence streams behind them
__global__ void __launch_bounds__(64)
k_decode(const DecJob *__restrict__ jobs, unsigned n_blocks, TabView seq_tab, TabView qual_tab) {
  __shared__ uint16_t state[QualModel::B];
  __shared__ uint32_t dt_off[QualModel::B];
  if (blockIdx.x < n_blocks) decode_stream<QualModel>(jobs[blockIdx.x], qual_tab, state, dt_off);
  else decode_stream<SeqModel>(jobs[blockIdx.x - n_blocks], seq_tab, state, dt_off);
}

// batch-wide record arrays: N counts widened for the scan
__global__ void __launch_bounds__(256)
k_gather_ncount(const DecJob *__restrict__ jobs, uint32_t *__restrict__ cnt32) {
  const DecJob j = jobs[blockIdx.y];
  for (unsigned r = blockIdx.x * blockDim.x + threadIdx.x; r < j.n_recs; r += gridDim.x * blockDim.x)
    cnt32[j.rec_base + r] = j.n_count[r];
}

// N restoration (tail of SequenceDecoder::decodeRecord, src/fse_sequence.cpp:138-142).
// The reference pops counts and deltas from the END of n_count / n_pos while walking the
// records backwards, which equals forward indexing from (n_pos_len - total N of the block).
__global__ void __launch_bounds__(256)
k_npatch(const DecJob *__restrict__ jobs, const uint32_t *__restrict__ off) {
  const DecJob j = jobs[blockIdx.y];
  const unsigned first = off[j.rec_base], total = off[j.rec_base + j.n_recs] - first;
  if (total > j.n_pos_len) {
    if (blockIdx.x == 0 && threadIdx.x == 0) j.res->s[0].corrupt = 1;
    return;
  }
  const unsigned shift = j.n_pos_len - total;
  for (unsigned r = blockIdx.x * blockDim.x + threadIdx.x; r < j.n_recs; r += gridDim.x * blockDim.x) {
    const unsigned cnt = j.n_count[r];
    if (!cnt) continue;
    const fqgpu_rec rec = j.recs[r];
    const uint16_t *d = j.n_pos + shift + (off[j.rec_base + r] - first);
    unsigned at = 0;
    for (unsigned k = 0; k < cnt; k++) {
      at += d[k];
      if (at >= rec.len) { j.res->s[0].corrupt = 1; break; }
      j.raw[rec.seq_off + at] = 'N';
    }
  }
}

// overwrites every sequence / quality byte of the block (decode target, tests, bench)
__global__ void __launch_bounds__(256)
k_wipe(uint8_t *__restrict__ raw, const fqgpu_rec *__restrict__ recs, unsigned R) {
  const unsigned waves = (gridDim.x * blockDim.x) >> 6;
  const unsigned lane = fq_lane();
  for (unsigned r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < R; r += waves) {
    const fqgpu_rec rec = recs[r];
    for (unsigned i = lane; i < rec.len; i += 64) {
      raw[rec.seq_off + i] = '?';
      raw[rec.qual_off + i] = '?';
    }
  }
}

}  // namespace

int fq_wipe_launch(fqgpu_ctx *ctx, fqgpu_dblock *b) {
  const unsigned R = (unsigned)b->n_recs;
  int rcs = fqgpu_sync(ctx);  // the block may still be in an encode lane
  if (rcs) return rcs;
  if (!R) return FQGPU_OK;
  const unsigned blocks = (unsigned)min((size_t)(R + 3) / 4, (size_t)8192);
  hipLaunchKernelGGL(k_wipe, dim3(blocks), dim3(256), 0, ctx->stream, b->raw, b->recs, R);
  FQ_HIP(hipGetLastError());
  return FQGPU_OK;
}

int fq_decode_launch(fqgpu_ctx *ctx, fqgpu_dblock *const *blocks, size_t n_blocks) {
  hipStream_t st = ctx->stream;
  if (!n_blocks) return FQGPU_OK;
  int rcs = fqgpu_sync(ctx);  // blocks may still be in an encode lane
  if (rcs) return rcs;
  // job descriptors: pinned host staging is not worth it for a few KB; plain async copy
  // from a host vector that outlives the copy (we synchronise the copy right away)
  DecJob *host = new DecJob[n_blocks];
  size_t r_tot = 0, r_max = 0;
  for (size_t i = 0; i < n_blocks; i++) {
    const fqgpu_dblock *b = blocks[i];
    DecJob &j = host[i];
    j.seq = b->seq;   j.seq_len = (unsigned)b->seq_len;
    j.qual = b->qual; j.qual_len = (unsigned)b->qual_len;
    j.recs = b->recs; j.n_recs = (unsigned)b->n_recs;
    j.n_count = b->n_count;
    j.n_pos = b->n_pos; j.n_pos_len = (unsigned)b->n_pos_len;
    j.raw = b->raw;
    j.res = b->result;
    j.rec_base = (unsigned)r_tot;
    r_tot += b->n_recs;
    if (b->n_recs > r_max) r_max = b->n_recs;
  }
  int rc;
  if ((rc = ctx->dec_desc.reserve(n_blocks * sizeof(DecJob)))) { delete[] host; return rc; }
  if ((rc = ctx->n_cnt32.reserve((r_tot + 1) * 4))) { delete[] host; return rc; }
  if ((rc = ctx->n_off.reserve((r_tot + 1) * 4))) { delete[] host; return rc; }
  hipError_t he = hipMemcpyAsync(ctx->dec_desc.p, host, n_blocks * sizeof(DecJob), hipMemcpyHostToDevice, st);
  if (he == hipSuccess) he = hipStreamSynchronize(st);
  delete[] host;
  if (he != hipSuccess) return fq_hip_error(he, __FILE__, __LINE__);
  const DecJob *jobs = ctx->dec_desc.as<DecJob>();

  for (size_t i = 0; i < n_blocks; i++)
    FQ_HIP(hipMemsetAsync(blocks[i]->result, 0, sizeof(BlockResult), st));
  TabView ts = {ctx->tab[0].logs, ctx->tab[0].log_prefix, ctx->tab[0].dt, ctx->tab[0].dt_off};
  TabView tq = {ctx->tab[1].logs, ctx->tab[1].log_prefix, ctx->tab[1].dt, ctx->tab[1].dt_off};
  fq_timer_span_begin(ctx, "decode", st);
  hipLaunchKernelGGL(k_decode, dim3((unsigned)(2 * n_blocks)), dim3(64), 0, st, jobs, (unsigned)n_blocks, ts, tq);
  fq_timer_span_end(ctx, st);
  fq_timer_span_begin(ctx, "npatch", st);
  const unsigned gx = (unsigned)min((r_max + 255) / 256, (size_t)4096);
  hipLaunchKernelGGL(k_gather_ncount, dim3(gx ? gx : 1, (unsigned)n_blocks), dim3(256), 0, st, jobs,
                     ctx->n_cnt32.as<uint32_t>());
  if ((rc = fq_scan_u32_to_u32(st, ctx->n_cnt32.as<uint32_t>(), r_tot, ctx->n_off.as<uint32_t>(),
                               ctx->scan_tmp)))
    return rc;
  hipLaunchKernelGGL(k_npatch, dim3(gx ? gx : 1, (unsigned)n_blocks), dim3(256), 0, st, jobs,
                     ctx->n_off.as<uint32_t>());
  fq_timer_span_end(ctx, st);
  FQ_HIP(hipGetLastError());
  return FQGPU_OK;
}
