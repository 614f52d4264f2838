// C ABI of libfqgpu.so (include/fqgpu.h): handle and buffer management around the
// kernels of tables.hip / encode.hip / decode.hip.  Host code only.
#include "fqgpu_internal.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <map>
#include <mutex>
#include <vector>

// The encode lanes use up to 16 streams (two per block in flight: sequence and quality pipelines).  ROCm maps streams onto
// GPU_MAX_HW_QUEUES hardware queues (default 4) and kernels of streams that share a queue
// run back to back; ask for 16 before the runtime initialises (one per stream of 8 lanes;
// 24 helps blocks of 16 MiB and less by a fifth and costs 256 MiB blocks 2 %).  This touches the
// host process's environment, so it is narrow and can be switched off: it never overrides a value
// the application (or the user) set, it has no effect once HIP is initialised, and
// FQGPU_KEEP_HW_QUEUES=1 in the environment makes the library leave the variable alone
// ("Environment").
__attribute__((constructor)) static void fq_ask_for_hw_queues() {
  if (getenv("FQGPU_KEEP_HW_QUEUES")) return;
  setenv("GPU_MAX_HW_QUEUES", "16", 0);
}

// ------------------------------------------------------------------ errors
static thread_local char g_hip_msg[256] = "";

int fq_hip_error(hipError_t e, const char *file, int line) {
  snprintf(g_hip_msg, sizeof(g_hip_msg), "HIP error %d (%s) at %s:%d", (int)e, hipGetErrorString(e), file, line);
  if (getenv("FQGPU_VERBOSE")) fprintf(stderr, "fqgpu: %s\n", g_hip_msg);
  return (e == hipErrorOutOfMemory) ? FQGPU_E_NOMEM
         : (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? FQGPU_E_NO_DEVICE
                                                                 : FQGPU_E_HIP;
}

extern "C" const char *fqgpu_strerror(int code) {
  switch (code) {
  case FQGPU_OK: return "ok";
  case FQGPU_E_OVERFLOW: return "compressed stream exceeds the reference capacity bound";
  case FQGPU_E_SHORT_READ: return "read shorter than 3 bases (undefined in the reference coder)";
  case FQGPU_E_CORRUPT: return "corrupt stream (end mark / leftover bits / N table)";
  case FQGPU_E_ARG: return "bad argument or symbol outside the model alphabet";
  case FQGPU_E_NO_DEVICE: return "no usable MI355X / HIP device (there is no CPU fallback)";
  case FQGPU_E_NOMEM: return "out of device or host memory";
  case FQGPU_E_HIP: return g_hip_msg[0] ? g_hip_msg : "HIP runtime error";
  case FQGPU_E_HEADER: return "read header the field coder cannot code (numeric field not an int32 / string field of 255 or more bytes)";
  default: return "unknown fqgpu error";
  }
}

extern "C" const char *fqgpu_version(void) { return "fqgpu 0.1 (gfx950)"; }

extern "C" int fqgpu_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// Workspace::compressBoundSequence / compressBoundQuality (reference src/workspace.h:21-35)
extern "C" size_t fqgpu_bound_seq(size_t n) {
  if (n < 1024) return (size_t)1024 * FQGPU_SEQ_MODELS;
  return n / 4 + 1024;
}
extern "C" size_t fqgpu_bound_qual(size_t n) {
  const size_t a = (size_t)1024 * FQGPU_QUAL_MODELS, b = n * 7 / 8 + 1024;
  return a > b ? a : b;
}

static int use_device(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return FQGPU_E_NO_DEVICE;
  if (device < 0 || device >= n) return FQGPU_E_NO_DEVICE;
  FQ_HIP(hipSetDevice(device));
  return FQGPU_OK;
}

// ------------------------------------------------------------------ pinned host memory
// A 64-byte header in front of the block says how it was allocated (pinned, or plain heap when
// there is no usable device: host-only tools and tests) and how large it is.
//
// hipHostFree waits until the DEVICE is idle (measured: 1.25 s behind a 1.6 s decode kernel of
// another thread, 6 ms on an idle device): a worker that lets a buffer grow stalls behind every
// other worker's kernels, and four decoding workers of the block farm ran two at a time because of
// it.  So freed pinned blocks go back to a cache by size class (sizes are rounded up to 1/8 of their
// power of two: at most 12.5 % more) and are handed out again; only what exceeds the cache limit
// (FQGPU_PINNED_CACHE_MB, default 8192) is really freed.  Blocks below 64 KiB are not pinned at all:
// they never carry a stream, and a farm makes thousands of them.
namespace {
struct HostHdr { uint64_t magic, pinned, size, pad[5]; };  // size: bytes of the whole block, header included
constexpr uint64_t HOST_MAGIC = 0x46514850494E4E44ull;
constexpr size_t PIN_MIN = 64 << 10;
struct PinCache {
  std::mutex mutex;
  std::multimap<size_t, void *> blocks;  // block size -> free pinned block
  size_t bytes = 0;
};
// (never destroyed: fqgpu_host_free is called from the destructors of the callers' own statics at exit)
PinCache &pin_cache() {
  static PinCache *c = new PinCache;
  return *c;
}
size_t pin_class(size_t bytes) {  // rounded up to a multiple of 1/8 of the largest power of two below it
  size_t step = 1;
  while ((step << 4) <= bytes) step <<= 1;
  return (bytes + step - 1) & ~(step - 1);
}
size_t pin_cache_limit() {
  static const size_t limit = [] {
    const char *e = getenv("FQGPU_PINNED_CACHE_MB");
    return (size_t)(e ? strtoull(e, nullptr, 10) : 8192ull) << 20;
  }();
  return limit;
}
}  // namespace
extern "C" void *fqgpu_host_alloc(size_t bytes) {
  void *p = nullptr;
  size_t total = bytes + sizeof(HostHdr);
  bool pinned = false;
  if (total >= PIN_MIN) {
    total = pin_class(total);
    {
      PinCache &pc = pin_cache();
      std::lock_guard<std::mutex> lock(pc.mutex);
      auto it = pc.blocks.find(total);
      if (it != pc.blocks.end()) { p = it->second; pc.blocks.erase(it); pc.bytes -= total; pinned = true; }
    }
    if (!p) {
      int n = 0;
      const bool gpu = hipGetDeviceCount(&n) == hipSuccess && n > 0;
      // Portable: the cache is process-wide, a block pinned while one device was current may be handed to a worker of another
      pinned = gpu && hipHostMalloc(&p, total, hipHostMallocPortable) == hipSuccess;
      if (!pinned) { (void)hipGetLastError(); p = nullptr; }
    }
  }
  if (!p) {
    total = (total + 63) & ~(size_t)63;
    p = aligned_alloc(64, total);
    if (!p) return nullptr;
  }
  HostHdr *h = static_cast<HostHdr *>(p);
  h->magic = HOST_MAGIC;
  h->pinned = pinned ? 1 : 0;
  h->size = total;
  return h + 1;
}
extern "C" void fqgpu_host_free(void *p) {
  if (!p) return;
  HostHdr *h = static_cast<HostHdr *>(p) - 1;
  if (h->magic != HOST_MAGIC) return;  // not ours
  h->magic = 0;
  if (!h->pinned) { free(h); return; }
  const size_t total = h->size;
  {
    PinCache &pc = pin_cache();
    std::lock_guard<std::mutex> lock(pc.mutex);
    if (pc.bytes + total <= pin_cache_limit()) {
      pc.blocks.emplace(total, h);
      pc.bytes += total;
      return;
    }
  }
  (void)hipHostFree(h);  // (waits for the device)
}
// gives the cached pinned blocks back to the system (waits for the device); returns the bytes freed
extern "C" size_t fqgpu_host_trim(void) {
  std::multimap<size_t, void *> drop;
  size_t bytes;
  {
    PinCache &pc = pin_cache();
    std::lock_guard<std::mutex> lock(pc.mutex);
    drop.swap(pc.blocks);
    bytes = pc.bytes;
    pc.bytes = 0;
  }
  for (auto &kv : drop) (void)hipHostFree(kv.second);
  return bytes;
}

// ------------------------------------------------------------------ kernel timing
// Spans = pairs of HIP events recorded on the stream the kernels run on.  Spans accumulate
// from fqgpu_ctx_enable_timing(ctx, 1) until they are read; fqgpu_ctx_last_timing sums them
// per label (total device time and number of launches of every kernel group).
struct KernelTimer {
  bool on = false;
  std::string only;  // if not empty: only spans of this label get events (fqgpu_ctx_timing_only)
  static constexpr size_t SKIPPED = (size_t)-2;
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  struct Span { const char *name; size_t b, e; };
  std::vector<Span> spans;
  hipEvent_t get(size_t *idx) {
    if (used == pool.size()) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return nullptr;
      pool.push_back(e);
    }
    *idx = used;
    return pool[used++];
  }
};

void fq_timer_span_begin(fqgpu_ctx *ctx, const char *name, hipStream_t st) {
  KernelTimer *t = ctx->timer;
  if (!t || !t->on) return;
  if (!t->only.empty() && t->only != name) {  // an open span without events, closed by its span_end
    t->spans.push_back({name, KernelTimer::SKIPPED, (size_t)-1});
    return;
  }
  size_t i;
  hipEvent_t e = t->get(&i);
  if (!e) return;
  (void)hipEventRecord(e, st);
  t->spans.push_back({name, i, (size_t)-1});
}
void fq_timer_span_end(fqgpu_ctx *ctx, hipStream_t st) {
  KernelTimer *t = ctx->timer;
  if (!t || !t->on || t->spans.empty()) return;
  // the matching begin is the last open span (launch code nests nothing across streams)
  for (size_t k = t->spans.size(); k-- > 0;) {
    if (t->spans[k].e == (size_t)-1) {
      if (t->spans[k].b == KernelTimer::SKIPPED) { t->spans.erase(t->spans.begin() + (long)k); return; }
      size_t i;
      hipEvent_t e = t->get(&i);
      if (!e) return;
      (void)hipEventRecord(e, st);
      t->spans[k].e = i;
      return;
    }
  }
}

extern "C" int fqgpu_ctx_enable_timing(fqgpu_ctx *ctx, int on) {
  if (!ctx) return FQGPU_E_ARG;
  if (!ctx->timer) ctx->timer = new (std::nothrow) KernelTimer();
  if (!ctx->timer) return FQGPU_E_NOMEM;
  ctx->timer->on = on != 0;
  if (on) { ctx->timer->used = 0; ctx->timer->spans.clear(); }
  return FQGPU_OK;
}

extern "C" int fqgpu_ctx_timing_only(fqgpu_ctx *ctx, const char *name) {
  if (!ctx) return FQGPU_E_ARG;
  if (!ctx->timer) ctx->timer = new (std::nothrow) KernelTimer();
  if (!ctx->timer) return FQGPU_E_NOMEM;
  ctx->timer->only = name ? name : "";
  return FQGPU_OK;
}

extern "C" int fqgpu_ctx_last_timing(fqgpu_ctx *ctx, fqgpu_timing *out) {
  if (!ctx || !out || !ctx->timer) return FQGPU_E_ARG;
  KernelTimer *t = ctx->timer;
  memset(out, 0, sizeof(*out));
  int rc = fqgpu_sync(ctx);
  if (rc) return rc;
  float first_to_last = 0;
  for (const KernelTimer::Span &sp : t->spans) {
    if (sp.e == (size_t)-1) continue;
    float ms = 0, span_end = 0;
    FQ_HIP(hipEventElapsedTime(&ms, t->pool[sp.b], t->pool[sp.e]));
    FQ_HIP(hipEventElapsedTime(&span_end, t->pool[t->spans[0].b], t->pool[sp.e]));
    if (span_end > first_to_last) first_to_last = span_end;
    int k = 0;
    for (; k < out->n_kernels; k++) if (strcmp(out->kernel_name[k], sp.name) == 0) break;
    if (k == out->n_kernels) {
      if (k == 32) continue;
      out->kernel_name[k] = sp.name;
      out->n_kernels++;
    }
    out->kernel_ms[k] += ms;
    out->kernel_calls[k] += 1;
  }
  out->total_ms = first_to_last;
  return FQGPU_OK;
}

// ------------------------------------------------------------------ encode lanes
// Blocks take the lanes in turn -- unless the handle keeps coding the SAME few blocks (no more of them than lanes): then
// a block goes back to the lane of its last encode.  Its encodes are ordered anyway (same streams, same result words: ev_encoded),
// and on its own lane that order costs nothing, where the rotation sends the next encode to another lane that must also finish
// ANOTHER block's encode first: the chains of unrelated blocks get coupled and every step ends in a partial barrier -- four blocks
// coded over and over on six lanes ran at 56-58 GB/s against 69 on their own four.  With more blocks than lanes the rotation
// stays: a fixed assignment (sixteen blocks: 3 3 3 3 2 2) leaves two lanes idle at the end of a batch.
// `b` is only compared with the blocks coded last, never followed (it may be gone); nullptr: reserving, lanes in turn.
EncLane *fq_next_lane(fqgpu_ctx *ctx, size_t n_bases, fqgpu_dblock *b) {
  const unsigned n = fq_lanes_for(ctx, n_bases);
  unsigned pick = n;
  if (b) {
    unsigned distinct = 0;
    bool seen = false;
    for (unsigned i = 0; i < FQ_RECENT_BLOCKS; i++) {
      const void *p = ctx->recent[i];
      if (!p) continue;
      bool first = true;
      for (unsigned j = 0; j < i; j++) first = first && ctx->recent[j] != p;
      distinct += first;
      seen = seen || p == b;
    }
    ctx->recent[ctx->recent_at++ % FQ_RECENT_BLOCKS] = b;
    if (seen && distinct <= n && b->home_lane >= 0 && (unsigned)b->home_lane < n) pick = (unsigned)b->home_lane;
  }
  if (pick == n) {
    pick = ctx->next_lane % n;
    ctx->next_lane++;
  }
  if (b) b->home_lane = (int)pick;
  EncLane &l = ctx->lanes[pick];
  if (!l.st_seq) {
    // Both pipelines at the same priority: while the sequence chains were serial their stream
    // ran at high priority; with every kernel a throughput kernel that costs 2 % (measured).
    if (hipStreamCreateWithFlags(&l.st_seq, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&l.st_qual, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&l.ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&l.ev_join, hipEventDisableTiming) != hipSuccess)
      return nullptr;
  }
  return &l;
}

static void free_lane(EncLane &l) {
  DevBuf *bufs[] = {&l.rec_start, &l.n_cnt32, &l.n_off, &l.scan_tmp, &l.first_sym, &l.rscan};
  for (DevBuf *b : bufs) b->release();
  for (int s = 0; s < 2; s++) {
    EncScratch &e = l.enc[s];
    DevBuf *eb[] = {&e.slot_of, &e.keys, &e.sorted_sym, &e.out16, &e.tile_hist, &e.tile_base, &e.group_sum,
                    &e.ctx_arrays, &e.seg_state, &e.seg_arrays, &e.seq_bdesc, &e.seq_plan, &e.seq_fbuf, &e.seq_cbuf, &e.tile_bits, &e.tile_bit_base, &e.scan_tmp, &e.tile_runs, &e.tile_sync, &e.dbg_enc16};
    for (DevBuf *b : eb) b->release();
  }
  hipEvent_t evs[] = {l.ev_fork, l.ev_join};
  for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
  hipStream_t sts[] = {l.st_seq, l.st_qual};
  for (hipStream_t q : sts) if (q) (void)hipStreamDestroy(q);
  l = EncLane();
}

// ------------------------------------------------------------------ record validation (host)
static int check_recs(const fqgpu_rec *recs, size_t n_recs, size_t raw_len, size_t *n_bases) {
  size_t tot = 0;
  for (size_t i = 0; i < n_recs; i++) {
    const fqgpu_rec &r = recs[i];
    if (r.len > 65535u) return FQGPU_E_ARG;  // readlen_t is u16 (src/defs.h:14)
    if ((size_t)r.seq_off + r.len > raw_len || (size_t)r.qual_off + r.len > raw_len) return FQGPU_E_ARG;
    if (r.len < 3) return FQGPU_E_SHORT_READ;
    tot += r.len;
  }
  if (tot >= 0xFFF00000ull) return FQGPU_E_ARG;  // block sizes are u32 in the reference too
  *n_bases = tot;
  return FQGPU_OK;
}

// ------------------------------------------------------------------ frequency tables
struct FtLayout {
  int n_models, alpha;
  size_t norm_bytes, logs_off, maxlog_off, total;
};
static FtLayout ft_layout(int stream) {
  FtLayout l;
  l.n_models = stream ? FQGPU_QUAL_MODELS : FQGPU_SEQ_MODELS;
  l.alpha = stream ? FQGPU_QUAL_ALPHA : FQGPU_SEQ_ALPHA;
  l.norm_bytes = (size_t)l.n_models * l.alpha * 2;
  l.logs_off = l.norm_bytes;
  l.maxlog_off = l.logs_off + (size_t)l.n_models * 4;
  l.total = l.maxlog_off + 4;
  return l;
}

// counts (device) -> FreqTable POD (host)
static int normalize_to_host(hipStream_t st, const uint32_t *counts_dev, int stream, void *ft_out) {
  const FtLayout l = ft_layout(stream);
  uint8_t *dev = fq_dev_alloc<uint8_t>(l.total + 8);
  if (!dev) return FQGPU_E_NOMEM;
  int rc = FQGPU_OK;
  uint32_t err = 0;
  do {
    if (hipMemsetAsync(dev, 0, l.total + 8, st) != hipSuccess) { rc = FQGPU_E_HIP; break; }
    uint32_t *err_dev = reinterpret_cast<uint32_t *>(dev + ((l.total + 3) & ~(size_t)3));
    rc = fq_normalize_counts(st, counts_dev, l.n_models, l.alpha, reinterpret_cast<int16_t *>(dev),
                             reinterpret_cast<uint32_t *>(dev + l.logs_off),
                             reinterpret_cast<uint32_t *>(dev + l.maxlog_off), err_dev);
    if (rc) break;
    if (hipMemcpyAsync(ft_out, dev, l.total, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(&err, err_dev, 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) { rc = FQGPU_E_HIP; break; }
    if (err) rc = FQGPU_E_ARG;
  } while (0);
  (void)hipFree(dev);
  return rc;
}

extern "C" int fqgpu_tables_from_counts(int device, const uint32_t *seq_counts, const uint32_t *qual_counts,
                                        void *seq_ft_out, void *qual_ft_out) {
  int rc = use_device(device);
  if (rc) return rc;
  if (!seq_counts || !qual_counts || !seq_ft_out || !qual_ft_out) return FQGPU_E_ARG;
  const size_t ns = (size_t)FQGPU_SEQ_MODELS * FQGPU_SEQ_ALPHA, nq = (size_t)FQGPU_QUAL_MODELS * FQGPU_QUAL_ALPHA;
  uint32_t *dev = fq_dev_alloc<uint32_t>(ns + nq);
  if (!dev) return FQGPU_E_NOMEM;
  hipStream_t st = nullptr;
  if (hipMemcpy(dev, seq_counts, ns * 4, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(dev + ns, qual_counts, nq * 4, hipMemcpyHostToDevice) != hipSuccess) rc = FQGPU_E_HIP;
  if (!rc) rc = normalize_to_host(st, dev, 0, seq_ft_out);
  if (!rc) rc = normalize_to_host(st, dev + ns, 1, qual_ft_out);
  (void)hipFree(dev);
  return rc;
}

extern "C" int fqgpu_freq_tables(int device, const uint8_t *raw, size_t raw_len, const fqgpu_rec *recs,
                                 size_t n_recs, void *seq_ft_out, void *qual_ft_out,
                                 uint32_t *seq_counts_out, uint32_t *qual_counts_out) {
  int rc = use_device(device);
  if (rc) return rc;
  if (!raw || !recs || !seq_ft_out || !qual_ft_out) return FQGPU_E_ARG;
  // the histogram pass itself accepts reads of any length (src/fse_sequence.cpp:145-169)
  size_t n_bases = 0;
  unsigned min_len = 0xFFFFFFFFu;
  for (size_t i = 0; i < n_recs; i++) {
    if ((size_t)recs[i].seq_off + recs[i].len > raw_len || (size_t)recs[i].qual_off + recs[i].len > raw_len)
      return FQGPU_E_ARG;
    n_bases += recs[i].len;
    if (recs[i].len < min_len || recs[i].len > 65535u) min_len = recs[i].len > 65535u ? 0u : recs[i].len;
  }
  const size_t ns = (size_t)FQGPU_SEQ_MODELS * FQGPU_SEQ_ALPHA, nq = (size_t)FQGPU_QUAL_MODELS * FQGPU_QUAL_ALPHA;
  uint8_t *raw_dev = fq_dev_alloc<uint8_t>(raw_len + 64);
  fqgpu_rec *recs_dev = fq_dev_alloc<fqgpu_rec>(n_recs + 1);
  uint32_t *cnt = fq_dev_alloc<uint32_t>(ns + nq + 1);
  hipStream_t st = nullptr;
  uint32_t err = 0;
  if (!raw_dev || !recs_dev || !cnt) rc = FQGPU_E_NOMEM;
  do {
    if (rc) break;
    if (hipMemcpy(raw_dev, raw, raw_len, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(recs_dev, recs, n_recs * sizeof(fqgpu_rec), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(cnt + ns + nq, 0, 4) != hipSuccess) { rc = FQGPU_E_HIP; break; }
    if ((rc = fq_build_freq_tables(device, st, raw_dev, recs_dev, n_recs, cnt, cnt + ns, n_bases, min_len))) break;
    if (hipMemcpy(&err, cnt + ns + nq, 4, hipMemcpyDeviceToHost) != hipSuccess) { rc = FQGPU_E_HIP; break; }
    if (err) { rc = FQGPU_E_ARG; break; }  // quality above Q63: the reference throws (src/fse_quality.cpp:88); a base byte outside ACGTN
    if (seq_counts_out && hipMemcpy(seq_counts_out, cnt, ns * 4, hipMemcpyDeviceToHost) != hipSuccess) { rc = FQGPU_E_HIP; break; }
    if (qual_counts_out && hipMemcpy(qual_counts_out, cnt + ns, nq * 4, hipMemcpyDeviceToHost) != hipSuccess) { rc = FQGPU_E_HIP; break; }
    if ((rc = normalize_to_host(st, cnt, 0, seq_ft_out))) break;
    if ((rc = normalize_to_host(st, cnt + ns, 1, qual_ft_out))) break;
  } while (0);
  if (raw_dev) (void)hipFree(raw_dev);
  if (recs_dev) (void)hipFree(recs_dev);
  if (cnt) (void)hipFree(cnt);
  return rc;
}

// ------------------------------------------------------------------ handle
static void free_tables(DevTables &t) {
  void *ps[] = {t.norm, t.logs, t.log_prefix, t.ct, t.ct_off, t.dt, t.dt_off, t.next1, t.next2, t.reset_mask,
                t.seq_pow[0], t.seq_pow[1], t.seq_pow[2], t.seq_pow[3]};
  for (void *p : ps) if (p) (void)hipFree(p);
  t = DevTables();
}

static int upload_tables(fqgpu_ctx *ctx, int stream, const void *ft) {
  const FtLayout l = ft_layout(stream);
  DevTables &t = ctx->tab[stream];
  const uint8_t *p = static_cast<const uint8_t *>(ft);
  const uint32_t *logs = reinterpret_cast<const uint32_t *>(p + l.logs_off);
  uint32_t max_log = 0;
  for (int i = 0; i < l.n_models; i++) {
    if (logs[i] < 5 || logs[i] > 12) return FQGPU_E_ARG;  // FSE_MIN_TABLELOG .. FSE_MAX_TABLELOG
    if (logs[i] > max_log) max_log = logs[i];
  }
  t.max_log = max_log;
  t.norm = fq_dev_alloc<int16_t>((size_t)l.n_models * l.alpha);
  t.logs = fq_dev_alloc<uint32_t>(l.n_models);
  uint32_t *err_dev = fq_dev_alloc<uint32_t>(1);
  if (!t.norm || !t.logs || !err_dev) return FQGPU_E_NOMEM;
  FQ_HIP(hipMemcpyAsync(t.norm, p, l.norm_bytes, hipMemcpyHostToDevice, ctx->stream));
  FQ_HIP(hipMemcpyAsync(t.logs, logs, (size_t)l.n_models * 4, hipMemcpyHostToDevice, ctx->stream));
  FQ_HIP(hipMemsetAsync(err_dev, 0, 4, ctx->stream));
  int rc = fq_build_tables(ctx->stream, t, l.n_models, l.alpha, err_dev);
  uint32_t err = 0;
  if (!rc) {
    FQ_HIP(hipMemcpyAsync(&err, err_dev, 4, hipMemcpyDeviceToHost, ctx->stream));
    FQ_HIP(hipStreamSynchronize(ctx->stream));
    if (err) rc = FQGPU_E_ARG;  // counts of a context do not sum to 2^log
  }
  (void)hipFree(err_dev);
  return rc;
}

extern "C" int fqgpu_ctx_create(int device, const void *seq_ft, const void *qual_ft, fqgpu_ctx **out) {
  if (!out) return FQGPU_E_ARG;
  *out = nullptr;
  int rc = use_device(device);
  if (rc) return rc;
  if (!seq_ft || !qual_ft) return FQGPU_E_ARG;
  fqgpu_ctx *ctx = new (std::nothrow) fqgpu_ctx();
  if (!ctx) return FQGPU_E_NOMEM;
  ctx->device = device;
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) ctx->n_cus = (unsigned)cus;
  }
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return FQGPU_E_HIP;
  }
  rc = fq_probe_lds_atomic_order(ctx->stream, &ctx->lds_atomics_ordered);
  if (getenv("FQGPU_NO_LDS_ATOMIC_RANK")) ctx->lds_atomics_ordered = false;  // force the ballot kernel
  if (getenv("FQGPU_SLOT_PARTITION")) ctx->tile_sorted = false;  // the slot-based partition/gather (same bits, for comparisons)
  if (const char *e = getenv("FQGPU_SETFUNC_WGS")) ctx->setfunc_wgs = (unsigned)atoi(e);  // experiments: same bits
  if (const char *e = getenv("FQGPU_SEQ_GROUP")) ctx->seq_group = (unsigned)atoi(e);  // experiments
  if (const char *e = getenv("FQGPU_SEQ_GROUP_MIN")) ctx->seq_group_min = (unsigned)atoi(e);
  if (!rc) rc = upload_tables(ctx, 0, seq_ft);
  if (!rc) rc = upload_tables(ctx, 1, qual_ft);
  if (rc) { fqgpu_ctx_destroy(ctx); return rc; }
  *out = ctx;
  return FQGPU_OK;
}

#ifdef FQGPU_EXPERIMENTS
void fq_ts_prof_dump();
#endif

extern "C" void fqgpu_ctx_destroy(fqgpu_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)fqgpu_sync(ctx);
#ifdef FQGPU_EXPERIMENTS
  fq_ts_prof_dump();
#endif
  free_tables(ctx->tab[0]);
  free_tables(ctx->tab[1]);
  DevBuf *bufs[] = {&ctx->n_cnt32, &ctx->n_off, &ctx->scan_tmp, &ctx->dec_desc, &ctx->dec_chunks, &ctx->dec_recstart};
  for (DevBuf *b : bufs) b->release();
  ctx->hp_parse.release();  // the device parser's scratch (fqgpu_ctx_reserve / fqgpu_encode_begin without a record table)
  ctx->hp_hdr.release();
  for (int i = 0; i < FQ_MAX_LANES; i++) free_lane(ctx->lanes[i]);
  if (ctx->hp_block) fqgpu_dblock_destroy(ctx->hp_block);
  if (ctx->hp_ev_h2d) (void)hipEventDestroy(ctx->hp_ev_h2d);
  if (ctx->hp_result) (void)hipHostFree(ctx->hp_result);
  if (ctx->timer) {
    for (hipEvent_t e : ctx->timer->pool) (void)hipEventDestroy(e);
    delete ctx->timer;
  }
  if (ctx->dec_stream2) (void)hipStreamDestroy(ctx->dec_stream2);
  if (ctx->dec_fork) (void)hipEventDestroy(ctx->dec_fork);
  if (ctx->dec_join) (void)hipEventDestroy(ctx->dec_join);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" int fqgpu_ctx_set_chain_params(fqgpu_ctx *ctx, unsigned segment, unsigned flags) {
  if (!ctx) return FQGPU_E_ARG;
  if (segment) {
    if (segment < 2 || segment > (1u << 24)) return FQGPU_E_ARG;
    ctx->seg_len = segment;
  }
  ctx->seq_generic = (flags & FQGPU_CHAIN_SEQ_GENERIC) ? 1 : 0;
  return FQGPU_OK;
}

extern "C" int fqgpu_ctx_set_seq_group(fqgpu_ctx *ctx, unsigned max_segments, unsigned min_groups) {
  if (!ctx) return FQGPU_E_ARG;
  ctx->seq_group = max_segments ? max_segments : 8u;
  ctx->seq_group_min = min_groups ? min_groups : 16u;
  return FQGPU_OK;
}

extern "C" int fqgpu_ctx_set_seq_segment(fqgpu_ctx *ctx, unsigned symbols) {
  if (!ctx) return FQGPU_E_ARG;
  ctx->seq_segment = symbols;
  if (symbols) {  // the power tables for this segment length (encode.hip rounds it the same way), before any encode uses them
    int rc = use_device(ctx->device);
    if (!rc) rc = fqgpu_sync(ctx);
    if (rc) return rc;
    const unsigned S = (unsigned)std::min((((size_t)symbols + 1023) / 1024) * 1024, (size_t)1 << 30);
    if ((rc = fq_seq_pow_ensure(ctx->stream, ctx->tab[0], FQGPU_SEQ_MODELS, S))) return rc;
    FQ_HIP(hipStreamSynchronize(ctx->stream));
  }
  return FQGPU_OK;
}

extern "C" int fqgpu_ctx_set_index_stride(fqgpu_ctx *ctx, unsigned symbols) {
  if (!ctx || symbols == 0 || symbols > (1u << 30)) return FQGPU_E_ARG;
  ctx->index_stride = (symbols + 65535u) & ~65535u;  // whole partition tiles of both streams
  return FQGPU_OK;
}

extern "C" int fqgpu_dblock_index_bytes(const fqgpu_dblock *b, int stream, size_t *bytes) {
  if (!b || !bytes || stream < 0 || stream > 1) return FQGPU_E_ARG;
  *bytes = b->index_bytes[stream];
  return FQGPU_OK;
}

extern "C" int fqgpu_dblock_fetch_index(fqgpu_ctx *ctx, const fqgpu_dblock *b, int stream, void *out, size_t cap) {
  if (!ctx || !b || !out || stream < 0 || stream > 1 || cap < b->index_bytes[stream]) return FQGPU_E_ARG;
  int rc = fqgpu_sync(ctx);
  if (rc) return rc;
  if (b->index_bytes[stream]) FQ_HIP(hipMemcpy(out, b->index[stream], b->index_bytes[stream], hipMemcpyDeviceToHost));
  return FQGPU_OK;
}

// an index for this block's stream: its header must describe the block; room for it on the device.  (The decode checks
// that every stride consumes exactly the bits between two snapshots; the states in a snapshot are taken as they are:
// a container that stores an index protects it with a checksum -- archive.hpp's DecodeIndexFile does.)
static int index_accept(fqgpu_dblock *b, int stream, const void *data, size_t len) {
  const unsigned B = stream ? FQGPU_QUAL_MODELS : FQGPU_SEQ_MODELS;
  FqIndexHeader h;
  if (len < sizeof(h)) return FQGPU_E_CORRUPT;
  memcpy(&h, data, sizeof(h));
  if (h.magic != FQ_INDEX_MAGIC || h.stream != (uint32_t)stream || h.stride == 0 || (h.stride & 65535u) ||
      h.n_sym != b->n_bases || h.n_snap != (h.n_sym ? (uint32_t)((h.n_sym - 1) / h.stride) : 0u) ||
      len != sizeof(h) + (size_t)h.n_snap * fq_index_snap_bytes(B))
    return FQGPU_E_CORRUPT;
  if (len > b->index_cap[stream]) {
    if (b->index[stream]) (void)hipFree(b->index[stream]);
    b->index[stream] = fq_dev_alloc<uint8_t>(len + 64);
    b->index_cap[stream] = b->index[stream] ? len : 0;
    if (!b->index[stream]) return FQGPU_E_NOMEM;
  }
  return FQGPU_OK;
}

extern "C" int fqgpu_dblock_load_index(fqgpu_ctx *ctx, fqgpu_dblock *b, int stream, const void *data, size_t len) {
  if (!ctx || !b || stream < 0 || stream > 1 || (!data && len)) return FQGPU_E_ARG;
  int rc = fqgpu_sync(ctx);
  if (rc) return rc;
  if (len == 0) { b->index_bytes[stream] = 0; return FQGPU_OK; }
  if ((rc = index_accept(b, stream, data, len))) return rc;
  FQ_HIP(hipMemcpy(b->index[stream], data, len, hipMemcpyHostToDevice));
  b->index_bytes[stream] = len;
  return FQGPU_OK;
}

extern "C" int fqgpu_ctx_set_lanes(fqgpu_ctx *ctx, unsigned lanes) {
  if (!ctx || lanes > FQ_MAX_LANES) return FQGPU_E_ARG;  // (0: by block size)
  int rc = fqgpu_sync(ctx);
  if (rc) return rc;
  ctx->n_lanes = lanes;
  ctx->next_lane = 0;
  return FQGPU_OK;
}

extern "C" int fqgpu_ctx_dump_tables(fqgpu_ctx *ctx, int stream, unsigned model, uint32_t *ctable_out,
                                     size_t ctable_cap_words, uint32_t *dtable_out, size_t dtable_cap_words) {
  if (!ctx || stream < 0 || stream > 1) return FQGPU_E_ARG;
  const FtLayout l = ft_layout(stream);
  if (model >= (unsigned)l.n_models) return FQGPU_E_ARG;
  int rc = use_device(ctx->device);
  if (rc) return rc;
  const DevTables &t = ctx->tab[stream];
  uint32_t lg = 0, co = 0, dof = 0;
  FQ_HIP(hipMemcpy(&lg, t.logs + model, 4, hipMemcpyDeviceToHost));
  FQ_HIP(hipMemcpy(&co, t.ct_off + model, 4, hipMemcpyDeviceToHost));
  FQ_HIP(hipMemcpy(&dof, t.dt_off + model, 4, hipMemcpyDeviceToHost));
  const size_t cw = 1 + ((size_t)1 << (lg - 1)) + 2 * (size_t)l.alpha, dw = 1 + ((size_t)1 << lg);
  if (ctable_out) {
    if (ctable_cap_words < cw) return FQGPU_E_ARG;
    FQ_HIP(hipMemcpy(ctable_out, t.ct + co, cw * 4, hipMemcpyDeviceToHost));
  }
  if (dtable_out) {
    if (dtable_cap_words < dw) return FQGPU_E_ARG;
    FQ_HIP(hipMemcpy(dtable_out, t.dt + dof, dw * 4, hipMemcpyDeviceToHost));
    for (size_t k = 1; k < dw; k++) dtable_out[k] = FQ_DENTRY_TO_ZSTD(dtable_out[k]);  // the device keeps the walk's arrangement
  }
  return FQGPU_OK;
}

// ------------------------------------------------------------------ device-resident blocks
extern "C" void fqgpu_dblock_destroy(fqgpu_dblock *b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  void *ps[] = {b->raw, b->recs, b->seq, b->qual, b->readlens, b->n_count, b->n_pos, b->result, b->index[0], b->index[1]};
  for (void *p : ps) if (p) (void)hipFree(p);
  if (b->ev_encoded) (void)hipEventDestroy(b->ev_encoded);
  delete b;
}

// stream / side-stream buffers of a block whose raw_len, n_recs, n_bases and n_pos_cap are set
static int alloc_block_outputs(fqgpu_dblock *b) {
  b->seq_cap = fqgpu_bound_seq(b->n_bases);
  b->qual_cap = fqgpu_bound_qual(b->n_bases);
  b->seq_alloc = b->seq_cap + 64;
  b->qual_alloc = b->qual_cap + 64;
  b->seq = fq_dev_alloc<uint8_t>(b->seq_alloc);
  b->qual = fq_dev_alloc<uint8_t>(b->qual_alloc);
  b->readlens = fq_dev_alloc<uint16_t>(b->n_recs);
  b->n_count = fq_dev_alloc<uint16_t>(b->n_recs);
  b->n_pos = fq_dev_alloc<uint16_t>(b->n_pos_cap + 16);
  b->result = fq_dev_alloc<BlockResult>(1);
  if (!b->seq || !b->qual || !b->readlens || !b->n_count || !b->n_pos || !b->result) return FQGPU_E_NOMEM;
  memset(&b->host_result, 0, sizeof(b->host_result));
  return FQGPU_OK;
}

extern "C" int fqgpu_dblock_create(fqgpu_ctx *ctx, const uint8_t *raw, size_t raw_len, const fqgpu_rec *recs,
                                   size_t n_recs, fqgpu_dblock **out) {
  if (!ctx || !raw || !recs || !out || !n_recs) return FQGPU_E_ARG;
  *out = nullptr;
  int rc = use_device(ctx->device);
  if (rc) return rc;
  size_t n_bases = 0;
  if ((rc = check_recs(recs, n_recs, raw_len, &n_bases))) return rc;
  // number of N bases = entries of n_pos (the reference grows the vector as it goes)
  size_t n_n = 0;
  for (size_t i = 0; i < n_recs; i++) {
    const uint8_t *s = raw + recs[i].seq_off, *e = s + recs[i].len;
    while ((s = static_cast<const uint8_t *>(memchr(s, 'N', (size_t)(e - s)))) != nullptr) { n_n++; s++; }
  }
  fqgpu_dblock *b = new (std::nothrow) fqgpu_dblock();
  if (!b) return FQGPU_E_NOMEM;
  b->device = ctx->device;
  b->owner = ctx;
  b->raw_len = raw_len; b->n_recs = n_recs; b->n_bases = n_bases;
  b->n_pos_cap = n_n;
  b->raw = fq_dev_alloc<uint8_t>(raw_len + 64);
  b->recs = fq_dev_alloc<fqgpu_rec>(n_recs);
  if (!b->raw || !b->recs || alloc_block_outputs(b) != FQGPU_OK) {
    fqgpu_dblock_destroy(b);
    return FQGPU_E_NOMEM;
  }
  hipError_t he = hipMemcpyAsync(b->raw, raw, raw_len, hipMemcpyHostToDevice, ctx->stream);
  if (he == hipSuccess) he = hipMemcpyAsync(b->recs, recs, n_recs * sizeof(fqgpu_rec), hipMemcpyHostToDevice, ctx->stream);
  if (he == hipSuccess) he = hipMemsetAsync(b->result, 0, sizeof(BlockResult), ctx->stream);
  if (he == hipSuccess) he = hipStreamSynchronize(ctx->stream);
  if (he != hipSuccess) { fqgpu_dblock_destroy(b); return fq_hip_error(he, __FILE__, __LINE__); }
  *out = b;
  return FQGPU_OK;
}

int fq_parse_on_device(hipStream_t st, const uint8_t *raw_dev, size_t raw_len, DevBuf &scan_tmp,
                       fqgpu_rec **recs_dev, size_t *n_recs, size_t *n_bases, size_t *n_n);

// Raw block in, record table built on the GPU (parse.hip): replaces FastqReader::parseRecords
// (reference src/fastq_io.cpp:67-125) for callers that hand over unparsed chunks.
extern "C" int fqgpu_dblock_create_from_raw(fqgpu_ctx *ctx, const uint8_t *raw, size_t raw_len, fqgpu_dblock **out) {
  if (!ctx || !raw || !out || !raw_len) return FQGPU_E_ARG;
  *out = nullptr;
  int rc = use_device(ctx->device);
  if (rc) return rc;
  fqgpu_dblock *b = new (std::nothrow) fqgpu_dblock();
  if (!b) return FQGPU_E_NOMEM;
  b->device = ctx->device;
  b->owner = ctx;
  b->raw = fq_dev_alloc<uint8_t>(raw_len + 64);
  if (!b->raw) { fqgpu_dblock_destroy(b); return FQGPU_E_NOMEM; }
  hipError_t he = hipMemsetAsync(b->raw + raw_len, 0, 64, ctx->stream);
  if (he == hipSuccess) he = hipMemcpyAsync(b->raw, raw, raw_len, hipMemcpyHostToDevice, ctx->stream);
  if (he != hipSuccess) { fqgpu_dblock_destroy(b); return fq_hip_error(he, __FILE__, __LINE__); }
  size_t n_n = 0;
  rc = fq_parse_on_device(ctx->stream, b->raw, raw_len, ctx->scan_tmp, &b->recs, &b->n_recs, &b->n_bases, &n_n);
  if (rc) { fqgpu_dblock_destroy(b); return rc; }
  // the block ends with the last complete record (partial tail ignored like the reference)
  fqgpu_rec last;
  he = hipMemcpy(&last, b->recs + (b->n_recs - 1), sizeof(last), hipMemcpyDeviceToHost);
  if (he != hipSuccess) { fqgpu_dblock_destroy(b); return fq_hip_error(he, __FILE__, __LINE__); }
  b->raw_len = (size_t)last.qual_off + last.len + 1;
  b->n_pos_cap = n_n;
  if ((rc = alloc_block_outputs(b))) { fqgpu_dblock_destroy(b); return rc; }
  he = hipMemsetAsync(b->result, 0, sizeof(BlockResult), ctx->stream);
  if (he == hipSuccess) he = hipStreamSynchronize(ctx->stream);
  if (he != hipSuccess) { fqgpu_dblock_destroy(b); return fq_hip_error(he, __FILE__, __LINE__); }
  *out = b;
  return FQGPU_OK;
}

extern "C" int fqgpu_dblock_records(fqgpu_ctx *ctx, const fqgpu_dblock *b, fqgpu_rec *recs_out, size_t cap,
                                    size_t *n_recs, size_t *raw_len) {
  if (!ctx || !b) return FQGPU_E_ARG;
  int rc = use_device(ctx->device);
  if (rc) return rc;
  if (n_recs) *n_recs = b->n_recs;
  if (raw_len) *raw_len = b->raw_len;
  if (recs_out) {
    const size_t n = cap < b->n_recs ? cap : b->n_recs;
    FQ_HIP(hipMemcpy(recs_out, b->recs, n * sizeof(fqgpu_rec), hipMemcpyDeviceToHost));
  }
  return FQGPU_OK;
}

extern "C" int fqgpu_dblock_encode(fqgpu_ctx *ctx, fqgpu_dblock *b, unsigned flags) {
  if (!ctx || !b || b->device != ctx->device) return FQGPU_E_ARG;
  int rc = use_device(ctx->device);
  if (rc) return rc;
  b->last_op = 1;
  b->result_pulled = false;
  return fq_encode_launch(ctx, b, flags);
}

extern "C" int fqgpu_dblock_wipe(fqgpu_ctx *ctx, fqgpu_dblock *b) {
  if (!ctx || !b || b->device != ctx->device) return FQGPU_E_ARG;
  int rc = use_device(ctx->device);
  if (rc) return rc;
  return fq_wipe_launch(ctx, b);
}

extern "C" int fqgpu_sync(fqgpu_ctx *ctx) {
  if (!ctx) return FQGPU_E_ARG;
  int rc = use_device(ctx->device);
  if (rc) return rc;
  for (int i = 0; i < FQ_MAX_LANES; i++) {
    if (ctx->lanes[i].st_seq) FQ_HIP(hipStreamSynchronize(ctx->lanes[i].st_seq));
    if (ctx->lanes[i].st_qual) FQ_HIP(hipStreamSynchronize(ctx->lanes[i].st_qual));
  }
  if (ctx->stream) FQ_HIP(hipStreamSynchronize(ctx->stream));
  return FQGPU_OK;
}

// Reads the device result block; call after fqgpu_sync.  Encode results also set the
// stream sizes a later decode of this block uses.
static int pull_result(fqgpu_dblock *b, bool from_encode) {
  FQ_HIP(hipMemcpy(&b->host_result, b->result, sizeof(BlockResult), hipMemcpyDeviceToHost));
  b->result_pulled = true;
  const BlockResult &r = b->host_result;
  if (r.s[1].bad_symbol || r.s[0].bad_symbol) return FQGPU_E_ARG;
  if (from_encode) {
    if (r.s[0].overflow || r.s[1].overflow) return FQGPU_E_OVERFLOW;
    b->seq_len = (size_t)r.s[0].len;
    b->qual_len = (size_t)r.s[1].len;
    b->n_pos_len = (size_t)r.n_pos_len;
  } else if (r.s[0].corrupt || r.s[1].corrupt) {
    return FQGPU_E_CORRUPT;
  }
  return FQGPU_OK;
}

extern "C" int fqgpu_dblock_status(const fqgpu_dblock *b, size_t *seq_len, size_t *qual_len, size_t *n_pos_len,
                                   size_t *n_bases) {
  if (!b) return FQGPU_E_ARG;
  fqgpu_dblock *mb = const_cast<fqgpu_dblock *>(b);
  (void)hipSetDevice(b->device);
  // The lanes run on non-blocking streams: a copy on the null stream is not ordered behind them.
  // An operation still in flight is waited for here (its sizes would otherwise read zero or stale).
  int rc = FQGPU_OK;
  if (b->last_op && !b->result_pulled && b->owner && (rc = fqgpu_sync(b->owner))) return rc;
  rc = b->last_op ? pull_result(mb, b->last_op == 1) : FQGPU_OK;
  if (seq_len) *seq_len = b->seq_len;
  if (qual_len) *qual_len = b->qual_len;
  if (n_pos_len) *n_pos_len = b->n_pos_len;
  if (n_bases) *n_bases = b->n_bases;
  return rc;
}

extern "C" int fqgpu_dblock_longest_chain(const fqgpu_dblock *b, unsigned *seq_segments, unsigned *qual_segments) {
  if (!b) return FQGPU_E_ARG;
  (void)hipSetDevice(b->device);
  BlockResult tmp;
  FQ_HIP(hipMemcpy(&tmp, b->result, sizeof(tmp), hipMemcpyDeviceToHost));
  if (seq_segments) *seq_segments = tmp.s[0].refixed;
  if (qual_segments) *qual_segments = tmp.s[1].refixed;
  return FQGPU_OK;
}

extern "C" int fqgpu_dblock_qual_segment_classes(fqgpu_ctx *ctx, const fqgpu_dblock *b, size_t counts[4]) {
  if (!ctx || !b || !counts) return FQGPU_E_ARG;
  counts[0] = counts[1] = counts[2] = counts[3] = 0;
  int rc = fqgpu_sync(ctx);
  if (rc) return rc;
  if (!b->diag_cls || !b->diag_n_segs) return FQGPU_E_ARG;  // no encode of this block yet
  uint32_t n = 0;
  FQ_HIP(hipMemcpy(&n, b->diag_n_segs, 4, hipMemcpyDeviceToHost));
  if (!n) return FQGPU_OK;
  uint8_t *cls = static_cast<uint8_t *>(malloc(n));
  if (!cls) return FQGPU_E_NOMEM;
  const hipError_t he = hipMemcpy(cls, b->diag_cls, n, hipMemcpyDeviceToHost);
  if (he == hipSuccess)
    for (uint32_t i = 0; i < n; i++) counts[cls[i] == 1 ? 0 : cls[i] == 0xFF ? 2 : cls[i] == 0 ? 3 : 1]++;
  free(cls);
  return he == hipSuccess ? FQGPU_OK : fq_hip_error(he, __FILE__, __LINE__);
}

extern "C" int fqgpu_dblock_fetch(fqgpu_ctx *ctx, const fqgpu_dblock *b, uint8_t *seq_out, uint8_t *qual_out,
                                  uint16_t *readlens_out, uint16_t *n_count_out, uint16_t *n_pos_out,
                                  uint8_t *raw_out) {
  if (!ctx || !b) return FQGPU_E_ARG;
  int rc = fqgpu_sync(ctx);
  if (rc) return rc;
  // sizes of an encode nobody has asked the status of yet
  if (b->last_op && !b->result_pulled && (rc = pull_result(const_cast<fqgpu_dblock *>(b), b->last_op == 1))) return rc;
  if (seq_out && b->seq_len) FQ_HIP(hipMemcpy(seq_out, b->seq, b->seq_len, hipMemcpyDeviceToHost));
  if (qual_out && b->qual_len) FQ_HIP(hipMemcpy(qual_out, b->qual, b->qual_len, hipMemcpyDeviceToHost));
  if (readlens_out) FQ_HIP(hipMemcpy(readlens_out, b->readlens, b->n_recs * 2, hipMemcpyDeviceToHost));
  if (n_count_out) FQ_HIP(hipMemcpy(n_count_out, b->n_count, b->n_recs * 2, hipMemcpyDeviceToHost));
  if (n_pos_out && b->n_pos_len) FQ_HIP(hipMemcpy(n_pos_out, b->n_pos, b->n_pos_len * 2, hipMemcpyDeviceToHost));
  if (raw_out) FQ_HIP(hipMemcpy(raw_out, b->raw, b->raw_len, hipMemcpyDeviceToHost));
  return FQGPU_OK;
}

extern "C" int fqgpu_dblock_load_streams(fqgpu_ctx *ctx, fqgpu_dblock *b, const uint8_t *seq, size_t seq_len,
                                         const uint8_t *qual, size_t qual_len, const uint16_t *n_count,
                                         const uint16_t *n_pos, size_t n_pos_len) {
  if (!ctx || !b || !seq || !qual || !n_count || (!n_pos && n_pos_len)) return FQGPU_E_ARG;
  int rc = fqgpu_sync(ctx);
  if (rc) return rc;
  // foreign streams may be larger than what this block's own encode would need
  // (seq_cap / qual_cap stay the capacities a later re-encode of this block is judged against)
  if (seq_len + 64 > b->seq_alloc) {
    (void)hipFree(b->seq);
    b->seq = fq_dev_alloc<uint8_t>(seq_len + 64);
    b->seq_alloc = b->seq ? seq_len + 64 : 0;
  }
  if (qual_len + 64 > b->qual_alloc) {
    (void)hipFree(b->qual);
    b->qual = fq_dev_alloc<uint8_t>(qual_len + 64);
    b->qual_alloc = b->qual ? qual_len + 64 : 0;
  }
  if (n_pos_len > b->n_pos_cap) {
    (void)hipFree(b->n_pos);
    b->n_pos = fq_dev_alloc<uint16_t>(n_pos_len + 16);
    b->n_pos_cap = n_pos_len;
  }
  if (!b->seq || !b->qual || !b->n_pos) return FQGPU_E_NOMEM;
  FQ_HIP(hipMemset(b->seq + seq_len, 0, 16));  // the bit reader loads whole dwords
  FQ_HIP(hipMemset(b->qual + qual_len, 0, 16));
  FQ_HIP(hipMemcpy(b->seq, seq, seq_len, hipMemcpyHostToDevice));
  FQ_HIP(hipMemcpy(b->qual, qual, qual_len, hipMemcpyHostToDevice));
  FQ_HIP(hipMemcpy(b->n_count, n_count, b->n_recs * 2, hipMemcpyHostToDevice));
  if (n_pos_len) FQ_HIP(hipMemcpy(b->n_pos, n_pos, n_pos_len * 2, hipMemcpyHostToDevice));
  b->seq_len = seq_len; b->qual_len = qual_len; b->n_pos_len = n_pos_len;
  b->last_op = 0;  // the result block no longer describes these streams
  b->result_pulled = true;
  b->index_bytes[0] = b->index_bytes[1] = 0;  // an index belongs to the streams it was made for
  return FQGPU_OK;
}

extern "C" int fqgpu_dblocks_decode(fqgpu_ctx *ctx, fqgpu_dblock *const *blocks, size_t n_blocks) {
  if (!ctx || (!blocks && n_blocks)) return FQGPU_E_ARG;
  int rc = use_device(ctx->device);
  if (rc) return rc;
  for (size_t i = 0; i < n_blocks; i++)
    if (!blocks[i] || blocks[i]->device != ctx->device) return FQGPU_E_ARG;
  // blocks that come straight out of an encode: wait for it and take over its stream sizes
  bool synced = false;
  for (size_t i = 0; i < n_blocks; i++) {
    fqgpu_dblock *b = blocks[i];
    if (b->last_op == 1 && !b->result_pulled) {
      if (!synced && (rc = fqgpu_sync(ctx))) return rc;
      synced = true;
      if ((rc = pull_result(b, true))) return rc;
    }
    if (!b->seq_len || !b->qual_len) return FQGPU_E_ARG;
  }
  for (size_t i = 0; i < n_blocks; i++) { blocks[i]->last_op = 2; blocks[i]->result_pulled = false; }
  return fq_decode_launch(ctx, blocks, n_blocks);
}

// ------------------------------------------------------------------ host-pointer convenience calls
// The two calls below stage through one device block the handle keeps between calls: a worker
// that codes chunk after chunk (reference src/process.cpp:49-54) pays hipMalloc only while its
// chunks are still growing.
template <class T>
static bool hp_grow(T *&p, size_t &have, size_t need) {
  if (p && need <= have) return true;
  if (p) (void)hipFree(p);
  have = need + need / 8;
  p = fq_dev_alloc<T>(have);
  if (!p) have = 0;
  return p != nullptr;
}

static int hp_block_acquire(fqgpu_ctx *ctx, size_t raw_len, size_t n_recs, size_t n_bases, size_t seq_cap,
                            size_t qual_cap, size_t n_pos_cap, fqgpu_dblock **out) {
  fqgpu_dblock *b = ctx->hp_block;
  if (!b) {
    b = new (std::nothrow) fqgpu_dblock();
    if (!b) return FQGPU_E_NOMEM;
    b->device = ctx->device;
    b->owner = ctx;
    b->result = fq_dev_alloc<BlockResult>(1);
    if (!b->result) { delete b; return FQGPU_E_NOMEM; }  // published only when complete
    ctx->hp_block = b;
  }
  size_t side2 = ctx->hp_side;
  const bool ok = hp_grow(b->raw, ctx->hp_raw, raw_len + 64) && hp_grow(b->recs, ctx->hp_recs, n_recs) &&
                  hp_grow(b->seq, ctx->hp_seq, seq_cap + 64) && hp_grow(b->qual, ctx->hp_qual, qual_cap + 64) &&
                  hp_grow(b->readlens, ctx->hp_side, n_recs) && hp_grow(b->n_count, side2, n_recs) &&
                  hp_grow(b->n_pos, ctx->hp_npos, n_pos_cap + 16);
  if (!ok) {  // leave nothing half-sized behind
    fqgpu_dblock_destroy(b);
    ctx->hp_block = nullptr;
    ctx->hp_raw = ctx->hp_recs = ctx->hp_seq = ctx->hp_qual = ctx->hp_side = ctx->hp_npos = 0;
    return FQGPU_E_NOMEM;
  }
  b->raw_len = raw_len; b->n_recs = n_recs; b->n_bases = n_bases;
  b->seq_cap = seq_cap; b->qual_cap = qual_cap; b->n_pos_cap = n_pos_cap;
  b->seq_alloc = ctx->hp_seq; b->qual_alloc = ctx->hp_qual;
  b->seq_len = b->qual_len = b->n_pos_len = 0;
  b->index_bytes[0] = b->index_bytes[1] = 0;
  b->last_op = 0;
  b->result_pulled = true;
  memset(&b->host_result, 0, sizeof(b->host_result));
  *out = b;
  return FQGPU_OK;
}

// Everything a block of this shape will need -- the staging block of the host-pointer calls and the
// scratch of every encode lane -- allocated now: the first block of a worker otherwise pays half a
// second of hipMalloc (2.3 GB of scratch per 256 MiB block and lane) inside its timed loop.
extern "C" int fqgpu_ctx_reserve(fqgpu_ctx *ctx, size_t raw_len, size_t n_recs, size_t n_bases) {
  if (!ctx || !raw_len || !n_recs || !n_bases || n_bases >= 0xFFF00000ull) return FQGPU_E_ARG;
  int rc = use_device(ctx->device);
  if (rc) return rc;
  if ((rc = fqgpu_sync(ctx))) return rc;
  ctx->hp_pending = false;
  fqgpu_dblock *b = nullptr;
  if ((rc = hp_block_acquire(ctx, raw_len, n_recs, n_bases, fqgpu_bound_seq(n_bases), fqgpu_bound_qual(n_bases), n_bases / 64 + 1024, &b))) return rc;
  if ((rc = ctx->hp_parse.cnt.reserve((raw_len / 4096 + 2) * 4)) || (rc = ctx->hp_parse.base.reserve((raw_len / 4096 + 3) * 4)) ||
      (rc = ctx->hp_parse.nl_pos.reserve((n_recs * 4 + 8) * 4)))
    return rc;
  const unsigned first = ctx->next_lane;
  for (unsigned l = 0; l < fq_lanes_for(ctx, n_bases) && !rc; l++) rc = fq_encode_launch(ctx, b, 0, nullptr, nullptr, true);
  ctx->next_lane = first;
  return rc;
}

// Waits for everything the handle has queued before an error return: the caller's buffers (page-locked
// vectors that go back to the pin cache, where another thread may pick them up) must not be read or
// written by a copy that is still in flight.
static int hp_fail(fqgpu_ctx *ctx, int rc) {
  ctx->hp_pending = false;
  (void)fqgpu_sync(ctx);
  return rc;
}
#define FQ_HIP_HP(call)                                                                       \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) return hp_fail(ctx, fq_hip_error(e_, __FILE__, __LINE__));          \
  } while (0)

// caller_seq_cap / caller_qual_cap: capacities the overflow rule is judged against (0: the reference's bounds)
static int hp_encode_begin(fqgpu_ctx *ctx, const uint8_t *raw, size_t raw_len, const fqgpu_rec *recs, size_t n_recs,
                           unsigned flags, size_t caller_seq_cap, size_t caller_qual_cap, size_t *n_recs_out,
                           size_t *n_bases_out, size_t *used_len) {
  if (!ctx || !raw || !raw_len || (recs && !n_recs)) return FQGPU_E_ARG;
  int rc = use_device(ctx->device);
  if (rc) return rc;
  ctx->hp_pending = false;  // (a block begun and never collected is dropped: fqgpu_sync below waits for it)
  ctx->hp_hdr.pending = ctx->hp_hdr.collected = false;
  size_t n_bases = 0, n_n = 0, used = raw_len;
  if (recs && (rc = check_recs(recs, n_recs, raw_len, &n_bases))) return rc;
  if ((rc = fqgpu_sync(ctx))) return rc;
  fqgpu_dblock *b = nullptr;
  // the raw buffer first: without a record table the other sizes are known only after the device has counted the lines
  if ((rc = hp_block_acquire(ctx, raw_len, recs ? n_recs : 0, n_bases, 0, 0, 0, &b))) return rc;
  if (!ctx->hp_ev_h2d) FQ_HIP(hipEventCreateWithFlags(&ctx->hp_ev_h2d, hipEventDisableTiming));
  if (!ctx->hp_result) FQ_HIP(hipHostMalloc(reinterpret_cast<void **>(&ctx->hp_result), sizeof(BlockResult), hipHostMallocPortable));
  FQ_HIP_HP(hipMemsetAsync(b->raw + raw_len, 0, 64, ctx->stream));
  FQ_HIP_HP(hipMemcpyAsync(b->raw, raw, raw_len, hipMemcpyHostToDevice, ctx->stream));
  if (!recs) {
    if ((rc = fq_parse_count(ctx->stream, b->raw, raw_len, ctx->hp_parse, &n_recs))) return hp_fail(ctx, rc);
    if (!hp_grow(b->recs, ctx->hp_recs, n_recs)) return hp_fail(ctx, FQGPU_E_NOMEM);
    if ((rc = fq_parse_records(ctx->stream, b->raw, raw_len, ctx->hp_parse, b->recs, n_recs, &n_bases, &n_n, &used))) return hp_fail(ctx, rc);
  }
  // The reference's capacities are the ones the overflow rule is judged against.  n_pos: the device
  // parser counted the N's; with a caller's table the worst case (every base an N) is provided for.
  if ((rc = hp_block_acquire(ctx, used, n_recs, n_bases, caller_seq_cap ? caller_seq_cap : fqgpu_bound_seq(n_bases),
                             caller_qual_cap ? caller_qual_cap : fqgpu_bound_qual(n_bases), recs ? n_bases : n_n, &b)))
    return hp_fail(ctx, rc);
  if (recs) FQ_HIP_HP(hipMemcpyAsync(b->recs, recs, n_recs * sizeof(fqgpu_rec), hipMemcpyHostToDevice, ctx->stream));
  FQ_HIP_HP(hipEventRecord(ctx->hp_ev_h2d, ctx->stream));
  b->last_op = 1;
  b->result_pulled = false;
  hipStream_t st = nullptr;
  // (the device copy of raw is scratch here: its N's are patched on the host by fqgpu_encode_end)
  if ((rc = fq_encode_launch(ctx, b, flags & ~FQGPU_F_WRITE_BACK_N, ctx->hp_ev_h2d, &st))) return hp_fail(ctx, rc);
  FQ_HIP_HP(hipMemcpyAsync(ctx->hp_result, b->result, sizeof(BlockResult), hipMemcpyDeviceToHost, st));
  ctx->hp_pending = true;
  ctx->hp_flags = flags;
  ctx->hp_done = st;
  ctx->hp_used = used;
  if (n_recs_out) *n_recs_out = n_recs;
  if (n_bases_out) *n_bases_out = n_bases;
  if (used_len) *used_len = used;
  return FQGPU_OK;
}

extern "C" int fqgpu_encode_begin(fqgpu_ctx *ctx, const uint8_t *raw, size_t raw_len, const fqgpu_rec *recs, size_t n_recs,
                                  unsigned flags, size_t *n_recs_out, size_t *n_bases_out, size_t *used_len) {
  return hp_encode_begin(ctx, raw, raw_len, recs, n_recs, flags, 0, 0, n_recs_out, n_bases_out, used_len);
}

extern "C" int fqgpu_encode_records(fqgpu_ctx *ctx, fqgpu_rec *recs_out, size_t cap) {
  if (!ctx || !recs_out || !ctx->hp_pending || !ctx->hp_block || cap < ctx->hp_block->n_recs) return FQGPU_E_ARG;
  int rc = use_device(ctx->device);
  if (rc) return rc;
  // on the handle's copy stream: the table was built (or uploaded) there; the lanes' kernels are not waited for
  FQ_HIP(hipMemcpyAsync(recs_out, ctx->hp_block->recs, ctx->hp_block->n_recs * sizeof(fqgpu_rec), hipMemcpyDeviceToHost, ctx->stream));
  FQ_HIP(hipStreamSynchronize(ctx->stream));
  return FQGPU_OK;
}

// waits for the block in flight and reads its result block; the block stays pending
static int hp_collect(fqgpu_ctx *ctx) {
  fqgpu_dblock *b = ctx->hp_block;
  if (b->result_pulled) return FQGPU_OK;
  FQ_HIP_HP(hipStreamSynchronize(ctx->hp_done));
  b->host_result = *ctx->hp_result;
  b->result_pulled = true;
  const BlockResult &r = b->host_result;
  if (r.s[0].bad_symbol || r.s[1].bad_symbol) return hp_fail(ctx, FQGPU_E_ARG);
  if (r.s[0].overflow || r.s[1].overflow) return hp_fail(ctx, FQGPU_E_OVERFLOW);
  b->seq_len = (size_t)r.s[0].len; b->qual_len = (size_t)r.s[1].len; b->n_pos_len = (size_t)r.n_pos_len;
  return FQGPU_OK;
}

extern "C" int fqgpu_encode_wait(fqgpu_ctx *ctx, size_t *seq_len, size_t *qual_len, size_t *n_pos_len) {
  if (!ctx || !ctx->hp_pending || !ctx->hp_block) return FQGPU_E_ARG;
  int rc = use_device(ctx->device);
  if (rc) return rc;
  if ((rc = hp_collect(ctx))) return rc;
  if (seq_len) *seq_len = ctx->hp_block->seq_len;
  if (qual_len) *qual_len = ctx->hp_block->qual_len;
  if (n_pos_len) *n_pos_len = ctx->hp_block->n_pos_len;
  return FQGPU_OK;
}

extern "C" int fqgpu_encode_cancel(fqgpu_ctx *ctx) {
  if (!ctx) return FQGPU_E_ARG;
  int rc = use_device(ctx->device);
  if (rc) return rc;
  ctx->hp_pending = false;
  ctx->hp_hdr.pending = ctx->hp_hdr.collected = false;
  return fqgpu_sync(ctx);
}

// The header fields of the block in flight (headers.hip), on the handle's copy stream -- where the chunk arrived and
// its record table was uploaded or built -- beside the lane's encode kernels.
extern "C" int fqgpu_encode_headers_begin(fqgpu_ctx *ctx, const uint8_t *field_types, const char *separators, unsigned n_fields,
                                          const uint8_t *first_header, size_t first_header_len) {
  if (!ctx || !ctx->hp_pending || !ctx->hp_block || ctx->hp_hdr.pending || !field_types || !first_header || (n_fields > 1 && !separators))
    return FQGPU_E_ARG;
  int rc = use_device(ctx->device);
  if (rc) return rc;
  const fqgpu_dblock *b = ctx->hp_block;
  if ((rc = fq_headers_launch(ctx->stream, b->raw, ctx->hp_used, b->recs, b->n_recs, b->n_bases, field_types, separators, n_fields,
                              first_header, first_header_len, ctx->hp_hdr)))
    return rc == FQGPU_E_ARG ? rc : hp_fail(ctx, rc);
  ctx->hp_hdr.pending = true;
  ctx->hp_hdr.collected = false;
  return FQGPU_OK;
}

extern "C" int fqgpu_encode_headers_wait(fqgpu_ctx *ctx, fqgpu_field_sizes *sizes, size_t *total_bytes, size_t *bad_record) {
  if (!ctx || !ctx->hp_hdr.pending) return FQGPU_E_ARG;
  int rc = use_device(ctx->device);
  if (rc) return rc;
  HdrScratch &hs = ctx->hp_hdr;
  if (!hs.collected) {
    FQ_HIP_HP(hipStreamSynchronize(ctx->stream));
    hs.collected = true;
  }
  const HdrResult &r = *hs.host_res;
  if (r.first_error != ~0ull) {
    if ((r.first_error & 0xFFu) == 3u || r.total > hs.bound) return FQGPU_E_ARG;  // a record table whose headers overlap: not this chunk's
    if (bad_record) *bad_record = (size_t)(r.first_error >> 8);
    return FQGPU_E_HEADER;
  }
  for (unsigned i = 0; sizes && i < hs.n_fields; i++) sizes[i] = {r.size[3 * i], r.size[3 * i + 1], r.size[3 * i + 2]};
  if (total_bytes) *total_bytes = (size_t)r.total;
  return FQGPU_OK;
}

extern "C" int fqgpu_encode_headers_end(fqgpu_ctx *ctx, uint8_t *out, size_t out_cap) {
  if (!ctx || !ctx->hp_hdr.pending || !out) return FQGPU_E_ARG;
  size_t total = 0;
  int rc = fqgpu_encode_headers_wait(ctx, nullptr, &total, nullptr);
  if (rc) return rc;
  if (out_cap < total) return FQGPU_E_ARG;
  HdrScratch &hs = ctx->hp_hdr;
  FQ_HIP_HP(hipMemcpyAsync(out, hs.out.p, total, hipMemcpyDeviceToHost, ctx->stream));
  FQ_HIP_HP(hipStreamSynchronize(ctx->stream));
  hs.pending = hs.collected = false;
  return FQGPU_OK;
}

extern "C" int fqgpu_encode_end(fqgpu_ctx *ctx, uint8_t *raw, uint8_t *seq_out, size_t seq_cap, size_t *seq_len,
                                uint8_t *qual_out, size_t qual_cap, size_t *qual_len, uint16_t *readlens_out,
                                uint16_t *n_count_out, uint16_t *n_pos_out, size_t n_pos_cap, size_t *n_pos_len) {
  if (!ctx || !ctx->hp_pending || !ctx->hp_block) return FQGPU_E_ARG;
  int rc = use_device(ctx->device);
  if (rc) return rc;
  if (!seq_out || !qual_out || !seq_len || !qual_len) return hp_fail(ctx, FQGPU_E_ARG);
  fqgpu_dblock *b = ctx->hp_block;
  hipStream_t st = ctx->hp_done;
  const size_t n_recs = b->n_recs;
  if ((rc = hp_collect(ctx))) return rc;
  ctx->hp_pending = false;
  if (b->seq_len > seq_cap || b->qual_len > qual_cap) return FQGPU_E_OVERFLOW;
  if (n_pos_out && b->n_pos_len > n_pos_cap) return FQGPU_E_ARG;
  // N -> A on the host needs the N tables even if the caller does not want them
  std::vector<uint16_t> tmp_cnt, tmp_pos;
  const bool patch = raw && (ctx->hp_flags & FQGPU_F_WRITE_BACK_N) && b->n_pos_len;
  uint16_t *cnt_h = n_count_out, *pos_h = n_pos_out;
  if (patch && !cnt_h) { tmp_cnt.resize(n_recs); cnt_h = tmp_cnt.data(); }
  if (patch && !pos_h) { tmp_pos.resize(b->n_pos_len); pos_h = tmp_pos.data(); }
  std::vector<fqgpu_rec> tmp_recs;
  if (patch) tmp_recs.resize(n_recs);
  // (the side buffers -- record table, readlens, n_count, n_pos -- are best left PAGEABLE: as page-locked
  // buffers their small copies queue up in the DMA engines behind the other workers' 256 MiB uploads;
  // four threads: 47.9 GB/s with pageable, 40-42 with page-locked side buffers, same box)
  FQ_HIP_HP(hipMemcpyAsync(seq_out, b->seq, b->seq_len, hipMemcpyDeviceToHost, st));
  FQ_HIP_HP(hipMemcpyAsync(qual_out, b->qual, b->qual_len, hipMemcpyDeviceToHost, st));
  if (readlens_out) FQ_HIP_HP(hipMemcpyAsync(readlens_out, b->readlens, n_recs * 2, hipMemcpyDeviceToHost, st));
  if (cnt_h) FQ_HIP_HP(hipMemcpyAsync(cnt_h, b->n_count, n_recs * 2, hipMemcpyDeviceToHost, st));
  if (pos_h && b->n_pos_len) FQ_HIP_HP(hipMemcpyAsync(pos_h, b->n_pos, b->n_pos_len * 2, hipMemcpyDeviceToHost, st));
  if (patch) FQ_HIP_HP(hipMemcpyAsync(tmp_recs.data(), b->recs, n_recs * sizeof(fqgpu_rec), hipMemcpyDeviceToHost, st));
  FQ_HIP_HP(hipStreamSynchronize(st));
  if (patch) {  // replaceAndEncodeNs (src/fse_sequence.cpp:35-51): deltas to the previous N, the first one absolute
    size_t at = 0;
    for (size_t r = 0; r < n_recs; r++) {
      uint8_t *s = raw + tmp_recs[r].seq_off;
      unsigned pos = 0;
      for (unsigned k = cnt_h[r]; k > 0; k--) { pos += pos_h[at++]; s[pos] = 'A'; }
    }
  }
  *seq_len = b->seq_len;
  *qual_len = b->qual_len;
  if (n_pos_len) *n_pos_len = b->n_pos_len;
  return FQGPU_OK;
}

// The decode index of the block fqgpu_encode_begin coded with FQGPU_F_DECODE_INDEX: after fqgpu_encode_wait or _end,
// until the handle's next host-pointer call.
extern "C" int fqgpu_encode_index(fqgpu_ctx *ctx, int stream, uint8_t *out, size_t cap, size_t *len) {
  if (!ctx || !ctx->hp_block || stream < 0 || stream > 1 || !len) return FQGPU_E_ARG;
  const fqgpu_dblock *b = ctx->hp_block;
  if (b->last_op != 1 || !b->result_pulled) return FQGPU_E_ARG;
  *len = b->index_bytes[stream];
  if (!out || !*len) return FQGPU_OK;
  if (cap < *len) return FQGPU_E_ARG;
  int rc = use_device(ctx->device);
  if (rc) return rc;
  FQ_HIP(hipMemcpyAsync(out, b->index[stream], *len, hipMemcpyDeviceToHost, ctx->hp_done));
  FQ_HIP(hipStreamSynchronize(ctx->hp_done));
  return FQGPU_OK;
}

// Host-pointer encode, one block per call, as asynchronous as one call can be: the inputs go up on
// the handle's copy stream, the lane's kernels wait for that event (not for the host), the result
// block lands in page-locked memory behind the last kernel, and the streams come down with their
// exact sizes on the stream the encode ran on.  The host waits twice: for the sizes, for the
// streams.  With page-locked caller buffers (fqgpu_host_alloc; the shim's chunk and stream vectors
// use it) every copy runs at link rate and several worker threads -- one handle each, like the
// reference's one workspace per thread (src/process.cpp:49-54) -- overlap their copies with each
// other's kernels.  Pageable buffers work too, at the rate of the runtime's staging copies.
// N -> A write-back happens on the HOST from n_count / n_pos (a few thousand bytes to touch)
// instead of copying the whole raw block back over PCIe.
extern "C" int fqgpu_encode_block(fqgpu_ctx *ctx, uint8_t *raw, size_t raw_len, const fqgpu_rec *recs,
                                  size_t n_recs, uint8_t *seq_out, size_t seq_cap, size_t *seq_len,
                                  uint8_t *qual_out, size_t qual_cap, size_t *qual_len,
                                  uint16_t *readlens_out, uint16_t *n_count_out, uint16_t *n_pos_out,
                                  size_t n_pos_cap, size_t *n_pos_len, unsigned flags) {
  if (!ctx || !raw || !recs || !n_recs || !seq_out || !qual_out || !seq_len || !qual_len) return FQGPU_E_ARG;
  if (!seq_cap || !qual_cap) return FQGPU_E_ARG;
  // the caller's capacities are the ones the overflow rule is judged against
  int rc = hp_encode_begin(ctx, raw, raw_len, recs, n_recs, flags, seq_cap, qual_cap, nullptr, nullptr, nullptr);
  if (rc) return rc;
  return fqgpu_encode_end(ctx, raw, seq_out, seq_cap, seq_len, qual_out, qual_cap, qual_len, readlens_out, n_count_out, n_pos_out,
                          n_pos_cap, n_pos_len);
}

static int hp_decode(fqgpu_ctx *ctx, const uint8_t *seq, size_t seq_len, const uint8_t *qual, size_t qual_len, const uint16_t *n_count,
                     size_t n_count_len, const uint16_t *n_pos, size_t n_pos_len, const fqgpu_rec *recs, size_t n_recs, uint8_t *raw_out,
                     size_t raw_len, const uint8_t *const index[2], const size_t index_len[2]) {
  if (!ctx || !seq || !qual || !n_count || !recs || !raw_out || !seq_len || !qual_len) return FQGPU_E_ARG;
  if (n_count_len < n_recs) return FQGPU_E_CORRUPT;
  int rc = use_device(ctx->device);
  if (rc) return rc;
  size_t n_bases = 0;
  if ((rc = check_recs(recs, n_recs, raw_len, &n_bases))) return rc;
  if ((rc = fqgpu_sync(ctx))) return rc;
  fqgpu_dblock *b = nullptr;
  const size_t seq_cap = seq_len > fqgpu_bound_seq(n_bases) ? seq_len : fqgpu_bound_seq(n_bases);
  const size_t qual_cap = qual_len > fqgpu_bound_qual(n_bases) ? qual_len : fqgpu_bound_qual(n_bases);
  if ((rc = hp_block_acquire(ctx, raw_len, n_recs, n_bases, seq_cap, qual_cap, n_pos_len, &b))) return rc;
  // raw_out holds the skeleton the first decode pass laid out (headers, newlines, '+')
  // everything on the handle's stream (the decode kernels run there too): no host wait in between
  hipStream_t st = ctx->stream;
  FQ_HIP_HP(hipMemcpyAsync(b->raw, raw_out, raw_len, hipMemcpyHostToDevice, st));
  FQ_HIP_HP(hipMemcpyAsync(b->recs, recs, n_recs * sizeof(fqgpu_rec), hipMemcpyHostToDevice, st));
  FQ_HIP_HP(hipMemsetAsync(b->seq + seq_len, 0, 16, st));  // the bit reader loads whole dwords
  FQ_HIP_HP(hipMemsetAsync(b->qual + qual_len, 0, 16, st));
  FQ_HIP_HP(hipMemcpyAsync(b->seq, seq, seq_len, hipMemcpyHostToDevice, st));
  FQ_HIP_HP(hipMemcpyAsync(b->qual, qual, qual_len, hipMemcpyHostToDevice, st));
  // the reference pops from the END of n_count (src/fse_sequence.cpp:115-126)
  FQ_HIP_HP(hipMemcpyAsync(b->n_count, n_count + (n_count_len - n_recs), n_recs * 2, hipMemcpyHostToDevice, st));
  if (n_pos_len) FQ_HIP_HP(hipMemcpyAsync(b->n_pos, n_pos, n_pos_len * 2, hipMemcpyHostToDevice, st));
  for (int s = 0; s < 2; s++)  // (hp_block_acquire has dropped whatever index the staging block held)
    if (index[s] && index_len[s]) {
      if ((rc = index_accept(b, s, index[s], index_len[s]))) return hp_fail(ctx, rc);
      FQ_HIP_HP(hipMemcpyAsync(b->index[s], index[s], index_len[s], hipMemcpyHostToDevice, st));
      b->index_bytes[s] = index_len[s];
    }
  b->seq_len = seq_len; b->qual_len = qual_len; b->n_pos_len = n_pos_len;
  b->last_op = 2;
  b->result_pulled = false;
  fqgpu_dblock *one[1] = {b};
  if ((rc = fq_decode_launch(ctx, one, 1))) return hp_fail(ctx, rc);
  if (!ctx->hp_result) FQ_HIP_HP(hipHostMalloc(reinterpret_cast<void **>(&ctx->hp_result), sizeof(BlockResult), hipHostMallocPortable));
  // The copies back are issued only when the kernels are through: a copy that waits in a DMA
  // engine's queue for a 13 s decode kernel holds that engine, and the uploads of the next workers'
  // blocks queue up behind it (measured in the block farm: four decoding workers ran two and two,
  // the third worker's fifth hipMemcpyAsync returning after 12.8 s).
  FQ_HIP_HP(hipStreamSynchronize(st));
  FQ_HIP_HP(hipMemcpyAsync(ctx->hp_result, b->result, sizeof(BlockResult), hipMemcpyDeviceToHost, st));
  FQ_HIP_HP(hipMemcpyAsync(raw_out, b->raw, raw_len, hipMemcpyDeviceToHost, st));
  FQ_HIP_HP(hipStreamSynchronize(st));
  b->host_result = *ctx->hp_result;
  b->result_pulled = true;
  if (b->host_result.s[0].bad_symbol || b->host_result.s[1].bad_symbol) return FQGPU_E_ARG;
  if (b->host_result.s[0].corrupt || b->host_result.s[1].corrupt) return FQGPU_E_CORRUPT;
  return FQGPU_OK;
}

extern "C" int fqgpu_decode_block(fqgpu_ctx *ctx, const uint8_t *seq, size_t seq_len, const uint8_t *qual,
                                  size_t qual_len, const uint16_t *n_count, size_t n_count_len,
                                  const uint16_t *n_pos, size_t n_pos_len, const fqgpu_rec *recs,
                                  size_t n_recs, uint8_t *raw_out, size_t raw_len) {
  const uint8_t *const index[2] = {nullptr, nullptr};
  const size_t index_len[2] = {0, 0};
  return hp_decode(ctx, seq, seq_len, qual, qual_len, n_count, n_count_len, n_pos, n_pos_len, recs, n_recs, raw_out, raw_len, index, index_len);
}

// The same with the decode index the block's encode left (FQGPU_F_DECODE_INDEX, fqgpu_encode_index): each stream is
// decoded from every snapshot at once instead of by one lane from its end.
extern "C" int fqgpu_decode_block_indexed(fqgpu_ctx *ctx, const uint8_t *seq, size_t seq_len, const uint8_t *qual, size_t qual_len,
                                          const uint16_t *n_count, size_t n_count_len, const uint16_t *n_pos, size_t n_pos_len,
                                          const fqgpu_rec *recs, size_t n_recs, uint8_t *raw_out, size_t raw_len,
                                          const uint8_t *seq_index, size_t seq_index_len, const uint8_t *qual_index, size_t qual_index_len) {
  if ((seq_index_len && !seq_index) || (qual_index_len && !qual_index)) return FQGPU_E_ARG;
  const uint8_t *const index[2] = {seq_index, qual_index};
  const size_t index_len[2] = {seq_index_len, qual_index_len};
  return hp_decode(ctx, seq, seq_len, qual, qual_len, n_count, n_count_len, n_pos, n_pos_len, recs, n_recs, raw_out, raw_len, index, index_len);
}
