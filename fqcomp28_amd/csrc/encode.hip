// Bit-exact parallel encoder of one (block, stream) -- replaces the per-record
// loop of CompressionWorkspace::encodeChunk (reference src/workspace.cpp:25-31)
// with its SequenceEncoder::encodeRecord (src/fse_sequence.cpp:53-112) /
// QualityEncoder::encodeRecord (src/fse_quality.cpp:5-53) bodies and
// FSE_Encoder::startChunk/endChunk (src/fse_common.hpp:77-90).
//
// The reference interleaves one tANS state per context into ONE bitstream:
//   for every symbol e in encode order (records in file order, positions L-1..0):
//       emit low nb(e) bits of state[ctx(e)];  state[ctx(e)] = T[ctx(e)][sym(e)](state)
//   then flush all states (ctx ascending), one '1' end-mark bit, zero pad.
// ctx(e)/sym(e) depend on the INPUT only, each context's state chain only on the
// symbols of that context.  Hence the decomposition:
//   K1 tile_hist / K2 layout : counting sort keys = context, per-tile histograms + scan
//   K3 scatter               : stable partition of the symbols by context (rank inside a
//                              (tile, context) = one lane-ordered LDS atomic per symbol); the
//                              sequence stream sorts every 4 K batch by context in LDS and
//                              writes runs instead of scattered bytes
//   K4 chains                : a tANS encoder state never forgets its history completely, but
//                              the SET of states it can be in collapses fast.  A chain is cut
//                              into segments of 4096 symbols; the state is known without its
//                              past right after a symbol whose normalised count is 1 or -1 (one
//                              table cell: "reset"), and for a segment without such a symbol
//                              one wave computes the segment's FUNCTION entry state -> exit
//                              state for every possible entry state over collapsing state sets.
//                              Entry states follow by applying the functions along the chain;
//                              then one lane per segment emits (nb, bits).  No serial chain,
//                              no speculation.
//   K6 bitcount/scan/pack    : per-symbol (nb,bits) brought back into encode order (quality:
//                              gather by slot; sequence: runs of the batch read into LDS),
//                              exclusive scan of nb = bit offsets, bit packing through LDS
//   K7 epilogue              : state flush + end mark + size/overflow
#include "fqgpu_internal.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#ifdef FQGPU_EXPERIMENTS
__device__ unsigned long long g_ts_prof[32];
void fq_ts_prof_dump() {
  unsigned long long h[32];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ts_prof), sizeof(h)) != hipSuccess) return;
  const char *names[4] = {"partition<Qual>", "partition<Seq>", "gather_pack<Qual>", "gather_pack<Seq>"};
  for (int k = 0; k < 4; k++)
    fprintf(stderr, "ts_prof %-18s phase ticks (100 MHz, summed over workgroups): %llu %llu %llu %llu | ranker alone %llu\n", names[k], h[8 * k], h[8 * k + 1],
            h[8 * k + 2], h[8 * k + 3], h[8 * k + 4]);
}
#endif

namespace {

constexpr unsigned TILE_SEQ = 32768;   // symbols per partition tile (one wave each in K3)
constexpr unsigned TILE_QUAL = 65536;
constexpr unsigned GROUP_TILES = 64;   // tiles per group in the two-level tile scan
constexpr unsigned PACK_TILE = 4096;   // symbols per bit-packing tile
constexpr unsigned PACK_THREADS = 256;
constexpr unsigned PACK_PER_THREAD = PACK_TILE / PACK_THREADS;  // 16
constexpr unsigned CTX_PAD = 16;       // every context's sorted run starts 16-aligned

template <class M> constexpr unsigned tile_size() { return M::STREAM == 0 ? TILE_SEQ : TILE_QUAL; }

#include "enc_records_hist.h"
#include "enc_partition.h"
#include "enc_chains_seq.h"
#include "enc_chains_generic.h"
#include "enc_gather.h"
#include "enc_seq_batch.h"
#include "enc_index.h"
#include "enc_pack.h"
#include "enc_tile_sort.h"

// ------------------------------------------------------------------ host orchestration
#define FQ_SPAN_BEGIN(name) fq_timer_span_begin(ctx, name, st)
#define FQ_SPAN_END() fq_timer_span_end(ctx, st)

// Timing experiments (tools/traffic_experiment.py) exist only in a library built with
// -DFQGPU_EXPERIMENTS (make experiments -> tools/_build/libfqgpu_experiments.so); the product
// library contains none of these switches.  FQGPU_DEBUG_SKIP = bit mask of kernel groups that are
// NOT launched (their outputs keep the values of the previous encode of the same lane: wrong output
// by design); FQGPU_DEBUG_NO_SYM_STORE: 1 = both streams, 2 = sequence only, 3 = quality only.
#ifdef FQGPU_EXPERIMENTS
static int fq_debug_no_sym(int stream) {
  const char *e = getenv("FQGPU_DEBUG_NO_SYM_STORE");
  const int v = e ? atoi(e) : 0;
  return v == 1 || (v == 2 && stream == 0) || (v == 3 && stream == 1);
}
static unsigned fq_debug_skip() {
  const char *e = getenv("FQGPU_DEBUG_SKIP");
  return e ? (unsigned)strtoul(e, nullptr, 0) : 0u;
}
static int fq_debug_k1() { const char *e = getenv("FQGPU_DEBUG_K1"); return e ? atoi(e) : 0; }
static bool fq_debug_flag(const char *name) { return getenv(name) != nullptr; }
// FQGPU_DEBUG_SKIPK=name,name,...: single kernels that are not launched (k1 k3s k3q setfunc resolve emit scan stage1 heads qresolve walk2 k6s k6q k2s k2q)
static bool fq_debug_skipk(const char *name) {
  const char *e = getenv("FQGPU_DEBUG_SKIPK");
  if (!e) return false;
  const size_t n = strlen(name);
  for (const char *p = e; (p = strstr(p, name)) != nullptr; p += n)
    if ((p == e || p[-1] == ',') && (p[n] == 0 || p[n] == ',')) return true;
  return false;
}
#else
static constexpr int fq_debug_no_sym(int) { return 0; }
static constexpr unsigned fq_debug_skip() { return 0u; }
static constexpr int fq_debug_k1() { return 0; }
static constexpr bool fq_debug_flag(const char *) { return false; }
static constexpr bool fq_debug_skipk(const char *) { return false; }
#endif

// true: both streams take the tile-sorted path, so K1 can run once for both (k_tile_hist2)
static bool fused_k1(const fqgpu_ctx *ctx) {
  return ctx->lds_atomics_ordered && ctx->tile_sorted && !ctx->seq_generic && !fq_debug_skip() && !fq_debug_k1();
}

// keys and tile histograms of one stream exist before its pipeline starts (fused K1)
template <class M>
int reserve_k1(EncScratch &sc, unsigned n_sym) {
  const unsigned n_tiles = (n_sym + TS_TILE - 1) / TS_TILE;
  const size_t n_pad = ((size_t)n_sym + SC_BATCH_SEQ + 15) & ~(size_t)15;
  int rc;
  if ((rc = sc.keys.reserve(n_pad * 3))) return rc;
  if ((rc = sc.tile_hist.reserve((size_t)n_tiles * M::B * 4))) return rc;
  return FQGPU_OK;
}

template <class M>
int encode_stream(fqgpu_ctx *ctx, EncLane &lane, hipStream_t st, fqgpu_dblock *b,
                  const uint32_t *rec_start, uint8_t *out_dev, size_t cap, unsigned flags, bool reserve_only = false) {
  EncScratch &sc = lane.enc[M::STREAM];
  const DevTables &tab = ctx->tab[M::STREAM];
  constexpr unsigned B = M::B;
  const unsigned n_sym = (unsigned)b->n_bases;
  const unsigned R = (unsigned)b->n_recs;
  const bool serial_seq = M::STREAM == 0 && !ctx->seq_generic;
  // tile-sorted partition + fused gather/pack: whenever the rank may come from lane-ordered LDS atomics
  // (the generic sequence mode keeps the slot-based kernels: it exists for comparisons only)
  const bool tile_path = ctx->lds_atomics_ordered && ctx->tile_sorted && (M::STREAM == 1 || serial_seq);
  const unsigned T = tile_path ? TS_TILE : tile_size<M>();
  // segment length of the generic chain kernels: whole 1024-symbol blocks
  // Default segment: 4096 symbols; shorter for small blocks, which are chains of short,
  // latency-bound kernels (a lane of the walk/emit kernels walks one segment): 16 MiB blocks
  // +8 %, 4 MiB blocks +25 % with 1024-2048 (measured); the result never depends on it.
  const unsigned auto_S = n_sym >= (24u << 20) ? 4096u : n_sym >= (4u << 20) ? 2048u : 1024u;
  const unsigned S = (unsigned)min(((size_t)(ctx->seg_len ? ctx->seg_len : auto_S) + SETS_BLOCK - 1) / SETS_BLOCK * SETS_BLOCK, (size_t)1 << 30);
  const unsigned n_tiles = (n_sym + T - 1) / T;
  const unsigned n_groups = (n_tiles + GROUP_TILES - 1) / GROUP_TILES;
  const unsigned n_ptiles = (n_sym + PACK_TILE - 1) / PACK_TILE;
  const size_t padded = (size_t)n_sym + (size_t)CTX_PAD * B + 64;
  const unsigned max_items = (unsigned)((size_t)n_sym / ((size_t)S * 64) + B + 1);
  StreamResult *res = &b->result->s[M::STREAM];

  int rc;
  unsigned dbg_mask = fq_debug_skip();
  if ((dbg_mask & 256u) && M::STREAM == 1) dbg_mask = 0;  // 256: sequence stream only
  if ((dbg_mask & 512u) && M::STREAM == 0) dbg_mask = 0;  // 512: quality stream only
  bool dbg_off = false;
  // keys: ckey u16 | csym u8 (quality), later enc16 u16 over both;  slot_of: u32 -- padded by one batch
  const size_t n_pad = ((size_t)n_sym + SC_BATCH_SEQ + 15) & ~(size_t)15;  // keeps every sub-array 16-byte aligned
  static_assert(TILE_SEQ % SEQ_BATCH == 0 && SEQ_BATCH % PACK_TILE == 0 && TILE_QUAL % PACK_TILE == 0, "a packing tile lies inside one partition tile");
  if ((rc = sc.slot_of.reserve(n_pad * (tile_path ? 2 : 4)))) return rc;  // tile path: lpos16 (u16) lives here; slot path: slot_of (u32)
  if ((rc = sc.keys.reserve(n_pad * 3))) return rc;
  if ((rc = sc.sorted_sym.reserve(padded))) return rc;
  if ((rc = sc.out16.reserve(padded * 2))) return rc;
  if ((rc = sc.tile_hist.reserve((size_t)n_tiles * B * 4))) return rc;
  if ((rc = sc.tile_base.reserve((size_t)n_tiles * B * 4))) return rc;
  if ((rc = sc.group_sum.reserve((size_t)n_groups * B * 4))) return rc;
  if ((rc = sc.ctx_arrays.reserve((size_t)(4 * B + 3) * 4))) return rc;
  if ((rc = sc.seg_state.reserve((size_t)B * 2 + (size_t)B * 8))) return rc;
  // sequence chains: segment length of the candidate-set kernels
  // (sequence chains: a lane of k_seq_emit walks one segment, a pure latency chain -- blocks of 64 MiB
  // and less gain 4 % from 2048-symbol segments, 256 MiB blocks nothing)
  const unsigned auto_seq_S = n_sym >= (48u << 20) ? 4096u : n_sym >= (4u << 20) ? 2048u : 1024u;
  unsigned seq_S = ctx->seq_segment ? ctx->seq_segment : auto_seq_S;
  seq_S = (unsigned)min(((size_t)seq_S + SETS_BLOCK - 1) / SETS_BLOCK * SETS_BLOCK, (size_t)1 << 30);
  const unsigned seq_max_segs = n_sym / seq_S + B + 1;
  const unsigned seq_fstride = 1u << tab.max_log;
  // generic chain kernels (quality stream; sequence stream with FQGPU_CHAIN_SEQ_GENERIC)
  const unsigned gen_max_segs = n_sym / S + B + 1;
  const unsigned gen_fstride = 1u << tab.max_log;
  // segment tables of the generic chain kernels (SegArrays, then ItemArrays), every array 16-byte aligned
  auto al16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
  const size_t gs = gen_max_segs;
  const size_t sa_anchor = 0, sa_usym = al16(sa_anchor + gs * 4),
               sa_entry = al16(sa_usym + (size_t)B * 4), sa_cand = al16(sa_entry + gs * 2), sa_cls = al16(sa_cand + gs * SEG_MAX_CAND * 2),
               ia_has_g = al16(sa_cls + gs), ia_entry = al16(ia_has_g + (size_t)max_items * 4), ia_g = al16(ia_entry + (size_t)max_items * 2),
               seg_arrays_bytes = ia_g + (size_t)max_items * gen_fstride * 2 + 64;
  const unsigned gen_pbase = gen_max_segs;  // function slots: one per segment, then one power table per context
  if (!serial_seq) {
    if ((rc = sc.seg_arrays.reserve(seg_arrays_bytes))) return rc;
    if ((rc = sc.seq_fbuf.reserve(((size_t)gen_max_segs + B) * gen_fstride * 2 + 64))) return rc;
  }
  if (serial_seq) {
    if ((rc = sc.seq_plan.reserve((size_t)SEGPLAN_WORDS * 4 + (size_t)seq_max_segs * 2 + 64))) return rc;
    if ((rc = sc.seq_fbuf.reserve((size_t)seq_max_segs * seq_fstride * 2 + 64))) return rc;
    if ((rc = sc.seq_cbuf.reserve((size_t)(seq_max_segs / SEQ_ITEM_GROUPS + B + 1) * (seq_fstride * 2 + 2) + 64))) return rc;
  }
  if (serial_seq && (rc = sc.seq_bdesc.reserve((size_t)(n_ptiles + 1) * SeqModel::B * 6 + 64))) return rc;
  if ((rc = sc.tile_bits.reserve((size_t)n_ptiles * 4))) return rc;
  if ((rc = sc.tile_bit_base.reserve((size_t)(n_ptiles + 1) * 8))) return rc;
  // tile_sync: K6's look-back status u64 [tiles] | ticket counter, pad | run counts u32 [tiles] (16-byte aligned) | edge words uint4 [tiles]
  const size_t sync_counter_off = (size_t)n_tiles * 8, sync_runcount_off = sync_counter_off + 16;
  const size_t sync_edges_off = (sync_runcount_off + (size_t)n_tiles * 4 + 15) & ~(size_t)15;
  if (tile_path) {
    if ((rc = sc.tile_runs.reserve((size_t)n_tiles * ts_run_stride<M>() * sizeof(uint2)))) return rc;
    if ((rc = sc.tile_sync.reserve(sync_edges_off + (size_t)n_tiles * 16))) return rc;
  }

  if (reserve_only) return FQGPU_OK;  // (fqgpu_ctx_reserve: the scratch of a block of this shape exists now)

  uint16_t *ckey = sc.keys.as<uint16_t>();
  uint8_t *csym = reinterpret_cast<uint8_t *>(ckey + n_pad);
  uint16_t *enc16 = ckey;  // the keys are dead after K3
  static const bool dbg_no_alias = fq_debug_flag("FQGPU_DEBUG_NO_ALIAS");  // timing experiments that skip K1 or its stores
  if (dbg_no_alias) {
    if ((rc = sc.dbg_enc16.reserve(n_pad * 2))) return rc;
    enc16 = sc.dbg_enc16.as<uint16_t>();
  }
  uint32_t *arrays = sc.ctx_arrays.as<uint32_t>();
  uint16_t *final_state = sc.seg_state.as<uint16_t>();
  const unsigned lds_ct = (1u + (1u << (tab.max_log - 1)) + 2u * M::A) * 4u;
  const char *pfx = M::STREAM ? "qual." : "seq.";
  (void)pfx;

  FQ_SPAN_BEGIN(M::STREAM ? "qual.tile_hist" : "seq.tile_hist");  dbg_off = (dbg_mask & 1u) != 0 || fused_k1(ctx);
  if (!dbg_off) {
    if (tile_path) hipLaunchKernelGGL((k_tile_hist<M, uint16_t>), dim3(n_tiles), dim3(256), 0, st, b->raw, b->recs,
                                      rec_start, R, n_sym, T, sc.tile_hist.as<uint16_t>(), ckey, csym, res, fq_debug_k1());
    else hipLaunchKernelGGL((k_tile_hist<M, uint32_t>), dim3(n_tiles), dim3(256), 0, st, b->raw, b->recs,
                            rec_start, R, n_sym, T, sc.tile_hist.as<uint32_t>(), ckey, csym, res, fq_debug_k1());
  }
  FQ_SPAN_END();
  FQ_SPAN_BEGIN(M::STREAM ? "qual.layout" : "seq.layout");  dbg_off = (dbg_mask & 2u) != 0 || fq_debug_skipk(M::STREAM ? "k2q" : "k2s");
  if (!dbg_off) {
    if (tile_path) hipLaunchKernelGGL(k_group_sum<uint16_t>, dim3((B + 255) / 256, n_groups), dim3(256), 0, st,
                                      sc.tile_hist.as<uint16_t>(), n_tiles, B, sc.group_sum.as<uint32_t>());
    else hipLaunchKernelGGL(k_group_sum<uint32_t>, dim3((B + 255) / 256, n_groups), dim3(256), 0, st,
                            sc.tile_hist.as<uint32_t>(), n_tiles, B, sc.group_sum.as<uint32_t>());
  }
  if (!dbg_off) hipLaunchKernelGGL(k_group_prefix, dim3((B + 255) / 256), dim3(256), 0, st, sc.group_sum.as<uint32_t>(), n_groups, B, arrays);
  if (!dbg_off) hipLaunchKernelGGL(k_ctx_layout, dim3(1), dim3(1024), 0, st, B, S, arrays,
                     serial_seq ? nullptr : reinterpret_cast<uint32_t *>(sc.seg_arrays.as<uint8_t>() + sa_usym));
  if (!dbg_off) {
    if (tile_path) hipLaunchKernelGGL(k_tile_base<uint16_t>, dim3((B + 255) / 256, n_groups), dim3(256), 0, st,
                                      sc.tile_hist.as<uint16_t>(), sc.group_sum.as<uint32_t>(), arrays + B, n_tiles, B, sc.tile_base.as<uint32_t>());
    else hipLaunchKernelGGL(k_tile_base<uint32_t>, dim3((B + 255) / 256, n_groups), dim3(256), 0, st,
                            sc.tile_hist.as<uint32_t>(), sc.group_sum.as<uint32_t>(), arrays + B, n_tiles, B, sc.tile_base.as<uint32_t>());
  }
  FQ_SPAN_END();
  FQ_SPAN_BEGIN(M::STREAM ? "qual.scatter" : "seq.scatter");  dbg_off = (dbg_mask & 4u) != 0 || fq_debug_skipk(M::STREAM ? "k3q" : "k3s");
  SeqBatchDesc bd;
  bd.start = sc.seq_bdesc.as<uint32_t>();
  bd.pre = reinterpret_cast<uint16_t *>(bd.start + (size_t)(n_ptiles + 1) * SeqModel::B);
  uint16_t *lpos16 = reinterpret_cast<uint16_t *>(sc.slot_of.as<uint32_t>());  // the sequence stream has no slot_of
  uint8_t *sync_base = sc.tile_sync.as<uint8_t>();
  unsigned long long *ts_status = reinterpret_cast<unsigned long long *>(sync_base);
  unsigned *ts_counter = reinterpret_cast<unsigned *>(sync_base + sync_counter_off);
  uint32_t *ts_run_count = reinterpret_cast<uint32_t *>(sync_base + sync_runcount_off);
  uint4 *ts_edges = reinterpret_cast<uint4 *>(sync_base + sync_edges_off);
  if (tile_path) {
    if (!dbg_off)
    {
      // keys of the fused K1: contexts alone, the symbols derived from them (first symbols of the records from lane.first_sym)
      const uint8_t *first_sym = lane.first_sym.as<uint8_t>() + (M::STREAM ? R : 0u);
      if (fused_k1(ctx))
        hipLaunchKernelGGL((k_tile_partition<M, true>), dim3(n_tiles), dim3(TS_THREADS), 0, st, ckey, csym, n_sym, sc.tile_hist.as<uint16_t>(),
                           sc.tile_base.as<uint32_t>(), sc.sorted_sym.as<uint8_t>(), lpos16, sc.tile_runs.as<uint2>(), ts_run_count, ts_status, ts_counter,
                           rec_start, R, first_sym);
      else
        hipLaunchKernelGGL((k_tile_partition<M, false>), dim3(n_tiles), dim3(TS_THREADS), 0, st, ckey, csym, n_sym, sc.tile_hist.as<uint16_t>(),
                           sc.tile_base.as<uint32_t>(), sc.sorted_sym.as<uint8_t>(), lpos16, sc.tile_runs.as<uint2>(), ts_run_count, ts_status, ts_counter,
                           rec_start, R, (const uint8_t *)nullptr);
    }
  } else if (!dbg_off && serial_seq) {
    if (ctx->lds_atomics_ordered)
      hipLaunchKernelGGL(k_scatter_seq<true>, dim3(n_tiles), dim3(64), 0, st, ckey, n_sym, T, sc.tile_base.as<uint32_t>(),
                         sc.sorted_sym.as<uint8_t>(), lpos16, bd, fq_debug_no_sym(0));
    else
      hipLaunchKernelGGL(k_scatter_seq<false>, dim3(n_tiles), dim3(64), 0, st, ckey, n_sym, T, sc.tile_base.as<uint32_t>(),
                         sc.sorted_sym.as<uint8_t>(), lpos16, bd, fq_debug_no_sym(0));
  } else if (!dbg_off) {
    if (ctx->lds_atomics_ordered)
      hipLaunchKernelGGL((k_scatter<M, true>), dim3(n_tiles), dim3(64), 0, st, ckey, csym, n_sym, T,
                         sc.tile_base.as<uint32_t>(), sc.sorted_sym.as<uint8_t>(), sc.slot_of.as<uint32_t>(),
                         fq_debug_no_sym(M::STREAM));
    else
      hipLaunchKernelGGL((k_scatter<M, false>), dim3(n_tiles), dim3(64), 0, st, ckey, csym, n_sym, T,
                         sc.tile_base.as<uint32_t>(), sc.sorted_sym.as<uint8_t>(), sc.slot_of.as<uint32_t>(),
                         fq_debug_no_sym(M::STREAM));
  }
  FQ_SPAN_END();
  FQ_SPAN_BEGIN(M::STREAM ? "qual.scan" : (serial_seq ? "seq.plan" : "seq.scan"));  dbg_off = (dbg_mask & 8u) != 0;
  if (serial_seq) {
    uint32_t *plan = sc.seq_plan.as<uint32_t>();
    uint16_t *entry = reinterpret_cast<uint16_t *>(plan + SEGPLAN_WORDS);
    uint16_t *fbuf = sc.seq_fbuf.as<uint16_t>();
    const unsigned next_stride = 4u << tab.max_log;
    const bool two = tab.next2 != nullptr;  // two-symbol tables exist up to log 11
    const unsigned wpg = two ? SETS_WAVES2 : SETS_WAVES;
    const unsigned Q = min(max(ctx->seq_group, 1u), SETS_MAX_GROUP), gmin = max(ctx->seq_group_min, 1u);
    const unsigned rounds = max(SETS_ROUNDS / Q, 1u);
    const unsigned max_fitems = seq_max_segs / (wpg * rounds) + B + 1, max_eitems = seq_max_segs / 64 + B + 1;
    const unsigned max_citems = seq_max_segs / SEQ_ITEM_GROUPS + B + 1;
    uint16_t *cbuf = sc.seq_cbuf.as<uint16_t>(), *item_entry = cbuf + (size_t)max_citems * seq_fstride;
    const uint16_t *pow = nullptr;  // power tables for this segment length (uniform segments), if the handle has them
    for (unsigned i = 0; i < FQ_SEQ_POW_SETS; i++)
      if (tab.seq_pow[i] && tab.seq_pow_S[i] == seq_S) pow = tab.seq_pow[i];
    const bool dbg_skip = fq_debug_flag("FQGPU_DEBUG_SKIP_SEQ_CHAIN");  // timing experiment only: wrong output
    if (!dbg_off) hipLaunchKernelGGL(k_seq_segplan, dim3(1), dim3(256), 0, st, arrays, seq_S, Q, gmin, wpg * rounds, plan);
    FQ_SPAN_END();
    FQ_SPAN_BEGIN("seq.setfunc");
    {  // all of k_seq_setfunc's LDS is dynamic (118 KB with the two-symbol table): say so once per process
      static const bool raised = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_seq_setfunc<32, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_seq_setfunc<64, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
        return true;
      }();
      (void)raised;
    }
    if (!dbg_skip) {
      if (dbg_off || fq_debug_skipk("setfunc")) {
      } else if (two)
        hipLaunchKernelGGL((k_seq_setfunc<32, true>), dim3(min(max_fitems, ctx->setfunc_wgs ? ctx->setfunc_wgs : ctx->n_cus)), dim3(SETS_WAVES2 * 64),
                           (32u << tab.max_log) + SETS_WAVES2 * sizeof(SetsWaveLds11) + 16, st, sc.sorted_sym.as<uint8_t>(), arrays, plan, tab.logs, tab.next2,
                           4 * next_stride, pow, next_stride, seq_S, Q, gmin, rounds, seq_fstride, fbuf, plan + 5 * (B + 1), 32u << tab.max_log);
      else
        hipLaunchKernelGGL((k_seq_setfunc<64, false>), dim3(min(max_fitems, 2 * ctx->n_cus)), dim3(SETS_WAVES * 64),
                           (8u << tab.max_log) + SETS_WAVES * sizeof(SetsWaveLds) + 16, st, sc.sorted_sym.as<uint8_t>(), arrays, plan, tab.logs, tab.next1,
                           next_stride, pow, next_stride, seq_S, Q, gmin, rounds, seq_fstride, fbuf, plan + 5 * (B + 1), 8u << tab.max_log);
      FQ_SPAN_END();
      FQ_SPAN_BEGIN("seq.resolve");  dbg_off = (dbg_mask & 8u) != 0 || fq_debug_skipk("resolve");
      if (!dbg_off) {
        if (tab.max_log <= 11)
          hipLaunchKernelGGL(k_seq_resolve<32>, dim3(1), dim3(SEQ_RESOLVE_THREADS), 0, st, plan, tab.logs, fbuf, seq_fstride, Q, gmin, cbuf, item_entry, entry);
        else
          hipLaunchKernelGGL(k_seq_resolve<64>, dim3(1), dim3(SEQ_RESOLVE_THREADS), 0, st, plan, tab.logs, fbuf, seq_fstride, Q, gmin, cbuf, item_entry, entry);
      }
      FQ_SPAN_END();
      FQ_SPAN_BEGIN("seq.chains");  dbg_off = (dbg_mask & 8u) != 0 || fq_debug_skipk("emit");
      if (!dbg_off) hipLaunchKernelGGL(k_seq_emit, dim3(max_eitems), dim3(64), 8u << tab.max_log, st, sc.sorted_sym.as<uint8_t>(),
                         sc.out16.as<uint16_t>(), arrays, plan, tab.ct, tab.ct_off, tab.next1, next_stride, seq_S,
                         entry, final_state, res);
    }
  } else {
    uint8_t *sab = sc.seg_arrays.as<uint8_t>();
    SegArrays sa;
    sa.anchor = reinterpret_cast<uint32_t *>(sab + sa_anchor);
    sa.usym = reinterpret_cast<uint32_t *>(sab + sa_usym);
    sa.entry_state = reinterpret_cast<uint16_t *>(sab + sa_entry);
    sa.cand_exit = reinterpret_cast<uint16_t *>(sab + sa_cand);
    sa.cls = sab + sa_cls;
    if (M::STREAM == 1) { b->diag_cls = sa.cls; b->diag_n_segs = arrays + B + (B + 1) + B; }  // seg_base[B]
    uint16_t *fbuf = sc.seq_fbuf.as<uint16_t>();
    const unsigned cand_grid = min(gen_max_segs / (64 / SEG_SLOT) + 1, 32u * ctx->n_cus);
    if (!dbg_off && !fq_debug_skipk("scan")) hipLaunchKernelGGL(k_seg_scan<M>, dim3((gen_max_segs + 3) / 4), dim3(256), 0, st, sc.sorted_sym.as<uint8_t>(),
                       arrays, tab.reset_mask, tab.norm, tab.logs, S, sa);
    FQ_SPAN_END();
    FQ_SPAN_BEGIN(M::STREAM ? "qual.stage1" : "seq.stage1");  dbg_off = (dbg_mask & 8u) != 0 || fq_debug_skipk("stage1");
    if (dbg_off) {
    } else if (tab.max_log <= 11)
      hipLaunchKernelGGL((k_seg_stage1<M, 32>), dim3(max_items + cand_grid + gen_pbase + B), dim3(64), lds_ct, st, sc.sorted_sym.as<uint8_t>(),
                         sc.out16.as<uint16_t>(), arrays, tab.ct, tab.ct_off, final_state, S, max_items, cand_grid, gen_pbase, gen_fstride, sa, fbuf, res);
    else
      hipLaunchKernelGGL((k_seg_stage1<M, 64>), dim3(max_items + cand_grid + gen_pbase + B), dim3(64), lds_ct, st, sc.sorted_sym.as<uint8_t>(),
                         sc.out16.as<uint16_t>(), arrays, tab.ct, tab.ct_off, final_state, S, max_items, cand_grid, gen_pbase, gen_fstride, sa, fbuf, res);
    FQ_SPAN_END();
    FQ_SPAN_BEGIN(M::STREAM ? "qual.heads" : "seq.heads");  dbg_off = (dbg_mask & 8u) != 0 || fq_debug_skipk("heads");
    if (!dbg_off)
      hipLaunchKernelGGL(k_seg_heads<M>, dim3(cand_grid), dim3(64), lds_ct, st, sc.sorted_sym.as<uint8_t>(), arrays, tab.ct,
                         tab.ct_off, S, gen_fstride, sa, fbuf);
    FQ_SPAN_END();
    FQ_SPAN_BEGIN(M::STREAM ? "qual.resolve" : "seq.resolve");  dbg_off = (dbg_mask & 8u) != 0 || fq_debug_skipk("qresolve");
    if (!dbg_off) {
      ItemArrays ia;
      ia.has_g = reinterpret_cast<uint32_t *>(sab + ia_has_g);
      ia.item_entry = reinterpret_cast<uint16_t *>(sab + ia_entry);
      ia.g = reinterpret_cast<uint16_t *>(sab + ia_g);
      if (tab.max_log <= 11)
        hipLaunchKernelGGL((k_seg_compose<M, 32>), dim3(max_items), dim3(64), 0, st, arrays, tab.logs, fbuf, gen_pbase, gen_fstride, sa, ia);
      else
        hipLaunchKernelGGL((k_seg_compose<M, 64>), dim3(max_items), dim3(64), 0, st, arrays, tab.logs, fbuf, gen_pbase, gen_fstride, sa, ia);
      hipLaunchKernelGGL(k_seg_resolve2<M>, dim3((B + 255) / 256), dim3(256), 0, st, arrays, tab.logs, gen_fstride, sa, ia);
      hipLaunchKernelGGL(k_seg_resolve3<M>, dim3((max_items + 255) / 256), dim3(256), 0, st, arrays, tab.logs, fbuf,
                         gen_pbase, gen_fstride, sa, ia);
    }
    FQ_SPAN_END();
    FQ_SPAN_BEGIN(M::STREAM ? "qual.walk2" : "seq.walk2");  dbg_off = (dbg_mask & 8u) != 0 || fq_debug_skipk("walk2");
    if (!dbg_off) hipLaunchKernelGGL((k_seg_walk<M, 2>), dim3(max_items), dim3(64), lds_ct, st, sc.sorted_sym.as<uint8_t>(),
                       sc.out16.as<uint16_t>(), arrays, tab.ct, tab.ct_off, final_state, S, sa, res);
  }
  FQ_SPAN_END();
  if (tile_path) {
    // the stream starts from zeros (words shared by two tiles are OR-ed into), the look-back from clean flags
    FQ_SPAN_BEGIN(M::STREAM ? "qual.gatherpack" : "seq.gatherpack");  dbg_off = (dbg_mask & 16u) != 0 || fq_debug_skipk(M::STREAM ? "k6q" : "k6s");
    if (!dbg_off) {
      hipLaunchKernelGGL(k_tile_gather_pack<M>, dim3(n_tiles), dim3(TS_GP_THREADS), 0, st, lpos16, sc.tile_runs.as<uint2>(), ts_run_count,
                         sc.out16.as<uint16_t>(), n_sym, n_tiles, ts_status, ts_counter, tab.log_prefix, (unsigned long long)cap,
                         reinterpret_cast<uint32_t *>(out_dev), res, sc.tile_bit_base.as<unsigned long long>(), ts_edges);
    }
    FQ_SPAN_END();
  } else {
  FQ_SPAN_BEGIN(M::STREAM ? "qual.bitcount" : "seq.bitcount");  dbg_off = (dbg_mask & 16u) != 0;
  if (!dbg_off && serial_seq)
    hipLaunchKernelGGL(k_bitcount_seq, dim3((n_sym + SEQ_BATCH - 1) / SEQ_BATCH), dim3(PACK_THREADS), 0, st, lpos16, bd, sc.out16.as<uint16_t>(), n_sym,
                       sc.tile_bits.as<uint32_t>(), enc16);
  else if (!dbg_off)
    hipLaunchKernelGGL(k_bitcount, dim3(n_ptiles), dim3(PACK_THREADS), 0, st, sc.slot_of.as<uint32_t>(),
                       sc.out16.as<uint16_t>(), n_sym, sc.tile_bits.as<uint32_t>(), enc16);
  FQ_SPAN_END();
  FQ_SPAN_BEGIN(M::STREAM ? "qual.bitscan" : "seq.bitscan");  dbg_off = (dbg_mask & 32u) != 0;
  if (!dbg_off) hipLaunchKernelGGL(k_bitscan, dim3(1), dim3(1024), 0, st, sc.tile_bits.as<uint32_t>(), n_ptiles,
                     sc.tile_bit_base.as<unsigned long long>(), tab.log_prefix, B, (unsigned long long)cap,
                     reinterpret_cast<uint32_t *>(out_dev), res);
  FQ_SPAN_END();
  FQ_SPAN_BEGIN(M::STREAM ? "qual.pack" : "seq.pack");  dbg_off = (dbg_mask & 64u) != 0;
  if (!dbg_off) hipLaunchKernelGGL(k_pack, dim3(n_ptiles), dim3(PACK_THREADS), 0, st, enc16, n_sym,
                     sc.tile_bit_base.as<unsigned long long>(), reinterpret_cast<uint32_t *>(out_dev), res);
  FQ_SPAN_END();
  }
  FQ_SPAN_BEGIN(M::STREAM ? "qual.epilogue" : "seq.epilogue");  dbg_off = (dbg_mask & 128u) != 0;
  if (!dbg_off) hipLaunchKernelGGL(k_epilogue<M>, dim3(1), dim3(256), 0, st, arrays, final_state, tab.logs,
                     tab.log_prefix, reinterpret_cast<uint32_t *>(out_dev), res, tile_path ? ts_edges : (const uint4 *)nullptr, n_tiles);
  FQ_SPAN_END();
  b->index_bytes[M::STREAM] = 0;
  if ((flags & FQGPU_F_DECODE_INDEX) && !dbg_mask) {
    const unsigned stride = ctx->index_stride;
    const unsigned n_snap = n_sym ? (n_sym - 1) / stride : 0u;
    const size_t bytes = sizeof(FqIndexHeader) + (size_t)n_snap * fq_index_snap_bytes(B);
    if (bytes > b->index_cap[M::STREAM]) {
      if (b->index[M::STREAM]) FQ_HIP(hipFree(b->index[M::STREAM]));
      b->index[M::STREAM] = fq_dev_alloc<uint8_t>(bytes + 64);
      b->index_cap[M::STREAM] = b->index[M::STREAM] ? bytes : 0;
      if (!b->index[M::STREAM]) return FQGPU_E_NOMEM;
    }
    FQ_SPAN_BEGIN(M::STREAM ? "qual.index" : "seq.index");
    hipLaunchKernelGGL(k_index_meta<M>, dim3(n_snap / 256 + 1), dim3(256), 0, st, b->raw, b->recs, rec_start, R, n_sym,
                       stride, sc.tile_bit_base.as<unsigned long long>(), tile_path ? TS_TILE : PACK_TILE, b->index[M::STREAM]);
    if (n_snap) {
      const uint32_t *seg_prefix = serial_seq ? sc.seq_plan.as<uint32_t>() + 2 * (B + 1) : arrays + B + (B + 1);
      const uint16_t *entry = serial_seq ? reinterpret_cast<const uint16_t *>(sc.seq_plan.as<uint32_t>() + SEGPLAN_WORDS)
                                         : reinterpret_cast<const uint16_t *>(sc.seg_arrays.as<uint8_t>() + sa_entry);
      hipLaunchKernelGGL(k_index_states<M>, dim3(B, (n_snap + 63) / 64), dim3(64), lds_ct, st, sc.sorted_sym.as<uint8_t>(),
                         arrays, sc.tile_base.as<uint32_t>(), T, n_sym, stride, seg_prefix, entry, serial_seq ? 1 : 0,
                         serial_seq ? seq_S : S, tab.ct, tab.ct_off, final_state, b->index[M::STREAM]);
    }
    FQ_SPAN_END();
    b->index_bytes[M::STREAM] = bytes;
  }
  FQ_HIP(hipGetLastError());
  return FQGPU_OK;
}

}  // namespace

namespace {
// Verifies on the device that same-address LDS atomics of one wave instruction are applied in
// lane order (what k_scatter<ORDERED> relies on): random keys, some lanes idle, plain and packed
// counters, compared with the rank counted by shuffles.
__global__ void __launch_bounds__(256)
k_probe_lds_atomic_order(unsigned iters, unsigned *__restrict__ bad) {
  __shared__ unsigned cnt[4][1024];
  const unsigned wave = threadIdx.x >> 6, lane = fq_lane();
  unsigned s = (blockIdx.x * blockDim.x + threadIdx.x) * 40503u + 12345u;
  unsigned nbad = 0;
  for (unsigned it = 0; it < iters; it++) {
    for (unsigned c = lane; c < 1024; c += 64) cnt[wave][c] = 0;
    fq_lds_wave_sync();
    s ^= s << 13; s ^= s >> 17; s ^= s << 5;
    const unsigned range = 1u << ((it % 6) * 2);  // 1, 4, 16, ... 1024 distinct keys
    const unsigned key = (s >> 8) % range;
    const bool active = (s & 15u) != 0;
    const bool packed = (it & 1u) != 0;
    unsigned got = 0;
    if (active) {
      if (packed) got = (atomicAdd(&cnt[wave][key >> 1], 1u << (16 * (key & 1u))) >> (16 * (key & 1u))) & 0xFFFFu;
      else got = atomicAdd(&cnt[wave][key], 1u);
    }
    unsigned ref = 0;
    for (int l = 0; l < 64; l++) {
      const unsigned k2 = (unsigned)__shfl((int)key, l);
      const int a2 = __shfl((int)active, l);
      if ((unsigned)l < lane && a2 && k2 == key) ref++;
    }
    if (active && got != ref) nbad++;
    fq_lds_wave_sync();
  }
  if (nbad) atomicAdd(bad, nbad);
}

}  // namespace

int fq_probe_lds_atomic_order(hipStream_t st, bool *ordered) {
  unsigned *bad = fq_dev_alloc<unsigned>(1);
  if (!bad) return FQGPU_E_NOMEM;
  unsigned h = 1;
  hipError_t e = hipMemsetAsync(bad, 0, 4, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_probe_lds_atomic_order, dim3(512), dim3(256), 0, st, 96u, bad);
    e = hipMemcpyAsync(&h, bad, 4, hipMemcpyDeviceToHost, st);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(bad);
  if (e != hipSuccess) return FQGPU_E_HIP;
  *ordered = h == 0;
  return FQGPU_OK;
}

namespace {
// (context, symbol) counts from the sorted symbols: one wave per 4096 symbols of one context's run,
// a 64-bin histogram per wave in LDS (lanes that meet on the run's most frequent symbols are folded
// into one add first), one global add per non-empty bin.
__global__ void __launch_bounds__(256)
k_hist_sorted_qual(const uint8_t *__restrict__ sorted_sym, const uint32_t *__restrict__ arrays, unsigned S,
                   uint32_t *__restrict__ counts) {
  constexpr unsigned B = QualModel::B, A = QualModel::A;
  __shared__ uint32_t s_bins[4][A];
  const uint32_t *ctx_count = arrays, *ctx_start = arrays + B, *seg_base = ctx_start + B + 1;
  const unsigned wave = threadIdx.x >> 6, lane = fq_lane(), seg = blockIdx.x * 4 + wave;
  if (seg >= seg_base[B]) return;  // the grid is an upper bound
  const unsigned c = seg_ctx_of<QualModel>(seg_base, seg), k = seg - seg_base[c];
  const unsigned n = ctx_count[c], begin = k * S, end = min(n, begin + S);
  const uint8_t *sym = sorted_sym + ctx_start[c];
  uint32_t *bins = s_bins[wave];
  bins[lane] = 0;
  fq_lds_wave_sync();
  for (unsigned p0 = begin; p0 < end; p0 += 1024) {
    const unsigned p = p0 + 16 * lane;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (p < end) v = *reinterpret_cast<const uint4 *>(sym + p);  // (runs are padded to 16 bytes)
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const unsigned s = (w[j >> 2] >> (8 * (j & 3))) & (A - 1);
      bool on = p + j < end;
      // the symbol of the first live lane: everybody who has it too is counted by that lane
      const unsigned long long live = __ballot(on);
      if (!live) break;  // (uniform)
      const unsigned lead = (unsigned)__ffsll((long long)live) - 1u;
      const unsigned s_lead = (unsigned)__shfl((int)s, (int)lead);
      const unsigned long long same = __ballot(on && s == s_lead);
      if (lane == lead) atomicAdd(&bins[s_lead], (unsigned)__popcll(same));
      if (on && s != s_lead) atomicAdd(&bins[s], 1u);
    }
  }
  fq_lds_wave_sync();
  if (bins[lane]) atomicAdd(&counts[(size_t)c * A + lane], bins[lane]);
}
}  // namespace

// FSE_Quality::calculateFreqTable (reference src/fse_quality.cpp:69-97) at the encoder's speed: the
// (context, symbol) pairs of the sample are the ones the encoder's K1 computes (the context of a
// position is the same from either end of the read), so K1 + K2 + K3 sort the symbols by context and
// the counts are small local histograms of contiguous runs -- instead of 60 M atomics scattered
// over a 2 MiB table in HBM (9 ms for a 128 MiB sample, 250 ms when one context holds most of it).
// counts_dev: [8192][64], already filled with the reference's initial 1.  Reads of length >= 3 only.
// Waits for st before it returns (its scratch is local).
int fq_qual_counts_sorted(hipStream_t st, const uint8_t *raw_dev, const fqgpu_rec *recs_dev, size_t n_recs, size_t n_bases,
                          uint32_t *counts_dev, uint32_t *err_dev) {
  constexpr unsigned B = QualModel::B, S = 4096;
  const unsigned R = (unsigned)n_recs, n_sym = (unsigned)n_bases;
  const unsigned n_tiles = (n_sym + TS_TILE - 1) / TS_TILE, n_groups = (n_tiles + GROUP_TILES - 1) / GROUP_TILES;
  const size_t n_pad = ((size_t)n_sym + SC_BATCH_SEQ + 15) & ~(size_t)15, padded = (size_t)n_sym + (size_t)CTX_PAD * B + 64;
  const unsigned max_segs = n_sym / S + B + 1;
  EncLane lane;
  EncScratch &sq = lane.enc[1], &ss = lane.enc[0];
  DevBuf readlens, result;
  int rc = FQGPU_OK;
  do {
    if ((rc = lane.rec_start.reserve((size_t)(R + 1) * 4)) || (rc = lane.n_cnt32.reserve((size_t)R * 8)) ||
        (rc = readlens.reserve((size_t)R * 2)) || (rc = result.reserve(sizeof(BlockResult))) || (rc = lane.first_sym.reserve((size_t)R * 2 + 16)) ||
        (rc = reserve_k1<SeqModel>(ss, n_sym)) || (rc = reserve_k1<QualModel>(sq, n_sym)) ||
        (rc = sq.slot_of.reserve(n_pad * 2)) || (rc = sq.sorted_sym.reserve(padded)) || (rc = sq.tile_base.reserve((size_t)n_tiles * B * 4)) ||
        (rc = sq.group_sum.reserve((size_t)n_groups * B * 4)) || (rc = sq.ctx_arrays.reserve((size_t)(4 * B + 3) * 4)) ||
        (rc = sq.tile_runs.reserve((size_t)n_tiles * ts_run_stride<QualModel>() * sizeof(uint2))) || (rc = sq.tile_sync.reserve((size_t)n_tiles * 4 + 64)))
      break;
    uint32_t *n_cnt32 = lane.n_cnt32.as<uint32_t>(), *lens32 = n_cnt32 + R, *rec_start = lane.rec_start.as<uint32_t>();
    BlockResult *res = result.as<BlockResult>();
    uint32_t *arrays = sq.ctx_arrays.as<uint32_t>();
    uint16_t *kq = sq.keys.as<uint16_t>();
    if (hipMemsetAsync(res, 0, sizeof(BlockResult), st) != hipSuccess) { rc = FQGPU_E_HIP; break; }
    hipLaunchKernelGGL(k_readlens, dim3((unsigned)min((size_t)(R + 255) / 256, (size_t)2048)), dim3(256), 0, st, recs_dev, R,
                       readlens.as<uint16_t>(), n_cnt32, lens32, (BlockResult *)nullptr);
    if ((rc = fq_scan_u32_to_u32(st, lens32, R, rec_start, lane.scan_tmp))) break;
    hipLaunchKernelGGL(k_tile_hist2, dim3(n_tiles), dim3(256), 0, st, raw_dev, recs_dev, rec_start, R, n_sym, TS_TILE,
                       ss.tile_hist.as<uint16_t>(), ss.keys.as<uint8_t>(), sq.tile_hist.as<uint16_t>(), kq,
                       lane.first_sym.as<uint8_t>(), lane.first_sym.as<uint8_t>() + R, n_cnt32, res);
    hipLaunchKernelGGL(k_group_sum<uint16_t>, dim3((B + 255) / 256, n_groups), dim3(256), 0, st, sq.tile_hist.as<uint16_t>(), n_tiles, B, sq.group_sum.as<uint32_t>());
    hipLaunchKernelGGL(k_group_prefix, dim3((B + 255) / 256), dim3(256), 0, st, sq.group_sum.as<uint32_t>(), n_groups, B, arrays);
    hipLaunchKernelGGL(k_ctx_layout, dim3(1), dim3(1024), 0, st, B, S, arrays, (uint32_t *)nullptr);
    hipLaunchKernelGGL(k_tile_base<uint16_t>, dim3((B + 255) / 256, n_groups), dim3(256), 0, st, sq.tile_hist.as<uint16_t>(), sq.group_sum.as<uint32_t>(),
                       arrays + B, n_tiles, B, sq.tile_base.as<uint32_t>());
    // (the rank only has to be a permutation here, not a stable one: no lane-ordered atomics are relied on)
    hipLaunchKernelGGL((k_tile_partition<QualModel, true>), dim3(n_tiles), dim3(TS_THREADS), 0, st, kq, (const uint8_t *)nullptr, n_sym,
                       sq.tile_hist.as<uint16_t>(), sq.tile_base.as<uint32_t>(), sq.sorted_sym.as<uint8_t>(),
                       reinterpret_cast<uint16_t *>(sq.slot_of.as<uint32_t>()), sq.tile_runs.as<uint2>(), sq.tile_sync.as<uint32_t>(),
                       (unsigned long long *)nullptr, (unsigned *)nullptr, rec_start, R, lane.first_sym.as<uint8_t>() + R);
    hipLaunchKernelGGL(k_hist_sorted_qual, dim3((max_segs + 3) / 4), dim3(256), 0, st, sq.sorted_sym.as<uint8_t>(), arrays, S, counts_dev);
    BlockResult h;
    if (hipMemcpyAsync(&h, res, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess ||
        hipGetLastError() != hipSuccess) { rc = FQGPU_E_HIP; break; }
    if (h.s[0].bad_symbol || h.s[1].bad_symbol) {  // a quality above Q63 / a byte that is no base: the caller reads the flag
      uint32_t one = 1;
      if (hipMemcpy(err_dev, &one, 4, hipMemcpyHostToDevice) != hipSuccess) rc = FQGPU_E_HIP;
    }
  } while (0);
  (void)hipStreamSynchronize(st);
  DevBuf *own[] = {&readlens, &result, &lane.rec_start, &lane.n_cnt32, &lane.n_off, &lane.scan_tmp, &lane.first_sym};
  for (DevBuf *b : own) b->release();
  for (EncScratch *e : {&ss, &sq}) {
    DevBuf *eb[] = {&e->slot_of, &e->keys, &e->sorted_sym, &e->out16, &e->tile_hist, &e->tile_base, &e->group_sum, &e->ctx_arrays,
                    &e->tile_runs, &e->tile_sync, &e->scan_tmp};
    for (DevBuf *b : eb) b->release();
  }
  return rc;
}

// One block = one encode lane: two HIP streams (sequence pipeline, quality pipeline) forked
// after the record-level kernels and joined before the N-position pass.  Blocks handed to
// different lanes overlap on the device: the serial sequence chains of one block hide behind
// the bandwidth-bound passes of the others.
int fq_encode_launch(fqgpu_ctx *ctx, fqgpu_dblock *b, unsigned flags, hipEvent_t wait, hipStream_t *done, bool reserve_only) {
  const unsigned R = (unsigned)b->n_recs;
  if (R == 0 || b->n_bases == 0) return FQGPU_E_ARG;
  EncLane *lp = fq_next_lane(ctx, b->n_bases, reserve_only ? nullptr : b);  // (reserving walks the lanes in turn)
  if (!lp) return FQGPU_E_NOMEM;
  EncLane &lane = *lp;
  hipStream_t st = lane.st_seq;
  if (done) *done = st;
  int rc;
  if (wait) FQ_HIP(hipStreamWaitEvent(st, wait, 0));
  if (!reserve_only && b->ev_encoded) FQ_HIP(hipStreamWaitEvent(st, b->ev_encoded, 0));  // the block's previous encode (fqgpu_internal.h)
  if ((rc = lane.rec_start.reserve((size_t)(R + 1) * 4))) return rc;
  if ((rc = lane.n_cnt32.reserve((size_t)R * 4 * 2))) return rc;  // n_cnt32 | lens32
  if ((rc = lane.n_off.reserve((size_t)(R + 1) * 4))) return rc;
  if ((rc = lane.first_sym.reserve((size_t)R * 2 + 16))) return rc;
  uint32_t *n_cnt32 = lane.n_cnt32.as<uint32_t>();
  uint32_t *lens32 = n_cnt32 + R;
  uint32_t *rec_start = lane.rec_start.as<uint32_t>();
  if (reserve_only) {  // every allocation a block of this shape needs on this lane, nothing launched
    const unsigned n_sym = (unsigned)b->n_bases;
    if (fused_k1(ctx) && ((rc = reserve_k1<SeqModel>(lane.enc[0], n_sym)) || (rc = reserve_k1<QualModel>(lane.enc[1], n_sym)))) return rc;
    if ((rc = lane.scan_tmp.reserve(((size_t)R / 1024 + 64) * 16))) return rc;
    if ((rc = encode_stream<QualModel>(ctx, lane, lane.st_qual, b, rec_start, nullptr, 0, flags, true))) return rc;
    return encode_stream<SeqModel>(ctx, lane, lane.st_seq, b, rec_start, nullptr, 0, flags, true);
  }

  const unsigned rec_blocks = (unsigned)min((size_t)(R + 3) / 4, (size_t)8192);  // k_npos: a wave per record
  const unsigned len_blocks = (unsigned)min((size_t)(R + 15) / 16, (size_t)4096);  // four records per wave
  const bool fused = fused_k1(ctx);
  FQ_SPAN_BEGIN("records");
  // k_record_scan's look-back words: zero when (re)allocated, then epochs and tickets count on from launch to launch
  const unsigned rscan_chunks = (R + RSCAN_CHUNK - 1) / RSCAN_CHUNK;
  auto rscan_launch = [&](int mode) -> int {
    const size_t need = (size_t)rscan_chunks * 8 + 16;
    if (need > lane.rscan.cap) {
      int rr = lane.rscan.reserve(need);
      if (rr) return rr;
      FQ_HIP(hipMemsetAsync(lane.rscan.p, 0, lane.rscan.cap, st));
      lane.rscan_tickets = 0;
    }
    lane.rscan_epoch = (lane.rscan_epoch + 1u) & 0xFFFFFFu;
    if (lane.rscan_epoch == 0u) {  // (every 16 M launches: words of the previous round of epochs must not be taken for new ones)
      FQ_HIP(hipMemsetAsync(lane.rscan.p, 0, lane.rscan.cap, st));
      lane.rscan_tickets = 0;
      lane.rscan_epoch = 1u;
    }
    unsigned *ticket = lane.rscan.as<unsigned>();
    unsigned long long *status = reinterpret_cast<unsigned long long *>(lane.rscan.as<uint8_t>() + 16);
    if (mode == 0)
      hipLaunchKernelGGL(k_record_scan<0>, dim3(rscan_chunks), dim3(RSCAN_THREADS), 0, st, b->recs, n_cnt32, R, b->readlens, rec_start, status, ticket,
                         lane.rscan_tickets, lane.rscan_epoch, b->result);
    else
      hipLaunchKernelGGL(k_record_scan<1>, dim3(rscan_chunks), dim3(RSCAN_THREADS), 0, st, b->recs, n_cnt32, R, b->n_count, lane.n_off.as<uint32_t>(), status, ticket,
                         lane.rscan_tickets, lane.rscan_epoch, b->result);
    lane.rscan_tickets += rscan_chunks;
    return FQGPU_OK;
  };
  if (fused) {  // lengths from the record table alone, scanned in the same launch; K1 counts the N's (the raw block is read once less)
    if ((rc = rscan_launch(0))) return rc;
  } else {
    hipLaunchKernelGGL(k_readlens_ncount, dim3(len_blocks), dim3(256), 0, st, b->raw, b->recs, R,
                       b->readlens, b->n_count, n_cnt32, lens32, b->result);
    if ((rc = fq_scan2_u32_to_u32(st, lens32, n_cnt32, R, rec_start, lane.n_off.as<uint32_t>(), lane.scan_tmp))) return rc;
    hipLaunchKernelGGL(k_store_npos_len, dim3(1), dim3(1), 0, st, lane.n_off.as<uint32_t>(), R, b->result);
  }
  FQ_SPAN_END();

  if (fused) {  // K1 of both streams in one pass, in front of the fork
    const unsigned n_sym = (unsigned)b->n_bases, n_tiles = (n_sym + TS_TILE - 1) / TS_TILE;
    if ((rc = reserve_k1<SeqModel>(lane.enc[0], n_sym)) || (rc = reserve_k1<QualModel>(lane.enc[1], n_sym))) return rc;
    uint16_t *kq = lane.enc[1].keys.as<uint16_t>();
    FQ_SPAN_BEGIN("tile_hist2");
    if (!fq_debug_skipk("k1")) hipLaunchKernelGGL(k_tile_hist2, dim3(n_tiles), dim3(256), 0, st, b->raw, b->recs, rec_start, R, n_sym, TS_TILE,
                       lane.enc[0].tile_hist.as<uint16_t>(), lane.enc[0].keys.as<uint8_t>(), lane.enc[1].tile_hist.as<uint16_t>(), kq,
                       lane.first_sym.as<uint8_t>(), lane.first_sym.as<uint8_t>() + R, n_cnt32, b->result);
    FQ_SPAN_END();
  }
  FQ_HIP(hipEventRecord(lane.ev_fork, lane.st_seq));
  FQ_HIP(hipStreamWaitEvent(lane.st_qual, lane.ev_fork, 0));
  // timing experiments only (wrong output): one stream at a time
  const bool dbg_no_qual = fq_debug_flag("FQGPU_DEBUG_SKIP_QUAL"), dbg_no_seq = fq_debug_flag("FQGPU_DEBUG_SKIP_SEQ");
  if (!dbg_no_qual && (rc = encode_stream<QualModel>(ctx, lane, lane.st_qual, b, rec_start, b->qual, b->qual_cap, flags))) return rc;
  if (!dbg_no_seq && (rc = encode_stream<SeqModel>(ctx, lane, lane.st_seq, b, rec_start, b->seq, b->seq_cap, flags))) return rc;
  FQ_HIP(hipEventRecord(lane.ev_join, lane.st_qual));
  FQ_HIP(hipStreamWaitEvent(lane.st_seq, lane.ev_join, 0));

  // after both streams: the optional in-place N -> A must not race with their reads of raw
  FQ_SPAN_BEGIN("npos");
  if (fused) {  // K1 left the N counts: 16-bit copy for the caller, offsets of the position deltas, their total -- one launch
    if ((rc = rscan_launch(1))) return rc;
  }
  hipLaunchKernelGGL(k_npos, dim3(rec_blocks), dim3(256), 0, st, b->raw, b->recs, R,
                     lane.n_off.as<uint32_t>(), b->n_pos, (flags & FQGPU_F_WRITE_BACK_N) ? 1 : 0);
  FQ_SPAN_END();
  FQ_HIP(hipGetLastError());
  if (!b->ev_encoded) FQ_HIP(hipEventCreateWithFlags(&b->ev_encoded, hipEventDisableTiming));
  FQ_HIP(hipEventRecord(b->ev_encoded, st));
  return FQGPU_OK;
}
