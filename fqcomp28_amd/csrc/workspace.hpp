// workspace.hpp -- the reference's block-codec surface on top of the fqgpu C ABI.
//
// Mirrors, name for name, what the reference exposes for this path so that its pipeline
// (src/process.cpp:46-68, 93-104) and its tests read the same:
//   FastqRecord / FastqChunk                       src/defs.h:22-52
//   CompressedBuffersDst / CompressedBuffersSrc    src/compressed_buffers.h:34-101
//   DatasetMeta (ft_seq / ft_qual part)            src/prepare.h:14-57
//   CompressionWorkspace::encodeChunk              src/workspace.h:69,  src/workspace.cpp:14-45
//   DecompressionWorkspace::decodeChunk            src/workspace.h:112, src/workspace.cpp:47-88
//   Workspace::compressBoundSequence/Quality       src/workspace.h:21-35
//   FSE_{Sequence,Quality}::calculateFreqTable     src/fse_sequence.h:72, src/fse_quality.h:50
//   CompressionWorkspace::encodeHeader / DecompressionWorkspace::decodeHeader   src/workspace.cpp:95-157
// The seq/qual FSE streams are coded on the GPU; the header fields are tokenised and delta-coded
// on the host (headers.hpp).  The misc streams (readlens, n_count, n_pos,
// header fields) go through compressMiscBuffers / decompressMiscBuffers like in the reference
// (src/workspace.cpp:176-256) -- with the library's own coder (fq_misc.cpp) in place of libbsc, whose
// source is absent: the compressed misc BYTES are out of parity scope, everything in front of that
// pass (the streams themselves, cbs.original_size) is the reference's, byte for byte.
// Error behaviour:
// the reference asserts / silently returns size 0; this shim throws std::runtime_error with
// fqgpu_strerror().  Header-only; link with libfqgpu.so.
#pragma once

#include <chrono>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <istream>
#include <memory>
#include <new>
#include <ostream>
#include <stdexcept>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

#include "../../include/fqgpu.h"
#include "headers.hpp"

namespace fqcomp28 {

using readlen_t = uint16_t;  // src/defs.h:14

/** Allocator of the buffers that cross PCIe: page-locked host memory (fqgpu_host_alloc), so that
 *  the copies to and from the GPU run at link rate and asynchronously.  The reference's
 *  FastqData / std::vector<std::byte> with another allocator: same interface, same contents. */
template <class T> struct HostAllocator {
  using value_type = T;
  HostAllocator() = default;
  template <class U> HostAllocator(const HostAllocator<U> &) {}
  T *allocate(std::size_t n) {
    void *p = fqgpu_host_alloc(n * sizeof(T));
    if (!p) throw std::bad_alloc();
    return static_cast<T *>(p);
  }
  void deallocate(T *p, std::size_t) { fqgpu_host_free(p); }
  /** resize() of a chunk or stream buffer must not sweep hundreds of megabytes with zeros that the
   *  file read / the copy from the GPU overwrites the next moment: elements are default-initialised */
  template <class U> void construct(U *p) { ::new (static_cast<void *>(p)) U; }
  template <class U, class A0, class... A> void construct(U *p, A0 &&a0, A &&...a) {
    ::new (static_cast<void *>(p)) U(std::forward<A0>(a0), std::forward<A>(a)...);
  }
  template <class U> bool operator==(const HostAllocator<U> &) const { return true; }
  template <class U> bool operator!=(const HostAllocator<U> &) const { return false; }
};
using FastqData = std::vector<char, HostAllocator<char>>;            // src/defs.h:20
using stream_bytes_t = std::vector<std::byte, HostAllocator<std::byte>>;  // cbs.seq / cbs.qual
// The side buffers (record table, readlens, n_count, n_pos) stay PAGEABLE on purpose: page-locked, their
// small copies queue up in the DMA engines behind the other workers' block uploads (fqgpu_encode_block
// from four threads: 47.9 GB/s with pageable side buffers, 40-42 with page-locked ones, same box).
using RecordTable = std::vector<fqgpu_rec>;
using u16_buffer_t = std::vector<uint16_t>;

/** Non-owning - holds pointers into outside allocated data (src/defs.h:22-32) */
struct FastqRecord {
  char *seqp = nullptr, *qualp = nullptr, *headerp = nullptr;
  readlen_t length = 0, header_length = 0;
  [[nodiscard]] std::string_view header() const { return {headerp, header_length}; }
  [[nodiscard]] std::string_view seq() const { return {seqp, length}; }
  [[nodiscard]] std::string_view qual() const { return {qualp, length}; }
};

struct FastqChunk {  // src/defs.h:34-52
  FastqData raw_data;
  std::vector<FastqRecord> records;
  std::size_t tot_reads_length = 0;
  std::size_t headers_length = 0;
  unsigned idx = 0;
  void clear() {
    idx = 0; tot_reads_length = 0; headers_length = 0;
    raw_data.clear(); records.clear();
  }
};

struct cb_original_sizes_t {  // src/compressed_buffers.h:10-32
  std::vector<headers::FieldStorage::sizes> header_fields;
  uint32_t total = 0, readlens = 0, n_records = 0, n_count = 0, n_pos = 0;
  void clear() {
    total = readlens = n_records = n_count = n_pos = 0;
    for (auto &sz : header_fields) sz = {};
  }
};

struct CompressedBuffers {  // src/compressed_buffers.h:34-69
  stream_bytes_t seq, qual;
  std::vector<std::byte> readlens, compressed_readlens;
  std::vector<headers::CompressedFieldStorage> compressed_header_fields;
  std::vector<std::byte> n_count, compressed_n_count;
  std::vector<std::byte> n_pos, compressed_n_pos;
  cb_original_sizes_t original_size;
  uint32_t chunk_idx = 0;
  /** Extension, not in the reference and not in the .fqc container: the decode index of the sequence / quality
   *  stream (fqgpu_encode_index), filled by encodeChunk when the workspace was asked for it, used by decodeChunk
   *  when present.  The farm keeps it in a file beside the archive (archive.hpp: DecodeIndexFile). */
  std::vector<std::byte> decode_index[2];
  /* like the reference, clear() does NOT clear n_count / n_pos */
  virtual void clear() {
    seq.clear(); qual.clear();
    decode_index[0].clear(); decode_index[1].clear();
    readlens.clear(); compressed_readlens.clear();
    for (auto &chf : compressed_header_fields) chf.clear();
    original_size.clear();
  }
  virtual ~CompressedBuffers() = default;
};

/** memcompress / memdecompress (src/memcompress.h:5-28) over the library's own misc-stream coder
 *  (fq_misc.cpp; libbsc's bytes are out of parity scope) */
inline std::size_t memcompress(std::byte *dst, const std::byte *src, std::size_t src_size) {
  return fqgpu_memcompress(reinterpret_cast<uint8_t *>(dst), fqgpu_memcompress_bound(src_size),
                           reinterpret_cast<const uint8_t *>(src), src_size);
}
inline std::size_t memdecompress(std::byte *dst, std::size_t dst_size, const std::byte *src, std::size_t src_size) {
  const std::size_t n = fqgpu_memdecompress(reinterpret_cast<uint8_t *>(dst), dst_size,
                                            reinterpret_cast<const uint8_t *>(src), src_size);
  if (n == static_cast<std::size_t>(-1) || (src_size && n != dst_size)) throw std::runtime_error("memdecompress: malformed misc stream");
  return n;
}
/** Every misc stream of a block (everything but the two FSE streams) as f(plain, compressed,
 *  original size): the one list both directions of the misc pass walk (reference
 *  src/workspace.cpp:176-256 runs libbsc over the same streams).  A NUMERIC header field has a
 *  content stream only. */
template <class Buffers, class F> void miscStreams(Buffers &cbs, const headers::HeaderFormatSpeciciation &fmt, F &&f) {
  f(cbs.readlens, cbs.compressed_readlens, cbs.original_size.readlens);
  f(cbs.n_count, cbs.compressed_n_count, cbs.original_size.n_count);
  f(cbs.n_pos, cbs.compressed_n_pos, cbs.original_size.n_pos);
  for (std::size_t i = 0; i < fmt.n_fields(); ++i) {
    auto &plain = cbs.header_fields[i];
    auto &packed = cbs.compressed_header_fields[i];
    auto &sizes = cbs.original_size.header_fields[i];
    f(plain.content, packed.content, sizes.content);
    if (fmt.field_types[i] == headers::FieldType::STRING) {
      f(plain.isDifferentFlag, packed.isDifferentFlag, sizes.isDifferentFlag);
      f(plain.contentLength, packed.contentLength, sizes.contentLength);
    }
  }
}

struct CompressedBuffersDst : CompressedBuffers {
  std::vector<headers::FieldStorageDst> header_fields;
  void clear() override {
    CompressedBuffers::clear();
    for (auto &hf : header_fields) hf.clear();
  }
};
struct CompressedBuffersSrc : CompressedBuffers {
  std::vector<headers::FieldStorageSrc> header_fields;
  struct { std::size_t n_count = 0, n_pos = 0; } index;  // src/compressed_buffers.h:90-93
  void clear() override {
    CompressedBuffers::clear();
    for (auto &hf : header_fields) hf.clear();
    index = {};
  }
};

/** FQGPU_SHIM_TRACE=1: one line per block on stderr with the milliseconds of every stage of
 *  encodeChunk (where a worker's time goes: DESIGN.md section 9) */
struct StageClock {
  bool on = std::getenv("FQGPU_SHIM_TRACE") != nullptr;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  std::string line;
  void lap(const char *name) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    char buf[64];
    std::snprintf(buf, sizeof(buf), " %s %.1f", name, std::chrono::duration<double, std::milli>(now - t).count());
    line += buf;
    t = now;
  }
  void done(unsigned idx) { if (on) std::fprintf(stderr, "block %u:%s\n", idx, line.c_str()); }
};

inline void fqgpuCheck(int rc, const char *what) {
  if (rc != FQGPU_OK) throw std::runtime_error(std::string(what) + ": " + fqgpu_strerror(rc));
}

/** FreqTable PODs exactly as the archive stores them (src/fse_common.hpp:147-174) */
struct DatasetMeta {
  /** used by the host's header coder for delta-ing the first header of each chunk (src/prepare.h:30) */
  std::string first_header;
  headers::HeaderFormatSpeciciation header_fmt;  // src/prepare.h:31
  std::unique_ptr<std::byte[]> ft_seq{new std::byte[FQGPU_SEQ_FT_BYTES]};
  std::unique_ptr<std::byte[]> ft_qual{new std::byte[FQGPU_QUAL_FT_BYTES]};
  DatasetMeta() = default;
  explicit DatasetMeta(std::string_view header)
      : first_header(header), header_fmt(headers::HeaderFormatSpeciciation::fromHeader(first_header)) {}
  /** DatasetMeta(const FastqChunk&) (src/prepare.h:23-27): dataset analysis on the GPU */
  explicit DatasetMeta(const FastqChunk &chunk, int device = 0)
      : first_header(chunk.records.empty() ? std::string_view() : chunk.records.front().header()) {
    if (!first_header.empty()) header_fmt = headers::HeaderFormatSpeciciation::fromHeader(first_header);
    RecordTable recs = toRecordTable(chunk);
    fqgpuCheck(fqgpu_freq_tables(device, reinterpret_cast<const uint8_t *>(chunk.raw_data.data()),
                                 chunk.raw_data.size(), recs.data(), recs.size(), ft_seq.get(),
                                 ft_qual.get(), nullptr, nullptr),
               "calculateFreqTable");
  }
  /** bytes of the metadata section of an archive (src/prepare.h:44-52) */
  [[nodiscard]] std::size_t size() const { return sizeof(readlen_t) + first_header.length() + FQGPU_SEQ_FT_BYTES + FQGPU_QUAL_FT_BYTES; }

  /** The metadata section of an archive (src/prepare.cpp:12-21 writes the same bytes): u16 header
   *  length, the header, then the two FreqTable PODs as they lie in memory */
  void appendTo(std::vector<uint8_t> &out) const {
    if (first_header.size() > 0xFFFFu) throw std::invalid_argument("first header longer than a readlen_t");
    const readlen_t hlen = static_cast<readlen_t>(first_header.size());
    const std::size_t at = out.size();
    out.resize(at + size());
    uint8_t *p = out.data() + at;
    std::memcpy(p, &hlen, sizeof(hlen));
    std::memcpy(p + sizeof(hlen), first_header.data(), hlen);
    std::memcpy(p + sizeof(hlen) + hlen, ft_seq.get(), FQGPU_SEQ_FT_BYTES);
    std::memcpy(p + sizeof(hlen) + hlen + FQGPU_SEQ_FT_BYTES, ft_qual.get(), FQGPU_QUAL_FT_BYTES);
  }
  static DatasetMeta fromBytes(const uint8_t *p, std::size_t n) {
    readlen_t hlen = 0;
    if (n < sizeof(hlen)) throw std::runtime_error("truncated dataset metadata");
    std::memcpy(&hlen, p, sizeof(hlen));
    if (n < sizeof(hlen) + hlen + FQGPU_SEQ_FT_BYTES + FQGPU_QUAL_FT_BYTES) throw std::runtime_error("truncated dataset metadata");
    DatasetMeta meta{std::string_view(reinterpret_cast<const char *>(p) + sizeof(hlen), hlen)};
    std::memcpy(meta.ft_seq.get(), p + sizeof(hlen) + hlen, FQGPU_SEQ_FT_BYTES);
    std::memcpy(meta.ft_qual.get(), p + sizeof(hlen) + hlen + FQGPU_SEQ_FT_BYTES, FQGPU_QUAL_FT_BYTES);
    return meta;
  }
  /** the same through streams, under the reference's names (src/prepare.cpp:12-41) */
  static void storeToStream(const DatasetMeta &meta, std::ostream &os) {
    std::vector<uint8_t> bytes;
    meta.appendTo(bytes);
    os.write(reinterpret_cast<const char *>(bytes.data()), static_cast<std::streamsize>(bytes.size()));
  }
  static DatasetMeta loadFromStream(std::istream &is) {
    std::vector<uint8_t> bytes(sizeof(readlen_t));
    is.read(reinterpret_cast<char *>(bytes.data()), sizeof(readlen_t));
    readlen_t hlen = 0;
    std::memcpy(&hlen, bytes.data(), sizeof(hlen));
    bytes.resize(sizeof(hlen) + hlen + FQGPU_SEQ_FT_BYTES + FQGPU_QUAL_FT_BYTES);
    is.read(reinterpret_cast<char *>(bytes.data()) + sizeof(hlen), static_cast<std::streamsize>(bytes.size() - sizeof(hlen)));
    if (!is.good()) throw std::runtime_error("truncated dataset metadata");
    return fromBytes(bytes.data(), bytes.size());
  }
  friend bool operator==(const DatasetMeta &a, const DatasetMeta &b) {
    return a.first_header == b.first_header && std::memcmp(a.ft_seq.get(), b.ft_seq.get(), FQGPU_SEQ_FT_BYTES) == 0 &&
           std::memcmp(a.ft_qual.get(), b.ft_qual.get(), FQGPU_QUAL_FT_BYTES) == 0;
  }

  static RecordTable toRecordTable(const FastqChunk &chunk) {
    RecordTable recs(chunk.records.size());
    const char *base = chunk.raw_data.data();
    for (std::size_t i = 0; i < recs.size(); ++i) {
      const FastqRecord &r = chunk.records[i];
      recs[i] = {static_cast<uint32_t>(r.seqp - base), static_cast<uint32_t>(r.qualp - base), r.length};
    }
    return recs;
  }
};

class Workspace {
public:
  static std::size_t compressBoundSequence(std::size_t n) { return fqgpu_bound_seq(n); }
  static std::size_t compressBoundQuality(std::size_t n) { return fqgpu_bound_qual(n); }

protected:
  explicit Workspace(const DatasetMeta *meta, int device)
      : meta_(meta), fmt_(meta->header_fmt), first_header_fields_(headers::fromHeader(meta->first_header, fmt_)) {
    fqgpuCheck(fqgpu_ctx_create(device, meta->ft_seq.get(), meta->ft_qual.get(), &ctx_), "Workspace");
    // one block at a time per workspace (like the reference's): one encode lane, a quarter of the scratch
    fqgpuCheck(fqgpu_ctx_set_lanes(ctx_, 1), "Workspace");
    for (const auto t : fmt_.field_types) field_types_.push_back(t == headers::FieldType::STRING ? 1 : 0);
  }
  /** every chunk codes its first header against the dataset's first header (src/workspace.cpp:90-93) */
  void startNewChunk() { prev_header_fields_ = first_header_fields_; }
  ~Workspace() { fqgpu_ctx_destroy(ctx_); }
  Workspace(const Workspace &) = delete;
  Workspace &operator=(const Workspace &) = delete;
  const DatasetMeta *const meta_;
  const headers::HeaderFormatSpeciciation fmt_;
  const headers::header_fields_t first_header_fields_;
  headers::header_fields_t prev_header_fields_;
  std::vector<uint8_t> field_types_;  // fmt_.field_types as the C ABI takes them (0 = NUMERIC, 1 = STRING)
  fqgpu_ctx *ctx_ = nullptr;
};

class CompressionWorkspace : public Workspace {
public:
  explicit CompressionWorkspace(const DatasetMeta *meta, int device = 0) : Workspace(meta, device) {}

  /** Device memory for chunks of up to chunk_bytes now instead of inside the first encodeChunk
   *  (the farm calls it while it sets its workers up; the sizes assume records of 64 bytes or more) */
  void reserve(std::size_t chunk_bytes) {
    if (chunk_bytes) fqgpuCheck(fqgpu_ctx_reserve(ctx_, chunk_bytes, chunk_bytes / 64 + 1, chunk_bytes / 2 + 1), "reserve");
  }

  /** encodeChunk also leaves cbs.decode_index (extension: a decoder that has it decodes a stream from every
   *  snapshot at once; seq / qual and everything else in cbs are unchanged) */
  void setDecodeIndex(bool on, unsigned stride_symbols = 0) {
    decode_index_ = on;
    // symbols between two snapshots (a multiple of 64 Ki; 0 = the library's 1 Mi): a stride is what ONE wave decodes, so
    // a block alone takes a stride's time -- 105 ms at 1 Mi -- and the index grows as the stride shrinks (2 % of the
    // archive at 1 Mi, 8 % at 256 Ki)
    if (on && stride_symbols) fqgpuCheck(fqgpu_ctx_set_index_stride(ctx_, stride_symbols), "setDecodeIndex");
  }

  /** Encodes reads into cbs, allocating memory in cbs as needed; mutates the chunk (N -> A).
   *  The chunk may come UNPARSED (records empty, as FastqReader hands it out): the GPU then finds the
   *  records, and chunk.records / the length sums are filled in from its table.  The header fields are coded on
   *  the GPU as well (fqgpu_encode_headers_*: round 3 coded them on this thread, 78 ms per 256 MiB chunk beside
   *  3 ms of GPU work); FQGPU_SHIM_HOST_HEADERS=1 keeps the host coder (headers.hpp), same bytes. */
  void encodeChunk(FastqChunk &chunk, CompressedBuffersDst &cbs) {
    StageClock clk;
    cbs.clear();
    cbs.chunk_idx = chunk.idx;
    auto *raw = reinterpret_cast<uint8_t *>(chunk.raw_data.data());
    const bool parsed = !chunk.records.empty();
    RecordTable recs;
    if (parsed) recs = DatasetMeta::toRecordTable(chunk);
    std::size_t R = 0, n_bases = 0, used = 0;
    fqgpuCheck(fqgpu_encode_begin(ctx_, raw, chunk.raw_data.size(), parsed ? recs.data() : nullptr, recs.size(),
                                  FQGPU_F_WRITE_BACK_N | (decode_index_ ? FQGPU_F_DECODE_INDEX : 0u), &R, &n_bases, &used),
               "encodeChunk");
    // From here to fqgpu_encode_end the block is in flight: copies of chunk.raw_data and of `recs` may still be queued.
    // Whatever throws in between (a chunk that ends inside a record, the header coder, a failing call) must not
    // unwind past them: the guard waits for the handle before the local table dies and the caller's page-locked
    // vectors can go back to the pin cache.
    struct InFlight {
      fqgpu_ctx *ctx;
      bool armed = true;
      ~InFlight() { if (armed) (void)fqgpu_encode_cancel(ctx); }
    } in_flight{ctx_};
    static const bool host_headers = std::getenv("FQGPU_SHIM_HOST_HEADERS") != nullptr;
    if (!host_headers)
      fqgpuCheck(fqgpu_encode_headers_begin(ctx_, field_types_.data(), fmt_.separators.data(), static_cast<unsigned>(fmt_.n_fields()),
                                            reinterpret_cast<const uint8_t *>(meta_->first_header.data()), meta_->first_header.size()),
                 "encodeChunk");
    clk.lap("begin");
    if (!parsed) {
      recs.resize(R);
      fqgpuCheck(fqgpu_encode_records(ctx_, recs.data(), R), "encodeChunk");
      if (used != chunk.raw_data.size()) throw std::invalid_argument("encodeChunk: the chunk does not end with a complete record");
      recordViews(chunk, recs);
    }
    clk.lap("records");
    // ---- the header fields
    cbs.header_fields.resize(fmt_.n_fields());
    cbs.original_size.header_fields.resize(fmt_.n_fields());
    for (auto &field : cbs.header_fields) field.clear();
    bool on_host = host_headers;
    if (!on_host) {
      std::vector<fqgpu_field_sizes> sizes(fmt_.n_fields());
      std::size_t total = 0, bad = 0;
      const int rc = fqgpu_encode_headers_wait(ctx_, sizes.data(), &total, &bad);
      if (rc == FQGPU_E_HEADER) {
        on_host = true;  // the host coder throws the reference-side exception for that header (below)
      } else {
        fqgpuCheck(rc, "encodeChunk");
        header_stage_.resize(total);
        fqgpuCheck(fqgpu_encode_headers_end(ctx_, reinterpret_cast<uint8_t *>(header_stage_.data()), header_stage_.size()), "encodeChunk");
        const std::byte *at = header_stage_.data();
        for (std::size_t i = 0; i < fmt_.n_fields(); ++i) {
          auto &f = cbs.header_fields[i];
          f.isDifferentFlag.assign(at, at + sizes[i].isDifferentFlag); at += sizes[i].isDifferentFlag;
          f.content.assign(at, at + sizes[i].content); at += sizes[i].content;
          f.contentLength.assign(at, at + sizes[i].contentLength); at += sizes[i].contentLength;
        }
      }
    }
    if (on_host) {
      startNewChunk();
      for (const FastqRecord &r : chunk.records) headers::encodeHeader(r.header(), fmt_, prev_header_fields_, cbs.header_fields);
      if (!host_headers) throw std::logic_error("encodeChunk: the device refused a header the host coder takes");
    }
    clk.lap("headers");
    // ---- the streams, at their exact sizes
    std::size_t seq_len = 0, qual_len = 0, n_pos_len = 0;
    fqgpuCheck(fqgpu_encode_wait(ctx_, &seq_len, &qual_len, &n_pos_len), "encodeChunk");
    clk.lap("wait");
    cbs.seq.resize(seq_len);
    cbs.qual.resize(qual_len);
    cbs.readlens.resize(R * sizeof(readlen_t));
    // n_count / n_pos are appended, never cleared: what a reused CompressedBuffersDst holds in the reference
    const std::size_t cnt_at = cbs.n_count.size(), pos_at = cbs.n_pos.size();
    cbs.n_count.resize(cnt_at + R * sizeof(uint16_t));
    cbs.n_pos.resize(pos_at + n_pos_len * sizeof(uint16_t));
    fqgpuCheck(fqgpu_encode_end(ctx_, raw, reinterpret_cast<uint8_t *>(cbs.seq.data()), cbs.seq.size(), &seq_len,
                                reinterpret_cast<uint8_t *>(cbs.qual.data()), cbs.qual.size(), &qual_len,
                                reinterpret_cast<uint16_t *>(cbs.readlens.data()), reinterpret_cast<uint16_t *>(cbs.n_count.data() + cnt_at),
                                reinterpret_cast<uint16_t *>(cbs.n_pos.data() + pos_at), n_pos_len, &n_pos_len),
               "encodeChunk");
    in_flight.armed = false;
    for (int s = 0; s < 2 && decode_index_; ++s) {
      std::size_t n = 0;
      fqgpuCheck(fqgpu_encode_index(ctx_, s, nullptr, 0, &n), "encodeChunk");
      cbs.decode_index[s].resize(n);
      fqgpuCheck(fqgpu_encode_index(ctx_, s, reinterpret_cast<uint8_t *>(cbs.decode_index[s].data()), n, &n), "encodeChunk");
    }
    clk.lap("end");
    cbs.original_size.n_records = static_cast<uint32_t>(R);
    cbs.original_size.total = static_cast<uint32_t>(chunk.raw_data.size());
    compressMiscBuffers(cbs);
    clk.lap("misc");
    clk.done(chunk.idx);
  }

private:
  stream_bytes_t header_stage_;  // page-locked landing place of the header field streams
  bool decode_index_ = false;

public:
  /** The misc pass (the reference's compressMiscBuffers, src/workspace.cpp:176-213): readlens, n_count,
   *  n_pos and every header field stream through memcompress, original sizes recorded for the container */
  void compressMiscBuffers(CompressedBuffersDst &cbs) const { compressMiscBuffers(cbs, fmt_); }
  /** the same without a workspace (host-only tools and tests: no GPU involved) */
  static void compressMiscBuffers(CompressedBuffersDst &cbs, const headers::HeaderFormatSpeciciation &fmt) {
    cbs.compressed_header_fields.resize(fmt.n_fields());
    cbs.original_size.header_fields.assign(fmt.n_fields(), {});
    for (auto &packed : cbs.compressed_header_fields) packed.clear();
    miscStreams(cbs, fmt, [](const auto &plain, std::vector<std::byte> &packed, uint32_t &size) {
      size = static_cast<uint32_t>(plain.size());
      packed.resize(fqgpu_memcompress_bound(plain.size()));
      packed.resize(memcompress(packed.data(), plain.data(), plain.size()));
    });
  }

  /** chunk.records (pointers into the chunk) and the length sums from a record table */
  static void recordViews(FastqChunk &chunk, const RecordTable &recs) {
    chunk.records.resize(recs.size());
    chunk.tot_reads_length = chunk.headers_length = 0;
    char *base = chunk.raw_data.data();
    std::size_t line = 0;  // start of the record's header line
    for (std::size_t i = 0; i < recs.size(); ++i) {
      FastqRecord &r = chunk.records[i];
      r.headerp = base + line;
      r.header_length = static_cast<readlen_t>(recs[i].seq_off - 1 - line);
      r.seqp = base + recs[i].seq_off;
      r.qualp = base + recs[i].qual_off;
      r.length = static_cast<readlen_t>(recs[i].len);
      chunk.tot_reads_length += r.length;
      chunk.headers_length += r.header_length;
      line = static_cast<std::size_t>(recs[i].qual_off) + recs[i].len + 1;
    }
  }

};

class DecompressionWorkspace : public Workspace {
public:
  explicit DecompressionWorkspace(const DatasetMeta *meta, int device = 0) : Workspace(meta, device) {}

  /** Both passes of decodeChunk (src/workspace.cpp:47-88): the first lays the chunk out
   *  (headers decoded on the host, lengths from readlens, '+' and newlines), the second fills the
   *  sequence and quality lines on the GPU */
  void decodeChunk(FastqChunk &chunk, CompressedBuffersSrc &cbs) {
    StageClock clk;
    chunk.clear();  // prepareFastqChunk (src/workspace.h:127-133)
    chunk.idx = cbs.chunk_idx;
    // (chunks of one archive differ by a record or two: room for the next ones, or every slightly longer chunk costs a
    // fresh page-locked block -- 45 ms of hipHostMalloc for 256 MiB)
    if (chunk.raw_data.capacity() < cbs.original_size.total) chunk.raw_data.reserve(cbs.original_size.total + cbs.original_size.total / 16 + 4096);
    chunk.raw_data.resize(cbs.original_size.total);
    chunk.records.resize(cbs.original_size.n_records);
    startNewChunk();
    clk.lap("resize");
    decompressMiscBuffers(cbs);
    clk.lap("misc");
    if (cbs.header_fields.size() != fmt_.n_fields()) throw std::invalid_argument("decodeChunk: header field streams do not match the format");
    if (cbs.readlens.size() < chunk.records.size() * sizeof(readlen_t)) throw std::invalid_argument("decodeChunk: readlens too short");
    char *dst = chunk.raw_data.data();
    char *const end = dst + chunk.raw_data.size();
    for (std::size_t i = 0, E = chunk.records.size(); i < E; ++i) {
      FastqRecord &r = chunk.records[i];
      std::memcpy(&r.length, cbs.readlens.data() + sizeof(readlen_t) * i, sizeof(readlen_t));
      r.headerp = dst;
      r.header_length = static_cast<readlen_t>(decodeHeaderChecked(dst, end, cbs));
      dst += r.header_length;
      if (static_cast<std::size_t>(end - dst) < 2u * r.length + 5u) throw std::out_of_range("decodeChunk: original_size.total too small");
      *dst++ = '\n';
      r.seqp = dst;  dst += r.length;  *dst++ = '\n';
      *dst++ = '+';  *dst++ = '\n';
      r.qualp = dst; dst += r.length;  *dst++ = '\n';
      chunk.tot_reads_length += r.length;
      chunk.headers_length += r.header_length;
    }
    clk.lap("headers+layout");
    RecordTable recs = DatasetMeta::toRecordTable(chunk);
    clk.lap("table");
    // (with cbs.decode_index -- an extension, empty in a reference archive -- every stream is decoded from all its
    // snapshots at once; without, by one lane from its end: the format's own pace)
    fqgpuCheck(fqgpu_decode_block_indexed(ctx_, reinterpret_cast<const uint8_t *>(cbs.seq.data()), cbs.seq.size(),
                                          reinterpret_cast<const uint8_t *>(cbs.qual.data()), cbs.qual.size(),
                                          reinterpret_cast<const uint16_t *>(cbs.n_count.data()),
                                          cbs.index.n_count / sizeof(uint16_t),
                                          reinterpret_cast<const uint16_t *>(cbs.n_pos.data()),
                                          cbs.index.n_pos / sizeof(uint16_t), recs.data(), recs.size(),
                                          reinterpret_cast<uint8_t *>(chunk.raw_data.data()), chunk.raw_data.size(),
                                          reinterpret_cast<const uint8_t *>(cbs.decode_index[0].data()), cbs.decode_index[0].size(),
                                          reinterpret_cast<const uint8_t *>(cbs.decode_index[1].data()), cbs.decode_index[1].size()),
               "decodeChunk");
    clk.lap("gpu");
    clk.done(chunk.idx);
  }

  /** The misc pass backwards (the reference's decompressMiscBuffers, src/workspace.cpp:215-256): every
   *  misc stream is restored from its compressed twin to the size the container recorded; index.n_count /
   *  index.n_pos are set to the ends of the buffers (the decoder pops from there) */
  void decompressMiscBuffers(CompressedBuffersSrc &cbs) const { decompressMiscBuffers(cbs, fmt_); }
  static void decompressMiscBuffers(CompressedBuffersSrc &cbs, const headers::HeaderFormatSpeciciation &fmt) {
    if (cbs.compressed_header_fields.size() != fmt.n_fields() || cbs.original_size.header_fields.size() != fmt.n_fields())
      throw std::invalid_argument("decodeChunk: header field streams do not match the format");
    cbs.header_fields.resize(fmt.n_fields());
    for (auto &plain : cbs.header_fields) plain.clear();
    miscStreams(cbs, fmt, [](auto &plain, const std::vector<std::byte> &packed, const uint32_t &size) {
      plain.resize(size);
      memdecompress(plain.data(), plain.size(), packed.data(), packed.size());
    });
    cbs.index.n_count = cbs.n_count.size();
    cbs.index.n_pos = cbs.n_pos.size();
  }

private:
  /** decodeHeader with the output bound checked: near the end of the chunk the header goes through
   *  a local buffer, since the field decoders may write up to FIELDLEN_MAX bytes per field */
  unsigned decodeHeaderChecked(char *dst, char *end, CompressedBuffersSrc &cbs) {
    const std::size_t worst = 1 + fmt_.n_fields() * (headers::FIELDLEN_MAX + 1);
    if (static_cast<std::size_t>(end - dst) >= worst) return headers::decodeHeader(dst, fmt_, prev_header_fields_, cbs.header_fields);
    tail_.resize(worst);
    const unsigned n = headers::decodeHeader(tail_.data(), fmt_, prev_header_fields_, cbs.header_fields);
    if (static_cast<std::size_t>(end - dst) < n) throw std::out_of_range("decodeChunk: original_size.total too small");
    std::memcpy(dst, tail_.data(), n);
    // string fields of the previous header must point at bytes that stay: re-anchor them in dst
    headers::header_fields_t anchored = headers::fromHeader(std::string_view(dst, n), fmt_);
    for (std::size_t i = 0; i < anchored.size(); ++i)
      if (fmt_.field_types[i] == headers::FieldType::STRING) prev_header_fields_[i] = anchored[i];
    return n;
  }
  std::vector<char> tail_;
};

/** The host parser (FastqReader::parseRecords, src/fastq_io.cpp:67-125) on top of fqgpu_parse_fastq: fills
 *  chunk.records; returns the bytes up to the end of the last complete record */
inline std::size_t parseRecords(FastqChunk &chunk) {
  const auto *raw = reinterpret_cast<const uint8_t *>(chunk.raw_data.data());
  const long n = fqgpu_parse_fastq(raw, chunk.raw_data.size(), nullptr, 0);
  if (n < 0) throw std::invalid_argument("malformed FASTQ block");
  RecordTable recs(static_cast<std::size_t>(n));
  fqgpu_parse_fastq(raw, chunk.raw_data.size(), recs.data(), recs.size());
  CompressionWorkspace::recordViews(chunk, recs);
  return recs.empty() ? 0 : static_cast<std::size_t>(recs.back().qual_off) + recs.back().len + 1;
}

}  // namespace fqcomp28
