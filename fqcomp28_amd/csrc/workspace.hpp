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

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <istream>
#include <memory>
#include <new>
#include <ostream>
#include <stdexcept>
#include <string>
#include <string_view>
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
  /* like the reference, clear() does NOT clear n_count / n_pos */
  virtual void clear() {
    seq.clear(); qual.clear();
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

  /** DatasetMeta::storeToStream (src/prepare.cpp:12-21), byte for byte: u16 header length, the
   *  header, then the two FreqTable PODs as they lie in memory */
  static void storeToStream(const DatasetMeta &meta, std::ostream &os) {
    if (meta.first_header.size() > 0xFFFFu) throw std::invalid_argument("first header longer than a readlen_t");
    const readlen_t hlen = static_cast<readlen_t>(meta.first_header.size());
    os.write(reinterpret_cast<const char *>(&hlen), sizeof(hlen));
    os.write(meta.first_header.data(), hlen);
    os.write(reinterpret_cast<const char *>(meta.ft_seq.get()), FQGPU_SEQ_FT_BYTES);
    os.write(reinterpret_cast<const char *>(meta.ft_qual.get()), FQGPU_QUAL_FT_BYTES);
  }
  /** DatasetMeta::loadFromStream (src/prepare.cpp:23-41) */
  static DatasetMeta loadFromStream(std::istream &is) {
    readlen_t hlen = 0;
    is.read(reinterpret_cast<char *>(&hlen), sizeof(hlen));
    std::string header(hlen, '!');
    is.read(header.data(), hlen);
    DatasetMeta meta{std::string_view(header)};
    is.read(reinterpret_cast<char *>(meta.ft_seq.get()), FQGPU_SEQ_FT_BYTES);
    is.read(reinterpret_cast<char *>(meta.ft_qual.get()), FQGPU_QUAL_FT_BYTES);
    if (!is.good()) throw std::runtime_error("truncated dataset metadata");
    return meta;
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
  fqgpu_ctx *ctx_ = nullptr;
};

class CompressionWorkspace : public Workspace {
public:
  explicit CompressionWorkspace(const DatasetMeta *meta, int device = 0) : Workspace(meta, device) {}

  /** Encodes reads into cbs, allocating memory in cbs as needed; mutates the chunk (N -> A) */
  void encodeChunk(FastqChunk &chunk, CompressedBuffersDst &cbs) {
    cbs.clear();
    cbs.chunk_idx = chunk.idx;
    const std::size_t R = chunk.records.size();
    cbs.seq.resize(compressBoundSequence(chunk.tot_reads_length));
    cbs.qual.resize(compressBoundQuality(chunk.tot_reads_length));
    cbs.readlens.resize(R * sizeof(readlen_t));
    cbs.header_fields.resize(fmt_.n_fields());
    cbs.original_size.header_fields.resize(fmt_.n_fields());
    for (auto &field : cbs.header_fields) field.clear();
    startNewChunk();
    for (const FastqRecord &r : chunk.records) headers::encodeHeader(r.header(), fmt_, prev_header_fields_, cbs.header_fields);
    for (std::size_t i = 0; i < fmt_.n_fields(); ++i) cbs.original_size.header_fields[i] = cbs.header_fields[i].originalSizes();
    RecordTable recs = DatasetMeta::toRecordTable(chunk);
    u16_buffer_t n_count(R), n_pos(chunk.tot_reads_length);
    std::size_t seq_len = 0, qual_len = 0, n_pos_len = 0;
    fqgpuCheck(fqgpu_encode_block(ctx_, reinterpret_cast<uint8_t *>(chunk.raw_data.data()),
                                  chunk.raw_data.size(), recs.data(), R,
                                  reinterpret_cast<uint8_t *>(cbs.seq.data()), cbs.seq.size(), &seq_len,
                                  reinterpret_cast<uint8_t *>(cbs.qual.data()), cbs.qual.size(), &qual_len,
                                  reinterpret_cast<uint16_t *>(cbs.readlens.data()), n_count.data(),
                                  n_pos.data(), n_pos.size(), &n_pos_len, FQGPU_F_WRITE_BACK_N),
               "encodeChunk");
    cbs.seq.resize(seq_len);
    cbs.qual.resize(qual_len);
    // appended, never cleared: what a reused CompressedBuffersDst holds in the reference
    append(cbs.n_count, n_count.data(), R);
    append(cbs.n_pos, n_pos.data(), n_pos_len);
    cbs.original_size.n_records = static_cast<uint32_t>(R);
    cbs.original_size.total = static_cast<uint32_t>(chunk.raw_data.size());
    cbs.original_size.readlens = static_cast<uint32_t>(cbs.readlens.size());
    cbs.original_size.n_count = static_cast<uint32_t>(cbs.n_count.size());
    cbs.original_size.n_pos = static_cast<uint32_t>(cbs.n_pos.size());
    compressMiscBuffers(cbs);
  }

  /** CompressionWorkspace::compressMiscBuffers (src/workspace.cpp:176-213): readlens, n_count, n_pos
   *  and every header field stream through memcompress; original sizes recorded for the container */
  void compressMiscBuffers(CompressedBuffersDst &cbs) const { compressMiscBuffers(cbs, fmt_); }
  /** the same without a workspace (host-only tools and tests: no GPU involved) */
  static void compressMiscBuffers(CompressedBuffersDst &cbs, const headers::HeaderFormatSpeciciation &fmt_) {
    cbs.original_size.readlens = static_cast<uint32_t>(cbs.readlens.size());
    compressBuffer(cbs.compressed_readlens, cbs.readlens);
    cbs.original_size.n_count = static_cast<uint32_t>(cbs.n_count.size());
    compressBuffer(cbs.compressed_n_count, cbs.n_count);
    cbs.original_size.n_pos = static_cast<uint32_t>(cbs.n_pos.size());
    compressBuffer(cbs.compressed_n_pos, cbs.n_pos);
    cbs.compressed_header_fields.resize(fmt_.n_fields());
    cbs.original_size.header_fields.resize(fmt_.n_fields());
    for (std::size_t i = 0, E = fmt_.n_fields(); i < E; ++i) {
      const auto &field_data = cbs.header_fields[i];
      auto &field_cdata = cbs.compressed_header_fields[i];
      auto &original_size = cbs.original_size.header_fields[i];
      if (fmt_.field_types[i] == headers::FieldType::STRING) {
        compressBuffer(field_cdata.isDifferentFlag, field_data.isDifferentFlag);
        compressBuffer(field_cdata.content, field_data.content);
        compressBuffer(field_cdata.contentLength, field_data.contentLength);
        original_size = field_data.originalSizes();
      } else { /* NUMERIC */
        compressBuffer(field_cdata.content, field_data.content);
        field_cdata.isDifferentFlag.clear(); field_cdata.contentLength.clear();
        original_size = {};
        original_size.content = static_cast<uint32_t>(field_data.content.size());
      }
    }
  }

private:
  /** compressBuffer (src/workspace.cpp:258-265) */
  template <class Bytes> static std::size_t compressBuffer(std::vector<std::byte> &dst, const Bytes &src) {
    dst.resize(fqgpu_memcompress_bound(src.size()));
    const std::size_t csize = memcompress(dst.data(), src.data(), src.size());
    dst.resize(csize);
    return csize;
  }
  template <class Bytes> static void append(Bytes &dst, const uint16_t *src, std::size_t n) {
    const std::size_t old = dst.size();
    dst.resize(old + n * sizeof(uint16_t));
    std::memcpy(dst.data() + old, src, n * sizeof(uint16_t));
  }
};

class DecompressionWorkspace : public Workspace {
public:
  explicit DecompressionWorkspace(const DatasetMeta *meta, int device = 0) : Workspace(meta, device) {}

  /** Both passes of decodeChunk (src/workspace.cpp:47-88): the first lays the chunk out
   *  (headers decoded on the host, lengths from readlens, '+' and newlines), the second fills the
   *  sequence and quality lines on the GPU */
  void decodeChunk(FastqChunk &chunk, CompressedBuffersSrc &cbs) {
    chunk.clear();  // prepareFastqChunk (src/workspace.h:127-133)
    chunk.idx = cbs.chunk_idx;
    chunk.raw_data.resize(cbs.original_size.total);
    chunk.records.resize(cbs.original_size.n_records);
    startNewChunk();
    decompressMiscBuffers(cbs);
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
    RecordTable recs = DatasetMeta::toRecordTable(chunk);
    fqgpuCheck(fqgpu_decode_block(ctx_, reinterpret_cast<const uint8_t *>(cbs.seq.data()), cbs.seq.size(),
                                  reinterpret_cast<const uint8_t *>(cbs.qual.data()), cbs.qual.size(),
                                  reinterpret_cast<const uint16_t *>(cbs.n_count.data()),
                                  cbs.index.n_count / sizeof(uint16_t),
                                  reinterpret_cast<const uint16_t *>(cbs.n_pos.data()),
                                  cbs.index.n_pos / sizeof(uint16_t), recs.data(), recs.size(),
                                  reinterpret_cast<uint8_t *>(chunk.raw_data.data()), chunk.raw_data.size()),
               "decodeChunk");
  }

  /** DecompressionWorkspace::decompressMiscBuffers (src/workspace.cpp:215-256): every misc stream
   *  is restored from its compressed twin to the size the container recorded; index.n_count /
   *  index.n_pos are set to the ends of the buffers (the decoder pops from there) */
  void decompressMiscBuffers(CompressedBuffersSrc &cbs) const { decompressMiscBuffers(cbs, fmt_); }
  static void decompressMiscBuffers(CompressedBuffersSrc &cbs, const headers::HeaderFormatSpeciciation &fmt_) {
    cbs.readlens.resize(cbs.original_size.readlens);
    memdecompress(cbs.readlens.data(), cbs.readlens.size(), cbs.compressed_readlens.data(), cbs.compressed_readlens.size());
    cbs.index.n_count = cbs.original_size.n_count;
    cbs.n_count.resize(cbs.original_size.n_count);
    memdecompress(cbs.n_count.data(), cbs.n_count.size(), cbs.compressed_n_count.data(), cbs.compressed_n_count.size());
    cbs.index.n_pos = cbs.original_size.n_pos;
    cbs.n_pos.resize(cbs.original_size.n_pos);
    memdecompress(cbs.n_pos.data(), cbs.n_pos.size(), cbs.compressed_n_pos.data(), cbs.compressed_n_pos.size());
    if (cbs.compressed_header_fields.size() != fmt_.n_fields() || cbs.original_size.header_fields.size() != fmt_.n_fields())
      throw std::invalid_argument("decodeChunk: header field streams do not match the format");
    cbs.header_fields.resize(fmt_.n_fields());
    for (std::size_t i = 0, E = fmt_.n_fields(); i < E; ++i) {
      const auto &field_cdata = cbs.compressed_header_fields[i];
      const auto &original_size = cbs.original_size.header_fields[i];
      auto &field_data = cbs.header_fields[i];
      field_data.clear();
      field_data.content.resize(original_size.content);
      memdecompress(field_data.content.data(), field_data.content.size(), field_cdata.content.data(), field_cdata.content.size());
      if (fmt_.field_types[i] == headers::FieldType::STRING) {
        field_data.isDifferentFlag.resize(original_size.isDifferentFlag);
        field_data.contentLength.resize(original_size.contentLength);
        memdecompress(field_data.isDifferentFlag.data(), field_data.isDifferentFlag.size(), field_cdata.isDifferentFlag.data(), field_cdata.isDifferentFlag.size());
        memdecompress(field_data.contentLength.data(), field_data.contentLength.size(), field_cdata.contentLength.data(), field_cdata.contentLength.size());
      }
    }
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

/** FastqReader::parseRecords (src/fastq_io.cpp:67-125) on top of fqgpu_parse_fastq */
inline std::size_t parseRecords(FastqChunk &chunk) {
  const auto *raw = reinterpret_cast<const uint8_t *>(chunk.raw_data.data());
  const long n = fqgpu_parse_fastq(raw, chunk.raw_data.size(), nullptr, 0);
  if (n < 0) throw std::invalid_argument("malformed FASTQ block");
  std::vector<fqgpu_rec> recs(static_cast<std::size_t>(n));
  fqgpu_parse_fastq(raw, chunk.raw_data.size(), recs.data(), recs.size());
  chunk.records.resize(recs.size());
  char *base = chunk.raw_data.data();
  std::size_t prev_end = 0;
  for (std::size_t i = 0; i < recs.size(); ++i) {
    FastqRecord &r = chunk.records[i];
    r.headerp = base + prev_end;
    r.header_length = static_cast<readlen_t>(recs[i].seq_off - 1 - prev_end);
    r.seqp = base + recs[i].seq_off;
    r.qualp = base + recs[i].qual_off;
    r.length = static_cast<readlen_t>(recs[i].len);
    chunk.tot_reads_length += r.length;
    chunk.headers_length += r.header_length;
    prev_end = recs[i].qual_off + recs[i].len + 1;
  }
  return prev_end;
}

}  // namespace fqcomp28
