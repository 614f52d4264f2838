// archive.hpp -- the reference's `.fqc` block container on top of the shim's buffers
// (SURVEY.md 8(f) row 2).  Host code; mirrors, name for name:
//   Archive, BlockInfo, writeBlock / readBlock / writeIndex / meta / indexBytes   src/archive.h:18-96
//   layout of the file                                                            src/archive.h:10-17
//   field order inside a block                                                    src/archive.cpp:57-106, 108-163
//   FastqReader::readNextChunk / FastqWriter::writeChunk                          src/fastq_io.cpp:23-65, 131-143
//   analyzeDataset                                                                src/prepare.cpp:42-47
//
// File layout (all integers little-endian, as the reference writes them from memory):
//   u32  n_blocks                      written LAST, at offset 0 (writeIndex, src/archive.cpp:45-55)
//   meta u16 hlen | first header | FreqTable<256,4> (3076 B) | FreqTable<8192,64> (1 081 348 B)
//   block x n_blocks, in COMPLETION order:
//        u32 total | u32 n_records
//        u32 orig | u32 csize | bytes          readlens
//        u32 orig | u32 csize | bytes          n_count
//        u32 orig | u32 csize | bytes          n_pos
//        u32 csize | bytes                     seq     (the FSE stream, bit-identical to the reference's)
//        u32 csize | bytes                     qual
//        per header field: STRING  3 x (u32 orig | u32 csize | bytes)  flags, content, lengths
//                          NUMERIC 1 x (u32 orig | u32 csize | bytes)  content
//   index: n_blocks x { i64 offset; u32 idx; 4 bytes of padding }  = sizeof(BlockInfo) = 16
// Byte-compatible with the reference for every field except the csize/bytes of the misc streams,
// which hold this library's own coder's output instead of libbsc's (out of parity scope).
#pragma once

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <filesystem>
#include <fstream>
#include <mutex>

#include "workspace.hpp"

namespace fqcomp28 {

using path_t = std::filesystem::path;

inline void checkStreamState(std::ios &s, const path_t &path) {  // src/utils.cpp:4-7
  if (!s.good()) throw std::system_error(std::make_error_code(std::errc::io_error), path.string());
}

template <class Target, class Source> Target narrow_cast(Source v) {  // src/utils.h:17-23
  auto r = static_cast<Target>(v);
  if (static_cast<Source>(r) != v) throw std::runtime_error("narrow_cast<>() failed");
  return r;
}

/** Reads data from a single .fastq file (src/fastq_io.h:10-42): chunks of about reading_size
 *  bytes that end with a complete record; the partial record at the end is carried over */
class FastqReader {
public:
  FastqReader(const std::string &mates1, std::size_t reading_size)
      : reading_size_(reading_size), ifs1_(mates1, std::ios::binary), bytes_left1_(std::filesystem::file_size(mates1)) {
    checkStreamState(ifs1_, mates1);
  }

  /** @return true if reading was successful */
  bool readNextChunk(FastqChunk &chunk) {
    chunk.clear();
    const std::lock_guard guard(mtx_);
    if (bytes_left1_ == 0) return false;  // (a partial record left at the very end is dropped like in the reference)
    chunk.idx = chunks_read_++;           // under the lock: the reference increments before it
    const std::size_t to_read = std::min(reading_size_ - partial1_.size(), bytes_left1_);
    chunk.raw_data.resize(partial1_.size() + to_read);
    char *buf = chunk.raw_data.data();
    std::memcpy(buf, partial1_.data(), partial1_.size());
    buf += partial1_.size();
    partial1_.clear();
    ifs1_.read(buf, static_cast<std::streamsize>(to_read));
    if (!ifs1_) throw std::runtime_error("FastqReader: short read");
    const std::size_t actual_chunk_size = parseRecords(chunk);
    partial1_.assign(chunk.raw_data.begin() + static_cast<std::ptrdiff_t>(actual_chunk_size), chunk.raw_data.end());
    chunk.raw_data.resize(actual_chunk_size);
    bytes_left1_ -= to_read;
    if (chunk.records.empty()) throw std::runtime_error("FastqReader: reading size smaller than one record");
    return true;
  }

private:
  const std::size_t reading_size_;
  std::vector<char> partial1_;
  std::ifstream ifs1_;
  std::size_t bytes_left1_;
  unsigned chunks_read_ = 0;
  std::mutex mtx_;
};

/** Writes data to a single .fastq file, chunks in their original order (src/fastq_io.h:44-59) */
class FastqWriter {
public:
  explicit FastqWriter(const std::string &mates1) : ofs1_(mates1, std::ios::binary) { checkStreamState(ofs1_, mates1); }
  void writeChunk(FastqChunk const &chunk) {
    std::unique_lock guard(mtx_);
    cv_.wait(guard, [&] { return chunk.idx == chunks_written_; });
    ofs1_.write(chunk.raw_data.data(), static_cast<std::streamsize>(chunk.raw_data.size()));
    chunks_written_++;
    guard.unlock();
    cv_.notify_all();
  }
  void flush() { ofs1_.flush(); }

private:
  std::ofstream ofs1_;
  unsigned chunks_written_ = 0;
  std::mutex mtx_;
  std::condition_variable cv_;
};

/** analyzeDataset (src/prepare.cpp:42-47): the first sample_size_bytes of the file, tables on the GPU */
inline DatasetMeta analyzeDataset(const path_t &fastq_file, std::size_t sample_size_bytes, int device = 0) {
  FastqChunk chunk;
  FastqReader reader(fastq_file.string(), sample_size_bytes);
  if (!reader.readNextChunk(chunk)) throw std::runtime_error("analyzeDataset: empty input");
  return DatasetMeta(chunk, device);
}

class Archive {
  /** Describes location and size of a data block in the archive file (src/archive.h:20-27) */
  struct BlockInfo {
    int64_t offset;
    uint32_t idx;  // position (order) of the corresponding chunk in the input file
    uint32_t pad_; // the reference's struct has 4 bytes of tail padding here; written as zeros
    bool operator==(const BlockInfo &o) const { return offset == o.offset && idx == o.idx; }
  };
  static_assert(sizeof(BlockInfo) == 16, "index entries are 16 bytes in the reference");
  constexpr static std::streamoff OFFSET_META = sizeof(uint32_t);

public:
  /** Creates Archive to read compressed data from an existing file (src/archive.cpp:6-11) */
  explicit Archive(const path_t &archive_path) : fs_(archive_path, std::ios_base::binary | std::ios_base::in) {
    checkStreamState(fs_, archive_path);
    readArchiveHeader();
  }
  /** Creates an archive to write compressed data to; meta gathered from the first
   *  sample_size_bytes of file_to_gather_meta (src/archive.cpp:13-20) */
  Archive(const path_t &archive_path, const path_t &file_to_gather_meta, std::size_t sample_size_bytes, int device = 0)
      : Archive(archive_path, analyzeDataset(file_to_gather_meta, sample_size_bytes, device)) {}
  /** ... or handed over (tables computed elsewhere) */
  Archive(const path_t &archive_path, DatasetMeta meta)
      : fs_(archive_path, std::ios_base::binary | std::ios_base::out | std::ios_base::trunc), meta_(std::move(meta)) {
    checkStreamState(fs_, archive_path);
    writeMeta();
  }

  /** src/archive.cpp:57-106 */
  void writeBlock(const CompressedBuffersDst &cb) {
    BlockInfo binfo = {};
    binfo.idx = cb.chunk_idx;
    const std::lock_guard guard(mtx_);
    binfo.offset = narrow_cast<int64_t>(static_cast<std::streamoff>(fs_.tellp()));
    writeInteger(cb.original_size.total);
    writeInteger(cb.original_size.n_records);
    writeInteger(cb.original_size.readlens);
    writeBytes(cb.compressed_readlens);
    writeInteger(cb.original_size.n_count);
    writeBytes(cb.compressed_n_count);
    writeInteger(cb.original_size.n_pos);
    writeBytes(cb.compressed_n_pos);
    writeBytes(cb.seq);
    writeBytes(cb.qual);
    if (cb.compressed_header_fields.size() != meta_.header_fmt.n_fields() ||
        cb.original_size.header_fields.size() != meta_.header_fmt.n_fields())
      throw std::invalid_argument("writeBlock: header field streams do not match the archive's format");
    for (std::size_t i = 0, E = meta_.header_fmt.n_fields(); i < E; ++i) {
      const auto &field_cdata = cb.compressed_header_fields[i];
      const auto &field_original_size = cb.original_size.header_fields[i];
      if (meta_.header_fmt.field_types[i] == headers::FieldType::STRING) {
        writeInteger(field_original_size.isDifferentFlag);
        writeBytes(field_cdata.isDifferentFlag);
        writeInteger(field_original_size.content);
        writeBytes(field_cdata.content);
        writeInteger(field_original_size.contentLength);
        writeBytes(field_cdata.contentLength);
      } else {
        writeInteger(field_original_size.content);
        writeBytes(field_cdata.content);
      }
    }
    index_.push_back(binfo);
  }

  /** src/archive.cpp:108-163; blocks come in the order of the input file (sorted index) */
  bool readBlock(CompressedBuffersSrc &cb) {
    cb.clear();
    const std::lock_guard guard(mtx_);
    if (blocks_processed_ == index_.size()) return false;
    const auto &binfo = index_[blocks_processed_++];
    fs_.seekg(binfo.offset);
    cb.chunk_idx = binfo.idx;
    cb.original_size.total = readInteger<uint32_t>();
    cb.original_size.n_records = readInteger<uint32_t>();
    cb.original_size.readlens = readInteger<uint32_t>();
    readBytes(cb.compressed_readlens);
    cb.original_size.n_count = readInteger<uint32_t>();
    readBytes(cb.compressed_n_count);
    cb.original_size.n_pos = readInteger<uint32_t>();
    readBytes(cb.compressed_n_pos);
    readBytes(cb.seq);
    readBytes(cb.qual);
    const auto n_fields = meta_.header_fmt.n_fields();
    cb.original_size.header_fields.resize(n_fields);
    cb.header_fields.resize(n_fields);
    cb.compressed_header_fields.resize(n_fields);
    for (std::size_t i = 0; i < n_fields; ++i) {
      auto &field_original_size = cb.original_size.header_fields[i];
      auto &field_cdata = cb.compressed_header_fields[i];
      field_original_size = {};
      if (meta_.header_fmt.field_types[i] == headers::FieldType::STRING) {
        field_original_size.isDifferentFlag = readInteger<uint32_t>();
        readBytes(field_cdata.isDifferentFlag);
        field_original_size.content = readInteger<uint32_t>();
        readBytes(field_cdata.content);
        field_original_size.contentLength = readInteger<uint32_t>();
        readBytes(field_cdata.contentLength);
      } else {
        field_original_size.content = readInteger<uint32_t>();
        readBytes(field_cdata.content);
      }
    }
    if (!fs_.good()) throw std::runtime_error("readBlock: truncated archive");
    return true;
  }

  /** src/archive.cpp:45-55: the block count goes to offset 0, the index behind the last block */
  void writeIndex() {
    const std::lock_guard guard(mtx_);
    const std::streamoff data_end_pos = fs_.tellp();
    fs_.seekp(0);
    const auto index_size = narrow_cast<uint32_t>(index_.size());
    writeInteger(index_size);
    fs_.seekp(data_end_pos);
    fs_.write(reinterpret_cast<const char *>(index_.data()), narrow_cast<std::streamsize>(index_.size() * sizeof(BlockInfo)));
  }
  void flush() { fs_.flush(); }

  const DatasetMeta &meta() const { return meta_; }
  [[nodiscard]] std::size_t indexBytes() const { return sizeof(uint32_t) + index_.size() * sizeof(BlockInfo); }
  [[nodiscard]] std::size_t nBlocks() const { return index_.size(); }
  /** (offset, idx) of every block, in index order -- for tests (the reference's ArchiveTester) */
  [[nodiscard]] std::vector<std::pair<int64_t, uint32_t>> indexEntries() const {
    std::vector<std::pair<int64_t, uint32_t>> v;
    for (const auto &b : index_) v.emplace_back(b.offset, b.idx);
    return v;
  }

private:
  void writeMeta() {  // src/archive.cpp:22-25
    fs_.seekp(OFFSET_META);
    DatasetMeta::storeToStream(meta_, fs_);
  }
  void readArchiveHeader() {  // src/archive.cpp:27-43
    uint32_t n_blocks = 0;
    fs_.read(reinterpret_cast<char *>(&n_blocks), sizeof(n_blocks));
    meta_ = DatasetMeta::loadFromStream(fs_);
    const auto data_start_pos = fs_.tellg();
    fs_.seekg(-narrow_cast<std::streamoff>(static_cast<std::size_t>(n_blocks) * sizeof(BlockInfo)), std::ios_base::end);
    index_.resize(n_blocks);
    fs_.read(reinterpret_cast<char *>(index_.data()), narrow_cast<std::streamsize>(static_cast<std::size_t>(n_blocks) * sizeof(BlockInfo)));
    if (!fs_.good()) throw std::runtime_error("archive header or index truncated");
    fs_.seekg(data_start_pos);
    sortIndex();
  }
  template <class Vec> void writeBytes(const Vec &bytes) {
    const auto sz = narrow_cast<uint32_t>(bytes.size());
    writeInteger(sz);
    fs_.write(reinterpret_cast<const char *>(bytes.data()), sz);
  }
  template <class Vec> void readBytes(Vec &bytes) {
    const auto sz = readInteger<uint32_t>();
    bytes.resize(sz);
    fs_.read(reinterpret_cast<char *>(bytes.data()), sz);
  }
  template <class T> void writeInteger(const T val) { fs_.write(reinterpret_cast<const char *>(&val), sizeof(T)); }
  template <class T> T readInteger() {
    T ret{};
    fs_.read(reinterpret_cast<char *>(&ret), sizeof(T));
    return ret;
  }
  /** index entries in the order of the corresponding input chunks (src/archive.h:85-89) */
  void sortIndex() {
    std::sort(index_.begin(), index_.end(), [](const auto &l, const auto &r) { return l.idx < r.idx; });
  }

  std::vector<BlockInfo> index_;
  std::size_t blocks_processed_ = 0;
  std::fstream fs_;
  DatasetMeta meta_;
  std::mutex mtx_;
};

}  // namespace fqcomp28
