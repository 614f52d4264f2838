// archive.hpp -- the `.fqc` container and the FASTQ file ends of the block farm, written from the
// FORMAT (reference src/archive.h:10-17 states it; src/archive.cpp:45-55, 57-106 and
// src/prepare.cpp:12-21 fill it in), for a farm of many workers per GPU:
//
//   u32  n_blocks                      at offset 0, known only at the end
//   meta u16 hlen | first header | FreqTable<256,4> (3076 B) | FreqTable<8192,64> (1 081 348 B)
//   block x n_blocks, in the order their space was claimed:
//        u32 total | u32 n_records
//        u32 orig | u32 csize | bytes          readlens
//        u32 orig | u32 csize | bytes          n_count
//        u32 orig | u32 csize | bytes          n_pos
//        u32 csize | bytes                     seq     (the FSE stream, bit-identical to the reference's)
//        u32 csize | bytes                     qual
//        per header field: STRING  3 x (u32 orig | u32 csize | bytes)  flags, content, lengths
//                          NUMERIC 1 x (u32 orig | u32 csize | bytes)  content
//   index: n_blocks x { i64 offset; u32 idx; 4 zero bytes }, ascending offsets
// (all integers little-endian).  Byte-compatible with the reference for every field except the
// csize/bytes of the misc streams, which hold this library's own coder's output instead of libbsc's.
//
// What is different from the reference's classes of the same names (src/archive.h:18-96,
// src/fastq_io.h:10-59), and why -- its reader parses every chunk and its archive and writer move
// every byte under ONE mutex each (src/fastq_io.cpp:29-52, src/archive.cpp:57-106,
// src/fastq_io.cpp:131-143), which is what bounds a farm whose coder takes 3 ms per block:
//   * files are positional (pread / pwrite on a descriptor): no cursor, hence nothing to lock around
//     the bytes.  A block is serialised into ONE buffer from a field list (blockFields: the one
//     place that knows the order; writer and reader both walk it), claims its space with one atomic
//     add and goes out with one pwrite; a block comes in with one pread of its extent and is taken
//     apart by a bounds-checked cursor.
//   * FastqReader only FINDS the end of a chunk under its lock -- a backwards search for the last
//     complete record in the last megabyte of the range -- the bulk of the chunk is read outside,
//     and the chunk goes to the GPU unparsed (fqgpu_encode_begin builds the record table).
//   * FastqWriter knows where every chunk goes (the archive's blocks carry their original sizes):
//     chunks are written wherever they belong the moment they are ready; no worker waits for another.
//   * abort(): a worker that fails stops the others at their next block instead of leaving them waiting.
#pragma once

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <filesystem>
#include <mutex>
#include <system_error>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include "workspace.hpp"

namespace fqcomp28 {

using path_t = std::filesystem::path;

/** A file addressed by position; any number of threads. */
class PosFile {
public:
  enum class Mode { Read, Create };
  PosFile(const path_t &p, Mode m) : path_(p.string()) {
    fd_ = m == Mode::Read ? ::open(path_.c_str(), O_RDONLY) : ::open(path_.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (fd_ < 0) fail("open");
  }
  ~PosFile() { if (fd_ >= 0) ::close(fd_); }
  PosFile(const PosFile &) = delete;
  PosFile &operator=(const PosFile &) = delete;

  [[nodiscard]] uint64_t size() const {
    struct stat st;
    if (::fstat(fd_, &st) != 0) fail("fstat");
    return static_cast<uint64_t>(st.st_size);
  }
  /** exactly n bytes from off, or an exception */
  void readAt(uint64_t off, void *dst, std::size_t n) const {
    char *p = static_cast<char *>(dst);
    while (n) {
      const ssize_t got = ::pread(fd_, p, n, static_cast<off_t>(off));
      if (got < 0 && errno == EINTR) continue;
      if (got <= 0) throw std::runtime_error(path_ + ": file ends inside a read of " + std::to_string(n) + " bytes at " + std::to_string(off));
      p += got; off += static_cast<uint64_t>(got); n -= static_cast<std::size_t>(got);
    }
  }
  void writeAt(uint64_t off, const void *src, std::size_t n) {
    const char *p = static_cast<const char *>(src);
    while (n) {
      const ssize_t put = ::pwrite(fd_, p, n, static_cast<off_t>(off));
      if (put < 0 && errno == EINTR) continue;
      if (put <= 0) fail("pwrite");
      p += put; off += static_cast<uint64_t>(put); n -= static_cast<std::size_t>(put);
    }
  }
  void resize(uint64_t n) { if (::ftruncate(fd_, static_cast<off_t>(n)) != 0) fail("ftruncate"); }

private:
  [[noreturn]] void fail(const char *what) const { throw std::system_error(errno, std::generic_category(), path_ + ": " + what); }
  std::string path_;
  int fd_ = -1;
};

namespace blockfmt {
constexpr std::size_t INDEX_ENTRY = 16;  // i64 offset | u32 idx | 4 zero bytes

/** The sized fields of a block in file order, each as f(orig, bytes): orig points at the u32 in
 *  front of the field's size, or is null for the two FSE streams, which carry no original size. */
template <class Buffers, class F> void blockFields(Buffers &cb, const headers::HeaderFormatSpeciciation &fmt, F &&f) {
  uint32_t *const none = nullptr;
  f(&cb.original_size.readlens, cb.compressed_readlens);
  f(&cb.original_size.n_count, cb.compressed_n_count);
  f(&cb.original_size.n_pos, cb.compressed_n_pos);
  f(none, cb.seq);
  f(none, cb.qual);
  for (std::size_t i = 0; i < fmt.n_fields(); ++i) {
    auto &sizes = cb.original_size.header_fields[i];
    auto &streams = cb.compressed_header_fields[i];
    if (fmt.field_types[i] == headers::FieldType::STRING) {
      f(&sizes.isDifferentFlag, streams.isDifferentFlag);
      f(&sizes.content, streams.content);
      f(&sizes.contentLength, streams.contentLength);
    } else {
      f(&sizes.content, streams.content);
    }
  }
}

inline void put32(uint8_t *&p, uint32_t v) { std::memcpy(p, &v, 4); p += 4; }

/** read side: nothing is taken from beyond `end` */
struct Cursor {
  const uint8_t *p, *end;
  uint32_t u32() {
    if (end - p < 4) throw std::runtime_error("archive block: truncated");
    uint32_t v;
    std::memcpy(&v, p, 4);
    p += 4;
    return v;
  }
  template <class Bytes> void bytes(Bytes &dst) {
    const uint32_t n = u32();
    if (static_cast<std::size_t>(end - p) < n) throw std::runtime_error("archive block: a field runs past the block");
    dst.resize(n);
    if (n) std::memcpy(dst.data(), p, n);
    p += n;
  }
};
}  // namespace blockfmt

/** The archive's own view of an index entry (16 bytes in the file) */
struct BlockRef {
  uint64_t offset = 0, end = 0;  // extent of the block in the file
  uint32_t idx = 0;              // position of its chunk in the input
};

class Archive {
public:
  /** an existing archive, to read */
  explicit Archive(const path_t &archive_path) : file_(archive_path, PosFile::Mode::Read) { load(); }
  /** a new archive whose tables come from the first sample_size_bytes of a FASTQ file */
  Archive(const path_t &archive_path, const path_t &file_to_gather_meta, std::size_t sample_size_bytes, int device = 0);
  /** ... or are handed over */
  Archive(const path_t &archive_path, DatasetMeta meta) : file_(archive_path, PosFile::Mode::Create), meta_(std::move(meta)) {
    std::vector<uint8_t> head(sizeof(uint32_t), 0);  // the block count: filled in by writeIndex
    meta_.appendTo(head);
    file_.writeAt(0, head.data(), head.size());
    claim_.store(head.size());
  }

  /** Thread-safe; concurrent callers write concurrently. */
  void writeBlock(const CompressedBuffersDst &cb) {
    const auto &fmt = meta_.header_fmt;
    if (cb.compressed_header_fields.size() != fmt.n_fields() || cb.original_size.header_fields.size() != fmt.n_fields())
      throw std::invalid_argument("writeBlock: header field streams do not match the archive's format");
    std::size_t bytes = 2 * sizeof(uint32_t);
    blockfmt::blockFields(cb, fmt, [&](const uint32_t *orig, const auto &data) {
      if (data.size() > 0xFFFFFFFFull) throw std::length_error("writeBlock: a field of 4 GiB or more");
      bytes += (orig ? 8 : 4) + data.size();
    });
    BlockRef ref;
    ref.offset = claim_.fetch_add(bytes);
    ref.end = ref.offset + bytes;
    ref.idx = cb.chunk_idx;
    // The two FSE streams are nine tenths of a block: they go out from where they lie.  Everything else -- the size
    // words, the side streams, the header fields -- is gathered into a small image in between (round 3 built ONE image
    // of the whole block: 90 MB allocated, zeroed and copied per block, a third of the 34 ms a block's write took).
    constexpr std::size_t DIRECT = 256u << 10;
    std::vector<uint8_t> image;
    image.reserve(bytes < (8u << 20) ? bytes : (8u << 20));
    uint64_t at = ref.offset;
    const auto flush = [&] {
      if (image.empty()) return;
      file_.writeAt(at, image.data(), image.size());
      at += image.size();
      image.clear();
    };
    const auto word = [&](uint32_t v) {
      uint8_t b[4];
      uint8_t *p = b;
      blockfmt::put32(p, v);
      image.insert(image.end(), b, b + 4);
    };
    word(cb.original_size.total);
    word(cb.original_size.n_records);
    blockfmt::blockFields(cb, fmt, [&](const uint32_t *orig, const auto &data) {
      if (orig) word(*orig);
      word(static_cast<uint32_t>(data.size()));
      const auto *bytes_of = reinterpret_cast<const uint8_t *>(data.data());
      if (data.size() >= DIRECT) {
        flush();
        file_.writeAt(at, bytes_of, data.size());
        at += data.size();
      } else if (!data.empty()) {
        image.insert(image.end(), bytes_of, bytes_of + data.size());
      }
    });
    flush();
    if (at != ref.end) throw std::logic_error("writeBlock: the block's fields do not add up to its size");
    const std::lock_guard<std::mutex> guard(index_mutex_);
    index_.push_back(ref);
  }

  /** Thread-safe; blocks are handed out in the order of the input file.  false: no block left (or abort()). */
  bool readBlock(CompressedBuffersSrc &cb) {
    cb.clear();
    if (aborted_.load()) return false;
    const std::size_t k = next_.fetch_add(1);
    if (k >= index_.size()) return false;
    const BlockRef &ref = index_[k];
    std::vector<uint8_t> image(ref.end - ref.offset);
    file_.readAt(ref.offset, image.data(), image.size());
    blockfmt::Cursor cur{image.data(), image.data() + image.size()};
    const auto &fmt = meta_.header_fmt;
    cb.chunk_idx = ref.idx;
    cb.original_size.total = cur.u32();
    cb.original_size.n_records = cur.u32();
    cb.original_size.header_fields.assign(fmt.n_fields(), {});
    cb.compressed_header_fields.resize(fmt.n_fields());
    cb.header_fields.resize(fmt.n_fields());
    blockfmt::blockFields(cb, fmt, [&](uint32_t *orig, auto &data) {
      if (orig) *orig = cur.u32();
      cur.bytes(data);
    });
    return true;
  }

  /** the index behind the last block, then the block count at offset 0 */
  void writeIndex() {
    const std::lock_guard<std::mutex> guard(index_mutex_);
    if (index_.size() > 0xFFFFFFFFull) throw std::length_error("writeIndex: too many blocks");
    std::sort(index_.begin(), index_.end(), [](const BlockRef &a, const BlockRef &b) { return a.offset < b.offset; });
    std::vector<uint8_t> tail(index_.size() * blockfmt::INDEX_ENTRY, 0);
    for (std::size_t i = 0; i < index_.size(); ++i) {
      const int64_t off = static_cast<int64_t>(index_[i].offset);
      std::memcpy(tail.data() + i * blockfmt::INDEX_ENTRY, &off, 8);
      std::memcpy(tail.data() + i * blockfmt::INDEX_ENTRY + 8, &index_[i].idx, 4);
    }
    file_.writeAt(claim_.load(), tail.data(), tail.size());
    const uint32_t n = static_cast<uint32_t>(index_.size());
    file_.writeAt(0, &n, sizeof(n));
  }
  void flush() {}  // (nothing is buffered on this side of the descriptor)
  /** readBlock hands out nothing more: a worker has failed */
  void abort() { aborted_.store(true); }

  const DatasetMeta &meta() const { return meta_; }
  [[nodiscard]] std::size_t indexBytes() const { return sizeof(uint32_t) + index_.size() * blockfmt::INDEX_ENTRY; }
  [[nodiscard]] std::size_t nBlocks() const { return index_.size(); }
  /** (offset, idx) of every block, in the order blocks are handed out -- for tests */
  [[nodiscard]] std::vector<std::pair<int64_t, uint32_t>> indexEntries() const {
    std::vector<std::pair<int64_t, uint32_t>> v;
    for (const BlockRef &b : index_) v.emplace_back(static_cast<int64_t>(b.offset), b.idx);
    return v;
  }
  /** Where every chunk starts in the restored file (entry k: chunk k; one more entry: the file's
   *  size): the first word of a block is the size of its chunk. */
  [[nodiscard]] std::vector<uint64_t> chunkOffsets() const {
    std::vector<uint64_t> at(index_.size() + 1, 0);
    for (std::size_t k = 0; k < index_.size(); ++k) {
      uint32_t total = 0;
      file_.readAt(index_[k].offset, &total, sizeof(total));
      at[k + 1] = at[k] + total;
    }
    return at;
  }

private:
  void load() {
    const uint64_t size = file_.size();
    uint8_t head[6];
    file_.readAt(0, head, sizeof(head));
    uint32_t n_blocks;
    uint16_t hlen;
    std::memcpy(&n_blocks, head, 4);
    std::memcpy(&hlen, head + 4, 2);
    std::vector<uint8_t> m(sizeof(hlen) + hlen + FQGPU_SEQ_FT_BYTES + FQGPU_QUAL_FT_BYTES);
    file_.readAt(sizeof(uint32_t), m.data(), m.size());
    meta_ = DatasetMeta::fromBytes(m.data(), m.size());
    const uint64_t data_begin = sizeof(uint32_t) + m.size(), index_bytes = static_cast<uint64_t>(n_blocks) * blockfmt::INDEX_ENTRY;
    if (size < data_begin + index_bytes) throw std::runtime_error("archive: the index does not fit the file (truncated?)");
    const uint64_t index_begin = size - index_bytes;
    std::vector<uint8_t> tail(index_bytes);
    if (index_bytes) file_.readAt(index_begin, tail.data(), tail.size());
    index_.resize(n_blocks);
    for (std::size_t i = 0; i < n_blocks; ++i) {
      int64_t off;
      std::memcpy(&off, tail.data() + i * blockfmt::INDEX_ENTRY, 8);
      std::memcpy(&index_[i].idx, tail.data() + i * blockfmt::INDEX_ENTRY + 8, 4);
      if (off < static_cast<int64_t>(data_begin) || static_cast<uint64_t>(off) + 2 * sizeof(uint32_t) > index_begin)
        throw std::runtime_error("archive: a block offset lies outside the data section");
      index_[i].offset = static_cast<uint64_t>(off);
    }
    // a block ends where the next one in the file begins
    std::vector<std::size_t> by_offset(n_blocks);
    for (std::size_t i = 0; i < n_blocks; ++i) by_offset[i] = i;
    std::sort(by_offset.begin(), by_offset.end(), [&](std::size_t a, std::size_t b) { return index_[a].offset < index_[b].offset; });
    for (std::size_t r = 0; r < n_blocks; ++r)
      index_[by_offset[r]].end = r + 1 < n_blocks ? index_[by_offset[r + 1]].offset : index_begin;
    // and blocks are read in the order of their chunks in the input
    std::stable_sort(index_.begin(), index_.end(), [](const BlockRef &a, const BlockRef &b) { return a.idx < b.idx; });
    claim_.store(index_begin);
  }

  PosFile file_;
  DatasetMeta meta_;
  std::vector<BlockRef> index_;
  std::mutex index_mutex_;
  std::atomic<uint64_t> claim_{0};     // writing: first byte behind the blocks so far
  std::atomic<std::size_t> next_{0};   // reading: next block to hand out
  std::atomic<bool> aborted_{false};
};

/** Offset behind the last complete 4-line record that ends inside [buf, buf + n); 0 if there is none.
 *  `starts_at_line`: buf begins at the start of a line (else the bytes before its first newline belong
 *  to a line that began earlier and are skipped).  A record is recognised from its shape alone --
 *  '@' line, line, '+' line, line of the second line's length -- which is unambiguous for FASTQ whose
 *  sequence lines hold bases: shifted by one, two or three lines the '@' / '+' tests look at a
 *  sequence line (the coder refuses anything but ACGTN there, fqgpu.h). */
inline std::size_t lastRecordEnd(const char *buf, std::size_t n, bool starts_at_line) {
  // line ends, from the back: e[0] is the last newline, e[1] the one before ...
  std::size_t e[5];
  int have = 0;
  std::size_t scan = n;
  auto pull = [&]() -> bool {  // one more newline towards the front
    while (scan > 0) {
      const void *hit = ::memrchr(buf, '\n', scan);
      if (!hit) { scan = 0; return false; }
      scan = static_cast<std::size_t>(static_cast<const char *>(hit) - buf);
      return true;
    }
    return false;
  };
  for (;;) {
    // window of five newlines: lines 1..4 of a candidate record lie between them
    while (have < 5) {
      if (!pull()) break;
      e[have++] = scan;
    }
    if (have < 4) return 0;
    const bool front_known = have == 5;
    if (!front_known && !starts_at_line) return 0;  // the first line's start is not in the buffer
    const std::size_t l1 = front_known ? e[4] + 1 : 0, l2 = e[3] + 1, l3 = e[2] + 1, l4 = e[1] + 1;
    if (buf[l1] == '@' && buf[l3] == '+' && e[2] - l2 == e[0] - l4) return e[0] + 1;
    if (!front_known) return 0;
    // slide the window one line towards the front
    for (int i = 0; i < 4; ++i) e[i] = e[i + 1];
    have = 4;
  }
}

/** Chunks of a FASTQ file: about reading_size bytes each, whole records only, the next chunk begins
 *  where the last one ended -- the blocks the reference's reader cuts (src/fastq_io.cpp:23-65), found
 *  without parsing them.  The chunk arrives UNPARSED (records empty): fqgpu_encode_begin builds the
 *  record table on the GPU.  Thread-safe; only the search for the chunk's end is serial. */
class FastqReader {
public:
  FastqReader(const path_t &mates1, std::size_t reading_size)
      : file_(mates1, PosFile::Mode::Read), size_(file_.size()), reading_size_(reading_size) {
    if (reading_size_ == 0) throw std::invalid_argument("FastqReader: empty reading size");
  }

  /** @return false when the file is used up (a partial record at its very end is dropped) or after abort() */
  bool readNextChunk(FastqChunk &chunk) {
    chunk.clear();
    uint64_t begin, end;
    {
      const std::lock_guard<std::mutex> guard(mutex_);
      if (aborted_ || next_begin_ >= size_) return false;
      begin = next_begin_;
      const uint64_t limit = std::min<uint64_t>(size_, begin + reading_size_);
      end = begin + findEnd(begin, limit);
      if (end == begin) {
        if (limit == size_) { next_begin_ = size_; return false; }
        throw std::runtime_error("FastqReader: reading size smaller than one record");
      }
      next_begin_ = end;
      chunk.idx = chunks_read_++;
    }
    chunk.raw_data.resize(end - begin);
    file_.readAt(begin, chunk.raw_data.data(), chunk.raw_data.size());
    return true;
  }
  void abort() {
    const std::lock_guard<std::mutex> guard(mutex_);
    aborted_ = true;
  }

private:
  /** bytes of [begin, limit) up to the end of its last complete record */
  std::size_t findEnd(uint64_t begin, uint64_t limit) {
    for (std::size_t window = std::size_t(1) << 20;; window <<= 2) {
      const uint64_t from = limit - begin > window ? limit - window : begin;
      tail_.resize(limit - from);
      file_.readAt(from, tail_.data(), tail_.size());
      const std::size_t rel = lastRecordEnd(tail_.data(), tail_.size(), from == begin);
      if (rel) return static_cast<std::size_t>(from - begin) + rel;
      if (from == begin) return 0;
    }
  }

  PosFile file_;
  const uint64_t size_;
  const std::size_t reading_size_;
  std::mutex mutex_;
  uint64_t next_begin_ = 0;
  unsigned chunks_read_ = 0;
  bool aborted_ = false;
  std::vector<char> tail_;
};

/** The restored FASTQ file.  chunk_offsets (Archive::chunkOffsets) says where every chunk goes, so
 *  chunks are written as they come, by any number of threads, in any order. */
class FastqWriter {
public:
  /** The chunks are written wherever they belong the moment they are ready, into `<mates1>.part` at its final size;
   *  flush() -- every chunk is there -- gives the file its name.  A restore that fails (damaged block, farm aborted)
   *  therefore leaves NO file under the requested name instead of a full-size one with zero-filled holes
   *  (the reference's ordered writer left a short prefix, src/fastq_io.cpp:127-136): the partial file is removed. */
  FastqWriter(const path_t &mates1, std::vector<uint64_t> chunk_offsets)
      : final_(mates1), part_(mates1.string() + ".part"), file_(part_, PosFile::Mode::Create), at_(std::move(chunk_offsets)) {
    if (at_.empty()) at_.push_back(0);
    file_.resize(at_.back());
  }
  ~FastqWriter() {
    if (!done_) { std::error_code ec; std::filesystem::remove(part_, ec); }
  }
  void writeChunk(const FastqChunk &chunk) {
    if (chunk.idx + 1 >= at_.size() || at_[chunk.idx + 1] - at_[chunk.idx] != chunk.raw_data.size())
      throw std::runtime_error("FastqWriter: chunk " + std::to_string(chunk.idx) + " does not have the size the archive recorded");
    file_.writeAt(at_[chunk.idx], chunk.raw_data.data(), chunk.raw_data.size());
  }
  void flush() {
    if (done_) return;
    std::filesystem::rename(part_, final_);
    done_ = true;
  }

private:
  path_t final_, part_;
  PosFile file_;
  std::vector<uint64_t> at_;
  bool done_ = false;
};

/** The decode indexes of an archive's blocks, in a file of their own beside it (`<archive>.fqx`): an extension the
 *  reference knows nothing of, so the .fqc file stays exactly what the reference reads and writes, and an archive
 *  without the file decodes at the format's own pace.  A block's entry is found by the position of its chunk in the
 *  input (CompressedBuffers::chunk_idx), and is only handed out if its checksum holds: the decoder takes the states
 *  in a snapshot as they are (api.hip: index_accept).
 *    "FQX1" | entries as they were claimed: { u32 chunk_idx, u32 0, u64 seq bytes, u64 qual bytes, u64 fnv1a-64 of the
 *    two indexes } seq index | qual index ... | table: n x { u64 offset of the entry } | u64 n | u64 size of the
 *    archive | u64 fnv1a-64 of the archive's first 64 KiB (block count, first header, tables) | "FQX1"
 *  The last two words tie the file to ITS archive: a file left behind by an earlier archive of the same name is
 *  recognised (belongsTo) and not used. */
class DecodeIndexFile {
public:
  static path_t pathFor(const path_t &archive_path) { return path_t(archive_path.string() + ".fqx"); }
  static constexpr uint32_t MAGIC = 0x31585146u;  // "FQX1"
  static constexpr std::size_t HEAD = 32, TAIL = 28;  // entry head; n, archive size, archive hash, magic

  struct Identity {
    uint64_t size = 0, hash = 0;
    friend bool operator==(const Identity &a, const Identity &b) { return a.size == b.size && a.hash == b.hash; }
  };
  /** of a finished archive file */
  static Identity identityOf(const path_t &archive_path) {
    const PosFile f(archive_path, PosFile::Mode::Read);
    Identity id;
    id.size = f.size();
    std::vector<std::byte> head(static_cast<std::size_t>(std::min<uint64_t>(id.size, 64u << 10)));
    if (!head.empty()) f.readAt(0, head.data(), head.size());
    id.hash = checksum(head, {});
    return id;
  }
  [[nodiscard]] bool belongsTo(const Identity &archive) const { return identity_ == archive; }

  /** to write (Create) or to read an existing one */
  DecodeIndexFile(const path_t &p, PosFile::Mode m) : file_(p, m) {
    if (m == PosFile::Mode::Create) {
      file_.writeAt(0, &MAGIC, 4);
      claim_.store(4);
      return;
    }
    const uint64_t size = file_.size();
    uint32_t magic = 0;
    uint64_t n = 0;
    if (size < 4 + TAIL) throw std::runtime_error("decode index file: too short");
    file_.readAt(0, &magic, 4);
    if (magic != MAGIC) throw std::runtime_error("decode index file: not one");
    file_.readAt(size - 4, &magic, 4);
    file_.readAt(size - TAIL, &n, 8);
    file_.readAt(size - TAIL + 8, &identity_.size, 8);
    file_.readAt(size - TAIL + 16, &identity_.hash, 8);
    if (magic != MAGIC || n > (size - 4 - TAIL) / 8) throw std::runtime_error("decode index file: damaged trailer (was it closed?)");
    std::vector<uint64_t> offsets(n);
    end_of_entries_ = size - TAIL - 8 * n;
    if (n) file_.readAt(end_of_entries_, offsets.data(), 8 * n);
    at_.assign(n, 0);
    for (const uint64_t off : offsets) {
      if (end_of_entries_ < 4 + HEAD || off < 4 || off > end_of_entries_ - HEAD) throw std::runtime_error("decode index file: entry outside the file");
      uint32_t idx = 0;
      file_.readAt(off, &idx, 4);
      if (idx >= n) throw std::runtime_error("decode index file: entry of a chunk the file cannot hold");
      at_[idx] = off;
    }
  }

  /** Thread-safe; concurrent callers write concurrently. */
  void put(const CompressedBuffers &cb) {
    const auto &s = cb.decode_index[0], &q = cb.decode_index[1];
    uint8_t head[HEAD];
    const uint32_t idx = cb.chunk_idx, zero = 0;
    const uint64_t ns = s.size(), nq = q.size(), sum = checksum(s, q);
    std::memcpy(head, &idx, 4); std::memcpy(head + 4, &zero, 4);
    std::memcpy(head + 8, &ns, 8); std::memcpy(head + 16, &nq, 8); std::memcpy(head + 24, &sum, 8);
    const uint64_t off = claim_.fetch_add(HEAD + ns + nq);
    file_.writeAt(off, head, HEAD);
    if (ns) file_.writeAt(off + HEAD, s.data(), ns);
    if (nq) file_.writeAt(off + HEAD + ns, q.data(), nq);
    const std::lock_guard<std::mutex> guard(m_);
    offsets_.push_back(off);
  }
  /** after the last put and when the archive is complete: the table and the trailer */
  void close(const Identity &archive) {
    const uint64_t n = offsets_.size(), at = claim_.load();
    if (n) file_.writeAt(at, offsets_.data(), 8 * n);
    file_.writeAt(at + 8 * n, &n, 8);
    file_.writeAt(at + 8 * n + 8, &archive.size, 8);
    file_.writeAt(at + 8 * n + 16, &archive.hash, 8);
    file_.writeAt(at + 8 * n + 24, &MAGIC, 4);
  }

  /** the indexes of chunk cb.chunk_idx into cb.decode_index; false: the file has none for it.  Thread-safe. */
  bool get(CompressedBuffers &cb) const {
    cb.decode_index[0].clear(); cb.decode_index[1].clear();
    if (cb.chunk_idx >= at_.size() || at_[cb.chunk_idx] == 0) return false;
    const uint64_t off = at_[cb.chunk_idx];
    uint8_t head[HEAD];
    file_.readAt(off, head, HEAD);
    uint32_t idx = 0;
    uint64_t ns = 0, nq = 0, sum = 0;
    std::memcpy(&idx, head, 4); std::memcpy(&ns, head + 8, 8); std::memcpy(&nq, head + 16, 8); std::memcpy(&sum, head + 24, 8);
    if (idx != cb.chunk_idx || ns > end_of_entries_ || nq > end_of_entries_ || off + HEAD + ns + nq > end_of_entries_)
      throw std::runtime_error("decode index file: damaged entry of chunk " + std::to_string(cb.chunk_idx));
    cb.decode_index[0].resize(ns);
    cb.decode_index[1].resize(nq);
    if (ns) file_.readAt(off + HEAD, cb.decode_index[0].data(), ns);
    if (nq) file_.readAt(off + HEAD + ns, cb.decode_index[1].data(), nq);
    if (checksum(cb.decode_index[0], cb.decode_index[1]) != sum)
      throw std::runtime_error("decode index file: checksum of chunk " + std::to_string(cb.chunk_idx) + " does not hold");
    return true;
  }

private:
  static uint64_t checksum(const std::vector<std::byte> &a, const std::vector<std::byte> &b) {
    // FNV-1a over 8-byte words (the indexes are u16 / u64 tables: a multiple of 8 bytes up to a short tail)
    uint64_t h = 0xcbf29ce484222325ull;
    for (const auto *v : {&a, &b}) {
      const std::size_t n = v->size();
      const auto *p = reinterpret_cast<const uint8_t *>(v->data());
      std::size_t i = 0;
      for (; i + 8 <= n; i += 8) {
        uint64_t w;
        std::memcpy(&w, p + i, 8);
        h = (h ^ w) * 0x100000001b3ull;
      }
      for (; i < n; ++i) h = (h ^ p[i]) * 0x100000001b3ull;
      h = (h ^ n) * 0x100000001b3ull;
    }
    return h;
  }
  PosFile file_;
  std::atomic<uint64_t> claim_{0};
  std::mutex m_;
  std::vector<uint64_t> offsets_;   // writing: entries so far
  std::vector<uint64_t> at_;        // reading: entry of chunk idx (0 = none: offset 0 holds the magic)
  uint64_t end_of_entries_ = 0;
  Identity identity_;               // reading: the archive the file was written for
};

/** Dataset analysis (src/prepare.cpp:42-47): the tables of the first sample_size_bytes of the file, on the GPU */
inline DatasetMeta analyzeDataset(const path_t &fastq_file, std::size_t sample_size_bytes, int device = 0) {
  FastqChunk chunk;
  FastqReader reader(fastq_file, sample_size_bytes);
  if (!reader.readNextChunk(chunk)) throw std::runtime_error("analyzeDataset: empty input");
  parseRecords(chunk);
  return DatasetMeta(chunk, device);
}

inline Archive::Archive(const path_t &archive_path, const path_t &file_to_gather_meta, std::size_t sample_size_bytes, int device)
    : Archive(archive_path, analyzeDataset(file_to_gather_meta, sample_size_bytes, device)) {}

}  // namespace fqcomp28
