// The reference's hot-path tests restated against the drop-in shim
// (reference test/workspace_test.cpp:45-69 "Workspace::encodeChunk",
//  test/fse_sequence_test.cpp:17-50, test/fse_quality_test.cpp:17-49: all round-trips).
// Beyond the reference's round trips the shim's streams are compared BYTE FOR BYTE with the CPU
// oracle's (written to files by the Python test): cbs.seq / cbs.qual / readlens / n_count / n_pos.
// Usage: workspace_test <fastq> <oracle stream prefix> [<fastq> <prefix> ...]   (needs a GPU)
#include "../../fqcomp28_amd/csrc/workspace.hpp"

#include <cstdio>
#include <sstream>
#include <fstream>
#include <iterator>

using namespace fqcomp28;

static FastqChunk loadFastqFileContents(const char *path) {  // test/test_utils.cpp:21-26
  FastqChunk chunk;
  std::ifstream ifs(path, std::ios::binary);
  chunk.raw_data.assign(std::istreambuf_iterator<char>(ifs), std::istreambuf_iterator<char>());
  parseRecords(chunk);
  return chunk;
}

static CompressedBuffersSrc convertToSrcBuffers(CompressedBuffersDst &&in) {  // test/test_utils.h:27-48
  CompressedBuffersSrc src;
  src.original_size = in.original_size;
  src.seq = std::move(in.seq);
  src.qual = std::move(in.qual);
  src.n_count = std::move(in.n_count);
  src.index.n_count = src.n_count.size();
  src.n_pos = std::move(in.n_pos);
  src.index.n_pos = src.n_pos.size();
  src.readlens = std::move(in.readlens);
  src.compressed_readlens = std::move(in.compressed_readlens);
  src.compressed_n_count = std::move(in.compressed_n_count);
  src.compressed_n_pos = std::move(in.compressed_n_pos);
  src.compressed_header_fields = std::move(in.compressed_header_fields);
  src.chunk_idx = in.chunk_idx;
  src.header_fields.resize(in.header_fields.size());
  for (std::size_t i = 0; i < in.header_fields.size(); ++i) {
    src.header_fields[i].isDifferentFlag = std::move(in.header_fields[i].isDifferentFlag);
    src.header_fields[i].content = std::move(in.header_fields[i].content);
    src.header_fields[i].contentLength = std::move(in.header_fields[i].contentLength);
  }
  return src;
}

#define CHECK(x) do { if (!(x)) { std::printf("CHECK failed: %s (%s:%d)\n", #x, __FILE__, __LINE__); return 1; } } while (0)

static std::vector<std::byte> fileBytes(const std::string &path) {
  std::ifstream ifs(path, std::ios::binary);
  std::vector<char> c((std::istreambuf_iterator<char>(ifs)), std::istreambuf_iterator<char>());
  std::vector<std::byte> b(c.size());
  std::memcpy(b.data(), c.data(), c.size());
  return b;
}
template <class A, class B> static bool sameBytes(const A &a, const B &b) {
  return a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size()) == 0;
}

static int encodeChunkRoundTrip(const char *path, const std::string &oracle_prefix) {
  FastqChunk chunk_in = loadFastqFileContents(path);
  const FastqData original = chunk_in.raw_data;
  const DatasetMeta meta(chunk_in);
  CompressionWorkspace cwksp(&meta);
  DecompressionWorkspace dwksp(&meta);
  CompressedBuffersDst cbs;

  // a reused buffer set: the second encode appends to n_count / n_pos
  for (int pass = 0; pass < 2; ++pass) {
    chunk_in.raw_data = original;
    cwksp.encodeChunk(chunk_in, cbs);
    if (pass == 0) {  // fresh buffers: exactly the oracle's bytes (tables from the same chunk on both sides)
      CHECK(sameBytes(cbs.seq, fileBytes(oracle_prefix + ".seq")));
      CHECK(sameBytes(cbs.qual, fileBytes(oracle_prefix + ".qual")));
      CHECK(sameBytes(cbs.readlens, fileBytes(oracle_prefix + ".readlens")));
      CHECK(sameBytes(cbs.n_count, fileBytes(oracle_prefix + ".n_count")));
      CHECK(sameBytes(cbs.n_pos, fileBytes(oracle_prefix + ".n_pos")));
      CHECK(std::memcmp(chunk_in.raw_data.data(), fileBytes(oracle_prefix + ".raw_after").data(), original.size()) == 0);
    }
  }
  // the misc streams went through compressMiscBuffers: compressed twins within the bound, sizes recorded
  CHECK(cbs.compressed_readlens.size() <= cbs.readlens.size() + 28 && !cbs.compressed_readlens.empty());
  CHECK(cbs.original_size.readlens == cbs.readlens.size() && cbs.original_size.n_pos == cbs.n_pos.size());
  CHECK(cbs.compressed_header_fields.size() == meta.header_fmt.n_fields());
  CHECK(cbs.seq.size() <= Workspace::compressBoundSequence(chunk_in.tot_reads_length));
  CHECK(cbs.n_count.size() == 2 * chunk_in.records.size() * sizeof(readlen_t));

  // the header fields went through the tokeniser: one stream set per field of the first header,
  // flags for every record of a STRING field, four bytes per record of a NUMERIC one
  const auto fmt = headers::HeaderFormatSpeciciation::fromHeader(chunk_in.records.front().header());
  CHECK(fmt == meta.header_fmt);
  CHECK(cbs.header_fields.size() == fmt.n_fields());
  CHECK(cbs.original_size.header_fields.size() == fmt.n_fields());
  for (std::size_t i = 0; i < fmt.n_fields(); ++i) {
    const auto &f = cbs.header_fields[i];
    CHECK(f.originalSizes() == cbs.original_size.header_fields[i]);
    if (fmt.field_types[i] == headers::FieldType::STRING) {
      CHECK(f.isDifferentFlag.size() == chunk_in.records.size());
    } else {
      CHECK(f.isDifferentFlag.empty() && f.contentLength.empty());
      CHECK(f.content.size() == chunk_in.records.size() * sizeof(headers::numeric_t));
    }
  }

  // decodeChunk from the buffers alone: both passes (layout + headers on the host, reads on the GPU)
  FastqChunk chunk_out;
  chunk_out.raw_data.assign(7, 'x');  // stale contents are dropped
  CompressedBuffersSrc src = convertToSrcBuffers(std::move(cbs));
  // what decodeChunk gets from an archive is the compressed twins: wipe the plain copies
  src.readlens.clear(); src.n_count.clear(); src.n_pos.clear();
  for (auto &f : src.header_fields) f.clear();
  dwksp.decodeChunk(chunk_out, src);
  for (const auto &f : src.header_fields) {  // every stream fully consumed
    CHECK(f.index.isDifferentPos == f.isDifferentFlag.size());
    CHECK(f.index.contentPos == f.content.size());
    CHECK(f.index.contentLengthPos == f.contentLength.size());
  }

  CHECK(chunk_out.records.size() == chunk_in.records.size());
  FastqChunk ref;
  ref.raw_data = original;
  parseRecords(ref);
  for (std::size_t i = 0, E = ref.records.size(); i < E; ++i) {
    CHECK(ref.records[i].header() == chunk_out.records[i].header());
    CHECK(ref.records[i].seq() == chunk_out.records[i].seq());
    CHECK(ref.records[i].qual() == chunk_out.records[i].qual());
  }
  CHECK(chunk_out.raw_data == original);

  // DatasetMeta store/load (reference test/prepare_test.cpp): u16 length + first header + the two
  // FreqTable PODs, round trip and exact size
  {
    std::stringstream ss;
    DatasetMeta::storeToStream(meta, ss);
    const std::string bytes = ss.str();
    CHECK(bytes.size() == meta.size());
    CHECK(bytes.size() == 2 + meta.first_header.size() + 3076 + 1081348);
    CHECK(meta.first_header == ref.records.front().header());
    readlen_t hlen = 0;
    std::memcpy(&hlen, bytes.data(), 2);
    CHECK(hlen == meta.first_header.size());
    CHECK(std::memcmp(bytes.data() + 2 + hlen, meta.ft_seq.get(), 3076) == 0);
    const DatasetMeta back = DatasetMeta::loadFromStream(ss);
    CHECK(back == meta);
  }
  std::printf("ok %s: %zu records, seq %zu B, qual %zu B\n", path, ref.records.size(), src.seq.size(),
              src.qual.size());
  return 0;
}

int main(int argc, char **argv) {
  if (fqgpu_device_count() < 1) { std::printf("no GPU: the shim has no CPU fallback\n"); return 2; }
  int bad = 0;
  for (int i = 1; i + 1 < argc; i += 2) {
    try {
      bad += encodeChunkRoundTrip(argv[i], argv[i + 1]);
    } catch (const std::exception &e) {
      std::printf("exception on %s: %s\n", argv[i], e.what());
      bad++;
    }
  }
  return bad ? 1 : 0;
}
