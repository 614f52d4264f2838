// Host-only exerciser of the shim's container code (fqcomp28_amd/csrc/archive.hpp) -- no GPU call:
//   archive_tool copy <in.fqc> <out.fqc>      readBlock every block (index order) -> writeBlock -> writeIndex
//   archive_tool dump <in.fqc>                per block: idx, sizes, and the misc streams after
//                                             decompressMiscBuffers + header decoding, as hex / text
//   archive_tool chunks <in.fastq> <bytes>    FastqReader with that reading size: "chunk <idx> <bytes>" per chunk
//   archive_tool rejoin <in.fastq> <out.fastq> <bytes>   the same chunks through FastqWriter, LAST chunk first
//                                             (reference test/fastq_io_test.cpp:15-53: reader -> writer identity)
//   archive_tool threads <in.fqc> <out.fqc> <T> <rounds>   what the farm's workers share, from T threads at once (for
//                                             ThreadSanitizer): readBlock of one archive -> through a Gate of two ->
//                                             writeBlock + DecodeIndexFile::put of another, `rounds` times over; then
//                                             T threads read it back and check every block's index entry
//   archive_tool sidecheck <in.fqc>           the decode index file beside an archive: "foreign" if it was written for
//                                             another archive, else every block's entry read (checksums) and counted
// tests/test_archive.py drives it against oracle/fqc_archive.py (an independent Python reading of
// src/archive.h:10-17, src/archive.cpp:57-106).
#include "../../fqcomp28_amd/csrc/process.hpp"

#include <cstdio>
#include <cstdlib>

using namespace fqcomp28;

static CompressedBuffersDst toDst(CompressedBuffersSrc &&in) {
  CompressedBuffersDst d;
  d.original_size = in.original_size;
  d.chunk_idx = in.chunk_idx;
  d.seq = std::move(in.seq);
  d.qual = std::move(in.qual);
  d.compressed_readlens = std::move(in.compressed_readlens);
  d.compressed_n_count = std::move(in.compressed_n_count);
  d.compressed_n_pos = std::move(in.compressed_n_pos);
  d.compressed_header_fields = std::move(in.compressed_header_fields);
  return d;
}

template <class Bytes> static void hex(const char *name, const Bytes &v) {
  std::printf("%s %zu ", name, v.size());
  for (std::byte b : v) std::printf("%02x", static_cast<unsigned>(b));
  std::printf("\n");
}

int main(int argc, char **argv) {
  try {
    if (argc == 4 && std::string(argv[1]) == "copy") {
      Archive in(argv[2]);
      DatasetMeta meta(std::string_view(in.meta().first_header));
      std::memcpy(meta.ft_seq.get(), in.meta().ft_seq.get(), FQGPU_SEQ_FT_BYTES);
      std::memcpy(meta.ft_qual.get(), in.meta().ft_qual.get(), FQGPU_QUAL_FT_BYTES);
      Archive out(argv[3], std::move(meta));
      CompressedBuffersSrc cbs;
      std::size_t n = 0;
      while (in.readBlock(cbs)) { out.writeBlock(toDst(std::move(cbs))); ++n; }
      out.writeIndex();
      out.flush();
      std::printf("copied %zu blocks, index %zu bytes\n", n, out.indexBytes());
      return 0;
    }
    if (argc == 3 && std::string(argv[1]) == "dump") {
      Archive in(argv[2]);
      const auto &fmt = in.meta().header_fmt;
      std::printf("first_header %s\nfields %zu\nblocks %zu\n", in.meta().first_header.c_str(), fmt.n_fields(), in.nBlocks());
      for (const auto &e : in.indexEntries()) std::printf("index %lld %u\n", static_cast<long long>(e.first), e.second);
      CompressedBuffersSrc cbs;
      const headers::header_fields_t first = headers::fromHeader(in.meta().first_header, fmt);
      while (in.readBlock(cbs)) {
        DecompressionWorkspace::decompressMiscBuffers(cbs, fmt);
        std::printf("block %u total %u n_records %u seq %zu qual %zu\n", cbs.chunk_idx, cbs.original_size.total,
                    cbs.original_size.n_records, cbs.seq.size(), cbs.qual.size());
        hex("readlens", cbs.readlens);
        hex("n_count", cbs.n_count);
        hex("n_pos", cbs.n_pos);
        headers::header_fields_t prev = first;
        // headers are decoded one behind the other like decodeChunk lays them out: the string
        // fields of `prev` keep pointing at bytes that stay
        std::vector<char> arena(cbs.original_size.total + 1 + fmt.n_fields() * (headers::FIELDLEN_MAX + 1));
        char *dst = arena.data();
        for (uint32_t r = 0; r < cbs.original_size.n_records; ++r) {
          const unsigned n = headers::decodeHeader(dst, fmt, prev, cbs.header_fields);
          std::printf("h %.*s\n", static_cast<int>(n), dst);
          dst += n;
        }
      }
      return 0;
    }
    if ((argc == 4 && std::string(argv[1]) == "chunks") || ((argc == 5 || argc == 6) && std::string(argv[1]) == "rejoin")) {
      const bool rejoin = argv[1][0] == 'r';
      FastqReader reader(argv[2], static_cast<std::size_t>(std::atoll(argv[rejoin ? 4 : 3])));
      std::vector<FastqChunk> chunks;
      std::vector<uint64_t> at{0};
      for (;;) {
        chunks.emplace_back();
        if (!reader.readNextChunk(chunks.back())) { chunks.pop_back(); break; }
        if (!chunks.back().records.empty()) throw std::runtime_error("the reader hands out unparsed chunks");
        std::printf("chunk %u %zu\n", chunks.back().idx, chunks.back().raw_data.size());
        at.push_back(at.back() + chunks.back().raw_data.size());
      }
      if (rejoin) {
        FastqWriter writer(argv[3], at);
        for (std::size_t i = chunks.size(); i-- > 0;) writer.writeChunk(chunks[i]);
        if (argc > 5 && argv[5][0] == 'a') return 3;  // "abandon": leave without flush() -- no output file may remain
        writer.flush();
      }
      return 0;
    }
    if (argc == 3 && std::string(argv[1]) == "sidecheck") {
      Archive in(argv[2]);
      const DecodeIndexFile side(DecodeIndexFile::pathFor(argv[2]), PosFile::Mode::Read);
      if (!side.belongsTo(DecodeIndexFile::identityOf(argv[2]))) { std::printf("foreign\n"); return 0; }
      CompressedBuffersSrc cbs;
      std::size_t with = 0, without = 0, bytes = 0;
      while (in.readBlock(cbs)) {
        if (side.get(cbs)) { ++with; bytes += cbs.decode_index[0].size() + cbs.decode_index[1].size(); } else ++without;
      }
      std::printf("entries %zu missing %zu bytes %zu\n", with, without, bytes);
      return 0;
    }
    if (argc == 6 && std::string(argv[1]) == "threads") {
      const unsigned T = static_cast<unsigned>(std::atoi(argv[4])), rounds = static_cast<unsigned>(std::atoi(argv[5]));
      // a made-up decode index of a block: a function of its bytes, so that the readers can check what they get
      const auto fake_index = [](const CompressedBuffers &cb, int s) {
        std::vector<std::byte> v(64 + (cb.seq.size() + 7 * static_cast<std::size_t>(s)) % 4000);
        for (std::size_t i = 0; i < v.size(); ++i) v[i] = static_cast<std::byte>((i * 131 + cb.chunk_idx * 7 + cb.qual.size() + static_cast<std::size_t>(s)) & 0xFF);
        return v;
      };
      std::atomic<std::size_t> written{0}, checked{0};
      {
        Archive in(argv[2]);
        DatasetMeta meta(std::string_view(in.meta().first_header));
        std::memcpy(meta.ft_seq.get(), in.meta().ft_seq.get(), FQGPU_SEQ_FT_BYTES);
        std::memcpy(meta.ft_qual.get(), in.meta().ft_qual.get(), FQGPU_QUAL_FT_BYTES);
        const std::size_t per_round = in.nBlocks();
        Archive out(argv[3], std::move(meta));
        DecodeIndexFile side(DecodeIndexFile::pathFor(argv[3]), PosFile::Mode::Create);
        detail::Gate gate(2);
        std::atomic<unsigned> ticket{0};
        detail::runWorkers(T, [&](unsigned) {
          for (;;) {
            const unsigned k = ticket.fetch_add(1);
            if (k >= rounds) break;
            Archive again(argv[2]);  // (every round reads the input from its start)
            CompressedBuffersSrc cbs;
            for (;;) {
              {
                const detail::Gate::Pass pass(gate);
                if (!again.readBlock(cbs)) break;
              }
              CompressedBuffersDst d = toDst(std::move(cbs));
              d.chunk_idx = static_cast<uint32_t>(k * per_round + d.chunk_idx);
              d.decode_index[0] = fake_index(d, 0);
              d.decode_index[1] = fake_index(d, 1);
              out.writeBlock(d);
              side.put(d);
              written.fetch_add(1);
            }
          }
        });
        out.writeIndex();
        out.flush();
        side.close(DecodeIndexFile::identityOf(argv[3]));
      }
      Archive back(argv[3]);
      const DecodeIndexFile side(DecodeIndexFile::pathFor(argv[3]), PosFile::Mode::Read);
      if (!side.belongsTo(DecodeIndexFile::identityOf(argv[3]))) throw std::runtime_error("the index file does not know its archive");
      detail::runWorkers(T, [&](unsigned) {
        CompressedBuffersSrc cbs;
        while (back.readBlock(cbs)) {
          if (!side.get(cbs)) throw std::runtime_error("a block without its index entry");
          if (cbs.decode_index[0] != fake_index(cbs, 0) || cbs.decode_index[1] != fake_index(cbs, 1)) throw std::runtime_error("an index entry with another block's bytes");
          checked.fetch_add(1);
        }
      });
      std::printf("threads %u: %zu blocks written, %zu read back with their index entries\n", T, written.load(), checked.load());
      return written.load() == checked.load() && written.load() > 0 ? 0 : 1;
    }
  } catch (const std::exception &e) {
    std::printf("exception: %s\n", e.what());
    return 1;
  }
  std::printf("usage: archive_tool copy <in> <out> | dump <in> | chunks <fastq> <bytes> | rejoin <fastq> <out> <bytes>\n");
  return 2;
}
