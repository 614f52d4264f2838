// process.hpp -- the reference's block pipeline (src/process.cpp:32-105) over the GPU workspaces:
// N worker threads, each owning chunk + buffers + workspace (src/process.cpp:49-54, 95-98), pull
// whole blocks from a reader, code them, and hand them to a writer; blocks land in the archive in
// the order they claim their space and the index records chunk_idx (src/archive.h:85-89).  Reader,
// archive and writer (archive.hpp) move the bytes OUTSIDE their locks, and a failing worker stops
// the farm instead of leaving the others waiting (the reference's ordered writer would wait forever).
//
// What is new against the reference is only where a worker's workspace lives: worker t of T uses
// GPU devices[t mod G] (blocks are independent given the tables, every GPU holds a
// replica of them, no collective, no peer traffic).  With T > G several workers share a GPU; each
// has its own handle (own streams, own staging block in HBM), so worker A's H2D copy, worker B's
// kernels and worker C's D2H copy overlap on the device -- the chunk and stream buffers are
// page-locked (workspace.hpp: HostAllocator), the copies asynchronous (api.hip: fqgpu_encode_block).
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>

#include "archive.hpp"

namespace fqcomp28 {

struct Settings {  // the part of src/settings.h:36-50 this path needs, plus the device list
  unsigned n_threads = 1;
  std::size_t reading_chunk_size = std::size_t(256) << 20;  // -R, MiB
  std::size_t sample_chunk_size = std::size_t(128) << 20;   // -S, MiB
  std::vector<int> devices = {0};
  /** The reference never clears cbs.n_count / cbs.n_pos (src/compressed_buffers.h:58-68), so block k
   *  of a worker carries the N tables of all its earlier blocks in front (decode pops
   *  from the end, so both forms decode everywhere).  Default: fresh tables per block -- the
   *  accumulation makes misc-stream work grow quadratically with the number of blocks. */
  bool accumulate_n_buffers = false;
  /** Workers that may be inside the source (reading a chunk) at once; 0 = a quarter of the workers, at least two.
   *  Sixteen workers that all read 256 MiB at the same moment share the host's memory bandwidth, finish together, then
   *  share the PCIe link, then the page cache: every stage waits for the slowest of sixteen.  Through a gate the first
   *  chunks are on the GPU while the others are still being read and the stages overlap from the first block on
   *  (FQGPU_FARM_READ_GATE overrides; a value of n_threads or more = no gate). */
  unsigned read_gate = 0;
  /** Extension: compress also writes `<archive>.fqx`, the decode indexes of every block (archive.hpp: DecodeIndexFile;
   *  about 2 % of the archive's size); decompress uses the file whenever it lies beside the archive. */
  bool decode_index = false;
  unsigned index_stride = 0;  // symbols between two snapshots of a decode index (multiple of 64 Ki; 0 = 1 Mi)
};

struct InputStats {  // src/report.h
  std::size_t seq = 0, header = 0, n_records = 0, raw = 0;
  InputStats &operator+=(const InputStats &o) { seq += o.seq; header += o.header; n_records += o.n_records; raw += o.raw; return *this; }
};
struct CompressedStats {
  std::size_t seq = 0, qual = 0, misc = 0, n_blocks = 0;
  CompressedStats &operator+=(const CompressedStats &o) { seq += o.seq; qual += o.qual; misc += o.misc; n_blocks += o.n_blocks; return *this; }
};
struct FarmReport {
  InputStats in;
  CompressedStats out;
  double seconds = 0;          // wall clock over the worker threads (tables and handles built before)
  std::vector<unsigned> blocks_per_worker;
};

namespace detail {
inline std::size_t miscBytes(const CompressedBuffersDst &cbs) {
  std::size_t n = cbs.compressed_readlens.size() + cbs.compressed_n_count.size() + cbs.compressed_n_pos.size();
  for (const auto &f : cbs.compressed_header_fields) n += f.isDifferentFlag.size() + f.content.size() + f.contentLength.size();
  return n;
}
/** runs body(t) on n threads; a worker that throws calls on_failure() (which must make the sources
 *  of work run dry, so that the others finish their block and stop: nobody waits for anybody here);
 *  the first exception is rethrown after all have joined */
template <class Body, class OnFailure> void runWorkers(unsigned n, Body &&body, OnFailure &&on_failure) {
  std::vector<std::thread> threads;
  std::vector<std::exception_ptr> errors(n);
  threads.reserve(n);
  for (unsigned t = 0; t < n; ++t)
    threads.emplace_back([&, t] {
      try { body(t); } catch (...) { errors[t] = std::current_exception(); on_failure(); }
    });
  for (auto &th : threads) th.join();
  for (auto &e : errors) if (e) std::rethrow_exception(e);
}
template <class Body> void runWorkers(unsigned n, Body &&body) { runWorkers(n, body, [] {}); }

/** at most `slots` holders at a time (C++17: no std::counting_semaphore) */
class Gate {
public:
  explicit Gate(unsigned slots) : free_(slots) {}
  class Pass {
  public:
    explicit Pass(Gate &g) : g_(g) {
      std::unique_lock<std::mutex> lock(g_.m_);
      g_.cv_.wait(lock, [&] { return g_.free_ > 0; });
      --g_.free_;
    }
    ~Pass() {
      { const std::lock_guard<std::mutex> lock(g_.m_); ++g_.free_; }
      g_.cv_.notify_one();
    }
    Pass(const Pass &) = delete;
    Pass &operator=(const Pass &) = delete;
  private:
    Gate &g_;
  };
private:
  std::mutex m_;
  std::condition_variable cv_;
  unsigned free_;
};
}  // namespace detail

/** The compression farm: `next_chunk(chunk)` and `write_block(cbs)` are called concurrently from
 *  the workers and must be thread-safe (FastqReader::readNextChunk and Archive::writeBlock are);
 *  `stop()` is called when a worker fails and must make next_chunk return false from then on. */
template <class Source, class Sink, class Stop>
FarmReport compressFarm(const DatasetMeta &meta, Source &&next_chunk, Sink &&write_block, Stop &&stop, const Settings &set) {
  const unsigned T = std::max(1u, set.n_threads);
  if (set.devices.empty()) throw std::invalid_argument("compressFarm: no device");
  // every worker builds its workspace first (256 + 8192 tables on its GPU: the reference does the
  // same once per thread, src/workspace.h:62-64); the clock starts when all are ready
  // ... and so are its buffers: device scratch for chunks of the reading size, page-locked chunk and
  // stream buffers (half a second of hipMalloc / hipHostMalloc per worker that would otherwise sit
  // inside its first block)
  // (declared BEFORE the workspaces: locals die in reverse order, so the handles -- whose destruction waits for
  // everything they have queued -- go first and the page-locked buffers return to the pin cache after that)
  std::vector<FastqChunk> chunks(T);
  std::vector<CompressedBuffersDst> buffers(T);
  std::vector<std::unique_ptr<CompressionWorkspace>> wksp(T);
  detail::runWorkers(T, [&](unsigned t) {
    wksp[t] = std::make_unique<CompressionWorkspace>(&meta, set.devices[t % set.devices.size()]);
    wksp[t]->reserve(set.reading_chunk_size);
    wksp[t]->setDecodeIndex(set.decode_index, set.index_stride);
    chunks[t].raw_data.reserve(set.reading_chunk_size);
    buffers[t].seq.reserve(set.reading_chunk_size / 8 + (1u << 20));
    buffers[t].qual.reserve(set.reading_chunk_size / 3 + (1u << 20));
  });
  std::vector<InputStats> istats(T);
  std::vector<CompressedStats> cstats(T);
  FarmReport rep;
  rep.blocks_per_worker.assign(T, 0);
  unsigned gate_slots = set.read_gate ? set.read_gate : std::max(2u, T / 4);
  if (const char *e = std::getenv("FQGPU_FARM_READ_GATE")) gate_slots = std::max(1, std::atoi(e));
  detail::Gate read_gate(std::min(gate_slots, T));
  const auto t0 = std::chrono::steady_clock::now();
  detail::runWorkers(T, [&](unsigned t) {
    FastqChunk &chunk = chunks[t];
    CompressedBuffersDst &cbs = buffers[t];
    for (;;) {
      StageClock clk;
      {
        const detail::Gate::Pass pass(read_gate);
        if (!next_chunk(chunk)) break;
      }
      clk.lap("read");
      if (!set.accumulate_n_buffers) { cbs.n_count.clear(); cbs.n_pos.clear(); }
      wksp[t]->encodeChunk(chunk, cbs);  // (an unparsed chunk has its records found on the GPU: the sums are known afterwards)
      clk.lap("encodeChunk");
      istats[t].seq += chunk.tot_reads_length;
      istats[t].header += chunk.headers_length;
      istats[t].n_records += chunk.records.size();
      istats[t].raw += chunk.raw_data.size();
      cstats[t].seq += cbs.seq.size();
      cstats[t].qual += cbs.qual.size();
      cstats[t].misc += detail::miscBytes(cbs);
      cstats[t].n_blocks++;
      rep.blocks_per_worker[t]++;
      write_block(cbs);
      clk.lap("write");
      clk.done(chunk.idx);
    }
  }, stop);
  rep.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (unsigned t = 0; t < T; ++t) { rep.in += istats[t]; rep.out += cstats[t]; }
  return rep;
}
template <class Source, class Sink>
FarmReport compressFarm(const DatasetMeta &meta, Source &&next_chunk, Sink &&write_block, const Settings &set) {
  std::atomic<bool> stopped{false};
  return compressFarm(meta, [&](FastqChunk &c) { return !stopped.load() && next_chunk(c); }, write_block, [&] { stopped.store(true); }, set);
}

/** processReads (src/process.cpp:32-82): file in, archive out */
inline FarmReport processReads(const path_t &mates1, const path_t &archive_path, const Settings &set) {
  Archive archive(archive_path, mates1, set.sample_chunk_size, set.devices.at(0));
  FastqReader reader(mates1, set.reading_chunk_size);
  std::unique_ptr<DecodeIndexFile> sidecar;
  if (set.decode_index) {
    sidecar = std::make_unique<DecodeIndexFile>(DecodeIndexFile::pathFor(archive_path), PosFile::Mode::Create);
  } else {  // what an earlier archive of this name left behind is not this one's
    std::error_code ec;
    std::filesystem::remove(DecodeIndexFile::pathFor(archive_path), ec);
  }
  FarmReport rep = compressFarm(
      archive.meta(), [&](FastqChunk &c) { return reader.readNextChunk(c); },
      [&](const CompressedBuffersDst &cbs) {
        archive.writeBlock(cbs);
        if (sidecar) sidecar->put(cbs);
      },
      [&] { reader.abort(); }, set);
  archive.writeIndex();
  archive.flush();
  if (sidecar) sidecar->close(DecodeIndexFile::identityOf(archive_path));
  return rep;
}

/** The decompression farm (src/process.cpp:84-105): `next_block(cbs)` / `write_chunk(chunk)` thread-safe;
 *  `stop()`: a worker has failed, next_block must return false from now on */
template <class Source, class Sink, class Stop>
FarmReport decompressFarm(const DatasetMeta &meta, Source &&next_block, Sink &&write_chunk, Stop &&stop, const Settings &set) {
  const unsigned T = std::max(1u, set.n_threads);
  if (set.devices.empty()) throw std::invalid_argument("decompressFarm: no device");
  std::vector<std::unique_ptr<DecompressionWorkspace>> wksp(T);
  detail::runWorkers(T, [&](unsigned t) { wksp[t] = std::make_unique<DecompressionWorkspace>(&meta, set.devices[t % set.devices.size()]); });
  std::vector<InputStats> istats(T);
  FarmReport rep;
  rep.blocks_per_worker.assign(T, 0);
  const auto t0 = std::chrono::steady_clock::now();
  detail::runWorkers(T, [&](unsigned t) {
    CompressedBuffersSrc cbs;
    FastqChunk chunk;
    for (;;) {
      StageClock clk;
      if (!next_block(cbs)) break;
      clk.lap("read");
      wksp[t]->decodeChunk(chunk, cbs);
      clk.lap("decodeChunk");
      istats[t].raw += chunk.raw_data.size();
      istats[t].n_records += chunk.records.size();
      rep.blocks_per_worker[t]++;
      write_chunk(chunk);
      clk.lap("write");
      clk.done(chunk.idx);
    }
  }, stop);
  rep.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (unsigned t = 0; t < T; ++t) rep.in += istats[t];
  return rep;
}
template <class Source, class Sink>
FarmReport decompressFarm(const DatasetMeta &meta, Source &&next_block, Sink &&write_chunk, const Settings &set) {
  std::atomic<bool> stopped{false};
  return decompressFarm(meta, [&](CompressedBuffersSrc &c) { return !stopped.load() && next_block(c); }, write_chunk, [&] { stopped.store(true); }, set);
}

/** processArchiveParts (src/process.cpp:84-105): archive in, file out (chunks in original order) */
inline FarmReport processArchiveParts(const path_t &archive_path, const path_t &mates1_out, const Settings &set) {
  Archive archive(archive_path);
  FastqWriter writer(mates1_out, archive.chunkOffsets());
  std::unique_ptr<DecodeIndexFile> sidecar;
  if (std::filesystem::exists(DecodeIndexFile::pathFor(archive_path))) {
    sidecar = std::make_unique<DecodeIndexFile>(DecodeIndexFile::pathFor(archive_path), PosFile::Mode::Read);
    if (!sidecar->belongsTo(DecodeIndexFile::identityOf(archive_path))) {
      std::fprintf(stderr, "%s was written for another archive: not used\n", DecodeIndexFile::pathFor(archive_path).string().c_str());
      sidecar.reset();
    }
  }
  FarmReport rep = decompressFarm(
      archive.meta(),
      [&](CompressedBuffersSrc &cbs) {
        if (!archive.readBlock(cbs)) return false;
        if (sidecar) (void)sidecar->get(cbs);  // (a chunk the file has no entry for is decoded without)
        return true;
      },
      [&](const FastqChunk &chunk) { writer.writeChunk(chunk); }, [&] { archive.abort(); }, set);
  writer.flush();
  return rep;
}

}  // namespace fqcomp28
