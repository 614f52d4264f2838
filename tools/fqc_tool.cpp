// fqc_tool -- the reference's two commands over the GPU block farm (fqcomp28_amd/csrc/process.hpp):
//   fqc_tool c <in.fastq> <out.fqc> [-t threads] [-R block MiB] [-S sample MiB] [-d dev,dev,...] [--accumulate-n]
//              [--index [--index-stride Ki symbols, a multiple of 64]]   (extension: decode indexes in <out.fqc>.fqx)
//   fqc_tool d <in.fqc> <out.fastq> [-t threads] [-d dev,dev,...]
// (fqcomp28 c --i1 in.fastq -o out.fqc -t N / fqcomp28 d -i out.fqc --o1 out.fastq, src/app.cpp:29-76.)
// Prints one JSON line with sizes, seconds and blocks per worker.  Needs a GPU: no CPU fallback.
#include "../fqcomp28_amd/csrc/process.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace fqcomp28;

int main(int argc, char **argv) {
  if (argc < 4 || (strcmp(argv[1], "c") && strcmp(argv[1], "d"))) {
    std::fprintf(stderr, "usage: fqc_tool c|d <in> <out> [-t N] [-R MiB] [-S MiB] [-d 0,1,..] [--accumulate-n] [--index] [--index-stride KiSymbols]\n");
    return 2;
  }
  Settings set;
  for (int i = 4; i < argc; ++i) {
    const std::string a = argv[i];
    auto val = [&]() -> const char * { if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", a.c_str()); std::exit(2); } return argv[++i]; };
    if (a == "-t") set.n_threads = (unsigned)std::atoi(val());
    else if (a == "-R") set.reading_chunk_size = (std::size_t)std::atoll(val()) << 20;
    else if (a == "-S") set.sample_chunk_size = (std::size_t)std::atoll(val()) << 20;
    else if (a == "--accumulate-n") set.accumulate_n_buffers = true;
    else if (a == "--index") set.decode_index = true;
    else if (a == "--index-stride") { set.decode_index = true; set.index_stride = static_cast<unsigned>(std::atoi(val())) << 10; }  // Ki symbols
    else if (a == "-d") {
      set.devices.clear();
      for (const char *p = val(); *p;) { set.devices.push_back(std::atoi(p)); while (*p && *p != ',') ++p; if (*p) ++p; }
    } else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
  }
  try {
    const bool comp = argv[1][0] == 'c';
    const FarmReport r = comp ? processReads(argv[2], argv[3], set) : processArchiveParts(argv[2], argv[3], set);
    std::printf("{\"cmd\": \"%s\", \"threads\": %u, \"devices\": %zu, \"raw_bytes\": %zu, \"records\": %zu, \"blocks\": %zu, "
                "\"seq_bytes\": %zu, \"qual_bytes\": %zu, \"misc_bytes\": %zu, \"seconds\": %.6f, \"blocks_per_worker\": [",
                argv[1], set.n_threads, set.devices.size(), r.in.raw, r.in.n_records, comp ? r.out.n_blocks : (std::size_t)0,
                r.out.seq, r.out.qual, r.out.misc, r.seconds);
    for (std::size_t i = 0; i < r.blocks_per_worker.size(); ++i) std::printf("%s%u", i ? ", " : "", r.blocks_per_worker[i]);
    std::printf("]}\n");
  } catch (const std::exception &e) {
    std::fprintf(stderr, "fqc_tool: %s\n", e.what());
    return 1;
  }
  return 0;
}
