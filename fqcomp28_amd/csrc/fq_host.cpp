// Host-side helpers of the path's callers: the minimal FASTQ parser that turns a raw
// block into the record table the kernels consume, and the deterministic synthetic
// FASTQ generator for the BASELINE.json configurations.  No GPU code.
#include "../../include/fqgpu.h"

#include <cstdio>
#include <cstring>

// FastqReader::parseRecords semantics (reference src/fastq_io.cpp:67-125): four lines per
// record, '@' header, sequence, '+' line, quality of the same length; a trailing partial
// record is ignored (the reference carries it over to the next chunk).
extern "C" long fqgpu_parse_fastq(const uint8_t *raw, size_t len, fqgpu_rec *recs, size_t cap) {
  size_t pos = 0;
  long n = 0;
  while (pos < len) {
    size_t start[4], end[4];
    size_t p = pos;
    int ln;
    for (ln = 0; ln < 4; ln++) {
      const void *nl = p < len ? memchr(raw + p, '\n', len - p) : nullptr;
      if (!nl) break;
      start[ln] = p;
      end[ln] = (size_t)(static_cast<const uint8_t *>(nl) - raw);
      p = end[ln] + 1;
    }
    if (ln < 4) break;  // partial record at the end of the block
    if (raw[start[0]] != '@' || raw[start[2]] != '+') return -1;
    const size_t l1 = end[1] - start[1], l3 = end[3] - start[3];
    if (l1 != l3 || l1 > 65535 || start[3] > 0xFFFFFFFFull) return -1;
    if ((size_t)n < cap && recs) {
      recs[n].seq_off = (uint32_t)start[1];
      recs[n].qual_off = (uint32_t)start[3];
      recs[n].len = (uint32_t)l1;
    }
    n++;
    pos = p;
  }
  return n;
}

namespace {
struct SplitMix {
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
};

// Phred ~ round(N(34, 5)) clipped to [2, 41], integer arithmetic only: the sum of twelve
// 16-bit uniforms is the classic Irwin-Hall stand-in for a unit normal.
inline unsigned phred_normal(SplitMix &g) {
  uint64_t a = g.next(), b = g.next(), c = g.next();
  int64_t sum = 0;
  for (int i = 0; i < 4; i++) {
    sum += (int64_t)((a >> (16 * i)) & 0xFFFF) + (int64_t)((b >> (16 * i)) & 0xFFFF) +
           (int64_t)((c >> (16 * i)) & 0xFFFF);
  }
  // z = (sum - 6*65536) / 65536 ; q = floor(34 + 5 z + 0.5)
  int64_t q = (34 * 65536 + 5 * (sum - 6 * 65536) + 32768) >> 16;
  if (q < 2) q = 2;
  if (q > 41) q = 41;
  return (unsigned)q;
}
}  // namespace

extern "C" size_t fqgpu_synth_fastq(uint8_t *dst, size_t cap, int mode, uint64_t seed, uint64_t first_read_id,
                                    uint64_t *n_reads_out) {
  static const char ACGT[4] = {'A', 'C', 'G', 'T'};
  size_t pos = 0;
  uint64_t id = first_read_id, n = 0;
  for (;;) {
    // every read draws from its own generator so blocks can be produced independently
    SplitMix g{seed * 0xD1342543DE82EF95ull + id * 0x2545F4914F6CDD1Dull + 28};
    unsigned L = 150;
    if (mode == 4) L = 50 + (unsigned)(g.next() % 251);
    char hdr[96];
    const int hl = snprintf(hdr, sizeof(hdr), "@SYN.%llu %llu length=%u\n", (unsigned long long)id,
                            (unsigned long long)id, L);
    const size_t need = (size_t)hl + 2 * (size_t)L + 4;
    if (pos + need > cap) break;
    memcpy(dst + pos, hdr, (size_t)hl);
    uint8_t *s = dst + pos + hl;
    uint8_t *q = s + L + 3;
    uint64_t bits = 0;
    int have = 0;
    for (unsigned i = 0; i < L; i++) {
      if (have < 2) { bits = g.next(); have = 64; }
      s[i] = (uint8_t)ACGT[bits & 3];
      bits >>= 2; have -= 2;
    }
    s[L] = '\n'; s[L + 1] = '+'; s[L + 2] = '\n';
    if (mode == 1) {
      memset(q, 'I', L);
      for (unsigned i = 0; i < L; i++) if (g.next() % 1000 == 0) s[i] = 'N';
    } else if (mode == 3) {  // binned: four levels at 5/10/15/70 %, the level of the previous position kept with p = 0.85
      static const char LEVELS[4] = {'#', '-', '8', 'F'};
      unsigned level = 3;
      uint64_t r = 0;
      for (unsigned i = 0; i < L; i++) {
        if ((i & 1u) == 0) r = g.next();
        const unsigned keep = (unsigned)(r & 0xFFFF), pick = (unsigned)((r >> 16) & 0xFFFF);
        r >>= 32;
        if (i == 0 || keep >= 55705u)  // 0.85 * 65536
          level = pick < 3277u ? 0u : pick < 9830u ? 1u : pick < 19661u ? 2u : 3u;
        q[i] = (uint8_t)LEVELS[level];
      }
    } else if (mode == 5) {  // constant: one base, one quality (one context per stream from the fourth symbol on)
      memset(s, 'A', L);
      memset(q, 'F', L);
    } else if (mode == 6) {  // two quality levels, i.i.d. at 30/70 %, nothing else: no reset symbol, no narrow symbol, no uniform segment
      uint64_t r = 0;
      for (unsigned i = 0; i < L; i++) {
        if ((i & 3u) == 0) r = g.next();
        q[i] = (uint8_t)(((r & 0xFFFFu) < 19661u) ? '-' : 'F');  // 0.3 * 65536
        r >>= 16;
      }
    } else {
      for (unsigned i = 0; i < L; i++) q[i] = (uint8_t)(33 + phred_normal(g));
      if (mode == 4)
        for (unsigned i = 0; i < L; i++) if (g.next() % 100 == 0) { s[i] = 'N'; q[i] = '#'; }
    }
    q[L] = '\n';
    pos += need;
    id++; n++;
  }
  if (n_reads_out) *n_reads_out = n;
  return pos;
}
