// The misc-stream coder (fqcomp28_amd/csrc/fq_misc.cpp: fqgpu_memcompress / fqgpu_memdecompress) under
// AddressSanitizer + UBSan, CPU only: round trips of the shapes the container feeds it, then every compressed
// stream damaged -- truncated at every length, single bytes flipped, random bytes -- and handed to the decoder
// with exact-size buffers: it must refuse ((size_t)-1) or return dst_size, and never touch a byte outside.
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined tests/cpp/misc_fuzz.cpp
//       fqcomp28_amd/csrc/fq_misc.cpp -Iinclude -o misc_fuzz && ./misc_fuzz
#include "../../include/fqgpu.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

static std::mt19937_64 rng(28);

static std::vector<uint8_t> make(int kind, size_t n) {
  std::vector<uint8_t> v(n);
  switch (kind) {
    case 0: for (auto &b : v) b = (uint8_t)rng(); break;                                    // noise
    case 1: for (size_t i = 0; i < n; i++) v[i] = (i & 1) ? 0 : 150; break;                 // u16 read lengths
    case 2: for (size_t i = 0; i < n; i++) v[i] = (i & 3) ? 0 : 1; break;                   // i32 "+1" deltas
    case 3: for (auto &b : v) b = (uint8_t)("ACGT:0123"[rng() % 9]); break;                 // header text
    case 4: for (auto &b : v) b = (rng() % 50) ? 0 : (uint8_t)rng(); break;                 // mostly zero
    case 6: {                                                                                     // a random skewed alphabet: every
      const unsigned k = 2 + (unsigned)(rng() % 200), sh = 1 + (unsigned)(rng() % 6);             // size of frequency, per plane
      for (size_t i = 0; i < n; i++) {
        unsigned s = 0;
        while (s + 1 < k && (rng() & ((1u << sh) - 1)) == 0) s += 1 + (unsigned)(rng() % 3);
        v[i] = (uint8_t)((s % k) * 7 + (i & 3));
      }
      break;
    }
    default: if (n) std::memset(v.data(), 7, n);                                                   // constant
  }
  return v;
}

static unsigned long checks = 0, refused = 0;

static void decode_damaged(const std::vector<uint8_t> &z, size_t n) {
  std::vector<uint8_t> out(n);  // exact size: one byte too far is the sanitizer's
  const size_t r = fqgpu_memdecompress(out.data(), n, z.data(), z.size());
  if (r == (size_t)-1) refused++;
  else if (r != n && !(r == 0 && z.empty())) { std::printf("decoder returned %zu for dst_size %zu\n", r, n); std::exit(1); }
  checks++;
}

int main() {
  // many frequency tables through the coder's division by reciprocal and its four interleaved states: round trips only
  for (int rep = 0; rep < 300; rep++) {
    const size_t n = 1000 + (size_t)(rng() % 150000);
    const std::vector<uint8_t> src = make(6, n);
    std::vector<uint8_t> z(fqgpu_memcompress_bound(n)), back(n);
    const size_t zn = fqgpu_memcompress(z.data(), z.size(), src.data(), n);
    if (zn == 0 || zn > z.size() || fqgpu_memdecompress(back.data(), n, z.data(), zn) != n || back != src) { std::printf("round trip: skewed table %d n %zu\n", rep, n); return 1; }
  }
  for (int kind = 0; kind < 7; kind++)
    for (size_t n : {size_t(0), size_t(1), size_t(2), size_t(3), size_t(5), size_t(64), size_t(257), size_t(4096), size_t(70001)}) {
      const std::vector<uint8_t> src = make(kind, n);
      std::vector<uint8_t> z(fqgpu_memcompress_bound(n));  // exact bound (src_size + 28)
      const size_t zn = fqgpu_memcompress(z.data(), z.size(), src.data(), n);
      if (zn == (size_t)-1 || zn > z.size() || (n == 0) != (zn == 0)) { std::printf("compress: kind %d n %zu -> %zu\n", kind, n, zn); return 1; }
      z.resize(zn);
      std::vector<uint8_t> back(n);
      if (fqgpu_memdecompress(back.data(), n, z.data(), zn) != (n ? n : 0) || back != src) { std::printf("round trip: kind %d n %zu\n", kind, n); return 1; }
      // a destination too small for the stream is refused or cut, never overrun
      if (n > 1) decode_damaged(z, n - 1);
      decode_damaged(z, n + 1);
      // truncated at every length (long streams: 200 lengths)
      const size_t step = zn > 200 ? zn / 200 : 1;
      for (size_t cut = 0; cut < zn; cut += step) decode_damaged(std::vector<uint8_t>(z.begin(), z.begin() + cut), n);
      // single bytes damaged, the header and table bytes all of them
      for (size_t i = 0; i < zn; i += (i < 64 ? 1 : step)) {
        std::vector<uint8_t> d = z;
        d[i] ^= (uint8_t)(1u << (rng() % 8));
        decode_damaged(d, n);
        d[i] = (uint8_t)rng();
        decode_damaged(d, n);
      }
      // noise of the same length behind a valid first byte
      for (int t = 0; t < 20 && zn; t++) {
        std::vector<uint8_t> d(zn);
        for (auto &b : d) b = (uint8_t)rng();
        d[0] = z[0];
        decode_damaged(d, n);
      }
    }
  std::printf("misc coder: %lu damaged streams decoded or refused (%lu refused), no report\n", checks, refused);
  return 0;
}
