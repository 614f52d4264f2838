// Misc-stream compressor behind fqgpu_memcompress / fqgpu_memdecompress -- the shim's
// memcompress()/memdecompress() (reference src/memcompress.h:5-28, src/memcompress.cpp:39-67),
// which CompressionWorkspace::compressMiscBuffers / DecompressionWorkspace::decompressMiscBuffers
// (src/workspace.cpp:176-256) run over readlens, n_count, n_pos and every header field stream.
//
// The reference hands these streams to libbsc (BWT + QLFC + LZP, cmake/Libbsc.cmake:3-7); its
// source is not part of the reference tree, so the BYTES of the compressed misc streams are out of
// parity scope.  What is kept is the interface and its contract: the caller
// sizes dst as src_size + 28 (extra_csize_misc = LIBBSC_HEADER_SIZE, src/workspace.h:18), an empty
// input gives an empty output, and decompression needs the original size from the container.
//
// Own format "FQM": the streams are little-endian u16 (readlens, n_count, n_pos), i32 deltas
// (numeric header fields) or bytes (flags, lengths, string content), so the coder splits the
// input into `stride` byte planes (1, 2 or 4: whichever the order-0 entropy estimate likes best)
// and codes every plane with a static order-0 rANS (12-bit frequencies).  A constant plane -- the
// high byte of 150 bp read lengths, the three upper bytes of a "+1" read-number delta -- costs its
// table and the state words.
//   byte 0      mode: 0 = stored (src follows); 4 + log2(stride) = planes, each coded with FOUR
//               interleaved rANS states (symbol i of the plane belongs to state i mod 4) and 16-bit
//               renormalisation; 1 + log2(stride) = round 3's one state with byte renormalisation
//               (still decoded, no longer written)
//   per plane   u8 k (symbols in the table, 0 = 256) | k x {u8 symbol, u16 freq} | u32 n | n bytes
//               mode 4..6: the n bytes = four u32 states, then the 16-bit words in decoding order; a plane of
//               ONE symbol has n = 0 (nothing to code)
// Four states because the coder is a serial chain otherwise -- one multiply-shift-add per symbol
// that the next symbol waits for, and a renormalisation branch nobody can predict: 47 ms of a
// farm worker's 190 ms per 256 MiB block at 120 MB/s.  Four independent
// chains overlap in the core.
// Host code: the inputs are the small side streams of a block.
#include "../../include/fqgpu.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace {

constexpr unsigned SCALE_BITS = 12, SCALE = 1u << SCALE_BITS;
constexpr uint32_t RANS_L = 1u << 23;   // modes 1..3 (decode only)
constexpr uint32_t RANS_L4 = 1u << 15;  // modes 4..6: states in [2^15, 2^31), one 16-bit word per renormalisation
constexpr unsigned WAYS = 4;

struct Plane {
  uint32_t count[256];
  size_t n;
};

// counts -> frequencies that sum to 4096, every used symbol >= 1
void normalise(const uint32_t *count, size_t n, uint16_t *freq) {
  unsigned used = 0, best = 0;
  uint32_t sum = 0;
  for (unsigned s = 0; s < 256; s++) {
    freq[s] = 0;
    if (!count[s]) continue;
    used++;
    uint64_t f = (uint64_t)count[s] * SCALE / n;
    if (f == 0) f = 1;
    freq[s] = (uint16_t)f;
    sum += (uint32_t)f;
    if (count[s] > count[best] || !count[best]) best = s;
  }
  (void)used;
  // the rounding error goes to the most frequent symbol; if that would empty it, take it
  // from the others one by one (only possible with many rare symbols)
  if (sum <= SCALE) {
    freq[best] = (uint16_t)(freq[best] + (SCALE - sum));
  } else {
    uint32_t over = sum - SCALE;
    while (over) {
      unsigned big = 0;
      for (unsigned s = 1; s < 256; s++)
        if (freq[s] > freq[big]) big = s;
      const uint32_t take = freq[big] - 1u < over ? freq[big] - 1u : over;
      freq[big] = (uint16_t)(freq[big] - take);
      over -= take;
    }
  }
}

double plane_cost_bits(const Plane &p) {
  if (!p.n) return 0.0;
  double bits = 0.0;
  unsigned used = 0;
  for (unsigned s = 0; s < 256; s++)
    if (p.count[s]) {
      used++;
      bits += (double)p.count[s] * std::log2((double)p.n / (double)p.count[s]);
    }
  return bits + 8.0 * (1 + 3 * used + 4 + (used > 1 ? 4 * WAYS : 0));
}

// what the encoder needs per symbol: x -> ((x / f) << 12) + (x % f) + cum = x + bias + (x / f) * (4096 - f), the
// quotient by a multiplication with the rounded-up reciprocal (exact for x < 2^31: Alverson, "Integer division using
// reciprocals"; tests/cpp/misc_fuzz.cpp checks every frequency at the edges of every quotient step)
struct EncSym {
  uint32_t x_max, rcp, bias;
  uint16_t cmpl, shift;
};

inline void enc_sym_init(EncSym &e, uint32_t start, uint32_t f) {
  e.x_max = ((RANS_L4 >> SCALE_BITS) << 16) * f;
  e.cmpl = (uint16_t)(SCALE - f);
  if (f < 2) {  // x / 1 = x: q = (x * (2^32 - 1)) >> 32 = x - 1 for x > 0, made up for in the bias
    e.rcp = ~0u;
    e.shift = 0;
    e.bias = start + SCALE - 1;
  } else {
    unsigned sh = 0;
    while (f > (1u << sh)) sh++;
    e.rcp = (uint32_t)(((1ull << (sh + 31)) + f - 1) / f);
    e.shift = (uint16_t)(sh - 1);
    e.bias = start;
  }
}

inline uint32_t enc_put(uint32_t x, const EncSym &e, uint16_t *&ptr) {
  if (x >= e.x_max) { *--ptr = (uint16_t)x; x >>= 16; }
  const uint32_t q = (uint32_t)(((uint64_t)x * e.rcp) >> 32) >> e.shift;
  return x + e.bias + q * e.cmpl;
}

// one plane: symbols src[first], src[first + stride], ...  -> appended to out
void encode_plane(const uint8_t *src, size_t first, size_t stride, const Plane &p, std::vector<uint8_t> &out, std::vector<uint16_t> &words) {
  uint16_t freq[256];
  normalise(p.count, p.n, freq);
  EncSym sym[256];
  uint32_t cum = 0;
  unsigned used = 0;
  for (unsigned s = 0; s < 256; s++) {
    if (freq[s]) { enc_sym_init(sym[s], cum, freq[s]); used++; }
    cum += freq[s];
  }
  out.push_back((uint8_t)(used & 0xFFu));  // 256 -> 0
  for (unsigned s = 0; s < 256; s++)
    if (freq[s]) {
      out.push_back((uint8_t)s);
      out.push_back((uint8_t)(freq[s] & 0xFFu));
      out.push_back((uint8_t)(freq[s] >> 8));
    }
  if (used == 1) {  // a constant plane: its table says it all
    for (int b = 0; b < 4; b++) out.push_back(0);
    return;
  }
  // rANS runs backwards over the plane and writes its words backwards; state i mod 4 codes symbol i
  words.resize(p.n + 8);
  uint16_t *const end = words.data() + words.size(), *ptr = end;
  uint32_t x[WAYS] = {RANS_L4, RANS_L4, RANS_L4, RANS_L4};
  size_t k = p.n;
  const uint8_t *const plane = src + first;
  while (k & (WAYS - 1)) { --k; x[k & (WAYS - 1)] = enc_put(x[k & (WAYS - 1)], sym[plane[k * stride]], ptr); }
  for (; k; k -= WAYS) {  // k is a multiple of four: symbols k-1 .. k-4 with the states 3 .. 0
    const uint8_t *b = plane + (k - WAYS) * stride;
    x[3] = enc_put(x[3], sym[b[3 * stride]], ptr);
    x[2] = enc_put(x[2], sym[b[2 * stride]], ptr);
    x[1] = enc_put(x[1], sym[b[stride]], ptr);
    x[0] = enc_put(x[0], sym[b[0]], ptr);
  }
  const uint32_t nb = (uint32_t)(4 * WAYS + 2 * (size_t)(end - ptr));
  for (int b = 0; b < 4; b++) out.push_back((uint8_t)((nb >> (8 * b)) & 0xFFu));
  for (unsigned j = 0; j < WAYS; j++)
    for (int b = 0; b < 4; b++) out.push_back((uint8_t)((x[j] >> (8 * b)) & 0xFFu));
  const size_t o = out.size();
  out.resize(o + 2 * (size_t)(end - ptr));
  uint8_t *dst = out.data() + o;
  for (const uint16_t *w = ptr; w < end; w++) { *dst++ = (uint8_t)(*w & 0xFFu); *dst++ = (uint8_t)(*w >> 8); }
}

}  // namespace

extern "C" size_t fqgpu_memcompress_bound(size_t src_size) { return src_size + 28; }

extern "C" size_t fqgpu_memcompress(uint8_t *dst, size_t dst_cap, const uint8_t *src, size_t src_size) {
  if (src_size == 0) return 0;
  if (!dst || !src || dst_cap < src_size + 1) return 0;
  if (src_size < 16) {  // not worth a table; also keeps every plane of every stride non-empty below
    dst[0] = 0;
    memcpy(dst + 1, src, src_size);
    return src_size + 1;
  }
  // plane histograms for stride 4; strides 2 and 1 are sums of them.  Eight bytes at a time, and a run of equal
  // 8-byte words is counted in one go: most of a block's side streams are (nearly) constant -- read lengths, N counts,
  // "+1" read numbers, "same as before" flags -- and a counter that is incremented again before its last store has
  // retired costs a store-forwarding round trip per byte (10 ms per 256 MiB block of the farm went into this loop)
  Plane p4[4], p2[2], p1;
  memset(p4, 0, sizeof(p4));
  {
    size_t i = 0;
    while (i + 8 <= src_size) {
      uint64_t w;
      memcpy(&w, src + i, 8);
      size_t run = 1;
      while (i + 8 * (run + 1) <= src_size) {
        uint64_t v;
        memcpy(&v, src + i + 8 * run, 8);
        if (v != w) break;
        run++;
      }
      for (int k = 0; k < 8; k++) p4[k & 3].count[src[i + k]] += (uint32_t)run;  // (i is a multiple of 8: byte k lies in plane k & 3)
      i += 8 * run;
    }
    for (; i < src_size; i++) p4[i & 3].count[src[i]]++;
  }
  for (int k = 0; k < 4; k++) p4[k].n = (src_size + 3 - (size_t)k) / 4;
  memset(p2, 0, sizeof(p2));
  memset(&p1, 0, sizeof(p1));
  for (int k = 0; k < 4; k++)
    for (unsigned s = 0; s < 256; s++) {
      p2[k & 1].count[s] += p4[k].count[s];
      p1.count[s] += p4[k].count[s];
    }
  p2[0].n = (src_size + 1) / 2; p2[1].n = src_size / 2;
  p1.n = src_size;
  const double c1 = plane_cost_bits(p1), c2 = plane_cost_bits(p2[0]) + plane_cost_bits(p2[1]),
               c4 = plane_cost_bits(p4[0]) + plane_cost_bits(p4[1]) + plane_cost_bits(p4[2]) + plane_cost_bits(p4[3]);
  unsigned lg = 0;
  double best = c1;
  if (c2 < best) { best = c2; lg = 1; }
  if (c4 < best) { best = c4; lg = 2; }
  if (best / 8.0 + 1.0 < (double)src_size) {
    std::vector<uint8_t> out;
    std::vector<uint16_t> words;
    out.reserve(src_size / 2 + 64);
    out.push_back((uint8_t)(4 + lg));
    const size_t stride = (size_t)1 << lg;
    const Plane *planes = lg == 0 ? &p1 : lg == 1 ? p2 : p4;
    for (size_t k = 0; k < stride; k++) encode_plane(src, k, stride, planes[k], out, words);
    if (out.size() < src_size + 1 && out.size() <= dst_cap) {
      memcpy(dst, out.data(), out.size());
      return out.size();
    }
  }
  dst[0] = 0;  // stored
  memcpy(dst + 1, src, src_size);
  return src_size + 1;
}

// returns dst_size on success, 0 on an empty input, (size_t)-1 on a malformed stream
extern "C" size_t fqgpu_memdecompress(uint8_t *dst, size_t dst_size, const uint8_t *src, size_t src_size) {
  if (src_size == 0) return 0;  // src/memcompress.cpp:56-57
  if (!src || (!dst && dst_size)) return (size_t)-1;
  const unsigned mode = src[0];
  if (mode == 0) {
    if (src_size != dst_size + 1) return (size_t)-1;
    memcpy(dst, src + 1, dst_size);
    return dst_size;
  }
  if (mode > 6) return (size_t)-1;
  const bool four = mode >= 4;
  const size_t stride = (size_t)1 << (four ? mode - 4 : mode - 1);
  size_t at = 1;
  std::vector<uint8_t> slot_sym(SCALE);
  std::vector<uint32_t> slot_tab(four ? SCALE : 0);  // freq | (slot - cum) << 16
  for (size_t k = 0; k < stride; k++) {
    const size_t n = (dst_size + stride - 1 - k) / stride;
    if (at + 1 > src_size) return (size_t)-1;
    unsigned used = src[at++];
    if (used == 0) used = 256;
    uint16_t freq[256];
    uint32_t cum[257];
    memset(freq, 0, sizeof(freq));
    if (at + 3 * (size_t)used > src_size) return (size_t)-1;
    for (unsigned u = 0; u < used; u++, at += 3) freq[src[at]] = (uint16_t)(src[at + 1] | (src[at + 2] << 8));
    cum[0] = 0;
    for (unsigned s = 0; s < 256; s++) cum[s + 1] = cum[s] + freq[s];
    if (cum[256] != SCALE) return (size_t)-1;
    for (unsigned s = 0; s < 256; s++)
      for (uint32_t j = cum[s]; j < cum[s + 1]; j++) slot_sym[j] = (uint8_t)s;
    if (at + 4 > src_size) return (size_t)-1;
    const uint32_t nb = (uint32_t)src[at] | ((uint32_t)src[at + 1] << 8) | ((uint32_t)src[at + 2] << 16) | ((uint32_t)src[at + 3] << 24);
    at += 4;
    if (four && nb == 0) {  // a constant plane
      if (used != 1) return (size_t)-1;
      size_t i = k;
      for (size_t j = 0; j < n; j++, i += stride) dst[i] = slot_sym[0];
      continue;
    }
    if (nb < 4 || at + nb > src_size) return (size_t)-1;
    const uint8_t *p = src + at, *const pe = p + nb;
    at += nb;
    if (four) {
      if (nb < 4 * WAYS || ((nb - 4 * WAYS) & 1u)) return (size_t)-1;
      for (unsigned s = 0; s < 256; s++)
        for (uint32_t j = cum[s]; j < cum[s + 1]; j++) slot_tab[j] = (uint32_t)freq[s] | ((j - cum[s]) << 16);
      uint32_t x[WAYS];
      for (unsigned j = 0; j < WAYS; j++, p += 4) {
        x[j] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
        if (x[j] < RANS_L4 || x[j] >= (RANS_L4 << 16)) return (size_t)-1;
      }
      size_t i = k;
      for (size_t j = 0; j < n; j++, i += stride) {
        uint32_t &xs = x[j & (WAYS - 1)];
        const uint32_t slot = xs & (SCALE - 1), t = slot_tab[slot];
        dst[i] = slot_sym[slot];
        xs = (t & 0xFFFFu) * (xs >> SCALE_BITS) + (t >> 16);
        if (xs < RANS_L4) {
          if (p == pe) return (size_t)-1;
          xs = (xs << 16) | (uint32_t)p[0] | ((uint32_t)p[1] << 8);
          p += 2;
        }
      }
      // every word consumed, every state back at the encoder's start
      if (p != pe || x[0] != RANS_L4 || x[1] != RANS_L4 || x[2] != RANS_L4 || x[3] != RANS_L4) return (size_t)-1;
      continue;
    }
    uint32_t x = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
    p += 4;
    size_t i = k;
    for (size_t j = 0; j < n; j++, i += stride) {
      const uint32_t slot = x & (SCALE - 1);
      const unsigned s = slot_sym[slot];
      dst[i] = (uint8_t)s;
      x = freq[s] * (x >> SCALE_BITS) + slot - cum[s];
      while (x < RANS_L) {
        if (p == pe) return (size_t)-1;
        x = (x << 8) | *p++;
      }
    }
    if (p != pe || x != RANS_L) return (size_t)-1;  // every byte consumed, the encoder's start state reached
  }
  return at == src_size ? dst_size : (size_t)-1;
}
