// Header field coder on the device -- SURVEY.md 8(f) row 3, first half, for the block that
// fqgpu_encode_begin has in flight: what CompressionWorkspace::encodeHeader does record by record
// (reference src/workspace.cpp:95-126 over FieldStorageDst::storeString / storeNumeric,
// src/headers.cpp:76-91, 110-120), for all records at once.
//
// Why it is parallel although "every field is coded against the previous header": the previous
// header is INPUT.  Record r tokenises its own header and header r-1 (record 0: the dataset's first
// header, Workspace::startNewChunk, src/workspace.cpp:90-93) side by side:
//   STRING  field: flag = (text differs); a differing field adds its length byte and its bytes,
//                  whose places are prefix sums over the records (per field)
//   NUMERIC field: int32 difference of the two values, four little-endian bytes at 4 r
// (FieldStorageDst keeps `prev_val` unchanged when the text is the same, so "the previous header's
// field" and "the last stored value" are the same thing.)
//   k_hdr_count   per workgroup of 256 records and string field: differing fields, their bytes
//   k_hdr_layout  prefix over the workgroups per field, then the places of every field's three
//                 streams in ONE output buffer: per field, in order, flags | content | lengths
//                 (FieldStorage::isDifferentFlag / content / contentLength) and their sizes
//   k_hdr_write   tokenises again, scans inside the workgroup, writes
// A header the host coder throws on (a numeric field without digits or outside int32, a differing
// string field of 255 or more bytes) is reported with the index of its record; the reference has
// an assert there (src/headers.cpp:20, 88, 117).
#include "fqgpu_internal.h"

#include <cstring>

namespace {

constexpr unsigned HDR_THREADS = 256, HDR_WAVES = HDR_THREADS / 64;
constexpr unsigned HDR_CNT_BITS = 9;  // a workgroup's count of differing fields (<= 256) under its byte count

struct HdrFormat {
  uint8_t type[FQGPU_HDR_MAX_FIELDS];  // 0 = NUMERIC, 1 = STRING (headers::FieldType, src/headers.h:12)
  uint8_t sep[FQGPU_HDR_MAX_FIELDS];   // separator behind field i (none behind the last)
  uint32_t n_fields;
};

struct HdrText {
  const uint8_t *s, *end;  // the rest of a header: the next field begins at s
};

// the next field [fs, fe): it ends at the first `sep` found from its SECOND byte on (a field has at least one
// character: std::find(field_start + 1, ...), src/workspace.cpp:99), the last field is the rest
__device__ __forceinline__ void hdr_field(HdrText &t, bool last, uint8_t sep, const uint8_t *&fs, const uint8_t *&fe) {
  fs = t.s;
  if (last) {
    fe = t.end;
    t.s = t.end;
    return;
  }
  const uint8_t *p = t.s < t.end ? t.s + 1 : t.end;
  while (p < t.end && *p != sep) ++p;
  fe = p;
  t.s = p < t.end ? p + 1 : t.end;
}

// std::from_chars(s, e, int32) with "no error" as the result: an optional '-', then at least one digit, read up to
// the first other character, inside int32
__device__ __forceinline__ bool hdr_int32(const uint8_t *s, const uint8_t *e, uint32_t &v) {
  bool neg = false;
  if (s < e && *s == '-') { neg = true; ++s; }
  unsigned long long acc = 0;
  unsigned digits = 0;
  for (; s < e; ++s) {
    const unsigned d = (unsigned)*s - (unsigned)'0';
    if (d > 9u) break;
    acc = acc * 10ull + d;
    if (acc > 0x80000000ull) acc = 0x80000001ull;  // out of range for either sign, and stays so
    digits++;
  }
  v = neg ? 0u - (uint32_t)acc : (uint32_t)acc;
  return digits && acc <= (neg ? 0x80000000ull : 0x7FFFFFFFull);
}

__device__ __forceinline__ void hdr_error(HdrResult *res, unsigned record, unsigned code) {
  atomicMin(&res->first_error, ((unsigned long long)record << 8) | code);
}

// Both headers of record r; threads behind the last record get two empty texts
__device__ __forceinline__ void hdr_texts(const uint8_t *raw, const fqgpu_rec *recs, unsigned r, unsigned R, const uint8_t *first_hdr,
                                          unsigned first_len, HdrText &cur, HdrText &prev) {
  cur.s = cur.end = prev.s = prev.end = raw;
  if (r >= R) return;
  // (offsets are clamped so that a caller's table that does not describe the chunk -- records out of order, a sequence
  // at offset 0 -- yields nonsense fields, never an address outside the block: seq_off and qual_off + len are inside it)
  const fqgpu_rec me = recs[r];
  unsigned line = 0;
  if (r) {
    const fqgpu_rec before = recs[r - 1];
    const unsigned line0 = r > 1 ? recs[r - 2].qual_off + recs[r - 2].len + 1u : 0u;
    line = before.qual_off + before.len + 1u;
    const unsigned end0 = before.seq_off ? before.seq_off - 1u : 0u;
    prev.end = raw + end0;
    prev.s = raw + min(line0 + 1u, end0);
  } else {
    prev.end = first_hdr + first_len;
    prev.s = first_hdr + min(1u, first_len);
  }
  const unsigned end1 = me.seq_off ? me.seq_off - 1u : 0u;
  cur.end = raw + end1;
  cur.s = raw + min(line + 1u, end1);  // behind the '@'
}

// Field i of record r against the same field of the header in front.  STRING: returns count | bytes << 9 of what the
// field adds to lengths / content (0: same text, or an error), `differs` for the flag.  NUMERIC: `delta`.
__device__ __forceinline__ unsigned hdr_code_field(HdrText &cur, HdrText &prev, const HdrFormat &fmt, unsigned i, unsigned r, bool active,
                                                   HdrResult *res, const uint8_t *&fs, unsigned &len, bool &differs, uint32_t &delta) {
  const bool last = i + 1 == fmt.n_fields;
  const uint8_t *fe, *ps, *pe;
  hdr_field(cur, last, fmt.sep[i], fs, fe);
  hdr_field(prev, last, fmt.sep[i], ps, pe);
  len = (unsigned)(fe - fs);
  differs = false;
  delta = 0;
  if (!active) return 0;
  if (fmt.type[i]) {
    differs = len != (unsigned)(pe - ps);
    for (unsigned k = 0; !differs && k < len; k++) differs = fs[k] != ps[k];
    if (!differs) return 0;
    if (len >= 255u) {  // FIELDLEN_MAX, src/headers.h:25 / src/headers.cpp:83
      hdr_error(res, r, 2);
      differs = false;
      return 0;
    }
    return 1u | (len << HDR_CNT_BITS);
  }
  uint32_t v = 0, pv = 0;
  if (!hdr_int32(fs, fe, v)) hdr_error(res, r, 1);
  if (!hdr_int32(ps, pe, pv) && r == 0) hdr_error(res, 0, 1);  // (r > 0: record r - 1 reports its own field)
  delta = v - pv;
  return 0;
}

__global__ void __launch_bounds__(HDR_THREADS)
k_hdr_count(const uint8_t *__restrict__ raw, const fqgpu_rec *__restrict__ recs, unsigned R, const uint8_t *__restrict__ first_hdr,
            unsigned first_len, const HdrFormat fmt, uint32_t *__restrict__ wg_sum, HdrResult *__restrict__ res) {
  __shared__ unsigned wsum[FQGPU_HDR_MAX_FIELDS][HDR_WAVES];
  const unsigned r = blockIdx.x * HDR_THREADS + threadIdx.x;
  HdrText cur, prev;
  hdr_texts(raw, recs, r, R, first_hdr, first_len, cur, prev);
  for (unsigned i = 0; i < fmt.n_fields; i++) {
    const uint8_t *fs;
    unsigned len;
    bool differs;
    uint32_t delta;
    unsigned v = hdr_code_field(cur, prev, fmt, i, r, r < R, res, fs, len, differs, delta);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    if (fq_lane() == 0) wsum[i][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < fmt.n_fields) {
    unsigned v = 0;
    for (unsigned w = 0; w < HDR_WAVES; w++) v += wsum[threadIdx.x][w];
    wg_sum[(size_t)blockIdx.x * fmt.n_fields + threadIdx.x] = v;
  }
}

// one workgroup: wave w takes the fields w, w + 4, ...; wg_base[wg][field] = {bytes, count} in front of the workgroup
__global__ void __launch_bounds__(HDR_THREADS)
k_hdr_layout(const uint32_t *__restrict__ wg_sum, unsigned n_wg, unsigned R, const HdrFormat fmt, uint2 *__restrict__ wg_base,
             HdrResult *__restrict__ res, unsigned long long cap) {
  __shared__ unsigned long long tot_bytes[FQGPU_HDR_MAX_FIELDS];  // (64 bits: a table that is not the chunk's can add up to anything)
  __shared__ unsigned tot_cnt[FQGPU_HDR_MAX_FIELDS];
  const unsigned lane = fq_lane(), F = fmt.n_fields;
  for (unsigned i = threadIdx.x >> 6; i < F; i += HDR_WAVES) {
    unsigned long long carry_b = 0;
    unsigned carry_c = 0;
    if (fmt.type[i])
      for (unsigned base = 0; base < n_wg; base += 64) {
        const unsigned w = base + lane;
        const unsigned v = w < n_wg ? wg_sum[(size_t)w * F + i] : 0u;
        const unsigned c = v & ((1u << HDR_CNT_BITS) - 1u), b = v >> HDR_CNT_BITS;
        unsigned ic = c, ib = b;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const unsigned oc = __shfl_up(ic, d), ob = __shfl_up(ib, d);
          if (lane >= (unsigned)d) { ic += oc; ib += ob; }
        }
        if (w < n_wg) wg_base[(size_t)w * F + i] = make_uint2((unsigned)carry_b + ib - b, carry_c + ic - c);
        carry_b += __shfl(ib, 63);
        carry_c += __shfl(ic, 63);
      }
    if (lane == 0) { tot_bytes[i] = carry_b; tot_cnt[i] = carry_c; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long off = 0;
    for (unsigned i = 0; i < F; i++) {
      const unsigned long long s_flags = fmt.type[i] ? R : 0u, s_content = fmt.type[i] ? tot_bytes[i] : 4ull * R, s_len = fmt.type[i] ? tot_cnt[i] : 0u;
      res->size[3 * i] = (uint32_t)s_flags; res->size[3 * i + 1] = (uint32_t)s_content; res->size[3 * i + 2] = (uint32_t)s_len;
      res->off[3 * i] = off; off += s_flags;
      res->off[3 * i + 1] = off; off += s_content;
      res->off[3 * i + 2] = off; off += s_len;
    }
    res->total = off;
    // more than every header byte once: only a record table whose "headers" overlap can ask for that; nothing is written
    if (off > cap) hdr_error(res, 0, 3);
  }
}

__global__ void __launch_bounds__(HDR_THREADS)
k_hdr_write(const uint8_t *__restrict__ raw, const fqgpu_rec *__restrict__ recs, unsigned R, const uint8_t *__restrict__ first_hdr,
            unsigned first_len, const HdrFormat fmt, const uint2 *__restrict__ wg_base, HdrResult *__restrict__ res,
            uint8_t *__restrict__ out, unsigned long long cap) {
  __shared__ unsigned wtot[FQGPU_HDR_MAX_FIELDS][HDR_WAVES];
  if (res->total > cap) return;  // (uniform: written by k_hdr_layout)
  const unsigned r = blockIdx.x * HDR_THREADS + threadIdx.x, lane = fq_lane(), wave = threadIdx.x >> 6;
  const bool active = r < R;
  HdrText cur, prev;
  hdr_texts(raw, recs, r, R, first_hdr, first_len, cur, prev);
  for (unsigned i = 0; i < fmt.n_fields; i++) {
    const uint8_t *fs;
    unsigned len;
    bool differs;
    uint32_t delta;
    const unsigned v = hdr_code_field(cur, prev, fmt, i, r, active, res, fs, len, differs, delta);
    if (!fmt.type[i]) {  // (uniform: the format is a kernel argument)
      if (active) {
        uint8_t *dst = out + res->off[3 * i + 1] + 4ull * r;
#pragma unroll
        for (int k = 0; k < 4; k++) dst[k] = (uint8_t)(delta >> (8 * k));
      }
      continue;
    }
    unsigned inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned o = __shfl_up(inc, d);
      if (lane >= (unsigned)d) inc += o;
    }
    if (lane == 63) wtot[i][wave] = inc;
    __syncthreads();
    unsigned excl = inc - v;
    for (unsigned w = 0; w < wave; w++) excl += wtot[i][w];
    if (!active) continue;
    out[res->off[3 * i] + r] = differs ? 1 : 0;
    if (differs) {
      const uint2 base = wg_base[(size_t)blockIdx.x * fmt.n_fields + i];
      out[res->off[3 * i + 2] + base.y + (excl & ((1u << HDR_CNT_BITS) - 1u))] = (uint8_t)len;
      uint8_t *dst = out + res->off[3 * i + 1] + base.x + (excl >> HDR_CNT_BITS);
      for (unsigned k = 0; k < len; k++) dst[k] = fs[k];
    }
  }
}

}  // namespace

size_t fq_headers_bound(size_t raw_len, size_t n_recs, size_t n_bases, unsigned n_fields) {
  // per record and field: a flag and a length byte or four bytes of a difference; content: at most every header byte once
  const size_t text = raw_len > 2 * n_bases ? raw_len - 2 * n_bases : 0;
  return n_recs * 4 * (size_t)n_fields + text + 64;
}

// Queues the three kernels and the copy of the result words behind whatever `st` holds (the chunk and its record
// table arrive or are built on that stream); hs.host_res is valid once `st` is through.
int fq_headers_launch(hipStream_t st, const uint8_t *raw_dev, size_t raw_len, const fqgpu_rec *recs_dev, size_t n_recs, size_t n_bases,
                      const uint8_t *field_types, const char *separators, unsigned n_fields, const uint8_t *first_header,
                      size_t first_header_len, HdrScratch &hs) {
  if (!n_fields || n_fields > FQGPU_HDR_MAX_FIELDS || !n_recs || n_recs >= 0xFFFFFF00ull || !first_header_len || first_header_len > 0xFFFFu)
    return FQGPU_E_ARG;
  HdrFormat fmt;
  memset(&fmt, 0, sizeof(fmt));
  fmt.n_fields = n_fields;
  for (unsigned i = 0; i < n_fields; i++) {
    if (field_types[i] > 1) return FQGPU_E_ARG;
    fmt.type[i] = field_types[i];
    fmt.sep[i] = i + 1 < n_fields ? (uint8_t)separators[i] : 0;
  }
  const unsigned n_wg = (unsigned)((n_recs + HDR_THREADS - 1) / HDR_THREADS);
  const size_t bound = fq_headers_bound(raw_len, n_recs, n_bases, n_fields);
  int rc;
  if ((rc = hs.wg_sum.reserve((size_t)n_wg * n_fields * 4)) || (rc = hs.wg_base.reserve((size_t)n_wg * n_fields * 8)) ||
      (rc = hs.out.reserve(bound)) || (rc = hs.first.reserve(first_header_len + 64)) || (rc = hs.res.reserve(sizeof(HdrResult))))
    return rc;
  if (!hs.host_res) FQ_HIP(hipHostMalloc(reinterpret_cast<void **>(&hs.host_res), sizeof(HdrResult), hipHostMallocPortable));
  if (!hs.host_first) FQ_HIP(hipHostMalloc(reinterpret_cast<void **>(&hs.host_first), 0x10000, hipHostMallocPortable));
  memcpy(hs.host_first, first_header, first_header_len);  // (the caller's header may be gone before the copy runs)
  HdrResult *res = hs.res.as<HdrResult>();
  FQ_HIP(hipMemsetAsync(res, 0xFF, sizeof(unsigned long long), st));  // first_error = none
  FQ_HIP(hipMemcpyAsync(hs.first.p, hs.host_first, first_header_len, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_hdr_count, dim3(n_wg), dim3(HDR_THREADS), 0, st, raw_dev, recs_dev, (unsigned)n_recs, hs.first.as<uint8_t>(),
                     (unsigned)first_header_len, fmt, hs.wg_sum.as<uint32_t>(), res);
  hipLaunchKernelGGL(k_hdr_layout, dim3(1), dim3(HDR_THREADS), 0, st, hs.wg_sum.as<uint32_t>(), n_wg, (unsigned)n_recs, fmt,
                     hs.wg_base.as<uint2>(), res, (unsigned long long)bound);
  hipLaunchKernelGGL(k_hdr_write, dim3(n_wg), dim3(HDR_THREADS), 0, st, raw_dev, recs_dev, (unsigned)n_recs, hs.first.as<uint8_t>(),
                     (unsigned)first_header_len, fmt, hs.wg_base.as<uint2>(), res, hs.out.as<uint8_t>(), (unsigned long long)bound);
  FQ_HIP(hipGetLastError());
  FQ_HIP(hipMemcpyAsync(hs.host_res, res, sizeof(HdrResult), hipMemcpyDeviceToHost, st));
  hs.n_fields = n_fields;
  hs.bound = bound;
  return FQGPU_OK;
}
