// GPU FASTQ record parser -- SURVEY.md 8(f) "next" row 1: replaces FastqReader::parseRecords
// (reference src/fastq_io.cpp:67-125), which the reference runs serially under the reader
// mutex (src/fastq_io.cpp:29-52).  On the device it is three HBM-bound passes:
//   k_nl_count  : newlines per 4 KB chunk (16-byte loads)
//   scan        : chunk bases
//   k_nl_write  : byte offset of every newline, in order
//   k_records   : every 4 consecutive lines = one record {seq_off, qual_off, len}; checks
//                 '@' / '+' line starts, equal sequence/quality lengths, u16 length; sums
//                 bases and N's (sizes of the coded block's buffers)
// A trailing partial record is ignored, like the reference's carry-over to the next chunk.
#include "fqgpu_internal.h"

#include <utility>

namespace {

constexpr unsigned PCHUNK = 4096;  // bytes per workgroup: 256 threads x 16 bytes

__device__ __forceinline__ unsigned count_nl16(const uint4 v, unsigned limit) {
  // number of '\n' among the first `limit` of 16 bytes
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
  unsigned n = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) n += ((unsigned)i < limit) && (((w[i >> 2] >> (8 * (i & 3))) & 0xFFu) == '\n');
  return n;
}

__global__ void __launch_bounds__(256)
k_nl_count(const uint8_t *__restrict__ raw, size_t len, uint32_t *__restrict__ chunk_count) {
  __shared__ unsigned wsum[4];
  const size_t off = (size_t)blockIdx.x * PCHUNK + (size_t)threadIdx.x * 16;
  unsigned n = 0;
  if (off < len) {
    const uint4 v = *reinterpret_cast<const uint4 *>(raw + off);  // buffer is padded to 64 bytes
    n = count_nl16(v, (unsigned)min((size_t)16, len - off));
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) n += __shfl_xor(n, d);
  if (fq_lane() == 0) wsum[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) chunk_count[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ void __launch_bounds__(256)
k_nl_write(const uint8_t *__restrict__ raw, size_t len, const uint32_t *__restrict__ chunk_base,
           uint32_t *__restrict__ nl_pos) {
  __shared__ unsigned wsum[4];
  const size_t off = (size_t)blockIdx.x * PCHUNK + (size_t)threadIdx.x * 16;
  uint4 v = make_uint4(0, 0, 0, 0);
  unsigned limit = 0;
  if (off < len) {
    v = *reinterpret_cast<const uint4 *>(raw + off);
    limit = (unsigned)min((size_t)16, len - off);
  }
  const unsigned n = count_nl16(v, limit);
  unsigned inc = n;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned o = __shfl_up(inc, d);
    if (fq_lane() >= (unsigned)d) inc += o;
  }
  if (fq_lane() == 63) wsum[threadIdx.x >> 6] = inc;
  __syncthreads();
  unsigned at = chunk_base[blockIdx.x] + inc - n;
  for (unsigned w = 0; w < (threadIdx.x >> 6); w++) at += wsum[w];
  const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 16; i++)
    if ((unsigned)i < limit && ((w4[i >> 2] >> (8 * (i & 3))) & 0xFFu) == '\n') nl_pos[at++] = (uint32_t)(off + i);
}

struct ParseSummary {
  unsigned long long n_bases, n_n;
  unsigned long long used_len;  // bytes up to and including the last complete record
  unsigned int bad_format;   // line 1 not '@', line 3 not '+', seq/qual length mismatch
  unsigned int short_read;   // a read shorter than 3
  unsigned int too_long;     // a read longer than 65535 (readlen_t)
};

__global__ void __launch_bounds__(256)
k_records(const uint8_t *__restrict__ raw, const uint32_t *__restrict__ nl_pos, unsigned n_recs,
          fqgpu_rec *__restrict__ recs, ParseSummary *__restrict__ sum) {
  const unsigned r = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long bases = 0;
  if (r < n_recs) {
    const unsigned l0 = r == 0 ? 0u : nl_pos[4 * r - 1] + 1u;
    const unsigned e0 = nl_pos[4 * r], e1 = nl_pos[4 * r + 1], e2 = nl_pos[4 * r + 2], e3 = nl_pos[4 * r + 3];
    const unsigned l1 = e0 + 1, l2 = e1 + 1, l3 = e2 + 1;
    const unsigned len = e1 - l1;
    if (raw[l0] != '@' || raw[l2] != '+' || (e3 - l3) != len) atomicOr(&sum->bad_format, 1u);
    if (len > 65535u) atomicOr(&sum->too_long, 1u);
    else if (len < 3u) atomicOr(&sum->short_read, 1u);
    recs[r].seq_off = l1;
    recs[r].qual_off = l3;
    recs[r].len = len;
    bases = len;
    if (r == n_recs - 1) sum->used_len = (unsigned long long)e3 + 1ull;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) bases += __shfl_xor(bases, d);
  if (fq_lane() == 0 && bases) atomicAdd(&sum->n_bases, bases);
}

// number of N bases (= entries of the n_pos side stream), one wave per record
__global__ void __launch_bounds__(256)
k_count_n(const uint8_t *__restrict__ raw, const fqgpu_rec *__restrict__ recs, unsigned R, ParseSummary *sum) {
  const unsigned waves = (gridDim.x * blockDim.x) >> 6;
  const unsigned lane = fq_lane();
  unsigned long long cnt = 0;
  for (unsigned r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < R; r += waves) {
    const fqgpu_rec rec = recs[r];
    if (rec.len > 65535u) continue;
    const uint8_t *s = raw + rec.seq_off;
    for (unsigned i = lane; i < rec.len; i += 64) cnt += s[i] == 'N';
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d);
  if (lane == 0 && cnt) atomicAdd(&sum->n_n, cnt);
}

}  // namespace

// raw_dev: device copy of the block, padded by >= 64 readable bytes.  Two steps, because the record
// table can only be sized once the lines are counted; both wait for st (a few bytes come back).
// The scratch is the caller's and only grows: a worker that parses chunk after chunk allocates
// nothing (hipFree waits for the device to be idle).
int fq_parse_count(hipStream_t st, const uint8_t *raw_dev, size_t raw_len, ParseScratch &ps, size_t *n_recs) {
  *n_recs = 0;
  if (raw_len == 0 || raw_len >= 0xFFF00000ull) return FQGPU_E_ARG;
  const size_t n_chunks = (raw_len + PCHUNK - 1) / PCHUNK;
  int rc;
  if ((rc = ps.cnt.reserve(n_chunks * 4)) || (rc = ps.base.reserve((n_chunks + 1) * 4)) || (rc = ps.sum.reserve(sizeof(ParseSummary)))) return rc;
  hipLaunchKernelGGL(k_nl_count, dim3((unsigned)n_chunks), dim3(256), 0, st, raw_dev, raw_len, ps.cnt.as<uint32_t>());
  if ((rc = fq_scan_u32_to_u32(st, ps.cnt.as<uint32_t>(), n_chunks, ps.base.as<uint32_t>(), ps.scan_tmp))) return rc;
  uint32_t total_nl = 0;
  FQ_HIP(hipMemcpyAsync(&total_nl, ps.base.as<uint32_t>() + n_chunks, 4, hipMemcpyDeviceToHost, st));
  FQ_HIP(hipStreamSynchronize(st));
  ps.total_nl = total_nl;
  *n_recs = total_nl / 4;  // a trailing partial record is ignored
  return *n_recs ? FQGPU_OK : FQGPU_E_ARG;
}

int fq_parse_records(hipStream_t st, const uint8_t *raw_dev, size_t raw_len, ParseScratch &ps, fqgpu_rec *recs_dev,
                     size_t n_recs, size_t *n_bases, size_t *n_n, size_t *used_len) {
  const size_t n_chunks = (raw_len + PCHUNK - 1) / PCHUNK;
  int rc;
  if ((rc = ps.nl_pos.reserve(((size_t)ps.total_nl + 4) * 4))) return rc;
  ParseSummary *sum = ps.sum.as<ParseSummary>();
  FQ_HIP(hipMemsetAsync(sum, 0, sizeof(ParseSummary), st));
  hipLaunchKernelGGL(k_nl_write, dim3((unsigned)n_chunks), dim3(256), 0, st, raw_dev, raw_len, ps.base.as<uint32_t>(), ps.nl_pos.as<uint32_t>());
  hipLaunchKernelGGL(k_records, dim3((unsigned)((n_recs + 255) / 256)), dim3(256), 0, st, raw_dev, ps.nl_pos.as<uint32_t>(),
                     (unsigned)n_recs, recs_dev, sum);
  const unsigned nb = (unsigned)min((n_recs + 3) / 4, (size_t)8192);
  hipLaunchKernelGGL(k_count_n, dim3(nb), dim3(256), 0, st, raw_dev, recs_dev, (unsigned)n_recs, sum);
  ParseSummary h;
  FQ_HIP(hipMemcpyAsync(&h, sum, sizeof(h), hipMemcpyDeviceToHost, st));
  FQ_HIP(hipStreamSynchronize(st));
  if (h.bad_format || h.too_long) return FQGPU_E_ARG;
  if (h.short_read) return FQGPU_E_SHORT_READ;
  if (h.n_bases >= 0xFFF00000ull) return FQGPU_E_ARG;
  *n_bases = (size_t)h.n_bases;
  *n_n = (size_t)h.n_n;
  if (used_len) *used_len = (size_t)h.used_len;
  return FQGPU_OK;
}

// the two steps with a record table allocated here; on success *recs_dev is hipMalloc'ed and the caller's
int fq_parse_on_device(hipStream_t st, const uint8_t *raw_dev, size_t raw_len, DevBuf &scan_tmp,
                       fqgpu_rec **recs_dev, size_t *n_recs, size_t *n_bases, size_t *n_n) {
  *recs_dev = nullptr;
  *n_recs = *n_bases = *n_n = 0;
  ParseScratch ps;
  std::swap(ps.scan_tmp, scan_tmp);
  size_t R = 0;
  int rc = fq_parse_count(st, raw_dev, raw_len, ps, &R);
  fqgpu_rec *recs = nullptr;
  if (!rc && !(recs = fq_dev_alloc<fqgpu_rec>(R))) rc = FQGPU_E_NOMEM;
  if (!rc) rc = fq_parse_records(st, raw_dev, raw_len, ps, recs, R, n_bases, n_n, nullptr);
  std::swap(ps.scan_tmp, scan_tmp);
  DevBuf *bufs[] = {&ps.cnt, &ps.base, &ps.sum, &ps.nl_pos};
  for (DevBuf *b : bufs) b->release();
  if (rc) { if (recs) (void)hipFree(recs); return rc; }
  *recs_dev = recs;
  *n_recs = R;
  return FQGPU_OK;
}
