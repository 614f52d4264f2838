// Part of encode.hip (included there, inside its anonymous namespace): decode index kernels (extension).

// ---- decode index (extension, include/fqgpu.h FQGPU_F_DECODE_INDEX) ------------------------
// Snapshot k sits at encode index e = k * stride (a multiple of both partition tile sizes).
// k_index_meta: bit position (= bit offset of packing tile e / 4096), the bytes in front of
// symbol e - 1 in its record.  k_index_states: the state of every context at that point = the
// state in front of the context's first symbol at or behind e, found by walking from the entry
// state of the segment that holds it (at most one segment); one lane per (context, snapshot),
// the context's CTable in LDS.
template <class M>
__global__ void __launch_bounds__(256)
k_index_meta(const uint8_t *__restrict__ raw, const fqgpu_rec *__restrict__ recs,
             const uint32_t *__restrict__ rec_start, unsigned R, unsigned n_sym, unsigned stride,
             const unsigned long long *__restrict__ tile_bit_base, unsigned bit_base_unit, uint8_t *__restrict__ index) {
  const unsigned n_snap = n_sym ? (n_sym - 1) / stride : 0u;
  const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;  // 0: header, 1 .. n_snap: snapshots
  if (k == 0) {
    FqIndexHeader h;
    h.magic = FQ_INDEX_MAGIC; h.stream = M::STREAM; h.stride = stride; h.n_snap = n_snap;
    h.n_sym = n_sym; h.reserved = 0;
    *reinterpret_cast<FqIndexHeader *>(index) = h;
    return;
  }
  if (k > n_snap) return;
  const unsigned e = k * stride;
  uint8_t *snap = index + sizeof(FqIndexHeader) + (size_t)(k - 1) * (FQ_INDEX_SNAP_HEAD + 2 * (size_t)M::B);
  *reinterpret_cast<unsigned long long *>(snap) = tile_bit_base[e / bit_base_unit];  // stride is a multiple of both the packing tile and the sorted tile
  // symbol e - 1: record r, position p (encode order walks a record from its last position)
  const unsigned r = fq_locate(rec_start, 0, R - 1, e - 1);
  const fqgpu_rec rec = recs[r];
  const unsigned p = rec.len - 1u - (e - 1u - rec_start[r]);
  const uint8_t *line = raw + (M::STREAM == 0 ? rec.seq_off : rec.qual_off);
  unsigned packed = 0;
  for (unsigned i = 0; i < 4; i++) packed |= (p >= i + 1 ? (unsigned)line[p - 1 - i] : 0xFFu) << (8 * i);
  reinterpret_cast<uint32_t *>(snap)[2] = packed;
  reinterpret_cast<uint32_t *>(snap)[3] = 0;
}

template <class M>
__global__ void __launch_bounds__(64)
k_index_states(const uint8_t *__restrict__ sorted_sym, const uint32_t *__restrict__ arrays,
               const uint32_t *__restrict__ tile_base, unsigned T, unsigned n_sym, unsigned stride,
               const uint32_t *__restrict__ seg_prefix, const uint16_t *__restrict__ entry, int entry_is_xo,
               unsigned S, const uint32_t *__restrict__ ct, const uint32_t *__restrict__ ct_off,
               const uint16_t *__restrict__ final_state, uint8_t *__restrict__ index) {
  extern __shared__ uint32_t lds[];
  constexpr unsigned B = M::B;
  const unsigned c = blockIdx.x;
  const unsigned n_snap = n_sym ? (n_sym - 1) / stride : 0u;
  const unsigned k = blockIdx.y * 64 + fq_lane() + 1;
  const unsigned n = arrays[c], run0 = arrays[B + c];
  if (n == 0) {  // (uniform) a context without symbols keeps its initial state: decoder state 0
    if (k <= n_snap)
      reinterpret_cast<uint16_t *>(index + sizeof(FqIndexHeader) + (size_t)(k - 1) * (FQ_INDEX_SNAP_HEAD + 2 * (size_t)B) +
                                   FQ_INDEX_SNAP_HEAD)[c] = 0;
    return;
  }
  const LdsCTable t = stage_ctable<M>(lds, ct + ct_off[c]);
  if (k > n_snap) return;
  const unsigned size = 1u << t.log;
  const unsigned rel = tile_base[(size_t)((k * stride) / T) * B + c] - run0;  // symbols of c in front of e
  unsigned x;
  {
    if (rel >= n) {
      x = final_state[c];
    } else {
      const unsigned seg = rel / S;
      const unsigned ev = entry[seg_prefix[c] + seg];
      x = entry_is_xo ? size + (ev >> 1) : ev;
      const uint8_t *sym = sorted_sym + run0;
      for (unsigned i = seg * S; i < rel; i++) (void)chain_step(t, x, sym[i] & (unsigned)(M::A - 1));
    }
  }
  uint16_t *st = reinterpret_cast<uint16_t *>(index + sizeof(FqIndexHeader) + (size_t)(k - 1) * (FQ_INDEX_SNAP_HEAD + 2 * (size_t)B) +
                                              FQ_INDEX_SNAP_HEAD);
  st[c] = (uint16_t)(x - size);
}
