"""ctypes binding of include/fqgpu.h (one Python name per C entry point)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
# FQGPU_LIB: another build of the same ABI (tools/traffic_experiment.py loads the -DFQGPU_EXPERIMENTS one)
LIB_PATH = os.environ.get("FQGPU_LIB") or os.path.join(HERE, "libfqgpu.so")

SEQ_MODELS, SEQ_ALPHA = 256, 4
QUAL_MODELS, QUAL_ALPHA = 8192, 64
REC_DTYPE = np.dtype([("seq_off", "<u4"), ("qual_off", "<u4"), ("len", "<u4")])
SEQ_FT_DTYPE = np.dtype(
    [("norm", "<i2", (SEQ_MODELS, SEQ_ALPHA)), ("logs", "<u4", (SEQ_MODELS,)), ("max_log", "<u4")]
)
QUAL_FT_DTYPE = np.dtype(
    [("norm", "<i2", (QUAL_MODELS, QUAL_ALPHA)), ("logs", "<u4", (QUAL_MODELS,)), ("max_log", "<u4")]
)
F_WRITE_BACK_N = 1
F_DECODE_INDEX = 2  # extension: the encode also leaves a decode index per stream

ERRORS = {0: "OK", -1: "OVERFLOW", -2: "SHORT_READ", -3: "CORRUPT", -4: "ARG", -5: "NO_DEVICE",
          -6: "NOMEM", -7: "HIP"}


class FqgpuError(RuntimeError):
    def __init__(self, code, where=""):
        self.code = code
        msg = lib().fqgpu_strerror(code).decode() if _lib is not None else ""
        super().__init__("fqgpu %s: %s (%d) %s" % (where, ERRORS.get(code, "?"), code, msg))


class Timing(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("kernel_ms", C.c_float * 32), ("kernel_calls", C.c_int * 32),
                ("kernel_name", C.c_char_p * 32), ("n_kernels", C.c_int)]


def lib_path():
    return LIB_PATH


def build(verbose=False):
    """Compile every HIP translation unit for gfx950 and link libfqgpu.so in-tree."""
    cmd = ["make", "-C", os.path.join(HERE, "csrc"), "-j", "6"]
    subprocess.run(cmd, check=True, stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


_lib = None

_PROTOS = {
    # name: (restype, argtypes)
    "fqgpu_device_count": (C.c_int, []),
    "fqgpu_strerror": (C.c_char_p, [C.c_int]),
    "fqgpu_version": (C.c_char_p, []),
    "fqgpu_bound_seq": (C.c_size_t, [C.c_size_t]),
    "fqgpu_bound_qual": (C.c_size_t, [C.c_size_t]),
    "fqgpu_freq_tables": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fqgpu_tables_from_counts": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fqgpu_ctx_create": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "fqgpu_ctx_destroy": (None, [C.c_void_p]),
    "fqgpu_ctx_set_chain_params": (C.c_int, [C.c_void_p, C.c_uint, C.c_uint]),
    "fqgpu_ctx_dump_tables": (C.c_int, [C.c_void_p, C.c_int, C.c_uint, C.c_void_p, C.c_size_t,
                                        C.c_void_p, C.c_size_t]),
    "fqgpu_encode_block": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                     C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                     C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.POINTER(C.c_size_t), C.c_uint]),
    "fqgpu_encode_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint,
                                     C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "fqgpu_encode_records": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "fqgpu_encode_wait": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "fqgpu_encode_cancel": (C.c_int, [C.c_void_p]),
    "fqgpu_encode_headers_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint, C.c_void_p, C.c_size_t]),
    "fqgpu_encode_headers_wait": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "fqgpu_encode_headers_end": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "fqgpu_encode_end": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                   C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_size_t, C.POINTER(C.c_size_t)]),
    "fqgpu_decode_block": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                     C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                     C.c_size_t, C.c_void_p, C.c_size_t]),
    "fqgpu_encode_index": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "fqgpu_decode_block_indexed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                             C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                             C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "fqgpu_dblock_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                      C.POINTER(C.c_void_p)]),
    "fqgpu_dblock_create_from_raw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "fqgpu_dblock_records": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                       C.POINTER(C.c_size_t)]),
    "fqgpu_dblock_destroy": (None, [C.c_void_p]),
    "fqgpu_dblock_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint]),
    "fqgpu_dblock_wipe": (C.c_int, [C.c_void_p, C.c_void_p]),
    "fqgpu_dblocks_decode": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t]),
    "fqgpu_sync": (C.c_int, [C.c_void_p]),
    "fqgpu_dblock_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                      C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "fqgpu_dblock_longest_chain": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]),
    "fqgpu_dblock_qual_segment_classes": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]),
    "fqgpu_ctx_set_index_stride": (C.c_int, [C.c_void_p, C.c_uint]),
    "fqgpu_dblock_index_bytes": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]),
    "fqgpu_dblock_fetch_index": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "fqgpu_dblock_load_index": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "fqgpu_ctx_set_lanes": (C.c_int, [C.c_void_p, C.c_uint]),
    "fqgpu_ctx_reserve": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t]),
    "fqgpu_ctx_set_seq_segment": (C.c_int, [C.c_void_p, C.c_uint]),
    "fqgpu_ctx_set_seq_group": (C.c_int, [C.c_void_p, C.c_uint, C.c_uint]),
    "fqgpu_dblock_fetch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "fqgpu_dblock_load_streams": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                            C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]),
    "fqgpu_ctx_enable_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "fqgpu_ctx_last_timing": (C.c_int, [C.c_void_p, C.POINTER(Timing)]),
    "fqgpu_ctx_timing_only": (C.c_int, [C.c_void_p, C.c_char_p]),
    "fqgpu_host_alloc": (C.c_void_p, [C.c_size_t]),
    "fqgpu_host_free": (None, [C.c_void_p]),
    "fqgpu_host_trim": (C.c_size_t, []),
    "fqgpu_memcompress_bound": (C.c_size_t, [C.c_size_t]),
    "fqgpu_memcompress": (C.c_size_t, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "fqgpu_memdecompress": (C.c_size_t, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "fqgpu_parse_fastq": (C.c_long, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "fqgpu_synth_fastq": (C.c_size_t, [C.c_void_p, C.c_size_t, C.c_int, C.c_uint64, C.c_uint64,
                                       C.POINTER(C.c_uint64)]),
}
EXPORTS = sorted(_PROTOS)


def lib():
    """The loaded extension.  Raises if libfqgpu.so has not been built: no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libfqgpu.so is not built (run __graft_entry__.build() or "
                               "`make -C fqcomp28_amd/csrc`); the product path has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(L, name)  # AttributeError = a symbol the header declares is missing
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _check(rc, where):
    if rc != 0:
        raise FqgpuError(rc, where)


def device_count():
    return lib().fqgpu_device_count()


def bound_seq(n):
    return lib().fqgpu_bound_seq(n)


def bound_qual(n):
    return lib().fqgpu_bound_qual(n)


def parse_fastq(raw):
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    n = lib().fqgpu_parse_fastq(_p(raw), raw.size, None, 0)
    if n < 0:
        raise FqgpuError(-4, "parse_fastq")
    recs = np.zeros(n, dtype=REC_DTYPE)
    lib().fqgpu_parse_fastq(_p(raw), raw.size, _p(recs), n)
    return recs


def memcompress(data):
    """misc-stream compressor (host code; own format, see fq_misc.cpp) -> uint8 array"""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    out = np.empty(lib().fqgpu_memcompress_bound(data.size), dtype=np.uint8)
    n = lib().fqgpu_memcompress(_p(out), out.size, _p(data), data.size)
    return out[:n].copy()


def memdecompress(cdata, original_size):
    cdata = np.ascontiguousarray(cdata, dtype=np.uint8)
    out = np.empty(original_size, dtype=np.uint8)
    n = lib().fqgpu_memdecompress(_p(out), out.size, _p(cdata), cdata.size)
    if n == 2 ** 64 - 1 or (cdata.size and n != original_size):
        raise FqgpuError(-3, "memdecompress")
    return out[:0] if cdata.size == 0 else out


def pinned_empty(n_bytes):
    """uint8 array in page-locked host memory (fqgpu_host_alloc); freed when the array dies"""
    p = lib().fqgpu_host_alloc(max(1, n_bytes))
    if not p:
        raise MemoryError("fqgpu_host_alloc")
    buf = (C.c_uint8 * max(1, n_bytes)).from_address(p)
    arr = np.frombuffer(buf, dtype=np.uint8, count=n_bytes)
    import weakref
    weakref.finalize(buf, lib().fqgpu_host_free, p)
    return arr


def synth_fastq(n_bytes, mode, seed=28, first_read_id=0):
    """-> (uint8 array of whole records, number of reads)"""
    buf = np.empty(n_bytes, dtype=np.uint8)
    n_reads = C.c_uint64(0)
    used = lib().fqgpu_synth_fastq(_p(buf), n_bytes, mode, seed, first_read_id, C.byref(n_reads))
    return buf[:used], int(n_reads.value)


def freq_tables(raw, recs, device=0, want_counts=False):
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    sft = np.zeros(1, dtype=SEQ_FT_DTYPE)
    qft = np.zeros(1, dtype=QUAL_FT_DTYPE)
    sc = np.zeros((SEQ_MODELS, SEQ_ALPHA), dtype=np.uint32) if want_counts else None
    qc = np.zeros((QUAL_MODELS, QUAL_ALPHA), dtype=np.uint32) if want_counts else None
    _check(lib().fqgpu_freq_tables(device, _p(raw), raw.size, _p(recs), len(recs), _p(sft), _p(qft),
                                   _p(sc), _p(qc)), "freq_tables")
    return (sft, qft, sc, qc) if want_counts else (sft, qft)


def tables_from_counts(seq_counts, qual_counts, device=0):
    sft = np.zeros(1, dtype=SEQ_FT_DTYPE)
    qft = np.zeros(1, dtype=QUAL_FT_DTYPE)
    sc = np.ascontiguousarray(seq_counts, dtype=np.uint32)
    qc = np.ascontiguousarray(qual_counts, dtype=np.uint32)
    _check(lib().fqgpu_tables_from_counts(device, _p(sc), _p(qc), _p(sft), _p(qft)), "tables_from_counts")
    return sft, qft


class DBlock:
    """Device-resident block (fqgpu_dblock)."""

    def __init__(self, ctx, raw, recs=None):
        """recs=None: the record table is built on the GPU (fqgpu_dblock_create_from_raw)."""
        raw = np.ascontiguousarray(raw, dtype=np.uint8)
        self.ctx = ctx
        h = C.c_void_p()
        if recs is None:
            _check(lib().fqgpu_dblock_create_from_raw(ctx.h, _p(raw), raw.size, C.byref(h)), "dblock_create_from_raw")
            self.h = h
            n, rl = C.c_size_t(), C.c_size_t()
            _check(lib().fqgpu_dblock_records(ctx.h, h, None, 0, C.byref(n), C.byref(rl)), "dblock_records")
            self.raw_len, self.n_recs = rl.value, n.value
        else:
            recs = np.ascontiguousarray(recs, dtype=REC_DTYPE)
            self.raw_len, self.n_recs = raw.size, len(recs)
            _check(lib().fqgpu_dblock_create(ctx.h, _p(raw), raw.size, _p(recs), len(recs), C.byref(h)),
                   "dblock_create")
            self.h = h

    def records(self):
        recs = np.zeros(self.n_recs, dtype=REC_DTYPE)
        _check(lib().fqgpu_dblock_records(self.ctx.h, self.h, _p(recs), self.n_recs, None, None), "dblock_records")
        return recs

    def close(self):
        if getattr(self, "h", None):
            lib().fqgpu_dblock_destroy(self.h)
            self.h = None

    __del__ = close

    def encode(self, flags=0):
        _check(lib().fqgpu_dblock_encode(self.ctx.h, self.h, flags), "dblock_encode")

    def fetch_index(self, stream):
        """decode index of one stream (0 = sequence, 1 = quality) of the last encode with F_DECODE_INDEX"""
        n = C.c_size_t(0)
        _check(lib().fqgpu_dblock_index_bytes(self.h, stream, C.byref(n)), "dblock_index_bytes")
        out = np.zeros(n.value, dtype=np.uint8)
        if n.value:
            _check(lib().fqgpu_dblock_fetch_index(self.ctx.h, self.h, stream, _p(out), out.size), "dblock_fetch_index")
        return out

    def load_index(self, stream, data):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        return lib().fqgpu_dblock_load_index(self.ctx.h, self.h, stream, _p(data) if data.size else None, data.size)

    def wipe(self):
        _check(lib().fqgpu_dblock_wipe(self.ctx.h, self.h), "dblock_wipe")

    def status(self):
        a, b, c, d = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
        rc = lib().fqgpu_dblock_status(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        return rc, dict(seq_len=a.value, qual_len=b.value, n_pos_len=c.value, n_bases=d.value)

    def longest_chain(self):
        """(seq, qual): longest serial run of symbols one lane walked in the last encode."""
        a, b = C.c_uint(), C.c_uint()
        _check(lib().fqgpu_dblock_longest_chain(self.h, C.byref(a), C.byref(b)), "dblock_longest_chain")
        return a.value, b.value

    def qual_segment_classes(self):
        """segments of the quality chains of the last encode by class -> dict(transparent, anchored, uniform, opaque)"""
        c = (C.c_size_t * 4)()
        _check(lib().fqgpu_dblock_qual_segment_classes(self.ctx.h, self.h, c), "dblock_qual_segment_classes")
        return dict(transparent=c[0], anchored=c[1], uniform=c[2], opaque=c[3])

    def fetch(self, raw=False):
        rc, st = self.status()
        _check(rc, "dblock_status")
        seq = np.zeros(st["seq_len"], dtype=np.uint8)
        qual = np.zeros(st["qual_len"], dtype=np.uint8)
        rl = np.zeros(self.n_recs, dtype=np.uint16)
        nc = np.zeros(self.n_recs, dtype=np.uint16)
        npos = np.zeros(st["n_pos_len"], dtype=np.uint16)
        rw = np.zeros(self.raw_len, dtype=np.uint8) if raw else None
        _check(lib().fqgpu_dblock_fetch(self.ctx.h, self.h, _p(seq), _p(qual), _p(rl), _p(nc), _p(npos),
                                        _p(rw)), "dblock_fetch")
        return dict(seq=seq, qual=qual, readlens=rl, n_count=nc, n_pos=npos, raw=rw)

    def fetch_raw(self):
        rw = np.zeros(self.raw_len, dtype=np.uint8)
        _check(lib().fqgpu_dblock_fetch(self.ctx.h, self.h, None, None, None, None, None, _p(rw)),
               "dblock_fetch")
        return rw

    def load_streams(self, seq, qual, n_count, n_pos):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        qual = np.ascontiguousarray(qual, dtype=np.uint8)
        n_count = np.ascontiguousarray(n_count, dtype=np.uint16)
        n_pos = np.ascontiguousarray(n_pos, dtype=np.uint16)
        _check(lib().fqgpu_dblock_load_streams(self.ctx.h, self.h, _p(seq), seq.size, _p(qual), qual.size,
                                               _p(n_count), _p(n_pos), n_pos.size), "load_streams")


class Context:
    """fqgpu_ctx: the device-side equivalent of a reference Compression/DecompressionWorkspace."""

    def __init__(self, seq_ft, qual_ft, device=0):
        self.seq_ft = np.ascontiguousarray(seq_ft)
        self.qual_ft = np.ascontiguousarray(qual_ft)
        assert self.seq_ft.nbytes == SEQ_FT_DTYPE.itemsize and self.qual_ft.nbytes == QUAL_FT_DTYPE.itemsize
        h = C.c_void_p()
        _check(lib().fqgpu_ctx_create(device, _p(self.seq_ft), _p(self.qual_ft), C.byref(h)), "ctx_create")
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            lib().fqgpu_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    def set_chain_params(self, segment=0, seq_generic=False, seq_segment=None, seq_group=None):
        """seq_group: (max_segments, min_groups) of fqgpu_ctx_set_seq_group, or max_segments alone."""
        flags = 1 if seq_generic else 0
        if seq_segment is not None:
            _check(lib().fqgpu_ctx_set_seq_segment(self.h, seq_segment), "set_seq_segment")
        if seq_group is not None:
            q, g = seq_group if isinstance(seq_group, tuple) else (seq_group, 0)
            _check(lib().fqgpu_ctx_set_seq_group(self.h, q, g), "set_seq_group")
        _check(lib().fqgpu_ctx_set_chain_params(self.h, segment, flags), "set_chain_params")

    def set_index_stride(self, symbols):
        _check(lib().fqgpu_ctx_set_index_stride(self.h, symbols), "set_index_stride")

    def set_lanes(self, lanes):
        _check(lib().fqgpu_ctx_set_lanes(self.h, lanes), "set_lanes")

    def sync(self):
        _check(lib().fqgpu_sync(self.h), "sync")

    def enable_timing(self, on=True, only=None):
        """HIP-event spans around the kernel groups; only: restrict them to one group's label"""
        _check(lib().fqgpu_ctx_timing_only(self.h, only.encode() if only else None), "timing_only")
        _check(lib().fqgpu_ctx_enable_timing(self.h, 1 if on else 0), "enable_timing")

    def last_timing(self):
        t = Timing()
        _check(lib().fqgpu_ctx_last_timing(self.h, C.byref(t)), "last_timing")
        return t.total_ms, [(t.kernel_name[i].decode(), t.kernel_ms[i], t.kernel_calls[i])
                            for i in range(t.n_kernels)]

    def dump_tables(self, stream, model):
        alpha = QUAL_ALPHA if stream else SEQ_ALPHA
        ct = np.zeros(1 + 2048 + 2 * alpha, dtype=np.uint32)
        dt = np.zeros(1 + 4096, dtype=np.uint32)
        _check(lib().fqgpu_ctx_dump_tables(self.h, stream, model, _p(ct), ct.size, _p(dt), dt.size),
               "dump_tables")
        log = int(ct[0] & 0xFFFF)
        return ct[: 1 + (1 << (log - 1)) + 2 * alpha].copy(), dt[: 1 + (1 << log)].copy()

    def dblock(self, raw, recs=None):
        return DBlock(self, raw, recs)

    def decode_dblocks(self, blocks):
        arr = (C.c_void_p * len(blocks))(*[b.h for b in blocks])
        _check(lib().fqgpu_dblocks_decode(self.h, arr, len(blocks)), "dblocks_decode")

    @staticmethod
    def host_buffers(n_recs, n_bases, seq_cap=None, qual_cap=None):
        """Output buffers a worker keeps across chunks (the reference reuses its CompressedBuffersDst)."""
        return dict(seq=np.zeros(bound_seq(n_bases) if seq_cap is None else seq_cap, dtype=np.uint8),
                    qual=np.zeros(bound_qual(n_bases) if qual_cap is None else qual_cap, dtype=np.uint8),
                    readlens=np.zeros(n_recs, dtype=np.uint16), n_count=np.zeros(n_recs, dtype=np.uint16),
                    n_pos=np.zeros(n_bases + 1, dtype=np.uint16))

    def encode_block_into(self, raw, recs, bufs, flags=0):
        """fqgpu_encode_block on the caller's arrays, nothing copied or allocated on the Python side.
        raw is written to when flags has F_WRITE_BACK_N.  -> (rc, seq_len, qual_len, n_pos_len)"""
        sl, ql, nn = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        rc = lib().fqgpu_encode_block(self.h, _p(raw), raw.size, _p(recs), len(recs), _p(bufs["seq"]),
                                      bufs["seq"].size, C.byref(sl), _p(bufs["qual"]), bufs["qual"].size,
                                      C.byref(ql), _p(bufs["readlens"]), _p(bufs["n_count"]), _p(bufs["n_pos"]),
                                      bufs["n_pos"].size, C.byref(nn), flags)
        return rc, sl.value, ql.value, nn.value

    def encode_block(self, raw, recs, flags=0, seq_cap=None, qual_cap=None):
        """Host-pointer call (fqgpu_encode_block) -> dict like the oracle's."""
        raw = np.array(raw, dtype=np.uint8, copy=True)
        recs = np.ascontiguousarray(recs, dtype=REC_DTYPE)
        bufs = self.host_buffers(len(recs), int(recs["len"].sum()), seq_cap, qual_cap)
        rc, sl, ql, nn = self.encode_block_into(raw, recs, bufs, flags)
        return dict(rc=rc, seq=bufs["seq"][:sl].copy(), qual=bufs["qual"][:ql].copy(), readlens=bufs["readlens"],
                    n_count=bufs["n_count"], n_pos=bufs["n_pos"][:nn].copy(), raw_after=raw)

    def encode_raw(self, raw, flags=0, recs=None, header_format=None):
        """The two-halves call on an UNPARSED chunk (fqgpu_encode_begin / _records / _wait / _end): the
        record table comes back from the GPU.  -> dict like encode_block's, plus recs and used_len.
        header_format = (types, separators, first_header) -- types[i] 0 = NUMERIC / 1 = STRING, separators as bytes,
        first_header with its '@' -- also codes the header fields on the device (fqgpu_encode_headers_*):
        `header_fields` = [(flags, content, lengths) per field] or, for a header that cannot be coded,
        `headers_rc` = FQGPU_E_HEADER and `bad_record`."""
        raw = np.array(raw, dtype=np.uint8, copy=True)
        n, nb, used = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        if recs is not None:
            recs = np.ascontiguousarray(recs, dtype=REC_DTYPE)
        rc = lib().fqgpu_encode_begin(self.h, _p(raw), raw.size, _p(recs) if recs is not None else None,
                                      0 if recs is None else len(recs), flags, C.byref(n), C.byref(nb), C.byref(used))
        if rc:
            return dict(rc=rc)
        hdr = {}
        if header_format is not None:
            types, seps, first = header_format
            types = np.ascontiguousarray(types, dtype=np.uint8)
            first = np.frombuffer(bytes(first), dtype=np.uint8)
            rc = lib().fqgpu_encode_headers_begin(self.h, _p(types), bytes(seps), len(types), _p(first), first.size)
            if rc:
                lib().fqgpu_encode_cancel(self.h)
                return dict(rc=rc)
        table = np.zeros(n.value, dtype=REC_DTYPE)
        rc = lib().fqgpu_encode_records(self.h, _p(table), len(table))
        if rc:
            return dict(rc=rc)
        if header_format is not None:
            sizes = np.zeros((len(types), 3), dtype=np.uint32)
            total, bad = C.c_size_t(0), C.c_size_t(0)
            rc = lib().fqgpu_encode_headers_wait(self.h, _p(sizes), C.byref(total), C.byref(bad))
            hdr["headers_rc"] = rc
            if rc:
                hdr["bad_record"] = bad.value
            else:
                out = np.zeros(max(total.value, 1), dtype=np.uint8)
                _check(lib().fqgpu_encode_headers_end(self.h, _p(out), out.size), "fqgpu_encode_headers_end")
                fields, at = [], 0
                for f in range(len(types)):
                    parts = []
                    for k in range(3):
                        parts.append(out[at:at + int(sizes[f, k])].copy())
                        at += int(sizes[f, k])
                    fields.append(tuple(parts))
                assert at == total.value
                hdr["header_fields"] = fields
        sl, ql, nn = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        rc = lib().fqgpu_encode_wait(self.h, C.byref(sl), C.byref(ql), C.byref(nn))
        if rc:
            return dict(rc=rc)
        seq, qual = np.zeros(sl.value, np.uint8), np.zeros(ql.value, np.uint8)
        readlens, n_count, n_pos = np.zeros(n.value, np.uint16), np.zeros(n.value, np.uint16), np.zeros(nn.value, np.uint16)
        rc = lib().fqgpu_encode_end(self.h, _p(raw), _p(seq), seq.size, C.byref(sl), _p(qual), qual.size, C.byref(ql),
                                    _p(readlens), _p(n_count), _p(n_pos), n_pos.size, C.byref(nn))
        if rc == 0 and (flags & F_DECODE_INDEX):
            hdr["index"] = []
            for s in (0, 1):
                n_idx = C.c_size_t(0)
                _check(lib().fqgpu_encode_index(self.h, s, None, 0, C.byref(n_idx)), "fqgpu_encode_index")
                idx = np.zeros(n_idx.value, dtype=np.uint8)
                _check(lib().fqgpu_encode_index(self.h, s, _p(idx) if idx.size else None, idx.size, C.byref(n_idx)), "fqgpu_encode_index")
                hdr["index"].append(idx)
        return dict(rc=rc, seq=seq, qual=qual, readlens=readlens, n_count=n_count, n_pos=n_pos, raw_after=raw,
                    recs=table, used_len=used.value, n_bases=nb.value, **hdr)

    def decode_block(self, seq, qual, n_count, n_pos, recs, raw_skeleton, index=None):
        """index = (sequence index, quality index) as encode_raw(flags=F_DECODE_INDEX) returns them:
        fqgpu_decode_block_indexed, every stream decoded from all its snapshots at once"""
        out = np.array(raw_skeleton, dtype=np.uint8, copy=True)
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        qual = np.ascontiguousarray(qual, dtype=np.uint8)
        n_count = np.ascontiguousarray(n_count, dtype=np.uint16)
        n_pos = np.ascontiguousarray(n_pos, dtype=np.uint16)
        recs = np.ascontiguousarray(recs, dtype=REC_DTYPE)
        if index is not None:
            si, qi = (np.ascontiguousarray(x, dtype=np.uint8) for x in index)
            rc = lib().fqgpu_decode_block_indexed(self.h, _p(seq), seq.size, _p(qual), qual.size, _p(n_count), n_count.size,
                                                  _p(n_pos), n_pos.size, _p(recs), len(recs), _p(out), out.size,
                                                  _p(si) if si.size else None, si.size, _p(qi) if qi.size else None, qi.size)
            return rc, out
        rc = lib().fqgpu_decode_block(self.h, _p(seq), seq.size, _p(qual), qual.size, _p(n_count), n_count.size,
                                      _p(n_pos), n_pos.size, _p(recs), len(recs), _p(out), out.size)
        return rc, out
