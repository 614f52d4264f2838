"""ctypes binding of the CPU oracle (oracle/) for the tests.

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "libfqc_oracle.so")

SEQ_MODELS, SEQ_ALPHA = 256, 4
QUAL_MODELS, QUAL_ALPHA = 8192, 64

REC_DTYPE = np.dtype([("seq_off", "<u4"), ("qual_off", "<u4"), ("len", "<u4")])
SEQ_FT_DTYPE = np.dtype(
    [("norm", "<i2", (SEQ_MODELS, SEQ_ALPHA)), ("logs", "<u4", (SEQ_MODELS,)), ("max_log", "<u4")]
)
QUAL_FT_DTYPE = np.dtype(
    [("norm", "<i2", (QUAL_MODELS, QUAL_ALPHA)), ("logs", "<u4", (QUAL_MODELS,)), ("max_log", "<u4")]
)
assert SEQ_FT_DTYPE.itemsize == 3076 and QUAL_FT_DTYPE.itemsize == 1081348


def build_oracle():
    """(Re)build oracle/_build/libfqc_oracle.so with the committed Makefile."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)
    return ORACLE_SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        L = C.CDLL(ORACLE_SO)
        vp, sz, i = C.c_void_p, C.c_size_t, C.c_int
        L.fo_optimal_table_log.restype = C.c_uint
        L.fo_optimal_table_log.argtypes = [C.c_uint, sz, C.c_uint]
        L.fo_normalize_count.restype = i
        L.fo_normalize_count.argtypes = [vp, C.c_uint, vp, sz, C.c_uint, i]
        L.fo_ctable_words.restype = sz
        L.fo_ctable_words.argtypes = [C.c_uint, C.c_uint]
        L.fo_dtable_words.restype = sz
        L.fo_dtable_words.argtypes = [C.c_uint]
        L.fo_build_ctable.restype = i
        L.fo_build_ctable.argtypes = [vp, vp, C.c_uint, C.c_uint]
        L.fo_build_dtable.restype = i
        L.fo_build_dtable.argtypes = [vp, vp, C.c_uint, C.c_uint]
        L.fo_compress_using_ctable.restype = sz
        L.fo_compress_using_ctable.argtypes = [vp, sz, vp, sz, vp]
        L.fo_decompress_using_dtable.restype = sz
        L.fo_decompress_using_dtable.argtypes = [vp, sz, vp, sz, vp]
        L.fqo_seq_counts.restype = C.c_int
        L.fqo_seq_counts.argtypes = [vp, vp, sz, vp]
        L.fqo_qual_counts.restype = i
        L.fqo_qual_counts.argtypes = [vp, vp, sz, vp]
        L.fqo_seq_ft_from_counts.restype = i
        L.fqo_seq_ft_from_counts.argtypes = [vp, vp]
        L.fqo_qual_ft_from_counts.restype = i
        L.fqo_qual_ft_from_counts.argtypes = [vp, vp]
        L.fqo_bound_seq.restype = sz
        L.fqo_bound_seq.argtypes = [sz]
        L.fqo_bound_qual.restype = sz
        L.fqo_bound_qual.argtypes = [sz]
        L.fqo_ctx_create.restype = vp
        L.fqo_ctx_create.argtypes = [vp, vp]
        L.fqo_ctx_destroy.restype = None
        L.fqo_ctx_destroy.argtypes = [vp]
        L.fqo_encode_block.restype = i
        L.fqo_encode_block.argtypes = [vp, vp, vp, sz, vp, sz, vp, vp, sz, vp, vp, vp, vp, vp]
        L.fqo_decode_block.restype = i
        L.fqo_decode_block.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, sz, vp, sz, vp]
        L.fqo_bench_blocks.restype = C.c_double
        L.fqo_bench_blocks.argtypes = [vp, vp, i, i, vp, vp, vp, vp, i]
        _lib = L
    return _lib


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def parse_fastq(raw):
    """4-line FASTQ -> record table (numpy restatement of FastqReader::parseRecords,
    reference src/fastq_io.cpp:67-125).  raw: uint8 array ending in '\\n'."""
    raw = np.asarray(raw, dtype=np.uint8)
    nl = np.flatnonzero(raw == 10)
    assert nl.size % 4 == 0 and (raw.size == 0 or raw[-1] == 10), "not a whole 4-line FASTQ block"
    n = nl.size // 4
    nl = nl.reshape(n, 4)
    starts = np.empty((n, 4), dtype=np.int64)
    starts[:, 1:] = nl[:, :3] + 1
    starts[0, 0] = 0
    starts[1:, 0] = nl[:-1, 3] + 1
    recs = np.zeros(n, dtype=REC_DTYPE)
    recs["seq_off"] = starts[:, 1]
    recs["qual_off"] = starts[:, 3]
    recs["len"] = nl[:, 1] - starts[:, 1]
    assert np.all(nl[:, 3] - starts[:, 3] == recs["len"]), "seq/qual length mismatch"
    assert np.all(raw[starts[:, 0]] == ord("@")) and np.all(raw[starts[:, 2]] == ord("+"))
    return recs


def load_fastq(path):
    raw = np.fromfile(path, dtype=np.uint8)
    return raw, parse_fastq(raw)


def freq_tables(raw, recs):
    """Oracle calculateFreqTable for both streams -> (seq_counts, qual_counts, seq_ft, qual_ft)."""
    L = lib()
    sc = np.zeros((SEQ_MODELS, SEQ_ALPHA), dtype=np.uint32)
    qc = np.zeros((QUAL_MODELS, QUAL_ALPHA), dtype=np.uint32)
    rc = L.fqo_seq_counts(ptr(raw), ptr(recs), len(recs), ptr(sc))
    assert rc == 0, rc
    rc = L.fqo_qual_counts(ptr(raw), ptr(recs), len(recs), ptr(qc))
    assert rc == 0, rc
    sft = np.zeros(1, dtype=SEQ_FT_DTYPE)
    qft = np.zeros(1, dtype=QUAL_FT_DTYPE)
    assert L.fqo_seq_ft_from_counts(ptr(sc), ptr(sft)) == 0
    assert L.fqo_qual_ft_from_counts(ptr(qc), ptr(qft)) == 0
    return sc, qc, sft, qft


class OracleCtx:
    def __init__(self, sft, qft):
        self.sft, self.qft = sft, qft
        self.h = lib().fqo_ctx_create(ptr(sft), ptr(qft))
        assert self.h

    def close(self):
        if self.h:
            lib().fqo_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def encode(self, raw, recs, seq_cap=None, qual_cap=None):
        """-> dict(rc, seq, qual, readlens, n_count, n_pos, raw_after)"""
        L = lib()
        raw = np.array(raw, dtype=np.uint8, copy=True)
        n = len(recs)
        bases = int(recs["len"].sum())
        seq_cap = L.fqo_bound_seq(bases) if seq_cap is None else seq_cap
        qual_cap = L.fqo_bound_qual(bases) if qual_cap is None else qual_cap
        seq = np.zeros(seq_cap + 8, dtype=np.uint8)
        qual = np.zeros(qual_cap + 8, dtype=np.uint8)
        rl = np.zeros(n + 1, dtype=np.uint16)
        nc = np.zeros(n + 1, dtype=np.uint16)
        npos = np.zeros(bases + 1, dtype=np.uint16)
        sl, ql, nn = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        rc = L.fqo_encode_block(self.h, ptr(raw), ptr(recs), n, ptr(seq), seq_cap, C.byref(sl),
                                ptr(qual), qual_cap, C.byref(ql), ptr(rl), ptr(nc), ptr(npos),
                                C.byref(nn))
        return dict(rc=rc, seq=seq[: sl.value].copy(), qual=qual[: ql.value].copy(),
                    readlens=rl[:n].copy(), n_count=nc[:n].copy(), n_pos=npos[: nn.value].copy(),
                    raw_after=raw)

    def decode(self, seq, qual, n_count, n_pos, recs, raw_skeleton):
        """Fills seq/qual line bytes into a copy of raw_skeleton -> (rc, raw_out)."""
        L = lib()
        out = np.array(raw_skeleton, dtype=np.uint8, copy=True)
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        qual = np.ascontiguousarray(qual, dtype=np.uint8)
        n_count = np.ascontiguousarray(n_count, dtype=np.uint16)
        n_pos = np.ascontiguousarray(n_pos, dtype=np.uint16)
        rc = L.fqo_decode_block(self.h, ptr(seq), len(seq), ptr(qual), len(qual), ptr(n_count),
                                len(n_count), ptr(n_pos), len(n_pos), ptr(recs), len(recs), ptr(out))
        return rc, out


def blank_skeleton(raw, recs):
    """Copy of raw with every sequence and quality byte wiped ('?'): what
    decodeChunk's first pass leaves before the FSE pass (src/workspace.cpp:62-80)."""
    out = np.array(raw, dtype=np.uint8, copy=True)
    for off in ("seq_off", "qual_off"):
        starts = recs[off].astype(np.int64)
        lens = recs["len"].astype(np.int64)
        idx = np.repeat(starts - np.concatenate(([0], np.cumsum(lens)[:-1])), lens) + np.arange(lens.sum())
        out[idx] = ord("?")
    return out
