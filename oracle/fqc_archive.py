"""Independent restatement of the reference's `.fqc` container, in Python.

TEST INFRASTRUCTURE ONLY (tests/, bench.py's checker legs and tools that PREPARE inputs may import
it; the product -- fqcomp28_amd/csrc/archive.hpp -- never does).  It exists so that
  * the C++ Archive (the product's container code) is checked against a second reading of
    src/archive.h:10-17 and src/archive.cpp:45-55, 57-106, 108-163 -- field order, sizes, index --
    instead of against itself, and
  * BASELINE configs[4] ("decode a reference-produced archive") runs on a real `.fqc`-shaped file
    whose seq/qual streams come from the CPU oracle (the reference binary cannot be built here).

Layout (little-endian):
  u32 n_blocks                          src/archive.cpp:45-50 (written last, at offset 0)
  u16 hlen | first header | ft_seq (3076 B) | ft_qual (1081348 B)      src/prepare.cpp:12-21
  blocks in completion order            src/archive.cpp:57-106:
      u32 total | u32 n_records
      (u32 orig | u32 csize | bytes) x {readlens, n_count, n_pos}
      (u32 csize | bytes) x {seq, qual}
      per header field: STRING 3 x (u32 orig | u32 csize | bytes) [flags, content, lengths]
                        NUMERIC 1 x (u32 orig | u32 csize | bytes) [content]
  index: n_blocks x {i64 offset, u32 idx, 4 pad bytes}                 src/archive.h:20-27

The misc streams' compressed bytes are whatever `compress` produces (the reference: libbsc, out of
parity scope; the tests pass the product's fqgpu_memcompress through ctypes, or `None` = stored in
the product coder's "stored" mode: byte 0 then the data).
"""
import struct

import headers_oracle as HO

SEQ_FT_BYTES, QUAL_FT_BYTES = 3076, 1081348
BLOCKINFO = struct.Struct("<qI4x")  # sizeof(BlockInfo) == 16


def stored(data: bytes) -> bytes:
    """the product coder's stored mode (fq_misc.cpp): empty stays empty, else 0x00 + data"""
    return b"" if len(data) == 0 else b"\x00" + bytes(data)


def unstored(cdata: bytes, orig: int) -> bytes:
    if len(cdata) == 0:
        return b""
    assert cdata[0] == 0 and len(cdata) == orig + 1, "not a stored misc stream"
    return bytes(cdata[1:])


class Block:
    """One block as the container sees it: original sizes + compressed byte strings."""

    def __init__(self):
        self.idx = 0
        self.total = self.n_records = 0
        self.readlens = self.n_count = self.n_pos = (0, b"")  # (orig size, compressed bytes)
        self.seq = self.qual = b""
        self.fields = []  # per header field: list of (orig, cbytes): 3 for STRING, 1 for NUMERIC


def block_from_streams(idx, raw, recs, enc, first_header, compress=stored, n_count_prefix=b"", n_pos_prefix=b""):
    """Block of chunk `idx` from the oracle's encode result `enc` (dict with seq, qual, readlens,
    n_count, n_pos) and the chunk's headers, coded against the dataset's first header like
    CompressionWorkspace::encodeChunk does (src/workspace.cpp:14-45, 90-93)."""
    b = Block()
    b.idx = idx
    b.total, b.n_records = len(raw), len(recs)
    rl = enc["readlens"].astype("<u2").tobytes()
    nc = n_count_prefix + enc["n_count"].astype("<u2").tobytes()
    npos = n_pos_prefix + enc["n_pos"].astype("<u2").tobytes()
    b.readlens = (len(rl), compress(rl))
    b.n_count = (len(nc), compress(nc))
    b.n_pos = (len(npos), compress(npos))
    b.seq, b.qual = enc["seq"].tobytes(), enc["qual"].tobytes()
    headers = headers_of(raw, recs)
    types, _, streams = HO.encode_headers(headers, first_header)
    for t, s in zip(types, streams):
        if t == HO.STRING:
            b.fields.append([(len(s.flags), compress(bytes(s.flags))), (len(s.content), compress(bytes(s.content))),
                             (len(s.lengths), compress(bytes(s.lengths)))])
        else:
            b.fields.append([(len(s.content), compress(bytes(s.content)))])
    return b


def headers_of(raw, recs):
    """header lines (with '@', without newline) of a parsed chunk"""
    out, prev_end = [], 0
    rb = raw.tobytes() if hasattr(raw, "tobytes") else bytes(raw)
    for r in recs:
        out.append(rb[prev_end: int(r["seq_off"]) - 1])
        prev_end = int(r["qual_off"]) + int(r["len"]) + 1
    return out


def write_archive(path, first_header: bytes, seq_ft: bytes, qual_ft: bytes, blocks):
    """blocks are written in the order given (= completion order); the index keeps their idx"""
    assert len(seq_ft) == SEQ_FT_BYTES and len(qual_ft) == QUAL_FT_BYTES
    index = []
    with open(path, "wb") as f:
        f.write(struct.pack("<I", 0))  # patched by the index step below
        f.write(struct.pack("<H", len(first_header)) + first_header + seq_ft + qual_ft)
        for b in blocks:
            index.append((f.tell(), b.idx))
            f.write(struct.pack("<II", b.total, b.n_records))
            for orig, c in (b.readlens, b.n_count, b.n_pos):
                f.write(struct.pack("<II", orig, len(c)) + c)
            for c in (b.seq, b.qual):
                f.write(struct.pack("<I", len(c)) + c)
            for parts in b.fields:
                for orig, c in parts:
                    f.write(struct.pack("<II", orig, len(c)) + c)
        for off, idx in index:
            f.write(BLOCKINFO.pack(off, idx))
        f.seek(0)
        f.write(struct.pack("<I", len(index)))


def read_archive(path):
    """-> (first_header, seq_ft, qual_ft, blocks sorted by idx, index entries in FILE order)"""
    with open(path, "rb") as f:
        data = f.read()
    (n_blocks,) = struct.unpack_from("<I", data, 0)
    (hlen,) = struct.unpack_from("<H", data, 4)
    at = 6
    first_header = data[at: at + hlen]; at += hlen
    seq_ft = data[at: at + SEQ_FT_BYTES]; at += SEQ_FT_BYTES
    qual_ft = data[at: at + QUAL_FT_BYTES]; at += QUAL_FT_BYTES
    data_start = at
    idx_at = len(data) - n_blocks * BLOCKINFO.size
    entries = [BLOCKINFO.unpack_from(data, idx_at + i * BLOCKINFO.size) for i in range(n_blocks)]
    types, _ = HO.format_from_header(first_header)
    blocks = []
    ends = []
    for off, idx in entries:
        p = off
        b = Block()
        b.idx = idx
        b.total, b.n_records = struct.unpack_from("<II", data, p); p += 8
        triple = []
        for _ in range(3):
            orig, cs = struct.unpack_from("<II", data, p); p += 8
            triple.append((orig, data[p: p + cs])); p += cs
        b.readlens, b.n_count, b.n_pos = triple
        for name in ("seq", "qual"):
            (cs,) = struct.unpack_from("<I", data, p); p += 4
            setattr(b, name, data[p: p + cs]); p += cs
        for t in types:
            parts = []
            for _ in range(3 if t == HO.STRING else 1):
                orig, cs = struct.unpack_from("<II", data, p); p += 8
                parts.append((orig, data[p: p + cs])); p += cs
            b.fields.append(parts)
        ends.append(p)
        blocks.append(b)
    # the blocks tile the space between the meta section and the index exactly
    spans = sorted(zip([e[0] for e in entries], ends))
    pos = data_start
    for s, e in spans:
        assert s == pos, "gap or overlap between blocks"
        pos = e
    assert pos == idx_at, "bytes between the last block and the index"
    return first_header, seq_ft, qual_ft, sorted(blocks, key=lambda b: b.idx), entries
