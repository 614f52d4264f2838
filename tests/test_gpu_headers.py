"""The header field coder on the device (fqcomp28_amd/csrc/headers.hip: fqgpu_encode_headers_begin / _wait / _end,
SURVEY.md 8(f) row 3) against oracle/headers_oracle.py -- the restatement of the reference's
CompressionWorkspace::encodeHeader (src/workspace.cpp:95-126, src/headers.cpp:76-120) that tests/test_headers.py pins
to the reference's known answers.  Byte work: bit-exact, every stream of every field."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import headers_oracle as HO  # noqa: E402

pytestmark = pytest.mark.gpu

E_HEADER = -8


@pytest.fixture(scope="module")
def F():
    import fqcomp28_amd as F
    assert F.device_count() >= 1, "no GPU visible: the product path has no CPU fallback"
    return F


@pytest.fixture(scope="module")
def ctx(F, golden_dir):
    raw, recs = O.load_fastq(os.path.join(golden_dir, "SRR065390_sub_1.fastq"))
    _, _, sft, qft = O.freq_tables(raw, recs)
    c = F.Context(sft, qft)
    yield c
    c.close()


_READS = []


def fastq_of(headers):
    """a chunk with these headers over the reads of the fixture the context's tables come from (the header coder does
    not look at the reads; the block's encode must stay inside its capacity rule)"""
    if not _READS:
        golden = os.path.join(ROOT, "tests", "golden")
        raw, recs = O.load_fastq(os.path.join(golden, "SRR065390_sub_1.fastq"))
        b = raw.tobytes()
        for r in recs:
            s, q, n = int(r["seq_off"]), int(r["qual_off"]), int(r["len"])
            _READS.append(b"\n" + b[s: s + n] + b"\n+\n" + b[q: q + n] + b"\n")
    return np.frombuffer(b"".join(h + _READS[i % len(_READS)] for i, h in enumerate(headers)), dtype=np.uint8)


def headers_of(raw, recs):
    b = raw.tobytes()
    out, line = [], 0
    for r in recs:
        out.append(b[line: int(r["seq_off"]) - 1])
        line = int(r["qual_off"]) + int(r["len"]) + 1
    return out


def code_on_gpu(ctx, raw, first_header, recs=None):
    types, seps = HO.format_from_header(first_header)
    fmt = ([0 if t == HO.NUMERIC else 1 for t in types], bytes(seps), first_header)
    return ctx.encode_raw(raw, recs=recs, header_format=fmt)


def assert_fields_equal(g, headers, first_header, lossless=True):
    assert g["rc"] == 0 and g["headers_rc"] == 0, (g["rc"], g.get("headers_rc"), g.get("bad_record"))
    types, _, streams = HO.encode_headers(headers, first_header)
    assert len(g["header_fields"]) == len(types)
    for i, ((flags, content, lengths), s) in enumerate(zip(g["header_fields"], streams)):
        assert flags.tobytes() == bytes(s.flags), (i, "flags")
        assert lengths.tobytes() == bytes(s.lengths), (i, "lengths")
        assert content.tobytes() == bytes(s.content), (i, "content")
    if lossless:  # (a numeric field written "007", "-0" or "12ab" comes back as 7, 0, 12 -- in the reference too)
        assert HO.decode_headers(len(headers), first_header, streams) == headers


@pytest.mark.parametrize("name", ["SRR065390_sub_1", "SRR065390_sub_2", "without_ns", "SRR065390_1_first5"])
def test_fixture_headers(ctx, golden_dir, name):
    """the reference's own test files: real SRA headers (string, read number, string, length=...)"""
    raw, recs = O.load_fastq(os.path.join(golden_dir, name + ".fastq"))
    hdrs = headers_of(raw, recs)
    assert_fields_equal(code_on_gpu(ctx, raw, hdrs[0]), hdrs, hdrs[0])
    # with the caller's record table instead of the device parser's
    assert_fields_equal(code_on_gpu(ctx, raw, hdrs[0], recs=recs), hdrs, hdrs[0])


def test_first_header_of_the_dataset_is_not_the_chunks_first(ctx, golden_dir):
    """Workspace::startNewChunk (src/workspace.cpp:90-93): a chunk's first header is coded against the DATASET's"""
    raw, recs = O.load_fastq(os.path.join(golden_dir, "SRR065390_sub_2.fastq"))
    hdrs = headers_of(raw, recs)
    first = b"@SRR065390.1 HWUSI-EAS687_61DAJ:1:1:1055:3384 length=100"
    assert HO.format_from_header(first) == HO.format_from_header(hdrs[0])
    assert_fields_equal(code_on_gpu(ctx, raw, first), hdrs, first)


def test_strings_that_change_numbers_that_wrap_fields_that_are_missing(ctx):
    """string values that repeat and change, lengths 0..254, numbers going down / negative / jumping by more than
    2^31 (the difference wraps), a number followed by other characters (from_chars reads the digits in front),
    headers with fewer separators than the format (the rest of the fields is empty or everything), a separator as
    the first character of a field; 3 000 records: twelve workgroups, the last one ragged"""
    rng = np.random.default_rng(11)
    names = [b"EAS687", b"EAS688", b"EAS688", b"TIOBDUREN", b"B", b"", b"x" * 254, b"y" * 100, b"-lead"]
    nums = [b"5", b"4", b"2147483647", b"-2147483648", b"0", b"33808546", b"-7", b"12ab", b"007", b"-0"]
    hdrs = [b"@EAS687.1 5 length=50/1"]
    for i in range(2999):
        a = names[int(rng.integers(len(names)))] if rng.random() < 0.3 else hdrs[-1][1:].split(b".")[0]
        b = nums[int(rng.integers(len(nums)))] if rng.random() < 0.5 else b"%d" % int(rng.integers(0, 2**31))
        c = b"length" if rng.random() < 0.9 else b"len"
        h = b"@%s.%d %s %s=%d/%d" % (a, i + 2, b, c, 50 + i % 251, 1 + i % 2)
        if i % 97 == 0:
            h = h[: int(rng.integers(2, len(h)))]   # cut: later fields are missing
            if not h[-1:].isdigit():                   # (a missing numeric field would be an error: next test)
                h = b"@q.1 2 z=3/4"
        hdrs.append(h)
    # cut headers whose numeric fields ended up empty cannot be coded: keep those the oracle codes
    ok = []
    for h in hdrs:
        try:
            HO.encode_headers([h], hdrs[0])
            ok.append(h)
        except ValueError:
            ok.append(b"@q.1 2 z=3/4")
    raw = fastq_of(ok)
    assert_fields_equal(code_on_gpu(ctx, raw, ok[0]), ok, ok[0], lossless=False)


def test_single_field_and_many_fields(ctx):
    hdrs = [b"@%d" % (1000 - 3 * i) for i in range(700)]
    assert_fields_equal(code_on_gpu(ctx, fastq_of(hdrs), hdrs[0]), hdrs, hdrs[0])
    hdrs = [b"@" + b":".join(b"%s%d" % (b"f" if k % 3 else b"", (i * (k + 1)) % 1000) for k in range(64)) for i in range(300)]
    types, seps = HO.format_from_header(hdrs[0])
    assert len(types) == 64
    assert_fields_equal(code_on_gpu(ctx, fastq_of(hdrs), hdrs[0]), hdrs, hdrs[0])


def test_headers_that_cannot_be_coded_are_reported_with_their_record(ctx, F):
    """the host coder throws where the reference asserts (src/headers.cpp:20, 83, 117); the device reports the FIRST
    such record"""
    good = [b"@r.%d x" % (i + 1) for i in range(1000)]
    for at, bad in ((0, b"@r.x1 x"), (517, b"@r. x"), (999, b"@r.99999999999 x"), (300, b"@r.2147483648 x"), (301, b"@r.-2147483649 x"),
                    (640, b"@r.7 " + b"z" * 255)):
        hdrs = list(good)
        hdrs[at] = bad
        if at < 900:
            hdrs[950] = b"@r.+5 x"   # a later one must not win
        with pytest.raises(ValueError):
            HO.encode_headers(hdrs, good[0])
        g = code_on_gpu(ctx, fastq_of(hdrs), good[0])
        assert g["rc"] == 0 and g["headers_rc"] == E_HEADER and g["bad_record"] == at, (at, g.get("headers_rc"), g.get("bad_record"))
    # 254 bytes is fine, and so is a field of 255 that never changes
    hdrs = [b"@r.%d %s" % (i + 1, b"z" * 254) for i in range(300)]
    assert_fields_equal(code_on_gpu(ctx, fastq_of(hdrs), hdrs[0]), hdrs, hdrs[0])
    hdrs = [b"@r.%d %s" % (i + 1, b"z" * 300) for i in range(300)]
    g = code_on_gpu(ctx, fastq_of(hdrs), hdrs[0])
    assert g["headers_rc"] == 0 and all(not f.any() for f in [g["header_fields"][2][0]])
    assert "header" in F.lib().fqgpu_strerror(E_HEADER).decode()
    # more fields than the library takes
    types = [1] * 65
    raw = fastq_of(good)
    assert ctx.encode_raw(raw, header_format=(types, b":" * 64, b"@" + b":".join([b"a"] * 65)))["rc"] == -4


def test_headers_of_a_whole_synthetic_block(ctx, F):
    """32 MiB of BASELINE's synthetic reads: ~110 K headers through all three kernels, and the encode of the block is
    what it is without the header coder"""
    raw, _ = F.synth_fastq(32 << 20, 2, seed=5)
    recs = F.parse_fastq(raw)
    hdrs = headers_of(raw, recs)
    ctx = F.Context(*F.freq_tables(raw, recs))   # (tables of these reads: the block must fit its capacity rule)
    g = code_on_gpu(ctx, raw, hdrs[0])
    assert_fields_equal(g, hdrs, hdrs[0])
    plain = ctx.encode_raw(raw)
    for k in ("seq", "qual", "readlens", "n_count", "n_pos"):
        assert np.array_equal(g[k], plain[k]), k
    ctx.close()


def test_a_record_table_that_is_not_the_chunks_never_faults(ctx, golden_dir):
    """the header coder takes header r from the end of record r - 1 to the sequence of record r: with a caller's table
    in another order, or with every record the same one, the "headers" overlap or are empty.  The fields are then
    nonsense -- coded, refused as headers or refused as a table -- but every access stays inside the block and the
    handle works afterwards."""
    raw, recs = O.load_fastq(os.path.join(golden_dir, "SRR065390_sub_1.fastq"))
    hdrs = headers_of(raw, recs)
    rng = np.random.default_rng(3)
    same = recs.copy()
    same[:] = recs[len(recs) // 2]
    for table in (recs[::-1].copy(), recs[rng.permutation(len(recs))], same, recs[:1].repeat(3000)):
        g = code_on_gpu(ctx, raw, hdrs[0], recs=table)
        assert g["rc"] in (0, -1) and g.get("headers_rc", 0) in (0, -4, -8), (g["rc"], g.get("headers_rc"))
    assert_fields_equal(code_on_gpu(ctx, raw, hdrs[0], recs=recs), hdrs, hdrs[0])
