"""The `.fqc` block container and the misc-stream coder, host side, CPU only.

The product's container code (fqcomp28_amd/csrc/archive.hpp, driven through tests/cpp/archive_tool.cpp)
is checked against oracle/fqc_archive.py -- an independent Python reading of the reference's layout
(src/archive.h:10-17: block count, metadata, blocks, index; src/archive.cpp:57-106: field order;
src/archive.h:20-27: 16-byte index entries sorted by chunk index on load, :85-89) -- on archives
whose seq/qual streams are the CPU oracle's.  The misc streams' compressed bytes are the library's
own coder's (libbsc is out of parity scope); the coder itself is tested for round trip, bound
(src_size + 28, src/workspace.h:18), empty input and refusal of damaged input.
"""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fqc_archive as A  # noqa: E402
import oracle_lib as O  # noqa: E402

FIXTURES = ["SRR065390_sub_1", "without_ns", "SRR065390_sub_2", "SRR065390_1_first5"]


@pytest.fixture(scope="module")
def F():
    import fqcomp28_amd as F
    F.lib()
    return F


@pytest.fixture(scope="module")
def tool(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("arc") / "archive_tool")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "cpp", "archive_tool.cpp"),
                    "-L" + os.path.join(ROOT, "fqcomp28_amd"), "-lfqgpu", "-Wl,-rpath," + os.path.join(ROOT, "fqcomp28_amd"),
                    "-lpthread"], check=True)
    return exe


def split_blocks(raw, recs, n):
    """n blocks of whole records -> [(raw, recs)] with block-relative offsets"""
    cuts = np.linspace(0, len(recs), n + 1).astype(int)
    out = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        lo = 0 if a == 0 else int(recs[a - 1]["qual_off"] + recs[a - 1]["len"] + 1)
        hi = int(recs[b - 1]["qual_off"] + recs[b - 1]["len"] + 1)
        r = recs[a:b].copy()
        r["seq_off"] -= lo
        r["qual_off"] -= lo
        out.append((raw[lo:hi], r))
    return out


def oracle_archive(F, path, raw, recs, n_blocks, order=None, compress=None):
    """archive of `raw` in n_blocks blocks, coded by the CPU oracle, written by the Python writer"""
    _, _, sft, qft = O.freq_tables(raw, recs)
    octx = O.OracleCtx(sft, qft)
    parts = split_blocks(raw, recs, n_blocks)
    first_header = A.headers_of(raw, recs[:1])[0]
    comp = (lambda d: F.memcompress(np.frombuffer(d, dtype=np.uint8)).tobytes()) if compress is None else compress
    blocks, encs = [], []
    for i, (braw, brecs) in enumerate(parts):
        e = octx.encode(braw, brecs)
        assert e["rc"] == 0
        encs.append(e)
        blocks.append(A.block_from_streams(i, braw, brecs, e, first_header, compress=comp))
    order = list(range(n_blocks)) if order is None else order
    A.write_archive(path, first_header, sft.tobytes(), qft.tobytes(), [blocks[i] for i in order])
    return parts, encs, blocks, (sft, qft)


# ---------------------------------------------------------------- misc-stream coder
def test_memcompress_roundtrip_bound_and_edge_cases(F):
    rng = np.random.default_rng(7)
    cases = [np.zeros(0, np.uint8), np.arange(5, dtype=np.uint8), np.zeros(16, np.uint8), rng.integers(0, 256, 17, dtype=np.uint8),
             np.full(100000, 150, np.uint16).view(np.uint8),          # readlens of fixed-length reads
             np.ones(60000, np.int32).view(np.uint8),                 # "+1" read-number deltas
             rng.integers(0, 256, 50000, dtype=np.uint8),             # incompressible -> stored
             rng.integers(0, 3, 50001).astype(np.uint8),              # n_count-like
             rng.integers(0, 300, 1 << 18).astype(np.uint16).view(np.uint8),
             np.frombuffer(b"HWUSI-EAS687_61DAJ" * 2000, dtype=np.uint8)]
    for c in cases:
        z = F.memcompress(c)
        assert z.size <= c.size + 28                                   # extra_csize_misc, src/workspace.h:18
        assert (z.size == 0) == (c.size == 0)                          # src/memcompress.cpp:56-57
        assert np.array_equal(F.memdecompress(z, c.size), c)
    assert F.memcompress(cases[4]).size < 64 and F.memcompress(cases[5]).size < 64
    assert F.memcompress(cases[6]).size == cases[6].size + 1
    assert F.memcompress(cases[8]).size < 0.75 * cases[8].size


def test_memdecompress_refuses_damaged_streams(F):
    src = np.random.default_rng(3).integers(0, 40, 40000).astype(np.uint16).view(np.uint8)
    z = F.memcompress(src)
    refused = 0
    for k in list(range(0, 40)) + list(range(40, z.size, 97)):
        bad = z.copy()
        bad[k] ^= 0x55
        try:
            out = F.memdecompress(bad, src.size)
            assert out.size == src.size  # accepted: then it is at least a full-size output
        except F.FqgpuError:
            refused += 1
    assert refused > 20
    for cut in (1, 5, z.size // 2, z.size - 1):
        with pytest.raises(F.FqgpuError):
            F.memdecompress(z[:cut], src.size)
    with pytest.raises(F.FqgpuError):
        F.memdecompress(z, src.size + 1)


# ---------------------------------------------------------------- container
@pytest.mark.parametrize("name", FIXTURES)
def test_cpp_archive_reads_and_rewrites_the_python_archive_byte_for_byte(F, tool, tmp_path, golden_dir, name):
    raw, recs = O.load_fastq(os.path.join(golden_dir, name + ".fastq"))
    n_blocks = 1 if len(recs) < 10 else 4
    src = str(tmp_path / "py.fqc")
    parts, encs, blocks, _ = oracle_archive(F, src, raw, recs, n_blocks)
    dst = str(tmp_path / "cpp.fqc")
    r = subprocess.run([tool, "copy", src, dst], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert ("copied %d blocks, index %d bytes" % (n_blocks, 4 + 16 * n_blocks)) in r.stdout  # indexBytes(), src/archive.h:57-59
    a, b = open(src, "rb").read(), open(dst, "rb").read()
    assert a == b, "readBlock -> writeBlock -> writeIndex does not reproduce the file"
    # header of the file, by hand: block count first, u16 header length, the header, the two PODs
    first_header = A.headers_of(raw, recs[:1])[0]
    assert struct.unpack_from("<I", b, 0)[0] == n_blocks
    assert struct.unpack_from("<H", b, 4)[0] == len(first_header) and b[6: 6 + len(first_header)] == first_header
    meta_end = 6 + len(first_header) + 3076 + 1081348
    # index: last 16 * n bytes, offsets ascending from the end of the metadata, padding zero
    ents = [struct.unpack_from("<qI4s", b, len(b) - 16 * (n_blocks - i)) for i in range(n_blocks)]
    assert ents[0][0] == meta_end and [e[1] for e in ents] == list(range(n_blocks)) and all(e[2] == b"\0\0\0\0" for e in ents)


def test_completion_order_and_sorted_index(F, tool, tmp_path, golden_dir):
    """Blocks land in the file in completion order; readers see them in input order
    (sortIndex, src/archive.h:85-89).  Every stream survives the trip."""
    raw, recs = O.load_fastq(os.path.join(golden_dir, "SRR065390_sub_1.fastq"))
    order = [3, 0, 4, 1, 2]
    src = str(tmp_path / "shuffled.fqc")
    parts, encs, blocks, (sft, qft) = oracle_archive(F, src, raw, recs, 5, order=order)
    r = subprocess.run([tool, "dump", src], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr
    lines = r.stdout.splitlines()
    assert lines[1] == "fields 11" and lines[2] == "blocks 5"
    idx_lines = [ln.split() for ln in lines if ln.startswith("index ")]
    assert [int(x[2]) for x in idx_lines] == [0, 1, 2, 3, 4]                 # sorted on load
    offs = [int(x[1]) for x in idx_lines]
    assert sorted(offs) == [offs[i] for i in order]                          # file order = completion order
    seen = [ln for ln in lines if ln.startswith("block ")]
    assert [int(s.split()[1]) for s in seen] == [0, 1, 2, 3, 4]
    # per block: sizes, the misc streams after decompressMiscBuffers, every header
    it = iter(lines[lines.index(seen[0]):])
    for i, ((braw, brecs), e) in enumerate(zip(parts, encs)):
        head = next(it).split()
        assert head[:8] == ["block", str(i), "total", str(braw.size), "n_records", str(len(brecs)), "seq", str(len(e["seq"]))]
        assert head[8:] == ["qual", str(len(e["qual"]))]
        for name in ("readlens", "n_count", "n_pos"):
            p = next(it).split()
            want = e[name].astype("<u2").tobytes()
            assert p[0] == name and int(p[1]) == len(want) and (bytes.fromhex(p[2]) if len(p) > 2 else b"") == want
        for h in A.headers_of(braw, brecs):
            assert next(it) == "h " + h.decode()
    # the C++ copy keeps every block; Python reads it back
    dst = str(tmp_path / "copy.fqc")
    assert subprocess.run([tool, "copy", src, dst], capture_output=True).returncode == 0
    fh, s2, q2, blocks2, entries = A.read_archive(dst)
    assert s2 == sft.tobytes() and q2 == qft.tobytes() and [e[1] for e in entries] == [0, 1, 2, 3, 4]
    for b0, b1 in zip(blocks, blocks2):
        assert (b0.total, b0.n_records, b0.readlens, b0.n_count, b0.n_pos, b0.seq, b0.qual, b0.fields) == \
               (b1.total, b1.n_records, b1.readlens, b1.n_count, b1.n_pos, b1.seq, b1.qual, b1.fields)


def test_truncated_archive_is_refused(F, tool, tmp_path, golden_dir):
    raw, recs = O.load_fastq(os.path.join(golden_dir, "without_ns.fastq"))
    src = str(tmp_path / "a.fqc")
    oracle_archive(F, src, raw, recs, 2)
    data = open(src, "rb").read()
    cut = str(tmp_path / "cut.fqc")
    open(cut, "wb").write(data[: len(data) // 3])
    r = subprocess.run([tool, "dump", cut], capture_output=True, text=True)
    assert r.returncode == 1 and "exception" in r.stdout


# ---------------------------------------------------------------- the FASTQ file ends of the farm
def reference_chunks(raw, recs, reading_size):
    """the chunks FastqReader::readNextChunk cuts (src/fastq_io.cpp:23-65): whole records of the next
    reading_size bytes, the next chunk begins where this one ended"""
    ends = (recs["qual_off"].astype(np.int64) + recs["len"] + 1)
    out, begin = [], 0
    while begin < ends[-1]:
        k = int(np.searchsorted(ends, begin + reading_size, side="right")) - 1
        if k < 0 or ends[k] <= begin:
            raise ValueError("reading size smaller than one record")
        out.append(int(ends[k]) - begin)
        begin = int(ends[k])
    return out


@pytest.mark.parametrize("name", FIXTURES)
def test_reader_cuts_the_reference_chunks_without_parsing(tool, tmp_path, golden_dir, name):
    """The reader finds the end of a chunk by a backwards search for the last complete record (no
    parse under its lock): same chunks as the reference's parse-everything reader for every reading
    size, incl. files whose quality lines begin with '@' or '+' (these do), a file without a final
    newline (its partial record is dropped) and reading sizes just above one record."""
    path = os.path.join(golden_dir, name + ".fastq")
    raw, recs = O.load_fastq(path)
    biggest = int((np.diff(np.concatenate([[0], recs["qual_off"].astype(np.int64) + recs["len"] + 1]))).max())
    sizes = sorted({biggest, biggest + 1, 1000, 1777, 4096, 65536, raw.size // 3, raw.size - 1, raw.size, raw.size + 10})
    for size in sizes:
        if size < biggest:
            continue
        r = subprocess.run([tool, "chunks", path, str(size)], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout
        got = [int(ln.split()[2]) for ln in r.stdout.splitlines() if ln.startswith("chunk ")]
        assert got == reference_chunks(raw, recs, size), size
    # reader -> writer identity with the chunks written last first (reference test/fastq_io_test.cpp:15-53)
    back = str(tmp_path / "back.fastq")
    assert subprocess.run([tool, "rejoin", path, back, "5000"], capture_output=True).returncode == 0
    assert open(back, "rb").read() == raw.tobytes() and not os.path.exists(back + ".part")
    # a writer that is abandoned before its flush() (a failed restore) leaves no file under either name
    gone = str(tmp_path / "gone.fastq")
    assert subprocess.run([tool, "rejoin", path, gone, "5000", "abandon"], capture_output=True).returncode == 3
    assert not os.path.exists(gone) and not os.path.exists(gone + ".part")
    # a reading size below one record is an error, not a hang or an empty archive
    r = subprocess.run([tool, "chunks", path, str(biggest // 2)], capture_output=True, text=True)
    assert r.returncode == 1 and "smaller than one record" in r.stdout
    # no newline at the end of the file: the unfinished record is dropped
    cut = str(tmp_path / "cut.fastq")
    open(cut, "wb").write(raw.tobytes()[:-1])
    r = subprocess.run([tool, "chunks", cut, str(raw.size)], capture_output=True, text=True)
    last_full = int(recs[-2]["qual_off"] + recs[-2]["len"] + 1)
    assert r.returncode == 0 and [int(ln.split()[2]) for ln in r.stdout.splitlines() if ln.startswith("chunk ")] == [last_full]


def test_quality_lines_that_look_like_headers_do_not_move_a_chunk_end(tool, tmp_path):
    """'@' and '+' are quality characters: a record is recognised by its shape over four lines."""
    recs = []
    rng = np.random.default_rng(4)
    for i in range(400):
        L = int(rng.integers(3, 60))
        seq = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5, L)].tobytes()
        q = bytearray(rng.integers(33, 75, L).astype(np.uint8).tobytes())
        q[0] = ord("@") if i % 3 == 0 else ord("+") if i % 3 == 1 else q[0]
        recs.append(b"@r%d +x\n" % i + seq + b"\n+\n" + bytes(q) + b"\n")
    data = b"".join(recs)
    path = str(tmp_path / "tricky.fastq")
    open(path, "wb").write(data)
    raw = np.frombuffer(data, dtype=np.uint8)
    table = O.parse_fastq(raw)
    for size in (200, 333, 1000, 4097):
        r = subprocess.run([tool, "chunks", path, str(size)], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout
        assert [int(ln.split()[2]) for ln in r.stdout.splitlines() if ln.startswith("chunk ")] == reference_chunks(raw, table, size)


# ---------------------------------------------------------------- the same host code under sanitizers
@pytest.fixture(scope="module")
def tool_sanitized(tmp_path_factory):
    """archive_tool with AddressSanitizer and UBSan over the header-only host code it instantiates (archive.hpp,
    workspace.hpp, headers.hpp); libfqgpu.so itself is loaded as it is.  (GPU sanitizers do not exist on this pool.)"""
    exe = str(tmp_path_factory.mktemp("arc_san") / "archive_tool_san")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", exe,
                        os.path.join(ROOT, "tests", "cpp", "archive_tool.cpp"), "-L" + os.path.join(ROOT, "fqcomp28_amd"), "-lfqgpu",
                        "-Wl,-rpath," + os.path.join(ROOT, "fqcomp28_amd"), "-lpthread"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no sanitizer runtime for g++ here: " + r.stderr[-200:])
    return exe


def test_container_reader_and_writer_are_clean_under_asan_and_ubsan(F, tool, tool_sanitized, tmp_path, golden_dir):
    """Every path of the container code the other tests walk -- copy (readBlock / writeBlock / writeIndex), dump (misc
    streams and header decoding), the reader's backwards boundary search at awkward reading sizes, the writer fed
    last chunk first, a truncated archive -- once more with the sanitizers on: same output, no report."""
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")

    def run(exe, *args):
        r = subprocess.run([exe, *args], capture_output=True, text=True, env=env)
        assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-2000:]
        return r

    for name in FIXTURES:
        path = os.path.join(golden_dir, name + ".fastq")
        raw, recs = O.load_fastq(path)
        n_blocks = 1 if len(recs) < 10 else 4
        src = str(tmp_path / (name + ".fqc"))
        oracle_archive(F, src, raw, recs, n_blocks, order=list(reversed(range(n_blocks))))
        dst = str(tmp_path / (name + ".copy.fqc"))
        r = run(tool_sanitized, "copy", src, dst)
        assert r.returncode == 0, r.stdout + r.stderr
        assert run(tool_sanitized, "dump", src).stdout == run(tool, "dump", src).stdout  # (blocks in completion order, index unsorted)
        assert run(tool_sanitized, "dump", dst).stdout == run(tool, "dump", dst).stdout
        biggest = int((np.diff(np.concatenate([[0], recs["qual_off"].astype(np.int64) + recs["len"] + 1]))).max())
        for size in (biggest, biggest + 1, 1777, raw.size - 1, raw.size + 10):
            if size < biggest:
                continue
            assert run(tool_sanitized, "chunks", path, str(size)).stdout == run(tool, "chunks", path, str(size)).stdout
        back = str(tmp_path / (name + ".back.fastq"))
        assert run(tool_sanitized, "rejoin", path, back, str(biggest + 7)).returncode == 0
        assert open(back, "rb").read() == raw.tobytes()
        # a reading size below one record and a truncated archive: refused, cleanly
        assert run(tool_sanitized, "chunks", path, str(biggest // 2)).returncode == 1
        whole = open(src, "rb").read()
        cut = str(tmp_path / (name + ".cut.fqc"))
        open(cut, "wb").write(whole[: len(whole) // 2])
        assert run(tool_sanitized, "dump", cut).returncode == 1


def test_what_the_farms_workers_share_is_clean_under_thread_sanitizer(F, tmp_path, golden_dir):
    """Archive::readBlock / writeBlock, DecodeIndexFile::put / get and the read Gate from six threads at once under
    ThreadSanitizer (the farm itself needs a GPU; these are the objects its workers share): every block comes back
    with its own index entry, no data race reported."""
    exe = str(tmp_path / "archive_tool_tsan")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-o", exe, os.path.join(ROOT, "tests", "cpp", "archive_tool.cpp"),
                        "-L" + os.path.join(ROOT, "fqcomp28_amd"), "-lfqgpu", "-Wl,-rpath," + os.path.join(ROOT, "fqcomp28_amd"), "-lpthread"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no ThreadSanitizer runtime for g++ here: " + r.stderr[-200:])
    raw, recs = O.load_fastq(os.path.join(golden_dir, "SRR065390_sub_1.fastq"))
    src = str(tmp_path / "in.fqc")
    oracle_archive(F, src, raw, recs, 6, order=[3, 0, 5, 1, 4, 2])
    r = subprocess.run([exe, "threads", src, str(tmp_path / "out.fqc"), "6", "12"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0"))
    assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr, r.stdout + r.stderr[-3000:]
    assert "72 blocks written, 72 read back" in r.stdout


def test_decode_index_file_damaged_truncated_foreign_under_asan_and_ubsan(F, tool_sanitized, tmp_path, golden_dir):
    """archive.hpp's DecodeIndexFile (the farm's `<archive>.fqx`) with the sanitizers on: written by six threads, read back
    whole; then every kind of damage -- a bit in an entry's bytes, in an entry's head, in the table, in the trailer, the file
    cut at many lengths, random bytes -- is refused with an exception (exit 1), never a sanitizer report; a file that belongs
    to another archive says so."""
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    raw, recs = O.load_fastq(os.path.join(golden_dir, "SRR065390_sub_1.fastq"))
    src, arc = str(tmp_path / "in.fqc"), str(tmp_path / "out.fqc")
    oracle_archive(F, src, raw, recs, 5, order=[2, 0, 4, 1, 3])

    def run(*args):
        r = subprocess.run([tool_sanitized, *args], capture_output=True, text=True, env=env, timeout=120)
        assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-2000:]
        return r
    assert run("threads", src, arc, "6", "3").returncode == 0
    r = run("sidecheck", arc)
    assert r.returncode == 0 and r.stdout.startswith("entries 15 missing 0"), r.stdout
    side = arc + ".fqx"
    good = open(side, "rb").read()
    rng = np.random.default_rng(5)
    places = [10, 40, 100, len(good) // 2, len(good) - 150, len(good) - 30, len(good) - 20, len(good) - 10, len(good) - 2]
    refused = 0
    for at in places + [int(x) for x in rng.integers(4, len(good), 40)]:
        bad = bytearray(good)
        bad[at] ^= 1 << int(rng.integers(8))
        open(side, "wb").write(bad)
        r = run("sidecheck", arc)
        assert r.returncode in (0, 1), (at, r.stdout, r.stderr[-500:])
        if r.returncode == 1:
            assert "decode index file" in r.stdout, (at, r.stdout)
            refused += 1
        else:  # (a bit of the identity words: the file then looks like another archive's; padding bytes of an entry head)
            assert r.stdout.startswith("foreign") or r.stdout.startswith("entries 15 missing 0"), (at, r.stdout)
    assert refused >= 40
    for cut in [0, 3, 4, 20, 31, 32, len(good) // 3, len(good) - 29, len(good) - 28, len(good) - 1]:
        open(side, "wb").write(good[:cut])
        r = run("sidecheck", arc)
        assert r.returncode == 1 and "decode index file" in r.stdout, (cut, r.stdout, r.stderr[-300:])
    for _ in range(10):
        open(side, "wb").write(good[:4] + bytes(rng.integers(0, 256, len(good) - 8, dtype=np.uint8)) + good[-4:])
        assert run("sidecheck", arc).returncode in (0, 1)
    # the index file of ANOTHER archive under this one's name
    other = str(tmp_path / "other.fqc")
    assert run("threads", src, other, "2", "1").returncode == 0
    os.replace(other + ".fqx", side)
    assert run("sidecheck", arc).stdout.startswith("foreign")


def test_misc_coder_decodes_or_refuses_damaged_streams_under_asan_and_ubsan(tmp_path):
    """tests/cpp/misc_fuzz.cpp: fq_misc.cpp compiled with the sanitizers; round trips of the container's stream
    shapes, then thousands of truncated / bit-flipped / random streams into exact-size buffers."""
    exe = str(tmp_path / "misc_fuzz")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        os.path.join(ROOT, "tests", "cpp", "misc_fuzz.cpp"), os.path.join(ROOT, "fqcomp28_amd", "csrc", "fq_misc.cpp"),
                        "-I" + os.path.join(ROOT, "include"), "-o", exe], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no sanitizer runtime for g++ here: " + r.stderr[-200:])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stdout + r.stderr[-2000:]
    assert "no report" in r.stdout
