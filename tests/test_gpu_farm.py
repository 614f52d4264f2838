"""The block farm around the hot path, on a real GPU (north_star:
"independent FASTQ blocks farmed ... embarrassingly parallel, no collectives"):

* the C++ pipeline (fqcomp28_amd/csrc/process.hpp behind tools/fqc_tool.cpp: the reference's
  processReads / processArchiveParts, src/process.cpp:32-105) with three worker threads on device 0:
  every block of the archive it writes carries exactly the oracle's streams, the archive reads back
  through the independent Python container reader, and decompression restores the input file;
* decode of `.fqc` files written by the ORACLE-side writer (BASELINE configs[4] in small: the
  reference binary cannot be built, so the CPU oracle plays its part), completion order shuffled;
* two FRESH processes (one per rank, gloo, both on device 0 of this box) code their b mod 2 share
  of one job with broadcast tables: the union equals the oracle's streams.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fqc_archive as A  # noqa: E402
import headers_oracle as HO  # noqa: E402
import oracle_lib as O  # noqa: E402

pytestmark = pytest.mark.gpu
FIXTURES = ["SRR065390_sub_1", "without_ns", "SRR065390_sub_2", "SRR065390_1_first5"]


@pytest.fixture(scope="module")
def F():
    import fqcomp28_amd as F
    if F.device_count() < 1:
        pytest.fail("no GPU visible: the product path has no CPU fallback")
    return F


@pytest.fixture(scope="module")
def tool(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("farm") / "fqc_tool")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-o", exe, os.path.join(ROOT, "tools", "fqc_tool.cpp"),
                    "-L" + os.path.join(ROOT, "fqcomp28_amd"), "-lfqgpu", "-Wl,-rpath," + os.path.join(ROOT, "fqcomp28_amd"),
                    "-lpthread"], check=True)
    return exe


def run_tool(tool, *args):
    r = subprocess.run([tool] + [str(a) for a in args], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return json.loads(r.stdout.strip().splitlines()[-1])


def first_chunk(raw, size):
    """what FastqReader::readNextChunk(size) returns first (src/fastq_io.cpp:23-65): the whole records of raw[:size]"""
    import fqcomp28_amd
    recs = fqcomp28_amd.parse_fastq(raw[:size])  # host parser: drops the partial record at the end
    end = int(recs[-1]["qual_off"] + recs[-1]["len"] + 1)
    assert np.array_equal(recs, O.parse_fastq(raw[:end]))
    return raw[:end], recs


def check_archive_against_oracle(F, path, raw, sample_bytes, n_workers=None):
    """every field of every block of an archive written by the GPU farm"""
    first_header, sft_b, qft_b, blocks, entries = A.read_archive(path)
    sft = np.frombuffer(sft_b, dtype=O.SEQ_FT_DTYPE).copy()
    qft = np.frombuffer(qft_b, dtype=O.QUAL_FT_DTYPE).copy()
    # dataset analysis = oracle's calculateFreqTable on the sample chunk
    sraw, srecs = first_chunk(raw, sample_bytes)
    _, _, osft, oqft = O.freq_tables(sraw, srecs)
    assert sft.tobytes() == osft.tobytes() and qft.tobytes() == oqft.tobytes()
    assert first_header == A.headers_of(sraw, srecs[:1])[0]
    octx = O.OracleCtx(sft, qft)
    types, _ = HO.format_from_header(first_header)
    at = 0
    for i, b in enumerate(blocks):  # sorted by idx = input order: the blocks tile the input file
        assert b.idx == i
        braw = raw[at: at + b.total]
        brecs = O.parse_fastq(braw)
        assert len(brecs) == b.n_records
        e = octx.encode(braw, brecs)
        assert e["rc"] == 0
        assert b.seq == e["seq"].tobytes(), "seq stream of block %d" % i
        assert b.qual == e["qual"].tobytes(), "qual stream of block %d" % i
        for (orig, c), want in ((b.readlens, e["readlens"]), (b.n_count, e["n_count"]), (b.n_pos, e["n_pos"])):
            got = F.memdecompress(np.frombuffer(c, dtype=np.uint8), orig)
            assert got.tobytes() == want.astype("<u2").tobytes()
        _, _, streams = HO.encode_headers(A.headers_of(braw, brecs), first_header)
        for t, parts, s in zip(types, b.fields, streams):
            wants = [bytes(s.flags), bytes(s.content), bytes(s.lengths)] if t == HO.STRING else [bytes(s.content)]
            for (orig, c), want in zip(parts, wants):
                assert orig == len(want) and F.memdecompress(np.frombuffer(c, dtype=np.uint8), orig).tobytes() == want
        at += b.total
    assert at == raw.size
    return blocks, entries


def test_cpp_farm_three_workers_every_block_matches_oracle(F, tool, tmp_path):
    raw, _ = F.synth_fastq(26 << 20, 4, seed=11)       # mixed lengths, N bases: every side stream busy
    src = tmp_path / "in.fastq"
    raw.tofile(src)
    arc = tmp_path / "out.fqc"
    rep = run_tool(tool, "c", src, arc, "-t", 3, "-R", 3, "-S", 5)
    assert rep["raw_bytes"] == raw.size and rep["blocks"] >= 8 and sum(rep["blocks_per_worker"]) == rep["blocks"]
    assert len(rep["blocks_per_worker"]) == 3
    blocks, entries = check_archive_against_oracle(F, str(arc), raw, 5 << 20)
    assert len(blocks) == rep["blocks"]
    assert sorted(e[1] for e in entries) == list(range(len(blocks)))
    assert sum(len(b.seq) for b in blocks) == rep["seq_bytes"] and sum(len(b.qual) for b in blocks) == rep["qual_bytes"]
    # and back: three workers decode, the writer restores the input order
    back = tmp_path / "back.fastq"
    rep2 = run_tool(tool, "d", arc, back, "-t", 3)
    assert rep2["raw_bytes"] == raw.size
    assert np.array_equal(np.fromfile(back, dtype=np.uint8), raw)


def test_cpp_farm_two_logical_devices_and_four_workers(F, tool, tmp_path):
    """`-d 0,0`: the farm's device list with G = 2 entries (both device 0 on a one-GPU box), so that
    worker t -> devices[t mod G], one handle per worker on its device, and the process-wide cache of
    page-locked blocks (allocated portable: a block pinned by one worker is reused by any other)
    all run; the archive is the oracle's, block for block."""
    raw, _ = F.synth_fastq(21 << 20, 4, seed=13)
    src = tmp_path / "in.fastq"
    raw.tofile(src)
    arc = tmp_path / "out.fqc"
    rep = run_tool(tool, "c", src, arc, "-t", 4, "-R", 2, "-S", 4, "-d", "0,0")
    assert rep["devices"] == 2 and len(rep["blocks_per_worker"]) == 4 and sum(rep["blocks_per_worker"]) == rep["blocks"] >= 10
    check_archive_against_oracle(F, str(arc), raw, 4 << 20)
    back = tmp_path / "back.fastq"
    rep2 = run_tool(tool, "d", arc, back, "-t", 4, "-d", "0,0")
    assert rep2["devices"] == 2 and np.array_equal(np.fromfile(back, dtype=np.uint8), raw)


def test_a_damaged_block_ends_the_farm_with_an_error_not_a_hang(F, tool, tmp_path):
    """A worker that fails (corrupt FSE stream, malformed misc stream, truncated block) must bring the
    whole command down with exit code 1: the other workers finish their block and stop.  (With an
    ordered writer that waits for the failed worker's chunk this used to hang holding the GPU.)"""
    raw, _ = F.synth_fastq(8 << 20, 2, seed=14)
    src = tmp_path / "in.fastq"
    raw.tofile(src)
    arc = tmp_path / "ok.fqc"
    run_tool(tool, "c", src, arc, "-t", 2, "-R", 1, "-S", 2)
    data = bytearray(open(arc, "rb").read())
    _, _, _, blocks, entries = A.read_archive(str(arc))
    assert len(blocks) >= 6
    off = sorted(e[0] for e in entries)[2]  # the third block in the file
    for what, at in (("a bit inside the block's streams", off + 40 + len(blocks[0].seq) // 2), ("the size word of its first field", off + 12)):
        bad = bytearray(data)
        bad[at] ^= 0x10
        path = tmp_path / "bad.fqc"
        open(path, "wb").write(bad)
        r = subprocess.run([tool, "d", str(path), str(tmp_path / "bad.fastq"), "-t", "3"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 1, (what, r.stdout, r.stderr)
        assert "fqc_tool:" in r.stderr
        # a failed restore leaves nothing that looks like a restored file (the writer works in <name>.part and renames at the end)
        assert not os.path.exists(tmp_path / "bad.fastq") and not os.path.exists(str(tmp_path / "bad.fastq") + ".part")


def test_cpp_farm_with_decode_indexes_beside_the_archive(F, tool, tmp_path):
    """`c --index` (extension): the archive's blocks are what they are without it, `<archive>.fqx` holds every block's
    decode indexes, `d` uses the file when it lies there -- a damaged one ends the command (checksum), without one the
    archive decodes at the format's own pace"""
    raw, _ = F.synth_fastq(44 << 20, 4, seed=17)
    src = tmp_path / "in.fastq"
    raw.tofile(src)
    plain, arc = tmp_path / "plain.fqc", tmp_path / "indexed.fqc"
    run_tool(tool, "c", src, plain, "-t", 3, "-R", 8, "-S", 4)
    assert not os.path.exists(str(plain) + ".fqx")
    rep = run_tool(tool, "c", src, arc, "-t", 3, "-R", 8, "-S", 4, "--index")
    side = str(arc) + ".fqx"
    assert os.path.getsize(side) > 64 * rep["blocks"]
    a, b = A.read_archive(str(plain)), A.read_archive(str(arc))
    assert a[:3] == b[:3] and len(a[3]) == len(b[3]) == rep["blocks"] >= 5
    for x, y in zip(a[3], b[3]):   # blocks sorted by chunk: the same fields, byte for byte
        assert (x.idx, x.total, x.n_records, x.seq, x.qual, x.readlens, x.n_count, x.n_pos, x.fields) == \
               (y.idx, y.total, y.n_records, y.seq, y.qual, y.readlens, y.n_count, y.n_pos, y.fields)
    back = tmp_path / "back.fastq"
    run_tool(tool, "d", arc, back, "-t", 3)
    assert np.array_equal(np.fromfile(back, dtype=np.uint8), raw)
    os.remove(back)
    # a bit inside an index, a file without its trailer: the command fails, nothing is left behind
    data = bytearray(open(side, "rb").read())
    for what, bad in (("bit", bytes(data[:4000]) + bytes([data[4000] ^ 4]) + bytes(data[4001:])), ("cut", bytes(data[:-20]))):
        open(side, "wb").write(bad)
        r = subprocess.run([tool, "d", str(arc), str(back), "-t", "3"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 1 and "decode index file" in r.stderr, (what, r.stdout, r.stderr)
        assert not os.path.exists(back) and not os.path.exists(str(back) + ".part")
    os.remove(side)
    run_tool(tool, "d", arc, back, "-t", 3)
    assert np.array_equal(np.fromfile(back, dtype=np.uint8), raw)
    os.remove(back)
    # an index file that belongs to ANOTHER archive (left behind under the same name) is recognised and not used ...
    other = tmp_path / "other.fqc"
    raw2, _ = F.synth_fastq(20 << 20, 2, seed=18)
    src2 = tmp_path / "in2.fastq"
    raw2.tofile(src2)
    run_tool(tool, "c", src2, other, "-t", 2, "-R", 8, "-S", 4, "--index")
    os.replace(str(other) + ".fqx", side)
    r = subprocess.run([tool, "d", str(arc), str(back), "-t", "2"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "written for another archive" in r.stderr, (r.stdout, r.stderr)
    assert np.array_equal(np.fromfile(back, dtype=np.uint8), raw)
    # ... and a compression without --index to a name that has one removes it
    run_tool(tool, "c", src2, arc, "-t", 2, "-R", 8, "-S", 4)
    assert not os.path.exists(side)


def test_indexed_decode_through_the_host_pointer_calls(F):
    """fqgpu_encode_begin(FQGPU_F_DECODE_INDEX) .. fqgpu_encode_index, then fqgpu_decode_block_indexed: the streams are
    the oracle's with or without the index, the indexed decode restores the block from the ORACLE's streams, an index
    of another block is refused, one index alone is enough for its stream"""
    raw, _ = F.synth_fastq(20 << 20, 4, seed=19)
    recs = F.parse_fastq(raw)
    _, _, sft, qft = O.freq_tables(raw, recs)
    e = O.OracleCtx(sft, qft).encode(raw, recs)
    ctx = F.Context(sft, qft)
    ctx.set_index_stride(1 << 20)
    g = ctx.encode_raw(raw, flags=F.F_DECODE_INDEX)
    assert g["rc"] == 0 and all(np.array_equal(g[k], e[k]) for k in ("seq", "qual", "n_count", "n_pos"))
    si, qi = g["index"]
    n_sym = int(recs["len"].sum())
    assert si.size == 32 + ((n_sym - 1) >> 20) * (16 + 2 * 256) and qi.size == 32 + ((n_sym - 1) >> 20) * (16 + 2 * 8192)
    skeleton = O.blank_skeleton(raw, recs)
    for index in ((si, qi), (si, np.zeros(0, np.uint8)), (np.zeros(0, np.uint8), qi)):
        rc, out = ctx.decode_block(e["seq"], e["qual"], e["n_count"], e["n_pos"], recs, skeleton, index=index)
        assert rc == 0 and np.array_equal(out, raw)
    # an index made for another block (fewer symbols): refused as corrupt, and the handle works afterwards
    half = F.parse_fastq(raw[: raw.size // 2])
    end = int(half[-1]["qual_off"] + half[-1]["len"] + 1)
    g2 = ctx.encode_raw(raw[:end], flags=2)
    rc, _ = ctx.decode_block(e["seq"], e["qual"], e["n_count"], e["n_pos"], recs, skeleton, index=g2["index"])
    assert rc == -3   # FQGPU_E_CORRUPT
    # a snapshot's bit position damaged: the stride does not end where the next snapshot says
    bad = qi.copy()
    bad[32 + 2] ^= 1
    rc, _ = ctx.decode_block(e["seq"], e["qual"], e["n_count"], e["n_pos"], recs, skeleton, index=(si, bad))
    assert rc == -3
    rc, out = ctx.decode_block(e["seq"], e["qual"], e["n_count"], e["n_pos"], recs, skeleton)
    assert rc == 0 and np.array_equal(out, raw)
    ctx.close()


def test_encode_from_an_unparsed_chunk_through_the_c_abi(F):
    """fqgpu_encode_begin without a record table (the GPU finds the records), _records, _wait, _end:
    the table is the host parser's, the streams are the oracle's, a partial record at the end of
    the chunk is ignored and reported through used_len, N -> A lands in the caller's buffer."""
    raw, _ = F.synth_fastq(5 << 20, 4, seed=15)
    recs = F.parse_fastq(raw)
    _, _, sft, qft = O.freq_tables(raw, recs)
    e = O.OracleCtx(sft, qft).encode(raw, recs)
    ctx = F.Context(sft, qft)
    for tail in (b"", b"@SYN.x partial\nACGT"):
        chunk = np.concatenate([raw, np.frombuffer(tail, dtype=np.uint8)])
        g = ctx.encode_raw(chunk, flags=1)
        assert g["rc"] == 0 and g["used_len"] == raw.size and g["n_bases"] == int(recs["len"].sum())
        assert np.array_equal(g["recs"], recs)
        for k in ("seq", "qual", "readlens", "n_count", "n_pos"):
            assert np.array_equal(g[k], e[k]), k
        assert np.array_equal(g["raw_after"][: raw.size], e["raw_after"])
    # with the caller's table: the same bytes; and a second begin without an end drops the first block
    g = ctx.encode_raw(raw, flags=1, recs=recs)
    assert g["rc"] == 0 and all(np.array_equal(g[k], e[k]) for k in ("seq", "qual", "n_pos"))
    import ctypes
    n = ctypes.c_size_t(0)
    assert F.lib().fqgpu_encode_begin(ctx.h, raw.ctypes.data, raw.size, None, 0, 0, ctypes.byref(n), None, None) == 0
    g = ctx.encode_raw(raw, flags=0)
    assert g["rc"] == 0 and np.array_equal(g["seq"], e["seq"])
    # malformed chunks are refused, and the handle works afterwards
    bad = raw.copy()
    bad[int(recs[3]["qual_off"]) - 2] = ord("x")  # the '+' line
    assert ctx.encode_raw(bad)["rc"] == -4  # FQGPU_E_ARG
    assert ctx.encode_raw(raw)["rc"] == 0
    ctx.close()


def test_cpp_farm_accumulated_n_tables_like_the_reference(F, tool, tmp_path):
    """--accumulate-n: a worker's CompressedBuffersDst is never cleared of n_count / n_pos
    (src/compressed_buffers.h:58-68); the archive still decodes (pops from the end)."""
    raw, _ = F.synth_fastq(6 << 20, 4, seed=12)
    src = tmp_path / "in.fastq"
    raw.tofile(src)
    arc = tmp_path / "acc.fqc"
    rep = run_tool(tool, "c", src, arc, "-t", 1, "-R", 1, "-S", 1, "--accumulate-n")
    _, _, _, blocks, _ = A.read_archive(str(arc))
    sizes = [b.n_count[0] for b in blocks]
    assert all(b > a for a, b in zip(sizes, sizes[1:])) and sizes[-1] == 2 * sum(b.n_records for b in blocks)
    back = tmp_path / "back.fastq"
    run_tool(tool, "d", arc, back, "-t", 2)
    assert np.array_equal(np.fromfile(back, dtype=np.uint8), raw)
    assert rep["blocks"] == len(blocks)


@pytest.mark.parametrize("name", FIXTURES)
def test_fixture_archives_written_by_the_oracle_side_decode_on_the_gpu(F, tool, tmp_path, golden_dir, name):
    """write (Python writer, oracle streams) -> read (C++ Archive) -> GPU decode, and the other way
    round: GPU farm writes, Python reads, oracle decodes."""
    from test_archive import oracle_archive
    raw, recs = O.load_fastq(os.path.join(golden_dir, name + ".fastq"))
    n_blocks = 1 if len(recs) < 10 else 3
    arc = tmp_path / "oracle.fqc"
    oracle_archive(F, str(arc), raw, recs, n_blocks, order=list(range(n_blocks))[::-1])
    back = tmp_path / "back.fastq"
    run_tool(tool, "d", arc, back, "-t", 2)
    assert np.array_equal(np.fromfile(back, dtype=np.uint8), raw)
    # GPU side writes: one block, tables from the whole file = the oracle's own golden configuration
    src = tmp_path / "in.fastq"
    raw.tofile(src)
    arc2 = tmp_path / "gpu.fqc"
    run_tool(tool, "c", src, arc2, "-t", 1)
    blocks, _ = check_archive_against_oracle(F, str(arc2), raw, 128 << 20)
    assert len(blocks) == 1
    first_header, sft_b, qft_b, _, _ = A.read_archive(str(arc2))
    octx = O.OracleCtx(np.frombuffer(sft_b, dtype=O.SEQ_FT_DTYPE).copy(), np.frombuffer(qft_b, dtype=O.QUAL_FT_DTYPE).copy())
    b = blocks[0]
    n_count = F.memdecompress(np.frombuffer(b.n_count[1], dtype=np.uint8), b.n_count[0]).view(np.uint16)
    n_pos = F.memdecompress(np.frombuffer(b.n_pos[1], dtype=np.uint8), b.n_pos[0]).view(np.uint16)
    rc, out = octx.decode(np.frombuffer(b.seq, dtype=np.uint8), np.frombuffer(b.qual, dtype=np.uint8), n_count, n_pos, recs,
                          O.blank_skeleton(raw, recs))
    assert rc == 0 and np.array_equal(out, raw)


def test_config5_oracle_written_archive_mixed_lengths(F, tool, tmp_path):
    """BASELINE configs[4] in small, on a real `.fqc`-shaped file: config-4 reads, blocks coded by the
    oracle and written in a shuffled completion order, decoded by four GPU workers."""
    from test_archive import oracle_archive
    raw, n = F.synth_fastq(9 << 20, 4, seed=5)
    recs = F.parse_fastq(raw)
    arc = tmp_path / "cfg5.fqc"
    order = [4, 1, 6, 0, 3, 5, 2]
    oracle_archive(F, str(arc), raw, recs, 7, order=order)
    back = tmp_path / "back.fastq"
    rep = run_tool(tool, "d", arc, back, "-t", 4)
    assert sum(rep["blocks_per_worker"]) == 7 and rep["raw_bytes"] == raw.size
    assert np.array_equal(np.fromfile(back, dtype=np.uint8), raw)


def test_two_fresh_processes_share_one_job(F, tmp_path):
    """N > 1 as the driver launches it: one process per rank, started before any GPU call, gloo for the
    table broadcast and the barriers (no RCCL), block b -> rank b mod 2; here both ranks sit on
    device 0.  The union of what the two processes wrote equals the oracle's streams of the job."""
    out = tmp_path / "streams"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", WORLD_SIZE="2", LOCAL_RANK="0",
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "fqcomp28_amd.farm", "--mib", "12", "--block-mib", "2", "--sample-mib", "4",
           "--device", "0", "--out", str(out)]
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd=ROOT)
             for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    rep = json.loads([ln for ln in outs[0].splitlines() if ln.startswith("{")][-1])
    assert rep["world"] == 2 and rep["n_blocks"] == 6
    assert [r["blocks"] for r in rep["ranks"]] == [[0, 2, 4], [1, 3, 5]]
    assert rep["ranks"][0]["pid"] != rep["ranks"][1]["pid"]
    # the same job, here: oracle tables from the first 4 MiB, oracle streams of every block
    from fqcomp28_amd.farm import make_job
    job = make_job(F, 12 << 20, 2 << 20)
    sample = np.concatenate(job)[: 4 << 20]   # the first --sample-mib bytes of the job, whole records of them
    srecs = F.parse_fastq(sample)
    _, _, sft, qft = O.freq_tables(sample, srecs)
    assert np.fromfile(out / "tables.seq_ft", dtype=np.uint8).tobytes() == sft.tobytes()
    assert np.fromfile(out / "tables.qual_ft", dtype=np.uint8).tobytes() == qft.tobytes()
    octx = O.OracleCtx(sft, qft)
    for b, raw in enumerate(job):
        e = octx.encode(raw, F.parse_fastq(raw))
        for k in ("seq", "qual", "readlens", "n_count", "n_pos"):
            got = np.fromfile(out / ("block_%d.%s" % (b, k)), dtype=e[k].dtype)
            assert np.array_equal(got, e[k]), (b, k)
