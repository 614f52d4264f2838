"""GPU parity tests proper: the HIP path (through the C ABI of include/fqgpu.h) against the
CPU oracle on the same inputs, against the committed golden vectors, and -- at BASELINE.json's
full size -- through the encode -> wipe -> decode round trip.  Integer/byte work: bit-exact."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

FIXTURES = ["SRR065390_sub_1", "without_ns", "SRR065390_sub_2", "SRR065390_1_first5"]


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def F():
    import fqcomp28_amd as F
    assert F.device_count() >= 1, "no GPU visible: the product path has no CPU fallback"
    return F


@pytest.fixture(scope="module")
def expected(golden_dir):
    with open(os.path.join(golden_dir, "expected.json")) as fh:
        return json.load(fh)


def assert_same_encoding(g, e):
    assert g["rc"] == 0 and e["rc"] == 0
    for k in ("seq", "qual", "readlens", "n_count", "n_pos"):
        assert np.array_equal(np.asarray(g[k]), np.asarray(e[k])), k


# ---------------------------------------------------------------- reference fixtures
@pytest.mark.parametrize("name", FIXTURES)
def test_fixture_tables_and_streams(F, name, golden_dir, expected):
    raw, recs = O.load_fastq(os.path.join(golden_dir, name + ".fastq"))
    assert np.array_equal(F.parse_fastq(raw), recs)
    sc, qc, sft, qft = O.freq_tables(raw, recs)
    gs, gq, gsc, gqc = F.freq_tables(raw, recs, want_counts=True)
    # calculateFreqTable + makeNormalizedFreqTable (a3/a4/a5)
    assert np.array_equal(gsc, sc) and np.array_equal(gqc, qc)
    assert gs.tobytes() == sft.tobytes() and gq.tobytes() == qft.tobytes()
    x = expected[name]
    assert sha(gs) == x["seq_ft_sha1"] and sha(gq) == x["qual_ft_sha1"]
    # encodeChunk seq/qual part (a7-a13) from GPU-built tables, against oracle AND golden
    ctx = F.Context(gs, gq)
    g = ctx.encode_block(raw, recs, flags=1)
    e = O.OracleCtx(sft, qft).encode(raw, recs)
    assert_same_encoding(g, e)
    assert np.array_equal(g["raw_after"], e["raw_after"])  # N -> A written back
    assert (len(g["seq"]), sha(g["seq"])) == (x["seq_len"], x["seq_sha1"])
    assert (len(g["qual"]), sha(g["qual"])) == (x["qual_len"], x["qual_sha1"])
    assert sha(g["n_count"]) == x["n_count_sha1"] and sha(g["n_pos"]) == x["n_pos_sha1"]
    # decodeChunk second pass (a14-a17): GPU decodes the ORACLE's streams
    rc, out = ctx.decode_block(e["seq"], e["qual"], e["n_count"], e["n_pos"], recs, O.blank_skeleton(raw, recs))
    assert rc == 0 and np.array_equal(out, raw)
    ctx.close()


def test_first5_golden_bytes(F, golden_dir):
    name = "SRR065390_1_first5"
    raw, recs = O.load_fastq(os.path.join(golden_dir, name + ".fastq"))
    sft, qft = F.freq_tables(raw, recs)
    g = F.Context(sft, qft).encode_block(raw, recs)
    assert np.array_equal(g["seq"], np.fromfile(os.path.join(golden_dir, name + ".seq.bin"), dtype=np.uint8))
    assert np.array_equal(g["qual"], np.fromfile(os.path.join(golden_dir, name + ".qual.bin"), dtype=np.uint8))
    assert np.array_equal(g["n_pos"], np.fromfile(os.path.join(golden_dir, name + ".n_pos.bin"), dtype=np.uint16))
    assert sft.tobytes() == open(os.path.join(golden_dir, name + ".seq_ft.bin"), "rb").read()


def test_device_tables_match_oracle_word_for_word(F, golden_dir):
    """FSE_buildCTable_wksp / FSE_buildDTable_wksp for every context (a6, a14)."""
    raw, recs = O.load_fastq(os.path.join(golden_dir, "SRR065390_sub_1.fastq"))
    _, _, sft, qft = O.freq_tables(raw, recs)
    ctx = F.Context(sft, qft)
    L = O.lib()
    for stream, ft, alpha, models in ((0, sft, 4, range(256)), (1, qft, 64, range(0, 8192, 7))):
        for m in models:
            ct, dt = ctx.dump_tables(stream, m)
            log = int(ft["logs"][0][m])
            oc = np.zeros(L.fo_ctable_words(log, alpha - 1), dtype=np.uint32)
            od = np.zeros(L.fo_dtable_words(log), dtype=np.uint32)
            norm = np.ascontiguousarray(ft["norm"][0][m])
            assert L.fo_build_ctable(O.ptr(oc), O.ptr(norm), alpha - 1, log) == 0
            assert L.fo_build_dtable(O.ptr(od), O.ptr(norm), alpha - 1, log) == 0
            assert np.array_equal(ct, oc), (stream, m)
            assert np.array_equal(dt, od), (stream, m)
    ctx.close()


def test_tables_from_counts_normalisation_corner_cases(F):
    """Count vectors that hit the low-probability (-1), the rounding and the normalizeM2 paths."""
    rng = np.random.default_rng(5)
    sc = 1 + rng.integers(0, 3000, (256, 4)).astype(np.uint32)
    sc[3] = [1, 1, 1, 1]
    sc[4] = [1, 1, 1, 900000]
    qc = np.ones((8192, 64), dtype=np.uint32)
    for c in range(0, 8192, 3):
        k = rng.integers(2, 64)
        qc[c, :k] += rng.integers(0, 60, k).astype(np.uint32)
        qc[c, rng.integers(0, 64)] += np.uint32(rng.integers(0, 5000))
    for c in range(1, 8192, 97):
        qc[c] = 1 + (rng.pareto(0.6, 64) * 30).astype(np.uint32)
    L = O.lib()
    osft = np.zeros(1, dtype=O.SEQ_FT_DTYPE)
    oqft = np.zeros(1, dtype=O.QUAL_FT_DTYPE)
    assert L.fqo_seq_ft_from_counts(O.ptr(sc), O.ptr(osft)) == 0
    assert L.fqo_qual_ft_from_counts(O.ptr(qc), O.ptr(oqft)) == 0
    gs, gq = F.tables_from_counts(sc, qc)
    assert gs.tobytes() == osft.tobytes()
    assert gq.tobytes() == oqft.tobytes()
    # and the tables built from them (covers logs 5..11 and many -1 symbols)
    ctx = F.Context(gs, gq)
    for m in range(0, 8192, 11):
        ct, dt = ctx.dump_tables(1, m)
        log = int(oqft["logs"][0][m])
        oc = np.zeros(L.fo_ctable_words(log, 63), dtype=np.uint32)
        od = np.zeros(L.fo_dtable_words(log), dtype=np.uint32)
        norm = np.ascontiguousarray(oqft["norm"][0][m])
        L.fo_build_ctable(O.ptr(oc), O.ptr(norm), 63, log)
        L.fo_build_dtable(O.ptr(od), O.ptr(norm), 63, log)
        assert np.array_equal(ct, oc) and np.array_equal(dt, od), m
    ctx.close()


@pytest.mark.parametrize("mode", [2, 3, 4, 5])
def test_dataset_analysis_through_the_encoders_sort_counts_what_the_reference_counts(F, mode):
    """Samples of a million symbols and more take the quality histogram through the encoder's K1-K3
    (symbols sorted by context, local histograms) instead of scattered atomics: the raw counts --
    not only the normalised tables -- equal FSE_Quality::calculateFreqTable's (oracle), for the
    BASELINE configs, binned qualities (few hot contexts) and one context only."""
    raw, _ = F.synth_fastq(9 << 20, mode, seed=31)
    recs = F.parse_fastq(raw)
    assert int(recs["len"].sum()) > (1 << 20)
    sc, qc, sft, qft = O.freq_tables(raw, recs)
    gs, gq, gsc, gqc = F.freq_tables(raw, recs, want_counts=True)
    assert np.array_equal(gsc, sc) and np.array_equal(gqc, qc)
    assert gs.tobytes() == sft.tobytes() and gq.tobytes() == qft.tobytes()
    # a quality above Q63 is refused on this path too (src/fse_quality.cpp:88 throws)
    bad = raw.copy()
    bad[int(recs[len(recs) // 2]["qual_off"]) + 5] = 33 + 64
    with pytest.raises(F.FqgpuError):
        F.freq_tables(bad, recs)


# ---------------------------------------------------------------- how the chains are cut never shows in the output
@pytest.mark.parametrize("seg,seq_generic,lanes", [(2, True, 1), (16, True, 3), (64, False, 2), (100000, True, 1),
                                                   (1024, False, 4)])
def test_chain_parameters_never_change_the_bits(F, golden_dir, seg, seq_generic, lanes):
    raw, recs = O.load_fastq(os.path.join(golden_dir, "SRR065390_sub_2.fastq"))
    _, _, sft, qft = O.freq_tables(raw, recs)
    e = O.OracleCtx(sft, qft).encode(raw, recs)
    ctx = F.Context(sft, qft)
    ctx.set_chain_params(seg, seq_generic)
    ctx.set_lanes(lanes)
    blocks = [ctx.dblock(raw, recs) for _ in range(3)]  # several blocks in flight on the lanes
    for b in blocks:
        b.encode()
    ctx.sync()
    for b in blocks:
        g = b.fetch()
        for k in ("seq", "qual", "n_count", "n_pos"):
            assert np.array_equal(g[k], e[k]), k
        ls, lq = b.longest_chain()
        assert 0 < lq <= int(recs["len"].sum()) and 0 < ls
        if seg == 100000:  # one segment per context: every chain is walked by one lane
            assert lq == int(np.bincount(_qual_ctx_of(raw, recs), minlength=8192).max())
        b.close()
    ctx.close()


def _fastq_with_lengths(lengths, seed=3):
    rng = np.random.default_rng(seed)
    parts = []
    for i, L in enumerate(lengths):
        seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=L)]
        if L > 20:
            seq[rng.integers(0, L, size=max(1, L // 200))] = ord("N")
        q = np.clip(np.rint(rng.normal(30, 6, size=L)), 2, 41).astype(np.uint8) + 33
        parts.append(b"@r%d\n" % i + seq.tobytes() + b"\n+\n" + q.tobytes() + b"\n")
    raw = np.frombuffer(b"".join(parts), dtype=np.uint8).copy()
    return raw, O.parse_fastq(raw)


@pytest.mark.parametrize("case", ["min_length_3", "max_length_65535", "mixed_extremes"])
def test_extreme_read_lengths(F, case):
    """Reads of the minimum (3) and maximum (65535 = readlen_t) length and a mix of both: the
    record walker's 64-record windows cover anything from 192 symbols to several tiles."""
    if case == "min_length_3":
        lengths = [3] * 60000
    elif case == "max_length_65535":
        lengths = [65535] * 12
    else:
        lengths = ([3, 4, 5, 65535, 7, 150, 3] * 40) + [65535, 3, 3, 3, 30000]
    raw, recs = _fastq_with_lengths(lengths)
    assert list(recs["len"]) == lengths
    _, _, sft, qft = O.freq_tables(raw, recs)
    e = O.OracleCtx(sft, qft).encode(raw, recs)
    ctx = F.Context(sft, qft)
    g = ctx.encode_block(raw, recs, flags=1)
    assert_same_encoding(g, e)
    db = ctx.dblock(raw)  # GPU record parser on the same bytes
    assert np.array_equal(db.records(), recs)
    db.close()
    rc, out = ctx.decode_block(g["seq"], g["qual"], g["n_count"], g["n_pos"], recs, O.blank_skeleton(raw, recs))
    assert rc == 0 and np.array_equal(out, raw)
    ctx.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("FQ_FUZZ_SEEDS", "12"))))
def test_random_blocks_match_oracle(F, seed):
    """Randomised blocks: read lengths from 3 to a few thousand, quality alphabets from one value to
    all 64 (chars 33..96), bases with N runs, tables from the block itself or from another block
    (symbols the tables have never seen); encode must equal the oracle and decode must restore."""
    rng = np.random.default_rng(1000 + seed)
    n_reads = int(rng.integers(1, 4000))
    kind = seed % 4
    if kind == 0:
        lengths = rng.integers(3, 12, size=n_reads)
    elif kind == 1:
        lengths = rng.integers(3, 400, size=n_reads)
    elif kind == 2:
        lengths = np.where(rng.random(n_reads) < 0.02, rng.integers(2000, 9000, size=n_reads), rng.integers(30, 160, size=n_reads))
    else:
        lengths = np.full(n_reads, int(rng.integers(3, 300)))
    qlo = int(rng.integers(0, 60)); qhi = int(rng.integers(qlo, 64))

    def make(lengths, salt):
        r2 = np.random.default_rng(5000 + 17 * seed + salt)
        parts = []
        for i, L in enumerate(lengths):
            L = int(L)
            seq = np.frombuffer(b"ACGT", dtype=np.uint8)[r2.integers(0, 4, size=L)].copy()
            if r2.random() < 0.3:
                a = int(r2.integers(0, L)); seq[a: a + int(r2.integers(1, 8))] = ord("N")
            q = (r2.integers(qlo, qhi + 1, size=L) + 33).astype(np.uint8)
            parts.append(b"@x%d\n" % i + seq.tobytes() + b"\n+\n" + q.tobytes() + b"\n")
        raw = np.frombuffer(b"".join(parts), dtype=np.uint8).copy()
        return raw, O.parse_fastq(raw)

    raw, recs = make(lengths, 0)
    if seed % 3 == 0:   # tables from a different sample
        traw, trecs = make(lengths[: max(1, n_reads // 3)], 1)
    else:
        traw, trecs = raw, recs
    _, _, sft, qft = O.freq_tables(traw, trecs)
    gs, gq = F.freq_tables(traw, trecs)
    assert gs.tobytes() == sft.tobytes() and gq.tobytes() == qft.tobytes()
    e = O.OracleCtx(sft, qft).encode(raw, recs)
    ctx = F.Context(sft, qft)
    g = ctx.encode_block(raw, recs, flags=1)
    assert g["rc"] == e["rc"]
    if e["rc"] == 0:
        assert_same_encoding(g, e)
        rc, out = ctx.decode_block(g["seq"], g["qual"], g["n_count"], g["n_pos"], recs, O.blank_skeleton(raw, recs))
        assert rc == 0 and np.array_equal(out, raw)
    ctx.close()


def _rescale_tables(ft, new_log):
    """FreqTable POD with every context renormalised to 2^new_log (largest-remainder on the old
    normalised counts, -1 entries kept): tables a foreign writer could have produced."""
    out = ft.copy()
    norm = out["norm"][0]
    logs = out["logs"][0]
    for c in range(norm.shape[0]):
        old = norm[c].astype(np.int64)
        cnt = np.where(old == -1, 1, old)
        tot = int(cnt.sum())
        target = 1 << new_log
        scaled = np.where(cnt > 0, np.maximum(1, cnt * target // tot), 0)
        scaled[np.argmax(scaled)] += target - int(scaled.sum())
        assert scaled.min() >= 0 and int(scaled.sum()) == target and scaled[np.argmax(scaled)] > 0
        norm[c] = np.where((old == -1) & (scaled == 1), -1, scaled).astype(norm.dtype)
        logs[c] = new_log
    out["max_log"][0] = new_log
    return out


@pytest.mark.parametrize("log", [12, 5])
def test_foreign_table_logs(F, log):
    """Tables with log 12 everywhere (the reference never builds them: FSE_DEFAULT_TABLELOG = 11, but
    the format allows them) and with log 5 where the alphabet fits: 64 states per lane in the
    segment-function kernels, one-symbol table for the sequence stream, 8.7 KB CTables."""
    raw, recs = _synth(F, 2, 5 << 20)
    _, _, sft, qft = O.freq_tables(raw, recs)
    sft2 = _rescale_tables(sft, log)
    qft2 = _rescale_tables(qft, 12) if log == 12 else qft  # (64 symbols do not fit a tiny table)
    octx = O.OracleCtx(sft2, qft2)
    ctx = F.Context(sft2, qft2)
    e = octx.encode(raw, recs)
    g = ctx.encode_block(raw, recs)
    if log == 5:
        # 5-bit probabilities of uniform bases cost more than 2 bits per base: the sequence stream
        # does not fit the reference's capacity rule and both coders say so
        assert e["rc"] == -1 and g["rc"] == -1
        cap = int(recs["len"].sum())  # generous capacities: same bytes again
        e = octx.encode(raw, recs, seq_cap=cap, qual_cap=cap)
        g = ctx.encode_block(raw, recs, seq_cap=cap, qual_cap=cap)
    assert_same_encoding(g, e)
    rc, out = ctx.decode_block(g["seq"], g["qual"], g["n_count"], g["n_pos"], recs, O.blank_skeleton(raw, recs))
    assert rc == 0 and np.array_equal(out, raw)
    ctx.close()


def test_one_handle_per_thread(F):
    """The reference runs one workspace per worker thread (src/process.cpp:49-54): three host threads,
    each with its own handle on the same GPU, code different blocks at the same time."""
    import threading
    jobs = []
    for i, mode in enumerate((2, 4, 2)):
        raw, recs = _synth(F, mode, (3 + i) << 20, seed=40 + i)
        _, _, sft, qft = O.freq_tables(raw, recs)
        jobs.append((raw, recs, sft, qft, O.OracleCtx(sft, qft).encode(raw, recs)))
    errors = []

    def worker(raw, recs, sft, qft, e):
        try:
            ctx = F.Context(sft, qft)
            for _ in range(3):
                g = ctx.encode_block(raw, recs)
                assert_same_encoding(g, e)
                rc, out = ctx.decode_block(g["seq"], g["qual"], g["n_count"], g["n_pos"], recs, O.blank_skeleton(raw, recs))
                assert rc == 0 and np.array_equal(out, raw)
            ctx.close()
        except BaseException as ex:  # noqa: BLE001 - reported by the main thread
            errors.append(repr(ex))

    threads = [threading.Thread(target=worker, args=j) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_partition_fallback_without_lane_ordered_lds_atomics(F, golden_dir, monkeypatch):
    """The partition kernels rank with lane-ordered LDS atomics when the handle's probe confirms the
    ordering; FQGPU_NO_LDS_ATOMIC_RANK forces the ballot-match kernels a device without that
    property would get.  Same bits either way."""
    raw, recs = O.load_fastq(os.path.join(golden_dir, "SRR065390_sub_2.fastq"))
    _, _, sft, qft = O.freq_tables(raw, recs)
    e = O.OracleCtx(sft, qft).encode(raw, recs)
    monkeypatch.setenv("FQGPU_NO_LDS_ATOMIC_RANK", "1")
    ctx = F.Context(sft, qft)
    monkeypatch.delenv("FQGPU_NO_LDS_ATOMIC_RANK")
    g = ctx.encode_block(raw, recs)
    assert_same_encoding(g, e)
    big, brecs = _synth(F, 2, 6 << 20)
    _, _, sft2, qft2 = O.freq_tables(big, brecs)
    ctx.close()
    monkeypatch.setenv("FQGPU_NO_LDS_ATOMIC_RANK", "1")
    ctx = F.Context(sft2, qft2)
    monkeypatch.delenv("FQGPU_NO_LDS_ATOMIC_RANK")
    assert_same_encoding(ctx.encode_block(big, brecs), O.OracleCtx(sft2, qft2).encode(big, brecs))
    ctx.close()


@pytest.mark.parametrize("mode", [2, 3])
def test_partition_fallback_at_bench_scale(F, monkeypatch, mode):
    """The ballot-match partition (what a device without lane-ordered LDS atomics would run) on a block
    of the size class the bench uses -- 96 MiB: 1 400 tiles, 22 tile groups, quality runs in every
    context -- for BASELINE's reads and for binned qualities (one context holding half a tile)."""
    big, brecs = _synth(F, mode, 96 << 20)
    sample = brecs[brecs["qual_off"] < (32 << 20)]
    sft, qft = F.freq_tables(big, sample)
    monkeypatch.setenv("FQGPU_NO_LDS_ATOMIC_RANK", "1")
    ctx = F.Context(sft, qft)
    monkeypatch.delenv("FQGPU_NO_LDS_ATOMIC_RANK")
    g = ctx.encode_block(big, brecs)
    ctx.close()
    assert_same_encoding(g, O.OracleCtx(sft, qft).encode(big, brecs))


def _qual_ctx_of(raw, recs):
    """context of every quality symbol (numpy restatement of FSE_Quality::calcContext)."""
    out = []
    for r in recs:
        q = raw[r["qual_off"]: r["qual_off"] + r["len"]].astype(np.int64) - 33
        a = np.concatenate(([0], q[:-1])); b = np.concatenate(([0, 0], q[:-2])); c = np.concatenate(([0, 0, 0], q[:-3]))
        out.append(((np.maximum(b, c) << 6) + a) & 0xFFF | ((b == c).astype(np.int64) << 12))
    return np.concatenate(out)


# ---------------------------------------------------------------- BASELINE.json configurations (reduced sizes vs oracle)
def _synth(F, mode, size, seed=28):
    raw, n = F.synth_fastq(size, mode, seed=seed)
    recs = F.parse_fastq(raw)
    assert len(recs) == n
    return raw, recs


def test_config1_uniform_q40_degenerate_context(F):
    """configs[0]: 10k x 150 bp, all quals 'I' (one context holds ~98% of the symbols and its
    state chain never forgets: worst case for the speculation)."""
    raw, recs = _synth(F, 1, 10000 * 335 + 4096)
    raw = raw[: recs[9999]["qual_off"] + 151] if len(recs) > 10000 else raw
    recs = F.parse_fastq(raw)
    assert len(recs) == 10000
    sft, qft = F.freq_tables(raw, recs)
    _, _, osft, oqft = O.freq_tables(raw, recs)
    assert sft.tobytes() == osft.tobytes() and qft.tobytes() == oqft.tobytes()
    e = O.OracleCtx(osft, oqft).encode(raw, recs)
    ctx = F.Context(sft, qft)
    g = ctx.encode_block(raw, recs)
    assert_same_encoding(g, e)
    assert len(g["n_pos"]) > 0
    rc, out = ctx.decode_block(g["seq"], g["qual"], g["n_count"], g["n_pos"], recs, O.blank_skeleton(raw, recs))
    assert rc == 0 and np.array_equal(out, raw)
    ctx.close()


def _rewrite_bases(raw, recs, make):
    """synthetic block with its bases replaced by make(n) -> uint8 codes 0..3 (A, C, G, T)"""
    raw = raw.copy()
    n = int(recs["len"].sum())
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)[make(n)]
    at = 0
    for r in recs:
        raw[r["seq_off"]: r["seq_off"] + r["len"]] = letters[at: at + r["len"]]
        at += int(r["len"])
    return raw


@pytest.mark.parametrize("group", [(1, 1), (3, 1), (8, 1), (16, 4)])
@pytest.mark.parametrize("bases", ["uniform", "all_A"])
def test_sequence_segment_groups_are_exact(F, bases, group):
    """A wave of k_seq_setfunc walks a group of consecutive segments and writes one function per
    segment boundary, all from the group's entry state; k_seq_resolve reads a group's functions in
    one round.  Any group size (incl. ragged last groups, chains shorter than a group, and the
    never-collapsing single-symbol context) gives the oracle's bits."""
    raw, recs = _synth(F, 2, 6 << 20)
    if bases == "all_A":
        raw = _rewrite_bases(raw, recs, lambda n: np.zeros(n, dtype=np.int64))
    _, _, sft, qft = O.freq_tables(raw, recs)
    e = O.OracleCtx(sft, qft).encode(raw, recs)
    ctx = F.Context(sft, qft)
    ctx.set_chain_params(0, seq_segment=1024, seq_group=group)
    g = ctx.encode_block(raw, recs)
    assert_same_encoding(g, e)
    ctx.close()


@pytest.mark.parametrize("bases", ["uniform", "skewed", "markov", "all_A", "mostly_A"])
@pytest.mark.parametrize("segment", [1024, 4096, 20000, 1 << 30])
def test_sequence_segment_functions_are_exact(F, bases, segment):
    """The sequence chains are cut into segments whose entry states come from per-segment state
    functions (k_seq_setfunc / k_seq_resolve / k_seq_emit): any segment length gives the oracle's
    bits, for near-uniform tables (the state sets stay large), skewed ones (they collapse fast) and
    single-symbol contexts (no state ever merges: the function is carried for all 2^log states)."""
    raw, recs = _synth(F, 2, 6 << 20)
    rng = np.random.default_rng(5)
    if bases == "skewed":
        raw = _rewrite_bases(raw, recs, lambda n: rng.choice(4, size=n, p=[0.55, 0.05, 0.1, 0.3]))
    elif bases == "markov":
        def chain(n):  # 70 %: repeat the previous base, else a fresh uniform one
            fresh = rng.integers(0, 4, size=n)
            keep = rng.random(n) < 0.7
            keep[0] = False
            return fresh[np.maximum.accumulate(np.where(keep, 0, np.arange(n)))]
        raw = _rewrite_bases(raw, recs, chain)
    elif bases == "all_A":
        raw = _rewrite_bases(raw, recs, lambda n: np.zeros(n, dtype=np.int64))
    elif bases == "mostly_A":
        raw = _rewrite_bases(raw, recs, lambda n: rng.choice(4, size=n, p=[0.997, 0.001, 0.001, 0.001]))
    _, _, sft, qft = O.freq_tables(raw, recs)
    e = O.OracleCtx(sft, qft).encode(raw, recs)
    ctx = F.Context(sft, qft)
    ctx.set_chain_params(0, seq_segment=segment)
    g = ctx.encode_block(raw, recs)
    assert_same_encoding(g, e)
    ctx.close()


def test_long_sequence_chain_whose_functions_fill_its_last_item_exactly(F):
    """One context holding the block, with a number of segments such that its segment functions (all segments but
    the last) fill the 64-group items of the three-level resolve EXACTLY: the chain's last segment then lies
    behind the last item's groups.  (Found by tools/soak_roundtrip.py in round 3: the expand step stopped at the
    end of the last group and left that segment's entry state as the scratch buffer had it -- two bytes of the
    state flush wrong, or right by luck of what the buffer held; hence one handle for all sizes here.)
    Constant reads (every base A): context AAAA has 146 of a read's 150 symbols."""
    raw, recs = _synth(F, 5, 3 << 20)
    assert len(recs) >= 7192
    head = int(recs[999]["qual_off"] + recs[999]["len"] + 1)
    _, _, sft, qft = O.freq_tables(raw[:head], recs[:1000])
    octx = O.OracleCtx(sft, qft)
    for params, counts in (({"seq_segment": 1024, "seq_group": 4}, list(range(3588, 3600)) + list(range(5384, 5396))),   # 512 / 768 functions in groups of 4
                           ({}, list(range(7180, 7192)))):                                                        # the sizes the soak tripped over
        ctx = F.Context(sft, qft)
        ctx.set_lanes(1)
        if params:
            ctx.set_chain_params(0, **params)
        for n_recs in counts:
            end = int(recs[n_recs - 1]["qual_off"] + recs[n_recs - 1]["len"] + 1)
            braw, brecs = raw[:end], recs[:n_recs]
            e = octx.encode(braw, brecs)
            b = ctx.dblock(braw, brecs)
            b.encode()
            ctx.sync()
            assert b.status()[0] == e["rc"] == 0
            g = b.fetch()
            for k in ("seq", "qual", "readlens", "n_count", "n_pos"):
                assert np.array_equal(g[k], e[k]), (params, n_recs, k)
            b.close()
        ctx.close()
    octx.close()


def test_constant_phred0_qualities_fill_whole_tiles_with_one_context(F):
    """All qualities '!' (Phred 0): calcContext(0, 0, 0) for EVERY position, the read starts
    included, so whole 65536-symbol tiles hold one context -- the one input on which K1's 16-bit
    tile counters (two per word) overflow and the histogram is rebuilt from the first symbol's
    context, and on which K3's 16-bit cursor reaches 65536 with the last symbol of a tile."""
    raw, recs = _synth(F, 2, 6 << 20)
    raw = raw.copy()
    for r in recs:
        raw[r["qual_off"]: r["qual_off"] + r["len"]] = ord("!")
    _, _, sft, qft = O.freq_tables(raw, recs)
    e = O.OracleCtx(sft, qft).encode(raw, recs)
    ctx = F.Context(sft, qft)
    g = ctx.encode_block(raw, recs)
    assert_same_encoding(g, e)
    rc, out = ctx.decode_block(g["seq"], g["qual"], g["n_count"], g["n_pos"], recs, O.blank_skeleton(raw, recs))
    assert rc == 0 and np.array_equal(out, raw)
    ctx.close()


@pytest.mark.parametrize("segment", [1024, 4096])
@pytest.mark.parametrize("size_mib", [3, 24])
def test_two_quality_levels_every_segment_needs_its_full_function(F, segment, size_mib):
    """Synth mode 6: two quality levels at 30/70 %, nothing else -- no reset symbol, no narrow (anchor) symbol, no uniform
    segment: every quality segment is opaque and gets its entry state from full segment functions composed along chains of
    hundreds to thousands of segments (eight contexts hold the whole stream: three-level resolves, combining ranker in K3).
    Bit for bit, both directions.  (bench.py: encode_two_levels_MBps.)"""
    raw, recs = _synth(F, 6, size_mib << 20)
    _, _, sft, qft = O.freq_tables(raw, recs)
    e = O.OracleCtx(sft, qft).encode(raw, recs)
    ctx = F.Context(sft, qft)
    ctx.set_chain_params(segment)
    b = ctx.dblock(raw, recs)
    b.encode()
    g = b.fetch()
    assert_same_encoding(dict(g, rc=b.status()[0]), e)
    cls = b.qual_segment_classes()
    assert cls["opaque"] > 0.9 * sum(cls.values()) and cls["transparent"] == 0, cls
    b.wipe()
    ctx.decode_dblocks([b])
    assert b.status()[0] == 0 and np.array_equal(b.fetch_raw(), raw)
    b.close()
    ctx.close()


@pytest.mark.parametrize("segment", [1024, 4096])
@pytest.mark.parametrize("quals", ["binned", "two_levels_rare_third"])
def test_quality_tables_without_reset_symbols(F, quals, segment):
    """Binned qualities: no symbol has a normalised count of 1, so (almost) every quality segment is
    opaque and gets its entry state from segment functions composed over runs of many segments
    (k_seg_setfunc / k_seg_compose / k_seg_resolve2/3); a rare third level adds a few transparent
    segments in between."""
    raw, recs = _synth(F, 2, 6 << 20)
    rng = np.random.default_rng(11)
    n = int(recs["len"].sum())
    keep = rng.random(n) < 0.85
    keep[0] = False
    if quals == "binned":
        levels, p = np.frombuffer(b"#-8F", dtype=np.uint8), [0.05, 0.1, 0.15, 0.7]
    else:
        levels, p = np.frombuffer(b"#F:", dtype=np.uint8), [0.3, 0.69995, 0.00005]
    fresh = rng.choice(len(levels), size=n, p=p)
    q = levels[fresh[np.maximum.accumulate(np.where(keep, 0, np.arange(n)))]]
    raw = raw.copy()
    at = 0
    for r in recs:
        raw[r["qual_off"]: r["qual_off"] + r["len"]] = q[at: at + r["len"]]
        at += int(r["len"])
    _, _, sft, qft = O.freq_tables(raw, recs)
    e = O.OracleCtx(sft, qft).encode(raw, recs)
    ctx = F.Context(sft, qft)
    ctx.set_chain_params(segment)
    g = ctx.encode_block(raw, recs)
    assert_same_encoding(g, e)
    rc, out = ctx.decode_block(g["seq"], g["qual"], g["n_count"], g["n_pos"], recs, O.blank_skeleton(raw, recs))
    assert rc == 0 and np.array_equal(out, raw)
    ctx.close()


@pytest.mark.parametrize("mode,size", [(2, 12 << 20), (4, 12 << 20)])
def test_config2_and_config4_blocks_match_oracle(F, mode, size):
    """configs[1]/[3] at 12 MiB: tables from a different sample (first 4 MiB) than the coded block."""
    raw, recs = _synth(F, mode, size)
    sample_recs = recs[recs["qual_off"] < (4 << 20)]
    sft, qft = F.freq_tables(raw, sample_recs)
    _, _, osft, oqft = O.freq_tables(raw, sample_recs)
    assert sft.tobytes() == osft.tobytes() and qft.tobytes() == oqft.tobytes()
    e = O.OracleCtx(osft, oqft).encode(raw, recs)
    ctx = F.Context(sft, qft)
    g = ctx.encode_block(raw, recs, flags=1)
    assert_same_encoding(g, e)
    assert np.array_equal(g["raw_after"], e["raw_after"])
    if mode == 4:
        assert len(g["n_pos"]) > 10000 and recs["len"].min() < 60 and recs["len"].max() > 290
    rc, out = ctx.decode_block(e["seq"], e["qual"], e["n_count"], e["n_pos"], recs, O.blank_skeleton(raw, recs))
    assert rc == 0 and np.array_equal(out, raw)
    ctx.close()


def test_config5_decode_batch_of_oracle_archive_blocks(F):
    """configs[4] in small: blocks coded by the ORACLE (the reference binary cannot be built) are
    decoded by one GPU launch over the whole batch; every block restores byte-equal."""
    raw, recs = _synth(F, 4, 6 << 20, seed=5)
    _, _, sft, qft = O.freq_tables(raw, recs)
    octx = O.OracleCtx(sft, qft)
    ctx = F.Context(sft, qft)
    # cut into 12 blocks at record boundaries
    cuts = np.linspace(0, len(recs), 13).astype(int)
    blocks, originals = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        lo = 0 if a == 0 else int(recs[a - 1]["qual_off"] + recs[a - 1]["len"] + 1)
        hi = int(recs[b - 1]["qual_off"] + recs[b - 1]["len"] + 1)
        braw = raw[lo:hi]
        brecs = recs[a:b].copy()
        brecs["seq_off"] -= lo
        brecs["qual_off"] -= lo
        e = octx.encode(braw, brecs)
        assert e["rc"] == 0
        db = ctx.dblock(O.blank_skeleton(braw, brecs), brecs)
        db.load_streams(e["seq"], e["qual"], e["n_count"], e["n_pos"])
        blocks.append(db)
        originals.append(braw)
    ctx.decode_dblocks(blocks)
    ctx.sync()
    for db, braw in zip(blocks, originals):
        rc, _ = db.status()
        assert rc == 0
        assert np.array_equal(db.fetch_raw(), braw)
        db.close()
    ctx.close()


# ---------------------------------------------------------------- error behaviour
def test_errors(F):
    raw, recs = _synth(F, 2, 1 << 20)
    sft, qft = F.freq_tables(raw, recs)
    ctx = F.Context(sft, qft)
    # capacity rule violated -> explicit error (the reference returns size 0 silently)
    assert ctx.encode_block(raw, recs, seq_cap=4096)["rc"] == -1
    assert ctx.encode_block(raw, recs, qual_cap=8192)["rc"] == -1
    # read shorter than 3
    short = np.frombuffer(b"@r\nAC\n+\nII\n@s\nACGT\n+\nIIII\n", dtype=np.uint8)
    assert ctx.encode_block(short, F.parse_fastq(short))["rc"] == -2
    # quality above Q63 ('a' = 97 > 96): calculateFreqTable throws in the reference
    bad = np.frombuffer(b"@r\nACGTA\n+\nIIaII\n", dtype=np.uint8)
    with pytest.raises(F.FqgpuError) as ei:
        F.freq_tables(bad, F.parse_fastq(bad))
    assert ei.value.code == -4
    assert ctx.encode_block(bad, F.parse_fastq(bad))["rc"] == -4
    # bytes that are neither a base nor N: base2bits_arr has UINT_MAX there (src/fse_sequence.cpp:6-14);
    # coding them as 'A' would lose data silently -> refused by the analysis, the encoder and the oracle
    for seq in (b"ACgTA", b"ACRTA", b"AC.TA", b"ACGT\r", b"\x00CGTA"):
        badb = np.frombuffer(b"@r\n" + seq + b"\n+\nIIIII\n@s\nACGTN\n+\nIIII#\n", dtype=np.uint8)
        br = F.parse_fastq(badb)
        assert len(br) == 2
        with pytest.raises(F.FqgpuError) as ei:
            F.freq_tables(badb, br)
        assert ei.value.code == -4
        assert ctx.encode_block(badb, br)["rc"] == -4
        assert O.OracleCtx(sft, qft).encode(badb, br)["rc"] == -4
    okb = np.frombuffer(b"@r\nACNTA\n+\nIIIII\n", dtype=np.uint8)
    assert ctx.encode_block(okb, F.parse_fastq(okb))["rc"] == 0
    # record table pointing outside the block
    r2 = recs[:10].copy()
    r2["qual_off"][3] = raw.size
    assert ctx.encode_block(raw, r2)["rc"] == -4
    # corrupt streams: end mark gone, truncated, bit flipped near the end
    g = ctx.encode_block(raw, recs)
    skel = O.blank_skeleton(raw, recs)
    z = g["seq"].copy(); z[-1] = 0
    assert ctx.decode_block(z, g["qual"], g["n_count"], g["n_pos"], recs, skel)[0] == -3
    assert ctx.decode_block(g["seq"][:-5], g["qual"], g["n_count"], g["n_pos"], recs, skel)[0] == -3
    # damaged bits (end mark moved down by one; a payload bit flipped in the middle of either
    # stream): the verdict is the oracle's -- refused, or accepted with exactly the oracle's bytes
    octx = O.OracleCtx(sft, qft)
    q1 = g["qual"].copy(); top = int(q1[-1]).bit_length() - 1
    q1[-1] = (int(q1[-1]) & ~(1 << top)) | (1 << (top - 1)) if top > 0 else 0
    q2 = g["qual"].copy(); q2[q2.size // 2] ^= 0x10
    s2 = g["seq"].copy(); s2[s2.size // 3] ^= 0x04
    n_bad = 0
    for sq, ql in ((g["seq"], q1), (g["seq"], q2), (s2, g["qual"])):
        orc, oout = octx.decode(sq, ql, g["n_count"], g["n_pos"], recs, skel)
        rc, out = ctx.decode_block(sq, ql, g["n_count"], g["n_pos"], recs, skel)
        assert (rc == 0) == (orc == 0), (rc, orc)
        if rc == 0:
            assert np.array_equal(out, oout) and not np.array_equal(out, raw)
        else:
            assert rc == -3
            n_bad += 1
    assert n_bad >= 1  # a stream that is one bit short cannot be consumed exactly
    rc, out = ctx.decode_block(g["seq"], g["qual"], g["n_count"], g["n_pos"], recs, skel)
    assert rc == 0 and np.array_equal(out, raw)
    # tables that do not sum to 2^log are refused
    broken = sft.copy(); broken["norm"][0][7][0] += 1
    with pytest.raises(F.FqgpuError):
        F.Context(broken, qft)
    ctx.close()


def test_results_without_an_explicit_sync(F):
    """fqgpu_dblock_status / fetch / dblocks_decode right behind an asynchronous encode: the lanes run
    on non-blocking streams, so these calls wait for the block themselves (they used to read zero sizes)."""
    raw, recs = _synth(F, 2, 8 << 20)
    _, _, sft, qft = O.freq_tables(raw, recs)
    e = O.OracleCtx(sft, qft).encode(raw, recs)
    ctx = F.Context(sft, qft)
    b = ctx.dblock(raw, recs)
    b.encode()
    rc, st = b.status()          # no ctx.sync() in between
    assert rc == 0 and st["seq_len"] == len(e["seq"]) and st["qual_len"] == len(e["qual"])
    b2 = ctx.dblock(raw, recs)
    b2.encode()
    g = b2.fetch()               # status + fetch, nothing synchronised by the caller
    assert np.array_equal(g["seq"], e["seq"]) and np.array_equal(g["qual"], e["qual"])
    b3 = ctx.dblock(raw, recs)
    b3.encode()
    ctx.decode_dblocks([b3])     # encode -> decode back to back
    rc, _ = b3.status()
    assert rc == 0 and np.array_equal(b3.fetch_raw(), raw)
    # a block whose streams were replaced can be re-encoded against its own capacity rule
    b3.load_streams(np.concatenate([e["seq"], np.zeros(1 << 20, np.uint8)])[: len(e["seq"])], e["qual"], e["n_count"], e["n_pos"])
    b3.encode()
    g3 = b3.fetch()
    assert np.array_equal(g3["seq"], e["seq"])
    for x in (b, b2, b3):
        x.close()
    ctx.close()


def test_accumulated_n_buffers_decode_from_the_end(F, golden_dir):
    """a reused CompressedBuffersDst carries the n_count/n_pos of earlier blocks in
    front; the decoder pops from the END, so extra leading entries must not matter."""
    raw, recs = O.load_fastq(os.path.join(golden_dir, "SRR065390_sub_1.fastq"))
    _, _, sft, qft = O.freq_tables(raw, recs)
    e = O.OracleCtx(sft, qft).encode(raw, recs)
    ctx = F.Context(sft, qft)
    junk_c = np.arange(37, dtype=np.uint16)
    junk_p = np.arange(911, dtype=np.uint16)
    rc, out = ctx.decode_block(e["seq"], e["qual"], np.concatenate([junk_c, e["n_count"]]),
                               np.concatenate([junk_p, e["n_pos"]]), recs, O.blank_skeleton(raw, recs))
    assert rc == 0 and np.array_equal(out, raw)
    ctx.close()


# ---------------------------------------------------------------- BASELINE size: properties only
def test_full_size_roundtrip_1gib(F):
    """configs[2] layout: 1 GiB of config-2 reads in 16 blocks of 64 MiB, tables from the first
    128 MiB.  Size-independent properties: encode -> wipe -> batch decode restores every byte; the
    streams of two blocks are additionally compared with the oracle bit for bit."""
    n_blocks, bsz = 16, 64 << 20
    blocks, raws, recss = [], [], []
    next_id = 0
    for i in range(n_blocks):
        raw, n = F.synth_fastq(bsz, 2, seed=28, first_read_id=next_id)
        next_id += n
        raws.append(raw)
        recss.append(F.parse_fastq(raw))
    sample = np.concatenate(raws[:2])
    srecs = F.parse_fastq(sample)
    sft, qft = F.freq_tables(sample, srecs)
    ctx = F.Context(sft, qft)
    total = 0
    for raw, recs in zip(raws, recss):
        b = ctx.dblock(raw, recs)
        b.encode()
        blocks.append(b)
        total += raw.size
    ctx.sync()
    sizes = []
    for b in blocks:
        rc, st = b.status()
        assert rc == 0
        ls, lq = b.longest_chain()
        # sequence chains are serial per context (~1/256 of the bases); quality chains are cut at
        # single-state symbols, so no lane walks more than a small part of the hottest context
        # (context 0xD7 also receives the first base of every read)
        assert ls <= st["n_bases"] // 200 + len(recss[0]) + 4096 and lq < st["n_bases"] // 512
        sizes.append((st["seq_len"], st["qual_len"]))
    # uniform ACGT cannot beat 2 bits/base; the model mismatch costs well under 0.1 %
    bases = sum(int(r["len"].sum()) for r in recss)
    assert bases / 4 <= sum(s for s, _ in sizes) <= bases / 4 * 1.001 + 16 * 2048
    octx = O.OracleCtx(sft, qft)
    for i in (0, 11):
        e = octx.encode(raws[i], recss[i])
        g = blocks[i].fetch()
        assert np.array_equal(g["seq"], e["seq"]) and np.array_equal(g["qual"], e["qual"])
    for b in blocks:
        b.wipe()
    ctx.sync()
    assert not np.array_equal(blocks[3].fetch_raw(), raws[3])
    ctx.decode_dblocks(blocks)
    ctx.sync()
    for b, raw in zip(blocks, raws):
        rc, _ = b.status()
        assert rc == 0
        assert np.array_equal(b.fetch_raw(), raw)
        b.close()
    ctx.close()
    assert total > (1 << 30) - 16 * 400


def _big_block_vs_oracle(F, mib):
    """one block of `mib` MiB of config-2 reads, tables from its first 128 MiB (reference -S 128):
    byte-compared with the oracle, or refused by both under the capacity rule (src/workspace.h:21-35)"""
    raw, _ = F.synth_fastq(mib << 20, 2, seed=28)
    recs = F.parse_fastq(raw)
    srecs = recs[recs["qual_off"] + recs["len"] < (128 << 20)]
    sft, qft = F.freq_tables(raw, srecs)
    ctx = F.Context(sft, qft)
    ctx.set_lanes(1)
    b = ctx.dblock(raw, recs)
    b.encode()
    rc, st = b.status()
    octx = O.OracleCtx(sft, qft)
    e = octx.encode(raw, recs)
    assert rc == e["rc"], (rc, e["rc"])
    if rc == 0:
        g = b.fetch()
        for k in ("seq", "qual", "readlens", "n_count", "n_pos"):
            assert np.array_equal(g[k], e[k]), k
        b.wipe()
        ctx.decode_dblocks([b])
        assert b.status()[0] == 0 and np.array_equal(b.fetch_raw(), raw)
    else:
        assert rc == -1
    b.close()
    ctx.close()
    return rc, st


@pytest.mark.timeout(900)
def test_one_256mib_block_bench_layout_matches_oracle(F):
    """the block size bench.py times (-R 256: 120 M symbols per stream), against the oracle bit for bit"""
    rc, st = _big_block_vs_oracle(F, 256)
    assert rc == 0 and st["n_bases"] > 119_000_000


@pytest.mark.timeout(1200)
def test_single_768mib_block_360M_symbols_under_the_plain_capacity_rule(F):
    """The largest single block of config-2 reads that still FITS the reference's capacity rule with tables from
    its first 128 MiB (src/workspace.h:21-35: the sequence stream ends 47 bytes below n/4 + 1024): 360 M symbols per
    stream -- three times the bench's block -- all five streams byte for byte, then the round trip."""
    rc, st = _big_block_vs_oracle(F, 768)
    assert rc == 0 and st["n_bases"] > 360_000_000
    assert F.bound_seq(st["n_bases"]) - st["seq_len"] < 1024


@pytest.mark.timeout(1200)
def test_single_1gib_block_is_refused_under_the_plain_capacity_rule_by_both_coders(F):
    """configs[1] read literally ("single block stream", -R 1024) with tables from the first 128 MiB: the sequence
    stream needs 239 bytes more than n/4 + 1024, the reference's endChunk returns 0
    (src/fse_common.hpp:85-90) -- the oracle and the GPU coder both say FQGPU_E_OVERFLOW.  Nothing is compared here
    but the verdict; the bytes of a block of this size are compared in the next test."""
    rc, st = _big_block_vs_oracle(F, 1024)
    assert rc == -1 and st["n_bases"] > 479_000_000


@pytest.mark.timeout(1200)
def test_single_block_of_480M_symbols_with_room_matches_oracle_byte_for_byte(F):
    """The same 1 GiB block (480 M symbols per stream: u32 encode indices up to 4.8e8, 14 659 tiles, 117 K segments
    per stream, three-level resolves) with 64 KiB of room handed to BOTH coders (fqgpu_encode_block judges the
    overflow rule against the caller's capacity, like the oracle): rc == 0, every stream byte for byte; then the
    GPU decodes the ORACLE's streams back to the block."""
    raw, _ = F.synth_fastq(1024 << 20, 2, seed=28)
    recs = F.parse_fastq(raw)
    srecs = recs[recs["qual_off"] + recs["len"] < (128 << 20)]
    sft, qft = F.freq_tables(raw, srecs)
    n_bases = int(recs["len"].sum())
    assert n_bases > 480_000_000
    seq_cap = F.bound_seq(n_bases) + (64 << 10)
    octx = O.OracleCtx(sft, qft)
    e = octx.encode(raw, recs, seq_cap=seq_cap)
    octx.close()
    assert e["rc"] == 0 and e["seq"].size > F.bound_seq(n_bases)   # it does NOT fit the plain rule
    ctx = F.Context(sft, qft)
    ctx.set_lanes(1)
    g = ctx.encode_block(raw, recs, seq_cap=seq_cap)
    assert g["rc"] == 0
    for k in ("seq", "qual", "readlens", "n_count", "n_pos"):
        assert np.array_equal(g[k], e[k]), k
    del g
    rc, out = ctx.decode_block(e["seq"], e["qual"], e["n_count"], e["n_pos"], recs, O.blank_skeleton(raw, recs))
    assert rc == 0 and np.array_equal(out, raw)
    ctx.close()


# ---------------------------------------------------------------- extension: decode index
@pytest.mark.parametrize("mode", [2, 4, -2])
def test_decode_index_parallel_decode(F, mode):
    """FQGPU_F_DECODE_INDEX (extension): the streams stay the oracle's byte for
    byte; with the index the block decodes with one lane per (stream, stride) -- mid-record starts,
    N patching and mixed read lengths included -- and the index survives a trip through the host."""
    raw, recs = _synth(F, abs(mode), 14 << 20)
    _, _, sft, qft = O.freq_tables(raw, recs)
    e = O.OracleCtx(sft, qft).encode(raw, recs)
    ctx = F.Context(sft, qft)
    if mode < 0:  # the sequence stream through the generic (reset-aware) chain kernels
        ctx.set_chain_params(1024, seq_generic=True)
    ctx.set_index_stride(1 << 16)   # ~100 strides of 64 Ki symbols
    b = ctx.dblock(raw, recs)
    b.encode(flags=F.F_DECODE_INDEX)
    ctx.sync()
    g = b.fetch()
    for k in ("seq", "qual", "n_count", "n_pos"):
        assert np.array_equal(g[k], e[k]), k
    ix = [b.fetch_index(0), b.fetch_index(1)]
    n_sym = int(recs["len"].sum())
    n_snap = (n_sym - 1) // (1 << 16)
    assert ix[0].size == 32 + n_snap * (16 + 2 * 256) and ix[1].size == 32 + n_snap * (16 + 2 * 8192)
    # bit positions grow, the last one lies inside the stream
    bp = np.array([int(ix[1][32 + k * (16 + 2 * 8192):][:8].view(np.uint64)[0]) for k in range(n_snap)])
    assert np.all(np.diff(bp) > 0) and bp[-1] < 8 * g["qual"].size
    b.wipe()
    ctx.decode_dblocks([b])
    ctx.sync()
    assert b.status()[0] == 0 and np.array_equal(b.fetch_raw(), raw)
    b.close()
    # a fresh block: streams and indexes come from the host
    b2 = ctx.dblock(O.blank_skeleton(raw, recs), recs)
    b2.load_streams(g["seq"], g["qual"], g["n_count"], g["n_pos"])
    assert b2.load_index(0, ix[0]) == 0 and b2.load_index(1, ix[1]) == 0
    ctx.decode_dblocks([b2])
    ctx.sync()
    assert b2.status()[0] == 0 and np.array_equal(b2.fetch_raw(), raw)
    # a damaged index is refused, a missing one just means the one-lane-per-stream decoder
    bad = ix[1].copy(); bad[0] ^= 1
    assert b2.load_index(1, bad) != 0
    assert b2.load_index(1, ix[1][:-2]) != 0
    assert b2.load_index(1, np.zeros(0, dtype=np.uint8)) == 0
    b2.wipe()
    ctx.decode_dblocks([b2])
    ctx.sync()
    assert b2.status()[0] == 0 and np.array_equal(b2.fetch_raw(), raw)
    b2.close()
    ctx.close()


def test_decode_batches_with_more_chains_than_places(F):
    """More than two quality chains per CU: the two streams go into separate launches and the quality
    walk takes its compact form (entries alone in LDS, table offsets by scalar loads).  (a) one block
    with a decode index every 4096 symbols (mixed read lengths and N's: ~600 strides per stream);
    (b) 600 small blocks in one batch.  Both restore byte for byte."""
    raw, recs = _synth(F, 4, 6 << 20, seed=9)
    _, _, sft, qft = O.freq_tables(raw, recs)
    ctx = F.Context(sft, qft)
    ctx.set_index_stride(4096)
    b = ctx.dblock(raw, recs)
    b.encode(flags=F.F_DECODE_INDEX)
    ctx.sync()
    e = O.OracleCtx(sft, qft).encode(raw, recs)
    g = b.fetch()
    for k in ("seq", "qual", "n_count", "n_pos"):
        assert np.array_equal(g[k], e[k]), k
    assert (int(recs["len"].sum()) - 1) // 4096 > 520
    b.wipe()
    ctx.decode_dblocks([b])
    ctx.sync()
    assert b.status()[0] == 0 and np.array_equal(b.fetch_raw(), raw)
    b.close()
    # (b) 600 blocks of ~40 records
    cuts = np.linspace(0, len(recs), 601).astype(int)
    blocks, originals = [], []
    for a, z in zip(cuts[:-1], cuts[1:]):
        lo = 0 if a == 0 else int(recs[a - 1]["qual_off"] + recs[a - 1]["len"] + 1)
        hi = int(recs[z - 1]["qual_off"] + recs[z - 1]["len"] + 1)
        braw = raw[lo:hi]
        brecs = recs[a:z].copy()
        brecs["seq_off"] -= lo
        brecs["qual_off"] -= lo
        db = ctx.dblock(braw, brecs)
        db.encode()
        blocks.append(db)
        originals.append(braw)
    ctx.sync()
    for db in blocks:
        db.wipe()
    ctx.decode_dblocks(blocks)
    ctx.sync()
    for db, braw in zip(blocks, originals):
        assert db.status()[0] == 0
        assert np.array_equal(db.fetch_raw(), braw)
        db.close()
    ctx.close()


def test_page_locked_blocks_are_cached_on_free(F):
    """fqgpu_host_free keeps page-locked blocks (hipHostFree waits until the device is idle: a farm
    worker freeing a buffer would wait for every other worker's kernels): the next allocation of the
    size class gets the block back; small blocks are ordinary heap memory; fqgpu_host_trim empties
    the cache."""
    from fqcomp28_amd.binding import lib
    L = lib()
    L.fqgpu_host_trim()
    p = L.fqgpu_host_alloc((3 << 20) - 100)
    assert p
    L.fqgpu_host_free(p)
    q = L.fqgpu_host_alloc((3 << 20) - 4096)   # same class (sizes round up to a multiple of 256 KiB here)
    assert q == p
    L.fqgpu_host_free(q)
    assert L.fqgpu_host_trim() >= 3 << 20
    assert L.fqgpu_host_trim() == 0
    small = L.fqgpu_host_alloc(100)
    assert small
    L.fqgpu_host_free(small)
    assert L.fqgpu_host_trim() == 0   # not pinned, not cached


def test_handles_created_and_destroyed_in_a_loop_give_all_device_memory_back(F):
    """Advisor finding of round 3: the device parser's scratch (fqgpu_ctx::hp_parse, grown by fqgpu_ctx_reserve and
    by every fqgpu_encode_begin without a record table -- the farm's default path) was never released.  Handles that
    code an unparsed chunk are created and destroyed in a loop; the free device memory must not drift."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    def free_bytes():
        f, t = C.c_size_t(0), C.c_size_t(0)
        assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value
    raw, recs = _synth(F, 2, 24 << 20)
    _, _, sft, qft = O.freq_tables(raw, recs)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import headers_oracle as HO
    first = raw.tobytes()[: int(recs[0]["seq_off"]) - 1]
    types, seps = HO.format_from_header(first)
    fmt = ([0 if t == HO.NUMERIC else 1 for t in types], bytes(seps), first)
    def cycle():
        ctx = F.Context(sft, qft)
        ctx.set_lanes(1)
        # fqgpu_encode_begin(recs = NULL): table built on the device; the header fields coded there too; a decode index
        g = ctx.encode_raw(raw, flags=F.F_DECODE_INDEX, header_format=fmt)
        assert g["rc"] == 0 and g["headers_rc"] == 0 and len(g["recs"]) == len(recs) and g["index"][1].size > 32
        ctx.close()
    cycle()   # first use: the runtime's own pools, code objects, the pinned cache
    F.lib().fqgpu_host_trim()
    before = free_bytes()
    for _ in range(6):
        cycle()
    F.lib().fqgpu_host_trim()
    after = free_bytes()
    # the parser's scratch for a 24 MiB chunk is ~1.5 MB per handle: six leaked ones would be ~9 MB
    assert before - after < (2 << 20), (before, after)


# ---------------------------------------------------------------- the C++ drop-in shim
def test_cpp_workspace_shim_roundtrip(golden_dir):
    """fqcomp28_amd/csrc/workspace.hpp (the reference's Workspace/CompressedBuffers surface over the
    C ABI): the reference's own Workspace::encodeChunk round-trip test, restated in C++."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "workspace_test")
    subprocess.run(["g++", "-std=c++17", "-O2", "-o", exe, os.path.join(root, "tests", "cpp", "workspace_test.cpp"),
                    "-L" + os.path.join(root, "fqcomp28_amd"), "-lfqgpu",
                    "-Wl,-rpath," + os.path.join(root, "fqcomp28_amd"), "-lpthread"], check=True)
    import tempfile
    args = []
    with tempfile.TemporaryDirectory() as tmp:
        for f in FIXTURES:  # the oracle's streams of every fixture, for the byte comparison inside the C++ test
            path = os.path.join(golden_dir, f + ".fastq")
            raw, recs = O.load_fastq(path)
            _, _, sft, qft = O.freq_tables(raw, recs)
            e = O.OracleCtx(sft, qft).encode(raw, recs)
            assert e["rc"] == 0
            prefix = os.path.join(tmp, f)
            for k in ("seq", "qual", "readlens", "n_count", "n_pos", "raw_after"):
                e[k].tofile(prefix + "." + k)
            args += [path, prefix]
        out = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok ") == len(FIXTURES)


# ---------------------------------------------------------------- next row: GPU FASTQ record parser
def test_gpu_parser_matches_cpu_parser(F, golden_dir):
    """fqgpu_dblock_create_from_raw (newline scan + 4-line grouping on the GPU) against the host
    parser and the numpy restatement of FastqReader::parseRecords (src/fastq_io.cpp:67-125)."""
    raw0, recs0 = O.load_fastq(os.path.join(golden_dir, "SRR065390_sub_1.fastq"))
    _, _, sft, qft = O.freq_tables(raw0, recs0)
    ctx = F.Context(sft, qft)
    cases = [raw0, raw0[: raw0.size - 23]]  # whole block; block with a partial record at the end
    for mode in (2, 4):
        cases.append(F.synth_fastq(3 << 20, mode)[0])
    for raw in cases:
        want = F.parse_fastq(raw)
        b = ctx.dblock(raw)  # no record table handed over
        got = b.records()
        assert np.array_equal(got, want)
        assert b.raw_len == int(want[-1]["qual_off"] + want[-1]["len"] + 1)
        b.close()
    # an unparsed block encodes to the same streams as the parsed one
    e = O.OracleCtx(sft, qft).encode(raw0, recs0)
    b = ctx.dblock(raw0)
    b.encode()
    ctx.sync()
    g = b.fetch()
    for k in ("seq", "qual", "readlens", "n_count", "n_pos"):
        assert np.array_equal(g[k], e[k]), k
    b.close()
    # malformed input is refused
    for bad in (b"@r\nACGT\n-\nIIII\n", b"@r\nACGT\n+\nIII\n", b"r\nACGT\n+\nIIII\n", b"@r\nAC\n+\nII\n", b"no newline"):
        with pytest.raises(F.FqgpuError) as ei:
            ctx.dblock(np.frombuffer(bad, dtype=np.uint8))
        assert ei.value.code in (-4, -2)
    ctx.close()


def test_soak_sample_random_sizes_kinds_tables_parameters_and_call_paths():
    """Forty cases of tools/soak_roundtrip.py (the sweep that found round 3's resolve bug runs thousands): random block
    sizes, data kinds with runs and alternations written into the qualities, own or foreign tables, chain parameters
    off their defaults, decode index, device-resident / host-pointer / unparsed-chunk calls, several blocks in flight
    -- every case byte for byte against the oracle, both directions."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_roundtrip.py"), "40", "424242"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "soak: 40 cases" in r.stdout
    # ... and two cases with blocks of 64 to 96 MiB (round 3's sample stopped at 3 MiB)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_roundtrip.py"), "2", "515151", "96", "64"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "soak: 2 cases" in r.stdout
