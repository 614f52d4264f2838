"""Oracle model layer: the reference's own hot-path tests restated
(test/fse_sequence_test.cpp:17-50, test/fse_quality_test.cpp:17-49,
test/workspace_test.cpp:45-69 -- all round-trips) plus the committed golden vectors
and the surveyor's independent regression values."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as O

FIXTURES = ["SRR065390_sub_1", "without_ns", "SRR065390_sub_2", "SRR065390_1_first5"]

# SURVEY.md 8(c) "Cross-session regression values": produced by a different restatement
SURVEY_VALUES = {
    "SRR065390_sub_1": (1000, 100000, 23212, "1c402cc74006", 37539, "f9c92501aabd", 2000, 12526),
    "without_ns": (851, 85100, 20237, "7aea2a154594", 35055, "706e1974c915", 1702, 0),
    "SRR065390_sub_2": (1000, 100000, 23608, "f9771b92ad7b", 36950, "b6953e536149", 2000, 3200),
    "SRR065390_1_first5": (5, 500, 363, "311101ad6b9e", 7305, "5da365dfabb0", 10, 702),
}


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def expected(golden_dir):
    with open(os.path.join(golden_dir, "expected.json")) as fh:
        return json.load(fh)


@pytest.mark.parametrize("name", FIXTURES)
def test_fixture_roundtrip_and_golden(name, golden_dir, expected):
    raw, recs = O.load_fastq(os.path.join(golden_dir, name + ".fastq"))
    sc, qc, sft, qft = O.freq_tables(raw, recs)
    ctx = O.OracleCtx(sft, qft)
    e = ctx.encode(raw, recs)
    assert e["rc"] == 0
    x = expected[name]
    assert (len(recs), int(recs["len"].sum())) == (x["n_records"], x["n_bases"])
    assert sha(sc) == x["seq_counts_sha1"] and sha(qc) == x["qual_counts_sha1"]
    assert sha(sft) == x["seq_ft_sha1"] and sha(qft) == x["qual_ft_sha1"]
    assert (len(e["seq"]), sha(e["seq"])) == (x["seq_len"], x["seq_sha1"])
    assert (len(e["qual"]), sha(e["qual"])) == (x["qual_len"], x["qual_sha1"])
    assert sha(e["n_count"]) == x["n_count_sha1"] and sha(e["n_pos"]) == x["n_pos_sha1"]
    assert np.array_equal(e["readlens"], recs["len"].astype(np.uint16))
    sv = SURVEY_VALUES[name]
    assert (len(recs), int(recs["len"].sum()), len(e["seq"]), sha(e["seq"])[:12], len(e["qual"]),
            sha(e["qual"])[:12], e["n_count"].nbytes, e["n_pos"].nbytes) == sv
    # the encoder replaced N by A in place (src/fse_sequence.cpp:44)
    n_before = int((raw == ord("N")).sum())
    assert int((e["raw_after"] != raw).sum()) == e["n_pos"].size <= n_before
    # decode (records last -> first) restores the block byte for byte
    rc, out = ctx.decode(e["seq"], e["qual"], e["n_count"], e["n_pos"], recs, O.blank_skeleton(raw, recs))
    assert rc == 0 and np.array_equal(out, raw)


def test_first5_byte_vectors(golden_dir):
    name = "SRR065390_1_first5"
    raw, recs = O.load_fastq(os.path.join(golden_dir, name + ".fastq"))
    _, _, sft, qft = O.freq_tables(raw, recs)
    e = O.OracleCtx(sft, qft).encode(raw, recs)
    for key, suffix, dt in (("seq", ".seq.bin", np.uint8), ("qual", ".qual.bin", np.uint8),
                            ("n_pos", ".n_pos.bin", np.uint16)):
        want = np.fromfile(os.path.join(golden_dir, name + suffix), dtype=dt)
        assert np.array_equal(e[key], want), key
    assert np.array_equal(sft.view(np.uint8), np.fromfile(os.path.join(golden_dir, name + ".seq_ft.bin"), dtype=np.uint8))


def test_initial_contexts_and_bounds():
    L = O.lib()
    # Workspace::compressBound*, src/workspace.h:21-35
    assert L.fqo_bound_seq(1023) == 262144 and L.fqo_bound_seq(1024) == 1280
    assert L.fqo_bound_qual(100) == 8388608 and L.fqo_bound_qual(16 << 20) == (16 << 20) * 7 // 8 + 1024
    # a single read ACGTA: counts land in ctx 0xD7 first (INITIAL_CONTEXT, src/fse_sequence.h:42-63)
    raw = np.frombuffer(b"@r\nACGTA\n+\nIIIII\n", dtype=np.uint8)
    recs = O.parse_fastq(raw)
    sc, qc, _, _ = O.freq_tables(raw, recs)
    assert sc[0xD7, 0] == 2  # 'A' after virtual T,C,C,T
    assert sc[(0xD7 >> 2) + (0 << 6), 1] == 2  # then C in ctx addSymUpper(0xD7, A)
    assert qc[4096, 40] == 2  # first quality in ctx calcContext(0,0,0) = 1<<12
    assert qc[4096 + 40, 40] == 2  # second: q=40, q1=q2=0 -> eq flag set
    assert qc[(40 << 6) + 40, 40] == 2  # third: q=40, q1=40, q2=0 -> max 40, eq clear
    assert int(sc.sum()) == 256 * 4 + 5 and int(qc.sum()) == 8192 * 64 + 5


def test_n_skips_do_not_advance_histogram_context():
    # the frequency pass skips N without touching ctx
    a = np.frombuffer(b"@r\nACNNGT\n+\nIIIIII\n", dtype=np.uint8)
    b = np.frombuffer(b"@r\nACGT\n+\nIIII\n", dtype=np.uint8)
    sa = O.freq_tables(a, O.parse_fastq(a))[0]
    sb = O.freq_tables(b, O.parse_fastq(b))[0]
    assert np.array_equal(sa, sb)


def test_short_reads_and_overflow_are_errors():
    raw = np.frombuffer(b"@r\nAC\n+\nII\n@s\nACGT\n+\nIIII\n", dtype=np.uint8)
    recs = O.parse_fastq(raw)
    big = np.frombuffer(b"@s\nACGTACGT\n+\nIIIIIIII\n", dtype=np.uint8)
    _, _, sft, qft = O.freq_tables(big, O.parse_fastq(big))
    ctx = O.OracleCtx(sft, qft)
    assert ctx.encode(raw, recs)["rc"] == -2  # FQO_E_SHORT_READ
    # capacity too small -> endChunk()==0 in the reference (src/fse_common.hpp:85-90)
    assert ctx.encode(big, O.parse_fastq(big), seq_cap=64)["rc"] == -1


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_blocks_roundtrip(seed):
    rng = np.random.default_rng(seed)
    parts = []
    for i in range(300):
        L = int(rng.integers(3, 400))
        seq = rng.choice(list(b"ACGTN"), size=L, p=[0.24, 0.25, 0.25, 0.24, 0.02]).astype(np.uint8)
        q = np.clip(np.rint(rng.normal(30, 8, L)), 0, 63).astype(np.uint8) + 33
        parts.append(b"@r%d\n" % i + seq.tobytes() + b"\n+\n" + q.tobytes() + b"\n")
    raw = np.frombuffer(b"".join(parts), dtype=np.uint8)
    recs = O.parse_fastq(raw)
    _, _, sft, qft = O.freq_tables(raw, recs)
    ctx = O.OracleCtx(sft, qft)
    e = ctx.encode(raw, recs)
    assert e["rc"] == 0
    rc, out = ctx.decode(e["seq"], e["qual"], e["n_count"], e["n_pos"], recs, O.blank_skeleton(raw, recs))
    assert rc == 0 and np.array_equal(out, raw)
    # corrupt stream detection: flipping the end mark byte must not pass BIT_endOfDStream
    bad = e["seq"].copy()
    bad[-1] ^= 0x80 if bad[-1] < 0x80 else 0xC0
    rc2, _ = ctx.decode(bad, e["qual"], e["n_count"], e["n_pos"], recs, O.blank_skeleton(raw, recs))
    assert rc2 != 0


def test_oracle_refuses_bytes_that_are_not_bases(golden_dir):
    """base2bits_arr (src/fse_sequence.cpp:6-14) knows A, C, G, T only (N is replaced first): the
    oracle refuses everything else instead of coding it as 'A'."""
    raw, recs = O.load_fastq(os.path.join(golden_dir, "without_ns.fastq"))
    _, _, sft, qft = O.freq_tables(raw, recs)
    octx = O.OracleCtx(sft, qft)
    assert octx.encode(raw, recs)["rc"] == 0
    for ch in (b"a", b"R", b".", b"\r"):
        bad = raw.copy()
        bad[recs[3]["seq_off"] + 7] = ch[0]
        assert octx.encode(bad, recs)["rc"] == -4
        sc = np.zeros((256, 4), dtype=np.uint32)
        assert O.lib().fqo_seq_counts(O.ptr(bad), O.ptr(recs), len(recs), O.ptr(sc)) == -4
