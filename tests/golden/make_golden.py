"""Regenerates tests/golden/expected.json and the first5 byte vectors from the CPU oracle.

The reference cannot be built or run here; these vectors are the
oracle's outputs on the reference's own test inputs (test/data/*.fastq, MIT, public
SRA reads).  The seq/qual sizes and sha1[:12] were independently reproduced by the
surveyor's scratch restatement (SURVEY.md 8(c) "Cross-session regression values").
Run:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

FIXTURES = ["SRR065390_sub_1", "without_ns", "SRR065390_sub_2", "SRR065390_1_first5"]


def sha(a):
    return hashlib.sha1(a.tobytes()).hexdigest()


def main():
    exp = {}
    for f in FIXTURES:
        raw, recs = O.load_fastq(os.path.join(HERE, f + ".fastq"))
        sc, qc, sft, qft = O.freq_tables(raw, recs)
        ctx = O.OracleCtx(sft, qft)
        e = ctx.encode(raw, recs)
        assert e["rc"] == 0
        exp[f] = dict(
            n_records=int(len(recs)), n_bases=int(recs["len"].sum()),
            seq_counts_sha1=sha(sc), qual_counts_sha1=sha(qc),
            seq_ft_sha1=sha(sft), qual_ft_sha1=sha(qft),
            seq_max_log=int(sft["max_log"][0]), qual_max_log=int(qft["max_log"][0]),
            seq_len=int(len(e["seq"])), seq_sha1=sha(e["seq"]),
            qual_len=int(len(e["qual"])), qual_sha1=sha(e["qual"]),
            readlens_sha1=sha(e["readlens"]),
            n_count_bytes=int(e["n_count"].nbytes), n_count_sha1=sha(e["n_count"]),
            n_pos_bytes=int(e["n_pos"].nbytes), n_pos_sha1=sha(e["n_pos"]),
        )
        if f == "SRR065390_1_first5":
            e["seq"].tofile(os.path.join(HERE, f + ".seq.bin"))
            e["qual"].tofile(os.path.join(HERE, f + ".qual.bin"))
            e["n_pos"].tofile(os.path.join(HERE, f + ".n_pos.bin"))
            sft.tofile(os.path.join(HERE, f + ".seq_ft.bin"))
    with open(os.path.join(HERE, "expected.json"), "w") as fh:
        json.dump(exp, fh, indent=1, sort_keys=True)
    print(json.dumps(exp, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
