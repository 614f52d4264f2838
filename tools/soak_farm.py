"""Soak of the C++ block farm end to end (tools/fqc_tool.cpp over process.hpp / archive.hpp / workspace.hpp):
random FASTQ files (synthetic kinds, sizes from a few records to tens of MiB, real-looking headers from the
generator, some files without a final newline), random -R / -S / -t / --accumulate-n / -d 0,0; compress,
decompress with ANOTHER number of workers, compare the round trip byte for byte; every block of the archive is
read back by the independent Python reader of the format (oracle/fqc_archive.py) and its seq / qual streams are
compared with the CPU oracle's for the same block and tables.
    python tools/soak_farm.py [cases, default 40] [first seed]"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import fqcomp28_amd as F  # noqa: E402
import fqc_archive as A  # noqa: E402  (test infrastructure)
import oracle_lib as O  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    exe = os.path.join(ROOT, "tools", "_build", "fqc_tool")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run(["g++", "-std=c++17", "-O2", "-o", exe, os.path.join(ROOT, "tools", "fqc_tool.cpp"), "-L" + os.path.join(ROOT, "fqcomp28_amd"),
                    "-lfqgpu", "-Wl,-rpath," + os.path.join(ROOT, "fqcomp28_amd"), "-lpthread"], check=True)
    t0 = time.time()
    checked_blocks = refused = 0
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        for case in range(cases):
            rng = np.random.default_rng(seed0 + case)
            mode = int(rng.choice([2, 2, 3, 4, 4, 5]))
            size = int(rng.choice([2000, 50000, 1 << 20, 5 << 20, 24 << 20]) * (0.5 + rng.random()))
            raw, _ = F.synth_fastq(size, mode, seed=seed0 + case)
            data = raw.tobytes()
            cut_newline = rng.random() < 0.2
            src = os.path.join(tmp, "in.fastq")
            open(src, "wb").write(data[:-1] if cut_newline else data)
            expect = data  # a missing final newline drops the unfinished last record (the reference's reader does the same)
            if cut_newline:
                recs = F.parse_fastq(raw)
                expect = data[: int(recs[-2]["qual_off"] + recs[-2]["len"] + 1)] if len(recs) > 1 else b""
            R = int(rng.choice([1, 2, 4, 16]))
            args = ["-R", str(R), "-S", str(int(rng.choice([1, 4, 16]))), "-t", str(int(rng.choice([1, 2, 3, 5])))]
            if rng.random() < 0.3:
                args += ["--accumulate-n"]
            if rng.random() < 0.3:
                args += ["-d", "0,0"]
            arc, back = os.path.join(tmp, "a.fqc"), os.path.join(tmp, "back.fastq")
            c = subprocess.run([exe, "c", src, arc] + args, capture_output=True, text=True)
            if len(expect) == 0:
                assert c.returncode != 0 or os.path.getsize(arc) > 0, (case, "empty input")
                continue
            if c.returncode != 0 and "capacity bound" in c.stderr:
                # the reference's own rule (a stream longer than its capacity: src/fse_sequence.cpp:35-51 returns 0): the
                # sample's tables do not fit the data.  The oracle must refuse a block of the same file for the same reason.
                full = np.frombuffer(expect, dtype=np.uint8)
                frecs = F.parse_fastq(full)
                ends = frecs["qual_off"].astype(np.int64) + frecs["len"] + 1

                def chunks(limit):
                    out, lo = [], 0
                    while lo < len(full):
                        k = int(np.searchsorted(ends, lo + limit, side="right"))
                        hi = int(ends[k - 1]) if k and ends[k - 1] > lo else int(ends[np.searchsorted(ends, lo, side="right")])
                        out.append((lo, hi))
                        lo = hi
                    return out
                S_mib = int(args[args.index("-S") + 1])
                s_lo, s_hi = chunks(S_mib << 20)[0]
                sraw = full[s_lo:s_hi]
                _, _, sft, qft = O.freq_tables(sraw, F.parse_fastq(sraw))
                octx = O.OracleCtx(sft, qft)
                rcs = []
                for lo, hi in chunks(R << 20):
                    braw = full[lo:hi]
                    rcs.append(octx.encode(braw, F.parse_fastq(braw))["rc"])
                octx.close()
                assert any(rc != 0 for rc in rcs), (case, args, "the farm refused what the oracle codes", rcs)
                refused += 1
                continue
            assert c.returncode == 0, (case, args, c.stdout[-500:], c.stderr[-500:])
            d = subprocess.run([exe, "d", arc, back, "-t", str(int(rng.choice([1, 2, 4])))], capture_output=True, text=True)
            assert d.returncode == 0, (case, args, d.stdout[-500:], d.stderr[-500:])
            got = open(back, "rb").read()
            assert got == expect, (case, args, len(got), len(expect))
            # the archive, read by the independent reader: every block's streams against the oracle
            if "--accumulate-n" not in args and len(expect) < (8 << 20):
                _, seq_ft, qual_ft, blocks, _ = A.read_archive(arc)
                sft = np.frombuffer(seq_ft, dtype=F.binding.SEQ_FT_DTYPE)
                qft = np.frombuffer(qual_ft, dtype=F.binding.QUAL_FT_DTYPE)
                octx = O.OracleCtx(sft, qft)
                full = np.frombuffer(expect, dtype=np.uint8)
                pos = 0
                for blk in blocks:  # sorted by chunk index
                    braw = full[pos: pos + blk.total]
                    pos += blk.total
                    brecs = F.parse_fastq(braw)
                    assert len(brecs) == blk.n_records, (case, args, "records of block", blk.idx)
                    e = octx.encode(braw, brecs)
                    assert e["rc"] == 0 and bytes(e["seq"]) == blk.seq and bytes(e["qual"]) == blk.qual, (case, args, "block", blk.idx)
                    checked_blocks += 1
                assert pos == len(expect)
                octx.close()
            if case % 10 == 9:
                print("case %d of %d, %d archive blocks checked against the oracle, %.0f s" % (case + 1, cases, checked_blocks, time.time() - t0), flush=True)
    print("farm soak: %d cases, %d refused by farm and oracle alike (capacity rule), the rest round-tripped; %d archive blocks equal to the oracle's" % (cases, refused, checked_blocks))


if __name__ == "__main__":
    main()
