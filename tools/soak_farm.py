"""Soak of the C++ block farm end to end (tools/fqc_tool.cpp over process.hpp / archive.hpp / workspace.hpp):
random FASTQ files (synthetic kinds, sizes from a few records to tens of MiB, real-looking headers from the
generator, some files without a final newline), random -R / -S / -t / --accumulate-n / -d 0,0; compress,
decompress with ANOTHER number of workers, compare the round trip byte for byte; every block of the archive is
read back by the independent Python reader of the format (oracle/fqc_archive.py) and its seq / qual streams are
compared with the CPU oracle's for the same block and tables.  Two files in three get their headers rewritten in a
random SHAPE (one to a dozen fields; names that change, numbers that walk, jump, go negative; Illumina-like; one bare
number; one bare string) and every header field stream of every block is compared with oracle/headers_oracle.py --
the header fields are coded on the GPU (headers.hip).  One file in twenty has ONE header that cannot be coded: the
command must fail.  One in three is compressed with --index (decode indexes beside the archive, used by the restore).
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
import headers_oracle as HO  # noqa: E402
import oracle_lib as O  # noqa: E402


def rewrite_headers(raw, rng):
    """-> (bytes of the file with new headers, index of a record whose header cannot be coded or None)"""
    recs = F.parse_fastq(raw)
    b = raw.tobytes()
    shape = int(rng.integers(0, 6))
    names = [b"EAS%d" % int(rng.integers(1, 999)) for _ in range(4)] + [b"x" * int(rng.integers(1, 200)), b"HWUSI-EAS100R"]
    # (a '-' is a separator wherever the dataset's FIRST header has one, so numbers may only go negative behind it)
    bad = int(rng.integers(1, len(recs))) if len(recs) > 2 and rng.random() < 0.05 and shape not in (3, 4) else None
    out = []
    name, x, y, tile = names[0], 1000, 2000, 1
    for i, r in enumerate(recs):
        if rng.random() < 0.02:
            name = names[int(rng.integers(len(names)))]
        x += int(rng.integers(-50, 200)); y = int(rng.integers(0, 200000)); tile += int(rng.random() < 0.01)
        num = b"%d" % (i + 1) if bad != i else b"x%d" % i
        if shape == 0:
            h = b"@%s.%s %d length=%d" % (name, num, int(rng.integers(-2**31 if i else 0, 2**31)), int(r["len"]))
        elif shape == 1:
            h = b"@%s:%d:FC%d:%d:%d:%d:%d %d:N:0:%s" % (name, 7, 42, 1 + tile % 8, tile, x, y, 1 + i % 2, num)
        elif shape == 2:
            h = b"@%s" % num
        elif shape == 3:
            h = b"@r%s/%d" % (num, 1 + i % 2)
        elif shape == 4:
            h = b"@" + bytes(rng.integers(97, 123, int(rng.integers(1, 60)), dtype=np.uint8))
        else:
            h = b"@%s_%s-%d.%d|%d;%s,%d" % (name, num, abs(x), y, tile, names[i % 2], i * 40000000 % 2**31)
        s, q, n = int(r["seq_off"]), int(r["qual_off"]), int(r["len"])
        out.append(h + b"\n" + b[s: s + n] + b"\n+\n" + b[q: q + n] + b"\n")
    return b"".join(out), bad


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    exe = os.path.join(ROOT, "tools", "_build", "fqc_tool")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run(["g++", "-std=c++17", "-O2", "-o", exe, os.path.join(ROOT, "tools", "fqc_tool.cpp"), "-L" + os.path.join(ROOT, "fqcomp28_amd"),
                    "-lfqgpu", "-Wl,-rpath," + os.path.join(ROOT, "fqcomp28_amd"), "-lpthread"], check=True)
    t0 = time.time()
    checked_blocks = refused = bad_headers = checked_fields = 0
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        for case in range(cases):
            rng = np.random.default_rng(seed0 + case)
            mode = int(rng.choice([2, 2, 3, 4, 4, 5]))
            size = int(rng.choice([2000, 50000, 1 << 20, 5 << 20, 24 << 20]) * (0.5 + rng.random()))
            raw, _ = F.synth_fastq(size, mode, seed=seed0 + case)
            data = raw.tobytes()
            bad_header = None
            if rng.random() < 0.66 and size < (8 << 20):
                data, bad_header = rewrite_headers(raw, rng)
                raw = np.frombuffer(data, dtype=np.uint8)
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
            if rng.random() < 0.33:
                args += ["--index"]
            arc, back = os.path.join(tmp, "a.fqc"), os.path.join(tmp, "back.fastq")
            c = subprocess.run([exe, "c", src, arc] + args, capture_output=True, text=True)
            if len(expect) == 0:
                assert c.returncode != 0 or os.path.getsize(arc) > 0, (case, "empty input")
                continue
            if bad_header is not None and not (cut_newline and bad_header >= len(F.parse_fastq(raw)) - 1):
                if c.returncode != 0 and "capacity bound" in c.stderr:
                    continue
                assert c.returncode == 1 and "not an int32" in c.stderr, (case, args, "a header that cannot be coded", c.stdout[-300:], c.stderr[-300:])
                bad_headers += 1
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
                first_header, seq_ft, qual_ft, blocks, _ = A.read_archive(arc)
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
                    # the header fields (coded on the GPU) against the restatement of the reference's coder
                    types, _, streams = HO.encode_headers(A.headers_of(braw, brecs), first_header)
                    for t, parts, s in zip(types, blk.fields, streams):
                        wants = [bytes(s.flags), bytes(s.content), bytes(s.lengths)] if t == HO.STRING else [bytes(s.content)]
                        for (orig, cbytes), want in zip(parts, wants):
                            got_f = F.memdecompress(np.frombuffer(cbytes, dtype=np.uint8), orig).tobytes()
                            assert orig == len(want) and got_f == want, (case, args, "header field of block", blk.idx)
                            checked_fields += 1
                    checked_blocks += 1
                assert pos == len(expect)
                octx.close()
            if case % 10 == 9:
                print("case %d of %d, %d archive blocks checked against the oracle, %.0f s" % (case + 1, cases, checked_blocks, time.time() - t0), flush=True)
    print("farm soak: %d cases, %d refused by farm and oracle alike (capacity rule), %d refused for a header that cannot be coded, the rest round-tripped; "
          "%d archive blocks equal to the oracle's, %d header field streams equal to the header oracle's" % (cases, refused, bad_headers, checked_blocks, checked_fields))


if __name__ == "__main__":
    main()
