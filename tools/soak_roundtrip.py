"""Soak of the whole hot path on the GPU against the CPU oracle: many seeds x data kinds x odd sizes.
Per case: synthetic reads (fqgpu_synth_fastq modes 2..5, some with their qualities rewritten into runs and
alternations -- the contexts that come back at distance 0, 1, 2 are the decode walk's slow paths), tables from the
block itself or from another block (foreign tables: escapes and rare symbols), encode -> five streams byte-equal to
the oracle's -> decode of the ORACLE's streams -> raw block byte-equal.  Blocks are decoded alone and in batches
(batches of many small blocks take the walk's compact form).
    python tools/soak_roundtrip.py [cases, default 60] [first seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import fqcomp28_amd as F  # noqa: E402
import oracle_lib as O  # noqa: E402  (the checker: test infrastructure, never the product)


def rewrite_qualities(raw, recs, rng, how):
    raw = raw.copy()
    for r in recs:
        q = raw[int(r["qual_off"]): int(r["qual_off"]) + int(r["len"])]
        n = q.size
        if how == 1:    # runs of random length
            v, i = [], 0
            while i < n:
                k = int(rng.integers(1, 12))
                v += [int(rng.integers(35, 75))] * k
                i += k
            q[:] = np.array(v[:n], dtype=np.uint8)
        elif how == 2:  # a, b, a, b ... (contexts coming back at distance 2)
            a, b = int(rng.integers(35, 75)), int(rng.integers(35, 75))
            q[0::2] = a
            q[1::2] = b
        elif how == 3:  # period three
            for k in range(3):
                q[k::3] = int(rng.integers(35, 75))
    return raw


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    t0 = time.time()
    done = 0
    for case in range(cases):
        rng = np.random.default_rng(seed0 + case)
        mode = int(rng.choice([2, 2, 3, 4, 4, 5]))
        size = int(rng.integers(3000, 3 << 20))
        raw, _ = F.synth_fastq(size, mode, seed=seed0 + case)
        recs = F.parse_fastq(raw)
        how = int(rng.integers(0, 4))
        if how and mode in (2, 4):
            raw = rewrite_qualities(raw, recs, rng, how)
        # tables: own block, or those of a differently seeded block of another mode
        if rng.random() < 0.3:
            raw2, _ = F.synth_fastq(1 << 20, int(rng.choice([2, 3, 4])), seed=seed0 + case + 7777)
            recs2 = F.parse_fastq(raw2)
            _, _, sft, qft = O.freq_tables(raw2, recs2)
        else:
            _, _, sft, qft = O.freq_tables(raw, recs)
        ctx = F.Context(sft, qft)
        ctx.set_lanes(int(rng.integers(1, 5)))
        octx = O.OracleCtx(sft, qft)
        e = octx.encode(raw, recs)
        b = ctx.dblock(raw, recs)
        b.encode()
        ctx.sync()
        rc = b.status()[0]
        assert rc == e["rc"], (case, rc, e["rc"])
        if rc == 0:
            g = b.fetch()
            for k in ("seq", "qual", "readlens", "n_count", "n_pos"):
                assert np.array_equal(g[k], e[k]), (case, mode, how, k)
            b.load_streams(e["seq"], e["qual"], e["n_count"], e["n_pos"])
            b.wipe()
            ctx.decode_dblocks([b])
            ctx.sync()
            assert b.status()[0] == 0 and np.array_equal(b.fetch_raw(), raw), (case, mode, how, "decode")
            done += 1
        b.close()
        octx.close()
        ctx.close()
        if case % 10 == 9:
            print("case %d of %d, %d round trips, %.0f s" % (case + 1, cases, done, time.time() - t0), flush=True)
    print("soak: %d cases, %d encoded and decoded byte-exactly, the rest refused by both coders alike" % (cases, done))


if __name__ == "__main__":
    main()
