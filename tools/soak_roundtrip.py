"""Soak of the whole hot path on the GPU against the CPU oracle: many seeds x data kinds x odd sizes.
Per case: synthetic reads (fqgpu_synth_fastq modes 2..5, some with their qualities rewritten into runs and
alternations -- the contexts that come back at distance 0, 1, 2 are the decode walk's slow paths), tables from the
block itself or from another block (foreign tables: escapes and rare symbols), encode -> five streams byte-equal to
the oracle's -> decode of the ORACLE's streams -> raw block byte-equal.  A third of the cases move the chain stage's
segment lengths and group sizes off their defaults, a third code a decode index and decode through it; the block goes
through the device-resident calls, the host-pointer call or, unparsed, through the GPU's record finder; behind the
host-pointer calls the oracle's streams are damaged (bit flips, truncation, random bytes) and decoded by both.
    python tools/soak_roundtrip.py [cases, default 60] [first seed] [largest block in MiB, default 3] [smallest block in MiB, default: 3000 bytes]"""
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


def several_blocks_in_flight(case, seed):
    """2 .. 6 blocks of different sizes and kinds on ONE handle, encoded without waiting in between (each takes the
    next encode lane), then decoded in one batch: every block against the oracle."""
    rng = np.random.default_rng(seed)
    parts = []
    for k in range(int(rng.integers(2, 7))):
        mode = int(rng.choice([2, 3, 4, 5]))
        raw, _ = F.synth_fastq(int(rng.integers(3000, 2 << 20)), mode, seed=seed + 31 * k)
        parts.append((raw, F.parse_fastq(raw)))
    _, _, sft, qft = O.freq_tables(*parts[int(rng.integers(0, len(parts)))])
    ctx = F.Context(sft, qft)
    ctx.set_lanes(int(rng.integers(1, 9)))
    octx = O.OracleCtx(sft, qft)
    blocks = [ctx.dblock(raw, recs) for raw, recs in parts]
    for b in blocks:
        b.encode()
    ctx.sync()
    good = []
    for b, (raw, recs) in zip(blocks, parts):
        e = octx.encode(raw, recs)
        rc = b.status()[0]
        assert rc == e["rc"], (case, "several", rc, e["rc"])
        if rc == 0:
            g = b.fetch()
            for k in ("seq", "qual", "readlens", "n_count", "n_pos"):
                assert np.array_equal(g[k], e[k]), (case, "several", k)
            b.wipe()
            good.append((b, raw))
    if good:
        ctx.decode_dblocks([b for b, _ in good])
        ctx.sync()
        for b, raw in good:
            assert b.status()[0] == 0 and np.array_equal(b.fetch_raw(), raw), (case, "several", "decode")
    for b in blocks:
        b.close()
    octx.close()
    ctx.close()
    return len(good)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    max_bytes = (int(sys.argv[3]) if len(sys.argv) > 3 else 3) << 20
    min_bytes = (int(sys.argv[4]) << 20) if len(sys.argv) > 4 else 3000
    t0 = time.time()
    done = damaged = 0
    verbose = bool(os.environ.get("SOAK_VERBOSE"))
    if verbose:  # a case that does not come back: where it stands, after a minute
        import faulthandler
        faulthandler.enable()
    for case in range(cases):
        if verbose:
            print("case %d starts" % case, flush=True)
            faulthandler.cancel_dump_traceback_later()
            faulthandler.dump_traceback_later(60, exit=True)
        if case % 10 == 0 and case:
            print("case %d of %d, %d round trips, %.0f s" % (case, cases, done, time.time() - t0), flush=True)
        if case % 5 == 4:
            done += several_blocks_in_flight(case, seed0 + case) > 0
            continue
        rng = np.random.default_rng(seed0 + case)
        mode = int(rng.choice([2, 2, 3, 4, 4, 5, 6]))
        size = int(rng.integers(min_bytes, max_bytes))
        raw, _ = F.synth_fastq(size, mode, seed=seed0 + case)
        recs = F.parse_fastq(raw)
        how = int(rng.integers(0, 4))
        if how and mode in (2, 4) and len(recs) <= 20000:  # (a Python loop over the records: small blocks only)
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
        # chain parameters off their defaults in a third of the cases: segment lengths and group sizes move every
        # boundary of the chain stage (segments per chain, groups per item, items per chain)
        if rng.random() < 0.35:
            ctx.set_chain_params(int(rng.choice([0, 1024, 2048, 4096])), seq_generic=bool(rng.random() < 0.2),
                                 seq_segment=int(rng.choice([1024, 2048, 4096])), seq_group=int(rng.choice([1, 2, 4, 8, 16])))
        with_index = rng.random() < 0.3
        if with_index:
            ctx.set_index_stride(int(rng.choice([1, 2, 4])) << 16)
        octx = O.OracleCtx(sft, qft)
        e = octx.encode(raw, recs)
        path = int(rng.integers(0, 3))  # 0: device-resident block, 1: host-pointer call, 2: unparsed chunk (GPU parser)
        if path == 0:
            b = ctx.dblock(raw, recs)
            b.encode(flags=F.F_DECODE_INDEX if with_index else 0)
            ctx.sync()
            rc = b.status()[0]
            g = b.fetch() if rc == 0 else None
        else:
            g = ctx.encode_block(raw, recs) if path == 1 else ctx.encode_raw(raw)
            rc = g["rc"]
            b = None
        assert rc == e["rc"], (case, path, rc, e["rc"])
        if rc == 0:
            for k in ("seq", "qual", "readlens", "n_count", "n_pos"):
                assert np.array_equal(g[k], e[k]), (case, mode, how, path, k)
            if path == 2:
                assert np.array_equal(g["recs"], recs), (case, "record table of the GPU parser")
            if b is not None:
                # decode of the ORACLE's streams (with this block's own index when it has one: the index belongs to
                # the streams, which are the oracle's byte for byte)
                idx = [b.fetch_index(0), b.fetch_index(1)] if with_index else None
                b.load_streams(e["seq"], e["qual"], e["n_count"], e["n_pos"])
                if idx:
                    assert b.load_index(0, idx[0]) == 0 and b.load_index(1, idx[1]) == 0
                b.wipe()
                ctx.decode_dblocks([b])
                ctx.sync()
                assert b.status()[0] == 0 and np.array_equal(b.fetch_raw(), raw), (case, mode, how, "decode", with_index)
            else:
                skel = O.blank_skeleton(raw, recs)
                rc, out = ctx.decode_block(e["seq"], e["qual"], e["n_count"], e["n_pos"], recs, skel)
                assert rc == 0 and np.array_equal(out, raw), (case, mode, how, "decode_block")
                # damaged streams: the verdict is the oracle's -- refused by both, or accepted by both with the same bytes
                for _ in range(3):
                    sq, ql = e["seq"].copy(), e["qual"].copy()
                    victim = sq if rng.random() < 0.5 else ql
                    kind = int(rng.integers(0, 3))
                    if kind == 0 and victim.size:
                        victim[int(rng.integers(0, victim.size))] ^= np.uint8(1 << int(rng.integers(0, 8)))
                    elif kind == 1 and victim.size > 4:
                        cut = int(rng.integers(1, min(victim.size - 1, 64)))
                        if victim is sq: sq = sq[:-cut]
                        else: ql = ql[:-cut]
                    elif victim.size:
                        victim[int(rng.integers(0, victim.size))] = np.uint8(rng.integers(0, 256))
                    orc, oout = octx.decode(sq, ql, e["n_count"], e["n_pos"], recs, skel)
                    grc, gout = ctx.decode_block(sq, ql, e["n_count"], e["n_pos"], recs, skel)
                    assert (grc == 0) == (orc == 0), (case, "damaged", kind, grc, orc)
                    if grc == 0:
                        assert np.array_equal(gout, oout), (case, "damaged", kind, "accepted by both, different bytes")
                    else:
                        assert grc == -3, (case, "damaged", grc)
                    damaged += 1
            done += 1
        if b is not None:
            b.close()
        octx.close()
        ctx.close()
        if max_bytes > (8 << 20):
            print("case %d of %d, %d round trips, %.0f s" % (case + 1, cases, done, time.time() - t0), flush=True)
    print("soak: %d cases, %d encoded and decoded byte-exactly, the rest refused by both coders alike; %d damaged streams decoded with the oracle's verdict" % (cases, done, damaged))


if __name__ == "__main__":
    main()
