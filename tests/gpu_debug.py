"""Stage-by-stage GPU-vs-oracle comparison, for bring-up on a GPU box:
    python tests/gpu_debug.py [fixture ...]
Prints where the first difference is instead of asserting."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402
import fqcomp28_amd as F  # noqa: E402


def first_diff(a, b):
    n = min(len(a), len(b))
    d = np.flatnonzero(a[:n] != b[:n])
    return (int(d[0]) if d.size else (n if len(a) != len(b) else -1)), len(a), len(b)


def run(name, raw, recs, seg=None, warm=None):
    print("==", name, "records", len(recs), "bases", int(recs["len"].sum()))
    sc, qc, sft, qft = O.freq_tables(raw, recs)
    t0 = time.time()
    gs, gq, gsc, gqc = F.freq_tables(raw, recs, want_counts=True)
    print("  freq_tables %.3fs  seq_counts eq %s  qual_counts eq %s  seq_ft eq %s  qual_ft eq %s" % (
        time.time() - t0, np.array_equal(sc, gsc), np.array_equal(qc, gqc),
        np.array_equal(sft.view(np.uint8), gs.view(np.uint8)), np.array_equal(qft.view(np.uint8), gq.view(np.uint8))))
    if not np.array_equal(sc, gsc):
        d = np.argwhere(sc != gsc)
        print("   seq count diffs", len(d), d[:5], sc[tuple(d[0])], gsc[tuple(d[0])])
    if not np.array_equal(qc, gqc):
        d = np.argwhere(qc != gqc)
        print("   qual count diffs", len(d), d[:5], qc[tuple(d[0])], gqc[tuple(d[0])])
    if not np.array_equal(sft["norm"], gs["norm"]):
        d = np.argwhere(sft["norm"] != gs["norm"])
        print("   seq norm diffs", len(d), d[:5])
    if not np.array_equal(qft["norm"], gq["norm"]):
        d = np.argwhere(qft["norm"][0] != gq["norm"][0])
        print("   qual norm diffs", len(d), d[:5])
        c = d[0][0]
        print("   ctx", c, "counts", qc[c], "oracle", qft["norm"][0][c], "gpu", gq["norm"][0][c], qft["logs"][0][c], gq["logs"][0][c])

    ctx = F.Context(sft, qft)
    if seg is not None:
        ctx.set_chain_params(seg, bool(warm))
    L = O.lib()
    bad = 0
    for stream, ft, alpha, nm in ((0, sft, 4, 256), (1, qft, 64, 8192)):
        for m in list(range(0, nm, max(1, nm // 64))) + [nm - 1]:
            ct, dt = ctx.dump_tables(stream, m)
            log = int(ft["logs"][0][m])
            oct_ = np.zeros(L.fo_ctable_words(log, alpha - 1), dtype=np.uint32)
            odt = np.zeros(L.fo_dtable_words(log), dtype=np.uint32)
            norm = np.ascontiguousarray(ft["norm"][0][m])
            L.fo_build_ctable(O.ptr(oct_), O.ptr(norm), alpha - 1, log)
            L.fo_build_dtable(O.ptr(odt), O.ptr(norm), alpha - 1, log)
            if not (np.array_equal(ct, oct_) and np.array_equal(dt, odt)):
                bad += 1
                if bad < 4:
                    print("   table mismatch stream", stream, "model", m, "log", log, first_diff(ct, oct_), first_diff(dt, odt))
    print("  tables checked, mismatches:", bad)

    octx = O.OracleCtx(sft, qft)
    e = octx.encode(raw, recs)
    t0 = time.time()
    g = ctx.encode_block(raw, recs, flags=1)
    print("  encode rc", g["rc"], "%.3fs" % (time.time() - t0))
    for k in ("seq", "qual", "readlens", "n_count", "n_pos", "raw_after"):
        print("   %-9s first_diff %s" % (k, first_diff(np.asarray(g[k]), np.asarray(e[k]))))
    skel = O.blank_skeleton(raw, recs)
    t0 = time.time()
    rc, out = ctx.decode_block(e["seq"], e["qual"], e["n_count"], e["n_pos"], recs, skel)
    print("  decode rc", rc, "%.3fs" % (time.time() - t0), "roundtrip", first_diff(out, raw))
    ctx.close()


if __name__ == "__main__":
    print("devices", F.device_count())
    names = sys.argv[1:] or ["SRR065390_1_first5", "without_ns", "SRR065390_sub_1"]
    for n in names:
        if n.startswith("synth"):
            mode = int(n[5:6])
            size = int(n.split(":")[1]) if ":" in n else 1 << 20
            raw, _ = F.synth_fastq(size, mode)
            recs = F.parse_fastq(raw)
        else:
            raw, recs = O.load_fastq(os.path.join(HERE, "golden", n + ".fastq"))
        run(n, raw, recs)
        if len(recs) > 100:
            run(n + " seg=64 generic", raw, recs, seg=64, warm=True)
