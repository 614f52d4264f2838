"""How the GPU path depends on the DATA (bench.py measures BASELINE's synthetic reads only):
encode GB/s and decode ns per symbol of 4 x 64 MiB blocks for
  synthetic  -- BASELINE configs[1] reads: uniform bases, Phred ~ N(34, 5)
  binned     -- the same reads with four-level qualities ('#', '-', '8', 'F' at 5/10/15/70 %) that
                persist from one position to the next with probability 0.85 (long runs of one
                context, no reset symbols: what current Illumina output looks like to this model)
  constant   -- every quality 'F', every base 'A' (one context per stream from the fourth symbol on)
Every run is a round trip (decode output compared byte for byte).  One JSON line per data kind.
    python tools/datadep_bench.py [block MiB] [segment of the quality chain kernels, 0 = default] [kinds, comma separated]
(binned data with segments of 4096 / 8192 / 16384: 41.1 / 42.5 / 38.1 GB/s: the segment length is not the lever.)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bench  # noqa: E402
import fqcomp28_amd as F  # noqa: E402


def remake(blocks, kind, seed=11):
    if kind == "synthetic":
        return blocks
    rng = np.random.default_rng(seed)
    out = []
    for raw, recs in blocks:
        raw = raw.copy()
        n = int(recs["len"].sum())
        if kind == "binned":
            levels, p = np.frombuffer(b"#-8F", dtype=np.uint8), [0.05, 0.1, 0.15, 0.7]
            keep = rng.random(n) < 0.85
            keep[0] = False
            fresh = rng.choice(len(levels), size=n, p=p)
            q = levels[fresh[np.maximum.accumulate(np.where(keep, 0, np.arange(n)))]]
        else:
            q = np.full(n, ord("F"), dtype=np.uint8)
        # the reads of the generator have one length: write all quality lines through one index array
        L = int(recs["len"][0])
        assert (recs["len"] == L).all()
        idx = (recs["qual_off"].astype(np.int64)[:, None] + np.arange(L)[None, :]).ravel()
        raw[idx] = q
        if kind == "constant":
            sidx = (recs["seq_off"].astype(np.int64)[:, None] + np.arange(L)[None, :]).ravel()
            raw[sidx] = ord("A")
        out.append((raw, recs))
    return out


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    seg = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # segment of the quality chain kernels (0 = default)
    kinds = sys.argv[3].split(",") if len(sys.argv) > 3 else ("synthetic", "binned", "constant")
    base = bench.make_workload(F, 4 * mib << 20, mib << 20, seed=28)
    for kind in kinds:
        blocks = remake(base, kind)
        sft, qft = bench.sample_tables(F, blocks, max(32, mib // 2) << 20, 0)
        ctx = F.Context(sft, qft, device=0)
        ctx.set_lanes(4)
        if seg:
            ctx.set_chain_params(seg)
        db = [ctx.dblock(raw, recs) for raw, recs in blocks]
        for b in db:
            b.encode()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            for b in db:
                b.encode()
        ctx.sync()
        enc = (time.perf_counter() - t0) / 3
        sizes = [b.status()[1] for b in db]
        ctx.enable_timing(True)  # one more step with events around every kernel group
        for b in db:
            b.encode()
        ctx.sync()
        _, spans = ctx.last_timing()
        ctx.enable_timing(False)
        kern = {name: round(ms / max(n, 1), 3) for name, ms, n in spans}
        for b in db:
            b.wipe()
        ctx.sync()
        t0 = time.perf_counter()
        ctx.decode_dblocks(db)
        ctx.sync()
        dec = time.perf_counter() - t0
        ok = all(b.status()[0] == 0 for b in db) and all(np.array_equal(b.fetch_raw(), r) for b, (r, _) in zip(db, blocks))
        raw_bytes = sum(r.size for r, _ in blocks)
        nsym = max(int(r["len"].sum()) for _, r in blocks)
        print(json.dumps({"data": kind, "segment": seg, "blocks": "4 x %d MiB" % mib, "encode_GBps": round(raw_bytes / enc / 1e9, 1),
                          "decode_MBps": round(raw_bytes / dec / 1e6, 1), "decode_ns_per_symbol_per_lane": round(dec * 1e9 / nsym, 1),
                          "seq_bytes": int(sum(s["seq_len"] for s in sizes)), "qual_bytes": int(sum(s["qual_len"] for s in sizes)),
                          "roundtrip_ok": bool(ok),
                          "kernels_ms": dict(sorted(kern.items(), key=lambda kv: -kv[1])[:8])}), flush=True)
        for b in db:
            b.close()
        ctx.close()


if __name__ == "__main__":
    main()
