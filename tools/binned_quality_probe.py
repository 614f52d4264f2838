"""Encode time of blocks whose quality tables have NO reset symbols: constant quality (configs[0])
and four-level binned qualities.  Every quality segment is opaque there."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import fqcomp28_amd as F

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
raw, _ = F.synth_fastq(mib << 20, 2, seed=28)
recs = F.parse_fastq(raw)
rng = np.random.default_rng(3)

def with_quals(make):
    out = raw.copy()
    n = int(recs["len"].sum())
    q = make(n)
    L = int(recs["len"][0])
    assert (recs["len"] == L).all()
    idx = (recs["qual_off"].astype(np.int64)[:, None] + np.arange(L)[None, :]).ravel()
    out[idx] = q
    return out

def binned(n):  # four levels, sticky
    lv = np.frombuffer(b"#-8F", dtype=np.uint8)
    keep = rng.random(n) < 0.85
    keep[0] = False
    fresh = rng.choice(4, size=n, p=[0.05, 0.1, 0.15, 0.7])
    return lv[fresh[np.maximum.accumulate(np.where(keep, 0, np.arange(n)))]]

for name, data in (("normal (config 2)", raw), ("binned 4 levels", with_quals(binned)),
                   ("constant 'I'", with_quals(lambda n: np.full(n, ord("I"), dtype=np.uint8)))):
    smp = data[: 64 << 20]
    srecs = F.parse_fastq(smp)
    sft, qft = F.freq_tables(smp[: int(srecs[-1]["qual_off"]) + int(srecs[-1]["len"]) + 1], srecs)
    ctx = F.Context(sft, qft)
    ctx.set_lanes(1)
    b = ctx.dblock(data, recs)
    b.encode(); ctx.sync()
    ctx.enable_timing(True)
    t0 = time.perf_counter(); b.encode(); ctx.sync(); dt = time.perf_counter() - t0
    rc, st = b.status()
    tot, ks = ctx.last_timing()
    q = {k: round(v, 2) for k, v, _ in ks if k.startswith("qual.") and v > 0.2}
    print("%-18s rc %d  %.1f ms per %d MiB block (%.1f GB/s)  qual bytes %d  %s" %
          (name, rc, dt * 1e3, mib, data.size / dt / 1e9, st["qual_len"], q), flush=True)
    b.close(); ctx.close()
