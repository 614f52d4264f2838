"""Where the host-pointer encode call spends its time (PCIe-inclusive path)."""
import sys, time, threading
import numpy as np
sys.path.insert(0, ".")
import fqcomp28_amd as F

MB = 1e6
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
raw, _ = F.synth_fastq(mib << 20, 2, seed=11)
recs = F.parse_fastq(raw)
sft, qft = F.freq_tables(raw[: min(raw.size, 128 << 20)], F.parse_fastq(raw[: min(raw.size, 128 << 20)]))
ctx = F.Context(sft, qft)
for _ in range(4): ctx.encode_block(raw, recs)  # every lane allocates its scratch on first use
for it in range(4):
    t0 = time.perf_counter(); b = ctx.dblock(raw, recs); t1 = time.perf_counter()
    b.encode(); ctx.sync(); t2 = time.perf_counter()
    b.status(); out = b.fetch(); t3 = time.perf_counter()
    b.close(); t4 = time.perf_counter()
    print("create(H2D) %.1f ms  encode %.1f ms  fetch(D2H) %.1f ms  destroy %.1f ms" %
          ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3), flush=True)
bufs = ctx.host_buffers(len(recs), int(recs["len"].sum()))
for _ in range(3):
    t0 = time.perf_counter(); got = ctx.encode_block_into(raw, recs, bufs); dt = time.perf_counter() - t0
    assert got[0] == 0
    print("encode_block one thread: %.1f ms  %.1f MB/s" % (dt * 1e3, raw.size / dt / MB), flush=True)

for T in (2, 4):
    ctxs = [F.Context(sft, qft) for _ in range(T)]
    bb = [c.host_buffers(len(recs), int(recs["len"].sum())) for c in ctxs]
    rr = [raw.copy() for c in ctxs]
    for c, b, r in zip(ctxs, bb, rr):
        for _ in range(4): c.encode_block_into(r, recs, b)
    def work(c, b, r):
        for _ in range(4): c.encode_block_into(r, recs, b)
    th = [threading.Thread(target=work, args=(c, b, r)) for c, b, r in zip(ctxs, bb, rr)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print("encode_block %d threads: %.1f MB/s aggregate" % (T, 4 * T * raw.size / dt / MB), flush=True)
    for c in ctxs: c.close()
