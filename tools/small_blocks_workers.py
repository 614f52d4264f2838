"""Small blocks (-R 16 / -R 64) with the reference's threading model: W worker threads, each with
its own handle (src/process.cpp:46-68: one workspace per worker), every worker coding its share
of the blocks from HBM-resident inputs.  A small block is bound by host-side launch cost per
handle (about 50 kernel launches per block), so throughput scales with the workers until the GPU
is full.  Usage: python tools/small_blocks_workers.py [block MiB] [total MiB]"""
import sys, time, threading
import numpy as np
sys.path.insert(0, ".")
import fqcomp28_amd as F

bm = int(sys.argv[1]) if len(sys.argv) > 1 else 16
total = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
raw, _ = F.synth_fastq(total << 20, 2, seed=28)
recs = F.parse_fastq(raw)
smp_end = int(np.searchsorted(recs["qual_off"], 128 << 20))
sft, qft = F.freq_tables(raw[: int(recs[smp_end - 1]["qual_off"]) + int(recs[smp_end - 1]["len"]) + 1], recs[:smp_end])
# blocks of ~bm MiB on record boundaries
ends = recs["qual_off"].astype(np.int64) + recs["len"] + 1
cuts = [0]
while cuts[-1] < len(recs):
    start_byte = 0 if cuts[-1] == 0 else int(ends[cuts[-1] - 1])
    cuts.append(int(np.searchsorted(ends, start_byte + (bm << 20), side="right")) if start_byte + (bm << 20) < ends[-1] else len(recs))
blocks = []
for a, b in zip(cuts[:-1], cuts[1:]):
    if b <= a: continue
    s = 0 if a == 0 else int(ends[a - 1]); e = int(ends[b - 1])
    r = recs[a:b].copy(); r["seq_off"] -= s; r["qual_off"] -= s
    blocks.append((raw[s:e], r))
print("%d blocks of ~%d MiB" % (len(blocks), bm), flush=True)

for W, lanes in ((1, 4), (2, 4), (4, 2), (8, 2)):
    ctxs = [F.Context(sft, qft) for _ in range(W)]
    for c in ctxs: c.set_lanes(lanes)
    mine = [[c.dblock(*blocks[i]) for i in range(w, len(blocks), W)] for w, c in enumerate(ctxs)]
    def work(w, reps):
        for _ in range(reps):
            for b in mine[w]: b.encode()
            ctxs[w].sync()
    for w in range(W): work(w, 1)  # warm-up
    best = 1e9
    for _ in range(3):
        th = [threading.Thread(target=work, args=(w, 2)) for w in range(W)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        best = min(best, (time.perf_counter() - t0) / 2)
    ok = all(b.status()[0] == 0 for m in mine for b in m)
    print("workers %d x lanes %d: %.1f ms per %d MiB = %.1f GB/s  ok=%s" % (W, lanes, best * 1e3, total, raw.size / best / 1e9, ok), flush=True)
    for m in mine:
        for b in m: b.close()
    for c in ctxs: c.close()
