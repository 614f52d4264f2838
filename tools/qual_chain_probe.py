"""Longest serial run of the quality chain kernel and its sensitivity to the nominal segment."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import fqcomp28_amd as F

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
raw, _ = F.synth_fastq(1 << 30, int(sys.argv[2]) if len(sys.argv) > 2 else 2, seed=28)
blk = raw[: mib << 20]
recs = F.parse_fastq(blk)
last = recs[-1]
blk = blk[: int(last["qual_off"]) + int(last["len"]) + 1]
smp = raw[: 128 << 20]
srecs = F.parse_fastq(smp)
sft, qft, sc, qc = F.freq_tables(smp[: int(srecs[-1]["qual_off"]) + int(srecs[-1]["len"]) + 1], srecs, want_counts=True)
pop = (qc.sum(axis=1) > 0).sum()
print("populated qual contexts:", int(pop), "symbols in largest:", int(qc.sum(axis=1).max()))
ctx = F.Context(sft, qft)
ctx.set_lanes(1)
ctx.enable_timing(True)
b = ctx.dblock(blk, recs)
for seg in (1024, 256, 4096):
    ctx.set_chain_params(segment=seg)
    b.encode(); ctx.sync()
    ctx.enable_timing(True)   # restart the accumulators
    b.encode(); ctx.sync()
    tot, ks = ctx.last_timing()
    print("segment", seg, "longest (seq, qual):", b.longest_chain(),
          {k: round(v, 3) for k, v, _ in ks if k.startswith("qual")}, flush=True)
