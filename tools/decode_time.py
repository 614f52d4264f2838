"""Decode timing of one library (FQGPU_LIB to try another build): 4 blocks of [MiB] in one launch, ns per symbol
and lane, round trip compared.    python tools/decode_time.py [MiB]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import fqcomp28_amd as F
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 32
blocks = bench.make_workload(F, 4 * mib << 20, mib << 20, seed=28)
sft, qft = bench.sample_tables(F, blocks, 16 << 20, 0)
ctx = F.Context(sft, qft, device=0)
ctx.set_lanes(4)
db = [ctx.dblock(raw, recs) for raw, recs in blocks]
for b in db:
    b.encode()
ctx.sync()
nsym = max(int(r["len"].sum()) for _, r in blocks)
for rep in range(2):
    for b in db:
        b.wipe()
    ctx.sync()
    t0 = time.perf_counter()
    ctx.decode_dblocks(db)
    ctx.sync()
    dt = time.perf_counter() - t0
ok = all(b.status()[0] == 0 for b in db) and bool(np.array_equal(db[-1].fetch_raw(), blocks[-1][0]))
print(os.environ.get("FQGPU_LIB", "product"), "ns/symbol %.1f" % (dt * 1e9 / nsym), "roundtrip", ok, flush=True)
