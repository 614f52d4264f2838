"""Why does a handle created SECOND in a process run 5-9 % slower?  Times the same 16 x 64 MiB
job on a handle created first, on one created behind another handle that has coded blocks (its lanes' streams exist),
and on one created after that other handle was destroyed -- under whatever GPU_MAX_HW_QUEUES the environment sets.
    GPU_MAX_HW_QUEUES=16 python3 tools/second_handle_probe.py [first|second|after_destroy]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import fqcomp28_amd as F  # noqa: E402

how = sys.argv[1] if len(sys.argv) > 1 else "first"
blocks = bench.make_workload(F, 1 << 30, 64 << 20, seed=28)
sft, qft = bench.sample_tables(F, blocks, 128 << 20, 0)
other = None
if how != "first":
    big = bench.make_workload(F, 1 << 30, 256 << 20, seed=28)
    other = F.Context(sft, qft)
    ob = [other.dblock(raw, recs) for raw, recs in big]
    for _ in range(2):
        for b in ob:
            b.encode()
    other.sync()
    if how == "after_destroy":
        for b in ob:
            b.close()
        other.close()
        other = None
ctx = F.Context(sft, qft)
db = [ctx.dblock(raw, recs) for raw, recs in blocks]
for _ in range(2):
    for b in db:
        b.encode()
ctx.sync()
best = None
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(5):
        for b in db:
            b.encode()
    ctx.sync()
    dt = (time.perf_counter() - t0) / 5
    best = dt if best is None else min(best, dt)
print("%s queues=%s: %.3f ms per step, %.1f MB/s" % (how, os.environ.get("GPU_MAX_HW_QUEUES"), best * 1e3, sum(r.size for r, _ in blocks) / best / 1e6), flush=True)
