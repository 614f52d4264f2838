"""The timed encode loop alone on one kind of data, for profilers:
    rocprofv3 --kernel-trace --stats -- python3 tools/encode_loop.py [synth|real|binned|constant|config3|config4] [steps] [lanes] [quality segment] [sequence segment]
(lanes 0 = the library's default; segments 0 = the library's defaults)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import bench  # noqa: E402
import fqcomp28_amd as F  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "synth"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 4
if kind == "real":
    blocks = bench.make_real_workload(F, 1 << 30, 256 << 20)
elif kind == "config3":
    blocks = bench.make_workload(F, 1 << 30, 64 << 20, seed=28)
elif kind == "config4":
    blocks = bench.make_workload(F, 256 << 20, 64 << 20, seed=28, mode=4)
else:
    blocks = bench.make_workload(F, 1 << 30, 256 << 20, seed=28, mode={"synth": 2, "binned": 3, "constant": 5, "two_levels": 6}[kind])
sft, qft = bench.sample_tables(F, blocks, 128 << 20, 0)
ctx = F.Context(sft, qft)
ctx.set_lanes(lanes)
qseg = int(sys.argv[4]) if len(sys.argv) > 4 else 0
sseg = int(sys.argv[5]) if len(sys.argv) > 5 else 0
if qseg or sseg:
    ctx.set_chain_params(segment=qseg, seq_segment=sseg or None)
db = [ctx.dblock(raw, recs) for raw, recs in blocks]
for b in db:
    b.encode()
ctx.sync()
t0 = time.perf_counter()
for _ in range(steps):
    for b in db:
        b.encode()
ctx.sync()
dt = (time.perf_counter() - t0) / steps
print("%s: %.3f ms per step, %.1f MB/s, rc %s" % (kind, dt * 1e3, sum(r.size for r, _ in blocks) / dt / 1e6, [b.status()[0] for b in db]), flush=True)
