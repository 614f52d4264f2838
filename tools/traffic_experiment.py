"""Timing experiment (wrong output by design): marginal cost of kernel groups / classes of stores.
Runs the configs[1] encode step normally, then again with every NAME[=VALUE] given on the command
line set in the environment (FQGPU_DEBUG_SKIP=<mask>, FQGPU_DEBUG_K1=<bits>, FQGPU_DEBUG_NO_SYM_STORE=1
...): what is not launched or stored keeps the previous step's data.
FQGPU_DEBUG_NO_ALIAS=1 is set here: enc16 normally lives in the key buffer, so without it a step
that skips K1 (or its key stores) would partition the previous step's (nb, bits) as if they were
keys and everything behind K1 would run on garbage -- that made "K1's key stores" look like 3 ms."""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, ".")
# the switches exist only in the experiments build of the library (make experiments)
subprocess.run(["make", "-C", "fqcomp28_amd/csrc", "-j", "6", "experiments"], check=True, stdout=subprocess.DEVNULL)
os.environ["FQGPU_LIB"] = os.path.abspath("tools/_build/libfqgpu_experiments.so")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("FQGPU_DEBUG_NO_ALIAS", "1")
import fqcomp28_amd as F
import bench

blocks = bench.make_workload(F, 1 << 30, 256 << 20, seed=28)
sft, qft = bench.sample_tables(F, blocks, 128 << 20, 0)
ctx = F.Context(sft, qft)
dblocks = [ctx.dblock(raw, recs) for raw, recs in blocks]
def run(steps):
    for b in dblocks: b.encode()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for b in dblocks: b.encode()
    ctx.sync()
    return (time.perf_counter() - t0) / steps * 1e3
print("normal             %.2f ms/step" % run(4), flush=True)
for arg in sys.argv[1:]:  # NAME or NAME=VALUE
    name, _, val = arg.partition("=")
    os.environ[name] = val or "1"
    print("%-28s %.2f ms/step" % (arg, run(4)), flush=True)
    del os.environ[name]
print("normal again       %.2f ms/step" % run(4), flush=True)
