"""A/B timing of library builds on ONE box: the boxes of the pool differ by +-2 % and drift, which is more than most
single changes to the encoder are worth, so variants are never compared across gpurun calls.
    python3 tools/ab_bench.py [--data synth|real|binned|constant|config3|config4] [--rounds 3] [--steps 5] [--lanes 4] name=path/to/lib.so ...
Every variant runs in FRESH PROCESSES, one after the other, `rounds` times round-robin (A B C A B C ...): two handles
in one process are NOT comparable -- the handle created second runs 5 % slower whatever its code is (measured with two
copies of one library: where the driver places the second handle's scratch), a fresh process always gets the first
placement.  The workload is generated once (into /dev/shm) and mapped by the children.  Prints per variant the median
and the best ms per step and the ratio to the first one; --check compares every variant's streams of the last block
with the first variant's (sha1); --spans adds the HIP-event kernel spans of one extra step."""
import argparse
import hashlib
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np  # noqa: E402


def child(args):
    import fqcomp28_amd as F   # FQGPU_LIB set by the parent
    d = np.load(args.child, allow_pickle=False)
    n = int(d["n"])
    blocks = [(d["raw%d" % i], d["recs%d" % i].view(F.REC_DTYPE)) for i in range(n)]
    sft, qft = d["sft"].view(F.SEQ_FT_DTYPE), d["qft"].view(F.QUAL_FT_DTYPE)
    ctx = F.Context(sft, qft)
    ctx.set_lanes(args.lanes)
    db = [ctx.dblock(raw, recs) for raw, recs in blocks]
    for _ in range(2):
        for b in db:
            b.encode()
    ctx.sync()
    ts = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            for b in db:
                b.encode()
        ctx.sync()
        ts.append((time.perf_counter() - t0) / args.steps * 1e3)
    out = {"ms": ts, "rc": [b.status()[0] for b in db]}
    if args.check:
        g = db[-1].fetch()
        h = hashlib.sha1()
        for k in ("seq", "qual", "readlens", "n_count", "n_pos"):
            h.update(np.ascontiguousarray(g[k]).tobytes())
        out["sha1"] = h.hexdigest()
    if args.spans:
        ctx.enable_timing(True)
        for b in db:
            b.encode()
        ctx.sync()
        _, spans = ctx.last_timing()
        ctx.enable_timing(False)
        out["spans"] = {k: round(ms / max(c, 1), 3) for k, ms, c in spans}
    print("ABRESULT " + json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="*", help="name=path")
    ap.add_argument("--data", default="synth", choices=["synth", "real", "binned", "constant", "two_levels", "config3", "config4"])
    ap.add_argument("--rounds", type=int, default=3, help="fresh processes per variant")
    ap.add_argument("--reps", type=int, default=3, help="timed repetitions of `steps` steps inside a process")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--lanes", type=int, default=0)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--spans", action="store_true")
    ap.add_argument("--child", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.child:
        return child(args)
    import bench
    import fqcomp28_amd as F0   # host-side helpers only (generator, parser, dataset analysis)
    libs = [(a.split("=", 1)[0], a.split("=", 1)[1]) for a in args.libs]
    if args.data == "real":
        blocks = bench.make_real_workload(F0, 1 << 30, 256 << 20)
    elif args.data == "config3":
        blocks = bench.make_workload(F0, 1 << 30, 64 << 20, seed=28)
    elif args.data == "config4":
        blocks = bench.make_workload(F0, 256 << 20, 64 << 20, seed=28, mode=4)
    else:
        blocks = bench.make_workload(F0, 1 << 30, 256 << 20, seed=28, mode={"synth": 2, "binned": 3, "constant": 5, "two_levels": 6}[args.data])
    sft, qft = bench.sample_tables(F0, blocks, 128 << 20, 0)
    raw_bytes = sum(r.size for r, _ in blocks)
    path = "/dev/shm/ab_bench_%d.npz" % os.getpid()
    arrs = {"n": np.array(len(blocks)), "sft": np.ascontiguousarray(sft).view(np.uint8), "qft": np.ascontiguousarray(qft).view(np.uint8)}
    for i, (raw, recs) in enumerate(blocks):
        arrs["raw%d" % i] = raw
        arrs["recs%d" % i] = np.ascontiguousarray(recs).view(np.uint8)
    np.savez(path, **arrs)
    del blocks, arrs
    res = {n: [] for n, _ in libs}
    extra = {}
    try:
        for r in range(args.rounds):
            for n, p in libs:
                env = dict(os.environ, FQGPU_LIB=os.path.abspath(p))
                cmd = [sys.executable, os.path.abspath(__file__), "--child", path, "--steps", str(args.steps), "--lanes", str(args.lanes), "--reps", str(args.reps)]
                if args.check and r == 0:
                    cmd.append("--check")
                if args.spans and r == 0:
                    cmd.append("--spans")
                out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
                line = [ln for ln in out.stdout.splitlines() if ln.startswith("ABRESULT ")]
                if out.returncode != 0 or not line:
                    print("%s: child failed rc=%d\n%s\n%s" % (n, out.returncode, out.stdout[-2000:], out.stderr[-2000:]), flush=True)
                    continue
                d = json.loads(line[-1][9:])
                res[n] += d["ms"]
                if any(d["rc"]):
                    print("%s: rc %s" % (n, d["rc"]), flush=True)
                extra.setdefault(n, {}).update({k: d[k] for k in ("sha1", "spans") if k in d})
    finally:
        os.unlink(path)
    base = statistics.median(res[libs[0][0]]) if res[libs[0][0]] else float("nan")
    for n, _ in libs:
        ts = res[n]
        if not ts:
            continue
        med = statistics.median(ts)
        print("%-14s median %7.3f ms/step  best %7.3f  %8.1f MB/s  x%.4f of %s   all: %s"
              % (n, med, min(ts), raw_bytes / med / 1e3, base / med, libs[0][0], " ".join("%.2f" % t for t in ts)), flush=True)
    if args.check:
        ref = extra.get(libs[0][0], {}).get("sha1")
        for n, _ in libs[1:]:
            print("%-14s streams of the last block equal to %s's: %s" % (n, libs[0][0], extra.get(n, {}).get("sha1") == ref and ref is not None), flush=True)
    if args.spans:
        for n, _ in libs:
            sp = extra.get(n, {}).get("spans", {})
            print("%-14s spans ms: %s" % (n, " ".join("%s=%.3f" % kv for kv in sorted(sp.items(), key=lambda kv: -kv[1]))), flush=True)


if __name__ == "__main__":
    main()
