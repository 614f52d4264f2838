"""Two properties of the HIP runtime that the host path is built around, measured:
  free   -- hipHostFree (through FQGPU_PINNED_CACHE_MB=0, i.e. fqgpu_host_free without its cache) while another
            thread's decode kernel runs, against an idle device
  stagger -- four handles decoding one block each from four threads that start 0.5 s apart: wall time against
            one decode (the copies back are issued behind the kernels: no DMA engine is held by a waiting copy)
    FQGPU_PINNED_CACHE_MB=0 python tools/host_runtime_probes.py free
    python tools/host_runtime_probes.py stagger [MiB]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402
import bench  # noqa: E402
import fqcomp28_amd as F  # noqa: E402
from fqcomp28_amd.binding import lib  # noqa: E402
import oracle_lib as O  # noqa: E402


def coded(ctx, raw, recs):
    db = ctx.dblock(raw, recs)
    db.encode()
    ctx.sync()
    e = db.fetch()
    db.close()
    return e


def probe_free():
    blocks = bench.make_workload(F, 32 << 20, 32 << 20, seed=28)
    sft, qft = bench.sample_tables(F, blocks, 16 << 20, 0)
    ctx = F.Context(sft, qft, device=0)
    raw, recs = blocks[0]
    e = coded(ctx, raw, recs)
    skel = O.blank_skeleton(raw, recs)

    def dec():
        t0 = time.perf_counter()
        ctx.decode_block(e["seq"], e["qual"], e["n_count"], e["n_pos"], recs, skel)
        print("decode took %.2f s" % (time.perf_counter() - t0), flush=True)
    th = threading.Thread(target=dec)
    th.start()
    time.sleep(0.4)
    for when in ("while a decode kernel runs", "idle device"):
        t0 = time.perf_counter()
        p = lib().fqgpu_host_alloc(64 << 20)
        t1 = time.perf_counter()
        lib().fqgpu_host_free(p)
        t2 = time.perf_counter()
        print("%s: host_alloc 64 MiB %.3f s, host_free %.3f s" % (when, t1 - t0, t2 - t1), flush=True)
        th.join()


def probe_stagger(mib):
    blocks = bench.make_workload(F, 4 * mib << 20, mib << 20, seed=28)
    sft, qft = bench.sample_tables(F, blocks, min(128, 4 * mib) << 20, 0)
    ctxs = [F.Context(sft, qft, device=0) for _ in range(4)]
    encs = [coded(ctxs[0], raw, recs) for raw, recs in blocks]
    skel = [O.blank_skeleton(raw, recs) for raw, recs in blocks]
    T0 = time.perf_counter()

    def work(i):
        time.sleep(0.5 * i)
        t0 = time.perf_counter()
        e = encs[i]
        rc, o = ctxs[i].decode_block(e["seq"], e["qual"], e["n_count"], e["n_pos"], blocks[i][1], skel[i])
        print("worker %d: start %.2f took %.2f s rc %d equal %s" % (i, t0 - T0, time.perf_counter() - t0, rc, bool(np.array_equal(o, blocks[i][0]))), flush=True)
    ths = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    print("wall %.2f s" % (time.perf_counter() - T0))


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "stagger"
    if what == "free":
        probe_free()
    else:
        probe_stagger(int(sys.argv[2]) if len(sys.argv) > 2 else 64)
