#!/usr/bin/env python3
"""bench.py -- throughput of the FASTQ block entropy-coding hot path on MI355X.

A step = one pass of the hot path over one batch: every block of the workload is ENCODED
(seq + qual FSE streams, readlens, N side streams) from inputs already resident in HBM.

`value`  BASELINE.json configs[1] on every GPU: rank r codes its own 1 GiB of synthetic 150 bp reads (uniform ACGT,
         Phred ~ N(34,5) clipped to [2,41]; own seed) in the reference's default -R 256 blocks (4 blocks of 256 MiB)
         with ONE sample's tables (first 128 MiB of rank 0's reads, broadcast as bytes).  The units (blocks) are
         independent given the tables and are sharded over the ranks: work per GPU fixed, "scaling": "weak".
`strong_config2_MBps`  (every N, after the timed region) BASELINE.json configs[2], the reference's scaling config:
         ONE 1 GiB job cut into 64 MiB blocks (16 of them), block b coded by rank b mod N, tables from the job's
         first 128 MiB: total work fixed (two blocks per GPU at N = 8).  At N = 1 the same number is also printed as
         `encode_config3_MBps` (16 x 64 MiB on one GPU).  `--layout strong` makes THIS the timed `value` instead.
`--gpus N` without a launcher (no WORLD_SIZE in the environment): the parent starts N fresh processes of this script
         with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set -- before anything touches a GPU -- and
         exits with their worst exit code; it refuses (exit 2) when fewer than N devices are visible.
One process per GPU; no data-path collective and no RCCL call anywhere: the table broadcast, the barriers and the
max-over-ranks of the elapsed time go over gloo.
value = raw FASTQ bytes all ranks coded per second (max-over-ranks time).  After the timed region
(not part of `value`): a whole timed block is byte-compared with the CPU oracle (also the source of
`ratio_vs_reference`), decode of the same archive (configs[4]: blocks dealt over the N ranks), the
CPU baselines, the host-pointer path incl. PCIe, other data (binned / constant / configs[3] / tiled REAL reads).
One JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # up to 8 blocks in flight x (seq, qual) streams, one hardware queue each

import numpy as np  # noqa: E402

MB = 1e6
PROFILE_TRAFFIC = "r04_traffic.json"  # counters and rocprof launch times of this round (tools/refresh_profiles.py)


def kernel_sources_sha():
    """sha256 over the device code: a committed counter profile is only quoted while this is unchanged"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "fqcomp28_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def profile_stamp():
    """which committed counter profile bench.py would quote, and whether it was taken on this device code"""
    out = {"file": "profiles/" + PROFILE_TRAFFIC, "current_kernel_sources_sha": kernel_sources_sha()}
    try:
        with open(os.path.join(ROOT, "profiles", PROFILE_TRAFFIC)) as fh:
            tj = json.load(fh)
        out.update(commit=tj.get("commit"), kernel_sources_sha=tj.get("kernel_sources_sha"))
    except Exception:
        out.update(commit=None, kernel_sources_sha=None)
    return out


def load_profile(block_mib):
    """profiles/<round>_traffic.json while it describes THIS device code and block size, else None"""
    try:
        with open(os.path.join(ROOT, "profiles", PROFILE_TRAFFIC)) as fh:
            tj = json.load(fh)
    except Exception:
        return None
    if tj.get("kernel_sources_sha") != kernel_sources_sha() or block_mib != tj.get("block_mib", 256):
        return None
    return tj


def make_workload(F, total_bytes, block_bytes, seed, first_id=0, mode=2):
    """-> list of (raw, recs) blocks of whole records, about block_bytes each (mode: include/fqgpu.h,
    fqgpu_synth_fastq: 2 = BASELINE configs[1], 4 = configs[3], 3 = binned qualities, 5 = constant)."""
    blocks, done, next_id = [], 0, first_id
    while done < total_bytes:
        want = min(block_bytes, total_bytes - done)
        raw, n = F.synth_fastq(want, mode, seed=seed, first_read_id=next_id)
        if n == 0:
            break
        next_id += n
        done += want
        blocks.append((raw, F.parse_fastq(raw)))
    return blocks


def sample_tables(F, blocks, sample_bytes, device):
    """Dataset analysis on the first sample_bytes (reference: --sample-size-Mb 128)."""
    got, parts = 0, []
    for raw, _ in blocks:
        take = min(raw.size, sample_bytes - got)
        parts.append(raw[:take])
        got += take
        if got >= sample_bytes:
            break
    sample = np.concatenate(parts)
    recs = F.parse_fastq(sample)
    return F.freq_tables(sample, recs, device=device)


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    return O


REAL_FILES = ("SRR065390_sub_1.fastq", "SRR065390_sub_2.fastq", "without_ns.fastq")


def make_real_workload(F, total_bytes, block_bytes, first_id=1):
    """REAL statistics: the 2 851 reads of SRR065390 the reference ships as its test data (tests/golden = the
    reference's test/data: 100 bp Illumina reads with N runs, '#' tails, poly-A) tiled to `total_bytes` with fresh
    read ids -- every pass over the 2 851 reads in another order -- in blocks of about block_bytes.  The reference's
    published numbers are on the full SRR065390 files (benchmark/Results.md:1-5), which cannot be fetched here.
    Fixed-width records (header @SRR065390.<10 digits> <10 digits> length=100), built with numpy."""
    seqs, quals = [], []
    for f in REAL_FILES:
        raw = np.fromfile(os.path.join(ROOT, "tests", "golden", f), dtype=np.uint8)
        recs = F.parse_fastq(raw)
        assert np.all(recs["len"] == 100)
        idx = np.arange(100)
        seqs.append(raw[recs["seq_off"][:, None].astype(np.int64) + idx])
        quals.append(raw[recs["qual_off"][:, None].astype(np.int64) + idx])
    seqs, quals = np.concatenate(seqs), np.concatenate(quals)
    n_src = len(seqs)
    head = np.frombuffer(b"@SRR065390.", dtype=np.uint8)
    tail = np.frombuffer(b" length=100\n", dtype=np.uint8)
    rec_bytes = head.size + 10 + 1 + 10 + tail.size + 100 + 3 + 100 + 1
    per_block = max(1, block_bytes // rec_bytes)
    blocks, done, next_id = [], 0, first_id
    pow10 = 10 ** np.arange(9, -1, -1, dtype=np.int64)
    while done < total_bytes:
        n = min(per_block, (total_bytes - done) // rec_bytes)
        if n == 0:
            break
        ids = next_id + np.arange(n, dtype=np.int64)
        src = (ids * 1237 + 7 * (ids // n_src)) % n_src   # a permutation of the source reads per pass (1237 and 2851 are coprime)
        digits = ((ids[:, None] // pow10) % 10 + 48).astype(np.uint8)
        rec = np.empty((n, rec_bytes), dtype=np.uint8)
        o = 0
        for part in (head, digits, np.frombuffer(b" ", np.uint8), digits, tail, seqs[src], np.frombuffer(b"\n+\n", np.uint8), quals[src],
                     np.frombuffer(b"\n", np.uint8)):
            w = part.shape[-1]
            rec[:, o:o + w] = part
            o += w
        assert o == rec_bytes
        raw = rec.reshape(-1)
        blocks.append((raw, F.parse_fastq(raw)))
        next_id += n
        done += n * rec_bytes
    return blocks


def child_environments(n, port, base=None):
    """Environment of every rank the parent of `--gpus N` starts (one process per GPU, gloo rendezvous on
    127.0.0.1): what `python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1` would set."""
    envs = []
    for r in range(n):
        e = dict(os.environ if base is None else base)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        envs.append(e)
    return envs


def spawn_ranks(n, all_on_gpu0, argv):
    """`--gpus N` from a bare shell: N fresh children of this script, started BEFORE this process has touched a GPU
    (counting devices does not initialise one); never re-executes a process that has.  -> exit code."""
    import socket
    import subprocess
    if not os.environ.get("FQ_BENCH_SPAWN_ONLY") and not all_on_gpu0:
        import torch
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write("bench.py: --gpus %d but %d device(s) visible (use --all-on-gpu0 to rehearse the ranks on one GPU)\n" % (n, have))
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    kids = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e) for e in child_environments(n, port)]
    return max(abs(k.wait()) for k in kids)


def strong_config2(F, farm, dist, rank, world, device, lanes, steps, mib, block_mib, sample_mib, oracle):
    """BASELINE.json configs[2] (the reference's scaling config, src/process.cpp:46-68 with one worker per GPU): ONE job
    of `mib` MiB in `block_mib` MiB blocks, block b -> rank b mod world, tables from the job's first `sample_mib` MiB
    (rank 0, broadcast).  Barrier, `steps` passes over this rank's blocks, barrier; max over ranks."""
    import torch
    job = make_workload(F, mib << 20, block_mib << 20, seed=28)   # the same job on every rank (deterministic per read)
    sft = qft = None
    if rank == 0:
        sft, qft = sample_tables(F, job, sample_mib << 20, device)
    sft, qft = farm.broadcast_tables(sft, qft, dist)
    mine = farm.shard_blocks(len(job), rank, world)
    ctx = F.Context(sft, qft, device=device)
    ctx.set_lanes(lanes)
    db = [ctx.dblock(*job[b]) for b in mine]
    for b in db:
        b.encode()
    ctx.sync()
    farm.barrier(dist)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for b in db:
            b.encode()
    ctx.sync()
    farm.barrier(dist)
    dt = farm.reduce_max(time.perf_counter() - t0, dist)
    ok = all(b.status()[0] == 0 for b in db)
    job_bytes = sum(r.size for r, _ in job)
    out = {"config": "BASELINE configs[2]: ONE job of %d MiB in %d MiB blocks, block b -> rank b mod %d, tables from the first %d MiB"
                     % (mib, block_mib, world, sample_mib),
           "MBps": round(job_bytes * steps / dt / MB, 1), "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
           "job_blocks": len(job), "blocks_this_gpu": len(mine), "scaling": "strong",
           "all_rc_zero": bool(farm.reduce_sum(0.0 if ok else 1.0, dist) == 0.0)}
    if oracle is not None and rank == 0 and db:
        octx = oracle.OracleCtx(sft, qft)
        e = octx.encode(*job[mine[-1]])
        g = db[-1].fetch()
        out["oracle_block_all_equal"] = bool(e["rc"] == 0 and all(np.array_equal(g[k], e[k]) for k in ("seq", "qual", "readlens", "n_count", "n_pos")))
        octx.close()
    for b in db:
        b.close()
    ctx.close()
    return out


def other_data(F, O, device, lanes, steps, kind, mode, total_mib, block_mib, sample_mib, decode_mib, blocks=None, classes=False):
    """The same measurement on data that is NOT the headline config (after the timed region, rank 0):
    encode MB/s of `total_mib` in `block_mib` blocks with `lanes` blocks in flight, one WHOLE block
    byte-compared with the oracle, and the decode walk's ns per symbol on 4 x `decode_mib` MiB of it.
    blocks: a prebuilt workload instead of synth mode `mode`; classes: the quality segments of one block by class."""
    make_small = None
    if blocks is None:
        blocks = make_workload(F, total_mib << 20, block_mib << 20, seed=28, mode=mode)
    else:
        make_small = lambda: make_real_workload(F, 4 * decode_mib << 20, decode_mib << 20, first_id=500_000_000)  # noqa: E731
    sft, qft = sample_tables(F, blocks, sample_mib << 20, device)
    ctx = F.Context(sft, qft, device=device)
    ctx.set_lanes(lanes)
    db = [ctx.dblock(raw, recs) for raw, recs in blocks]
    for b in db:
        b.encode()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for b in db:
            b.encode()
    ctx.sync()
    dt = time.perf_counter() - t0
    raw_bytes = sum(r.size for r, _ in blocks)
    out = {"data": kind, "blocks": "%d x %d MiB" % (len(blocks), block_mib), "steps": steps,
           "MBps": round(raw_bytes * steps / dt / MB, 1), "ms_per_step": round(dt / steps * 1e3, 3)}
    sizes = [b.status() for b in db]
    if classes:  # (the last block is the last one its lane coded: the lane's segment table is still this block's)
        out["quality_segments_by_class"] = db[-1].qual_segment_classes()
        out["quality_segment_symbols"] = "segments of one block's quality chains (4096 symbols each, one chain per context)"
    out["rc"] = [rc for rc, _ in sizes]
    out["seq_bytes"] = int(sum(st["seq_len"] for _, st in sizes))
    out["qual_bytes"] = int(sum(st["qual_len"] for _, st in sizes))
    if O is not None:  # a whole block of the last timed step against the oracle, every stream byte for byte
        octx = O.OracleCtx(sft, qft)
        e = octx.encode(*blocks[-1])
        g = db[-1].fetch()
        out["oracle_block_all_equal"] = bool(e["rc"] == 0 and all(np.array_equal(g[k], e[k]) for k in ("seq", "qual", "readlens", "n_count", "n_pos")))
        out["oracle_block_raw_bytes"] = int(blocks[-1][0].size)
        octx.close()
    for b in db:
        b.close()
    if decode_mib:  # the serial walk per (block, stream): 4 blocks = 8 chains side by side
        small = make_small() if make_small else make_workload(F, 4 * decode_mib << 20, decode_mib << 20, seed=29, mode=mode)
        sdb = [ctx.dblock(raw, recs) for raw, recs in small]
        for b in sdb:
            b.encode()
        ctx.sync()
        ok = all(b.status()[0] == 0 for b in sdb)
        for b in sdb:
            b.wipe()
        ctx.sync()
        t0 = time.perf_counter()
        ctx.decode_dblocks(sdb)
        ctx.sync()
        dd = time.perf_counter() - t0
        ok = ok and all(b.status()[0] == 0 for b in sdb) and all(np.array_equal(b.fetch_raw(), r) for b, (r, _) in zip(sdb, small))
        out["decode_ns_per_symbol_per_lane"] = round(dd * 1e9 / max(int(r["len"].sum()) for _, r in small), 1)
        out["decode_blocks"] = "4 x %d MiB" % decode_mib
        out["decode_roundtrip_ok"] = bool(ok)
        for b in sdb:
            b.close()
    ctx.close()
    return out


def farm_leg(blocks, workers=16, repeat=4, decompress_workers=4):
    """End to end through the C++ block farm (tools/fqc_tool.cpp over fqcomp28_amd/csrc/process.hpp: the reference's
    processReads / processArchiveParts, src/process.cpp:32-105): the job's blocks `repeat` times over as a FASTQ file in
    /tmp, compressed by `workers` threads (file in, archive out: read, upload, GPU parse, header fields and both streams
    on the GPU, misc coder, write), then once more with --index and restored from that archive.  The clocks are
    fqc_tool's (the workers' loops; tables and handles are built before).  Not part of `value`."""
    import shutil
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "tools", "_build", "fqc_tool")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run(["g++", "-std=c++17", "-O2", "-o", exe, os.path.join(ROOT, "tools", "fqc_tool.cpp"), "-L" + os.path.join(ROOT, "fqcomp28_amd"),
                    "-lfqgpu", "-Wl,-rpath," + os.path.join(ROOT, "fqcomp28_amd"), "-lpthread"], check=True, capture_output=True, timeout=300)
    tmp = tempfile.mkdtemp(prefix="fq_farm_", dir="/tmp")
    try:
        src, arc, back = (os.path.join(tmp, n) for n in ("in.fastq", "a.fqc", "back.fastq"))
        with open(src, "wb") as fh:
            for _ in range(repeat):
                for raw, _recs in blocks:
                    raw.tofile(fh)
        size = os.path.getsize(src)

        def tool(*a):
            r = subprocess.run([exe] + list(a), capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                raise RuntimeError("fqc_tool %s: %s" % (a[0], (r.stdout + r.stderr)[-300:]))
            return json.loads(r.stdout.strip().splitlines()[-1])
        c = tool("c", src, arc, "-t", str(workers))
        plain_bytes = os.path.getsize(arc)
        ci = tool("c", src, arc, "-t", str(workers), "--index")
        d = tool("d", arc, back, "-t", str(decompress_workers))
        same = subprocess.run(["cmp", "-s", src, back]).returncode == 0
        return {"file_MiB": size >> 20, "block_MiB": 256, "blocks": c["blocks"], "workers": workers,
                "compress_MBps": round(size / c["seconds"] / MB, 1), "compress_with_index_MBps": round(size / ci["seconds"] / MB, 1),
                "archive_bytes": plain_bytes, "archive_unchanged_by_index": os.path.getsize(arc) == plain_bytes,
                "decode_index_bytes": os.path.getsize(arc + ".fqx"), "decompress_workers": decompress_workers,
                "decompress_indexed_MBps": round(size / d["seconds"] / MB, 1), "roundtrip_equal": same,
                "what": "fqc_tool c / d on a file in /tmp: read, H2D, GPU parse, header fields + both streams on the GPU, misc coder, write; "
                        "the clock covers the worker threads' loops"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(blocks, sft, qft, seconds_budget=24.0):
    """The oracle ("port" of the reference loops) timed on the host cores, like the reference's
    thread pool: one workspace per thread, whole blocks per thread (src/process.cpp:46-68, 93-104).
    Encode = encodeChunk's seq/qual part; decode = the decodeRecord loops alone."""
    import ctypes as C
    O = _oracle()
    L = O.lib()
    threads = max(1, min(len(os.sched_getaffinity(0)), 64))
    raw0, recs0 = blocks[0]
    # bounded sample: every thread gets one 16 MiB block of the same workload
    per = 16 << 20
    nrec = int(np.searchsorted(recs0["qual_off"], per))
    nrec = max(1, min(nrec, len(recs0)))
    end = int(recs0[nrec - 1]["qual_off"] + recs0[nrec - 1]["len"] + 1)
    raws = [np.array(raw0[:end], copy=True) for _ in range(threads)]
    recs = np.ascontiguousarray(recs0[:nrec])
    nb = int(recs["len"].sum())
    Raw = (C.c_void_p * threads)(*[r.ctypes.data for r in raws])
    Recs = (C.c_void_p * threads)(*[recs.ctypes.data] * threads)
    NR = (C.c_size_t * threads)(*[nrec] * threads)
    NB = (C.c_size_t * threads)(*[nb] * threads)
    out = {}
    for what, mode in (("enc", 0), ("dec", 2)):
        for label, nt, nblk in (("1", 1, 1), ("all", threads, threads)):
            best, reps, t_start = None, 0, time.time()
            while reps < 3 and time.time() - t_start < seconds_budget / 4:
                dt = L.fqo_bench_blocks(O.ptr(sft), O.ptr(qft), nt, nblk, Raw, Recs, NR, NB, mode)
                assert dt > 0
                best = dt if best is None else min(best, dt)
                reps += 1
            out[what + label] = end * nblk / best / MB
    sample = "%d x %.0f MiB blocks of the same config-2 reads, one oracle workspace per thread, best of <=3" % (threads, end / 2**20)
    enc = {"value": round(out["encall"], 1), "unit": "MB/s", "cores": threads, "kind": "port",
           "single_thread_MBps": round(out["enc1"], 1), "sample": "encode of " + sample}
    dec = {"value": round(out["decall"], 1), "unit": "MB/s", "cores": threads, "kind": "port",
           "single_thread_MBps": round(out["dec1"], 1), "sample": "decode (decodeRecord loops alone) of " + sample}
    return enc, dec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--layout", default="auto", choices=["auto", "config1", "strong", "weak"],
                    help="auto: configs[1] at N=1 and on every GPU at N>1 (weak); strong: configs[2], one 1 GiB job in 64 MiB blocks dealt over the ranks")
    ap.add_argument("--mib", type=int, default=1024, help="raw FASTQ of the job (weak layout: per GPU)")
    ap.add_argument("--block-mib", type=int, default=None, help="-R of the reference (default 256; 64 in the strong layout)")
    ap.add_argument("--sample-mib", type=int, default=128, help="-S of the reference")
    ap.add_argument("--decode-block-mib", type=int, default=1, help="block size of the many-blocks decode run")
    ap.add_argument("--decode-mib", type=int, default=256, help="data decoded in the many-blocks run")
    ap.add_argument("--lanes", type=int, default=0, help="blocks in flight per GPU (encode lanes); 0 = the library's default: four, six for blocks below 48 M symbols")
    ap.add_argument("--seq-mode", default="sets", choices=["sets", "generic"],
                    help="sequence chain kernels: segment functions over state sets (default), reset-cut kernel")
    ap.add_argument("--seq-segment", type=int, default=None, help="segment length of the sequence chain kernels")
    ap.add_argument("--segment", type=int, default=0, help="segment length of the generic (quality) chain kernels")
    ap.add_argument("--all-on-gpu0", action="store_true", help="rehearsal of the multi-rank path on a one-GPU box")
    ap.add_argument("--index-stride", type=int, default=1 << 20, help="symbols between the snapshots of the decode index")
    ap.add_argument("--host-threads", type=int, default=4, help="worker threads of the host-pointer measurement")
    ap.add_argument("--skip-decode", action="store_true")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--skip-host", action="store_true")
    ap.add_argument("--skip-farm", action="store_true", help="no end-to-end run of the C++ block farm (4 GiB file in /tmp)")
    ap.add_argument("--all-legs", action="store_true", help="N > 1: also the rank-0 legs that describe ONE GPU and its host (CPU baseline and oracle check, "
                    "host-pointer path, other data); by default they run at N = 1 only")
    ap.add_argument("--skip-other-data", action="store_true", help="no binned / constant / configs[3] runs after the timed region")
    ap.add_argument("--skip-strong", action="store_true", help="no configs[2] run (ONE job in 64 MiB blocks dealt over the ranks) after the timed region")
    args = ap.parse_args()

    # `--gpus N` without a launcher: N fresh processes of this script, before anything here touches a GPU
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, args.all_on_gpu0, sys.argv[1:]))
    if os.environ.get("FQ_BENCH_SPAWN_ONLY"):  # CPU test of the launcher: what a rank was started with
        print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}), flush=True)
        return

    from fqcomp28_amd import farm
    rank, world, local = farm.dist_env()
    if world != max(args.gpus, 1) and rank == 0:
        sys.stderr.write("bench.py: note: --gpus %d but the launcher started %d rank(s); the launcher's count is used\n" % (args.gpus, world))
    dist = farm.init_dist()  # gloo; before anything touches the GPU
    import torch
    if args.all_on_gpu0:
        local = 0
    torch.cuda.set_device(local)
    import fqcomp28_amd as F
    assert F.device_count() > local, "bench.py needs a GPU (no CPU fallback)"
    device = local
    layout = args.layout if args.layout != "auto" else ("config1" if world == 1 else "weak")
    block_mib = args.block_mib or (64 if layout == "strong" else 256)
    if world > 1 and not args.all_legs:
        # what describes one GPU and the box's host cores is measured at N = 1 (the CPU baseline is "rank 0 at N = 1 only"):
        # the N > 1 runs keep to the timed region, the configs[2] job and the decode of the archive dealt over the ranks
        args.skip_cpu = args.skip_host = args.skip_other_data = True

    def barrier():
        farm.barrier(dist)
        torch.cuda.synchronize()

    # ---- workload resident in HBM before the timed region
    t0 = time.time()
    if layout == "strong":
        job = make_workload(F, args.mib << 20, block_mib << 20, seed=28)   # the same job on every rank
        sft = qft = None
        if rank == 0:
            sft, qft = sample_tables(F, job, args.sample_mib << 20, device)
        sft, qft = farm.broadcast_tables(sft, qft, dist)
        blocks = [job[b] for b in farm.shard_blocks(len(job), rank, world)]
        job_blocks = len(job)
        del job
    else:
        blocks = make_workload(F, args.mib << 20, block_mib << 20, seed=28 + (rank if layout == "weak" else 0))
        if layout == "weak" and world > 1:  # one job, one sample: rank 0's tables for every rank (src/archive.cpp:17-24 analyses one sample)
            sft = qft = None
            if rank == 0:
                sft, qft = sample_tables(F, blocks, args.sample_mib << 20, device)
            sft, qft = farm.broadcast_tables(sft, qft, dist)
        else:
            sft, qft = sample_tables(F, blocks, args.sample_mib << 20, device)
        job_blocks = len(blocks) * world
    ctx = F.Context(sft, qft, device=device)
    ctx.set_lanes(max(0, min(args.lanes, 8)))
    ctx.set_chain_params(args.segment, seq_generic=args.seq_mode == "generic", seq_segment=args.seq_segment)
    dblocks = [ctx.dblock(raw, recs) for raw, recs in blocks]
    raw_bytes = sum(raw.size for raw, _ in blocks)
    n_recs = sum(len(r) for _, r in blocks)
    n_bases = sum(int(r["len"].sum()) for _, r in blocks)
    setup_s = time.time() - t0

    def step():
        for b in dblocks:
            b.encode()

    # Warm-up, then ONE more untimed step with HIP events around EVERY kernel group: it gives the
    # table of all groups (kernels_ms) and names the dominant one.  The timed steps then carry events
    # around that group only (two per launch, on the stream it is launched on): events between all
    # kernels of a stream cost 2-3 % of the step.
    def spans_of():
        tot_ms, spans = ctx.last_timing()
        return {name: ms / max(n, 1) for name, ms, n in spans}, {name: n for name, ms, n in spans}
    for _ in range(args.warmup):
        step()
    ctx.sync()
    ctx.enable_timing(True)
    step()
    ctx.sync()
    kern_all, _ = spans_of()
    # The roofline kernel is the one the PROFILE names: largest total rocprofv3 duration in the committed kernel-trace
    # summary of this device code (profiles/<round>_traffic.json, bound by the hash over the device sources).  The
    # HIP-event table of this one untimed step only decides without such a profile: with four blocks in flight its
    # spans move by +-30 % from run to run.
    profile = load_profile(block_mib)
    dom_from = "largest total rocprofv3 duration in profiles/%s (same device code)" % PROFILE_TRAFFIC
    dom_name = None
    if profile:
        ranked = sorted(((k, v.get("rocprof_total_ms", 0.0)) for k, v in profile["kernels"].items() if k in kern_all), key=lambda kv: -kv[1])
        if ranked and ranked[0][1] > 0:
            dom_name = ranked[0][0]
    if dom_name is None:
        dom_name = max(kern_all.items(), key=lambda kv: kv[1])[0] if kern_all else None
        dom_from = "HIP-event spans of one untimed step (no committed profile of this device code)"
    ctx.enable_timing(True, only=None if os.environ.get("FQ_BENCH_ALL_EVENTS") else dom_name)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    # device time of the dominant group over the timed steps (or of every group without a warm-up step)
    kern_timed, calls = spans_of()
    kern = dict(kern_all)
    kern.update(kern_timed)
    ctx.enable_timing(False)
    elapsed = farm.reduce_max(elapsed, dist)
    total_raw = farm.reduce_sum(float(raw_bytes), dist)
    enc_MBps = total_raw * args.steps / elapsed / MB

    sizes = []
    for b in dblocks:
        rc, st = b.status()
        assert rc == 0, rc
        sizes.append(st)
    seq_bytes = sum(s["seq_len"] for s in sizes)
    qual_bytes = sum(s["qual_len"] for s in sizes)
    npos_bytes = 2 * sum(s["n_pos_len"] for s in sizes)
    longest = [b.longest_chain() for b in dblocks]

    # ---- the timed output against the oracle (rank 0): one WHOLE block of the last timed step, byte
    # for byte; the size ratio against the reference coder follows from the same oracle run
    check = None
    if rank == 0 and dblocks and not args.skip_cpu:
        O = _oracle()
        octx = O.OracleCtx(sft, qft)
        raw_c, recs_c = blocks[-1]
        t0 = time.perf_counter()
        e = octx.encode(raw_c, recs_c)
        t_or = time.perf_counter() - t0
        g = dblocks[-1].fetch()
        same = {k: bool(np.array_equal(g[k], e[k])) for k in ("seq", "qual", "readlens", "n_count", "n_pos")}
        check = {"block": len(dblocks) - 1, "block_raw_bytes": int(raw_c.size), "oracle_rc": int(e["rc"]),
                 "streams_equal": same, "all_equal": all(same.values()),
                 "gpu_bytes": int(g["seq"].size + g["qual"].size), "oracle_bytes": int(e["seq"].size + e["qual"].size),
                 "oracle_encode_s": round(t_or, 2)}
        octx.close()

    # ---- roofline of the dominant kernel: algorithmic bytes of ONE block's
    # stream = symbols read once + stream bytes written once; the whole-block figure alongside
    last = sizes[-1]
    last_raw = blocks[-1][0].size
    last_bases = int(blocks[-1][1]["len"].sum())
    alg_block = last_raw + last["seq_len"] + last["qual_len"] + 2 * len(blocks[-1][1]) * 2 + 2 * last["n_pos_len"]
    # the group named by the warm-up table, with its duration over the TIMED steps
    if dom_name and dom_name in kern_timed:
        dom = (dom_name, kern_timed[dom_name])
    else:
        dom = max(kern.items(), key=lambda kv: kv[1]) if kern else ("none", 0.0)
    stream_out = last["qual_len"] if dom[0].startswith("qual") else last["seq_len"]
    alg_dom = last_bases + stream_out if "." in dom[0] else alg_block
    dom_s = dom[1] / 1e3
    achieved = alg_dom / dom_s / 1e9 if dom_s > 0 else 0.0
    # HBM bytes of that kernel: bench.py cannot collect counters itself (separate rocprofv3 --pmc
    # passes of this script, tools/refresh_profiles.py).  The committed figure is quoted only while
    # the device code is the code it was measured on; otherwise null.
    traffic, rocprof_avg_ms = None, None
    traffic_src = profile_stamp()
    if profile:
        k = profile["kernels"].get(dom[0])
        if k:
            traffic = k.get("traffic_bytes_per_launch")
            rocprof_avg_ms = k.get("rocprof_avg_launch_ms")
        traffic_src["block_traffic_bytes"] = profile.get("block_traffic_bytes")
    # One figure per key: `achieved` / `frac` divide by the kernel's own duration -- the rocprofv3 launch
    # time of the committed profile while the device code is the code it was taken on, else (null
    # profile) the HIP-event span measured here; the span (which also holds the wait for room on a chip
    # busy with the other lanes' kernels) is always given beside it as frac_event_span.
    span_achieved = achieved
    if rocprof_avg_ms:
        achieved = alg_dom / (rocprof_avg_ms / 1e3) / 1e9
    roofline = {"bound": "hbm", "kernel": dom[0], "kernel_chosen_by": dom_from,
                "rocprof_kernel": (profile["kernels"].get(dom[0], {}).get("rocprof_kernel") if profile else None),
                "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 5), "frac_from": "rocprofv3 launch time of the committed profile (same device code)" if rocprof_avg_ms
                else "HIP-event span of this run (no committed profile of this device code)",
                "frac_event_span": round(span_achieved / 8000.0, 5), "achieved_event_span": round(span_achieved, 2),
                "traffic": traffic, "traffic_from_committed_profile": traffic_src,
                "rocprof_avg_launch_ms": rocprof_avg_ms,
                "event_span_ms": round(dom[1], 4), "algorithmic_bytes_per_launch": int(alg_dom),
                "launches_timed": calls.get(dom[0], 0),
                "job_GBps": round(alg_block * len(blocks) * args.steps / elapsed / 1e9, 2),
                "job_frac": round(alg_block * len(blocks) * args.steps / elapsed / 1e9 / 8000.0, 5),
                "kernels_ms_from": "one untimed step after the warm-up (events around every kernel group); the roofline kernel: the timed steps",
                "kernels_ms": {k: round(v, 4) for k, v in sorted(kern.items(), key=lambda kv: -kv[1])}}

    extra = {}
    # ---- decode (after the timed region): the same archive, blocks dealt over the ranks exactly as
    # they were coded (configs[4]); then extensions: decode index, many small blocks
    if not args.skip_decode:
        for b in dblocks:
            b.wipe()
        ctx.sync()
        ctx.enable_timing(True)
        barrier()
        t0 = time.perf_counter()
        ctx.decode_dblocks(dblocks)
        ctx.sync()
        barrier()
        dt = farm.reduce_max(time.perf_counter() - t0, dist)
        dkern, _ = spans_of()
        ctx.enable_timing(False)
        ok = True
        for b in dblocks:
            rc, _ = b.status()
            ok = ok and rc == 0
        ok = ok and bool(np.array_equal(dblocks[-1].fetch_raw(), blocks[-1][0]))  # a whole block, byte for byte
        extra["decode_MBps"] = round(total_raw / dt / MB, 1)
        extra["decode_blocks_per_gpu"] = len(dblocks)
        extra["decode_roundtrip_ok"] = bool(farm.reduce_sum(0.0 if ok else 1.0, dist) == 0.0)
        # decode roofline: the one kernel that matters; algorithmic bytes = streams read + read bytes written
        d_ms = dkern.get("decode", 0.0)
        d_alg = seq_bytes + qual_bytes + 2 * n_bases
        extra["decode_roofline"] = {"bound": "hbm", "kernel": "decode (k_decode: one lane per (block, stream))",
                                    "achieved": round(d_alg / (d_ms / 1e3) / 1e9, 3) if d_ms else None, "peak": 8000.0, "unit": "GB/s",
                                    "frac": round(d_alg / (d_ms / 1e3) / 1e9 / 8000.0, 7) if d_ms else None,
                                    "launch_ms": round(d_ms, 2), "algorithmic_bytes_per_launch": int(d_alg),
                                    "lanes": 2 * len(dblocks), "ns_per_symbol_per_lane": round(d_ms * 1e6 / max(1, max(int(r["len"].sum()) for _, r in blocks)), 1),
                                    "traffic": None,
                                    "limit": "not bytes: the format leaves one serial chain per (block, stream); one wave walks it and is "
                                             "bound by the chain of a symbol (nine instructions and an LDS round trip) and by its own instruction issue, 31 instructions per symbol (DESIGN.md section 5)"}
        # extension: the same blocks coded with a decode index (identical streams + a sidecar of
        # snapshots every --index-stride symbols), decoded with one lane per (stream, stride)
        ctx.set_index_stride(args.index_stride)
        ctx.sync()
        t0 = time.perf_counter()
        for b in dblocks:
            b.encode(flags=F.F_DECODE_INDEX)
        ctx.sync()
        t_enc_ix = time.perf_counter() - t0
        ix_bytes = sum(b.fetch_index(0).size + b.fetch_index(1).size for b in dblocks)
        same = all(b.status()[1]["seq_len"] == s["seq_len"] and b.status()[1]["qual_len"] == s["qual_len"]
                   for b, s in zip(dblocks, sizes))
        for b in dblocks:
            b.wipe()
        ctx.sync()
        barrier()
        t0 = time.perf_counter()
        ctx.decode_dblocks(dblocks)
        ctx.sync()
        barrier()
        dt_ix = farm.reduce_max(time.perf_counter() - t0, dist)
        ok_ix = all(b.status()[0] == 0 for b in dblocks) and \
            bool(np.array_equal(dblocks[-1].fetch_raw(), blocks[-1][0]))
        extra["decode_with_index"] = {"MBps": round(total_raw / dt_ix / MB, 1), "stride_symbols": args.index_stride,
                                      "index_bytes_per_gpu": int(ix_bytes),
                                      "index_over_streams": round(ix_bytes / max(1, seq_bytes + qual_bytes), 5),
                                      "encode_with_index_ms": round(t_enc_ix * 1e3, 2), "streams_unchanged": bool(same),
                                      "roundtrip_ok": ok_ix}
        for b in dblocks:
            b.close()
        dblocks = []
        small = make_workload(F, args.decode_mib << 20, args.decode_block_mib << 20, seed=28 + rank)
        sdb = [ctx.dblock(raw, recs) for raw, recs in small]
        for b in sdb:
            b.encode()
        ctx.sync()
        for b in sdb:
            rc, _ = b.status()
            assert rc == 0, rc
            b.wipe()
        ctx.sync()
        barrier()
        t0 = time.perf_counter()
        ctx.decode_dblocks(sdb)
        ctx.sync()
        barrier()
        dt = farm.reduce_max(time.perf_counter() - t0, dist)
        small_raw = farm.reduce_sum(float(sum(r.size for r, _ in small)), dist)
        ok = all(b.status()[0] == 0 for b in sdb) and bool(np.array_equal(sdb[-1].fetch_raw(), small[-1][0]))
        extra["decode_small_blocks_MBps"] = round(small_raw / dt / MB, 1)
        extra["decode_small_blocks"] = {"blocks_per_gpu": len(sdb), "block_MiB": args.decode_block_mib,
                                        "roundtrip_ok": ok}
        for b in sdb:
            b.close()
    for b in dblocks:
        b.close()
    dblocks = []
    # The timed handle goes before the next one comes: a handle created behind another shares the first one's hardware
    # queues (GPU_MAX_HW_QUEUES = 16: streams on one queue run back to back) and runs 5-9 % slower for it -- until round 4's
    # last day every later leg of this process paid that (tools/second_handle_probe.py).
    ctx.close()
    ctx = None

    # ---- BASELINE configs[2] at every N (after the timed region; all ranks): the reference's own scaling config
    if not args.skip_strong and layout != "strong":
        sc2 = strong_config2(F, farm, dist, rank, world, device, max(0, min(args.lanes, 8)), max(3, args.steps), 1024, 64, args.sample_mib,
                             None if args.skip_cpu else _oracle())
        extra["strong_config2_MBps"] = sc2["MBps"]
        extra["strong_config2"] = sc2
        if world == 1:
            extra["encode_config3_MBps"] = sc2["MBps"]   # 16 x 64 MiB on one GPU: the same run under the name the blocks' size gives it


    # ---- the callers either side of the path (not part of `value`): the host-pointer call incl.
    # PCIe -- what CompressionWorkspace::encodeChunk gets through the shim -- from T worker threads with
    # one handle each (reference: one workspace per thread, src/process.cpp:49-54), page-locked chunk
    # and stream buffers; and the GPU record parser against the host parser
    if rank == 0 and not args.skip_host:
        T = max(1, args.host_threads)
        per_thread = max(1, (len(blocks) * 4 + T - 1) // T)   # the job's blocks four times over in total
        ctxs = [F.Context(sft, qft, device=device) for _ in range(T)]
        hctx = ctxs[0]
        for c in ctxs:
            c.set_lanes(1)
        pins = []
        for t in range(T):
            raw_t, recs_t = blocks[t % len(blocks)]
            pr = F.pinned_empty(raw_t.size)
            pr[:] = raw_t
            nb = int(recs_t["len"].sum())
            # block and streams page-locked, the small side buffers pageable: what the C++ shim's FastqChunk /
            # CompressedBuffers hold (workspace.hpp says why the side buffers are better left pageable)
            bufs = dict(seq=F.pinned_empty(F.bound_seq(nb)), qual=F.pinned_empty(F.bound_qual(nb)),
                        readlens=np.zeros(len(recs_t), np.uint16), n_count=np.zeros(len(recs_t), np.uint16),
                        n_pos=np.zeros(nb + 1, np.uint16))
            pins.append((pr, recs_t, bufs))
        errs = []

        def work(t, n):
            pr, recs_t, bufs = pins[t]
            for _ in range(n):
                got = ctxs[t].encode_block_into(pr, recs_t, bufs)
                if got[0] != 0:
                    errs.append(got[0])
        for t in range(T):   # first calls grow the handles' staging blocks and lane scratch
            work(t, 1)
        th = [threading.Thread(target=work, args=(t, per_thread)) for t in range(T)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        dt = time.perf_counter() - t0
        assert not errs, errs
        coded = sum(pins[t][0].size for t in range(T)) * per_thread
        extra["host_pointer_encode_MBps"] = round(coded / dt / MB, 1)
        # one thread, pageable buffers: what a caller that changes nothing gets
        raw0 = np.array(blocks[0][0], dtype=np.uint8, copy=True)
        recs0 = blocks[0][1]
        bufs0 = hctx.host_buffers(len(recs0), int(recs0["len"].sum()))
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            got = hctx.encode_block_into(raw0, recs0, bufs0)
            d1 = time.perf_counter() - t0
            assert got[0] == 0
            best = d1 if best is None else min(best, d1)
        # a host-pointer result of the T-thread run against the resident path's streams of the same block, byte for byte
        t_chk = 0
        hp_bufs = pins[t_chk][2]
        ref_b = hctx.dblock(*blocks[t_chk % len(blocks)])
        ref_b.encode()
        ref = ref_b.fetch()
        ref_b.close()
        same_hp = bool(all(np.array_equal(np.asarray(hp_bufs[k])[: len(ref[k])], ref[k]) for k in ("seq", "qual", "readlens", "n_count")))
        extra["host_pointer"] = {"threads": T, "handles": T, "blocks_coded": T * per_thread, "buffers": "block and streams page-locked (fqgpu_host_alloc), side buffers pageable",
                                 "includes": "H2D of the block + record table, encode, D2H of both streams and the side streams",
                                 "one_thread_pageable_MBps": round(raw0.size / best / MB, 1),
                                 "streams_equal_resident_path": same_hp}

        t0 = time.perf_counter()
        hr = F.parse_fastq(raw0)
        t_host = time.perf_counter() - t0
        t0 = time.perf_counter()
        pb = hctx.dblock(raw0)  # H2D + newline scan + record table on the GPU
        hctx.sync()
        t_gpu = time.perf_counter() - t0
        ok = bool(np.array_equal(pb.records(), hr))
        pb.close()
        extra["parser"] = {"host_parse_MBps": round(raw0.size / t_host / MB, 1),
                           "gpu_create_from_raw_MBps_incl_h2d": round(raw0.size / t_gpu / MB, 1), "tables_equal": ok}
        for c in ctxs:
            c.close()
        if layout == "config1" and not args.skip_farm:
            try:
                extra["farm"] = farm_leg(blocks)
                extra["farm_compress_MBps"] = extra["farm"]["compress_MBps"]
            except Exception as e:   # (a box without g++ or without room in /tmp: the line says so, the run goes on)
                extra["farm"] = {"error": "%s: %s" % (type(e).__name__, str(e)[-300:])}

    # ---- data that is not the headline config (rank 0, after the timed region; not part of `value`)
    if rank == 0 and not args.skip_other_data:
        Oc = None if args.skip_cpu else _oracle()
        lanes = max(0, min(args.lanes, 8))
        od = [other_data(F, Oc, device, lanes, 5, "binned qualities: four levels at 5/10/15/70 %, kept w.p. 0.85 (synth mode 3)", 3, 1024, 256, 128, 16),
              other_data(F, Oc, device, lanes, 5, "constant: every base A, every quality F (synth mode 5)", 5, 1024, 256, 128, 16),
              other_data(F, Oc, device, lanes, 5, "BASELINE configs[3]: 256 MiB, length U[50,300], 1 % N (synth mode 4), in -R 64 blocks", 4, 256, 64, 128, 0)]
        od.append(other_data(F, Oc, device, lanes, 5, "two quality levels '-' / 'F' i.i.d. at 30/70 % (synth mode 6): every quality segment needs its full "
                             "entry-state -> exit-state function (the last opaque class)", 6, 1024, 256, 128, 0, classes=True))
        extra["encode_two_levels_MBps"] = od[3]["MBps"]
        real_blocks = make_real_workload(F, 1024 << 20, 256 << 20)
        od.append(other_data(F, Oc, device, lanes, 5, "REAL reads: the reference's 2 851 test reads of SRR065390 (100 bp, N runs, '#' tails) tiled to 4 x 256 MiB "
                             "with fresh read ids, tables from the first 128 MiB", None, 1024, 256, 128, 16, blocks=real_blocks, classes=True))
        del real_blocks
        extra["encode_real_MBps"] = od[4]["MBps"]
        extra["decode_real_ns_per_symbol"] = od[4].get("decode_ns_per_symbol_per_lane")
        extra["encode_binned_MBps"], extra["encode_constant_MBps"], extra["encode_config4_MBps"] = (o["MBps"] for o in od[:3])
        extra["decode_binned_ns_per_symbol"] = od[0].get("decode_ns_per_symbol_per_lane")
        extra["decode_constant_ns_per_symbol"] = od[1].get("decode_ns_per_symbol_per_lane")
        extra["other_data"] = od

    cpu = cpu_dec = None
    if rank == 0 and not args.skip_cpu:
        cpu, cpu_dec = cpu_baseline(blocks, sft, qft)

    if rank == 0:
        cfg_name = {"config1": "BASELINE configs[1]: %d MiB synthetic 150 bp reads" % args.mib,
                    "strong": "BASELINE configs[2]: ONE job of %d MiB synthetic 150 bp reads in %d MiB blocks dealt round-robin over %d GPU(s)"
                              % (args.mib, block_mib, world),
                    "weak": "BASELINE configs[1] on each of %d GPU(s) (weak scaling: rank r codes its own %d MiB of synthetic 150 bp reads, one sample's tables from rank 0)"
                            % (world, args.mib)}[layout]
        line = {
            "metric": "encode MB/s (raw FASTQ in)", "value": round(enc_MBps, 1), "unit": "MB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong" if layout == "strong" else "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s (uniform ACGT, Phred~N(34,5) clipped [2,41]), -R %d blocks, tables from the first %d MiB, "
                                   "inputs resident in HBM" % (cfg_name, block_mib, args.sample_mib),
                       "layout": layout, "job_blocks": job_blocks, "blocks_this_gpu": len(blocks),
                       "encode_lanes": ("%d blocks in flight per GPU" % args.lanes) if args.lanes else "library default: four blocks in flight per GPU, six for blocks below 48 M symbols",
                       "records_this_gpu": n_recs, "bases_this_gpu": n_bases,
                       "parallelism": ("block b -> rank b mod %d" % world if layout == "strong" else "every rank its own blocks (%d ranks)" % world)
                                      + ", one process per GPU, tables broadcast over gloo, no data-path collective, no RCCL"},
            "compressed": {"seq_bytes": seq_bytes, "qual_bytes": qual_bytes, "n_pos_bytes": npos_bytes,
                           "ratio_vs_reference": (round(check["gpu_bytes"] / check["oracle_bytes"], 6) if check else None),
                           "ratio_vs_reference_from": "seq+qual bytes of one whole timed block / the CPU oracle's bytes for the same block and tables",
                           "longest_serial_chain": [list(r) for r in longest]},
            "oracle_check": check, "kernel_sources_sha": kernel_sources_sha(),
            "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_decode": cpu_dec, "setup_s": round(setup_s, 1),
        }
        line.update(extra)
        if cpu:
            line["gpu_over_cpu_all_cores"] = round(enc_MBps / cpu["value"], 1)
            line["gpu_over_cpu_1_thread"] = round(enc_MBps / cpu["single_thread_MBps"], 1)
        print(json.dumps(line))
    if dist is not None:
        farm.barrier(dist)  # rank 0 runs the checks and baselines alone: the others wait for it here
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
