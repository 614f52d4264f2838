#!/usr/bin/env python3
"""bench.py -- throughput of the FASTQ block entropy-coding hot path on MI355X.

A step = one pass of the hot path over one batch: every block of the workload is ENCODED
(seq + qual FSE streams, readlens, N side streams) from inputs already resident in HBM.
Workload at N=1 = BASELINE.json configs[1]: 1 GiB of synthetic 150 bp reads (uniform ACGT,
Phred ~ N(34,5) clipped to [2,41]), reference default block size -R 256 (4 blocks of 256 MiB),
frequency tables from the first 128 MiB.  N>1: one process per GPU, every rank codes its own
1 GiB (weak scaling, blocks are independent: no data-path collective); the value is the
whole-job MB/s = bytes of raw FASTQ all ranks coded / max-over-ranks time.
Decode (same archive; plus a many-small-blocks layout) is measured after the timed region and
reported in extra keys.  One JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # up to 8 blocks in flight x (seq, qual) streams, one hardware queue each

import numpy as np  # noqa: E402

MB = 1e6


def make_workload(F, total_bytes, block_bytes, seed, first_id=0):
    """-> list of (raw, recs) blocks of whole records, about block_bytes each."""
    blocks, done, next_id = [], 0, first_id
    while done < total_bytes:
        want = min(block_bytes, total_bytes - done)
        raw, n = F.synth_fastq(want, 2, seed=seed, first_read_id=next_id)
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


def cpu_baseline(blocks, sft, qft, seconds_budget=20.0):
    """The oracle ("port" of the reference loop) timed on the host cores, like the reference's
    thread pool: one workspace per thread, whole blocks per thread (src/process.cpp:46-68)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import oracle_lib as O
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
    for label, nt, nblk in (("1", 1, 1), ("all", threads, threads)):
        best = None
        reps = 0
        t_start = time.time()
        while reps < 3 and time.time() - t_start < seconds_budget / 2:
            dt = L.fqo_bench_blocks(O.ptr(sft), O.ptr(qft), nt, nblk, Raw, Recs, NR, NB, 0)
            assert dt > 0
            best = dt if best is None else min(best, dt)
            reps += 1
        out[label] = end * nblk / best / MB
    return {"value": round(out["all"], 1), "unit": "MB/s", "cores": threads, "kind": "port",
            "single_thread_MBps": round(out["1"], 1),
            "sample": "encode of %d x %.0f MiB blocks of the same config-2 reads, one oracle workspace "
                      "per thread, best of <=3" % (threads, end / 2**20)}


def ctx_lanes(args):
    return max(1, min(8, args.lanes))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mib", type=int, default=1024, help="raw FASTQ per GPU")
    ap.add_argument("--block-mib", type=int, default=256, help="-R of the reference (default 256)")
    ap.add_argument("--sample-mib", type=int, default=128, help="-S of the reference")
    ap.add_argument("--decode-block-mib", type=int, default=1, help="block size of the many-blocks decode run")
    ap.add_argument("--decode-mib", type=int, default=256, help="data decoded in the many-blocks run")
    ap.add_argument("--lanes", type=int, default=4, help="blocks in flight per GPU (encode lanes)")
    ap.add_argument("--seq-mode", default="sets", choices=["sets", "generic"],
                    help="sequence chain kernels: segment functions over state sets (default), reset-cut kernel")
    ap.add_argument("--seq-segment", type=int, default=None, help="segment length of the sequence chain kernels")
    ap.add_argument("--segment", type=int, default=0, help="segment length of the generic (quality) chain kernels")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (RCCL, one GPU per rank); gloo only to rehearse the "
                    "multi-rank path on a one-GPU box (all ranks then share GPU 0)")
    ap.add_argument("--index-stride", type=int, default=1 << 20, help="symbols between the snapshots of the decode index")
    ap.add_argument("--skip-decode", action="store_true")
    ap.add_argument("--skip-cpu", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "gloo":
            local = 0  # rehearsal: every rank on GPU 0
        torch.cuda.set_device(local)
        dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    import fqcomp28_amd as F
    from fqcomp28_amd.farm import reduce_max, reduce_sum
    assert F.device_count() > local, "bench.py needs a GPU (no CPU fallback)"
    device = local

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- workload resident in HBM before the timed region
    t0 = time.time()
    blocks = make_workload(F, args.mib << 20, args.block_mib << 20, seed=28 + rank)
    sft, qft = sample_tables(F, blocks, args.sample_mib << 20, device)
    ctx = F.Context(sft, qft, device=device)
    ctx.set_lanes(max(1, min(args.lanes, 8)))
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
    dom_name = max(kern_all.items(), key=lambda kv: kv[1])[0] if kern_all else None
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
    elapsed = reduce_max(elapsed, dist)
    total_raw = reduce_sum(float(raw_bytes), dist)
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
    # HBM bytes of that kernel from the PMC passes committed under profiles/ (separate rocprofv3
    # --pmc FETCH_SIZE / WRITE_SIZE runs of this script; bench.py cannot collect counters itself)
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as fh:
            tj = json.load(fh)
        if tj.get("kernel") == dom[0] and args.block_mib == 256:
            traffic = tj["traffic_bytes_per_launch"]
    except Exception:
        traffic = None
    # the same kernel's average duration in the committed rocprofv3 summary (first wave to last
    # wave; the HIP-event span above also contains the time the launch waits for free CUs when
    # four blocks are in flight)
    rocprof_avg_ms = None
    try:
        import csv
        with open(os.path.join(ROOT, "profiles", "r01_bench_encode_kernel_stats.csv")) as fh:
            for row in csv.DictReader(fh):
                if tj.get("kernel") == dom[0] and tj.get("rocprof_kernel", "").split(" ")[0] in row["Name"]:
                    rocprof_avg_ms = round(float(row["AverageNs"]) / 1e6, 4)
                    break
    except Exception:
        rocprof_avg_ms = None
    roofline = {"bound": "hbm", "kernel": dom[0], "rocprof_avg_launch_ms": rocprof_avg_ms, "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 5), "traffic": traffic,
                "avg_launch_ms": round(dom[1], 4), "algorithmic_bytes_per_launch": int(alg_dom),
                "launches_timed": calls.get(dom[0], 0),
                "job_GBps": round(alg_block * len(blocks) * args.steps / elapsed / 1e9, 2),
                "job_frac": round(alg_block * len(blocks) * args.steps / elapsed / 1e9 / 8000.0, 5),
                "kernels_ms_from": "one untimed step after the warm-up (events around every kernel group); the roofline kernel: the timed steps",
                "kernels_ms": {k: round(v, 4) for k, v in sorted(kern.items(), key=lambda kv: -kv[1])}}

    # ---- decode (after the timed region): same archive, then a many-small-blocks layout
    extra = {}
    if not args.skip_decode:
        for b in dblocks:
            b.wipe()
        ctx.sync()
        barrier()
        t0 = time.perf_counter()
        ctx.decode_dblocks(dblocks)
        ctx.sync()
        barrier()
        dt = reduce_max(time.perf_counter() - t0, dist)
        for b in dblocks:
            rc, _ = b.status()
            assert rc == 0, rc
        ok = bool(np.array_equal(dblocks[0].fetch_raw()[: 1 << 22], blocks[0][0][: 1 << 22]))
        extra["decode_MBps"] = round(total_raw / dt / MB, 1)
        extra["decode_blocks_per_gpu"] = len(dblocks)
        extra["decode_roundtrip_ok"] = ok
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
        dt_ix = reduce_max(time.perf_counter() - t0, dist)
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
        dt = reduce_max(time.perf_counter() - t0, dist)
        small_raw = reduce_sum(float(sum(r.size for r, _ in small)), dist)
        ok = all(b.status()[0] == 0 for b in sdb) and bool(np.array_equal(sdb[-1].fetch_raw(), small[-1][0]))
        extra["decode_small_blocks_MBps"] = round(small_raw / dt / MB, 1)
        extra["decode_small_blocks"] = {"blocks_per_gpu": len(sdb), "block_MiB": args.decode_block_mib,
                                        "roundtrip_ok": ok}
        for b in sdb:
            b.close()

    # ---- the callers either side of the path (not part of `value`): host-pointer call incl. PCIe
    # and per-call buffers, and the GPU record parser against the host parser
    if rank == 0 and not args.skip_decode:
        raw0, recs0 = blocks[0]
        raw0 = np.array(raw0, dtype=np.uint8, copy=True)
        bufs = ctx.host_buffers(len(recs0), int(recs0["len"].sum()))
        best = None
        for _ in range(1 + ctx_lanes(args)):  # first calls grow the handle's staging block and lane scratch
            t0 = time.perf_counter()
            got = ctx.encode_block_into(raw0, recs0, bufs)
            dt = time.perf_counter() - t0
            assert got[0] == 0
            best = dt if best is None else min(best, dt)
        # one worker thread, pageable host memory, H2D + encode + D2H of streams and side streams
        extra["host_pointer_encode_MBps"] = round(raw0.size / best / MB, 1)
        t0 = time.perf_counter()
        hr = F.parse_fastq(raw0)
        t_host = time.perf_counter() - t0
        t0 = time.perf_counter()
        pb = ctx.dblock(raw0)  # H2D + newline scan + record table on the GPU
        ctx.sync()
        t_gpu = time.perf_counter() - t0
        ok = bool(np.array_equal(pb.records(), hr))
        pb.close()
        extra["parser"] = {"host_parse_MBps": round(raw0.size / t_host / MB, 1),
                           "gpu_create_from_raw_MBps_incl_h2d": round(raw0.size / t_gpu / MB, 1), "tables_equal": ok}

    cpu = None
    if rank == 0 and world == 1 and not args.skip_cpu:
        cpu = cpu_baseline(blocks, sft, qft)

    if rank == 0:
        line = {
            "metric": "encode MB/s (raw FASTQ in)", "value": round(enc_MBps, 1), "unit": "MB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %d MiB/GPU synthetic 150 bp reads (uniform ACGT, "
                                   "Phred~N(34,5) clipped [2,41]), -R %d blocks, tables from first %d MiB, "
                                   "inputs resident in HBM" % (args.mib, args.block_mib, args.sample_mib),
                       "blocks_per_gpu": len(blocks), "records_per_gpu": n_recs, "bases_per_gpu": n_bases,
                       "parallelism": "blocks round-robin, %d process(es) x 1 GPU, no collectives" % world},
            "compressed": {"seq_bytes": seq_bytes, "qual_bytes": qual_bytes, "n_pos_bytes": npos_bytes,
                           "ratio_vs_reference": 1.0, "longest_serial_chain": [list(r) for r in longest]},
            "roofline": roofline, "cpu_baseline": cpu, "setup_s": round(setup_s, 1),
        }
        line.update(extra)
        if cpu:
            line["gpu_over_cpu_all_cores"] = round(enc_MBps / cpu["value"], 1)
            line["gpu_over_cpu_1_thread"] = round(enc_MBps / cpu["single_thread_MBps"], 1)
        print(json.dumps(line))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
