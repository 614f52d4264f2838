"""Block farm: how the independent FASTQ blocks of ONE job are dealt to the GPUs of a node.

The reference's only parallelism is N worker threads pulling whole blocks
(src/process.cpp:46-68, 93-104); fqcomp28_amd/csrc/process.hpp is that pipeline in C++ with
one workspace per (thread, GPU).  This module is the same farm with one PROCESS per GPU
(SURVEY.md 8(e)), which is how bench.py and the driver's multi-GPU runs are launched:

* the frequency tables are per ARCHIVE: rank 0 analyses the sample and
  broadcasts the two FreqTable PODs as bytes (1 084 424 B) -- every rank then builds the same
  CTables/DTables on its own GPU;
* block b goes to rank b mod world; blocks never talk to each other, so there is no data-path
  collective and no RCCL anywhere: the broadcast of the tables, the start/stop barriers and the
  max-over-ranks of the elapsed time go over gloo (CPU tensors);
* results (stream sizes, or the streams themselves) are gathered on rank 0, which is where an
  archive writer would append them in completion order with their chunk index
  (src/archive.cpp:57-106).

`python -m fqcomp28_amd.farm --mib 12 --block-mib 2 --out DIR` runs a rank's share of a small
synthetic job and leaves every block's streams in DIR (tests/test_gpu_farm.py starts two fresh
processes this way and compares the union with the oracle).
"""
import json
import os
import sys
import time

import numpy as np


# ---------------------------------------------------------------- partition
def shard_blocks(n_blocks, rank, world):
    """Indices of the blocks rank `rank` of `world` codes (round-robin, like the
    reference's chunk dispenser when every worker is equally fast)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, n_blocks, world))


def blocks_for_weak_scaling(blocks_per_gpu, world):
    """Weak scaling: per-GPU work is fixed, so the job has blocks_per_gpu * world blocks."""
    return blocks_per_gpu * world


# ---------------------------------------------------------------- process group (gloo only)
def dist_env():
    """(rank, world, local_rank) from the launcher's environment (torch.distributed.run)"""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_dist():
    """gloo process group for world > 1 (MASTER_ADDR / MASTER_PORT from the launcher); None otherwise.
    The data path has no collective, so RCCL is never initialised."""
    rank, world, _ = dist_env()
    if world <= 1:
        return None
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


def _active(dist):
    return dist is not None and dist.is_available() and dist.is_initialized()


def barrier(dist=None):
    if _active(dist):
        dist.barrier()


def reduce_max(value, dist=None):
    """max over ranks of a python float; identity without a process group"""
    if not _active(dist):
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_sum(value, dist=None):
    if not _active(dist):
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def broadcast_tables(sft, qft, dist=None, src=0):
    """Rank `src`'s FreqTable PODs to every rank, as bytes.  sft/qft may be None on the others."""
    from .binding import SEQ_FT_DTYPE, QUAL_FT_DTYPE
    if not _active(dist):
        return sft, qft
    import torch
    n = SEQ_FT_DTYPE.itemsize + QUAL_FT_DTYPE.itemsize
    buf = torch.empty(n, dtype=torch.uint8)
    if dist.get_rank() == src:
        both = np.concatenate([np.ascontiguousarray(sft).view(np.uint8).ravel(), np.ascontiguousarray(qft).view(np.uint8).ravel()])
        buf.copy_(torch.from_numpy(both.copy()))
    dist.broadcast(buf, src=src)
    raw = buf.numpy()
    return (raw[: SEQ_FT_DTYPE.itemsize].copy().view(SEQ_FT_DTYPE),
            raw[SEQ_FT_DTYPE.itemsize:].copy().view(QUAL_FT_DTYPE))


def gather_objects(obj, dist=None, dst=0):
    """list of every rank's `obj` on rank dst (None elsewhere); [obj] without a process group"""
    if not _active(dist):
        return [obj]
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(obj, out, dst=dst)
    return out


# ---------------------------------------------------------------- one job, strong scaling
def make_job(F, total_bytes, block_bytes, seed=28, mode=2):
    """The job's blocks [(raw, recs)], identical on every rank (deterministic generator: a rank
    generates the whole job and keeps its share -- block k's read numbers depend on blocks < k)."""
    blocks, done, next_id = [], 0, 0
    while done < total_bytes:
        want = min(block_bytes, total_bytes - done)
        raw, n = F.synth_fastq(want, mode, seed=seed, first_read_id=next_id)
        if n == 0:
            break
        next_id += n
        done += want
        blocks.append(raw)
    return blocks


def sample_tables(F, blocks, sample_bytes, device):
    """Dataset analysis on the first sample_bytes of the job (reference: --sample-size-Mb 128)."""
    got, parts = 0, []
    for raw in blocks:
        take = min(raw.size, sample_bytes - got)
        parts.append(raw[:take])
        got += take
        if got >= sample_bytes:
            break
    sample = np.concatenate(parts)
    recs = F.parse_fastq(sample)
    return F.freq_tables(sample, recs, device=device)


class RankShare:
    """This rank's part of a job: its blocks resident in HBM, one handle with the job's tables."""

    def __init__(self, F, job_blocks, rank, world, device, sample_bytes, dist=None, lanes=4):
        self.F, self.rank, self.world, self.dist = F, rank, world, dist
        sft = qft = None
        if rank == 0:
            sft, qft = sample_tables(F, job_blocks, sample_bytes, device)
        self.sft, self.qft = broadcast_tables(sft, qft, dist)
        self.ctx = F.Context(self.sft, self.qft, device=device)
        self.ctx.set_lanes(lanes)
        self.mine = shard_blocks(len(job_blocks), rank, world)
        self.raws = [job_blocks[b] for b in self.mine]
        self.recs = [F.parse_fastq(r) for r in self.raws]
        self.dblocks = [self.ctx.dblock(r, rc) for r, rc in zip(self.raws, self.recs)]
        self.raw_bytes = sum(r.size for r in self.raws)

    def encode_all(self):
        for b in self.dblocks:
            b.encode()

    def timed(self, fn, steps=1):
        """barrier, steps x fn, sync, barrier -> max over ranks of the elapsed seconds"""
        self.ctx.sync()
        barrier(self.dist)
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        self.ctx.sync()
        barrier(self.dist)
        return reduce_max(time.perf_counter() - t0, self.dist)

    def close(self):
        for b in self.dblocks:
            b.close()
        self.ctx.close()


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="one rank's share of a small synthetic job; streams to --out")
    ap.add_argument("--mib", type=int, default=12)
    ap.add_argument("--block-mib", type=int, default=2)
    ap.add_argument("--sample-mib", type=int, default=4)
    ap.add_argument("--mode", type=int, default=2)
    ap.add_argument("--device", type=int, default=None, help="default: LOCAL_RANK")
    ap.add_argument("--out", required=True)
    args = ap.parse_args(argv)
    rank, world, local = dist_env()
    dist = init_dist()  # before anything touches the GPU
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import fqcomp28_amd as F
    device = local if args.device is None else args.device
    job = make_job(F, args.mib << 20, args.block_mib << 20, mode=args.mode)
    share = RankShare(F, job, rank, world, device, args.sample_mib << 20, dist)
    dt = share.timed(share.encode_all)
    os.makedirs(args.out, exist_ok=True)
    sizes = {}
    for b, db in zip(share.mine, share.dblocks):
        g = db.fetch()
        for k in ("seq", "qual", "readlens", "n_count", "n_pos"):
            g[k].tofile(os.path.join(args.out, "block_%d.%s" % (b, k)))
        sizes[b] = (int(g["seq"].size), int(g["qual"].size))
    if rank == 0:
        share.sft.tofile(os.path.join(args.out, "tables.seq_ft"))
        share.qft.tofile(os.path.join(args.out, "tables.qual_ft"))
    everyone = gather_objects({"rank": rank, "blocks": share.mine, "sizes": sizes, "pid": os.getpid()}, dist)
    if rank == 0:
        print(json.dumps({"world": world, "n_blocks": len(job), "seconds": dt, "ranks": everyone}))
    share.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
