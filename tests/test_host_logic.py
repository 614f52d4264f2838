"""CPU-only tests: the C-ABI library loads and exports every symbol include/fqgpu.h declares,
host-side helpers (parser, generator, bounds) behave like the reference's, the block farm shards
correctly across ranks (gloo, world_size 2), and the product path fails loudly without a GPU."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def F():
    import fqcomp28_amd as F
    if not os.path.exists(F.lib_path()):
        F.build()
    return F


def test_library_exports_every_declared_symbol(F):
    hdr = open(os.path.join(ROOT, "include", "fqgpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(fqgpu_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"fqgpu_bound_"}
    from fqcomp28_amd import binding
    assert declared == set(binding.EXPORTS), declared ^ set(binding.EXPORTS)
    L = F.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.fqgpu_version().startswith(b"fqgpu")


def test_bounds_match_reference_rule(F):
    # src/workspace.h:21-35
    for n in (0, 1, 1023, 1024, 1500000, 30 << 20, 480 << 20):
        assert F.bound_seq(n) == O.lib().fqo_bound_seq(n) == (262144 if n < 1024 else n // 4 + 1024)
        assert F.bound_qual(n) == O.lib().fqo_bound_qual(n) == max(8388608, n * 7 // 8 + 1024)


def test_parser_matches_numpy_restatement(F, golden_dir):
    for f in ("SRR065390_sub_1", "without_ns", "SRR065390_1_first5"):
        raw, recs = O.load_fastq(os.path.join(golden_dir, f + ".fastq"))
        assert np.array_equal(F.parse_fastq(raw), recs)
        # a truncated tail is ignored like the reference's partial-record carry-over
        cut = raw[: raw.size - 17]
        got = F.parse_fastq(cut)
        assert np.array_equal(got, recs[: len(got)]) and len(got) == len(recs) - 1
    bad = np.frombuffer(b"@r\nACGT\n-\nIIII\n", dtype=np.uint8)
    with pytest.raises(F.FqgpuError):
        F.parse_fastq(bad)


@pytest.mark.parametrize("mode", [1, 2, 4])
def test_synthetic_generator_is_deterministic_and_well_formed(F, mode):
    a, n = F.synth_fastq(1 << 20, mode, seed=28)
    b, _ = F.synth_fastq(1 << 20, mode, seed=28)
    c, _ = F.synth_fastq(1 << 20, mode, seed=29)
    assert np.array_equal(a, b) and not np.array_equal(a[: c.size], c[: a.size])
    recs = F.parse_fastq(a)
    assert len(recs) == n and np.array_equal(recs, O.parse_fastq(a))
    # blocks generated independently continue the same read stream
    k = n // 2
    first, nk = F.synth_fastq(int(recs[k - 1]["qual_off"] + recs[k - 1]["len"] + 1), mode, seed=28)
    rest, _ = F.synth_fastq(a.size - first.size, mode, seed=28, first_read_id=nk)
    assert nk == k and np.array_equal(np.concatenate([first, rest]), a)
    quals = np.concatenate([a[r["qual_off"]: r["qual_off"] + r["len"]] for r in recs[:500]]).astype(int) - 33
    seqs = np.concatenate([a[r["seq_off"]: r["seq_off"] + r["len"]] for r in recs[:500]])
    assert set(np.unique(seqs)) <= set(b"ACGTN")
    if mode == 1:
        assert (quals == 40).all() and (recs["len"] == 150).all()
    elif mode == 2:
        assert (recs["len"] == 150).all() and 2 <= quals.min() and quals.max() <= 41
        assert abs(quals.mean() - 33.8) < 0.5 and 4 < quals.std() < 5.5 and not (seqs == ord("N")).any()
    else:
        assert recs["len"].min() >= 50 and recs["len"].max() <= 300
        isn = seqs == ord("N")
        assert 0.005 < isn.mean() < 0.02 and (quals[isn] == 2).all()


def test_no_gpu_means_loud_failure_not_fallback(F, golden_dir):
    if F.device_count() > 0:
        pytest.skip("a GPU is visible")
    raw, recs = O.load_fastq(os.path.join(golden_dir, "SRR065390_1_first5.fastq"))
    with pytest.raises(F.FqgpuError) as ei:
        F.freq_tables(raw, recs)
    assert ei.value.code == -5
    _, _, sft, qft = O.freq_tables(raw, recs)
    with pytest.raises(F.FqgpuError) as ei:
        F.Context(sft, qft)
    assert ei.value.code == -5


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "fqcomp28_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_lib" not in txt and "fqc_oracle" not in txt and "fse_oracle" not in txt, f


def test_shard_blocks_partition():
    from fqcomp28_amd.farm import shard_blocks, blocks_for_weak_scaling
    for world in (1, 2, 4, 8):
        for n in (0, 1, 7, 16, 33):
            got = sorted(b for r in range(world) for b in shard_blocks(n, r, world))
            assert got == list(range(n))
        assert blocks_for_weak_scaling(4, world) == 4 * world
    with pytest.raises(ValueError):
        shard_blocks(4, 2, 2)


WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import torch.distributed as dist
import numpy as np
from fqcomp28_amd.farm import shard_blocks, reduce_max, reduce_sum, broadcast_tables, gather_objects, init_dist, barrier
from fqcomp28_amd.binding import SEQ_FT_DTYPE, QUAL_FT_DTYPE
dist = init_dist()
assert dist.get_backend() == "gloo"
r, w = dist.get_rank(), dist.get_world_size()
mine = shard_blocks(9, r, w)
n = reduce_sum(float(len(mine)), dist)
t = reduce_max(1.0 + r, dist)
barrier(dist)
assert n == 9.0 and t == float(w), (n, t)
# one sample's tables, broadcast as bytes from rank 0 (the others start with nothing)
sft = qft = None
if r == 0:
    rng = np.random.default_rng(5)
    sft = rng.integers(0, 256, SEQ_FT_DTYPE.itemsize, dtype=np.uint8).view(SEQ_FT_DTYPE)
    qft = rng.integers(0, 256, QUAL_FT_DTYPE.itemsize, dtype=np.uint8).view(QUAL_FT_DTYPE)
sft, qft = broadcast_tables(sft, qft, dist)
want = np.random.default_rng(5)
assert sft.tobytes() == want.integers(0, 256, SEQ_FT_DTYPE.itemsize, dtype=np.uint8).tobytes()
assert qft.tobytes() == want.integers(0, 256, QUAL_FT_DTYPE.itemsize, dtype=np.uint8).tobytes()
got = gather_objects({"rank": r, "blocks": mine}, dist)
if r == 0:
    assert sorted(b for g in got for b in g["blocks"]) == list(range(9))
else:
    assert got is None
print("rank", r, "ok", mine)
dist.destroy_process_group()
"""


def test_block_farm_two_ranks_gloo(tmp_path):
    """N>1 path: each rank takes its round-robin share, timing is max over ranks, no data-path collective."""
    script = tmp_path / "w.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    assert "[0, 2, 4, 6, 8]" in outs[0] and "[1, 3, 5, 7]" in outs[1]


# ---------------------------------------------------------------- bench.py: launcher, workloads, profile binding (no GPU)
def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_bench_gpus_n_from_a_bare_shell_starts_n_ranks():
    """`python bench.py --gpus 3` without a launcher: the parent starts three fresh processes with the environment
    torch.distributed.run would give them (gloo rendezvous on 127.0.0.1) and never touches a GPU itself; with fewer
    devices visible than ranks asked for it refuses with a non-zero exit instead of coding on one GPU."""
    import json
    env = dict(os.environ, FQ_BENCH_SPAWN_ONLY="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--all-on-gpu0"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    ranks = sorted((json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")), key=lambda d: int(d["RANK"]))
    assert [d["RANK"] for d in ranks] == ["0", "1", "2"] and [d["LOCAL_RANK"] for d in ranks] == ["0", "1", "2"]
    assert all(d["WORLD_SIZE"] == "3" and d["MASTER_ADDR"] == "127.0.0.1" for d in ranks)
    assert len({d["MASTER_PORT"] for d in ranks}) == 1 and 1024 < int(ranks[0]["MASTER_PORT"]) < 65536
    envs = _bench().child_environments(2, 29501, base={"PATH": "/bin"})
    assert [(e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"], e["MASTER_PORT"]) for e in envs] == [("0", "0", "2", "29501"), ("1", "1", "2", "29501")]
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and e["PATH"] == "/bin" for e in envs)
    # no GPU in this container: two ranks on real devices are refused, loudly
    env.pop("FQ_BENCH_SPAWN_ONLY")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "device(s) visible" in r.stderr and not r.stdout.strip()


def test_bench_real_read_workload_is_wellformed_and_tiles_the_reference_reads(F, golden_dir):
    """bench.py's REAL-statistics workload: the reference's 2 851 test reads tiled with fresh ids -- whole records, every
    read one of the source reads, every pass a permutation of all of them, blocks the oracle codes under the capacity rule."""
    bench = _bench()
    blocks = bench.make_real_workload(F, 3 << 20, 1 << 20)
    assert len(blocks) == 3
    src = set()
    for f in bench.REAL_FILES:
        raw, recs = O.load_fastq(os.path.join(golden_dir, f))
        for r in recs:
            src.add((raw[r["seq_off"]: r["seq_off"] + 100].tobytes(), raw[r["qual_off"]: r["qual_off"] + 100].tobytes()))
    seen, ids = [], []
    for raw, recs in blocks:
        assert np.array_equal(O.parse_fastq(raw), recs) and abs(raw.size - (1 << 20)) < 300
        for r in recs:
            seen.append((raw[r["seq_off"]: r["seq_off"] + 100].tobytes(), raw[r["qual_off"]: r["qual_off"] + 100].tobytes()))
        ids += [int(bytes(raw[o - 44 + 11: o - 44 + 21])) for o in recs["seq_off"]]
    assert ids == list(range(1, len(ids) + 1)) and set(seen) <= src
    p1, p2 = seen[2851 - 1: 2 * 2851 - 1], seen[2 * 2851 - 1: 3 * 2851 - 1]   # read ids 2851 .. 5701 and 5702 .. 8552: two whole passes
    assert len(seen) > 3 * 2851 and len(set(p1)) == len(src) == len(set(p2)) and p1 != p2
    _, _, sft, qft = O.freq_tables(*blocks[0])
    e = O.OracleCtx(sft, qft).encode(*blocks[1])
    assert e["rc"] == 0 and e["n_pos"].size > 1000   # the N runs of the real reads are there


def test_bench_quotes_a_profile_only_for_the_device_code_it_was_taken_on(tmp_path, monkeypatch):
    bench = _bench()
    import json
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    os.makedirs(tmp_path / "fqcomp28_amd" / "csrc")
    (tmp_path / "fqcomp28_amd" / "csrc" / "a.hip").write_text("kernel v1")
    sha = bench.kernel_sources_sha()
    prof = {"commit": "x", "kernel_sources_sha": sha, "block_mib": 256, "kernels": {"seq.setfunc": {"rocprof_total_ms": 40.0}, "qual.scatter": {"rocprof_total_ms": 30.0}}}
    (tmp_path / "profiles" / bench.PROFILE_TRAFFIC).write_text(json.dumps(prof))
    assert bench.load_profile(256)["kernels"]["seq.setfunc"]["rocprof_total_ms"] == 40.0
    assert bench.load_profile(64) is None                       # another block size: not this profile's kernels
    (tmp_path / "fqcomp28_amd" / "csrc" / "a.hip").write_text("kernel v2")
    assert bench.load_profile(256) is None and bench.profile_stamp()["kernel_sources_sha"] == sha != bench.profile_stamp()["current_kernel_sources_sha"]
