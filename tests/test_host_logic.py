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
