import os, sys, subprocess
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tools"); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/oracle")
import numpy as np
import fqcomp28_amd as F
import soak_farm as S
seed0, case = 7000, int(sys.argv[1])
rng = np.random.default_rng(seed0 + case)
mode = int(rng.choice([2, 2, 3, 4, 4, 5]))
size = int(rng.choice([2000, 50000, 1 << 20, 5 << 20, 24 << 20]) * (0.5 + rng.random()))
raw, _ = F.synth_fastq(size, mode, seed=seed0 + case)
data = raw.tobytes()
bad = None
rew = False
if rng.random() < 0.66 and size < (8 << 20):
    data, bad = S.rewrite_headers(raw, rng); rew = True
print("mode", mode, "size", size, "rewritten", rew, "bad", bad, "first header", data[:80].split(b"\n")[0])
import oracle_lib as O
full = np.frombuffer(data, dtype=np.uint8)
frecs = F.parse_fastq(full)
ends = frecs["qual_off"].astype(np.int64) + frecs["len"] + 1
def chunks(limit):
    out, lo = [], 0
    while lo < len(full):
        k = int(np.searchsorted(ends, lo + limit, side="right"))
        hi = int(ends[k - 1]) if k and ends[k - 1] > lo else int(ends[np.searchsorted(ends, lo, side="right")])
        out.append((lo, hi)); lo = hi
    return out
R, Smib = int(sys.argv[2]), int(sys.argv[3])
s_lo, s_hi = chunks(Smib << 20)[0]
sraw = full[s_lo:s_hi]
_, _, sft, qft = O.freq_tables(sraw, F.parse_fastq(sraw))
octx = O.OracleCtx(sft, qft)
ctx = F.Context(sft, qft); ctx.set_lanes(1)
for lo, hi in chunks(R << 20):
    braw = full[lo:hi].copy(); brecs = F.parse_fastq(braw)
    e = octx.encode(braw, brecs)
    g = ctx.encode_block(braw.copy(), brecs, flags=1)
    g2 = ctx.encode_raw(braw.copy(), flags=1)
    eq = {k: bool(np.array_equal(np.asarray(g[k]), np.asarray(e[k]))) for k in ("seq", "qual", "readlens", "n_count", "n_pos")}
    eq2 = {k: bool(np.array_equal(np.asarray(g2[k]), np.asarray(e[k]))) for k in ("seq", "qual", "readlens", "n_count", "n_pos")}
    rc, out = ctx.decode_block(e["seq"], e["qual"], e["n_count"], e["n_pos"], brecs, O.blank_skeleton(braw, brecs))
    print("block", lo, hi, "recs", len(brecs), "nsym", int(brecs["len"].sum()), "oracle rc", e["rc"], "gpu rc", g["rc"], g2["rc"], eq, eq2, "decode rc", rc, bool(np.array_equal(out, braw)), "lens", len(e["seq"]), len(e["qual"]))
exe = os.path.join(ROOT, "tools", "_build", "fqc_tool")
os.makedirs(os.path.dirname(exe), exist_ok=True)
subprocess.run(["g++", "-std=c++17", "-O2", "-o", exe, os.path.join(ROOT, "tools", "fqc_tool.cpp"), "-L" + os.path.join(ROOT, "fqcomp28_amd"),
                "-lfqgpu", "-Wl,-rpath," + os.path.join(ROOT, "fqcomp28_amd"), "-lpthread"], check=True)
open("/tmp/in.fastq", "wb").write(data)
import fqc_archive as A
import headers_oracle as HO
for env in ({}, {"FQGPU_SHIM_HOST_HEADERS": "1"}):
    e = dict(os.environ, FQGPU_VERBOSE="1", **env)
    c = subprocess.run([exe, "c", "/tmp/in.fastq", "/tmp/a.fqc", "-R", str(R), "-S", str(Smib), "-t", "1"], capture_output=True, text=True, env=e)
    d = subprocess.run([exe, "d", "/tmp/a.fqc", "/tmp/back.fastq", "-t", "1"], capture_output=True, text=True, env=e)
    print(env, "c", c.returncode, c.stderr[-300:], "d", d.returncode, d.stderr[-300:])
    first_header, seq_ft, qual_ft, blocks, _ = A.read_archive("/tmp/a.fqc")
    pos = 0
    for blk in blocks:
        braw = full[pos: pos + blk.total]; pos += blk.total
        brecs = F.parse_fastq(braw)
        e2 = octx.encode(braw, brecs)
        print(" block", blk.idx, blk.total, blk.n_records, len(brecs), "seq eq", bytes(e2["seq"]) == blk.seq, "qual eq", bytes(e2["qual"]) == blk.qual,
              "readlens", blk.readlens[0], len(blk.readlens[1]), "n_count", blk.n_count[0], len(blk.n_count[1]), "n_pos", blk.n_pos[0], len(blk.n_pos[1]))
        for name, (orig, cb), want in (("readlens", blk.readlens, e2["readlens"]), ("n_count", blk.n_count, e2["n_count"]), ("n_pos", blk.n_pos, e2["n_pos"])):
            got = F.memdecompress(np.frombuffer(cb, dtype=np.uint8), orig)
            print("   ", name, got.tobytes() == want.astype("<u2").tobytes())
