"""End-to-end timing of the C++ block farm (tools/fqc_tool.cpp over fqcomp28_amd/csrc/process.hpp):
writes a synthetic config-2 FASTQ file, compresses it with T worker threads, decompresses it, compares.
    python tools/farm_bench.py [MiB] [threads ...]
The clock of fqc_tool covers the worker threads only (tables and handles are built before).
FARM_COMPRESS_ONLY=1 skips the decompression and the comparison (A/B runs of the compressing side:
FQGPU_SHIM_HOST_HEADERS=1 is passed on to the workers); FARM_INDEX=1 compresses with --index (decode indexes in
<archive>.fqx, which the decompression then uses)."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fqcomp28_amd as F
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
threads = [int(x) for x in sys.argv[2:]] or [1, 4, 16]
exe = os.path.join(ROOT, "tools", "_build", "fqc_tool")
os.makedirs(os.path.dirname(exe), exist_ok=True)
subprocess.run(["g++", "-std=c++17", "-O2", "-o", exe, os.path.join(ROOT, "tools", "fqc_tool.cpp"), "-L" + os.path.join(ROOT, "fqcomp28_amd"),
                "-lfqgpu", "-Wl,-rpath," + os.path.join(ROOT, "fqcomp28_amd"), "-lpthread"], check=True)
with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
    src = os.path.join(tmp, "in.fastq")
    done, next_id = 0, 0
    with open(src, "wb") as f:
        while done < mib << 20:
            raw, n = F.synth_fastq(min(64 << 20, (mib << 20) - done), 2, seed=28, first_read_id=next_id)
            raw.tofile(f); next_id += n; done += 64 << 20
    size = os.path.getsize(src)
    for t in threads:
        arc, back = os.path.join(tmp, "a.fqc"), os.path.join(tmp, "back.fastq")
        t0 = time.time()
        for old in (arc + ".fqx", back):
            if os.path.exists(old):
                os.remove(old)
        index = ["--index"] if os.environ.get("FARM_INDEX") else []
        if os.environ.get("FARM_INDEX_STRIDE"):
            index += ["--index-stride", os.environ["FARM_INDEX_STRIDE"]]
        run = subprocess.run([exe, "c", src, arc, "-t", str(t)] + index, capture_output=True, text=True, check=True)
        c = json.loads(run.stdout.splitlines()[-1])
        wall_c = time.time() - t0
        if os.environ.get("FQGPU_SHIM_TRACE"):
            print(run.stderr, flush=True)
        if os.environ.get("FARM_COMPRESS_ONLY"):
            print(json.dumps({"threads": t, "raw_MiB": size >> 20, "blocks": c["blocks"], "compress_workers_s": round(c["seconds"], 3),
                              "compress_MBps": round(size / c["seconds"] / 1e6, 1), "compress_wall_s_incl_analysis_and_io": round(wall_c, 2),
                              "archive_bytes": os.path.getsize(arc), "host_headers": bool(os.environ.get("FQGPU_SHIM_HOST_HEADERS"))}), flush=True)
            continue
        t0 = time.time()
        run = subprocess.run([exe, "d", arc, back, "-t", str(t)], capture_output=True, text=True, check=True)
        d = json.loads(run.stdout.splitlines()[-1])
        if os.environ.get("FQGPU_SHIM_TRACE"):
            print(run.stderr, flush=True)
        wall_d = time.time() - t0
        same = subprocess.run(["cmp", "-s", src, back]).returncode == 0
        print(json.dumps({"threads": t, "raw_MiB": size >> 20, "blocks": c["blocks"], "compress_workers_s": round(c["seconds"], 3),
                          "compress_MBps": round(size / c["seconds"] / 1e6, 1), "compress_wall_s_incl_analysis_and_io": round(wall_c, 2),
                          "decompress_workers_s": round(d["seconds"], 3), "decompress_MBps": round(size / d["seconds"] / 1e6, 1),
                          "archive_bytes": os.path.getsize(arc), "ratio": round(size / os.path.getsize(arc), 3), "roundtrip_equal": same,
                          "decode_index_bytes": os.path.getsize(arc + ".fqx") if index else 0}), flush=True)
