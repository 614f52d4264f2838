"""Runs HERE after the call: copies the rocprofv3 summaries of tools/profile_round.sh <tag> (gpurun_out/<tag>_*) and the default bench
line (gpurun_out/<tag>_bench_default.log) into profiles/ and prints the figures DESIGN.md quotes.
    python tools/refresh_profiles.py [tag, default r03]
Everything is stamped with the commit it was measured on and with the hash of the device code
(bench.py quotes `traffic` only while that hash is unchanged)."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/"
sys.path.insert(0, R)
import bench  # noqa: E402  (kernel_sources_sha)

TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"


def newest(pat):
    return sorted(glob.glob(pat, recursive=True), key=os.path.getmtime)[-1]


def short(n):
    return re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))


def last_json(path):
    for ln in reversed(open(path).read().splitlines()):
        if ln.startswith('{"metric"'):
            return json.loads(ln)


# bench.py's span labels -> the kernel(s) launched inside them
SPAN_KERNELS = {
    "tile_hist2": ["k_tile_hist2"],
    "seq.scatter": ["k_tile_partition<SeqModel, true>"], "qual.scatter": ["k_tile_partition<QualModel, true>"],
    "seq.setfunc": ["k_seq_setfunc<32u, true>"], "seq.chains": ["k_seq_emit"], "seq.resolve": ["k_seq_resolve<32u>"],
    "qual.stage1": ["k_seg_stage1<QualModel, 32u>"], "qual.heads": ["k_seg_heads<QualModel>"], "qual.walk2": ["k_seg_walk<QualModel, 2>"],
    "qual.scan": ["k_seg_scan<QualModel>"],
    "seq.gatherpack": ["k_tile_gather_pack<SeqModel>"], "qual.gatherpack": ["k_tile_gather_pack<QualModel>"],
}

commit = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=R, capture_output=True, text=True).stdout.strip()
sha = bench.kernel_sources_sha()
f4 = newest(R + "gpurun_out/%s_" % TAG + "prof4/**/*kernel_stats.csv")
f1 = newest(R + "gpurun_out/%s_" % TAG + "prof1/**/*kernel_stats.csv")
shutil.copy(f4, R + "profiles/%s_bench_encode_kernel_stats.csv" % TAG)
shutil.copy(f1, R + "profiles/%s_encode_one_lane_kernel_stats.csv" % TAG)
# the traffic file the box wrote BEFORE its default bench run (tools/make_traffic_json.py): the one the default line quotes
shutil.copy(R + "gpurun_out/%s_traffic.json" % TAG, R + "profiles/%s_traffic.json" % TAG)
shutil.copy(R + "gpurun_out/%s_pmc_hbm_traffic_per_launch.csv" % TAG, R + "profiles/%s_pmc_hbm_traffic_per_launch.csv" % TAG)
tj = json.load(open(R + "profiles/%s_traffic.json" % TAG))
assert tj["kernel_sources_sha"] == sha, "the profiles were taken on other device code than this checkout's"
s4 = list(csv.DictReader(open(f4)))
one = {short(r["Name"]): float(r["AverageNs"]) / 1e6 for r in csv.DictReader(open(f1))}
for r in s4[:28]:
    n = short(r["Name"])
    print("%-40s calls %4s avg %7.3f ms  (1 lane %7.3f)  %5s%%" % (n[:40], r["Calls"], float(r["AverageNs"]) / 1e6, one.get(n, float("nan")), r["Percentage"][:5]))
skip = ("k_hist_", "k_build", "k_normalize", "k_fill", "k_log", "k_reset", "k_probe")

sq = sorted(glob.glob(R + "gpurun_out/%s_" % TAG + "pmc_sq/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
if sq:
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for r in csv.DictReader(open(sq[-1])):
        n = short(r["Kernel_Name"])
        acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            calls[n] += 1
    with open(R + "profiles/%s_pmc_sq_stalls_one_lane.csv" % TAG, "w") as f:
        f.write("kernel,launches,SQ_WAVE_CYCLES,wait_any_pct,wait_inst_any_pct,wait_inst_lds_pct,active_inst_any_pct,SQ_LDS_IDX_ACTIVE,lds_bank_conflict_pct_of_idx_active,SQ_BUSY_CYCLES\n")
        for n, c in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
            if n.startswith(skip) or c["SQ_WAVE_CYCLES"] < 1e6:
                continue
            wc = c["SQ_WAVE_CYCLES"]
            f.write("\"%s\",%d,%.4g,%.1f,%.1f,%.1f,%.1f,%.4g,%.1f,%.4g\n" % (n, calls[n], wc, 100 * c["SQ_WAIT_ANY"] / wc, 100 * c["SQ_WAIT_INST_ANY"] / wc,
                    100 * c["SQ_WAIT_INST_LDS"] / wc, 100 * c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_LDS_IDX_ACTIVE"],
                    100 * c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0), c["SQ_BUSY_CYCLES"]))

ins = sorted(glob.glob(R + "gpurun_out/%s_" % TAG + "pmc_insts/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
if ins:
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for r in csv.DictReader(open(ins[-1])):
        n = short(r["Kernel_Name"])
        acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            calls[n] += 1
    cols = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_WAVES"]
    with open(R + "profiles/%s_pmc_instructions_per_launch.csv" % TAG, "w") as f:
        f.write("kernel,launches,wave_insts_VALU,SALU,LDS,VMEM_RD,VMEM_WR,SMEM,waves\n")
        for n, c in sorted(acc.items(), key=lambda kv: -sum(kv[1][k] for k in cols[:6]) / max(calls[kv[0]], 1)):
            if n.startswith(skip):
                continue
            f.write("\"%s\",%d,%s\n" % (n, calls[n], ",".join("%.4g" % (c[k] / max(calls[n], 1)) for k in cols)))

d = last_json(R + "gpurun_out/%s_bench_default.log" % TAG)
json.dump(d, open(R + "profiles/%s_bench_default_line.json" % TAG, "w"), indent=1)
e = last_json(R + "gpurun_out/%s_" % TAG + "prof4.log")
json.dump(e, open(R + "profiles/%s_bench_encode_line.json" % TAG, "w"), indent=1)
box_sha = e.get("kernel_sources_sha")
print("commit", commit, "kernel sources sha", sha, "(profiled box saw %s)" % box_sha)
assert box_sha in (None, sha), "the profiles were taken on other device code than this checkout's"
# the default line must quote THIS round's file, number for number, and name the kernel the kernel-trace summary names
rk = d["roofline"]["kernel"]
assert d["roofline"]["traffic_from_committed_profile"]["kernel_sources_sha"] == sha
assert d["roofline"]["rocprof_avg_launch_ms"] == tj["kernels"][rk]["rocprof_avg_launch_ms"], (d["roofline"]["rocprof_avg_launch_ms"], tj["kernels"][rk])
assert d["roofline"]["traffic"] == tj["kernels"][rk]["traffic_bytes_per_launch"]
assert d["roofline"]["traffic_from_committed_profile"]["block_traffic_bytes"] == tj["block_traffic_bytes"]
assert tj["kernels"][rk]["rocprof_kernel"].split(",")[0] in short(s4[0]["Name"]) or short(s4[0]["Name"]).startswith("__amd"), (rk, s4[0]["Name"])
print("roofline kernel", rk, "= first row of the kernel-trace summary:", short(s4[0]["Name"]))
print("default:", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"].get("rocprof_avg_launch_ms"), d["cpu_baseline"]["value"],
      d["gpu_over_cpu_all_cores"], "rocprof run:", e["value"])
