"""Builds profiles/<tag>_traffic.json from the rocprofv3 passes of tools/profile_round.sh <tag> (gpurun_out/<tag>_prof4,
_prof1, _pmc_f, _pmc_w).  Runs ON THE GPU BOX, between the profile passes and the default bench.py run of the same
call, so that the default line quotes the file of ITS OWN round (round 3's default line quoted an earlier file than
the one committed); a copy goes to gpurun_out/ (the only directory that travels back).
    python3 tools/make_traffic_json.py [tag]
bench.py quotes the file only while the hash over the device sources is the one stamped here."""
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

# the box has no .git: tools/profile_round.sh's caller leaves the commit in profiles/.head_commit
try:
    commit = open(R + "profiles/.head_commit").read().strip()
except OSError:
    commit = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=R, capture_output=True, text=True).stdout.strip() or None
sha = bench.kernel_sources_sha()
f4 = newest(R + "gpurun_out/%s_" % TAG + "prof4/**/*kernel_stats.csv")
f1 = newest(R + "gpurun_out/%s_" % TAG + "prof1/**/*kernel_stats.csv")
s4, s1 = list(csv.DictReader(open(f4))), list(csv.DictReader(open(f1)))
one = {short(r["Name"]): float(r["AverageNs"]) / 1e6 for r in s1}
four = {short(r["Name"]): float(r["AverageNs"]) / 1e6 for r in s4}
four_total = {short(r["Name"]): float(r["TotalDurationNs"]) / 1e6 for r in s4}
four_calls = {short(r["Name"]): int(r["Calls"]) for r in s4}
for r in s4[:28]:
    n = short(r["Name"])
    print("%-40s calls %4s avg %7.3f ms  (1 lane %7.3f)  %5s%%" % (n[:40], r["Calls"], float(r["AverageNs"]) / 1e6, one.get(n, float("nan")), r["Percentage"][:5]))


def agg(path):
    a = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        n = short(r["Kernel_Name"])
        a[n][0] += 1
        a[n][1] += float(r["Counter_Value"])
    return a


fa = agg(newest(R + "gpurun_out/%s_" % TAG + "pmc_f/**/*counter_collection.csv"))
wa = agg(newest(R + "gpurun_out/%s_" % TAG + "pmc_w/**/*counter_collection.csv"))
skip = ("k_hist_", "k_build", "k_normalize", "k_fill", "k_log", "k_reset", "k_probe")
out = []
for n in sorted(fa, key=lambda k: -(fa[k][1] * 2 + wa.get(k, [0, 0])[1])):
    c = fa[n][0]
    fk = fa[n][1] / c
    wk = wa.get(n, [1, 0])[1] / max(wa.get(n, [1, 0])[0], 1)
    out.append((n, c, fk, wk, (2 * fk + wk) * 1024 / 1e6))
with open(R + "gpurun_out/%s_pmc_hbm_traffic_per_launch.csv" % TAG, "w") as f:
    f.write("kernel,launches,FETCH_SIZE_KB_per_launch_raw,WRITE_SIZE_KB_per_launch,HBM_MB_per_launch_fetch_doubled\n")
    for o in out:
        f.write("\"%s\",%d,%.1f,%.1f,%.1f\n" % o)
# the PMC passes run 1 warm-up + 1 table + 1 timed step of 4 blocks = 12 block encodes (memsets are not kernels of ours)
n_block_encodes = max(o[1] for o in out if o[0].startswith("k_tile_partition<QualModel"))
block_mb = sum(o[4] * o[1] for o in out if not o[0].startswith(skip)) / n_block_encodes
print("HBM MB per 256 MiB block (all encode kernels, FETCH doubled):", round(block_mb, 1), "over", n_block_encodes, "block encodes")
for o in out[:14]:
    print("%-40s x%3d  %8.1f MB per launch (fetch raw %7.1f MB, write %7.1f MB)" % (o[0][:40], o[1], o[4], o[2] * 1.024 / 1e3, o[3] * 1.024 / 1e3))
per = {o[0]: o for o in out}
kernels = {}
for span, names in SPAN_KERNELS.items():
    if all(n in per for n in names):
        kernels[span] = {"rocprof_kernel": ", ".join(names),
                         "traffic_bytes_per_launch": int(sum((2 * per[n][2] + per[n][3]) * 1024 for n in names)),
                         "rocprof_avg_launch_ms": round(sum(four.get(n, 0.0) for n in names), 4),
                         "rocprof_total_ms": round(sum(four_total.get(n, 0.0) for n in names), 3),
                         "rocprof_calls": sum(four_calls.get(n, 0) for n in names),
                         "rocprof_avg_launch_ms_one_lane": round(sum(one.get(n, 0.0) for n in names), 4)}
json.dump({"commit": commit, "kernel_sources_sha": sha, "block_mib": 256, "block_traffic_bytes": int(block_mb * 1e6), "kernels": kernels,
           "note": "separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of bench.py --steps 1 --warmup 1 --skip-cpu --skip-decode --skip-host "
                   "(256 MiB blocks); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads); "
                   "average launch times from the --kernel-trace --stats passes of the same commit"},
          open(R + "profiles/%s_traffic.json" % TAG, "w"), indent=1)
shutil.copy(R + "profiles/%s_traffic.json" % TAG, R + "gpurun_out/%s_traffic.json" % TAG)
print("wrote profiles/%s_traffic.json (device code %s); roofline kernel by total duration:" % (TAG, sha),
      max(kernels.items(), key=lambda kv: kv[1]["rocprof_total_ms"])[0])

