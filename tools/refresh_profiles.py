"""Copies the rocprofv3 summaries and bench lines collected under gpurun_out/ (prof4, prof1, pmc_f,
pmc_w, bench_default.log, prof4.log) into profiles/ and prints the figures DESIGN.md quotes."""
import csv, glob, re, collections, json, shutil, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/"
def newest(pat):
    return sorted(glob.glob(pat), key=os.path.getmtime)[-1]
def short(n):
    return re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))
f4 = newest(R + "gpurun_out/prof4/*/*kernel_stats.csv"); f1 = newest(R + "gpurun_out/prof1/*/*kernel_stats.csv")
s4 = list(csv.DictReader(open(f4))); s1 = list(csv.DictReader(open(f1)))
shutil.copy(f4, R + "profiles/r01_bench_encode_kernel_stats.csv"); shutil.copy(f1, R + "profiles/r01_encode_one_lane_kernel_stats.csv")
one = {short(r["Name"]): float(r["AverageNs"]) / 1e6 for r in s1}
for r in s4[:26]:
    n = short(r["Name"])
    print("%-36s calls %4s avg %7.3f ms  (1 lane %7.3f)  %5s%%" % (n[:36], r["Calls"], float(r["AverageNs"]) / 1e6, one.get(n, float("nan")), r["Percentage"][:5]))
F = list(csv.DictReader(open(newest(R + "gpurun_out/pmc_f/*/*counter_collection.csv"))))
W = list(csv.DictReader(open(newest(R + "gpurun_out/pmc_w/*/*counter_collection.csv"))))
def agg(rows):
    a = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        n = short(r["Kernel_Name"]); a[n][0] += 1; a[n][1] += float(r["Counter_Value"])
    return a
fa = agg(F); wa = agg(W); out = []
for n in sorted(fa, key=lambda k: -(fa[k][1] * 2 + wa.get(k, [0, 0])[1])):
    c = fa[n][0]; fk = fa[n][1] / c; wk = wa.get(n, [1, 0])[1] / max(wa.get(n, [1, 0])[0], 1)
    out.append((n, c, fk, wk, (2 * fk + wk) * 1024 / 1e6))
with open(R + "profiles/r01_pmc_hbm_traffic_per_launch.csv", "w") as f:
    f.write("kernel,launches,FETCH_SIZE_KB_per_launch_raw,WRITE_SIZE_KB_per_launch,HBM_MB_per_launch_fetch_doubled\n")
    for o in out: f.write("\"%s\",%d,%.1f,%.1f,%.1f\n" % o)
skip = ("k_hist", "k_build", "k_normalize", "k_fill", "k_log", "k_reset", "k_probe")
print("HBM MB per block:", sum(o[4] * o[1] for o in out if not o[0].startswith(skip)) / 8)
for o in out[:12]: print("%-36s x%3d total(f x2) %8.1f MB  (fetch raw %7.1f write %7.1f)" % (o[0][:36], o[1], o[4], o[2] * 1.024 / 1e3, o[3] * 1.024 / 1e3))
sf = [o for o in out if o[0].startswith("k_seq_setfunc")][0]
json.dump({"kernel": "seq.setfunc", "rocprof_kernel": sf[0], "fetch_size_kb": round(sf[2], 1), "write_size_kb": round(sf[3], 1),
           "traffic_bytes_per_launch": int((2 * sf[2] + sf[3]) * 1024),
           "note": "separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of bench.py --steps 1 --warmup 1 --skip-cpu --skip-decode (256 MiB blocks); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads)"},
          open(R + "profiles/r01_traffic.json", "w"), indent=1)
# wave-cycle breakdown per kernel, one block in flight (SQ counters; quad-cycle units, summed over all launches)
sq = sorted(glob.glob(R + "gpurun_out/pmc_sq/*/*counter_collection.csv"), key=os.path.getmtime)
if sq:
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for r in csv.DictReader(open(sq[-1])):
        n = short(r["Kernel_Name"]); acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": calls[n] += 1
    with open(R + "profiles/r01_pmc_sq_stalls_one_lane.csv", "w") as f:
        f.write("kernel,launches,SQ_WAVE_CYCLES,wait_any_pct,wait_inst_any_pct,wait_inst_lds_pct,active_inst_any_pct,SQ_LDS_IDX_ACTIVE,lds_bank_conflict_pct_of_idx_active,SQ_BUSY_CYCLES\n")
        for n, c in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
            if n.startswith(skip) or c["SQ_WAVE_CYCLES"] < 1e6: continue
            wc = c["SQ_WAVE_CYCLES"]
            f.write("\"%s\",%d,%.4g,%.1f,%.1f,%.1f,%.1f,%.4g,%.1f,%.4g\n" % (n, calls[n], wc, 100 * c["SQ_WAIT_ANY"] / wc, 100 * c["SQ_WAIT_INST_ANY"] / wc,
                    100 * c["SQ_WAIT_INST_LDS"] / wc, 100 * c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_LDS_IDX_ACTIVE"],
                    100 * c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0), c["SQ_BUSY_CYCLES"]))
def last_json(path):
    for l in reversed(open(path).read().splitlines()):
        if l.startswith('{"metric"'): return json.loads(l)
d = last_json(R + "gpurun_out/bench_default.log"); json.dump(d, open(R + "profiles/r01_bench_default_line.json", "w"), indent=1)
e = last_json(R + "gpurun_out/prof4.log"); json.dump(e, open(R + "profiles/r01_bench_encode_line.json", "w"), indent=1)
print("default:", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], d["cpu_baseline"]["value"], d["gpu_over_cpu_all_cores"], "rocprof run:", e["value"])
