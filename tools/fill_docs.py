"""Fills the profile numbers of DESIGN.md and profiles/README.md from the committed profile files of a round
(profiles/<tag>_*), so that every such number in the two documents is one the profile files hold.  A number sits between
two HTML comments, <!--@@KEY@@-->value<!--@@--> (invisible when rendered), so that a later profile round refreshes it in
place; a bare @@KEY@@ in newly written text is turned into that form.
    python3 tools/fill_docs.py [tag, default r04] [range of `value` across boxes, e.g. "99.6-102.9"]"""
import csv
import json
import os
import re
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/"
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"
RANGE = sys.argv[2] if len(sys.argv) > 2 else "?"


def short(n):
    return re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))


d = json.load(open(R + "profiles/%s_bench_default_line.json" % TAG))
e = json.load(open(R + "profiles/%s_bench_encode_line.json" % TAG))
tj = json.load(open(R + "profiles/%s_traffic.json" % TAG))
one = {short(r["Name"]): float(r["AverageNs"]) / 1e6 for r in csv.DictReader(open(R + "profiles/%s_encode_one_lane_kernel_stats.csv" % TAG))}
four = {short(r["Name"]): float(r["AverageNs"]) / 1e6 for r in csv.DictReader(open(R + "profiles/%s_bench_encode_kernel_stats.csv" % TAG))}
mb = {r["kernel"]: float(r["HBM_MB_per_launch_fetch_doubled"]) for r in csv.DictReader(open(R + "profiles/%s_pmc_hbm_traffic_per_launch.csv" % TAG))}
ins = list(csv.DictReader(open(R + "profiles/%s_pmc_instructions_per_launch.csv" % TAG)))
tot = {k: sum(float(r[k]) for r in ins if not r["kernel"].startswith("__amd")) for k in ("wave_insts_VALU", "SALU", "LDS", "VMEM_RD", "VMEM_WR")}


def pair(t, a, b, f="%.2f"):
    return (f % t.get(a, 0.0)) + " / " + (f % t.get(b, 0.0))


def sumk(t, names):
    return sum(t.get(n, 0.0) for n in names)


PS, PQ = "k_tile_partition<SeqModel, true>", "k_tile_partition<QualModel, true>"
GS, GQ = "k_tile_gather_pack<SeqModel>", "k_tile_gather_pack<QualModel>"
QCH = ["k_seg_scan<QualModel>", "k_seg_stage1<QualModel, 32u>", "k_seg_heads<QualModel>", "k_seg_compose<QualModel, 32u>", "k_seg_resolve2<QualModel>",
       "k_seg_resolve3<QualModel>", "k_seg_walk<QualModel, 2>"]
K2 = ["k_group_sum<unsigned short>", "k_group_prefix", "k_ctx_layout", "k_tile_base<unsigned short>"]
od = {o["data"].split(":")[0].split(" ")[0]: o for o in d.get("other_data", [])}
vals = {
    "VALUE": "%.1f" % (d["value"] / 1e3), "MS": "%.2f" % d["ms_per_step"], "RANGE": RANGE,
    "RATIO": "%.1f" % d["gpu_over_cpu_all_cores"], "RATIO1": "%.0f" % d["gpu_over_cpu_1_thread"], "CPU": "%.2f" % (d["cpu_baseline"]["value"] / 1e3),
    "TRAFFIC": "%.2f" % (tj["block_traffic_bytes"] / 1e9), "RK": d["roofline"]["kernel"] + " = " + str(d["roofline"].get("rocprof_kernel")),
    "CONFIG3": "%.1f" % (d["strong_config2_MBps"] / 1e3), "CONFIG4": "%.1f" % (d["encode_config4_MBps"] / 1e3),
    "REAL": "%.1f" % (d["encode_real_MBps"] / 1e3), "TWO": "%.1f" % (d["encode_two_levels_MBps"] / 1e3),
    "BINNED": "%.1f" % (d["encode_binned_MBps"] / 1e3), "CONST": "%.1f" % (d["encode_constant_MBps"] / 1e3),
    "DEC": "%.1f" % d["decode_MBps"], "DECR": str(d["decode_real_ns_per_symbol"]), "DECB": str(d["decode_binned_ns_per_symbol"]),
    "DECC": str(d["decode_constant_ns_per_symbol"]), "HP": "%.1f" % (d["host_pointer_encode_MBps"] / 1e3), "PROFV": "%.1f" % (e["value"] / 1e3),
    "COMMIT": str(tj.get("commit")), "SHA": str(tj.get("kernel_sources_sha")),
    "INSTS": "%.0f M VALU + %.0f M SALU + %.0f M LDS + %.0f M vector-memory" % (tot["wave_insts_VALU"] / 1e6, tot["SALU"] / 1e6, tot["LDS"] / 1e6,
                                                                              (tot["VMEM_RD"] + tot["VMEM_WR"]) / 1e6),
    "NPOS4": "%.2f" % sumk(four, ["k_record_scan<0>", "k_record_scan<1>", "k_npos"]),
    "K1": "%.2f" % one.get("k_tile_hist2", 0), "K14": "%.2f" % four.get("k_tile_hist2", 0), "K1MB": "%.0f" % mb.get("k_tile_hist2", 0),
    "K2MB": "%.0f (both streams)" % (2 * sumk(mb, K2)),
    "K3": pair(one, PS, PQ), "K34": pair(four, PS, PQ), "K3MB": pair(mb, PS, PQ, "%.0f"),
    "SF": "%.2f" % one.get("k_seq_setfunc<32u, true>", 0), "SF4": "%.2f" % four.get("k_seq_setfunc<32u, true>", 0), "SFMB": "%.0f" % mb.get("k_seq_setfunc<32u, true>", 0),
    "RE": "%.2f + %.2f" % (one.get("k_seq_resolve<32u>", 0), one.get("k_seq_emit", 0)), "RE4": "%.2f + %.2f" % (four.get("k_seq_resolve<32u>", 0), four.get("k_seq_emit", 0)),
    "QC": " + ".join("%.2f" % one.get(k, 0) for k in QCH), "QC4": " + ".join("%.2f" % four.get(k, 0) for k in QCH),
    "K6": pair(one, GS, GQ), "K64": pair(four, GS, GQ), "K6MB": pair(mb, GS, GQ, "%.0f"),
    "K7": pair(one, "k_epilogue<SeqModel>", "k_epilogue<QualModel>"),
}
for path in ("DESIGN.md", "profiles/README.md"):
    s = open(R + path).read()
    seen = set()

    def refill(m):
        seen.add(m.group(1))
        return "<!--@@%s@@-->%s<!--@@-->" % (m.group(1), vals.get(m.group(1), m.group(2)))

    s = re.sub(r"<!--@@([A-Z0-9]+)@@-->(.*?)<!--@@-->", refill, s, flags=re.S)
    s = re.sub(r"(?<!<!--)@@([A-Z0-9]+)@@(?!-->)", lambda m: refill(type("M", (), {"group": lambda self, i: (None, m.group(1), m.group(0))[i]})()), s)
    open(R + path, "w").write(s)
    print(path, "filled:", sorted(seen & set(vals)), "unknown keys:", sorted(seen - set(vals)))
