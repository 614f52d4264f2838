"""prints the per-kernel table of a rocprofv3 --kernel-trace --stats --output-format csv run: tools/kstats.py <dir>"""
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
def short(n):
    return re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))
for r in list(csv.DictReader(open(f)))[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print("%-44s calls %4s avg %9.1f us  total %8.2f ms  %5s%%" % (short(r["Name"])[:44], r["Calls"], float(r["AverageNs"]) / 1e3,
          float(r["TotalDurationNs"]) / 1e6, r["Percentage"][:5]))
