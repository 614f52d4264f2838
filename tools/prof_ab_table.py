"""Side-by-side per-kernel table of tools/prof_ab.sh's summaries: python3 tools/prof_ab_table.py <tag> name name ..."""
import csv, glob, re, sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/"
tag, names = sys.argv[1], sys.argv[2:]
def short(n):
    return re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))
def load(d):
    f = sorted(glob.glob(R + "gpurun_out/%s/**/*kernel_stats.csv" % d, recursive=True))[-1]
    return {short(r["Name"]): (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6) for r in csv.DictReader(open(f))}
for lanes in ("4", "1"):
    tabs = [load("%s_%s_%s" % (tag, n, lanes)) for n in names]
    keys = sorted(set().union(*tabs), key=lambda k: -max(t.get(k, (0, 0, 0))[2] for t in tabs))
    print("---- %s lane(s): avg us per launch (calls) | " % lanes + " | ".join(names))
    for k in keys[:26]:
        print("%-38s" % k[:38] + " | ".join("%9.1f (%3d)" % (t.get(k, (0, 0, 0))[1], t.get(k, (0, 0, 0))[0]) for t in tabs))
    print("%-38s" % "sum of totals ms" + " | ".join("%9.1f      " % sum(v[2] for v in t.values()) for t in tabs))
