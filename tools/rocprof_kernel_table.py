"""Per-kernel table (calls, total ms, average ms, share) from a rocprofv3 --kernel-trace result
database (rocprofv3 7.x writes <name>_results.db unless --output-format csv is given).
    python tools/rocprof_kernel_table.py <results.db> [rows]"""
import re
import sqlite3
import sys


def table(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    disp = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0]
    sym = disp.replace("rocpd_kernel_dispatch", "rocpd_info_kernel_symbol")
    q = ("select s.kernel_name, count(*), sum(d.end - d.start) / 1e6, avg(d.end - d.start) / 1e6 from %s d join %s s "
         "on d.kernel_id = s.id group by s.kernel_name order by 3 desc" % (disp, sym))
    return list(cur.execute(q))


def short(name):
    name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)
    name = re.sub(r"\.kd$", "", name)
    m = re.match(r"(k_[a-z0-9_]+?)(I.*?)?E(Ev|v)", name)
    if m:
        tags = re.findall(r"(Seq|Qual)Model|L[bij](\d+)", name[: m.end()])
        extra = ",".join(t[0] or t[1] for t in tags)
        return m.group(1) + ("<" + extra + ">" if extra else "")
    return name[:48]


if __name__ == "__main__":
    rows = table(sys.argv[1])
    lim = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    tot = sum(r[2] for r in rows)
    print("%-44s %6s %10s %9s %6s" % ("kernel", "calls", "total_ms", "avg_ms", "share"))
    for name, n, t, a in rows[:lim]:
        print("%-44s %6d %10.3f %9.4f %5.1f%%" % (short(name), n, t, a, 100 * t / tot))
