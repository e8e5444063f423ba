"""Per-kernel statistics (calls, total / average / min / max duration) from a rocprofv3 rocpd SQLite file:
    python tools/rocpd_stats.py results.db [out.csv]"""
import re
import sqlite3
import sys


def stats(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute(f"select s.kernel_name, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
                       f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
    return rows


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"<.*", "", name)
    return name.replace("void ", "").replace("dto::", "").replace("(anonymous namespace)::", "")


if __name__ == "__main__":
    rows = stats(sys.argv[1])
    total = sum(r[2] for r in rows)
    lines = ["kernel,calls,total_ms,avg_ms,min_ms,max_ms,percent"]
    for n, c, t, mn, mx in rows:
        lines.append(f"{short(n)},{c},{t / 1e6:.3f},{t / c / 1e6:.4f},{mn / 1e6:.4f},{mx / 1e6:.4f},{100.0 * t / total:.2f}")
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(txt + "\n")
