"""Per-slice kernel time of a solve from a rocprofv3 kernel trace (tools/trace_phases.sh): one line per 25 iterations."""
import collections
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"].replace("void ", "").replace("dto::", "").split("<")[0].split("(")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
rows.sort()
it, cur, out = 0, collections.Counter(), []
for s, e, n in rows:
    cur[n] += (e - s) * 1e-6
    if n == "k_update":
        it += 1
        if it % 25 == 0:
            out.append((it, dict(cur)))
            cur = collections.Counter()
keys = ["k_stage_eval", "k_kkt_fwd_seq", "k_kkt_fwd", "k_kkt_sep", "k_kkt_bwd_seq", "k_kkt_bwd", "k_linesearch", "k_update", "k_gather_rows"]
print("iters " + " ".join(f"{k[2:][:9]:>9s}" for k in keys) + "     other    sum   (ms per iteration, mean over the slice)")
for it, c in out:
    tot = sum(c.values())
    oth = tot - sum(c.get(k, 0.0) for k in keys)
    print(f"{it:5d} " + " ".join(f"{c.get(k, 0.0) / 25:9.2f}" for k in keys) + f" {oth / 25:9.2f} {tot / 25:7.1f}")
