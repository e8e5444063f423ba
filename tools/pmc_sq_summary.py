"""Summarise rocprofv3 --pmc SQ_* passes per kernel (full-batch launches only): python tools/pmc_sq_summary.py <out.json> <counter_collection.csv>..."""
import collections, csv, json, sys
import numpy as np

res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[2:]:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r["Kernel_Name"].replace("void ", "").split("<")[0].split("(")[0].replace("dto::", "")
            if not name.startswith("k_"):
                continue
            res[(name, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for (name, grid), d in sorted(res.items()):
    if grid < 60000:          # the bench batch (8192 instances) only; the side measurements use smaller grids
        continue
    out[f"{name} grid={grid}"] = {c: dict(launches=len(v), mean=float(np.mean(v)), median=float(np.median(v)), max=float(np.max(v)))
                                  for c, v in sorted(d.items())}
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, d in out.items():
    print(k)
    for c, v in d.items():
        print(f"   {c:22s} launches {v['launches']:4d} mean {v['mean']:16.0f} median {v['median']:16.0f} max {v['max']:16.0f}")
