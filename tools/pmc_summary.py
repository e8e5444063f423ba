"""Summarise rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected in separate runs) per kernel.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE/WRITE_SIZE are in KiB and, on gfx950,
FETCH_SIZE reports exactly half of the bytes of a coalesced streaming read (MI355X_MICROARCH.md, HBM section;
re-checked here on k_jac, whose read and write byte counts are known exactly).
usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [note]
"""
import collections
import csv
import json
import sys


def load(path):
    d = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].replace("void ", "").split("<")[0].split("(")[0].replace("dto::", "")
            d[name].append(float(r["Counter_Value"]))
    return d


def main():
    F, W = load(sys.argv[1]), load(sys.argv[2])
    out = {"note": sys.argv[4] if len(sys.argv) > 4 else "", "unit": "bytes per launch (mean over all launches of the kernel)",
           "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024", "kernels": {}}
    for k in sorted(F):
        if not k.startswith("k_"):
            continue
        f, w = F[k], W.get(k, [0.0])
        out["kernels"][k] = dict(launches=len(f), fetch_size_kib_mean=sum(f) / len(f), write_size_kib_mean=sum(w) / len(w),
                                 fetch_size_kib_max=max(f), write_size_kib_max=max(w),
                                 hbm_bytes_per_launch_mean=(2 * sum(f) / len(f) + sum(w) / len(w)) * 1024,
                                 hbm_bytes_per_launch_max=(2 * max(f) + max(w)) * 1024)
    with open(sys.argv[3], "w") as fo:
        json.dump(out, fo, indent=1)
    for k, v in out["kernels"].items():
        print(f"{k:16s} launches {v['launches']:4d}  HBM bytes/launch mean {v['hbm_bytes_per_launch_mean'] / 1e9:8.3f} GB  max {v['hbm_bytes_per_launch_max'] / 1e9:8.3f} GB")


if __name__ == "__main__":
    main()
