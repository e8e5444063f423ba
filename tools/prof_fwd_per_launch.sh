#!/bin/bash
# Per-launch durations of the headline loop's kernels from a rocprofv3 kernel trace of `bench.py --loop-only --steps 20 --warmup 5`
# (the driver's command without the side measurements): bench.py's roofline prices k_kkt_fwd_seq over the 20 TIMED iterations,
# rocprofv3 --stats averages over all 25 launches including the 5 warm-up iterations -- this file shows both.
# Run on the GPU box from the repo root: bash tools/prof_fwd_per_launch.sh [tag] -> gpurun_out/prof_<tag>/per_launch_ms.json
TAG=${1:-r04}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
timeout -k 5 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --loop-only --steps 20 --warmup 5 > $OUT/per_launch_loop.json 2> $OUT/per_launch.err
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    for k in ("k_kkt_fwd_seq", "k_kkt_bwd_early", "k_update_eval", "k_linesearch", "k_kkt_bwd_gate"):
        if k in name:
            d[k].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
res = {}
for k, v in d.items():
    v.sort()
    ms = [round(x[1], 3) for x in v]
    res[k] = dict(launches=len(ms), ms=ms, mean_all=round(sum(ms) / len(ms), 3),
                  mean_timed_region=round(sum(ms[-20:]) / len(ms[-20:]), 3), note="last 20 launches = the timed iterations (5 warm-up iterations before them)")
json.dump(dict(command="bench.py --loop-only --steps 20 --warmup 5", kernels=res), open(out + "/per_launch_ms.json", "w"), indent=1)
print(json.dumps({k: (v["mean_all"], v["mean_timed_region"]) for k, v in res.items()}))
PY
rm -rf $OUT/trace
cat $OUT/per_launch_loop.json | tail -1 | cut -c1-200
