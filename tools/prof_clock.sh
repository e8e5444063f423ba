#!/bin/bash
# Average engine clock per kernel of the headline loop: GRBM_GUI_ACTIVE (cycles the GPU was busy during the dispatch) over
# the dispatch duration.  bash tools/prof_clock.sh <batch> <tag>  -> gpurun_out/clock_<tag>/clock.txt
B=${1:-131072}; TAG=${2:-r03}
OUT=$PWD/gpurun_out/clock_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
[ -f "$ROOT/bench.py" ] || { echo "run from the repo root (bench.py not found under $ROOT)" >&2; exit 1; }
cd /tmp
timeout -k 5 400 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/raw -- python3 $ROOT/bench.py --loop-only --steps 6 --warmup 2 --batch $B > $OUT/run.log 2>&1
cd $ROOT
python3 - "$OUT" <<'PY' | tee $OUT/clock.txt
import csv, glob, sys, collections
out = sys.argv[1]
cc = glob.glob(out + "/raw/**/*counter_collection.csv", recursive=True)
kt = glob.glob(out + "/raw/**/*kernel_trace.csv", recursive=True)
dur = {}
for f in kt:
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for f in cc:
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Dispatch_Id"] not in dur:
            continue
        name, ns = dur[r["Dispatch_Id"]]
        name = name.replace("void ", "").split("<")[0].split("(")[0].replace("dto::", "")
        if ns < 1e6:
            continue
        a = acc[name]; a[0] += float(r["Counter_Value"]); a[1] += ns; a[2] += 1
for k, (cyc, ns, n) in sorted(acc.items()):
    print(f"{k:20s} launches {n:3d}  busy cycles / duration = {cyc / ns:6.3f} GHz (x number of XCD/SE instances the counter sums over)")
PY
head -3 $(ls $OUT/raw/*/*counter_collection.csv | head -1) | cut -c1-400
rm -rf $OUT/raw
