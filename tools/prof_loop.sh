#!/bin/bash
# Kernel-trace statistics of the headline loop, run on the GPU box from the repo root:
#   bash tools/prof_loop.sh <batch> <tag>   -> gpurun_out/loop_<tag>/kernel_stats.csv
B=${1:-131072}; TAG=${2:-r03}
OUT=$PWD/gpurun_out/loop_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
[ -f "$ROOT/bench.py" ] || { echo "run from the repo root (bench.py not found under $ROOT)" >&2; exit 1; }
cd /tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 $ROOT/bench.py --loop-only --steps ${STEPS:-8} --warmup ${WARMUP:-2} --batch $B > $OUT/run.log 2>&1
echo "exit $?" >> $OUT/run.log
cd $ROOT
F=$(find $OUT/raw -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp $F $OUT/kernel_stats.csv && head -12 $OUT/kernel_stats.csv | cut -c1-200
grep '"value"' $OUT/run.log | cut -c1-400
rm -rf $OUT/raw
