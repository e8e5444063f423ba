#!/bin/bash
# Kernel times of a few solver iterations on the 64-state configuration (T = 2000, 256 instances), run on the GPU box from the repo
# root:  bash tools/prof_wide_solve.sh [tag]  -> gpurun_out/wide_solve_<tag>/kernel_stats.csv
TAG=${1:-r04}
OUT=$PWD/gpurun_out/wide_solve_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 $ROOT/tools/wide_solve_demo.py 2000 256 physical 3 > $OUT/run.log 2>&1
echo "exit $?" >> $OUT/run.log
cd $ROOT
f=$(find $OUT/raw -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv
rm -rf $OUT/raw
head -8 $OUT/kernel_stats.csv | cut -c1-160
tail -3 $OUT/run.log
