#!/bin/bash
# Kernel statistics of a short instance-major run under rocprofv3 (run on the GPU box from the repo root):
#   bash tools/prof_im.sh <tag> <args of tools/im_vs_soa.py ...>
# writes gpurun_out/prof_<tag>/kernel_stats.csv.  Every step is under its own timeout: a profiler that does not exit must
# not eat the box's time limit.
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
[ -f "$ROOT/bench.py" ] || { echo "run from the repo root (bench.py not found under $ROOT)" >&2; exit 1; }
cd /tmp
timeout -k 5 420 rocprofv3 --kernel-trace --output-format rocpd -d $OUT -o run -- python3 $ROOT/tools/im_vs_soa.py "$@" > $OUT/run.log 2>&1
echo "rocprofv3 exit: $?" >> $OUT/run.log
cd $ROOT
DB=$(find $OUT -name "*.db" | head -1)
grep '^{' $OUT/run.log
[ -n "$DB" ] && python3 tools/rocpd_stats.py $DB $OUT/kernel_stats.csv | head -14
rm -f $OUT/*.db
