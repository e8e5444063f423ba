#!/bin/bash
# Kernel durations along a solve: rocprofv3 kernel trace of the headline loop, summarised per 25-iteration slice.
# bash tools/trace_phases.sh [steps] [batch]  (GPU box, repo root) -> gpurun_out/phases/
STEPS=${1:-300}; B=${2:-524288}
OUT=gpurun_out/phases; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --loop-only --steps $STEPS --warmup 0 --batch $B > $OUT/loop.json 2> $OUT/trace.err
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 tools/trace_phases.py "$F" > $OUT/phases_B${B}.txt; cat $OUT/phases_B${B}.txt; cat $OUT/loop.json
rm -rf $OUT/trace
