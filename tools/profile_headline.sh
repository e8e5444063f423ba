#!/bin/bash
# Profile (rounds 2 to 4) of the headline loop ALONE (acrobot T=1000, default batch): kernel statistics and the two HBM traffic
# passes.  Run on the GPU box from the repo root: bash tools/profile_headline.sh [batch] ; results under gpurun_out/prof_$TAG/ (TAG defaults to r04)
B=${1:-524288}
OUT=gpurun_out/prof_${TAG:-r04}
mkdir -p $OUT
export TMPDIR=/tmp
CMD="bench.py --loop-only --steps ${STEPS:-40} --warmup ${WARMUP:-5} --batch $B"
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_B$B -- python3 $CMD > $OUT/loop_B$B.json 2> $OUT/stats_B$B.err
timeout -k 5 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_B$B -- python3 $CMD > /dev/null 2> $OUT/fetch_B$B.err
timeout -k 5 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_B$B -- python3 $CMD > /dev/null 2> $OUT/write_B$B.err
find $OUT -name "*kernel_stats.csv" | head; find $OUT -name "*counter_collection.csv" | head
cat $OUT/loop_B$B.json
F=$(find $OUT/fetch_B$B -name "*counter_collection.csv" | head -1); W=$(find $OUT/write_B$B -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py "$F" "$W" $OUT/pmc_traffic_acrobot_T1000_B$B.json "headline loop only: $CMD"
S=$(find $OUT/stats_B$B -name "*kernel_stats.csv" | head -1); cp "$S" $OUT/headline_loop_B${B}_kernel_stats.csv; head -12 "$S"
# the raw per-dispatch counter files are large: keep the summaries only
rm -rf $OUT/fetch_B$B $OUT/write_B$B $OUT/stats_B$B
