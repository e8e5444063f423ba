#!/bin/bash
# Iteration throughput of the headline loop against the batch size (loop only, 20 timed iterations after 5):
#   bash tools/bench_batch_sweep.sh [tag]   -> gpurun_out/sweep_<tag>/batch_sweep.txt
TAG=${1:-r03}
OUT=gpurun_out/sweep_$TAG
mkdir -p $OUT
: > $OUT/batch_sweep.txt
for B in 64 1024 8192 12288 18432 24576 32768 49152 65536 66560 98304 131072 196608 262144 393216 524288; do
  timeout -k 5 400 python bench.py --loop-only --steps 20 --warmup 5 --batch $B 2> $OUT/err_$B.log | tail -1 >> $OUT/batch_sweep.txt
done
python - "$OUT/batch_sweep.txt" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    d = json.loads(ln)
    print(f"{d['instances_per_gpu']:8d} instances  P={d['time_partitions']:2d}  {d['ms_per_step']:9.3f} ms/iteration  {d['value'] / 1e6:7.3f} M it/s  {d['factorizations_per_iteration']} factorisations/iteration")
PY
