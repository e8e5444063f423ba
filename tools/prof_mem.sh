#!/bin/bash
# Memory-path counter passes (vector L1 / address translation / L2 latency) over the headline loop, run on the GPU box from the
# repo root:   bash tools/prof_mem.sh <batch> <tag>   -> gpurun_out/mem_<tag>/mem_counters.{json,txt}
B=${1:-131072}; TAG=${2:-r03}
OUT=$PWD/gpurun_out/mem_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
[ -f "$ROOT/bench.py" ] || { echo "run from the repo root (bench.py not found under $ROOT)" >&2; exit 1; }
CMD="$ROOT/bench.py --loop-only --steps 8 --warmup 2 --batch $B"
cd /tmp
timeout -k 5 120 rocprofv3 --list-avail > $OUT/avail.txt 2>&1
grep -o "TCP_[A-Z0-9_]*\|TCC_[A-Z0-9_]*\|TA_[A-Z0-9_]*\|TD_[A-Z0-9_]*" $OUT/avail.txt | sort -u > $OUT/avail_mem_names.txt
i=0
for SET in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_PERMISSION_MISS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum"; do
  i=$((i+1))
  timeout -k 5 300 rocprofv3 --pmc $SET --output-format csv -d $OUT/pass$i -- python3 $CMD > $OUT/pass$i.log 2>&1
  echo "pass $i ($SET) exit $?" >> $OUT/passes.log
done
cd $ROOT
python3 tools/pmc_sq_summary.py $OUT/mem_counters.json $(find $OUT -name "*counter_collection.csv") > $OUT/mem_counters.txt 2>&1
grep -A30 "k_kkt_fwd_seq\|k_kkt_bwd_seq" $OUT/mem_counters.txt | head -90
tail -3 $OUT/pass*.log | head -60
rm -rf $OUT/pass1 $OUT/pass2 $OUT/pass3 $OUT/pass4 $OUT/pass5 $OUT/pass6 $OUT/pass7
