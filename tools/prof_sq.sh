#!/bin/bash
# SQ counter passes over the headline loop (SoA engine), run on the GPU box from the repo root:
#   bash tools/prof_sq.sh <batch> <tag>    -> gpurun_out/sq_<tag>/sq_counters.{json,txt}
B=${1:-131072}; TAG=${2:-r03}
OUT=$PWD/gpurun_out/sq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
[ -f "$ROOT/bench.py" ] || { echo "run from the repo root (bench.py not found under $ROOT)" >&2; exit 1; }
CMD="$ROOT/bench.py --loop-only --steps ${STEPS:-8} --warmup ${WARMUP:-2} --batch $B"
cd /tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INST_LEVEL_VMEM SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout -k 5 300 rocprofv3 --pmc $SET --output-format csv -d $OUT/pass$i -- python3 $CMD > $OUT/pass$i.log 2>&1
  echo "pass $i exit $?" >> $OUT/passes.log
done
cd $ROOT
python3 tools/pmc_sq_summary.py $OUT/sq_counters.json $(find $OUT -name "*counter_collection.csv") > $OUT/sq_counters.txt 2>&1
grep -A26 "k_kkt_fwd_seq\|k_kkt_bwd_seq" $OUT/sq_counters.txt | head -80
rm -rf $OUT/pass1 $OUT/pass2 $OUT/pass3
