#!/bin/bash
# SQ counters of k_kkt_fwd_seq for two plugin build variants (GPU box, repo root): is a second wavefront per SIMD resident, what does it execute?
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/variant_sq_r06
mkdir -p $OUT
cd /tmp
i=0
for FL in "" "-DDTO_SEQ_FWD_OCC=2 -DDTO_SEQ_PREFETCH_FWD=0"; do
  i=$((i+1))
  export DTO_PLUGIN_CXXFLAGS="$FL"
  timeout -k 5 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAVES --output-format csv -d $OUT/a$i -- python3 $ROOT/bench.py --loop-only --steps 8 --warmup 2 --batch 131072 > $OUT/a$i.log 2>&1
  timeout -k 5 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES --output-format csv -d $OUT/b$i -- python3 $ROOT/bench.py --loop-only --steps 8 --warmup 2 --batch 131072 > $OUT/b$i.log 2>&1
  cd $ROOT
  python3 tools/pmc_sq_summary.py $OUT/v$i.json $(find $OUT/a$i $OUT/b$i -name "*counter_collection.csv") > $OUT/v$i.txt 2>&1
  echo "=== variant '$FL'"; grep -A18 "^k_kkt_fwd_seq" $OUT/v$i.txt | head -20; tail -1 $OUT/a$i.log
  rm -rf $OUT/a$i $OUT/b$i
  cd /tmp
done
