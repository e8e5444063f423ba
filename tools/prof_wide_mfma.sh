#!/bin/bash
# Matrix-pipe counters of k_wide_step (BASELINE configs[4]: 64 states, 129 x 129 blocks, T = 2000, 256 instances), run on the GPU
# box from the repo root:  bash tools/prof_wide_mfma.sh [tag]  -> gpurun_out/wide_mfma_<tag>/{counters.csv,summary.json}
TAG=${1:-r04}
OUT=$PWD/gpurun_out/wide_mfma_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
[ -f "$ROOT/tools/wide_bench.py" ] || { echo "run from the repo root" >&2; exit 1; }
cd /tmp
timeout -k 5 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/raw -- python3 $ROOT/tools/wide_bench.py 2000 256 1 > $OUT/run.log 2>&1
echo "exit $?" >> $OUT/run.log
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
d = collections.defaultdict(list)
per = collections.defaultdict(list)
for f in glob.glob(out + "/raw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        # the KKT step is two launches since round 4: forward sweep (k_wide_step) and backward sweep (k_wide_bwd) -- summed here
        if "k_wide_step" in r["Kernel_Name"] or "k_wide_bwd" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
            per[("fwd" if "k_wide_step" in r["Kernel_Name"] else "bwd", r["Counter_Name"])].append(float(r["Counter_Value"]))
nl = max(1, len(per.get(("fwd", "SQ_WAVE_CYCLES"), [])))
m = {k: sum(v) / nl for k, v in d.items()}
res = dict(kernel="dto::wide::k_wide_step + k_wide_bwd = one KKT step (acrobot embedded in 64 states, T=2000, 256 instances = one workgroup per CU)",
           steps=nl, counters_mean_per_step=m,
           per_kernel_mean={f"{a}:{c}": sum(v) / len(v) for (a, c), v in per.items()})
if m:
    mf = m.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) / 4.0        # MOPS counts 512-flop units, 4 per v_mfma_f64_16x16x4_f64
    cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0                   # summed over the 8 XCDs
    res["derived"] = dict(mfma_f64_instructions=mf, gpu_cycles_per_launch=cyc,
                          mfma_pipe_utilisation=m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024.0) if cyc else None,
                          wave_waiting_fraction=m.get("SQ_WAIT_ANY", 0.0) / m.get("SQ_WAVE_CYCLES", 1.0),
                          valu_active_fraction=m.get("SQ_ACTIVE_INST_VALU", 0.0) / m.get("SQ_WAVE_CYCLES", 1.0),
                          note="one f64 MFMA occupies the matrix pipe of its SIMD for 64 cycles (16 passes): pipe peak 32 flop/cycle/SIMD = 78.6 TFLOP/s at 2.4 GHz; "
                               "SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1 024 SIMDs (64 busy cycles per instruction, as in profiles/r01/pmc_wide_step_summary.json)")
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $OUT/raw
