#!/bin/bash
# occupancy 1 + next-stage prefetch (the product) against occupancy 2 without it, by batch size (GPU box, repo root)
OUT=gpurun_out/variant_ab_sizes_r06.txt
: > $OUT
for B in 65536 98304 131072 196608 262144 393216; do
  for FL in "" "-DDTO_SEQ_FWD_OCC=2 -DDTO_SEQ_PREFETCH_FWD=0"; do
    echo "== B=$B flags: '$FL'" >> $OUT
    DTO_PLUGIN_CXXFLAGS="$FL" timeout 600 python bench.py --loop-only --steps 20 --warmup 5 --batch $B >> $OUT 2>> gpurun_out/variant_ab_sizes_r06.err
  done
done
python3 - <<'PY'
import json,re
L=open('gpurun_out/variant_ab_sizes_r06.txt').read().split("\n")
cur=None
for l in L:
    if l.startswith("=="): cur=l
    elif l.startswith("{"):
        d=json.loads(l); print(cur, "->", round(d["ms_per_step"],2), "ms", round(d["value"]/1e6,3), "M it/s")
PY
