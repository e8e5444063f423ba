#!/bin/bash
# A/B of plugin build variants of the sequential forward sweep on the headline loop (run on the GPU box from the repo root; the
# variants are prebuilt with DTO_PLUGIN_CXXFLAGS=<flags> on the build host: the flags are part of the plugin cache key)
OUT=gpurun_out/variant_ab_r06.txt
: > $OUT
run() {
  echo "== flags: '$1'" >> $OUT
  DTO_PLUGIN_CXXFLAGS="$1" timeout 600 python bench.py --loop-only --steps 20 --warmup 5 --batch ${B:-524288} >> $OUT 2>> gpurun_out/variant_ab_r06.err
}
run ""
run "-DDTO_SEQ_FWD_OCC=2 -DDTO_SEQ_PREFETCH_FWD=0"
run "-DDTO_SEQ_FWD_OCC=2"
run "-DDTO_SEQ_FWD_RL=1 -DDTO_SEQ_FWD_OCC=2 -DDTO_SEQ_PREFETCH_FWD=0"
run "-DDTO_SEQ_FWD_RL=1"
run ""
cat $OUT
