#!/bin/bash
# A/B of compile-time variants of the acrobot plugin (variants/<tag>.so built by hand with -D flags), on the GPU box:
#   bash tools/variant_ab.sh <cache-name>.so tag1 tag2 ...
CACHE=directtrajectoryoptimization.jl_amd/_plugins/$1; shift
cp $CACHE /tmp/orig_plugin.so
for tag in base "$@"; do
  if [ $tag = base ]; then cp /tmp/orig_plugin.so $CACHE; else cp variants/$tag.so $CACHE; fi
  echo "== $tag"
  timeout -k 5 200 python tools/im_vs_soa.py --batch ${BATCH:-131072} --iters 25 --engines soa 2>&1 | tail -1 | cut -c1-260
done
cp /tmp/orig_plugin.so $CACHE
