#!/bin/bash
# The whole GPU suite under every build-selection override (the kernels must agree with the oracle whichever build runs):
#   tools/env_sweep.sh            (on the GPU box; about half a minute per setting)
cd "$(dirname "$0")/.." || exit 1
for e in "YMC_TILE_MLDA2=mem" "YMC_TILE_MLDA2=lds" "YMC_TILE_MLDA2=plain" "YMC_TILE_MLDA2=stream" "YMC_TILE_NESTED=stream" \
         "YMC_TILE_ASTREAM=1" "YMC_TILE_STAGE=0" "YMC_LV_ROWS=1 YMC_LV_PIECES=7" "YMC_LV_ROWS=0" "YMC_SPLIT=0" "YMC_SPLIT=1 YMC_SPLIT_NP=2" \
         "YMC_TILE_AEM=lds" "YMC_WIDE_FUSED=0" "YMC_WIDE_TC=1" "YMC_IAT_KERNEL=wave" \
         "YMC_WIDE_SPLIT=0" "YMC_WIDE_SPLIT_KERNEL=stream" "YMC_WIDE_SPLIT_KERNEL=stream YMC_WIDE_SPLIT_W=8" "YMC_WIDE_W=16" "YMC_WIDE2_W=8" \
         "YMC_WIDE_SPLIT_G=3"; do
  echo "== $e"; env $e timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
done
