for e in "YMC_TILE_MLDA2=mem" "YMC_TILE_MLDA2=lds" "YMC_TILE_MLDA2=plain" "YMC_TILE_ASTREAM=1" "YMC_TILE_STAGE=0" "YMC_LV_ROWS=1 YMC_LV_PIECES=7" "YMC_LV_ROWS=0" "YMC_SPLIT=0" "YMC_SPLIT=1 YMC_SPLIT_NP=2"; do
  echo "== $e"; env $e timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
done
