cd /root/repo
for e in "YMC_TILE_MLDA2=mem" "YMC_TILE_MLDA2=lds" "YMC_TILE_MLDA2=plain" "YMC_TILE_MLDA2=stream" "YMC_TILE_NESTED=stream" "YMC_TILE_ASTREAM=1" "YMC_TILE_STAGE=0" "YMC_LV_ROWS=1 YMC_LV_PIECES=7"; do
  echo "== $e"; env $e timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
done
