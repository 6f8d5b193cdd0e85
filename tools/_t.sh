cd /root/repo
for e in "YMC_WIDE_SPLIT=0" "YMC_WIDE_SPLIT_KERNEL=stream" "YMC_WIDE_SPLIT_KERNEL=stream YMC_WIDE_SPLIT_W=8" "YMC_WIDE_W=88" "YMC_WIDE2_W=16" "YMC_WIDE_SPLIT_G=3" "YMC_WIDE_FUSED=0" "YMC_WIDE_TC=1"; do
  echo "== $e"; env $e timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
done
