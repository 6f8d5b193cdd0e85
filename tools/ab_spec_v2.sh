#!/bin/bash
# the product's draw specification against the cheaper one of `make -C yagre-mcmc_amd/csrc v2x`
# (timing experiment: one Philox block per d = 2 transition, three normal pairs per two blocks beyond the first),
# C2 / C5 / C3 in ONE call on one box.  Output: gpurun_out/r4_spec_v2.txt
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/r4_spec_v2.txt
: > $OUT
for rep in 1 2; do
for lib in libymc.so libymc_v2x.so; do
  for wl in c2 c5 c3; do
    YMC_LIBRARY=$PWD/yagre-mcmc_amd/$lib python bench.py --workload $wl --steps 6 --warmup 2 --legs none --no-cpu-baseline > gpurun_out/_ab.json || exit 1
    python - "$lib" "$wl" >> $OUT <<'PY'
import json, sys
d = json.load(open("gpurun_out/_ab.json"))
r = d["roofline"]
print(f"{sys.argv[1]:16s} {sys.argv[2]:4s} {d['value']:.4e} steps/s  kernel {r['kernel_ms_avg']:.3f} ms  frac {r['frac']:.4f}  "
      f"clock {d['sustained_clock_mhz']:.0f} MHz  acceptance {d['acceptance_rate']:.5f}  mean {d['pooled_mean']}")
PY
  done
done
done
cat $OUT
