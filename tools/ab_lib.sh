#!/bin/bash
# A/B of two builds of the library on one box: tools/ab_lib.sh WORKLOAD libA.so libB.so [bench args...]
# (paths relative to yagre-mcmc_amd/; boxes differ by a few per cent, so both run in the same call, twice)
cd "$(dirname "$0")/.." || exit 1
WL=$1; A=$2; B=$3; shift 3
for rep in 1 2; do for lib in $A $B; do
    YMC_LIBRARY=$PWD/yagre-mcmc_amd/$lib python bench.py --workload $WL --no-cpu-baseline --steps 10 --warmup 3 "$@" > gpurun_out/ab_${WL}_$lib.json && python -c "
import json,sys;d=json.load(open(sys.argv[1]));r=d['roofline'];print('%-40s'%sys.argv[1],'%.4g'%d['value'],'frac %.3f'%r['frac'],'%.3f ms'%r['kernel_ms_avg'],'%.0f MHz'%d['sustained_clock_mhz'])" gpurun_out/ab_${WL}_$lib.json
done; done
