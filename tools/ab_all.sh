#!/bin/bash
# A/B of two builds of the library over the bench workloads on one box: tools/ab_all.sh libA.so libB.so [workloads...]
cd "$(dirname "$0")/.." || exit 1
A=$1; B=$2; shift 2
WLS=${@:-c2 c5 c4 c3}
for wl in $WLS; do for rep in 1 2; do for lib in $A $B; do
    YMC_LIBRARY=$PWD/yagre-mcmc_amd/$lib python bench.py --workload $wl --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/ab_all.json && python -c "
import json,sys;d=json.load(open(sys.argv[1]));r=d['roofline'];print('%-4s %-18s'%(sys.argv[2],sys.argv[3]),'%.4g'%d['value'],'frac %.3f'%r['frac'],'%.3f ms'%r['kernel_ms_avg'],'%.0f MHz'%d['sustained_clock_mhz'])" gpurun_out/ab_all.json $wl $lib
done; done; done
