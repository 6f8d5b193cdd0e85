#!/bin/bash
# A/B of the builds of the two-level matrix-core kernel on C3's two-level workload, in ONE process lifetime of the
# box (boxes differ by a few per cent): tools/ab_c3m.sh [modes...]
cd "$(dirname "$0")/.." || exit 1
MODES=${@:-mem regs lds}
for rep in 1 2; do for m in $MODES; do YMC_TILE_MLDA2=$m python bench.py --workload c3m --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/ab_c3m_$m.json && python -c "
import json,sys;d=json.load(open(sys.argv[1]));r=d['roofline'];print(sys.argv[1],'%.4g'%d['value'],'frac %.3f'%r['frac'],'%.3f ms'%r['kernel_ms_avg'],'%.0f MHz'%d['sustained_clock_mhz'])" gpurun_out/ab_c3m_$m.json; done; done
if [ -f yagre-mcmc_amd/libymc_prof.so ]; then for m in $MODES; do YMC_TILE_MLDA2=$m python tools/tile_sections.py 5; done; fi
