#!/bin/bash
# c3a (adaptive error model, d = m = 128, 65 536 chains) with the error model in LDS against the coarse matrix in LDS
for v in lds matrix; do
  YMC_TILE_AEM=$v python bench.py --workload c3a --steps 6 --warmup 2 --legs none --no-cpu-baseline > gpurun_out/r4_c3a_$v.json
  python -c "
import json; d=json.load(open('gpurun_out/r4_c3a_$v.json')); print('$v', '%.4g'%d['value'], '%.3f ms'%d['roofline']['kernel_ms_avg'], '%.3f'%d['roofline']['frac'], d['acceptance_rate'])"
done
