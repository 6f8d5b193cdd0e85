#!/bin/bash
# rocprofv3 evidence for the group kernels (round 4): kernel trace + stats and the WRITE_SIZE / FETCH_SIZE passes (each on its own)
# of 256 chains at d = m = 512, 200 transitions per launch.  Output: gpurun_out/r04_wsmall_*; summaries are copied to profiles/ by hand.
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out; export TMPDIR=/tmp
cat > /tmp/wide_small_one.py <<PY
import os, sys
sys.path.insert(0, "$ROOT")
os.chdir("$ROOT")
import numpy as np
from yagremcmc_amd import _ffi as ffi, lowering
ctx = ffi.default_context(0)
rng = np.random.default_rng(1)
d = m = 512
K, n = 256, 200
l = lowering.Lowered(ffi.POSTERIOR_LINEAR, d)
l.desc.n_data, l.desc.data_dim = 3, m
for k, v in (("data", rng.normal(size=(3, m))), ("noise_precision", np.full(m, 3.0)), ("prior_mean", np.zeros(d)),
             ("prior_precision", np.full(d, 0.2)), ("fwd_matrix", rng.normal(size=(m, d)) / np.sqrt(d)), ("fwd_shift", rng.normal(size=m))):
    l.set_array(k, v)
prob = lowering.make_problem(ctx, ffi.METHOD_MRW, l, np.full(d, 0.02))
run = ffi.Run(prob, 1, K, rng.normal(size=(K, d)) * 0.1, n, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT, 0)
for rep in range(6):
    run.rewind_record()
    run.advance(n); run.sync()
ms, launches = run.kernel_time_ms()
print("256 chains d = m = 512: %.3f ms per launch of %d transitions (HIP events, %d launches)" % (ms, n, launches))
PY
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_wsmall_stats -- python3 /tmp/wide_small_one.py > $OUT/r04_wsmall_stats.txt 2>&1 || exit 1
for pass in "WRITE_SIZE" "FETCH_SIZE"; do
  rocprofv3 --pmc $pass --output-format csv -d $OUT/r04_wsmall_pmc_$pass -- python3 /tmp/wide_small_one.py > /dev/null 2>&1 || echo "pass $pass failed"
done
python3 - <<PY
import csv, glob, json
src = sorted(glob.glob("$OUT/r04_wsmall_stats/**/*kernel_stats.csv", recursive=True))[-1]
open("$OUT/r04_wide_d512_m512_K256_kernel_stats.csv", "w").write(open(src).read())
out = {}
for f in glob.glob("$OUT/r04_wsmall_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "systolic" in r["Kernel_Name"] or "wide_split" in r["Kernel_Name"]:
            out.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
med = {k: sorted(v)[len(v) // 2] for k, v in out.items()}
K, T, d = 256, 200, 512
med["hbm_bytes_per_launch"] = med.get("WRITE_SIZE", 0) * 1024 + 2 * med.get("FETCH_SIZE", 0) * 1024
med["algorithmic_bytes_per_launch"] = K * T * (8 * d + 1)
med["exchange_bytes_per_launch"] = T * (K // 16) * 512 * 16 * 8 * 2   # the terms written once and read once by the leader
med["hip_events"] = [l.strip() for l in open("$OUT/r04_wsmall_stats.txt") if "256 chains" in l][-1]
json.dump(med, open("$OUT/r04_wide_d512_m512_K256_pmc.json", "w"), indent=1)
print(json.dumps(med, indent=1))
PY
