#!/bin/bash
# rocprofv3 evidence for the fused wide kernel (round 4): kernel trace + stats and the PMC passes (each on its own) of
# tools/exp_wide_fused.py's d = m = 512 case.  Output: gpurun_out/r04_wide_*; summaries are copied to profiles/ by hand.
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out; export TMPDIR=/tmp
cat > /tmp/wide_one.py <<PY
import os, sys
sys.path.insert(0, "$ROOT")
os.chdir("$ROOT")
sys.argv = ["exp", "65536"]
src = open("$ROOT/tools/exp_wide_fused.py").read().split("TA = ffi.RECORD_TRAJECTORY")[0]
exec(src)
timed("d=512 m=512 record, 30 steps", 512, 512, 30, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT)
PY
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_wide_stats -- python3 /tmp/wide_one.py > $OUT/r04_wide_stats.txt 2>&1 || exit 1
for pass in "WRITE_SIZE" "FETCH_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d $OUT/r04_wide_pmc_$tag -- python3 /tmp/wide_one.py > /dev/null 2>&1 || echo "pass $tag failed"
done
python3 - <<PY
import csv, glob, json
out = {}
rows = list(csv.DictReader(open(sorted(glob.glob("$OUT/r04_wide_stats/**/*kernel_stats.csv", recursive=True))[-1])))
open("$OUT/r04_wide_d512_m512_K65536_kernel_stats.csv", "w").write(open(sorted(glob.glob("$OUT/r04_wide_stats/**/*kernel_stats.csv", recursive=True))[-1]).read())
for f in glob.glob("$OUT/r04_wide_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wide_fused" in r["Kernel_Name"]:
            out.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
med = {k: sorted(v)[len(v) // 2] for k, v in out.items()}
K, T, d = 65536, 30, 512
med["hbm_bytes_per_launch"] = med.get("WRITE_SIZE", 0) * 1024 + 2 * med.get("FETCH_SIZE", 0) * 1024
med["algorithmic_bytes_per_launch"] = K * T * (8 * d + 1)
med["algorithmic_flops_per_launch"] = 2.0 * 512 * 512 * K * T
json.dump(med, open("$OUT/r04_wide_d512_m512_K65536_pmc.json", "w"), indent=1)
print(json.dumps(med, indent=1))
PY
