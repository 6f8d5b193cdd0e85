#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's roofline block, on the GPU box:
#   tools/collect_profiles.sh r03 [workloads...]
# One pass per kind of the SAME bench command: kernel trace + stats; then PMC passes each on their own
# (gpurun refuses PMC together with API tracing; FETCH_SIZE and WRITE_SIZE do not fit one pass):
#   WRITE_SIZE | FETCH_SIZE | SQ issue counters | SQ wait counters + the MFMA counters of the guide
# Raw output goes to gpurun_out/prof_*; the small summaries that are judged are written to profiles/ by
# tools/summarise_profiles.py.
set -o pipefail
TAG=${1:-r02}; shift
WLS=${@:-c2 c3 c3m c3a c5 c4 iat}
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
pmc_pass() { # name, tag, counters...
    local name=$1 tag=$2; shift 2
    ( cd /tmp && rocprofv3 --pmc "$@" --output-format csv -d "$OUT/prof_${name}_$tag" -- python3 "$ROOT/bench.py" --workload $name \
        --steps 1 --warmup 1 --launches-per-step 10 --no-cpu-baseline > "$OUT/prof_${name}_$tag.json" 2> "$OUT/prof_${name}_$tag.err" )
}
run_set() {
    local name=$1
    rm -rf "$OUT"/prof_${name}_*
    ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${name}_stats" -- python3 "$ROOT/bench.py" \
        --workload $name --steps 5 --warmup 2 --launches-per-step 20 --no-cpu-baseline > "$OUT/prof_${name}_stats.json" 2> "$OUT/prof_${name}_stats.err" ) || return 1
    pmc_pass $name WRITE_SIZE WRITE_SIZE || return 1
    pmc_pass $name FETCH_SIZE FETCH_SIZE || return 1
    pmc_pass $name SQ SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_MFMA \
        || echo "SQ counter pass failed for $name (not fatal)"
    pmc_pass $name SQW SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES \
        SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_LDS || echo "SQ wait/MFMA counter pass failed for $name (not fatal)"
    pmc_pass $name GRBM GRBM_GUI_ACTIVE || echo "GRBM pass failed for $name (not fatal)"
    echo "collected $name"
}
for wl in $WLS; do run_set $wl || exit 1; done
python3 tools/summarise_profiles.py "$TAG" $WLS
