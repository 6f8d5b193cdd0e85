#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's roofline block, on the GPU box:
#   tools/collect_profiles.sh r01
# One pass per kind (kernel trace + stats; PMC passes each on their own -- gpurun refuses PMC together
# with API tracing), all of the SAME bench command.  Raw output goes to gpurun_out/prof_*; the small
# summaries that are judged are written to profiles/ by tools/summarise_profiles.py.
set -o pipefail
TAG=${1:-r01}
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
run_set() { # name, bench args...
    local name=$1; shift
    rm -rf "$OUT"/prof_${name}_*
    ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${name}_stats" -- python3 "$ROOT/bench.py" "$@" --no-cpu-baseline \
        > "$OUT/prof_${name}_stats.json" 2> "$OUT/prof_${name}_stats.err" ) || return 1
    for ctr in WRITE_SIZE FETCH_SIZE; do
        ( cd /tmp && rocprofv3 --pmc $ctr --output-format csv -d "$OUT/prof_${name}_$ctr" -- python3 "$ROOT/bench.py" "$@" --steps 5 --warmup 1 --no-cpu-baseline \
            > "$OUT/prof_${name}_$ctr.json" 2> "$OUT/prof_${name}_$ctr.err" ) || return 1
    done
    ( cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d "$OUT/prof_${name}_SQ" \
        -- python3 "$ROOT/bench.py" "$@" --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/prof_${name}_SQ.json" 2> "$OUT/prof_${name}_SQ.err" ) \
        || echo "SQ counter pass failed for $name (not fatal)"
    echo "collected $name"
}
run_set c2 --workload c2 && run_set c3 --workload c3 && run_set c5 --workload c5 && run_set c4 --workload c4 \
    && python3 tools/summarise_profiles.py "$TAG"
