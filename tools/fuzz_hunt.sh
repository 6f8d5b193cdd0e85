#!/bin/bash
# The fuzz suites at other seeds than the fixed ones:   tools/fuzz_hunt.sh FIRST LAST   (on the GPU box; ~6 s per shift)
cd "$(dirname "$0")/.." || exit 1
for s in $(seq ${1:-1} ${2:-10}); do
    echo "== shift $s"; YMC_FUZZ_SHIFT=$s timeout -k 10 300 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -3
done
