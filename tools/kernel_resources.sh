#!/bin/bash
# Registers, spills, LDS and scratch of the gfx950 kernels in one object file:
#   tools/kernel_resources.sh yagre-mcmc_amd/csrc/ymc_kernels_p1.o [name-regex]
set -e
OBJ=$(readlink -f "$1"); PAT=${2:-.}
TMP=$(mktemp -d); trap 'rm -rf $TMP' EXIT
cp "$OBJ" $TMP/o.o
( cd $TMP && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading o.o > /dev/null )
CO=$(ls $TMP/o.o.*gfx950* | head -1)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$CO" > $TMP/notes.txt
if [ -n "$3" ]; then /opt/rocm/lib/llvm/bin/llvm-objdump -d "$CO" > "$3"; fi
python3 - $TMP/notes.txt "$PAT" <<'PY'
import sys, re
txt = open(sys.argv[1]).read()
for blk in txt.split("  - .agpr_count:")[1:]:
    m = re.search(r"\.name:\s+(\S+)", blk)
    if m and re.search(sys.argv[2], m.group(1)):
        g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)
        print(f"{m.group(1)[:100]:100s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} vspill {g('vgpr_spill_count'):>3s} "
              f"sspill {g('sgpr_spill_count'):>3s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size')}")
PY
