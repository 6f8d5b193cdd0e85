#!/usr/bin/env python3
"""Registers, spills and occupancy of every kernel of one translation unit (the compiler's own report):
    tools/kernel_resources.py yagre-mcmc_amd/csrc/ymc_wide_fused.hip [extra hipcc flags]
"""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-I/opt/rocm/include",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null", *sys.argv[2:]]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::", "", cur).split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z /\[\]]+): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'spill':>6s} {'sspill':>6s} {'occ':>4s} {'LDS':>7s}")
for k, r in rows.items():
    print(f"{k[:70]:70s} {r.get('VGPRs', 0):5d} {r.get('AGPRs', 0):5d} {r.get('VGPRs Spill', 0):6d} {r.get('SGPRs Spill', 0):6d} "
          f"{r.get('Occupancy [waves/SIMD]', 0):4d} {r.get('LDS Size [bytes/block]', 0):7d}")
