#!/usr/bin/env python3
"""The PCIe-inclusive rate quoted in NOTEBOOK.md (rounds 1-3, section 5): C2 (65 536 chains, 2-D Gaussian target) when every recorded
row is brought to the host, (a) the round-1 way -- synchronous fetch into pageable NumPy arrays after every
launch -- and (b) through the egress path (ymc_run_egress_*: page-locked double buffer on a copy stream,
overlapping the next launch), at record stride 1 and at stride = the median integrated autocorrelation time
(what the reference's examples keep after thinning, example_mcmc_2d_singleLevel.py:71-79).
    python tools/measure_pcie_inclusive.py [launches]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from yagremcmc_amd import _ffi as ffi, lowering  # noqa: E402

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ctx = ffi.Context(0)
wl = bench.make_workload("c2", ffi, ctx, lowering)
prob, x0, K, T = wl["problem"], wl["x0"], wl["K"], 1000
TA = ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT

# (a) synchronous fetch, pageable destination
run = ffi.Run(prob, 1, K, x0, T, TA | ffi.RECORD_MOMENTS)
run.advance(T)
run.sync()
t0 = time.perf_counter()
for _ in range(3):
    run.rewind_record()
    run.advance(T)
    traj, acc = run.trajectory(), run.accept()
dt = time.perf_counter() - t0
print(f"(a) fetch into pageable arrays : {3 * K * T / dt:.3e} steps/s  ({3 * (traj.nbytes + acc.nbytes) / dt / 1e9:.1f} GB/s)")
iat = int(np.median(run.iat(200, None, 1)))
del run


def egress_rate(stride, what, flags):
    rows = T // stride
    run = ffi.Run(prob, 1, K, x0, 2 * rows, flags, record_stride=stride)
    sink = 0.0
    for b in range(2):                       # first touch of the page-locked buffers is not timed
        run.advance(rows * stride)
        run.egress_begin(b, b * rows, rows, what)
    run.egress_wait(0), run.egress_wait(1)
    t0 = time.perf_counter()
    nbytes = 0
    for b in range(launches):
        half = b & 1
        if half == 0:
            run.rewind_record()
        run.advance(rows * stride)
        run.egress_begin(half, half * rows, rows, what)
        if b >= 1:
            v = run.egress_wait(1 - half)
            sink += float(v["trajectory"][-1, -1, 0])     # the host touches what arrived
            nbytes += sum(a.nbytes for k, a in v.items() if k != "first")
    v = run.egress_wait((launches - 1) & 1)
    nbytes += sum(a.nbytes for k, a in v.items() if k != "first")
    dt = time.perf_counter() - t0
    return launches * K * rows * stride / dt, nbytes / dt / 1e9


r1, bw1 = egress_rate(1, TA, TA | ffi.RECORD_MOMENTS)
print(f"(b) egress, stride 1           : {r1:.3e} steps/s  ({bw1:.1f} GB/s over PCIe, overlapped with the launches)")
rs, bws = egress_rate(iat, ffi.RECORD_TRAJECTORY, ffi.RECORD_TRAJECTORY | ffi.RECORD_MOMENTS)
print(f"(c) egress, stride {iat:3d} (median IAT, 'max'): {rs:.3e} steps/s  ({bws:.1f} GB/s)")
