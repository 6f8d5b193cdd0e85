import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import bench
from yagremcmc_amd import _ffi as ffi, lowering
ctx = ffi.Context(0)
prob, x0, K, T, desc, d, wl = bench.make_workload("c2", ffi, ctx, lowering)
run = ffi.Run(prob, 1, K, x0, T, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT | ffi.RECORD_MOMENTS)
run.advance(T); run.sync()
for rep in range(3):
    run.rewind_record()
    t0 = time.perf_counter(); run.advance(T); run.sync(); t1 = time.perf_counter()
    traj = run.trajectory(); acc = run.accept(); t2 = time.perf_counter()
    print(f"advance {1e3*(t1-t0):.2f} ms  fetch {1e3*(t2-t1):.1f} ms  ({(traj.nbytes+acc.nbytes)/(t2-t1)/1e9:.1f} GB/s)  "
          f"PCIe-inclusive {K*T/(t2-t0):.3e} steps/s  summary-only {K*T/(t1-t0):.3e}")
