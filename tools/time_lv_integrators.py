"""Wall time of the Lotka-Volterra random walk (BASELINE configs[3] shapes: nData 10, T 10, K chains x n steps) with the
fixed-step integrator (row-per-lane kernel, and lane-per-chain with YMC_LV_ROWS=0) and with the adaptive one.
    python tools/time_lv_integrators.py [K] [n]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd())
from yagremcmc_amd import _ffi as ffi, lowering

K = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
ctx = ffi.default_context(0)
rng = np.random.default_rng(3)
design = rng.uniform(0.5, 1.5, size=(10, 2))


def low(rtol):
    l = lowering.Lowered(ffi.POSTERIOR_LV, 2)
    l.desc.n_data, l.desc.data_dim = 10, 2
    for k, v in (("data", rng.uniform(0.3, 2.0, size=(10, 2))), ("noise_precision", np.full(2, 25.0)),
                 ("prior_mean", np.zeros(2)), ("prior_precision", np.full(2, 1 / 1.4)), ("design", design)):
        l.set_array(k, v)
    l.desc.t_final, l.desc.alpha, l.desc.gamma, l.desc.rk4_steps = 10.0, 0.8, 0.4, 50
    l.desc.lv_rtol, l.desc.lv_atol = rtol, 1e-6 if rtol > 0 else 0.0
    return l


x0 = np.log([0.4, 0.6]) + 0.1 * rng.normal(size=(K, 2))
for label, rtol, rows in (("RK4 x 50, row per lane", 0.0, "1"), ("RK4 x 50, lane per chain", 0.0, "0"),
                          ("adaptive rtol 1e-3", 1e-3, None), ("adaptive rtol 1e-4", 1e-4, None),
                          ("adaptive rtol 1e-6", 1e-6, None)):
    if rows is None:
        os.environ.pop("YMC_LV_ROWS", None)
    else:
        os.environ["YMC_LV_ROWS"] = rows
    rng = np.random.default_rng(4)
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low(rtol), np.full(2, np.sqrt(0.15)))
    run = ffi.Run(prob, 1, K, x0, 1, ffi.RECORD_MOMENTS, 0)
    run.advance(max(1, n // 10)); run.sync(); run.kernel_time_ms()
    run.advance(n); run.sync()
    ms, _ = run.kernel_time_ms()
    print("%-28s K=%d n=%d: %8.2f ms -> %.3g chain-steps/s" % (label, K, n, ms, K * n / ms * 1e3))
