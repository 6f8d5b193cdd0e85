"""Small ensembles beyond 256 parameters (random walk over a linear posterior), 200 transitions per launch:
    python tools/time_wide_small.py [K ...]      (YMC_WIDE_SPLIT=0: the fused kernel alone; YMC_WIDE_SPLIT_G / _W: the group)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from yagremcmc_amd import _ffi as ffi, lowering

Ks = [int(a) for a in sys.argv[1:]] or [16, 256, 1024, 2048]
ctx = ffi.default_context(0)
rng = np.random.default_rng(1)


def low(d, m, nd=3):
    l = lowering.Lowered(ffi.POSTERIOR_LINEAR, d)
    l.desc.n_data, l.desc.data_dim = nd, m
    for k, v in (("data", rng.normal(size=(nd, m))), ("noise_precision", np.full(m, 3.0)), ("prior_mean", np.zeros(d)),
                 ("prior_precision", np.full(d, 0.2)), ("fwd_matrix", rng.normal(size=(m, d)) / np.sqrt(d)), ("fwd_shift", rng.normal(size=m))):
        l.set_array(k, v)
    return l


n = 200
for d, m in ((512, 512), (320, 64), (512, 2048), (1024, 256)):
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low(d, m), np.full(d, 0.02))
    for K in Ks:
        x0 = rng.normal(size=(K, d)) * 0.1
        run = ffi.Run(prob, 1, K, x0, n, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT, 0)
        run.advance(n // 4); run.sync(); run.kernel_time_ms()
        run.rewind_record()
        run.advance(n); run.sync()
        ms, _ = run.kernel_time_ms()
        print("d=%d m=%d K=%d: %8.3f ms / %d steps = %6.1f us per transition -> %.3g steps/s" % (d, m, K, ms, n, ms / n * 1e3, K * n / ms * 1e3), flush=True)
        run.close()
