"""The fused wide kernel over the number of outputs at d = 512, 65 536 chains:  python tools/exp_wide_m.py [d]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from yagremcmc_amd import _ffi as ffi, lowering

d = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = 65536
ctx = ffi.default_context(0)
rng = np.random.default_rng(1)


def low(d, m, nd=3):
    l = lowering.Lowered(ffi.POSTERIOR_LINEAR, d)
    l.desc.n_data, l.desc.data_dim = nd, m
    for k, v in (("data", rng.normal(size=(nd, m))), ("noise_precision", np.full(m, 3.0)), ("prior_mean", np.zeros(d)),
                 ("prior_precision", np.full(d, 0.2)), ("fwd_matrix", rng.normal(size=(m, d)) / np.sqrt(d)), ("fwd_shift", rng.normal(size=m))):
        l.set_array(k, v)
    return l


for m in (128, 256, 512, 1024, 2048, 4096):
    n = 12 if m <= 1024 else 6
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low(d, m), np.full(d, 0.02))
    x0 = rng.normal(size=(K, d)) * 0.1
    run = ffi.Run(prob, 1, K, x0, n, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT, 0)
    run.advance(n); run.sync(); run.kernel_time_ms()
    run.rewind_record()
    run.advance(n); run.sync()
    ms, _ = run.kernel_time_ms()
    rate = K * n / ms * 1e3
    print("d=%d m=%d: %.3f ms per transition -> %.3g steps/s, fp64 %.2f" % (d, m, ms / n, rate, rate * 2.0 * m * d / 78.6e12), flush=True)
    run.close()
