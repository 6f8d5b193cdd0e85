"""Two-level delayed acceptance beyond 256 parameters, 65 536 chains, S = 5: python tools/time_wide2.py   (YMC_WIDE2_W=8|16)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from yagremcmc_amd import _ffi as ffi, lowering

K = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ctx = ffi.default_context(0)
rng = np.random.default_rng(1)


def low(d, m, nd=3):
    l = lowering.Lowered(ffi.POSTERIOR_LINEAR, d)
    l.desc.n_data, l.desc.data_dim = nd, m
    for k, v in (("data", rng.normal(size=(nd, m))), ("noise_precision", np.full(m, 3.0)), ("prior_mean", np.zeros(d)),
                 ("prior_precision", np.full(d, 0.2)), ("fwd_matrix", rng.normal(size=(m, d)) / np.sqrt(d)), ("fwd_shift", rng.normal(size=m))):
        l.set_array(k, v)
    return l


for d, mc, mf in ((320, 32, 64), (320, 64, 320), (512, 64, 512), (512, 256, 512), (640, 64, 256), (1024, 64, 256)):
    S, n = 5, 4
    prob = lowering.make_problem(ctx, ffi.METHOD_MLDA2, low(d, mf), np.full(d, 0.02), surrogate=low(d, mc), sub_chain_length=S)
    run = ffi.Run(prob, 1, K, rng.normal(size=(K, d)) * 0.1, n, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT, 0)
    run.advance(n); run.sync(); run.kernel_time_ms()
    run.rewind_record()
    run.advance(n); run.sync()
    ms, _ = run.kernel_time_ms()
    rate = K * n / ms * 1e3
    print("two-level d=%d coarse %d fine %d: %7.2f ms / %d steps -> %.3g fine steps/s, fp64 %.2f" % (d, mc, mf, ms, n, rate, rate * 2.0 * d * (S * mc + mf) / 78.6e12), flush=True)
    run.close()
