"""Kernel times of the streamed matrix-core builds beyond d = 128 / beyond the LDS (random walk, dense proposal, two-level,
nested), with the fp64 MFMA fraction of their forward maps:   python tools/time_wide.py [K]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from yagremcmc_amd import _ffi as ffi, lowering

K = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ctx = ffi.default_context(0)
rng = np.random.default_rng(1)
PEAK = 78.6e12


def low(d, m, nd=3):
    l = lowering.Lowered(ffi.POSTERIOR_LINEAR, d)
    l.desc.n_data, l.desc.data_dim = nd, m
    for k, v in (("data", rng.normal(size=(nd, m))), ("noise_precision", np.full(m, 3.0)), ("prior_mean", np.zeros(d)),
                 ("prior_precision", np.full(d, 0.2)), ("fwd_matrix", rng.normal(size=(m, d)) / np.sqrt(d)), ("fwd_shift", rng.normal(size=m))):
        l.set_array(k, v)
    return l


class Pre:
    def __init__(s, l):
        s.low = l


real = lowering.lower_density
lowering.lower_density = lambda dd, rk4_steps=None: dd.low if isinstance(dd, Pre) else real(dd, rk4_steps)


def timed(label, prob, d, n, flop_per_step):
    x0 = rng.normal(size=(K, d)) * 0.1
    run = ffi.Run(prob, 1, K, x0, n, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT, 0)
    run.advance(max(1, n // 2)); run.sync(); run.kernel_time_ms()
    run.rewind_record()
    run.advance(n); run.sync()
    ms, _ = run.kernel_time_ms()
    rate = K * n / ms * 1e3
    print("%-58s %8.2f ms / %d steps -> %.3g steps/s, fp64 %.2f" % (label, ms, n, rate, rate * flop_per_step / PEAK))
    run.close()


for d, m in ((128, 128), (128, 512), (192, 192), (256, 256)):
    timed(f"random walk d={d} m={m}", lowering.make_problem(ctx, ffi.METHOD_MRW, low(d, m), np.full(d, 0.03)), d, 20, 2.0 * m * d)
for d, m in ((320, 64), (512, 512), (1024, 256)):   # beyond 256: the wide path (five plain launches per transition)
    timed(f"random walk d={d} m={m} (wide path)", lowering.make_problem(ctx, ffi.METHOD_MRW, low(d, m), np.full(d, 0.02)), d, 6, 2.0 * m * d)
timed("two-level d=512 coarse 64 fine 512 S=5 (wide path)",
      lowering.make_problem(ctx, ffi.METHOD_MLDA2, low(512, 512), np.full(512, 0.02), surrogate=low(512, 64), sub_chain_length=5), 512, 4,
      2.0 * 512 * (5 * 64 + 512))
timed("nested (2 surrogates, 4 base steps) d=512 outputs (64, 128, 512) (wide path)",
      lowering.make_nested_problem(ctx, Pre(low(512, 512)), [Pre(low(512, 64)), Pre(low(512, 128))], [3, 4], np.full(512, 0.02)), 512, 4,
      2.0 * 512 * (4 * 64 + 2 * (128 + 512)))
for d, m in ((128, 128), (256, 256)):
    B = rng.normal(size=(d, d)) * 0.02 / np.sqrt(d)
    L = np.linalg.cholesky(B @ B.T + 0.001 * np.eye(d))
    timed(f"random walk, dense proposal d={d} m={m}", lowering.make_problem(ctx, ffi.METHOD_MRW, low(d, m), np.ones(d), chol=L), d, 20,
          2.0 * (m + d) * d)
B = rng.normal(size=(512, 512)) * 0.02 / np.sqrt(512)
timed("random walk, dense proposal d=512 m=128 (wide path)",
      lowering.make_problem(ctx, ffi.METHOD_MRW, low(512, 128), np.ones(512), chol=np.linalg.cholesky(B @ B.T + 0.001 * np.eye(512))), 512, 6,
      2.0 * (128 + 512) * 512)
for d, mc, mf, S in ((128, 128, 128, 5), (128, 256, 256, 5), (256, 64, 256, 5), (256, 256, 256, 5)):
    timed(f"two-level d={d} coarse {mc} fine {mf} S={S}",
          lowering.make_problem(ctx, ffi.METHOD_MLDA2, low(d, mf), np.full(d, 0.03), surrogate=low(d, mc), sub_chain_length=S), d, 10,
          2.0 * d * (S * mc + mf))
for d, ms in ((128, (128, 128, 128)), (256, (64, 128, 256))):
    prob = lowering.make_nested_problem(ctx, Pre(low(d, ms[2])), [Pre(low(d, ms[0])), Pre(low(d, ms[1]))], [3, 4], np.full(d, 0.03))
    timed(f"nested (2 surrogates, 4 base steps) d={d} outputs {ms}", prob, d, 10, 2.0 * d * (4 * ms[0] + ms[1] + ms[2]))
