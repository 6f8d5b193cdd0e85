import numpy as np, sys, os, time
sys.path.insert(0, os.getcwd())
from yagremcmc_amd import _ffi as ffi, lowering
ctx = ffi.default_context(0)
rng = np.random.default_rng(1)
d, m, nd, K = 128, 128, 5, 65536
def low():
    l = lowering.Lowered(ffi.POSTERIOR_LINEAR, d)
    l.desc.n_data, l.desc.data_dim = nd, m
    A = rng.normal(size=(m, d)) / np.sqrt(d)
    for k, v in (("data", rng.normal(size=(nd, m))), ("noise_precision", np.full(m, 3.0)), ("prior_mean", np.zeros(d)),
                 ("prior_precision", np.full(d, 0.2)), ("fwd_matrix", A), ("fwd_shift", rng.normal(size=m))):
        l.set_array(k, v)
    return l
class Pre:
    def __init__(s, l): s.low = l
real = lowering.lower_density
lowering.lower_density = lambda dd, rk4_steps=None: dd.low if isinstance(dd, Pre) else real(dd, rk4_steps)
lows = [low() for _ in range(3)]
prob = lowering.make_nested_problem(ctx, Pre(lows[2]), [Pre(lows[0]), Pre(lows[1])], [3, 4], np.full(d, 0.03))
x0 = rng.normal(size=(K, d)) * 0.1
run = ffi.Run(prob, 1, K, x0, 20, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT, 0)
run.advance(10); run.sync(); run.kernel_time_ms()
run.rewind_record()
run.advance(20); run.sync()
ms, n = run.kernel_time_ms()
print("YMC_TILE_STAGE", os.environ.get("YMC_TILE_STAGE"), "nested E=2 S=4 d=128 K=65536: %.2f ms per 20 top steps -> %.3g top steps/s" % (ms, K * 20 / ms * 1e3))

# the adaptive error model over two d = 128 models, 5 data rows (the LDS leaves room for 8 KB units only)
prob = lowering.make_problem(ctx, ffi.METHOD_AEM2, lows[2], np.full(d, 0.03), surrogate=lows[0], sub_chain_length=4, aem=(8, True))
run = ffi.Run(prob, 1, K, x0, 20, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT, 0)
run.advance(10); run.sync(); run.kernel_time_ms()
run.rewind_record()
run.advance(20); run.sync()
ms, n = run.kernel_time_ms()
print("YMC_TILE_STAGE", os.environ.get("YMC_TILE_STAGE"), "adaptive error model S=4 d=128 K=65536: %.2f ms per 20 fine steps -> %.3g fine steps/s" % (ms, K * 20 / ms * 1e3))

# the random walk with a dense proposal covariance at d = 128 (L read by every transition)
B = rng.normal(size=(d, d)) * 0.02 / np.sqrt(d)
L = np.linalg.cholesky(B @ B.T + 0.001 * np.eye(d))
prob = lowering.make_problem(ctx, ffi.METHOD_MRW, lows[2], np.ones(d), chol=np.tril(L))
run = ffi.Run(prob, 1, K, x0, 40, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT, 0)
run.advance(20); run.sync(); run.kernel_time_ms()
run.rewind_record()
run.advance(40); run.sync()
ms, n = run.kernel_time_ms()
print("YMC_TILE_STAGE", os.environ.get("YMC_TILE_STAGE"), "dense-proposal random walk d=128 K=65536: %.2f ms per 40 transitions -> %.3g steps/s" % (ms, K * 40 / ms * 1e3))
