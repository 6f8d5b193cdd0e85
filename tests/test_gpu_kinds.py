"""The worked extension kind on the device: a logistic-growth ODE solver registered through
csrc/ymc_kinds.def + csrc/kinds/logistic.h + yagremcmc_amd/kinds/logistic.py runs under every lane-per-chain sampler --
without a line of its own in any launch switch -- and equals its checker bit for bit."""
import numpy as np
import pytest

from oracle import oracle as o

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ffi():
    from yagremcmc_amd import _ffi
    return _ffi


@pytest.fixture(scope="module")
def ctx(ffi):
    return ffi.default_context(0)


def _spec(seed, n=6, steps=40):
    rng = np.random.default_rng(seed)
    x0 = rng.uniform(0.3, 2.5, size=n)
    truth = np.log(np.array([0.8, 5.0]))
    r, k = np.exp(truth)
    data = (k / (1.0 + (k / x0 - 1.0) * np.exp(-r * 3.0)) + 0.05 * rng.standard_normal(n)).reshape(n, 1)
    return dict(design=x0, t_final=3.0, rk4_steps=steps, data=data, noise_precision=np.array([400.0]),
                prior_mean=np.array([0.0, 1.0]), prior_precision=np.array([0.5, 0.5])), truth


def _low(ffi, spec, likelihood_only=False):
    from yagremcmc_amd import lowering
    low = lowering.Lowered(ffi.POSTERIOR_LOGISTIC, 2)
    low.desc.n_data, low.desc.data_dim = spec["data"].shape
    low.desc.t_final, low.desc.rk4_steps = spec["t_final"], spec["rk4_steps"]
    low.desc.likelihood_only = int(likelihood_only)
    for k in ("design", "data", "noise_precision", "prior_mean", "prior_precision"):
        low.set_array(k, spec[k])
    return low


def _run(ffi, prob, seed, x0, nsteps, splits=2, chain0=0, sub=False):
    flags = ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT | ffi.RECORD_LOGPOST | ffi.RECORD_MOMENTS | (ffi.RECORD_SUB_ACCEPT if sub else 0)
    run = ffi.Run(prob, seed, x0.shape[0], x0, nsteps, flags, chain0)
    base, rem = divmod(nsteps, splits)
    for i in range(splits):
        run.advance(base + (1 if i < rem else 0))
    run.sync()
    return run


def _same(run, e, out):
    assert np.array_equal(run.accept(), out["acc"]) and np.array_equal(run.trajectory(), out["traj"])
    assert np.array_equal(run.logpost(), out["logpost"])
    s = run.summary()
    assert np.array_equal(s["accepted"], e.acc_count) and np.array_equal(s["status"], e.status)
    assert np.array_equal(s["wf_mean"], e.wf_mean) and np.array_equal(s["wf_m2"], e.wf_m2)
    x, lp = run.state()
    assert np.array_equal(x, e.x) and np.array_equal(lp, e.lp)


@pytest.mark.parametrize("K", [70, 5000])       # 5 000 chains: the helper-wave (split) form of the random walk
def test_random_walk_over_the_extension_kind_equals_its_checker(ffi, ctx, K):
    from yagremcmc_amd import lowering
    spec, truth = _spec(11)
    rng = np.random.default_rng(K)
    x0 = truth + 0.2 * rng.standard_normal((K, 2))
    sigma = np.array([0.02, 0.015])
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW, _low(ffi, spec), sigma)
    run = _run(ffi, prob, 77, x0, 120, chain0=31)
    e = o.Ensemble(o.posterior_logistic(**spec), sigma, x0, 77, chain0=31)
    _same(run, e, e.advance(120))
    assert 0.1 < run.accept().mean() < 0.9
    # evaluate_log and ForwardModel.evaluate for batches
    thetas = truth + 0.3 * rng.standard_normal((200, 2))
    assert np.array_equal(prob.forward(thetas), o.forward(o.posterior_logistic(**spec), thetas))
    lp, st = o.logdensity(o.posterior_logistic(**spec), thetas)
    got = prob.logdensity(thetas)
    assert np.array_equal(got[0] if isinstance(got, tuple) else got, lp)


def test_pcn_and_two_level_over_the_extension_kind_equal_their_checkers(ffi, ctx):
    from yagremcmc_amd import lowering
    spec, truth = _spec(12)
    rng = np.random.default_rng(9)
    K = 130
    x0 = truth + 0.2 * rng.standard_normal((K, 2))
    psig = np.sqrt(np.reciprocal(spec["prior_precision"]))
    centred = dict(spec, prior_mean=np.zeros(2))
    prob = lowering.make_problem(ctx, ffi.METHOD_PCN, _low(ffi, centred, likelihood_only=True), psig, pcn_step_size=0.01)
    run = _run(ffi, prob, 5, x0, 90)
    e = o.Ensemble(dict(o.posterior_logistic(**centred), likelihood_only=1), psig, x0, 5, pcn_step_size=0.01)
    _same(run, e, e.advance(90))
    # two-level delayed acceptance: a coarse integrator (6 steps) under the fine one (40), S = 3
    coarse = dict(spec, rk4_steps=6)
    sigma = np.array([0.02, 0.015])
    prob = lowering.make_problem(ctx, ffi.METHOD_MLDA2, _low(ffi, spec), sigma, surrogate=_low(ffi, coarse), sub_chain_length=3)
    run = _run(ffi, prob, 6, x0, 80, sub=True)
    e = o.Ensemble(o.posterior_logistic(**spec), sigma, x0, 6, surrogate=o.posterior_logistic(**coarse), sub_chain_length=3)
    out = e.advance(80, want_sub=True)
    _same(run, e, out)
    assert np.array_equal(run.sub_accept(), out["sub_acc"])


def test_extension_kind_through_the_builders(ffi):
    """MRWBuilder.bayesModel over the registered solver class (what a user of the reference writes), the solver's own
    ForwardModel.evaluate on the device, and the description check of the kind"""
    from yagremcmc_amd import lowering
    from yagremcmc_amd.chain.method.mrw import MRWBuilder
    from yagremcmc_amd.kinds.logistic import LogisticGrowthSolver
    from yagremcmc_amd.model.forwardModel import ForwardModel
    from yagremcmc_amd.parameter.vector import ParameterVector
    from yagremcmc_amd.statistics.bayesModel import BayesianRegressionModel
    from yagremcmc_amd.statistics.covariance import DiagonalCovarianceMatrix, IIDCovarianceMatrix
    from yagremcmc_amd.statistics.data import Data
    from yagremcmc_amd.statistics.gaussian import Gaussian
    from yagremcmc_amd.statistics.likelihood import AdditiveGaussianNoiseLikelihood
    from yagremcmc_amd.statistics.noise import CentredGaussianNoise
    spec, truth = _spec(13)
    solver = LogisticGrowthSolver(spec["design"], {'T': 3.0, 'rk4Steps': 40})
    fm = ForwardModel(solver)
    ev = fm.evaluate(ParameterVector(truth))
    assert ev.shape == (6, 1) and np.array_equal(ev.reshape(-1), o.forward(o.posterior_logistic(**spec), truth[None])[0])
    assert np.abs(ev - solver.exact(truth)).max() < 1e-6
    noise = CentredGaussianNoise(IIDCovarianceMatrix(1, 1.0 / 400.0))
    prior = Gaussian(ParameterVector(spec["prior_mean"]), IIDCovarianceMatrix(2, 2.0))
    b = MRWBuilder()
    b.proposalCovariance = DiagonalCovarianceMatrix(np.array([0.02, 0.015]) ** 2)
    b.bayesModel = BayesianRegressionModel(AdditiveGaussianNoiseLikelihood(Data(spec["data"]), fm, noise), prior)
    b.nChains, b.seed, b.recordLogPosterior = 96, 21, True
    mc = b.build_method()
    mc.run(151, ParameterVector(truth), verbose=False)
    sigma = lowering.proposal_sigma(b.proposalCovariance, 2)
    p = o.posterior_logistic(**dict(spec, noise_precision=np.reciprocal(np.reciprocal(np.array([400.0])) * 1.0)))
    p["noise_precision"] = np.asarray(noise.covariance._precision, dtype=float).reshape(-1)
    p["prior_precision"] = np.asarray(prior.covariance._precision, dtype=float).reshape(-1)
    e = o.Ensemble(p, sigma, np.tile(truth, (96, 1)), 21)
    out = e.advance(150)
    assert np.array_equal(mc.chain.ensemble_trajectory[1:], out["traj"]) and np.array_equal(mc.accept_decisions, out["acc"])
    assert np.array_equal(mc.log_posterior[1:], out["logpost"])
    post = mc.chain.ensemble_trajectory[50:].reshape(-1, 2).mean(axis=0)
    assert np.abs(post - truth).max() < 0.25
    # the kind's own description check reaches the caller as the library's error
    bad = dict(spec, design=np.array([0.5, -1.0, 1.0, 1.0, 1.0, 1.0]))
    with pytest.raises(ffi.YmcError, match="logistic.*initial populations.*positive"):
        lowering.make_problem(ffi.default_context(0), ffi.METHOD_MRW, _low(ffi, bad), np.array([0.1, 0.1]))
