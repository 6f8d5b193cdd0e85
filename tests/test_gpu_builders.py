"""The drop-in surface on the GPU: the reference's own tests, restated for the
ensemble builders (test_mcmc_1d.py, test_mcmc_2d.py, test_mlda.py,
test_inference_mcmc_singleLevel.py), plus single-chain shape compatibility."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as o  # noqa: E402


def _imports():
    from yagremcmc_amd.chain.diagnostics import AcceptanceRateDiagnostics, DummyDiagnostics, FullDiagnostics
    from yagremcmc_amd.chain.method.mlda import MLDABuilder
    from yagremcmc_amd.chain.method.mrw import MetropolisedRandomWalk, MRWBuilder
    from yagremcmc_amd.parameter.scalar import ScalarParameter
    from yagremcmc_amd.parameter.vector import ParameterVector
    from yagremcmc_amd.statistics.covariance import DiagonalCovarianceMatrix, IIDCovarianceMatrix
    from yagremcmc_amd.test.testSetup import GaussianTargetDensity1d, GaussianTargetDensity2d
    return locals()


@pytest.mark.parametrize("diag", ["DummyDiagnostics", "AcceptanceRateDiagnostics", "FullDiagnostics"])
def test_mcmc_1d_single_chain_like_the_reference(diag):
    """test_mcmc_1d.py:58-116 (seeded statistical test) on ONE chain: trajectory is
    a list of chainLength coefficient arrays; mean/var within 1e-1."""
    m = _imports()
    tgt = m["GaussianTargetDensity1d"](m["ScalarParameter"](np.array([1.0])), 1.0)
    mc = m["MetropolisedRandomWalk"](tgt, m["IIDCovarianceMatrix"](1, 1.0), m[diag](), seed=18)
    n = 15000
    mc.run(n, m["ScalarParameter"](np.array([-3.0])), verbose=False)
    traj = mc.chain.trajectory
    assert isinstance(traj, list) and len(traj) == n and traj[0].shape == (1,)
    states = np.array(traj)
    samples = states[200::3]
    assert abs(samples.mean() - 1.0) < 1e-1 and abs(samples.var() - 1.0) < 1e-1
    # and it is the oracle's chain, bit for bit
    e = o.Ensemble(o.gauss_iso1d(1.0, 1.0), np.ones(1), np.array([[-3.0]]), 18, eq_mode=o.EQ_ISCLOSE)
    assert np.array_equal(states[1:, 0], e.advance(n - 1)["traj"][:, 0, 0])
    if diag != "DummyDiagnostics":
        assert mc.diagnostics.global_acceptance_rate() == e.acc_count[0] / (n - 1)
        assert isinstance(mc.diagnostics._decisions if diag == "AcceptanceRateDiagnostics"
                          else mc.diagnostics._diagnostics._decisions, list)
    if diag == "FullDiagnostics":
        assert np.array_equal(mc.diagnostics.mean(), e.wf_mean[0])
        assert np.array_equal(mc.diagnostics.marginal_variance(), e.wf_m2[0] / (n - 2.0))


@pytest.mark.parametrize("proposal", ["iid", "indep"])
def test_mcmc_2d_moments_ensemble(proposal):
    """test_mcmc_2d.py:18-81 with 2048 chains: ensemble moments are far tighter
    than the reference's single-chain tolerances (5e-2 / 1e-1)."""
    m = _imports()
    mean = np.array([1.0, 1.5])
    cov = np.array([[1.2, -0.2], [-0.2, 0.4]])
    b = m["MRWBuilder"]()
    b.explicitTarget = m["GaussianTargetDensity2d"](m["ParameterVector"](mean), cov)
    b.proposalCovariance = (m["IIDCovarianceMatrix"](2, 0.25) if proposal == "iid"
                            else m["DiagonalCovarianceMatrix"](np.array([cov[0, 0], cov[1, 1]])))
    b.diagnostics = m["FullDiagnostics"]()
    b.nChains, b.seed = 2048, 116
    mc = b.build_method()
    mc.run(2000, m["ParameterVector"](np.array([-2.0, 0.0])), verbose=True)
    states = mc.chain.trajectory
    assert states.shape == (2000, 2048, 2)
    samples = states[200::5].reshape(-1, 2)
    assert np.allclose(samples.mean(axis=0), mean, atol=2e-2)
    assert np.allclose(np.cov(samples.T), cov, atol=3e-2)
    rate = mc.diagnostics.global_acceptance_rate()
    assert 0.1 < rate < 0.9
    assert np.allclose(mc.diagnostics.per_chain_acceptance_rate().mean(), rate)
    assert mc.diagnostics.marginal_variance().shape == (2,)


def test_two_level_builder_and_perfect_surrogate():
    """test_mlda.py:94-186"""
    m = _imports()
    mean = np.array([1.0, 1.5])
    cov = np.array([[2.4, -0.5], [-0.5, 0.7]])
    fine = m["GaussianTargetDensity2d"](m["ParameterVector"](mean), cov)
    coarse = m["GaussianTargetDensity2d"](m["ParameterVector"](mean + np.array([0.15, -0.2])),
                                          2.0 * cov + np.array([[0.1, 1.2], [1.2, 0.05]]))
    b = m["MLDABuilder"]()
    b.explicitTarget, b.surrogateTargets = fine, [coarse]
    b.baseProposalCovariance = m["IIDCovarianceMatrix"](2, 1.0)
    b.subChainLengths = [10]
    b.nChains, b.seed = 1024, 42
    mc = b.build_method()
    mc.run(1500, m["ParameterVector"](np.array([-8.0, -7.0])), verbose=False)
    rate = mc.diagnostics.global_acceptance_rate()
    assert 0.1 < rate < 0.9
    samples = mc.chain.trajectory[100:].reshape(-1, 2)
    assert np.allclose(samples.mean(axis=0), mean, atol=0.05)
    # perfect surrogate => acceptance rate 1 up to the (never occurring) all-rejected sub-chains
    b2 = m["MLDABuilder"]()
    b2.explicitTarget, b2.surrogateTargets = fine, [fine]
    b2.baseProposalCovariance = m["IIDCovarianceMatrix"](2, 1.0)
    b2.subChainLengths = [10]
    b2.nChains = 512
    mc2 = b2.build_method()
    mc2.run(500, m["ParameterVector"](np.array([1.0, 1.5])), verbose=False)
    assert abs(mc2.diagnostics.global_acceptance_rate() - 1.0) < 1e-3


def test_two_level_surrogate_diagnostics_come_from_the_sub_chain_record():
    """mcmc.surrogate(0).diagnostics of a one-surrogate run (mlda.py:84-89: the coarse chain's own diagnostics): counted
    from the record of the sub-chain decisions -- without that record the request is refused, not answered with 0."""
    m = _imports()
    mean = np.array([1.0, 1.5])
    cov = np.array([[2.4, -0.5], [-0.5, 0.7]])
    fine = m["GaussianTargetDensity2d"](m["ParameterVector"](mean), cov)
    coarse = m["GaussianTargetDensity2d"](m["ParameterVector"](mean + np.array([0.15, -0.2])), 2.0 * cov)
    b = m["MLDABuilder"]()
    b.explicitTarget, b.surrogateTargets = fine, [coarse]
    b.baseProposalCovariance = m["IIDCovarianceMatrix"](2, 1.0)
    b.subChainLengths = [4]
    b.nChains, b.seed = 128, 21
    mc = b.build_method()
    mc.run(200, m["ParameterVector"](np.array([-8.0, -7.0])), verbose=False)
    with pytest.raises(RuntimeError, match="recordSubChain"):
        mc.surrogate(0).diagnostics
    b.recordSubChain = True
    ms = b.build_method()
    ms.run(200, m["ParameterVector"](np.array([-8.0, -7.0])), verbose=False)
    assert np.array_equal(ms.accept_decisions, mc.accept_decisions)
    sub = ms.sub_chain_decisions
    assert sub.shape == (199, 4, 128) and 0.05 < sub.mean() < 0.95
    assert ms.surrogate(0).diagnostics.global_acceptance_rate() == sub.mean()
    assert np.array_equal(ms.surrogate(0).diagnostics.per_chain_acceptance_rate(), sub.mean(axis=(0, 1)))
    assert ms.surrogate(0).target is coarse
    # diagnostics handed in for the surrogate (the reference's way) switch the record on by themselves
    b.recordSubChain = False
    b.surrogateDiagnostics = [m["AcceptanceRateDiagnostics"]()]
    md = b.build_method()
    md.run(200, m["ParameterVector"](np.array([-8.0, -7.0])), verbose=False)
    assert md.surrogate(0).diagnostics.global_acceptance_rate() == sub.mean()


def _mlda_test_data(m):
    """test_mlda.py:14-43"""
    mean = np.array([1.0, 1.5])
    cov = np.array([[2.4, -0.5], [-0.5, 0.7]])
    PV, G = m["ParameterVector"], m["GaussianTargetDensity2d"]
    return dict(
        tgtMean=mean, tgt=G(PV(mean), cov),
        base=G(PV(mean + np.array([-0.05, 0.01])), 3.0 * np.array([[2.8, -0.1], [-0.1, 1.7]])),
        fine=G(PV(mean + np.array([0.0, -0.01])), 1.5 * np.array([[2.4, -0.3], [-0.3, 1.1]])))


def test_mlda_chain_with_two_surrogates_like_the_reference():
    """test_mlda.py:47-91 (subChainLengths [6, 6]) and example_mcmc_2d_hierarchical.py, on an
    ensemble; every chain is the oracle's nested chain bit for bit."""
    m = _imports()
    d = _mlda_test_data(m)
    b = m["MLDABuilder"]()
    b.explicitTarget = d["tgt"]
    b.surrogateTargets = [d["base"], d["fine"]]
    b.baseProposalCovariance = m["IIDCovarianceMatrix"](2, 1.0)
    b.subChainLengths = [6, 6]
    b.nChains, b.seed = 256, 42
    mc = b.build_method()
    assert mc.nSurrogates == 2
    mc.run(2000, m["ParameterVector"](np.array([-8.0, -7.0])), verbose=False)
    assert mc.chain.trajectory.shape == (2000, 256, 2)
    assert 0.1 < mc.diagnostics.global_acceptance_rate() < 0.9
    est = mc.chain.trajectory[500::5].reshape(-1, 2).mean(axis=0)
    assert np.allclose(est, d["tgtMean"], atol=0.1)
    rates = mc.level_acceptance_rates()
    assert len(rates) == 2 and all(0.05 < r < 0.95 for r in rates)
    assert mc.surrogate(0).diagnostics.global_acceptance_rate() == rates[0]
    assert mc.surrogate(1).diagnostics.global_acceptance_rate() == 0.  # its accept step never runs
    assert mc.surrogate(-1).target is d["fine"]
    # == the oracle's nest
    from yagremcmc_amd import lowering
    w = [lowering._whitening_of(x) for x in (d["tgt"], d["base"], d["fine"])]
    og = [o.gauss_whitened(*x) for x in w]
    P, Q, steps = o.nested_levels(og[1:], [6, 6], og[0])
    e = o.NestedEnsemble(P, Q, steps, np.ones(2), np.tile([-8.0, -7.0], (256, 1)), 42)
    out = e.advance(300)
    assert np.array_equal(mc.chain.trajectory[1:301], out["traj"])
    assert np.array_equal(mc.accept_decisions[:300], out["acc"])


def test_mlda_five_level_method_like_the_reference():
    """test_mlda.py:189-256: four surrogates, subChainLengths [6, 4, 4, 3]."""
    m = _imports()
    d = _mlda_test_data(m)
    PV, G = m["ParameterVector"], m["GaussianTargetDensity2d"]
    tm = d["tgtMean"]
    means = [tm + np.array([0.2, -0.3]), tm + np.array([-0.1, 0.1]), tm + np.array([0.05, -0.05]), tm + 0.0]
    covs = [8. * np.array([[3.0, -0.2], [-0.2, 1.5]]), 6. * np.array([[2.7, -0.25], [-0.25, 1.3]]),
            4. * np.array([[2.5, -0.3], [-0.3, 1.1]]), 2. * np.array([[2.4, -0.35], [-0.35, 1.0]])]
    b = m["MLDABuilder"]()
    b.explicitTarget = d["tgt"]
    b.surrogateTargets = [G(PV(mu), c) for mu, c in zip(means, covs)]
    b.baseProposalCovariance = m["IIDCovarianceMatrix"](2, 1.0)
    b.subChainLengths = [6, 4, 4, 3]
    b.nChains, b.seed = 512, 789
    mc = b.build_method()
    mc.run(1000, PV(np.array([2.0, -3.0])), verbose=False)
    assert 0.1 < mc.diagnostics.global_acceptance_rate() < 0.9
    est = mc.chain.trajectory[200::3].reshape(-1, 2).mean(axis=0)
    assert np.allclose(est, tm, atol=0.2)
    assert len(mc.level_acceptance_rates()) == 3


def test_mlda_three_surrogates_fail_like_the_reference():
    """With an odd number >= 3 of surrogates the reference's run() raises AttributeError
    (mlda.py:23-33 calls run() of a level whose proposal method is the base random walk)."""
    m = _imports()
    d = _mlda_test_data(m)
    b = m["MLDABuilder"]()
    b.explicitTarget = d["tgt"]
    b.surrogateTargets = [d["base"], d["fine"], d["fine"]]
    b.baseProposalCovariance = m["IIDCovarianceMatrix"](2, 1.0)
    b.subChainLengths = [2, 2, 2]
    mc = b.build_method()
    with pytest.raises(AttributeError, match="set_state"):
        mc.run(10, m["ParameterVector"](np.array([0.0, 0.0])), verbose=False)
    b.subChainLengths = [2, 2]
    with pytest.raises(ValueError, match="Number of sub-chain lengths"):
        b.build_method()


def test_dense_proposal_covariance_through_the_builder(golden):
    """MRWBuilder.proposalCovariance = DenseCovarianceMatrix(...) (covariance.py:69-94)."""
    m = _imports()
    from yagremcmc_amd.statistics.covariance import DenseCovarianceMatrix
    g = golden("f13_mrw_2d_dense_proposal.npz")
    cov = DenseCovarianceMatrix(g["proposal_cov"])
    assert np.array_equal(cov.cholFactor_, g["chol"])                  # same LAPACK factor as the reference's
    assert np.allclose(cov.dense(), g["proposal_cov"]) and cov.dimension == 2
    x = np.array([0.3, -1.2])
    assert np.allclose(cov.apply_inverse(cov.dense() @ x), x)
    b = m["MRWBuilder"]()
    b.explicitTarget = m["GaussianTargetDensity2d"](m["ParameterVector"](np.array([1.0, 1.5])),
                                                    np.array([[2.4, -0.5], [-0.5, 0.7]]))
    b.proposalCovariance = cov
    b.nChains, b.seed = g["x0"].shape[0], int(g["seed"])
    mc = b.build_method()
    mc.run(int(g["n_states"]), m["ParameterVector"](g["x0"][0].copy()), verbose=False)
    assert np.array_equal(mc.accept_decisions, g["acc"])
    assert np.abs(mc.chain.trajectory - g["traj"]).max() <= 1e-12
    # not lowered for delayed acceptance
    from yagremcmc_amd.chain.method.mlda import MLDABuilder
    b2 = MLDABuilder()
    b2.explicitTarget, b2.surrogateTargets = b.explicitTarget, [b.explicitTarget]
    b2.baseProposalCovariance, b2.subChainLengths = cov, [3]
    with pytest.raises(NotImplementedError):
        b2.build_method().run(5, m["ParameterVector"](np.zeros(2)), verbose=False)


def test_adaptive_error_model_through_the_builder(golden):
    """example_inference_linearModel_twoLevel.py:95-100,181-191 wiring (AEMBuilder over a model
    hierarchy of AEMLikelihoods); chain 0..K-1 == the live reference's chains of fixture F12."""
    from yagremcmc_amd.chain.diagnostics import FullDiagnostics
    from yagremcmc_amd.chain.method.aem import AEMBuilder
    from yagremcmc_amd.chain.method.mlda import MLDABuilder
    from yagremcmc_amd.exampleSetup import ExampleLinearModelSolver
    from yagremcmc_amd.model.forwardModel import ForwardModel
    from yagremcmc_amd.parameter.vector import ParameterVector
    from yagremcmc_amd.statistics.covariance import IIDCovarianceMatrix
    from yagremcmc_amd.statistics.data import Data
    from yagremcmc_amd.statistics.gaussian import Gaussian
    from yagremcmc_amd.statistics.likelihood import AdditiveGaussianNoiseLikelihood, AEMLikelihood
    from yagremcmc_amd.statistics.modelHierarchy import BayesianRegressionModelHierarchy
    from yagremcmc_amd.statistics.noise import CentredGaussianNoise
    from yagremcmc_amd.utility.hierarchy import Hierarchy, SharedComponent
    g = golden("f12_aem_linear.npz")
    K, n = g["x0"].shape[0], int(g["n_states"])
    noise = CentredGaussianNoise(IIDCovarianceMatrix(2, np.sqrt(0.3) ** 2))  # as the example forms it
    assert np.array_equal(noise.covariance._precision, g["noise_precision"])
    prior = Gaussian(ParameterVector(g["prior_mean"].copy()), IIDCovarianceMatrix(2, 5.0))
    data = Data(g["data"])
    mk = lambda A, b, cls, *extra: cls(data, ForwardModel(ExampleLinearModelSolver(A, b)), noise, *extra)
    lik_s = mk(g["A_coarse"], g["b_coarse"], AEMLikelihood, int(g["min_data_size"]), True)
    lik_t = mk(g["A_fine"], g["b_fine"], AEMLikelihood, int(g["min_data_size"]), True)
    with pytest.raises(ValueError, match="Smallest"):
        mk(g["A_fine"], g["b_fine"], AEMLikelihood, 1)
    model = BayesianRegressionModelHierarchy(Hierarchy([lik_s, lik_t]), SharedComponent(prior, 2))
    b = AEMBuilder()
    b.baseProposalCovariance = IIDCovarianceMatrix(2, 0.5)
    b.subChainLengths = [int(g["S"])]
    b.bayesModel = model
    b.targetDiagnostics = FullDiagnostics()
    b.nChains, b.seed = K, int(g["seed"])
    mc = b.build_method()
    mc.run(n, ParameterVector(np.zeros(2)), verbose=False)
    assert np.array_equal(mc.chain.trajectory, g["traj"])
    assert np.array_equal(mc.accept_decisions, g["acc"])
    acc = lik_s.accumulator
    assert np.array_equal(acc.nData, g["err_n"])
    assert np.allclose(acc.mean(), g["err_mean"], rtol=1e-12, atol=0)
    assert np.allclose(acc.marginal_variance(), g["err_mvar"], rtol=1e-12, atol=0)
    assert lik_t.accumulator.nData == 0
    assert mc.diagnostics.marginal_variance().shape == (2,)
    # builder validation, aem.py:66-79
    plain = BayesianRegressionModelHierarchy(
        Hierarchy([mk(g["A_coarse"], g["b_coarse"], AdditiveGaussianNoiseLikelihood), lik_t]), SharedComponent(prior, 2))
    b.bayesModel = plain
    with pytest.raises(ValueError, match="Likelihood on level 0 is not adaptive"):
        b.build_method()
    # the learned shift moves the coarse posterior towards the fine one: more fine-level acceptances
    # than vanilla two-level sampling on the same models and draws
    v = MLDABuilder()
    v.baseProposalCovariance = IIDCovarianceMatrix(2, 0.5)
    v.subChainLengths = [int(g["S"])]
    v.bayesModel = plain
    v.nChains, v.seed = K, int(g["seed"])
    mv = v.build_method()
    mv.run(n, ParameterVector(np.zeros(2)), verbose=False)
    assert mc.diagnostics.global_acceptance_rate() > mv.diagnostics.global_acceptance_rate()


@pytest.mark.parametrize("proposal", ["iid", "indep"])
def test_lotka_volterra_inference_through_the_builder(proposal):
    """test_inference_mcmc_singleLevel.py:28-118 wiring; posterior mean near the truth."""
    import yagremcmc_amd.test.testSetup as setup
    from yagremcmc_amd.chain.method.mrw import MRWBuilder
    from yagremcmc_amd.model.forwardModel import ForwardModel
    from yagremcmc_amd.statistics.bayesModel import BayesianRegressionModel
    from yagremcmc_amd.statistics.covariance import DiagonalCovarianceMatrix, IIDCovarianceMatrix
    from yagremcmc_amd.statistics.gaussian import Gaussian
    from yagremcmc_amd.statistics.likelihood import AdditiveGaussianNoiseLikelihood
    from yagremcmc_amd.statistics.noise import CentredGaussianNoise
    np.random.seed(16)
    config = {'T': 6., 'alpha': 0.8, 'gamma': 0.4, 'nData': 5, 'dataDim': 2, 'solver': 'LSODA', 'rtol': 1e-4}
    design = np.array([[0.1, 0.9], [0.5, 0.5], [1., 0.5], [0.5, 1.], [2.5, 1.5]])
    solver = setup.LotkaVolterraSolver(design, config)
    truth = setup.LotkaVolterraParameter.from_interpolation(np.array([0.4, 0.6]))
    data = setup.generate_synthetic_data(truth, solver, 0.05)
    assert data.array.shape == (5, 2)
    # ForwardModel.evaluate goes through the device and agrees with the oracle's RK4
    f = ForwardModel(solver).evaluate(truth)
    p = o.posterior_lv(design, 6.0, 0.8, 0.4, 50, data.array, np.full(2, 20.0), np.zeros(2), np.full(2, 50.0))
    assert np.array_equal(f, o.forward(p, truth.coefficient).reshape(5, 2))
    prior = Gaussian(setup.LotkaVolterraParameter.from_coefficient(np.zeros(2)), IIDCovarianceMatrix(2, 0.02))
    like = AdditiveGaussianNoiseLikelihood(data, ForwardModel(solver), CentredGaussianNoise(IIDCovarianceMatrix(2, 0.05)))
    b = MRWBuilder()
    b.proposalCovariance = (IIDCovarianceMatrix(2, 0.02) if proposal == "iid"
                            else DiagonalCovarianceMatrix(np.array([0.02, 0.01])))
    b.bayesModel = BayesianRegressionModel(like, prior)
    b.nChains, b.recordLogPosterior = 256, True
    mc = b.build_method()
    init = setup.LotkaVolterraParameter.from_coefficient(np.array([-0.6, -0.3]))
    mc.run(1000, init, verbose=False)
    states = mc.chain.trajectory
    post = states[200::5].reshape(-1, 2).mean(axis=0)
    # the prior N(0, 0.02 I) dominates 5 noisy rows: the posterior mean sits between 0 and the truth
    assert np.all(np.abs(post) <= np.abs(truth.coefficient) + 0.1)
    assert (mc.chain_status == 0).all()
    # target.evaluate_log on the device == recorded log-posterior of that state
    lp = mc.log_posterior
    k, n = 3, 500
    got = mc.target.evaluate_log(setup.LotkaVolterraParameter.from_coefficient(states[n, k].copy()))
    assert got == lp[n, k]
    e = o.Ensemble(p | dict(noise_precision=np.reciprocal(np.full(2, 0.05)), prior_precision=np.reciprocal(np.full(2, 0.02))),
                   np.sqrt(np.reciprocal(b.proposalCovariance._precision)), np.tile(init.coefficient, (256, 1)), 0)
    assert np.array_equal(states[1:], e.advance(999)["traj"])


def test_lotka_volterra_example_from_its_own_start_with_the_adaptive_integrator_f24(golden):
    """example_inference_lotkaVolterra_singleLevel.py:29-106 through the builders with 'adaptive': True in the solver's
    config: from theta = (-7, 2.8) the chains are those of the reference's loop (fixture F24), no chain is flagged;
    without it (RK4 in 50 steps) the start itself is a non-finite evaluation and every chain is flagged."""
    import yagremcmc_amd.test.testSetup as setup
    from yagremcmc_amd.chain.method.mrw import MRWBuilder
    from yagremcmc_amd.model.forwardModel import ForwardModel
    from yagremcmc_amd.statistics.bayesModel import BayesianRegressionModel
    from yagremcmc_amd.statistics.covariance import IIDCovarianceMatrix
    from yagremcmc_amd.statistics.data import Data
    from yagremcmc_amd.statistics.gaussian import Gaussian
    from yagremcmc_amd.statistics.likelihood import AdditiveGaussianNoiseLikelihood
    from yagremcmc_amd.statistics.noise import CentredGaussianNoise
    g = golden("f24_lotka_volterra_adaptive.npz")
    n, K = int(g["n_states"]), g["x0"].shape[0]
    config = {'T': 10., 'alpha': 0.8, 'gamma': 0.4, 'nData': 10, 'dataDim': 2, 'solver': 'LSODA', 'rtol': 1e-4}
    with pytest.raises(ValueError, match="exclude"):
        setup.LotkaVolterraSolver(np.asarray(g["design"]), dict(config, adaptive=True, rk4Steps=80))

    def sampler(cfg):
        solver = setup.LotkaVolterraSolver(np.asarray(g["design"]), cfg)
        like = AdditiveGaussianNoiseLikelihood(Data(np.asarray(g["data"])), ForwardModel(solver),
                                               CentredGaussianNoise(IIDCovarianceMatrix(2, 0.04)))
        prior = Gaussian(setup.LotkaVolterraParameter.from_coefficient(np.zeros(2)), IIDCovarianceMatrix(2, 1.4))
        b = MRWBuilder()
        b.proposalCovariance = IIDCovarianceMatrix(2, 0.15)
        b.bayesModel = BayesianRegressionModel(like, prior)
        b.nChains, b.seed, b.recordLogPosterior = K, int(g["seed"]), True
        return b.build_method(), solver

    mc, solver = sampler(dict(config, adaptive=True))
    init = setup.LotkaVolterraParameter.from_coefficient(np.array([-7.0, 2.8]))
    mc.run(n, init, verbose=False)
    assert np.array_equal(mc.chain.trajectory, g["traj"]) and np.array_equal(mc.accept_decisions, g["acc"])
    assert np.abs(mc.log_posterior - g["logpost"]).max() <= 1e-12 * np.abs(g["logpost"]).max()
    assert (mc.chain_status == 0).all()
    # the forward map alone, and a trajectory for plotting (testSetup.py:142-162)
    f = ForwardModel(solver).evaluate(init)
    tight = g["forward_dop853_rtol1e12"][24]
    assert np.array_equal(g["thetas"][24], [-7.0, 2.8]) and (np.abs(f - tight) / (1e-3 + np.abs(tight))).max() < 1e-2
    t, y = solver.full_solution(init, np.asarray(g["design"])[0])
    assert t.shape == (101,) and y.shape == (2, 101) and np.array_equal(y[:, -1], f[0]) and np.isfinite(y).all()
    rk4, _ = sampler(config)
    rk4.run(20, init, verbose=False)
    assert (rk4.log_posterior[0] == -np.inf).all() and (rk4.chain_status != 0).any()


def test_reference_style_objects_are_accepted_by_duck_typing():
    """A user who keeps the reference's own plug-in instances: recognised by class
    name + attributes (here: stand-ins with the reference's attribute names)."""
    from types import SimpleNamespace
    from yagremcmc_amd.chain.method.mrw import MRWBuilder
    from yagremcmc_amd.parameter.vector import ParameterVector
    mean, cov = np.array([1.0, 1.5]), np.array([[2.4, -0.5], [-0.5, 0.7]])
    w, v = np.linalg.eigh(cov)
    cov_object = SimpleNamespace(_LP=v * np.sqrt(1.0 / w), _log_pdet=float(np.sum(np.log(w))), rank=2)
    RefTarget = type("GaussianTargetDensity2d", (), {})
    t = RefTarget()
    t.dist_ = SimpleNamespace(mean=mean, cov=cov, cov_object=cov_object)
    RefCov = type("IIDCovarianceMatrix", (), {})
    c = RefCov()
    c._precision = np.reciprocal(np.full(2, 1.0))
    RefDiag = type("FullDiagnostics", (), {})
    b = MRWBuilder()
    b.explicitTarget, b.proposalCovariance, b.diagnostics = t, c, RefDiag()
    b.nChains, b.seed = 64, 5
    mc = b.build_method()
    mc.run(300, ParameterVector(np.array([-8.0, -7.0])), verbose=False)
    e = o.Ensemble(o.gauss_whitened(mean, cov_object._LP, 2 * np.log(2 * np.pi) + cov_object._log_pdet),
                   np.ones(2), np.tile([-8.0, -7.0], (64, 1)), 5)
    assert np.array_equal(mc.chain.trajectory[1:], e.advance(299)["traj"])
    assert np.array_equal(mc.diagnostics.per_chain_mean(), e.wf_mean)


def test_unsupported_plugins_raise():
    from yagremcmc_amd import UnsupportedPlugin
    from yagremcmc_amd.chain.method.mrw import MRWBuilder
    from yagremcmc_amd.parameter.vector import ParameterVector
    from yagremcmc_amd.statistics.covariance import IIDCovarianceMatrix

    class MyDensity:
        def evaluate_log(self, p):
            return 0.0

    b = MRWBuilder()
    b.explicitTarget, b.proposalCovariance = MyDensity(), IIDCovarianceMatrix(2, 1.0)
    with pytest.raises(UnsupportedPlugin):
        b.build_method().run(10, ParameterVector(np.zeros(2)), verbose=False)


def test_two_level_over_a_model_hierarchy_f7(golden):
    """MLDABuilder.bayesModel = BayesianRegressionModelHierarchy (mlda.py:315-335), the shipped linear
    two-level example: the device chains equal the live reference's, bit for bit."""
    from yagremcmc_amd.chain.method.mlda import MLDABuilder
    from yagremcmc_amd.exampleSetup import ExampleLinearModelSolver
    from yagremcmc_amd.model.forwardModel import ForwardModel
    from yagremcmc_amd.parameter.vector import ParameterVector
    from yagremcmc_amd.statistics.covariance import IIDCovarianceMatrix
    from yagremcmc_amd.statistics.data import Data
    from yagremcmc_amd.statistics.gaussian import Gaussian
    from yagremcmc_amd.statistics.likelihood import AdditiveGaussianNoiseLikelihood
    from yagremcmc_amd.statistics.modelHierarchy import BayesianRegressionModelHierarchy
    from yagremcmc_amd.statistics.noise import CentredGaussianNoise
    from yagremcmc_amd.utility.hierarchy import Hierarchy, SharedComponent
    g = golden("f7_mlda2_linear_models.npz")
    n, S, K = int(g["n_states"]), int(g["S"]), g["x0"].shape[0]
    data = Data(np.asarray(g["data"]))
    noise = CentredGaussianNoise(IIDCovarianceMatrix(2, 0.3))
    assert np.allclose(noise.covariance._precision, g["noise_precision"])
    prior = Gaussian(ParameterVector(np.asarray(g["prior_mean"])), IIDCovarianceMatrix(2, 5.0))
    lik = [AdditiveGaussianNoiseLikelihood(data, ForwardModel(ExampleLinearModelSolver(g[f"A_{lv}"], g[f"b_{lv}"])), noise)
           for lv in ("coarse", "fine")]
    b = MLDABuilder()
    b.baseProposalCovariance = IIDCovarianceMatrix(2, 0.5)
    b.subChainLengths = [S]
    b.bayesModel = BayesianRegressionModelHierarchy(Hierarchy(lik), SharedComponent(prior, 2))
    b.nChains, b.seed, b.recordLogPosterior, b.recordSubChain = K, int(g["seed"]), True, True
    mc = b.build_method()
    mc.run(n, ParameterVector(np.zeros(2)), verbose=False)
    # noise precision of the fixture is reciprocal(0.3 from sqrt(0.3)**2); rebuild exactly if it differs in the last bit
    if not np.array_equal(noise.covariance._precision, g["noise_precision"]):
        noise.covariance._precision = np.asarray(g["noise_precision"])
        mc.run(n, ParameterVector(np.zeros(2)), verbose=False)
    assert np.array_equal(mc.chain.trajectory, g["traj"])
    assert np.array_equal(mc.accept_decisions, g["acc"])
    assert np.array_equal(mc.sub_chain_decisions, g["sub_acc"])
    lp = mc.log_posterior
    assert np.abs(lp - g["logpost"]).max() <= 1e-12 * np.abs(g["logpost"]).max()
    with pytest.raises(ValueError, match="Number of sub-chain lengths does not match"):
        b2 = MLDABuilder()
        b2.baseProposalCovariance, b2.subChainLengths = IIDCovarianceMatrix(2, 0.5), [3, 3]
        b2.bayesModel = b.bayesModel
        b2.build_method()


def test_pcn_builder_f8(golden):
    """PCNBuilder (pcn.py:60-88; test_inference_mcmc_singleLevel.py:121-148): device chains == the live
    reference's; the sampler's target is the likelihood; builder errors as in the reference."""
    import yagremcmc_amd.test.testSetup as setup
    from yagremcmc_amd.chain.method.pcn import PCNBuilder
    from yagremcmc_amd.model.forwardModel import ForwardModel
    from yagremcmc_amd.statistics.bayesModel import BayesianRegressionModel
    from yagremcmc_amd.statistics.covariance import DiagonalCovarianceMatrix, IIDCovarianceMatrix
    from yagremcmc_amd.statistics.data import Data
    from yagremcmc_amd.statistics.gaussian import Gaussian
    from yagremcmc_amd.statistics.likelihood import AdditiveGaussianNoiseLikelihood
    from yagremcmc_amd.statistics.noise import CentredGaussianNoise
    g = golden("f8_pcn_lotka_volterra.npz")
    n, K = int(g["n_states"]), g["x0"].shape[0]
    cfg = {'T': float(g["t_final"]), 'alpha': float(g["alpha"]), 'gamma': float(g["gamma"]), 'nData': 5, 'dataDim': 2}
    solver = setup.LotkaVolterraSolver(np.asarray(g["design"]), cfg)
    like = AdditiveGaussianNoiseLikelihood(Data(np.asarray(g["data"])), ForwardModel(solver),
                                           CentredGaussianNoise(IIDCovarianceMatrix(2, 0.05)))
    prior = Gaussian(setup.LotkaVolterraParameter.from_coefficient(np.zeros(2)),
                     DiagonalCovarianceMatrix(np.array([0.02, 0.05])))
    assert np.array_equal(prior.covariance._precision, g["prior_precision"])
    b = PCNBuilder()
    with pytest.raises(ValueError, match="Step size not set in PCN."):
        b.bayesModel = BayesianRegressionModel(like, prior)
        b.build_method()
    b.stepSize = float(g["step_size"])
    b.nChains, b.seed, b.recordLogPosterior = K, int(g["seed"]), True
    mc = b.build_method()
    assert mc.target is like
    mc.run(n, setup.LotkaVolterraParameter.from_coefficient(np.array([-0.6, -0.3])), verbose=False)
    assert np.array_equal(mc.chain.trajectory, g["traj"])
    assert np.array_equal(mc.accept_decisions, g["acc"])
    assert np.abs(mc.log_posterior - g["loglik"]).max() <= 1e-12 * np.abs(g["loglik"]).max()
    # and bit-exact against the oracle including diagnostics
    p = o.posterior_lv(g["design"], cfg['T'], cfg['alpha'], cfg['gamma'], 50, g["data"], g["noise_precision"],
                       g["prior_mean"], g["prior_precision"])
    p["likelihood_only"] = 1
    e = o.Ensemble(p, g["prior_sigma"], g["x0"], int(g["seed"]), pcn_step_size=float(g["step_size"]))
    out = e.advance(n - 1)
    assert np.array_equal(mc.log_posterior[1:], out["logpost"])
    # likelihood.evaluate_log on the device == the recorded value
    got = like.evaluate_log(setup.LotkaVolterraParameter.from_coefficient(mc.chain.trajectory[100, 3].copy()))
    assert got == mc.log_posterior[100, 3]
    b2 = PCNBuilder()
    b2.stepSize, b2.explicitTarget = 0.01, object()
    with pytest.raises(RuntimeError):
        b2.build_method()
    off = Gaussian(setup.LotkaVolterraParameter.from_coefficient(np.array([0.1, 0.0])), IIDCovarianceMatrix(2, 0.02))
    b3 = PCNBuilder()
    b3.stepSize, b3.bayesModel = 0.01, BayesianRegressionModel(like, off)
    with pytest.raises(ValueError, match="centred prior"):
        b3.build_method()


def test_integrated_autocorrelation_on_device_f9(golden):
    """postprocessing/autocorrelation.py:92-140 on the device == the reference's values for the F2 / F1 chains"""
    from yagremcmc_amd import _ffi
    from yagremcmc_amd.postprocessing import autocorrelation as ac
    g2, g9, g1 = golden("f2_mrw_2d.npz"), golden("f9_iat.npz"), golden("f1_mrw_1d.npz")
    traj = g2["traj"]
    for burnin in (0, 200):
        assert np.array_equal(ac.integrated_autocorrelation(traj[burnin:], 'mean'), g9[f"iat_mean_b{burnin}"])
        assert np.array_equal(ac.integrated_autocorrelation(traj[burnin:], 'max'), g9[f"iat_max_b{burnin}"])
    # one chain, (n, d) host array: the reference's call signature and return type
    one = ac.integrated_autocorrelation(traj[200:, 5, :], 'max')
    assert isinstance(one, int) and one == int(g9["iat_max_b200"][5])
    s1 = g1["traj"][:, 0, :]
    assert ac.integrated_autocorrelation(s1[200:], 'mean') == int(g9["f1_iat_mean_b200"])
    assert ac.integrated_autocorrelation(s1[200:], 'max') == int(g9["f1_iat_max_b200"])
    acf = ac.estimate_autocorrelation_function_1d(traj[200:, 0, 0])
    assert acf.shape == g9["acf_chain0_dim0_b200"].shape and acf[0] == 1.0
    assert np.abs(acf - g9["acf_chain0_dim0_b200"]).max() < 1e-12
    with pytest.raises(ValueError, match="Invalid IAT"):
        ac.integrated_autocorrelation(traj[:, 0, :], 'median')
    # from the record in HBM, through the sampler (the F2 problem, same seed => same chains)
    from yagremcmc_amd.chain.method.mrw import MRWBuilder
    from yagremcmc_amd.parameter.vector import ParameterVector
    from yagremcmc_amd.statistics.covariance import IIDCovarianceMatrix
    from yagremcmc_amd.test.testSetup import GaussianTargetDensity2d
    RefT = type("GaussianTargetDensity2d", (), {})
    t = RefT()
    from types import SimpleNamespace
    w, v = np.linalg.eigh(g2["cov"])
    t.dist_ = SimpleNamespace(mean=g2["mean"], cov=g2["cov"], cov_object=SimpleNamespace(
        _LP=g2["whiten"], _log_pdet=float(g2["log_norm"]) - 2 * np.log(2 * np.pi), rank=2))
    b = MRWBuilder()
    b.explicitTarget, b.proposalCovariance = t, IIDCovarianceMatrix(2, 1.0)
    b.nChains, b.seed = 64, int(g2["seed"])
    mc = b.build_method()
    mc.run(2000, ParameterVector(np.array([-8.0, -7.0])), verbose=False)
    assert np.array_equal(mc.chain.trajectory, traj)
    for burnin in (0, 200):
        assert np.array_equal(mc.integrated_autocorrelation('mean', burnin=burnin), g9[f"iat_mean_b{burnin}"])
        assert np.array_equal(mc.integrated_autocorrelation('max', burnin=burnin), g9[f"iat_max_b{burnin}"])
    # long series (beyond the LDS-resident size) and many series at once
    rng = np.random.default_rng(3)
    x = np.zeros((30000, 3, 1))
    e = rng.normal(size=x.shape)
    for i in range(1, x.shape[0]):
        x[i] = 0.9 * x[i - 1] + e[i]
    iat = ac.integrated_autocorrelation(x, 'mean')
    assert iat.shape == (3,) and (np.abs(iat - 19) <= 4).all()        # AR(1), phi = 0.9: tau = 19
    del GaussianTargetDensity2d, _ffi


def test_lotka_volterra_solver_like_the_reference_tests():
    """test_solver_setup.py:27-78 and test_solver_invoke.py:62-116 on the device-backed solver: attribute
    names, shapes, status, and agreement with an independent scipy solve at the reference's 1e-3."""
    from scipy.integrate import solve_ivp
    import yagremcmc_amd.test.testSetup as ts
    from yagremcmc_amd.model.forwardModel import EvaluationStatus
    from yagremcmc_amd.statistics.data import Data
    config = {'T': 10, 'alpha': 0.1, 'gamma': 0.1, 'nData': 10, 'dataDim': 2, 'solver': 'LSODA', 'rtol': 1e-4}
    design = np.random.default_rng(0).random((10, 2))
    coefficients = np.array([0.2, 0.3])
    parameter = ts.LotkaVolterraParameter.from_coefficient(coefficients)
    solver = ts.LotkaVolterraSolver(design, config)
    assert solver.x_.shape == design.shape and solver.tBoundary_ == (0., config['T'])
    assert solver.fixedParam_ == [config['alpha'], config['gamma']]
    assert solver.dataShape_ == (config['nData'], config['dataDim'])
    assert solver.param_ == [None, None] and solver.evaluation_ is None and solver.status_ == EvaluationStatus.NONE
    solver.interpolate(parameter)
    np.testing.assert_allclose(solver.param_, np.exp([0.2, 0.3]), rtol=1e-5, atol=1e-8)
    solver.invoke()
    assert solver.status_ == EvaluationStatus.SUCCESS and solver.status == EvaluationStatus.SUCCESS
    assert solver.evaluation_.shape == (config['nData'], config['dataDim']) and (solver.evaluation_ >= 0).all()

    def flow(t, y):
        return [0.1 * y[0] - np.exp(0.2) * y[0] * y[1], np.exp(0.3) * y[0] * y[1] - 0.1 * y[1]]

    for i in range(config['nData']):
        ref = solve_ivp(flow, (0, 10), design[i], method='LSODA', rtol=1e-10, atol=1e-12).y[:, -1]
        np.testing.assert_allclose(solver.evaluation_[i], ref, rtol=1e-3, atol=1e-6)
    t, y = solver.full_solution(parameter, np.array([1.0, 0.5]))
    assert len(t) == 51 and y.shape == (2, 51) and np.array_equal(y[:, 0], [1.0, 0.5])
    dense = solve_ivp(flow, (0, 10), [1.0, 0.5], method='LSODA', rtol=1e-10, atol=1e-12, t_eval=t).y
    np.testing.assert_allclose(y, dense, rtol=1e-3, atol=1e-6)
    data = ts.generate_synthetic_data(parameter, ts.LotkaVolterraSolver(design, config), 0.1)
    assert isinstance(data, Data) and data.size == config['nData'] and data.array[0, :].size == config['dataDim']


def test_two_level_lotka_volterra_model_hierarchy_vs_oracle():
    """Delayed acceptance over a hierarchy of Lotka-Volterra models (the shape of the reference's
    example_inference_lotkaVolterra_twoLevel.py): the coarse level integrates with 8 RK4 steps, the
    fine one with 50; every level keeps its own solver settings through the lowering."""
    import yagremcmc_amd.test.testSetup as setup
    from yagremcmc_amd.chain.method.mlda import MLDABuilder
    from yagremcmc_amd.model.forwardModel import ForwardModel
    from yagremcmc_amd.parameter.vector import ParameterVector
    from yagremcmc_amd.statistics.covariance import IIDCovarianceMatrix
    from yagremcmc_amd.statistics.data import Data
    from yagremcmc_amd.statistics.gaussian import Gaussian
    from yagremcmc_amd.statistics.likelihood import AdditiveGaussianNoiseLikelihood
    from yagremcmc_amd.statistics.modelHierarchy import BayesianRegressionModelHierarchy
    from yagremcmc_amd.statistics.noise import CentredGaussianNoise
    from yagremcmc_amd.utility.hierarchy import Hierarchy, SharedComponent
    rng = np.random.default_rng(3)
    design = rng.uniform(0.5, 1.5, size=(6, 2))
    cfg = {'T': 5., 'alpha': 0.8, 'gamma': 0.4, 'nData': 6, 'dataDim': 2}
    fine_solver = setup.LotkaVolterraSolver(design, dict(cfg, rk4Steps=50))
    coarse_solver = setup.LotkaVolterraSolver(design, dict(cfg, rk4Steps=8))
    truth = setup.LotkaVolterraParameter.from_coefficient(np.log([0.4, 0.6]))
    fine_solver.interpolate(truth)
    fine_solver.invoke()
    data = fine_solver.evaluation + 0.2 * rng.normal(size=(6, 2))
    noise = CentredGaussianNoise(IIDCovarianceMatrix(2, 0.04))
    prior = Gaussian(ParameterVector(np.zeros(2)), IIDCovarianceMatrix(2, 1.4))
    liks = [AdditiveGaussianNoiseLikelihood(Data(data), ForwardModel(s), noise) for s in (coarse_solver, fine_solver)]
    b = MLDABuilder()
    b.bayesModel = BayesianRegressionModelHierarchy(Hierarchy(liks), SharedComponent(prior, 2))
    b.baseProposalCovariance = IIDCovarianceMatrix(2, 0.02)
    b.subChainLengths = [3]
    b.nChains, b.seed, b.recordSubChain = 200, 9, True
    mc = b.build_method()
    mc.run(81, ParameterVector(np.array([-0.6, -0.3])), verbose=False)
    common = dict(data=data, noise_precision=np.reciprocal(np.full(2, 0.04)), prior_mean=np.zeros(2),
                  prior_precision=np.reciprocal(np.full(2, 1.4)))
    fine = o.posterior_lv(design, 5.0, 0.8, 0.4, 50, **common)
    coarse = o.posterior_lv(design, 5.0, 0.8, 0.4, 8, **common)
    sigma = np.sqrt(np.reciprocal(np.reciprocal(np.full(2, 0.02))))
    e = o.Ensemble(fine, sigma, np.tile([-0.6, -0.3], (200, 1)), 9, surrogate=coarse, sub_chain_length=3)
    out = e.advance(80, want_sub=True)
    assert np.array_equal(mc.chain.trajectory[1:], out["traj"]) and np.array_equal(mc.accept_decisions, out["acc"])
    assert np.array_equal(mc.sub_chain_decisions, out["sub_acc"])
    assert 0.05 < out["acc"].mean() < 0.999 and not np.array_equal(out["acc"], out["sub_acc"][:, -1, :])


@pytest.mark.parametrize("name,prop_var", [("f14_mlda2_linear_d128.npz", 1.1e-3), ("f14b_mlda2_linear_d192.npz", 7e-4),
                                           ("f14c_mlda2_linear_d300.npz", 4e-4)])
def test_two_level_over_a_model_hierarchy_d128_f14(golden, name, prop_var):
    """MLDABuilder.bayesModel = BayesianRegressionModelHierarchy of two d = 128 / d = 192 linear models
    (mlda.py:315-335): through the builders the chains are the live reference's (fixtures F14 / F14b), on the
    matrix-core kernel."""
    import hashlib
    from yagremcmc_amd.chain.diagnostics import AcceptanceRateDiagnostics
    from yagremcmc_amd.chain.method.mlda import MLDABuilder
    from yagremcmc_amd.exampleSetup import ExampleLinearModelSolver
    from yagremcmc_amd.model.forwardModel import ForwardModel
    from yagremcmc_amd.parameter.vector import ParameterVector
    from yagremcmc_amd.statistics.covariance import IIDCovarianceMatrix
    from yagremcmc_amd.statistics.data import Data
    from yagremcmc_amd.statistics.gaussian import Gaussian
    from yagremcmc_amd.statistics.likelihood import AdditiveGaussianNoiseLikelihood
    from yagremcmc_amd.statistics.modelHierarchy import BayesianRegressionModelHierarchy
    from yagremcmc_amd.statistics.noise import CentredGaussianNoise
    from yagremcmc_amd.utility.hierarchy import Hierarchy, SharedComponent
    g = golden(name)
    d, n, S, K = g["A_fine"].shape[1], int(g["n_states"]), int(g["S"]), g["x0"].shape[0]
    prior = Gaussian(ParameterVector(np.array(g["prior_mean"])), IIDCovarianceMatrix(d, 5.0))
    noise = CentredGaussianNoise(IIDCovarianceMatrix(d, 0.3))
    liks = [AdditiveGaussianNoiseLikelihood(Data(g["data"]), ForwardModel(ExampleLinearModelSolver(g[f"A_{lv}"], g[f"b_{lv}"])), noise)
            for lv in ("coarse", "fine")]
    b = MLDABuilder()
    b.bayesModel = BayesianRegressionModelHierarchy(Hierarchy(liks), SharedComponent(prior, 2))
    b.baseProposalCovariance = IIDCovarianceMatrix(d, prop_var)
    b.subChainLengths = [S]
    b.surrogateDiagnostics = [AcceptanceRateDiagnostics()]
    b.nChains, b.seed, b.recordSubChain = K, int(g["seed"]), True
    mc = b.build_method()
    mc.run(n, ParameterVector(np.zeros(d)), verbose=False)
    assert np.array_equal(mc.accept_decisions, g["acc"]) and np.array_equal(mc.sub_chain_decisions, g["sub_acc"])
    traj = mc.chain.ensemble_trajectory
    assert hashlib.sha256(np.ascontiguousarray(traj).tobytes()).hexdigest() == str(g["traj_sha256"])
    assert abs(mc.diagnostics.global_acceptance_rate() - g["acc"].mean()) < 1e-15


@pytest.mark.parametrize("name,prior_var,noise_var,prop_var", [("f18_mlda_nested2_linear_d128.npz", 5.0, 0.3, 1.1e-3),
                                                               ("f18b_mlda_nested4_linear_d24.npz", 5.0, 0.3, 6e-3),
                                                               ("f18c_mlda_nested8_linear_d24.npz", 5.0, 0.3, 6e-3),
                                                               ("f18d_mlda_nested2_linear_d192.npz", 5.0, 0.3, 7e-4),
                                                               ("f18e_mlda_nested2_linear_d300.npz", 5.0, 0.3, 4e-4)])
def test_nested_levels_over_model_hierarchies_f18(golden, name, prior_var, noise_var, prop_var):
    """MLDABuilder.bayesModel = a hierarchy of three (d = 128) / five / nine (d = 24) linear models, i.e. an even number
    of surrogates: through the builders the chains are the live reference's (fixtures F18), on the matrix cores."""
    import hashlib
    from yagremcmc_amd.chain.diagnostics import AcceptanceRateDiagnostics
    from yagremcmc_amd.chain.method.mlda import MLDABuilder
    from yagremcmc_amd.exampleSetup import ExampleLinearModelSolver
    from yagremcmc_amd.model.forwardModel import ForwardModel
    from yagremcmc_amd.parameter.vector import ParameterVector
    from yagremcmc_amd.statistics.covariance import IIDCovarianceMatrix
    from yagremcmc_amd.statistics.data import Data
    from yagremcmc_amd.statistics.gaussian import Gaussian
    from yagremcmc_amd.statistics.likelihood import AdditiveGaussianNoiseLikelihood
    from yagremcmc_amd.statistics.modelHierarchy import BayesianRegressionModelHierarchy
    from yagremcmc_amd.statistics.noise import CentredGaussianNoise
    from yagremcmc_amd.utility.hierarchy import Hierarchy, SharedComponent
    g = golden(name)
    n_models, d = g["A"].shape[0], g["A"].shape[2]
    n, K = int(g["n_states"]), g["x0"].shape[0]
    prior = Gaussian(ParameterVector(np.array(g["prior_mean"])), IIDCovarianceMatrix(d, prior_var))
    noise = CentredGaussianNoise(IIDCovarianceMatrix(d, noise_var))
    liks = [AdditiveGaussianNoiseLikelihood(Data(g["data"]), ForwardModel(ExampleLinearModelSolver(A, b_)), noise)
            for A, b_ in zip(g["A"], g["b"])]
    b = MLDABuilder()
    b.bayesModel = BayesianRegressionModelHierarchy(Hierarchy(liks), SharedComponent(prior, n_models))
    b.baseProposalCovariance = IIDCovarianceMatrix(d, prop_var)
    b.subChainLengths = [int(v) for v in g["sub_lengths"]]
    b.surrogateDiagnostics = [AcceptanceRateDiagnostics() for _ in range(n_models - 1)]
    b.nChains, b.seed = K, int(g["seed"])
    mc = b.build_method()
    mc.run(n, ParameterVector(np.zeros(d)), verbose=False)
    assert np.array_equal(mc.accept_decisions, g["acc"])
    traj = mc.chain.ensemble_trajectory
    assert hashlib.sha256(np.ascontiguousarray(traj).tobytes()).hexdigest() == str(g["traj_sha256"])
    assert abs(mc.diagnostics.global_acceptance_rate() - g["acc"].mean()) < 1e-15
    for e in range(g["level_acc"].shape[0]):   # the executed lower levels: surrogates 0, 2, ...
        made = (n - 1) * int(np.prod([int(g["sub_lengths"][2 * j + 1]) for j in range(e, g["level_acc"].shape[0])]))
        assert abs(mc.surrogate(2 * e).diagnostics.global_acceptance_rate() - g["level_acc"][e].sum() / (made * K)) < 1e-15


def test_dense_noise_and_prior_through_the_builders_f15(golden):
    """MRWBuilder.bayesModel with CentredGaussianNoise(DenseCovarianceMatrix) and a Gaussian prior over a
    DenseCovarianceMatrix: the chains of the live reference (fixture F15a)."""
    from yagremcmc_amd.chain.method.mrw import MRWBuilder
    from yagremcmc_amd.exampleSetup import ExampleLinearModelSolver
    from yagremcmc_amd.model.forwardModel import ForwardModel
    from yagremcmc_amd.parameter.vector import ParameterVector
    from yagremcmc_amd.statistics.bayesModel import BayesianRegressionModel
    from yagremcmc_amd.statistics.covariance import DenseCovarianceMatrix, DiagonalCovarianceMatrix
    from yagremcmc_amd.statistics.data import Data
    from yagremcmc_amd.statistics.gaussian import Gaussian
    from yagremcmc_amd.statistics.likelihood import AdditiveGaussianNoiseLikelihood
    from yagremcmc_amd.statistics.noise import CentredGaussianNoise
    g = golden("f15_dense_noise_prior.npz")
    like = AdditiveGaussianNoiseLikelihood(Data(g["data"]), ForwardModel(ExampleLinearModelSolver(g["A"], g["b"])),
                                           CentredGaussianNoise(DenseCovarianceMatrix(g["noise_cov"])))
    prior = Gaussian(ParameterVector(np.array(g["prior_mean"])), DenseCovarianceMatrix(g["prior_cov"]))
    b = MRWBuilder()
    b.bayesModel = BayesianRegressionModel(like, prior)
    b.proposalCovariance = DiagonalCovarianceMatrix(np.array([0.12, 0.2, 0.15]))
    b.nChains, b.seed, b.recordLogPosterior = g["x0"].shape[0], int(g["seed"]), True
    mc = b.build_method()
    mc.run(int(g["n_states"]), ParameterVector(np.zeros(3)), verbose=False)
    assert np.array_equal(mc.accept_decisions, g["acc"]) and np.array_equal(mc.chain.ensemble_trajectory, g["traj"])
    assert np.abs(mc.log_posterior - g["logpost"]).max() <= 1e-12 * np.abs(g["logpost"]).max()
    # evaluate_log of the posterior object itself goes through the same lowering
    k, n = 3, 200
    assert mc.target.evaluate_log(ParameterVector(mc.chain.ensemble_trajectory[n, k].copy())) == mc.log_posterior[n, k]


def test_dense_noise_beyond_the_lane_sizes_through_the_builders_f19(golden):
    """MRWBuilder / MLDABuilder over d = 24 linear models whose noise is CentredGaussianNoise(DenseCovarianceMatrix)
    (24 x 24): the chains of the live reference (fixture F19), on the matrix-core kernels; the posterior object's
    own evaluate_log and the solver's forward value go through the same lowering."""
    from yagremcmc_amd.chain.diagnostics import AcceptanceRateDiagnostics
    from yagremcmc_amd.chain.method.mlda import MLDABuilder
    from yagremcmc_amd.chain.method.mrw import MRWBuilder
    from yagremcmc_amd.exampleSetup import ExampleLinearModelSolver
    from yagremcmc_amd.model.forwardModel import ForwardModel
    from yagremcmc_amd.parameter.vector import ParameterVector
    from yagremcmc_amd.statistics.bayesModel import BayesianRegressionModel
    from yagremcmc_amd.statistics.covariance import DenseCovarianceMatrix, IIDCovarianceMatrix
    from yagremcmc_amd.statistics.data import Data
    from yagremcmc_amd.statistics.gaussian import Gaussian
    from yagremcmc_amd.statistics.likelihood import AdditiveGaussianNoiseLikelihood
    from yagremcmc_amd.statistics.modelHierarchy import BayesianRegressionModelHierarchy
    from yagremcmc_amd.statistics.noise import CentredGaussianNoise
    from yagremcmc_amd.utility.hierarchy import Hierarchy, SharedComponent
    g = golden("f19_dense_noise_d24.npz")
    d = 24
    noise = CentredGaussianNoise(DenseCovarianceMatrix(g["noise_cov"]))
    prior = Gaussian(ParameterVector(np.array(g["prior_mean"])), IIDCovarianceMatrix(d, 5.0))
    lik_f = AdditiveGaussianNoiseLikelihood(Data(g["data"]), ForwardModel(ExampleLinearModelSolver(g["A"], g["b"])), noise)
    lik_c = AdditiveGaussianNoiseLikelihood(Data(g["data"]), ForwardModel(ExampleLinearModelSolver(g["A_coarse"], g["b_coarse"])), noise)
    b = MRWBuilder()
    b.bayesModel = BayesianRegressionModel(lik_f, prior)
    b.proposalCovariance = IIDCovarianceMatrix(d, 4e-3)
    b.nChains, b.seed, b.recordLogPosterior = g["x0"].shape[0], int(g["seed"]), True
    mc = b.build_method()
    mc.run(int(g["n_states"]), ParameterVector(np.zeros(d)), verbose=False)
    assert np.array_equal(mc.accept_decisions, g["acc"]) and np.array_equal(mc.chain.ensemble_trajectory, g["traj"])
    assert np.abs(mc.log_posterior - g["logpost"]).max() <= 1e-12 * np.abs(g["logpost"]).max()
    k, n = 3, 200
    assert mc.target.evaluate_log(ParameterVector(mc.chain.ensemble_trajectory[n, k].copy())) == mc.log_posterior[n, k]
    b = MLDABuilder()
    b.bayesModel = BayesianRegressionModelHierarchy(Hierarchy([lik_c, lik_f]), SharedComponent(prior, 2))
    b.baseProposalCovariance = IIDCovarianceMatrix(d, 4e-3)
    b.subChainLengths = [int(g["ml_S"])]
    b.surrogateDiagnostics = [AcceptanceRateDiagnostics()]
    b.nChains, b.seed, b.recordSubChain = g["ml_x0"].shape[0], int(g["ml_seed"]), True
    mc = b.build_method()
    mc.run(int(g["ml_n_states"]), ParameterVector(np.zeros(d)), verbose=False)
    assert np.array_equal(mc.accept_decisions, g["ml_acc"]) and np.array_equal(mc.sub_chain_decisions, g["ml_sub_acc"])
    assert np.array_equal(mc.chain.ensemble_trajectory, g["ml_traj"])


@pytest.mark.parametrize("name,prop_var", [("f22_linear_d192_m160.npz", 8e-4), ("f22b_linear_d320_m48.npz", 5e-4)])
def test_linear_model_d192_through_the_builder_f22(golden, name, prop_var):
    """MRWBuilder over the reference's shape-agnostic linear model at d = 192, m = 160 and at d = 320, m = 48
    (exampleSetup.py:42-52, likelihood.py:74-84; fixtures F22 / F22b = the live reference's chains): the forward matrix
    streams from L2, one wave per SIMD holds 192 components per chain; 320 components take the wide path."""
    import hashlib
    from yagremcmc_amd.chain.method.mrw import MRWBuilder
    from yagremcmc_amd.exampleSetup import ExampleLinearModelSolver
    from yagremcmc_amd.model.forwardModel import ForwardModel
    from yagremcmc_amd.parameter.vector import ParameterVector
    from yagremcmc_amd.statistics.bayesModel import BayesianRegressionModel
    from yagremcmc_amd.statistics.covariance import IIDCovarianceMatrix
    from yagremcmc_amd.statistics.data import Data
    from yagremcmc_amd.statistics.gaussian import Gaussian
    from yagremcmc_amd.statistics.likelihood import AdditiveGaussianNoiseLikelihood
    from yagremcmc_amd.statistics.noise import CentredGaussianNoise
    g = golden(name)
    m, d = g["A"].shape
    like = AdditiveGaussianNoiseLikelihood(Data(g["data"]), ForwardModel(ExampleLinearModelSolver(g["A"], g["b"])),
                                           CentredGaussianNoise(IIDCovarianceMatrix(m, 0.3)))
    b = MRWBuilder()
    b.bayesModel = BayesianRegressionModel(like, Gaussian(ParameterVector(np.zeros(d)), IIDCovarianceMatrix(d, 5.0)))
    b.proposalCovariance = IIDCovarianceMatrix(d, prop_var)
    b.nChains, b.seed, b.recordLogPosterior = g["x0"].shape[0], int(g["seed"]), True
    mc = b.build_method()
    mc.run(int(g["n_states"]), ParameterVector(np.zeros(d)), verbose=False)
    assert np.array_equal(mc.accept_decisions, g["acc"])
    traj = mc.chain.ensemble_trajectory
    assert np.array_equal(traj[:, :2, :], g["traj_head"])
    assert hashlib.sha256(np.ascontiguousarray(traj).tobytes()).hexdigest() == str(g["traj_sha256"])
    assert np.abs(mc.log_posterior - g["logpost"]).max() <= 1e-12 * np.abs(g["logpost"]).max()
    assert mc.integrated_autocorrelation("mean", 20).shape == (g["x0"].shape[0],)


def test_iat_of_a_stuck_chain_is_reported_not_invented():
    """A chain that never moves (or was flagged non-finite and froze) has a constant series: c0 = 0 and the
    autocorrelation is 0 / 0.  The device reports IAT_UNDEFINED for it; the Python layer raises unless asked."""
    from yagremcmc_amd import _ffi
    from yagremcmc_amd.postprocessing import autocorrelation as ac
    rng = np.random.default_rng(4)
    x = np.empty((400, 3, 2))
    x[:, 0] = rng.standard_normal((400, 2)).cumsum(axis=0) * 0.1
    x[:, 1] = [0.25, -1.5]                      # stuck
    x[:, 2, 0] = rng.standard_normal(400)       # one moving, one constant component
    x[:, 2, 1] = 3.0
    with pytest.raises(ValueError, match="constant"):
        ac.integrated_autocorrelation(x, 'mean')
    got = ac.integrated_autocorrelation(x, 'mean', allowUndefined=True)
    assert got[1] == _ffi.IAT_UNDEFINED and got[0] > 0 and got[2] > 0
    got = ac.integrated_autocorrelation(x, 'max', allowUndefined=True)
    assert got[0] > 0 and got[1] == _ffi.IAT_UNDEFINED and got[2] == _ffi.IAT_UNDEFINED
    iat, _ = _ffi.default_context(0).iat_series(np.stack([x[:, 0, 0], np.full(400, 7.0)]), allow_undefined=True)
    assert iat[0] > 0 and iat[1] == _ffi.IAT_UNDEFINED


def _reference_iat(series, c=5.0):
    """postprocessing/autocorrelation.py:5-29,32-89 restated with NumPy (direct lag sums)."""
    x = np.asarray(series, dtype=float)
    x = x - x.mean()
    n = x.size
    c0 = float(np.dot(x, x))
    cum = 1.0
    for lag in range(1, n):
        cum += float(np.dot(x[:-lag], x[lag:])) / c0
        iat = 2.0 * cum - 1.0
        if not lag < c * iat:
            return int(np.rint(iat))
    return 1


@pytest.mark.parametrize("kernel", ["resident", "wave", "block"])
@pytest.mark.parametrize("N", [2, 9, 77, 513, 1500, 4100, 9800])
def test_iat_kernels_on_ragged_series(monkeypatch, kernel, N):
    """The persistent LDS-resident kernel (eight waves share a series' tiles, the next series requested into registers), the
    wave-per-series kernel (register-blocked lags, LDS tiles of 512) and the workgroup-per-series one
    against the NumPy restatement of autocorrelation.py: lengths around the tile and lag-block sizes, slowly
    and quickly decorrelating series (windows that close in the first block and after several), and more series than
    the card has CUs (the persistent kernel's second lap)."""
    from yagremcmc_amd import _ffi
    monkeypatch.setenv("YMC_IAT_KERNEL", kernel)
    rng = np.random.default_rng(N)
    S = 70 if N != 1500 else 600
    phi = rng.uniform(0.0, 0.97, size=S)
    x = np.zeros((S, N))
    e = rng.standard_normal((S, N))
    for t in range(1, N):
        x[:, t] = phi * x[:, t - 1] + e[:, t]
    x[:, 0] = e[:, 0]
    x += rng.normal(size=(S, 1)) * 50.0                   # a large mean: centring matters
    got, _ = _ffi.default_context(0).iat_series(x)
    want = np.array([_reference_iat(x[s]) for s in range(S)])
    assert np.array_equal(got, want), (got - want)[got != want]


@pytest.mark.parametrize("K,d,N", [(70, 1, 900), (300, 3, 700), (1000, 2, 2100)])
def test_iat_of_many_chains_from_states(K, d, N):
    """'max' and 'mean' on many chains through the transposition and the persistent LDS-resident kernel == the NumPy
    restatement of autocorrelation.py per series (more series than the card has CUs: several laps per workgroup)."""
    from yagremcmc_amd import _ffi
    rng = np.random.default_rng(K + d)
    phi = rng.uniform(0.0, 0.95, size=(K, d))
    x = np.zeros((N, K, d))
    e = rng.standard_normal((N, K, d))
    x[0] = e[0]
    for t in range(1, N):
        x[t] = phi * x[t - 1] + e[t]
    x += rng.normal(size=(1, K, d)) * 20.0
    ctx = _ffi.default_context(0)
    want = np.array([max(_reference_iat(x[:, k, j]) for j in range(d)) for k in range(K)])
    assert np.array_equal(ctx.iat_states(x, 1), want)
    want_mean = np.array([_reference_iat(np.mean(x[:, k, :], axis=1)) for k in range(K)])
    assert np.array_equal(ctx.iat_states(x, 0), want_mean)


@pytest.mark.parametrize("d", [8, 24, 128, 136, 200, 256])
def test_integrated_autocorrelation_from_the_tiled_record_and_numpy_mean_order(d):
    """'mean' IAT at d >= 8: the per-state mean is formed in NumPy's pairwise order (bit-identical to
    np.mean(seq, axis=1), whose blocks split beyond 128 terms); d > 8 reads the tile kernel's record layout in HBM."""
    from yagremcmc_amd import _ffi, lowering
    rng = np.random.default_rng(d)
    m = min(d, 16)
    A = rng.normal(size=(m, d)) / np.sqrt(d)
    low = lowering.Lowered(_ffi.POSTERIOR_LINEAR, d)
    low.desc.n_data, low.desc.data_dim = 2, m
    for k, v in dict(data=rng.normal(size=(2, m)), noise_precision=np.full(m, 2.0), prior_mean=np.zeros(d),
                     prior_precision=np.ones(d), fwd_matrix=A, fwd_shift=np.zeros(m)).items():
        low.set_array(k, v)
    ctx = _ffi.default_context(0)
    prob = lowering.make_problem(ctx, _ffi.METHOD_MRW, low, np.full(d, 0.35))
    K, N = 40, 600
    run = _ffi.Run(prob, 5, K, rng.normal(size=(K, d)), N, _ffi.RECORD_TRAJECTORY | _ffi.RECORD_ACCEPT)
    run.advance(N)
    run.sync()
    traj = run.trajectory()
    got = run.iat(0, None, 0, 5.0)
    means = np.mean(traj, axis=2)                       # NumPy's own order, per (row, chain)
    want = np.array([_reference_iat(means[:, k]) for k in range(K)])
    assert np.array_equal(got, want)
    # the device's means are NumPy's, bit for bit (through the host-array entry point, same kernel)
    assert np.array_equal(ctx.iat_states(traj, 0, 5.0), want)
    if d <= 24:
        got_max = run.iat(0, None, 1, 5.0)
        want_max = np.array([max(_reference_iat(traj[:, k, j]) for j in range(d)) for k in range(K)])
        assert np.array_equal(got_max, want_max)
