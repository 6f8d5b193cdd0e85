#!/usr/bin/env python3
"""Generate the golden fixtures of tests/golden/ from the LIVE reference.

Runs only in the build container (needs /root/reference on PYTHONPATH):

    PYTHONPATH=/root/reference MPLBACKEND=Agg python tests/golden/make_golden.py

For every fixture the UNMODIFIED reference classes are built through the
reference's own builders and run one chain at a time; the only intervention is
that the two module-level names the reference draws random numbers through,

    yagremcmc.statistics.gaussian.standard_normal      (gaussian.py:2,63)
    yagremcmc.chain.metropolisHastings.uniform          (metropolisHastings.py:2,68)

are rebound to closures replaying the ensemble's counter-based stream
(csrc/ymc_numerics.h, evaluated here through oracle.rng_fill) by ADDRESS:
the normals of transition t, and -- only if the reference asks for it -- the
uniform of transition t.  What is stored is data: the inputs of each problem
and the reference's outputs (trajectories, accept decisions, log-densities,
diagnostics).  Nothing of the reference's source travels.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle  # noqa: E402  (draw stream + RK4 plug-in only)

import yagremcmc.chain.metropolisHastings as ref_mh  # noqa: E402
import yagremcmc.statistics.gaussian as ref_gaussian  # noqa: E402
from yagremcmc.chain.diagnostics import AcceptanceRateDiagnostics, FullDiagnostics  # noqa: E402
from yagremcmc.chain.method.mlda import MLDABuilder  # noqa: E402
from yagremcmc.chain.method.mrw import MRWBuilder  # noqa: E402
from yagremcmc.model.evaluation import EvaluationStatus  # noqa: E402
from yagremcmc.model.forwardModel import ForwardModel  # noqa: E402
from yagremcmc.model.interface import SolverInterface  # noqa: E402
from yagremcmc.parameter.scalar import ScalarParameter  # noqa: E402
from yagremcmc.parameter.vector import ParameterVector  # noqa: E402
from yagremcmc.statistics.bayesModel import BayesianRegressionModel  # noqa: E402
from yagremcmc.statistics.covariance import (DenseCovarianceMatrix, DiagonalCovarianceMatrix,  # noqa: E402
                                             IIDCovarianceMatrix)
from yagremcmc.statistics.data import Data  # noqa: E402
from yagremcmc.statistics.gaussian import Gaussian  # noqa: E402
from yagremcmc.statistics.likelihood import AdditiveGaussianNoiseLikelihood  # noqa: E402
from yagremcmc.statistics.noise import CentredGaussianNoise  # noqa: E402
from yagremcmc.test.testSetup import (GaussianTargetDensity1d, GaussianTargetDensity2d,  # noqa: E402
                                      LotkaVolterraParameter, LotkaVolterraSolver)

sys.path.insert(0, "/root/reference")
from exampleSetup import ExampleLinearModelSolver  # noqa: E402


class Replay:
    """Serves the ensemble's draws of one chain to the reference by address."""

    def __init__(self, seed, chain, dim, n_transitions, sub_chain_length=0):
        self.S = sub_chain_length
        self.xi, self.u = oracle.rng_fill(seed, chain, 1, 0, n_transitions, dim)
        self.xi, self.u = self.xi[:, 0, :], self.u[:, 0]
        self.normal_calls = 0
        self.t_last = None
        self.fine = False
        self.used_u = []

    def _t_of_call(self, c):
        if self.S == 0:
            return c
        return (c // self.S) * (self.S + 1) + (c % self.S)

    def standard_normal(self, size=None):
        t = self._t_of_call(self.normal_calls)
        self.normal_calls += 1
        self.t_last = t
        out = self.xi[t].copy()
        assert size is None or int(size) == out.size
        return out

    def uniform(self, low=0.0, high=1.0, size=None):
        if self.S == 0:
            t = self.t_last
        elif self.fine:
            # fine accept/reject of the fine step whose sub-chain just ended
            t = (self.t_last // (self.S + 1)) * (self.S + 1) + self.S
        else:
            t = self.t_last
        self.used_u.append(t)
        return np.array([self.u[t]])

    def install(self):
        ref_gaussian.standard_normal = self.standard_normal
        ref_mh.uniform = self.uniform


def mark_fine_level(mc, replay):
    """Tell the replay when the FINE chain's _accept_reject is drawing."""
    inner = mc._accept_reject

    def wrapped(proposal, state):
        replay.fine = True
        try:
            return inner(proposal, state)
        finally:
            replay.fine = False

    mc._accept_reject = wrapped


class SequentialReplay:
    """Draws by address for the nested scheme: every accept/reject DECISION of any level takes the
    next transition index in execution order (csrc/ymc_numerics.h, draw addressing).  The index
    advances when a level's _accept_reject returns (watch()), whether or not it drew a uniform."""

    def __init__(self, seed, chain, dim, n_transitions):
        self.xi, self.u = oracle.rng_fill(seed, chain, 1, 0, n_transitions, dim)
        self.xi, self.u = self.xi[:, 0, :], self.u[:, 0]
        self.t = 0

    def standard_normal(self, size=None):
        out = self.xi[self.t].copy()
        assert size is None or int(size) == out.size
        return out

    def uniform(self, low=0.0, high=1.0, size=None):
        return np.array([self.u[self.t]])

    def install(self):
        ref_gaussian.standard_normal = self.standard_normal
        ref_mh.uniform = self.uniform

    def watch(self, mh):
        inner = mh._accept_reject

        def wrapped(proposal, state):
            try:
                return inner(proposal, state)
            finally:
                self.t += 1

        mh._accept_reject = wrapped


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def eval_log_along(target, traj, make_param):
    return np.array([float(np.asarray(target.evaluate_log(make_param(s))).reshape(-1)[0])
                     for s in traj])


def scipy_whitening(density2d):
    co = density2d.dist_.cov_object
    return (np.array(density2d.dist_.mean, dtype=float), np.array(co._LP, dtype=float),
            float(co.rank * np.log(2.0 * np.pi) + co._log_pdet))


def sigma_of(cov):
    # covariance.py:51-52: sqrt(reciprocal(precision)), precision = reciprocal(var)
    return np.sqrt(np.reciprocal(cov._precision))


# ---------------------------------------------------------------------------
# F1: example_mcmc_1d.py shape -- 1 chain, 1-D Gaussian target, ScalarParameter
# ---------------------------------------------------------------------------

def fixture_f1():
    seed, n = 20250103, 10000
    mean, var, prop_var, x0 = 1.5, 1.0, 1.5, -3.0
    target = GaussianTargetDensity1d(ScalarParameter(np.array([mean])), var)
    cov = IIDCovarianceMatrix(1, prop_var)
    from yagremcmc.chain.method.mrw import MetropolisedRandomWalk
    diag = FullDiagnostics()
    mc = MetropolisedRandomWalk(target, cov, diag)
    rp = Replay(seed, 0, 1, n - 1)
    rp.install()
    mc.run(n, ScalarParameter(np.array([x0])), verbose=False)
    traj = np.array(mc.chain.trajectory).reshape(n, 1, 1)
    acc = np.array(diag._diagnostics._decisions, dtype=np.uint8).reshape(n - 1, 1)
    lp = eval_log_along(target, traj[:, 0, :], lambda s: ScalarParameter(np.array(s))).reshape(n, 1)
    np.savez_compressed(
        os.path.join(HERE, "f1_mrw_1d.npz"), seed=seed, n_states=n, mean=mean, variance=var,
        sigma=sigma_of(cov), x0=np.array([[x0]]), traj=traj, acc=acc, logpost=lp,
        acc_rate=diag.global_acceptance_rate(), wf_mean=np.array(diag.mean()),
        wf_var=np.array(diag.marginal_variance()))
    print("F1", acc.mean())


# ---------------------------------------------------------------------------
# F2 (+F6): example_mcmc_2d_singleLevel.py shape -- 2-D Gaussian, MRW
# ---------------------------------------------------------------------------

TGT_MEAN = np.array([1.0, 1.5])
TGT_COV = np.array([[2.4, -0.5], [-0.5, 0.7]])


def run_mrw_2d(name, seed, K, n, proposal_cov, x0):
    target = GaussianTargetDensity2d(ParameterVector(TGT_MEAN.copy()), TGT_COV.copy())
    mu, U, c = scipy_whitening(target)
    traj = np.empty((n, K, 2))
    acc = np.empty((n - 1, K), dtype=np.uint8)
    lp = np.empty((n, K))
    margin = np.empty((n - 1, K))
    rate = np.empty(K)
    wmean = np.empty((K, 2))
    wvar = np.empty((K, 2))
    sig = sigma_of(proposal_cov)
    for k in range(K):
        b = MRWBuilder()
        b.explicitTarget = target
        b.proposalCovariance = proposal_cov
        b.diagnostics = FullDiagnostics()
        mc = b.build_method()
        rp = Replay(seed, k, 2, n - 1)
        rp.install()
        mc.run(n, ParameterVector(x0.copy()), verbose=False)
        traj[:, k, :] = np.array(mc.chain.trajectory)
        acc[:, k] = mc.diagnostics._diagnostics._decisions
        lp[:, k] = eval_log_along(target, traj[:, k, :], lambda s: ParameterVector(np.array(s)))
        # margin |u - a| of every decision, from reference values
        props = traj[:-1, k, :] + sig * rp.xi
        lpp = eval_log_along(target, props, lambda s: ParameterVector(np.array(s)))
        a = np.minimum(np.exp(lpp - lp[:-1, k]), 1.0)
        margin[:, k] = np.abs(rp.u - a)
        rate[k] = mc.diagnostics.global_acceptance_rate()
        wmean[k] = mc.diagnostics.mean()
        wvar[k] = mc.diagnostics.marginal_variance()
    np.savez_compressed(
        os.path.join(HERE, name), seed=seed, n_states=n, mean=mu, whiten=U, log_norm=c,
        cov=TGT_COV, sigma=sig, x0=np.tile(x0, (K, 1)), traj=traj, acc=acc, logpost=lp,
        min_margin=margin.min(), acc_rate=rate, wf_mean=wmean, wf_var=wvar)
    print(name, acc.mean(), "min margin", margin.min())


# ---------------------------------------------------------------------------
# F5: example_mcmc_2d_twoLevel.py shape -- two-level delayed acceptance, S = 6
# ---------------------------------------------------------------------------

def fixture_f5():
    seed, K, n, S = 20250105, 64, 1000, 6
    x0 = np.array([-8.0, -7.0])
    sur_mean = TGT_MEAN + np.array([0.15, -0.2])
    sur_cov = 2.0 * TGT_COV + np.array([[0.1, 1.2], [1.2, 0.05]])
    fine = GaussianTargetDensity2d(ParameterVector(TGT_MEAN.copy()), TGT_COV.copy())
    coarse = GaussianTargetDensity2d(ParameterVector(sur_mean.copy()), sur_cov.copy())
    mu_f, U_f, c_f = scipy_whitening(fine)
    mu_c, U_c, c_c = scipy_whitening(coarse)
    cov = IIDCovarianceMatrix(2, 1.0)
    traj = np.empty((n, K, 2))
    acc = np.empty((n - 1, K), dtype=np.uint8)
    sub = np.empty((n - 1, S, K), dtype=np.uint8)
    lp = np.empty((n, K))
    for k in range(K):
        b = MLDABuilder()
        b.explicitTarget = fine
        b.surrogateTargets = [coarse]
        b.baseProposalCovariance = cov
        b.subChainLengths = [S]
        b.surrogateDiagnostics = [AcceptanceRateDiagnostics()]
        mc = b.build_method()
        rp = Replay(seed, k, 2, (n - 1) * (S + 1), sub_chain_length=S)
        rp.install()
        mark_fine_level(mc, rp)
        mc.run(n, ParameterVector(x0.copy()), verbose=False)
        traj[:, k, :] = np.array(mc.chain.trajectory)
        acc[:, k] = mc.diagnostics._decisions
        sub[:, :, k] = np.array(mc.surrogate(0).diagnostics._decisions,
                                dtype=np.uint8).reshape(n - 1, S)
        lp[:, k] = eval_log_along(fine, traj[:, k, :], lambda s: ParameterVector(np.array(s)))
    np.savez_compressed(
        os.path.join(HERE, "f5_mlda2_2d.npz"), seed=seed, n_states=n, S=S,
        fine_mean=mu_f, fine_whiten=U_f, fine_log_norm=c_f, fine_cov=TGT_COV,
        coarse_mean=mu_c, coarse_whiten=U_c, coarse_log_norm=c_c, coarse_cov=sur_cov,
        sigma=sigma_of(cov), x0=np.tile(x0, (K, 1)), traj=traj, acc=acc, sub_acc=sub, logpost=lp)
    print("F5", acc.mean(), sub.mean())


# ---------------------------------------------------------------------------
# F3: linear forward model posterior (likelihood.py:49-87 over exampleSetup.py)
# ---------------------------------------------------------------------------

def linear_problem(d, n_data, gen_seed):
    """C3-shaped synthetic problem: A ~ N(0,1)/sqrt(d) etc."""
    z, _ = oracle.rng_fill(gen_seed, 0, d + 3 + n_data, 0, 1, d, uniforms=False)
    z = z[0]
    A = z[:d] / np.sqrt(d)
    b = z[d]
    theta_true = z[d + 1]
    noise_var = 0.3
    data = np.array([A @ theta_true + b + np.sqrt(noise_var) * z[d + 3 + i] for i in range(n_data)])
    return A, b, data, noise_var


def build_linear_reference(A, b, data, noise_var, prior_mean, prior_var):
    d = A.shape[1]
    solver = ExampleLinearModelSolver(A, b)
    like = AdditiveGaussianNoiseLikelihood(
        Data(data), ForwardModel(solver), CentredGaussianNoise(IIDCovarianceMatrix(A.shape[0], noise_var)))
    prior = Gaussian(ParameterVector(prior_mean.copy()), IIDCovarianceMatrix(d, prior_var))
    return BayesianRegressionModel(like, prior)


def run_mrw_model(model, seed, K, n, prop_cov, x0, make_param=ParameterVector, keep_traj=None):
    d = x0.size
    traj = np.empty((n, K, d))
    acc = np.empty((n - 1, K), dtype=np.uint8)
    lp = np.empty((n, K))
    for k in range(K):
        b = MRWBuilder()
        b.bayesModel = model
        b.proposalCovariance = prop_cov
        mc = b.build_method()
        rp = Replay(seed, k, d, n - 1)
        rp.install()
        mc.run(n, make_param(x0.copy()), verbose=False)
        traj[:, k, :] = np.array(mc.chain.trajectory)
        acc[:, k] = mc.diagnostics._decisions
        lp[:, k] = eval_log_along(mc.target, traj[:, k, :], lambda s: make_param(np.array(s)))
    return traj, acc, lp


def fixture_f3():
    d, n_data, seed, K, n = 128, 5, 20250107, 16, 500
    A, b, data, noise_var = linear_problem(d, n_data, 7)
    prior_mean, prior_var, prop_var = np.zeros(d), 5.0, 1.1e-3
    model = build_linear_reference(A, b, data, noise_var, prior_mean, prior_var)
    prop_cov = IIDCovarianceMatrix(d, prop_var)
    traj, acc, lp = run_mrw_model(model, seed, K, n, prop_cov, np.zeros(d))
    np.savez_compressed(
        os.path.join(HERE, "f3_linear_d128.npz"), seed=seed, n_states=n, A=A, b=b, data=data,
        noise_precision=np.reciprocal(np.full(d, noise_var)), prior_mean=prior_mean,
        prior_precision=np.reciprocal(np.full(d, prior_var)), sigma=sigma_of(prop_cov),
        x0=np.zeros((K, d)), traj_head=traj[:, :2, :], traj_sha256=sha(traj), acc=acc, logpost=lp)
    print("F3", acc.mean())


def fixture_f22(name="f22_linear_d192_m160.npz", d=192, m=160, seed=20250122, K=8, n=200, prop_var=8e-4):
    """Beyond the LDS-resident shapes: d = 192 parameters, m = 160 outputs (a 240 KB forward matrix), the reference's
    shape-agnostic linear model (exampleSetup.py:42-52, likelihood.py:74-84) under a random walk, 8 chains x 200 states.
    F22b: d = 320, m = 48 (beyond the matrix-core kernels' 256 parameters: the wide path), 4 chains x 120 states."""
    n_data = 4
    z, _ = oracle.rng_fill(22, 0, m + 3 + n_data, 0, 1, d, uniforms=False)
    z = z[0]
    A = z[:m] / np.sqrt(d)
    b = z[m][:m]
    theta_true = z[m + 1]
    noise_var = 0.3
    data = np.array([A @ theta_true + b + np.sqrt(noise_var) * z[m + 3 + i][:m] for i in range(n_data)])
    prior_mean, prior_var = np.zeros(d), 5.0
    model = build_linear_reference(A, b, data, noise_var, prior_mean, prior_var)
    prop_cov = IIDCovarianceMatrix(d, prop_var)
    traj, acc, lp = run_mrw_model(model, seed, K, n, prop_cov, np.zeros(d))
    np.savez_compressed(
        os.path.join(HERE, name), seed=seed, n_states=n, A=A, b=b, data=data,
        noise_precision=np.reciprocal(np.full(m, noise_var)), prior_mean=prior_mean,
        prior_precision=np.reciprocal(np.full(d, prior_var)), sigma=sigma_of(prop_cov),
        x0=np.zeros((K, d)), traj_head=traj[:, :2, :], traj_sha256=sha(traj), acc=acc, logpost=lp)
    print(name, acc.mean())


def fixture_f3b():
    """The shipped 2x2 example model (example_inference_linearModel_twoLevel.py:24-82)."""
    from numpy.random import seed as np_seed, standard_normal
    np_seed(2222)
    A = np.array([[1.4, -0.2], [-0.6, 0.7]])
    b = np.zeros(2)
    true = np.array([1.5, 0.5])
    sd = np.sqrt(0.3)
    data = np.array([A @ true + b + sd * standard_normal(2) for _ in range(5)])
    prior_mean = true + np.array([-0.2, 0.4])
    model = build_linear_reference(A, b, data, sd ** 2, prior_mean, 5.0)
    g = np.linspace(-2.0, 4.0, 25)
    grid = np.array([[x, y] for x in g for y in g])
    from yagremcmc.chain.target import UnnormalisedPosterior
    post = UnnormalisedPosterior(model.likelihood, model.prior)
    vals = np.array([post.evaluate_log(ParameterVector(p)) for p in grid])
    seed, K, n = 20250108, 8, 500
    prop_cov = IIDCovarianceMatrix(2, 0.5)
    traj, acc, lp = run_mrw_model(model, seed, K, n, prop_cov, np.array([0.0, 0.0]))
    np.savez_compressed(
        os.path.join(HERE, "f3b_linear_2x2.npz"), seed=seed, n_states=n, A=A, b=b, data=data,
        noise_precision=np.reciprocal(np.full(2, sd ** 2)), prior_mean=prior_mean,
        prior_precision=np.reciprocal(np.full(2, 5.0)), sigma=sigma_of(prop_cov),
        x0=np.zeros((K, 2)), grid=grid, grid_logpost=vals, traj=traj, acc=acc, logpost=lp)
    print("F3b", acc.mean())


# ---------------------------------------------------------------------------
# F4: Lotka-Volterra (test_inference_mcmc_singleLevel.py:28-73 wiring)
# ---------------------------------------------------------------------------

class RK4LotkaVolterraSolver(SolverInterface):
    """The build's fixed-step RK4 integrator as a reference SolverInterface
    plug-in, so the reference LOOP can be replayed with the build's forward
    map (loop parity exact; the integrator itself is checked separately)."""

    def __init__(self, problem):
        self._p = problem
        self._theta = None
        self._eval = None
        self._status = EvaluationStatus.NONE

    @property
    def status(self):
        return self._status

    @property
    def evaluation(self):
        return self._eval

    def interpolate(self, parameter):
        self._theta = np.array(parameter.coefficient, dtype=float)

    def invoke(self):
        out = oracle.forward(self._p, self._theta)
        self._eval = out.reshape(self._p["n_data"], 2)
        self._status = (EvaluationStatus.SUCCESS if np.all(np.isfinite(out))
                        else EvaluationStatus.FAILURE)


def fixture_f4():
    cfg = {'T': 6., 'alpha': 0.8, 'gamma': 0.4, 'nData': 5, 'dataDim': 2,
           'solver': 'LSODA', 'rtol': 1e-4}
    design = np.array([[0.1, 0.9], [0.5, 0.5], [1., 0.5], [0.5, 1.], [2.5, 1.5]])
    ref_solver = LotkaVolterraSolver(design, cfg)
    tight = LotkaVolterraSolver(design, dict(cfg, solver='DOP853', rtol=1e-12))
    # forward map at 32 parameter values around the test's posterior bulk
    zz, uu = oracle.rng_fill(4444, 0, 32, 0, 1, 2)
    thetas = np.log(np.array([0.4, 0.6])) + 0.25 * zz[0]
    f_ref = np.empty((32, 5, 2))
    f_tight = np.empty((32, 5, 2))
    for i, th in enumerate(thetas):
        p = LotkaVolterraParameter.from_coefficient(th)
        for s, out in ((ref_solver, f_ref), (tight, f_tight)):
            s.interpolate(p)
            s.invoke()
            assert s.status == EvaluationStatus.SUCCESS
            out[i] = s.evaluation
    # synthetic data from the tight solver + stream noise
    truth = np.log(np.array([0.4, 0.6]))
    tight.interpolate(LotkaVolterraParameter.from_coefficient(truth))
    tight.invoke()
    noise_var = 0.05
    zn, _ = oracle.rng_fill(5555, 0, 5, 0, 1, 2, uniforms=False)
    data = tight.evaluation + np.sqrt(noise_var) * zn[0]
    prior_mean, prior_var = np.zeros(2), 0.02
    problem = oracle.posterior_lv(design, cfg['T'], cfg['alpha'], cfg['gamma'], 50, data,
                                  np.reciprocal(np.full(2, noise_var)), prior_mean,
                                  np.reciprocal(np.full(2, prior_var)))
    like = AdditiveGaussianNoiseLikelihood(
        Data(data), ForwardModel(RK4LotkaVolterraSolver(problem)),
        CentredGaussianNoise(IIDCovarianceMatrix(2, noise_var)))
    prior = Gaussian(LotkaVolterraParameter.from_coefficient(prior_mean.copy()),
                     IIDCovarianceMatrix(2, prior_var))
    model = BayesianRegressionModel(like, prior)
    seed, K, n = 20250104, 16, 300
    prop_cov = DiagonalCovarianceMatrix(np.array([0.02, 0.01]))
    x0 = np.array([-0.6, -0.3])
    traj, acc, lp = run_mrw_model(model, seed, K, n, prop_cov, x0,
                                  make_param=LotkaVolterraParameter.from_coefficient)
    np.savez_compressed(
        os.path.join(HERE, "f4_lotka_volterra.npz"), seed=seed, n_states=n, design=design,
        t_final=cfg['T'], alpha=cfg['alpha'], gamma=cfg['gamma'], rk4_steps=50, data=data,
        noise_precision=np.reciprocal(np.full(2, noise_var)), prior_mean=prior_mean,
        prior_precision=np.reciprocal(np.full(2, prior_var)), sigma=sigma_of(prop_cov),
        x0=np.tile(x0, (K, 1)), thetas=thetas, forward_lsoda_rtol1e4=f_ref,
        forward_dop853_rtol1e12=f_tight, traj=traj, acc=acc, logpost=lp)
    print("F4", acc.mean(), "RK4-50 vs DOP853:",
          np.abs(oracle.forward(problem, thetas).reshape(32, 5, 2) - f_tight).max(),
          "LSODA vs DOP853:", np.abs(f_ref - f_tight).max())


def fixture_f24():
    """example_inference_lotkaVolterra_singleLevel.py as shipped -- T = 10, ten design rows in [0.5, 1.5]^2, noise variance
    0.04, prior N(0, 1.4 I), proposal variance 0.15 -- FROM ITS OWN STARTING POINT (-7, 2.8) (:106), where the predator
    first grows at rate 16 and RK4 in 50 steps overflows: the reference's loop with the build's ADAPTIVE integrator
    (ymc_lv_dp45_solve at the example's rtol 1e-4) as its solver plug-in; the forward map there and around the
    posterior bulk against the reference's LSODA and a tight DOP853."""
    cfg = {'T': 10., 'alpha': 0.8, 'gamma': 0.4, 'nData': 10, 'dataDim': 2, 'solver': 'LSODA', 'rtol': 1e-4}
    _, uu = oracle.rng_fill(2424, 0, 20, 0, 1, 2)
    design = 0.5 + uu[0, :20].reshape(10, 2)
    ref_solver = LotkaVolterraSolver(design, cfg)
    tight = LotkaVolterraSolver(design, dict(cfg, solver='DOP853', rtol=1e-12))
    zz, _ = oracle.rng_fill(2425, 0, 24, 0, 1, 2, uniforms=False)
    thetas = np.concatenate([np.log(np.array([0.4, 0.6])) + 0.3 * zz[0], [[-7.0, 2.8], [-6.5, 2.5], [-3.0, 3.0], [1.5, -2.0]]])
    f_ref = np.empty((len(thetas), 10, 2))
    f_tight = np.empty((len(thetas), 10, 2))
    for i, th in enumerate(thetas):
        prm = LotkaVolterraParameter.from_coefficient(th)
        for s_, out in ((ref_solver, f_ref), (tight, f_tight)):
            s_.interpolate(prm)
            s_.invoke()
            assert s_.status == EvaluationStatus.SUCCESS
            out[i] = s_.evaluation
    tight.interpolate(LotkaVolterraParameter.from_interpolation(np.array([0.4, 0.6])))
    tight.invoke()
    noise_var = 0.04
    zn, _ = oracle.rng_fill(2426, 0, 10, 0, 1, 2, uniforms=False)
    data = tight.evaluation + np.sqrt(noise_var) * zn[0]
    prior_mean, prior_var = np.zeros(2), 1.4
    rtol, atol = 1e-4, 1e-6
    problem = oracle.posterior_lv(design, cfg['T'], cfg['alpha'], cfg['gamma'], 50, data,
                                  np.reciprocal(np.full(2, noise_var)), prior_mean, np.reciprocal(np.full(2, prior_var)),
                                  rtol=rtol, atol=atol)
    like = AdditiveGaussianNoiseLikelihood(Data(data), ForwardModel(RK4LotkaVolterraSolver(problem)),
                                           CentredGaussianNoise(IIDCovarianceMatrix(2, noise_var)))
    prior = Gaussian(LotkaVolterraParameter.from_coefficient(prior_mean.copy()), IIDCovarianceMatrix(2, prior_var))
    model = BayesianRegressionModel(like, prior)
    seed, K, n = 20250124, 8, 500
    prop_cov = IIDCovarianceMatrix(2, 0.15)
    x0 = np.array([-7.0, 2.8])
    traj, acc, lp = run_mrw_model(model, seed, K, n, prop_cov, x0, make_param=LotkaVolterraParameter.from_coefficient)
    np.savez_compressed(
        os.path.join(HERE, "f24_lotka_volterra_adaptive.npz"), seed=seed, n_states=n, design=design, t_final=cfg['T'],
        alpha=cfg['alpha'], gamma=cfg['gamma'], rtol=rtol, atol=atol, data=data,
        noise_precision=np.reciprocal(np.full(2, noise_var)), prior_mean=prior_mean,
        prior_precision=np.reciprocal(np.full(2, prior_var)), sigma=sigma_of(prop_cov), x0=np.tile(x0, (K, 1)),
        thetas=thetas, forward_lsoda_rtol1e4=f_ref, forward_dop853_rtol1e12=f_tight, traj=traj, acc=acc, logpost=lp)
    f = oracle.forward(problem, thetas).reshape(-1, 10, 2)
    print("F24", acc.mean(), "final states", traj[-1, :3], "adaptive vs DOP853 (rel):",
          (np.abs(f - f_tight) / (1e-3 + np.abs(f_tight))).max(), "LSODA vs DOP853 (rel):",
          (np.abs(f_ref - f_tight) / (1e-3 + np.abs(f_tight))).max())


def fixture_f7():
    """Two-level delayed acceptance over a model hierarchy: the shipped linear example
    (example_inference_linearModel_twoLevel.py:24-177, vanilla MLDA, sub-chain length 5)."""
    from numpy.random import seed as np_seed, standard_normal
    from yagremcmc.statistics.modelHierarchy import BayesianRegressionModelHierarchy
    from yagremcmc.utility.hierarchy import Hierarchy, SharedComponent
    np_seed(2222)
    A_t, b_t = np.array([[1.4, -0.2], [-0.6, 0.7]]), np.zeros(2)
    A_s, b_s = A_t + np.array([[-0.6, -0.2], [0.4, 1.1]]), np.array([0.5, -0.9])
    true = np.array([1.5, 0.5])
    sd = np.sqrt(0.3)
    data = np.array([A_t @ true + b_t + sd * standard_normal(2) for _ in range(5)])
    prior_mean = true + np.array([-0.2, 0.4])
    prior = Gaussian(ParameterVector(prior_mean.copy()), IIDCovarianceMatrix(2, 5.0))
    noise = CentredGaussianNoise(IIDCovarianceMatrix(2, sd ** 2))
    lik_s = AdditiveGaussianNoiseLikelihood(Data(data), ForwardModel(ExampleLinearModelSolver(A_s, b_s)), noise)
    lik_t = AdditiveGaussianNoiseLikelihood(Data(data), ForwardModel(ExampleLinearModelSolver(A_t, b_t)), noise)
    model = BayesianRegressionModelHierarchy(Hierarchy([lik_s, lik_t]), SharedComponent(prior, 2))
    seed, K, n, S = 20250109, 32, 600, 5
    cov = IIDCovarianceMatrix(2, 0.5)
    x0 = np.zeros(2)
    traj = np.empty((n, K, 2))
    acc = np.empty((n - 1, K), dtype=np.uint8)
    sub = np.empty((n - 1, S, K), dtype=np.uint8)
    lp = np.empty((n, K))
    for k in range(K):
        b = MLDABuilder()
        b.baseProposalCovariance = cov
        b.subChainLengths = [S]
        b.bayesModel = model
        b.surrogateDiagnostics = [AcceptanceRateDiagnostics()]
        mc = b.build_method()
        rp = Replay(seed, k, 2, (n - 1) * (S + 1), sub_chain_length=S)
        rp.install()
        mark_fine_level(mc, rp)
        mc.run(n, ParameterVector(x0.copy()), verbose=False)
        traj[:, k, :] = np.array(mc.chain.trajectory)
        acc[:, k] = mc.diagnostics._decisions
        sub[:, :, k] = np.array(mc.surrogate(0).diagnostics._decisions, dtype=np.uint8).reshape(n - 1, S)
        lp[:, k] = eval_log_along(mc.target, traj[:, k, :], lambda s: ParameterVector(np.array(s)))
    np.savez_compressed(
        os.path.join(HERE, "f7_mlda2_linear_models.npz"), seed=seed, n_states=n, S=S, A_fine=A_t, b_fine=b_t,
        A_coarse=A_s, b_coarse=b_s, data=data, noise_precision=np.reciprocal(np.full(2, sd ** 2)),
        prior_mean=prior_mean, prior_precision=np.reciprocal(np.full(2, 5.0)), sigma=sigma_of(cov),
        x0=np.zeros((K, 2)), traj=traj, acc=acc, sub_acc=sub, logpost=lp)
    print("F7", acc.mean(), sub.mean())


def fixture_f8():
    """Preconditioned Crank-Nicolson on the Lotka-Volterra model (test_inference_mcmc_singleLevel.py:121-148
    wiring, the build's RK4 as the solver plug-in, centred prior) and on the 2x2 linear model."""
    from yagremcmc.chain.method.pcn import PCNBuilder
    g4 = np.load(os.path.join(HERE, "f4_lotka_volterra.npz"))
    problem = oracle.posterior_lv(g4["design"], float(g4["t_final"]), float(g4["alpha"]), float(g4["gamma"]), 50,
                                  g4["data"], g4["noise_precision"], g4["prior_mean"], g4["prior_precision"])
    like = AdditiveGaussianNoiseLikelihood(
        Data(g4["data"]), ForwardModel(RK4LotkaVolterraSolver(problem)),
        CentredGaussianNoise(IIDCovarianceMatrix(2, 0.05)))
    prior_cov = DiagonalCovarianceMatrix(np.array([0.02, 0.05]))
    prior = Gaussian(LotkaVolterraParameter.from_coefficient(np.zeros(2)), prior_cov)
    model = BayesianRegressionModel(like, prior)
    seed, K, n, h = 20250110, 16, 400, 0.01
    x0 = np.array([-0.6, -0.3])
    traj = np.empty((n, K, 2))
    acc = np.empty((n - 1, K), dtype=np.uint8)
    lp = np.empty((n, K))
    for k in range(K):
        b = PCNBuilder()
        b.stepSize = h
        b.bayesModel = model
        mc = b.build_method()
        rp = Replay(seed, k, 2, n - 1)
        rp.install()
        mc.run(n, LotkaVolterraParameter.from_coefficient(x0.copy()), verbose=False)
        traj[:, k, :] = np.array(mc.chain.trajectory)
        acc[:, k] = mc.diagnostics._decisions
        lp[:, k] = eval_log_along(mc.target, traj[:, k, :], lambda s: LotkaVolterraParameter.from_coefficient(np.array(s)))
    np.savez_compressed(
        os.path.join(HERE, "f8_pcn_lotka_volterra.npz"), seed=seed, n_states=n, step_size=h, design=g4["design"],
        t_final=g4["t_final"], alpha=g4["alpha"], gamma=g4["gamma"], rk4_steps=50, data=g4["data"],
        noise_precision=np.reciprocal(np.full(2, 0.05)), prior_mean=np.zeros(2), prior_precision=prior_cov._precision,
        prior_sigma=sigma_of(prior_cov), x0=np.tile(x0, (K, 1)), traj=traj, acc=acc, loglik=lp)
    print("F8", acc.mean())


def fixture_f9():
    """Post-processing: the reference's IAT / ACF (postprocessing/autocorrelation.py) of the F2 and F1 chains."""
    import yagremcmc.postprocessing.autocorrelation as ac
    g2 = np.load(os.path.join(HERE, "f2_mrw_2d.npz"))
    traj = g2["traj"]                                   # (2000, 64, 2), produced by the reference
    out = {}
    for burnin in (0, 200):
        out[f"iat_mean_b{burnin}"] = np.array([ac.integrated_autocorrelation(traj[burnin:, k, :], 'mean')
                                               for k in range(traj.shape[1])], dtype=np.int32)
        out[f"iat_max_b{burnin}"] = np.array([ac.integrated_autocorrelation(traj[burnin:, k, :], 'max')
                                              for k in range(traj.shape[1])], dtype=np.int32)
    out["acf_chain0_dim0_b200"] = ac.estimate_autocorrelation_function_1d(traj[200:, 0, 0])
    g1 = np.load(os.path.join(HERE, "f1_mrw_1d.npz"))
    s1 = g1["traj"][:, 0, :]                             # (10000, 1) as np.array(mcmc.chain.trajectory)
    out["f1_iat_mean_b200"] = ac.integrated_autocorrelation(s1[200:], 'mean')
    out["f1_iat_max_b200"] = ac.integrated_autocorrelation(s1[200:], 'max')
    np.savez_compressed(os.path.join(HERE, "f9_iat.npz"), **out)
    print("F9", out["iat_mean_b200"][:8], out["iat_max_b200"][:8], out["f1_iat_mean_b200"])


# ---------------------------------------------------------------------------
# F10 / F11: MLDA with 2 and with 4 surrogates as the reference executes them
# (example_mcmc_2d_hierarchical.py:13-47; test_mlda.py:189-256)
# ---------------------------------------------------------------------------

def run_nested(name, seed, K, n, tgt_mean, tgt_cov, sur_means, sur_covs, sub_lengths, x0):
    target = GaussianTargetDensity2d(ParameterVector(np.array(tgt_mean, dtype=float)), np.array(tgt_cov, dtype=float))
    surs = [GaussianTargetDensity2d(ParameterVector(np.array(m, dtype=float)), np.array(c, dtype=float))
            for m, c in zip(sur_means, sur_covs)]
    ns = len(surs)
    E = ns // 2 + 1
    steps = [sub_lengths[2 * e + 1] for e in range(E - 1)]
    cost = 1
    for s_ in steps:
        cost = s_ * cost + 1
    cov = IIDCovarianceMatrix(2, 1.0)
    traj = np.empty((n, K, 2))
    acc = np.empty((n - 1, K), dtype=np.uint8)
    lp = np.empty((n, K))
    level_acc = np.zeros((E - 1, K), dtype=np.uint64)
    for k in range(K):
        b = MLDABuilder()
        b.explicitTarget = target
        b.surrogateTargets = list(surs)
        b.baseProposalCovariance = cov
        b.subChainLengths = list(sub_lengths)
        b.surrogateDiagnostics = [AcceptanceRateDiagnostics() for _ in range(ns)]
        mc = b.build_method()
        rp = SequentialReplay(seed, k, 2, (n - 1) * cost)
        rp.install()
        rp.watch(mc)
        for i in range(ns):
            rp.watch(mc.surrogate(i))
        mc.run(n, ParameterVector(np.array(x0, dtype=float)), verbose=False)
        assert rp.t == (n - 1) * cost, (rp.t, (n - 1) * cost)
        traj[:, k, :] = np.array(mc.chain.trajectory)
        acc[:, k] = mc.diagnostics._decisions
        lp[:, k] = eval_log_along(target, traj[:, k, :], lambda s: ParameterVector(np.array(s)))
        for e in range(E - 1):
            dec = mc.surrogate(2 * e).diagnostics._decisions
            level_acc[e, k] = int(np.sum(dec))
        for i in range(1, ns, 2):   # the levels whose accept step never runs
            assert len(mc.surrogate(i).diagnostics._decisions) == 0
    mu_t, U_t, c_t = scipy_whitening(target)
    sw = [scipy_whitening(s_) for s_ in surs]
    np.savez_compressed(
        os.path.join(HERE, name), seed=seed, n_states=n, sub_lengths=np.array(sub_lengths),
        tgt_mean=mu_t, tgt_whiten=U_t, tgt_log_norm=c_t,
        sur_mean=np.array([w[0] for w in sw]), sur_whiten=np.array([w[1] for w in sw]),
        sur_log_norm=np.array([w[2] for w in sw]), sigma=sigma_of(cov),
        x0=np.tile(np.array(x0, dtype=float), (K, 1)), traj=traj, acc=acc, logpost=lp, level_acc=level_acc,
        cost_top=cost)
    print(name, acc.mean(), level_acc.sum(axis=1))


def fixture_f10():
    # example_mcmc_2d_hierarchical.py: two surrogates, subChainLengths [4, 4]
    tgt_mean, tgt_cov = [1.0, 1.5], [[2.4, -0.5], [-0.5, 0.7]]
    run_nested("f10_mlda_nested2.npz", 20250110, 32, 400, tgt_mean, tgt_cov,
               [[1.0 - 0.05, 1.5 + 0.01], [1.0, 1.5 - 0.01]],
               [3.0 * np.array([[2.8, -0.1], [-0.1, 1.7]]), 1.5 * np.array([[2.4, -0.3], [-0.3, 1.1]])],
               [4, 4], [-8.0, -7.0])


def fixture_f11():
    # test_mlda.py:189-256: four surrogates, subChainLengths [6, 4, 4, 3]
    tm = np.array([1.0, 1.5])
    means = [tm + np.array([0.2, -0.3]), tm + np.array([-0.1, 0.1]), tm + np.array([0.05, -0.05]), tm.copy()]
    covs = [8. * np.array([[3.0, -0.2], [-0.2, 1.5]]), 6. * np.array([[2.7, -0.25], [-0.25, 1.3]]),
            4. * np.array([[2.5, -0.3], [-0.3, 1.1]]), 2. * np.array([[2.4, -0.35], [-0.35, 1.0]])]
    run_nested("f11_mlda_nested4.npz", 20250111, 16, 300, tm, [[2.4, -0.5], [-0.5, 0.7]], means, covs,
               [6, 4, 4, 3], [2.0, -3.0])


def fixture_f23():
    # deeper than any of the reference's own examples: eight and fourteen surrogates (mlda.py:46-73 builds any
    # number of levels) = five and eight effective levels; the coarser a surrogate, the wider and the further off
    tm = np.array([1.0, 1.5])
    base = np.array([[2.4, -0.5], [-0.5, 0.7]])

    def hierarchy(ns):
        means = [tm + (ns - i) * np.array([0.04, -0.03]) * (-1.0) ** i for i in range(ns)]
        covs = [(1.0 + 0.45 * (ns - i)) * (base + 0.05 * (ns - i) * np.eye(2)) for i in range(ns)]
        return means, covs

    means, covs = hierarchy(8)
    run_nested("f23_mlda_nested8.npz", 20250123, 8, 150, tm, base, means, covs, [2, 3, 5, 2, 1, 2, 4, 2], [2.0, -3.0])
    means, covs = hierarchy(14)
    run_nested("f23b_mlda_nested14.npz", 20250124, 6, 40, tm, base, means, covs,
               [3, 2, 2, 2, 5, 1, 1, 2, 7, 2, 2, 2, 1, 2], [-1.0, 0.5])


# ---------------------------------------------------------------------------
# F12: adaptive error model (chain/method/aem.py) on the linear two-level example's models
# (example_inference_linearModel_twoLevel.py:33-56,95-100,181-191), with a milder surrogate
# error and a small minDataSize so that the shift AND the variance inflation become active
# within the fixture; fresh AEMLikelihood objects per chain (they carry the learned state)
# ---------------------------------------------------------------------------

def fixture_f12(name="f12_aem_linear.npz", use_heuristic=True, seed=20250112, min_data=8, K=24, n=500, S=5):
    from numpy.random import seed as np_seed, standard_normal
    from yagremcmc.chain.method.aem import AEMBuilder
    from yagremcmc.statistics.likelihood import AEMLikelihood
    from yagremcmc.statistics.modelHierarchy import BayesianRegressionModelHierarchy
    from yagremcmc.utility.hierarchy import Hierarchy, SharedComponent
    np_seed(2222)
    A_t, b_t = np.array([[1.4, -0.2], [-0.6, 0.7]]), np.zeros(2)
    A_s, b_s = A_t + 0.25 * np.array([[-0.6, -0.2], [0.4, 1.1]]), 0.25 * np.array([0.5, -0.9])
    true = np.array([1.5, 0.5])
    sd = np.sqrt(0.3)
    data = np.array([A_t @ true + b_t + sd * standard_normal(2) for _ in range(5)])
    prior_mean = true + np.array([-0.2, 0.4])
    cov = IIDCovarianceMatrix(2, 0.5)
    x0 = np.zeros(2)
    traj = np.empty((n, K, 2))
    acc = np.empty((n - 1, K), dtype=np.uint8)
    sub = np.empty((n - 1, S, K), dtype=np.uint8)
    lp = np.empty((n, K))
    err_n = np.zeros(K, dtype=np.uint64)
    err_mean = np.zeros((K, 2))
    err_mvar = np.zeros((K, 2))
    aem_prec = np.zeros((K, 2))
    hits = np.zeros(K, dtype=np.int64)
    for k in range(K):
        prior = Gaussian(ParameterVector(prior_mean.copy()), IIDCovarianceMatrix(2, 5.0))
        noise = CentredGaussianNoise(IIDCovarianceMatrix(2, sd ** 2))
        lik_s = AEMLikelihood(Data(data), ForwardModel(ExampleLinearModelSolver(A_s, b_s)), noise, min_data,
                              use_heuristic)
        lik_t = AEMLikelihood(Data(data), ForwardModel(ExampleLinearModelSolver(A_t, b_t)), noise, min_data,
                              use_heuristic)
        model = BayesianRegressionModelHierarchy(Hierarchy([lik_s, lik_t]), SharedComponent(prior, 2))
        b = AEMBuilder()
        b.baseProposalCovariance = cov
        b.subChainLengths = [S]
        b.bayesModel = model
        b.surrogateDiagnostics = [AcceptanceRateDiagnostics()]
        mc = b.build_method()
        rp = Replay(seed, k, 2, (n - 1) * (S + 1), sub_chain_length=S)
        rp.install()
        mark_fine_level(mc, rp)
        mc.run(n, ParameterVector(x0.copy()), verbose=False)
        traj[:, k, :] = np.array(mc.chain.trajectory)
        acc[:, k] = mc.diagnostics._decisions
        sub[:, :, k] = np.array(mc.surrogate(0).diagnostics._decisions, dtype=np.uint8).reshape(n - 1, S)
        plain_t = AdditiveGaussianNoiseLikelihood(Data(data), ForwardModel(ExampleLinearModelSolver(A_t, b_t)), noise)
        from yagremcmc.chain.target import UnnormalisedPosterior
        lp[:, k] = eval_log_along(UnnormalisedPosterior(plain_t, prior), traj[:, k, :],
                                  lambda s_: ParameterVector(np.array(s_)))
        err_n[k] = lik_s.accumulator.nData
        assert lik_t.accumulator.nData == 0
        if err_n[k] >= 1:
            err_mean[k] = lik_s.accumulator.mean()
        if err_n[k] >= 2:
            err_mvar[k] = lik_s.accumulator.marginal_variance()
        if lik_s._noiseModel._aemNoise is not None:
            aem_prec[k] = lik_s._noiseModel._aemNoise.covariance._precision
        hits[k] = lik_s._cache.hits
    np.savez_compressed(
        os.path.join(HERE, name), seed=seed, n_states=n, S=S, min_data_size=min_data,
        use_noise_heuristic=int(use_heuristic), A_fine=A_t, b_fine=b_t, A_coarse=A_s, b_coarse=b_s, data=data,
        noise_precision=np.reciprocal(np.full(2, sd ** 2)), prior_mean=prior_mean,
        prior_precision=np.reciprocal(np.full(2, 5.0)), sigma=sigma_of(cov), x0=np.zeros((K, 2)), traj=traj,
        acc=acc, sub_acc=sub, logpost=lp, err_n=err_n, err_mean=err_mean, err_mvar=err_mvar, aem_prec=aem_prec)
    print(name, acc.mean(), sub.mean(), "accepted per chain", err_n.min(), err_n.max(), "cache hits", hits.mean())


# ---------------------------------------------------------------------------
# F13: random walk with a DenseCovarianceMatrix proposal (covariance.py:69-94).  The reference forms
# L @ xi with BLAS' dgemv, whose summation order is not the ensemble's: states agree to rounding.
# ---------------------------------------------------------------------------

def fixture_f13():
    seed, K, n = 20250114, 16, 600
    x0 = np.array([-3.0, 2.0])
    target = GaussianTargetDensity2d(ParameterVector(TGT_MEAN.copy()), TGT_COV.copy())
    mu, U, c = scipy_whitening(target)
    prop = 0.6 * TGT_COV + np.array([[0.2, 0.1], [0.1, 0.3]])
    cov = DenseCovarianceMatrix(prop.copy())
    traj = np.empty((n, K, 2))
    acc = np.empty((n - 1, K), dtype=np.uint8)
    lp = np.empty((n, K))
    for k in range(K):
        b = MRWBuilder()
        b.explicitTarget = target
        b.proposalCovariance = cov
        mc = b.build_method()
        rp = Replay(seed, k, 2, n - 1)
        rp.install()
        mc.run(n, ParameterVector(x0.copy()), verbose=False)
        traj[:, k, :] = np.array(mc.chain.trajectory)
        acc[:, k] = mc.diagnostics._decisions
        lp[:, k] = eval_log_along(target, traj[:, k, :], lambda s_: ParameterVector(np.array(s_)))
    np.savez_compressed(os.path.join(HERE, "f13_mrw_2d_dense_proposal.npz"), seed=seed, n_states=n, mean=mu,
                        whiten=U, log_norm=c, proposal_cov=prop, chol=np.array(cov.cholFactor_),
                        x0=np.tile(x0, (K, 1)), traj=traj, acc=acc, logpost=lp)
    print("F13", acc.mean())


def fixture_f14(name="f14_mlda2_linear_d128.npz", d=128, seed=20250114, K=8, n=201, S=5, prop_var=1.1e-3):
    """Two-level delayed acceptance over a model hierarchy at d = 128 (the matrix-core kernel's size): two
    C3-shaped linear models (mlda.py:315-335, modelHierarchy.py:6-53), the coarse one a perturbation of the
    fine one; 8 chains x 200 fine steps, sub-chain length 5; decisions of both levels.  F14b: d = 192 (both
    matrices beyond the LDS: streamed), 4 chains x 80 fine steps, sub-chain length 3."""
    from yagremcmc.statistics.modelHierarchy import BayesianRegressionModelHierarchy
    from yagremcmc.utility.hierarchy import Hierarchy, SharedComponent
    n_data = 5
    A_f, b_f, data, noise_var = linear_problem(d, n_data, 7)
    z, _ = oracle.rng_fill(99, 0, d + 1, 0, 1, d, uniforms=False)
    A_c = A_f + 0.02 * z[0][:d] / np.sqrt(d)          # a surrogate close enough to be useful, different enough to reject
    b_c = b_f + 0.05 * z[0][d]
    prior_mean, prior_var = np.zeros(d), 5.0
    prior = Gaussian(ParameterVector(prior_mean.copy()), IIDCovarianceMatrix(d, prior_var))
    noise = CentredGaussianNoise(IIDCovarianceMatrix(d, noise_var))
    lik_c = AdditiveGaussianNoiseLikelihood(Data(data), ForwardModel(ExampleLinearModelSolver(A_c, b_c)), noise)
    lik_f = AdditiveGaussianNoiseLikelihood(Data(data), ForwardModel(ExampleLinearModelSolver(A_f, b_f)), noise)
    model = BayesianRegressionModelHierarchy(Hierarchy([lik_c, lik_f]), SharedComponent(prior, 2))
    cov = IIDCovarianceMatrix(d, prop_var)
    traj = np.empty((n, K, d))
    acc = np.empty((n - 1, K), dtype=np.uint8)
    sub = np.empty((n - 1, S, K), dtype=np.uint8)
    lp = np.empty((n, K))
    for k in range(K):
        b = MLDABuilder()
        b.baseProposalCovariance = cov
        b.subChainLengths = [S]
        b.bayesModel = model
        b.surrogateDiagnostics = [AcceptanceRateDiagnostics()]
        mc = b.build_method()
        rp = Replay(seed, k, d, (n - 1) * (S + 1), sub_chain_length=S)
        rp.install()
        mark_fine_level(mc, rp)
        mc.run(n, ParameterVector(np.zeros(d)), verbose=False)
        traj[:, k, :] = np.array(mc.chain.trajectory)
        acc[:, k] = mc.diagnostics._decisions
        sub[:, :, k] = np.array(mc.surrogate(0).diagnostics._decisions, dtype=np.uint8).reshape(n - 1, S)
        lp[:, k] = eval_log_along(mc.target, traj[:, k, :], lambda s_: ParameterVector(np.array(s_)))
    np.savez_compressed(
        os.path.join(HERE, name), seed=seed, n_states=n, S=S, A_fine=A_f, b_fine=b_f,
        A_coarse=A_c, b_coarse=b_c, data=data, noise_precision=np.reciprocal(np.full(d, noise_var)),
        prior_mean=prior_mean, prior_precision=np.reciprocal(np.full(d, prior_var)), sigma=sigma_of(cov),
        x0=np.zeros((K, d)), traj_head=traj[:, :2, :], traj_sha256=sha(traj), acc=acc, sub_acc=sub, logpost=lp)
    print(name, acc.mean(), sub.mean())


def fixture_f16(name="f16_aem_linear_d128.npz", use_heuristic=True, seed=20250116, min_data=6, K=8, n=161, S=4, d=128, n_data=5,
                prop_var=1.1e-3):
    """The adaptive error model over two d = 128 linear models (the matrix-core kernel's size): AEMBuilder
    (aem.py:61-82) with AEMLikelihoods on both levels, fresh likelihood objects per chain; chains, decisions of
    both levels, and what every chain's coarse likelihood learned (error count, mean, variance, inflated
    precision)."""
    from yagremcmc.chain.method.aem import AEMBuilder
    from yagremcmc.chain.target import UnnormalisedPosterior
    from yagremcmc.statistics.likelihood import AEMLikelihood
    from yagremcmc.statistics.modelHierarchy import BayesianRegressionModelHierarchy
    from yagremcmc.utility.hierarchy import Hierarchy, SharedComponent
    A_f, b_f, data, noise_var = linear_problem(d, n_data, 7)
    z, _ = oracle.rng_fill(99, 0, d + 1, 0, 1, d, uniforms=False)
    A_c = A_f + 0.02 * z[0][:d] / np.sqrt(d)
    b_c = b_f + 0.05 * z[0][d]
    prior_mean, prior_var = np.zeros(d), 5.0
    cov = IIDCovarianceMatrix(d, prop_var)
    traj = np.empty((n, K, d))
    acc = np.empty((n - 1, K), dtype=np.uint8)
    sub_acc = np.empty((n - 1, S, K), dtype=np.uint8)
    lp = np.empty((n, K))
    err_n = np.zeros(K, dtype=np.uint64)
    err_mean, err_mvar, aem_prec = np.zeros((K, d)), np.zeros((K, d)), np.zeros((K, d))
    hits = np.zeros(K, dtype=np.int64)
    for k in range(K):
        prior = Gaussian(ParameterVector(prior_mean.copy()), IIDCovarianceMatrix(d, prior_var))
        noise = CentredGaussianNoise(IIDCovarianceMatrix(d, noise_var))
        lik_c = AEMLikelihood(Data(data), ForwardModel(ExampleLinearModelSolver(A_c, b_c)), noise, min_data, use_heuristic)
        lik_f = AEMLikelihood(Data(data), ForwardModel(ExampleLinearModelSolver(A_f, b_f)), noise, min_data, use_heuristic)
        model = BayesianRegressionModelHierarchy(Hierarchy([lik_c, lik_f]), SharedComponent(prior, 2))
        b = AEMBuilder()
        b.baseProposalCovariance = cov
        b.subChainLengths = [S]
        b.bayesModel = model
        b.surrogateDiagnostics = [AcceptanceRateDiagnostics()]
        mc = b.build_method()
        rp = Replay(seed, k, d, (n - 1) * (S + 1), sub_chain_length=S)
        rp.install()
        mark_fine_level(mc, rp)
        mc.run(n, ParameterVector(np.zeros(d)), verbose=False)
        traj[:, k, :] = np.array(mc.chain.trajectory)
        acc[:, k] = mc.diagnostics._decisions
        sub_acc[:, :, k] = np.array(mc.surrogate(0).diagnostics._decisions, dtype=np.uint8).reshape(n - 1, S)
        plain = AdditiveGaussianNoiseLikelihood(Data(data), ForwardModel(ExampleLinearModelSolver(A_f, b_f)), noise)
        lp[:, k] = eval_log_along(UnnormalisedPosterior(plain, prior), traj[:, k, :], lambda s_: ParameterVector(np.array(s_)))
        err_n[k] = lik_c.accumulator.nData
        assert lik_f.accumulator.nData == 0
        if err_n[k] >= 1:
            err_mean[k] = lik_c.accumulator.mean()
        if err_n[k] >= 2:
            err_mvar[k] = lik_c.accumulator.marginal_variance()
        if lik_c._noiseModel._aemNoise is not None:
            aem_prec[k] = lik_c._noiseModel._aemNoise.covariance._precision
        hits[k] = lik_c._cache.hits
    np.savez_compressed(
        os.path.join(HERE, name), seed=seed, n_states=n, S=S, min_data_size=min_data, use_noise_heuristic=int(use_heuristic),
        A_fine=A_f, b_fine=b_f, A_coarse=A_c, b_coarse=b_c, data=data, noise_precision=np.reciprocal(np.full(d, noise_var)),
        prior_mean=prior_mean, prior_precision=np.reciprocal(np.full(d, prior_var)), sigma=sigma_of(cov), x0=np.zeros((K, d)),
        traj_head=traj[:, :2, :], traj_sha256=sha(traj), acc=acc, sub_acc=sub_acc, logpost=lp, err_n=err_n, err_mean=err_mean,
        err_mvar=err_mvar, aem_prec=aem_prec)
    print(name, acc.mean(), sub_acc.mean(), "accepted per chain", err_n.min(), err_n.max(), "cache hits", hits.mean())


def fixture_f17():
    """Random walk with a DENSE proposal covariance at d = 16 (the matrix-core kernel's path): the reference forms
    L @ xi with BLAS' dgemv (unspecified order), the ensemble as k-ordered fma chains: decisions equal, states to
    rounding."""
    d, n_data, seed, K, n = 16, 3, 20250117, 8, 300
    A, b, data, noise_var = linear_problem(d, n_data, 17)
    prior_mean, prior_var = np.zeros(d), 5.0
    model = build_linear_reference(A, b, data, noise_var, prior_mean, prior_var)
    z, _ = oracle.rng_fill(1717, 0, d, 0, 1, d, uniforms=False)
    B = z[0] * 0.05
    prop = B @ B.T + 0.02 * np.eye(d)
    cov = DenseCovarianceMatrix(prop.copy())
    traj, acc, lp = run_mrw_model(model, seed, K, n, cov, np.zeros(d))
    np.savez_compressed(
        os.path.join(HERE, "f17_mrw_dense_proposal_d16.npz"), seed=seed, n_states=n, A=A, b=b, data=data,
        noise_precision=np.reciprocal(np.full(d, noise_var)), prior_mean=prior_mean,
        prior_precision=np.reciprocal(np.full(d, prior_var)), proposal_cov=prop, chol=np.array(cov.cholFactor_),
        x0=np.zeros((K, d)), traj=traj, acc=acc, logpost=lp)
    print("F17", acc.mean())


def fixture_f15():
    """Dense covariances as NOISE and PRIOR models (CentredGaussianNoise / Gaussian over DenseCovarianceMatrix,
    covariance.py:69-94, noise.py:21-22, gaussian.py:19-24): (a) a linear model d = 3, 4 outputs, 5 data rows,
    dense 4 x 4 noise and dense 3 x 3 prior covariance; (b) Lotka-Volterra with a dense 2 x 2 noise covariance
    (reference loop over the RK4 plug-in).  The reference solves two triangular systems per norm with LAPACK;
    the fixture pins decisions, chains and log-posteriors (the latter to rounding)."""
    z, _ = oracle.rng_fill(1515, 0, 40, 0, 1, 4, uniforms=False)
    z = z[0]
    d, m, nd = 3, 4, 5
    A, b = z[:4, :3] * 0.8, z[4, :4]
    theta_true = z[5, :3]
    Bn = z[6:10, :4] * 0.4
    noise_cov = Bn @ Bn.T + 0.3 * np.eye(m)
    Bp = z[10:13, :3] * 0.7
    prior_cov = Bp @ Bp.T + 1.5 * np.eye(d)
    Ln = np.linalg.cholesky(noise_cov)
    data = np.array([A @ theta_true + b + Ln @ z[13 + i, :4] for i in range(nd)])
    prior_mean = z[20, :3] * 0.3
    like = AdditiveGaussianNoiseLikelihood(Data(data), ForwardModel(ExampleLinearModelSolver(A, b)),
                                           CentredGaussianNoise(DenseCovarianceMatrix(noise_cov)))
    prior = Gaussian(ParameterVector(prior_mean.copy()), DenseCovarianceMatrix(prior_cov))
    model = BayesianRegressionModel(like, prior)
    seed, K, n = 20250115, 16, 400
    prop_cov = DiagonalCovarianceMatrix(np.array([0.12, 0.2, 0.15]))
    traj, acc, lp = run_mrw_model(model, seed, K, n, prop_cov, np.zeros(d))
    out = dict(seed=seed, n_states=n, A=A, b=b, data=data, noise_cov=noise_cov, prior_cov=prior_cov, prior_mean=prior_mean,
               sigma=sigma_of(prop_cov), x0=np.zeros((K, d)), traj=traj, acc=acc, logpost=lp)
    # (b) Lotka-Volterra, dense 2 x 2 noise
    cfg = {'T': 6., 'alpha': 0.8, 'gamma': 0.4, 'nData': 5, 'dataDim': 2}
    design = np.array([[0.1, 0.9], [0.5, 0.5], [1., 0.5], [0.5, 1.], [2.5, 1.5]])
    lv_noise_cov = np.array([[0.05, 0.02], [0.02, 0.08]])
    pv, pm = 0.02, np.zeros(2)
    base = oracle.posterior_lv(design, cfg['T'], cfg['alpha'], cfg['gamma'], 50, np.zeros((5, 2)), np.ones(2), pm,
                               np.reciprocal(np.full(2, pv)))
    clean = oracle.forward(base, np.log(np.array([0.4, 0.6]))).reshape(5, 2)
    lv_data = clean + (np.linalg.cholesky(lv_noise_cov) @ z[21:26, :2].T).T
    problem = dict(base, data=lv_data)
    lv_like = AdditiveGaussianNoiseLikelihood(Data(lv_data), ForwardModel(RK4LotkaVolterraSolver(problem)),
                                              CentredGaussianNoise(DenseCovarianceMatrix(lv_noise_cov)))
    lv_prior = Gaussian(LotkaVolterraParameter.from_coefficient(pm.copy()), IIDCovarianceMatrix(2, pv))
    lv_prop = DiagonalCovarianceMatrix(np.array([0.02, 0.01]))
    Kl, nl = 8, 250
    ltraj, lacc, llp = run_mrw_model(BayesianRegressionModel(lv_like, lv_prior), seed + 1, Kl, nl, lv_prop,
                                     np.array([-0.6, -0.3]), make_param=LotkaVolterraParameter.from_coefficient)
    out.update(lv_seed=seed + 1, lv_n_states=nl, lv_design=design, lv_t_final=cfg['T'], lv_alpha=cfg['alpha'],
               lv_gamma=cfg['gamma'], lv_rk4_steps=50, lv_data=lv_data, lv_noise_cov=lv_noise_cov, lv_prior_mean=pm,
               lv_prior_precision=np.reciprocal(np.full(2, pv)), lv_sigma=sigma_of(lv_prop),
               lv_x0=np.tile([-0.6, -0.3], (Kl, 1)), lv_traj=ltraj, lv_acc=lacc, lv_logpost=llp)
    np.savez_compressed(os.path.join(HERE, "f15_dense_noise_prior.npz"), **out)
    print("F15", acc.mean(), lacc.mean())


def fixture_f19():
    """A dense NOISE covariance beyond the lane kernels' sizes (d = 24 parameters, 24 outputs, 5 data rows;
    CentredGaussianNoise over DenseCovarianceMatrix, covariance.py:69-94, noise.py:21-22): (a) the random walk,
    (b) two-level delayed acceptance over a hierarchy of two such models (mlda.py:315-335).  The reference
    solves two triangular systems per norm with LAPACK; the ensemble whitens the problem once (ymc_lower_times).
    Pinned: decisions, chains, log-posteriors (to rounding)."""
    from yagremcmc.statistics.modelHierarchy import BayesianRegressionModelHierarchy
    from yagremcmc.utility.hierarchy import Hierarchy, SharedComponent
    d, n_data = 24, 5
    A, b, _, _ = linear_problem(d, n_data, 19)
    z, _ = oracle.rng_fill(1919, 0, 3 * d + 8, 0, 1, d, uniforms=False)
    z = z[0]
    Bn = z[:d] * 0.12
    noise_cov = Bn @ Bn.T + 0.25 * np.eye(d)
    Ln = np.linalg.cholesky(noise_cov)
    theta_true = z[d]
    data = np.array([A @ theta_true + b + Ln @ z[d + 1 + i] for i in range(n_data)])
    A_c = A + 0.03 * z[d + 8:2 * d + 8] / np.sqrt(d)
    b_c = b + 0.05 * z[2 * d + 8]
    prior_mean, prior_var, prop_var = np.zeros(d), 5.0, 4e-3
    prior = Gaussian(ParameterVector(prior_mean.copy()), IIDCovarianceMatrix(d, prior_var))
    noise = CentredGaussianNoise(DenseCovarianceMatrix(noise_cov))
    lik_f = AdditiveGaussianNoiseLikelihood(Data(data), ForwardModel(ExampleLinearModelSolver(A, b)), noise)
    lik_c = AdditiveGaussianNoiseLikelihood(Data(data), ForwardModel(ExampleLinearModelSolver(A_c, b_c)), noise)
    cov = IIDCovarianceMatrix(d, prop_var)
    seed, K, n = 20250119, 8, 301
    traj, acc, lp = run_mrw_model(BayesianRegressionModel(lik_f, prior), seed, K, n, cov, np.zeros(d))
    out = dict(seed=seed, n_states=n, A=A, b=b, A_coarse=A_c, b_coarse=b_c, data=data, noise_cov=noise_cov,
               prior_mean=prior_mean, prior_precision=np.reciprocal(np.full(d, prior_var)), sigma=sigma_of(cov),
               x0=np.zeros((K, d)), traj=traj, acc=acc, logpost=lp)
    # (b) two levels
    model = BayesianRegressionModelHierarchy(Hierarchy([lik_c, lik_f]), SharedComponent(prior, 2))
    K2, n2, S = 6, 151, 3
    traj2 = np.empty((n2, K2, d))
    acc2 = np.empty((n2 - 1, K2), dtype=np.uint8)
    sub2 = np.empty((n2 - 1, S, K2), dtype=np.uint8)
    lp2 = np.empty((n2, K2))
    for k in range(K2):
        bld = MLDABuilder()
        bld.baseProposalCovariance = cov
        bld.subChainLengths = [S]
        bld.bayesModel = model
        bld.surrogateDiagnostics = [AcceptanceRateDiagnostics()]
        mc = bld.build_method()
        rp = Replay(seed + 1, k, d, (n2 - 1) * (S + 1), sub_chain_length=S)
        rp.install()
        mark_fine_level(mc, rp)
        mc.run(n2, ParameterVector(np.zeros(d)), verbose=False)
        traj2[:, k, :] = np.array(mc.chain.trajectory)
        acc2[:, k] = mc.diagnostics._decisions
        sub2[:, :, k] = np.array(mc.surrogate(0).diagnostics._decisions, dtype=np.uint8).reshape(n2 - 1, S)
        lp2[:, k] = eval_log_along(mc.target, traj2[:, k, :], lambda s_: ParameterVector(np.array(s_)))
    out.update(ml_seed=seed + 1, ml_n_states=n2, ml_S=S, ml_x0=np.zeros((K2, d)), ml_traj=traj2, ml_acc=acc2,
               ml_sub_acc=sub2, ml_logpost=lp2)
    np.savez_compressed(os.path.join(HERE, "f19_dense_noise_d24.npz"), **out)
    print("F19", acc.mean(), acc2.mean(), sub2.mean())


def fixture_f20(name="f20_dense_prior_noise_d20.npz", d=20, n_data=4, seed=20250120, K=8, n=301, prop_var=6e-3):
    """Dense PRIOR and dense NOISE covariance beyond the lane kernels' sizes (d = 20 parameters, 20 outputs, 4
    data rows; Gaussian / CentredGaussianNoise over DenseCovarianceMatrix, gaussian.py:19-24, noise.py:21-22,
    covariance.py:69-94), prior mean away from zero: a random walk.  Decisions, chains, log-posteriors.
    F20b: d = 128 (the prior's 128 more outputs take the forward map beyond the LDS: streamed)."""
    A, b, _, _ = linear_problem(d, n_data, 20)
    z, _ = oracle.rng_fill(2020, 0, 3 * d + 8, 0, 1, d, uniforms=False)
    z = z[0]
    Bn = z[:d] * 0.12
    noise_cov = Bn @ Bn.T + 0.25 * np.eye(d)
    Bp = z[d:2 * d] * 0.5
    prior_cov = Bp @ Bp.T + 2.0 * np.eye(d)
    prior_mean = 0.4 * z[2 * d]
    theta_true = z[2 * d + 1]
    data = np.array([A @ theta_true + b + np.linalg.cholesky(noise_cov) @ z[2 * d + 2 + i] for i in range(n_data)])
    prior = Gaussian(ParameterVector(prior_mean.copy()), DenseCovarianceMatrix(prior_cov))
    noise = CentredGaussianNoise(DenseCovarianceMatrix(noise_cov))
    lik = AdditiveGaussianNoiseLikelihood(Data(data), ForwardModel(ExampleLinearModelSolver(A, b)), noise)
    cov = IIDCovarianceMatrix(d, prop_var)
    traj, acc, lp = run_mrw_model(BayesianRegressionModel(lik, prior), seed, K, n, cov, np.zeros(d))
    np.savez_compressed(os.path.join(HERE, name), seed=seed, n_states=n, A=A, b=b, data=data,
                        noise_cov=noise_cov, prior_cov=prior_cov, prior_mean=prior_mean, sigma=sigma_of(cov),
                        x0=np.zeros((K, d)), traj=traj, acc=acc, logpost=lp)
    print(name, acc.mean())


def fixture_f21():
    """Preconditioned Crank-Nicolson (pcn.py:9-57) on a d = 64 linear model with a centred prior -- the size of
    the matrix-core kernel's pCN proposal: 8 chains x 301 states."""
    from yagremcmc.chain.method.pcn import PCNBuilder
    d, n_data = 64, 5
    A, b, data, noise_var = linear_problem(d, n_data, 21)
    prior_var = np.linspace(0.5, 2.0, d)
    prior_cov = DiagonalCovarianceMatrix(prior_var.copy())
    prior = Gaussian(ParameterVector(np.zeros(d)), prior_cov)
    like = AdditiveGaussianNoiseLikelihood(Data(data), ForwardModel(ExampleLinearModelSolver(A, b)),
                                           CentredGaussianNoise(IIDCovarianceMatrix(d, noise_var)))
    model = BayesianRegressionModel(like, prior)
    seed, K, n, h = 20250121, 8, 301, 0.004
    traj = np.empty((n, K, d))
    acc = np.empty((n - 1, K), dtype=np.uint8)
    lp = np.empty((n, K))
    for k in range(K):
        bld = PCNBuilder()
        bld.stepSize = h
        bld.bayesModel = model
        mc = bld.build_method()
        rp = Replay(seed, k, d, n - 1)
        rp.install()
        mc.run(n, ParameterVector(np.zeros(d)), verbose=False)
        traj[:, k, :] = np.array(mc.chain.trajectory)
        acc[:, k] = mc.diagnostics._decisions
        lp[:, k] = eval_log_along(mc.target, traj[:, k, :], lambda s_: ParameterVector(np.array(s_)))
    np.savez_compressed(
        os.path.join(HERE, "f21_pcn_linear_d64.npz"), seed=seed, n_states=n, step_size=h, A=A, b=b, data=data,
        noise_precision=np.reciprocal(np.full(d, noise_var)), prior_mean=np.zeros(d), prior_precision=prior_cov._precision,
        prior_sigma=sigma_of(prior_cov), x0=np.zeros((K, d)), traj=traj, acc=acc, loglik=lp)
    print("F21", acc.mean())


def fixture_f18(name="f18_mlda_nested2_linear_d128.npz", d=128, n_models=3, seed=20250118, K=6, n=101,
                sub_lengths=(3, 4), perturb=0.02, prop_var=1.1e-3):
    """Delayed acceptance over a hierarchy of THREE (and, second fixture, FIVE) linear models at large dimension
    (mlda.py:12-73,100-116,315-335; modelHierarchy.py:6-53): an even number of surrogates, executed by the
    reference as a nest of n/2 + 1 effective levels (include/ymc.h, YMC_METHOD_MLDA_NESTED).  The coarser a
    model, the larger its perturbation of the finest one's matrix and shift."""
    from yagremcmc.statistics.modelHierarchy import BayesianRegressionModelHierarchy
    from yagremcmc.utility.hierarchy import Hierarchy, SharedComponent
    n_data = 5
    A_f, b_f, data, noise_var = linear_problem(d, n_data, 11)
    ns = n_models - 1
    assert ns % 2 == 0 and len(sub_lengths) == ns
    As, bs = [], []
    for i in range(ns):
        z, _ = oracle.rng_fill(1234 + i, 0, d + 1, 0, 1, d, uniforms=False)
        scale = perturb * (ns - i)
        As.append(A_f + scale * z[0][:d] / np.sqrt(d))
        bs.append(b_f + 2.5 * scale * z[0][d])
    As.append(A_f)
    bs.append(b_f)
    prior_mean, prior_var = np.zeros(d), 5.0
    prior = Gaussian(ParameterVector(prior_mean.copy()), IIDCovarianceMatrix(d, prior_var))
    noise = CentredGaussianNoise(IIDCovarianceMatrix(d, noise_var))
    liks = [AdditiveGaussianNoiseLikelihood(Data(data), ForwardModel(ExampleLinearModelSolver(A, b)), noise)
            for A, b in zip(As, bs)]
    model = BayesianRegressionModelHierarchy(Hierarchy(liks), SharedComponent(prior, n_models))
    E = ns // 2 + 1
    steps = [sub_lengths[2 * e + 1] for e in range(E - 1)]
    cost = 1
    for s_ in steps:
        cost = s_ * cost + 1
    cov = IIDCovarianceMatrix(d, prop_var)
    traj = np.empty((n, K, d))
    acc = np.empty((n - 1, K), dtype=np.uint8)
    lp = np.empty((n, K))
    level_acc = np.zeros((E - 1, K), dtype=np.uint64)
    for k in range(K):
        b = MLDABuilder()
        b.baseProposalCovariance = cov
        b.subChainLengths = list(sub_lengths)
        b.bayesModel = model
        b.surrogateDiagnostics = [AcceptanceRateDiagnostics() for _ in range(ns)]
        mc = b.build_method()
        rp = SequentialReplay(seed, k, d, (n - 1) * cost)
        rp.install()
        rp.watch(mc)
        for i in range(ns):
            rp.watch(mc.surrogate(i))
        mc.run(n, ParameterVector(np.zeros(d)), verbose=False)
        assert rp.t == (n - 1) * cost, (rp.t, (n - 1) * cost)
        traj[:, k, :] = np.array(mc.chain.trajectory)
        acc[:, k] = mc.diagnostics._decisions
        lp[:, k] = eval_log_along(mc.target, traj[:, k, :], lambda s_: ParameterVector(np.array(s_)))
        for e in range(E - 1):
            level_acc[e, k] = int(np.sum(mc.surrogate(2 * e).diagnostics._decisions))
        for i in range(1, ns, 2):   # the levels whose accept step never runs
            assert len(mc.surrogate(i).diagnostics._decisions) == 0
    np.savez_compressed(
        os.path.join(HERE, name), seed=seed, n_states=n, sub_lengths=np.array(sub_lengths), A=np.array(As), b=np.array(bs),
        data=data, noise_precision=np.reciprocal(np.full(d, noise_var)), prior_mean=prior_mean,
        prior_precision=np.reciprocal(np.full(d, prior_var)), sigma=sigma_of(cov), x0=np.zeros((K, d)),
        traj_head=traj[:, :2, :], traj_sha256=sha(traj), acc=acc, logpost=lp, level_acc=level_acc, cost_top=cost)
    print(name, acc.mean(), level_acc.sum(axis=1))


def main():
    which = sys.argv[1:] or ["f1", "f2", "f2b", "f3", "f3b", "f4", "f5", "f7", "f8", "f9", "f10", "f11", "f12", "f13",
                             "f14", "f15", "f16", "f17", "f18", "f19", "f20", "f21", "f22", "f23", "f24"]
    if "f1" in which:
        fixture_f1()
    if "f2" in which:
        run_mrw_2d("f2_mrw_2d.npz", 20250102, 64, 2000, IIDCovarianceMatrix(2, 1.0),
                   np.array([-8.0, -7.0]))
    if "f2b" in which:
        run_mrw_2d("f2b_mrw_2d_indep.npz", 20250112, 8, 500,
                   DiagonalCovarianceMatrix(np.array([TGT_COV[0, 0], TGT_COV[1, 1]])),
                   np.array([0.5, 0.25]))
    if "f3" in which:
        fixture_f3()
    if "f3b" in which:
        fixture_f3b()
    if "f22" in which:
        fixture_f22()
        fixture_f22("f22b_linear_d320_m48.npz", d=320, m=48, seed=20250130, K=4, n=120, prop_var=5e-4)
    if "f4" in which:
        fixture_f4()
    if "f5" in which:
        fixture_f5()
    if "f7" in which:
        fixture_f7()
    if "f8" in which:
        fixture_f8()
    if "f9" in which:
        fixture_f9()
    if "f10" in which:
        fixture_f10()
    if "f11" in which:
        fixture_f11()
    if "f13" in which:
        fixture_f13()
    if "f14" in which:
        fixture_f14()
        fixture_f14("f14b_mlda2_linear_d192.npz", d=192, seed=20250126, K=4, n=81, S=3, prop_var=7e-4)
        fixture_f14("f14c_mlda2_linear_d300.npz", d=300, seed=20250131, K=4, n=41, S=3, prop_var=4e-4)
    if "f15" in which:
        fixture_f15()
    if "f16" in which:
        fixture_f16()
        fixture_f16("f16b_aem_linear_d48_rows12.npz", seed=20250128, min_data=5, K=6, n=121, S=3, d=48, n_data=12, prop_var=4e-3)
    if "f17" in which:
        fixture_f17()
    if "f19" in which:
        fixture_f19()
    if "f20" in which:
        fixture_f20()
        fixture_f20("f20b_dense_prior_noise_d128.npz", d=128, n_data=3, seed=20250129, K=4, n=121, prop_var=1.2e-3)
    if "f21" in which:
        fixture_f21()
    if "f18" in which:
        fixture_f18()
        fixture_f18("f18b_mlda_nested4_linear_d24.npz", d=24, n_models=5, seed=20250119, K=6, n=81,
                    sub_lengths=(2, 3, 2, 2), perturb=0.03, prop_var=6e-3)
        fixture_f18("f18c_mlda_nested8_linear_d24.npz", d=24, n_models=9, seed=20250125, K=4, n=25,
                    sub_lengths=(2, 2, 3, 2, 1, 2, 2, 2), perturb=0.015, prop_var=6e-3)
        fixture_f18("f18d_mlda_nested2_linear_d192.npz", d=192, n_models=3, seed=20250127, K=4, n=41, sub_lengths=(2, 3),
                    perturb=0.02, prop_var=7e-4)
        fixture_f18("f18e_mlda_nested2_linear_d300.npz", d=300, n_models=3, seed=20250132, K=4, n=31, sub_lengths=(2, 3),
                    perturb=0.02, prop_var=4e-4)
    if "f23" in which:
        fixture_f23()
    if "f24" in which:
        fixture_f24()
    if "f12" in which:
        fixture_f12()
        fixture_f12("f12b_aem_linear_noheuristic.npz", use_heuristic=False, seed=20250113, min_data=5, K=12,
                    n=400, S=3)


if __name__ == "__main__":
    main()
