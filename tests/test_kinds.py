"""The extension point for new forward models (INTEGRATION.md "Adding a forward model"), CPU side:
the registration table, the worked kind's checker pinned against the closed-form solution, the Python registry."""
import os
import re

import numpy as np
import pytest

from oracle import oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem(rng, n=6, steps=40):
    x0 = rng.uniform(0.3, 2.5, size=n)
    T = 3.0
    truth = np.log(np.array([0.8, 5.0]))
    r, k = np.exp(truth)
    clean = k / (1.0 + (k / x0 - 1.0) * np.exp(-r * T))
    data = (clean + 0.05 * rng.standard_normal(n)).reshape(n, 1)
    return dict(design=x0, t_final=T, rk4_steps=steps, data=data, noise_precision=np.array([400.0]),
                prior_mean=np.array([0.0, 1.0]), prior_precision=np.array([0.5, 0.5])), truth


def test_registration_table_header_and_bindings_agree():
    """every line of csrc/ymc_kinds.def names an enum value of include/ymc.h, a header under csrc/kinds/ that ymc_kinds.h
    includes, and the constant the Python binding and the checker use"""
    from yagremcmc_amd import _ffi
    text = open(os.path.join(ROOT, "yagre-mcmc_amd", "csrc", "ymc_kinds.def")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)      # the comment shows the macro's shape
    lines = re.findall(r"^YMC_EXT_KIND\((\w+),\s*(\w+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+)\)", text, flags=re.M)
    assert lines, "the table has at least the worked example"
    header = open(os.path.join(ROOT, "include", "ymc.h")).read()
    kinds_h = open(os.path.join(ROOT, "yagre-mcmc_amd", "csrc", "ymc_kinds.h")).read()
    for enum, name, dmin, dmax, outputs, cols in lines:
        m = re.search(rf"{enum}\s*=\s*(\d+)", header)
        assert m, f"{enum} is not in include/ymc.h"
        assert os.path.exists(os.path.join(ROOT, "yagre-mcmc_amd", "csrc", "kinds", f"{name}.h"))
        assert f'#include "kinds/{name}.h"' in kinds_h
        assert int(dmin) <= int(dmax) <= 8 and int(outputs) >= 1 and int(cols) >= 1
        assert getattr(_ffi, enum.replace("YMC_DENSITY_", "")) == int(m.group(1))
    assert ("YMC_DENSITY_POSTERIOR_LOGISTIC", "logistic", "2", "2", "1", "1") in lines
    assert _ffi.POSTERIOR_LOGISTIC == o.POST_LOGISTIC == 5


def test_logistic_checker_against_the_closed_form():
    """the oracle's counterpart of the worked kind has no reference to be compared with: it is pinned by the ODE's
    closed-form solution (RK4 error ~ h^4) and by the posterior assembled from it in NumPy"""
    rng = np.random.default_rng(5)
    spec, truth = _problem(rng)
    p = o.posterior_logistic(**spec)
    thetas = truth + 0.3 * rng.standard_normal((40, 2))
    F = o.forward(p, thetas)
    r, k = np.exp(thetas[:, :1]), np.exp(thetas[:, 1:])
    exact = k / (1.0 + (k / spec["design"][None, :] - 1.0) * np.exp(-r * spec["t_final"]))
    assert np.abs(F - exact).max() < 5e-6
    errs = []
    for steps in (20, 40, 80):                      # fourth order: halving the step divides the error by ~16
        F = o.forward(o.posterior_logistic(**dict(spec, rk4_steps=steps)), thetas)
        errs.append(np.abs(F - exact).max())
    assert 10.0 < errs[0] / errs[1] < 22.0 and 10.0 < errs[1] / errs[2] < 22.0
    lp, status = o.logdensity(p, thetas)
    res = o.forward(p, thetas) - spec["data"].reshape(1, -1)
    want = -0.5 * (res * res * spec["noise_precision"][0]).sum(axis=1) \
        - 0.5 * (((thetas - spec["prior_mean"]) ** 2) * spec["prior_precision"]).sum(axis=1)
    assert not status.any() and np.allclose(lp, want, rtol=1e-12, atol=0)
    # a run of the checker's random walk finds the truth
    e = o.Ensemble(p, np.array([0.02, 0.015]), np.tile(truth, (64, 1)), 3)
    out = e.advance(400)
    post = out["traj"][100:].reshape(-1, 2).mean(axis=0)
    assert np.abs(post - truth).max() < 0.2 and 0.1 < out["acc"].mean() < 0.9


def test_registered_solver_is_lowered_and_strangers_are_not():
    from yagremcmc_amd import _ffi, lowering
    from yagremcmc_amd.kinds.logistic import LogisticGrowthSolver
    from yagremcmc_amd.model.forwardModel import ForwardModel
    from yagremcmc_amd.parameter.vector import ParameterVector
    from yagremcmc_amd.plugins import SolverInterface
    from yagremcmc_amd.statistics.covariance import IIDCovarianceMatrix
    from yagremcmc_amd.statistics.data import Data
    from yagremcmc_amd.statistics.gaussian import Gaussian
    from yagremcmc_amd.statistics.likelihood import AdditiveGaussianNoiseLikelihood
    from yagremcmc_amd.statistics.noise import CentredGaussianNoise
    rng = np.random.default_rng(6)
    spec, _ = _problem(rng)
    solver = LogisticGrowthSolver(spec["design"], {'T': spec["t_final"], 'rk4Steps': 40})
    noise = CentredGaussianNoise(IIDCovarianceMatrix(1, 1.0 / 400.0))
    prior = Gaussian(ParameterVector(spec["prior_mean"]), IIDCovarianceMatrix(2, 2.0))
    low = lowering.lower_posterior(AdditiveGaussianNoiseLikelihood(Data(spec["data"]), ForwardModel(solver), noise), prior)
    assert low.desc.kind == _ffi.POSTERIOR_LOGISTIC and (low.desc.dim, low.desc.n_data, low.desc.data_dim) == (2, 6, 1)
    assert low.desc.rk4_steps == 40 and low.desc.t_final == 3.0
    with pytest.raises(ValueError, match="one data row per design row"):
        lowering.lower_posterior(AdditiveGaussianNoiseLikelihood(Data(spec["data"][:4]), ForwardModel(solver), noise), prior)

    class Mine(LogisticGrowthSolver):       # a subclass is another class: registered by identity, not by name or ancestry
        pass

    class Stranger(SolverInterface):
        pass
    for bad in (Mine(spec["design"], {'T': 3.0}), Stranger()):
        with pytest.raises(lowering.UnsupportedPlugin, match="no device lowering|not by the stock class"):
            lowering.lower_solver(bad, lowering.Lowered(0, 0))
