"""Lotka-Volterra parameter inference (shapes of the reference's
example_inference_lotkaVolterra_singleLevel.py: nData 10, T 10; wiring of
yagremcmc/test/test_inference_mcmc_singleLevel.py), fixed-step RK4 x 50 on the
device, an ensemble of chains started at the test's initial state.

    python examples/example_inference_lotkaVolterra_singleLevel.py [nChains] [nSteps] [adaptive]

With a third argument `adaptive` the chains start where the reference's example starts, theta = (-7, 2.8)
(:106) -- RK4 in 50 steps overflows there -- and the device integrates adaptively at the example's rtol."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import yagremcmc_amd.test.testSetup as setup
from yagremcmc_amd.chain.method.mrw import MRWBuilder
from yagremcmc_amd.model.forwardModel import ForwardModel
from yagremcmc_amd.statistics.bayesModel import BayesianRegressionModel
from yagremcmc_amd.statistics.covariance import IIDCovarianceMatrix
from yagremcmc_amd.statistics.gaussian import Gaussian
from yagremcmc_amd.statistics.likelihood import AdditiveGaussianNoiseLikelihood
from yagremcmc_amd.statistics.noise import CentredGaussianNoise

nChains = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nSteps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
adaptive = len(sys.argv) > 3 and sys.argv[3] == "adaptive"
np.random.seed(1111)

config = {'T': 10., 'alpha': 0.8, 'gamma': 0.4, 'nData': 10, 'dataDim': 2, 'solver': 'LSODA', 'rtol': 1e-4}
config.update({'adaptive': True} if adaptive else {'rk4Steps': 50})
design = np.array([np.random.uniform(0.5, 1.5, 2) for _ in range(config['nData'])])
solver = setup.LotkaVolterraSolver(design, config)
groundTruth = setup.LotkaVolterraParameter.from_interpolation(np.array([0.4, 0.6]))
dataNoiseVar = 0.04
data = setup.generate_synthetic_data(groundTruth, solver, dataNoiseVar)   # forward solve on the device

prior = Gaussian(setup.LotkaVolterraParameter.from_coefficient(np.zeros(2)), IIDCovarianceMatrix(2, 1.4))
likelihood = AdditiveGaussianNoiseLikelihood(data, ForwardModel(solver),
                                             CentredGaussianNoise(IIDCovarianceMatrix(2, dataNoiseVar)))
chainBuilder = MRWBuilder()
chainBuilder.proposalCovariance = IIDCovarianceMatrix(2, 0.15)
chainBuilder.bayesModel = BayesianRegressionModel(likelihood, prior)
chainBuilder.nChains = nChains
sampler = chainBuilder.build_method()
start = np.array([-7.0, 2.8]) if adaptive else np.array([-0.6, -0.3])
sampler.run(nSteps, setup.LotkaVolterraParameter.from_coefficient(start))

states = sampler.chain.ensemble_trajectory
post = states[nSteps // 4:].reshape(-1, 2)
print(f"acceptance rate: {sampler.diagnostics.global_acceptance_rate():.4f}")
print(f"true parameter (log beta, log delta): {groundTruth.coefficient}")
print(f"posterior mean: {post.mean(axis=0)}  -> (beta, delta) = {np.exp(post.mean(axis=0))}")
print(f"chains flagged non-finite: {int((sampler.chain_status != 0).sum())}")
