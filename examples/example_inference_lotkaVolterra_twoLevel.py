"""Two-level Lotka-Volterra inference (the reference's example_inference_lotkaVolterra_twoLevel.py: a
coarse RK23 / rtol 1e-2 model under a DOP853 / rtol 1e-5 target, sub-chain length 3).  On the device both
levels are fixed-step RK4; the step counts follow from the levels' rtol (16 and 89 steps,
lowering.rk4_steps_for_rtol) or are given explicitly as 'rk4Steps' -- levels that lower to the same
density would make delayed acceptance pointless and are reported with a RuntimeWarning.

    python examples/example_inference_lotkaVolterra_twoLevel.py [nChains] [nSteps] [adaptive]

With `adaptive` both levels integrate adaptively at their own rtol (as the reference's levels do), and the chains start at
the reference's starting point (-7, 2.8)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import yagremcmc_amd.test.testSetup as setup
from yagremcmc_amd.chain.method.mlda import MLDABuilder
from yagremcmc_amd.model.forwardModel import ForwardModel
from yagremcmc_amd.statistics.covariance import IIDCovarianceMatrix
from yagremcmc_amd.statistics.gaussian import Gaussian
from yagremcmc_amd.statistics.likelihood import AdditiveGaussianNoiseLikelihood
from yagremcmc_amd.statistics.modelHierarchy import BayesianRegressionModelHierarchy
from yagremcmc_amd.statistics.noise import CentredGaussianNoise
from yagremcmc_amd.utility.hierarchy import Hierarchy, SharedComponent

nChains = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nSteps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
adaptive = len(sys.argv) > 3 and sys.argv[3] == "adaptive"
np.random.seed(1112)

base = {'T': 10., 'alpha': 0.8, 'gamma': 0.4, 'nData': 10, 'dataDim': 2, 'adaptive': adaptive}
surrogateConfig = dict(base, solver='RK23', rtol=1e-2)
targetConfig = dict(base, solver='DOP853', rtol=1e-5)
design = np.array([np.random.uniform(0.5, 1.5, 2) for _ in range(base['nData'])])
surrogateSolver = setup.LotkaVolterraSolver(design, surrogateConfig)
targetSolver = setup.LotkaVolterraSolver(design, targetConfig)
print("adaptive integration at rtol 1e-2 / 1e-5" if adaptive else
      f"RK4 steps: surrogate {surrogateSolver.rk4Steps_}, target {targetSolver.rk4Steps_}")

groundTruth = setup.LotkaVolterraParameter.from_interpolation(np.array([0.4, 0.6]))
dataNoiseVar = 0.04
data = setup.generate_synthetic_data(groundTruth, targetSolver, dataNoiseVar)

prior = Gaussian(setup.LotkaVolterraParameter.from_coefficient(np.zeros(2)), IIDCovarianceMatrix(2, 1.4))
noise = CentredGaussianNoise(IIDCovarianceMatrix(2, dataNoiseVar))
likelihoods = Hierarchy([AdditiveGaussianNoiseLikelihood(data, ForwardModel(s), noise)
                         for s in (surrogateSolver, targetSolver)])
statModel = BayesianRegressionModelHierarchy(likelihoods, SharedComponent(prior, 2))

chainBuilder = MLDABuilder()
chainBuilder.bayesModel = statModel
chainBuilder.baseProposalCovariance = IIDCovarianceMatrix(2, 0.1)
chainBuilder.subChainLengths = [3]
chainBuilder.nChains = nChains
sampler = chainBuilder.build_method()
start = np.array([-7.0, 2.8]) if adaptive else np.array([-0.6, -0.3])
sampler.run(nSteps, setup.LotkaVolterraParameter.from_coefficient(start))

states = sampler.chain.ensemble_trajectory
post = states[nSteps // 4:].reshape(-1, 2)
print(f"acceptance rate (fine level): {sampler.diagnostics.global_acceptance_rate():.4f}")
print(f"true parameter (log beta, log delta): {groundTruth.coefficient}")
print(f"posterior mean: {post.mean(axis=0)}  -> (beta, delta) = {np.exp(post.mean(axis=0))}")
print(f"chains flagged non-finite: {int((sampler.chain_status != 0).sum())}")
