"""1-D Gaussian target, one chain -- the reference's example_mcmc_1d.py through
the device path (ScalarParameter, MetropolisedRandomWalk built directly)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from yagremcmc_amd.chain.diagnostics import FullDiagnostics
from yagremcmc_amd.chain.method.mrw import MetropolisedRandomWalk
from yagremcmc_amd.parameter.scalar import ScalarParameter
from yagremcmc_amd.statistics.covariance import IIDCovarianceMatrix
from yagremcmc_amd.test.testSetup import GaussianTargetDensity1d

tgtMean, tgtVar = ScalarParameter(np.array([1.5])), 1.
mcmc = MetropolisedRandomWalk(GaussianTargetDensity1d(tgtMean, tgtVar), IIDCovarianceMatrix(1, 1.5),
                              FullDiagnostics(), seed=1)
nSteps = int(1e5)
mcmc.run(nSteps, ScalarParameter(np.array([-3.])))
states = np.array(mcmc.chain.trajectory)          # (nSteps, 1), as in the reference
burnin = int(0.02 * nSteps)
print(f"acceptance rate: {mcmc.diagnostics.global_acceptance_rate()}")
print(f"chain sample variance: {mcmc.diagnostics.marginal_variance()}")
print(f"mean state: {mcmc.diagnostics.mean()}")
print(f"true mean: {tgtMean.coefficient[0]}  mean estimate: {states[burnin:].mean()}")
print(f"true variance: {tgtVar}  variance estimate: {states[burnin:].var()}")
