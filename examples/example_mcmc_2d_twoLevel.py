"""Two-level delayed acceptance on the 2-D Gaussian pair of the reference's
example_mcmc_2d_twoLevel.py (sub-chain length 6), as an ensemble."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from yagremcmc_amd.chain.method.mlda import MLDABuilder
from yagremcmc_amd.parameter.vector import ParameterVector
from yagremcmc_amd.statistics.covariance import IIDCovarianceMatrix
from yagremcmc_amd.test.testSetup import GaussianTargetDensity2d

nChains = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nSteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5000

tgtMeanCoeff = np.array([1., 1.5])
tgtCov = np.array([[2.4, -0.5], [-0.5, 0.7]])
surrCov = 2. * tgtCov + np.array([[0.1, 1.2], [1.2, 0.05]])
tgtDensity = GaussianTargetDensity2d(ParameterVector(tgtMeanCoeff), tgtCov)
surrDensity = GaussianTargetDensity2d(ParameterVector(tgtMeanCoeff + np.array([0.15, -0.2])), surrCov)

chainBuilder = MLDABuilder()
chainBuilder.explicitTarget = tgtDensity
chainBuilder.surrogateTargets = [surrDensity]
chainBuilder.baseProposalCovariance = IIDCovarianceMatrix(2, 1.)
chainBuilder.subChainLengths = [6]
chainBuilder.nChains = nChains
chainBuilder.seed = 7

mcmc = chainBuilder.build_method()
mcmc.run(nSteps, ParameterVector(np.array([-8., -7.])))
states = mcmc.chain.ensemble_trajectory

# the reference's post-processing (example_mcmc_2d_twoLevel.py:44-66): IATs per chain on the device, thinning by 'max'
burnin = 100
meanIAT = mcmc.integrated_autocorrelation('mean', burnin)
maxIAT = mcmc.integrated_autocorrelation('max', burnin)
thinningStep = max(int(np.median(maxIAT)), 1)
mcmcSamples = states[burnin::thinningStep].reshape(-1, 2)
print("\nAnalytics")
print("---------")
print(f"acceptance rate (fine level, pooled): {mcmc.diagnostics.global_acceptance_rate():.4f}")
print(f"mean IAT (median over chains): {int(np.median(meanIAT))}   max IAT: {int(np.median(maxIAT))}")
print("\nInference")
print("---------")
print(f"true mean: {tgtMeanCoeff}")
print(f"mean state: {states.reshape(-1, 2).mean(axis=0)}")
print(f"mean estimate (thinned by {thinningStep}): {mcmcSamples.mean(axis=0)}")
