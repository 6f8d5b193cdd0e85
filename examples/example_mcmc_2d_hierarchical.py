"""MLDA with two surrogates on the densities of the reference's
example_mcmc_2d_hierarchical.py (sub-chain lengths [4, 4]), as an ensemble.

The chains are the ones the reference produces for this setup: with an even
number of surrogates only every second level of its hierarchy takes accept
decisions (include/ymc.h, YMC_METHOD_MLDA_NESTED) -- here the base surrogate's
random walk and the fine-level correction against the fine surrogate."""
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
tgtDensity = GaussianTargetDensity2d(ParameterVector(tgtMeanCoeff), tgtCov)
baseSurrDensity = GaussianTargetDensity2d(ParameterVector(tgtMeanCoeff + np.array([-0.05, 0.01])),
                                          4. * np.array([[2.8, -0.1], [-0.1, 1.7]]))
fineSurrDensity = GaussianTargetDensity2d(ParameterVector(tgtMeanCoeff + np.array([0., -0.01])),
                                          2. * np.array([[2.4, -0.3], [-0.3, 1.1]]))

chainBuilder = MLDABuilder()
chainBuilder.explicitTarget = tgtDensity
chainBuilder.surrogateTargets = [baseSurrDensity, fineSurrDensity]
chainBuilder.baseProposalCovariance = IIDCovarianceMatrix(2, 1.)
chainBuilder.subChainLengths = [4, 4]
chainBuilder.nChains = nChains
chainBuilder.seed = 11

mcmc = chainBuilder.build_method()
mcmc.run(nSteps, ParameterVector(np.array([-8., -7.])))

# the reference's post-processing (example_mcmc_2d_hierarchical.py:53-75): IATs per chain on the device, thinning by 'max'
burnin = 100
states = mcmc.chain.ensemble_trajectory
meanIAT = mcmc.integrated_autocorrelation('mean', burnin)
maxIAT = mcmc.integrated_autocorrelation('max', burnin)
thinningStep = max(int(np.median(maxIAT)), 1)
mcmcSamples = states[burnin::thinningStep].reshape(-1, 2)
print("\nAnalytics")
print("---------")
print(f"acceptance rate (pooled): {mcmc.diagnostics.global_acceptance_rate():.4f}")
print(f"acceptance rate per level (coarsest first): {[round(r, 4) for r in mcmc.level_acceptance_rates()]}")
print(f"mean IAT (median over chains): {int(np.median(meanIAT))}   max IAT: {int(np.median(maxIAT))}")
print(f"true mean: {tgtMeanCoeff}  mean state: {states.reshape(-1, 2).mean(axis=0)}  "
      f"mean estimate (thinned by {thinningStep}): {mcmcSamples.mean(axis=0)}")
