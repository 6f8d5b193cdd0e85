"""2-D Gaussian target, Metropolised random walk -- the ensemble version of the
reference's example_mcmc_2d_singleLevel.py (same target, proposal, initial
state and diagnostics; `nChains` independent chains instead of one)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from yagremcmc_amd.chain.diagnostics import FullDiagnostics
import yagremcmc_amd.postprocessing.autocorrelation as ac
from yagremcmc_amd.chain.method.mrw import MRWBuilder
from yagremcmc_amd.parameter.vector import ParameterVector
from yagremcmc_amd.statistics.covariance import IIDCovarianceMatrix
from yagremcmc_amd.test.testSetup import GaussianTargetDensity2d

nChains = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nSteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10000

tgtMean = ParameterVector(np.array([1., 1.5]))
tgtCov = np.array([[2.4, -0.5], [-0.5, 0.7]])
tgtDensity = GaussianTargetDensity2d(tgtMean, tgtCov)

chainBuilder = MRWBuilder()
chainBuilder.explicitTarget = tgtDensity
chainBuilder.proposalCovariance = IIDCovarianceMatrix(tgtMean.dimension, 1.0)
chainBuilder.diagnostics = FullDiagnostics()
chainBuilder.nChains = nChains          # ensemble extension
chainBuilder.seed = 20250103

mcmc = chainBuilder.build_method()
mcmc.run(nSteps, ParameterVector(np.array([-8., -7.])))

states = mcmc.chain.ensemble_trajectory          # (nSteps, nChains, 2)
burnin = 1000
samples = states[burnin:].reshape(-1, 2)

# the reference's post-processing (example_mcmc_2d_singleLevel.py:62-82), per chain on the device from the record in HBM:
# the autocorrelation function (of chain 0's components, as the reference plots it), both IATs, thinning by the 'max' one
acf = [ac.estimate_autocorrelation_function_1d(states[burnin:, 0, j]) for j in range(tgtMean.dimension)]
meanIAT = mcmc.integrated_autocorrelation('mean', burnin=burnin)
maxIAT = mcmc.integrated_autocorrelation('max', burnin=burnin)
thinningStep = max(int(np.median(maxIAT)), 1)
mcmcSamples = states[burnin::thinningStep].reshape(-1, 2)
meanState = states.reshape(-1, 2).mean(axis=0)
meanEst = mcmcSamples.mean(axis=0)

print("\nAnalytics")
print("---------")
print(f"chains: {nChains}, states per chain: {nSteps}")
print(f"acceptance rate (pooled): {mcmc.diagnostics.global_acceptance_rate():.4f}")
print(f"kernel time per launch: {mcmc.kernel_time_ms()[0]:.3f} ms")
print(f"mean IAT (median over chains): {np.median(meanIAT):.0f}   max IAT (median over chains): {np.median(maxIAT):.0f}")
print("\nInference")
print("---------")
print(f"true mean: {tgtMean.coefficient}")
print(f"mean state (Welford, pooled): {mcmc.diagnostics.mean()}")
print(f"mean state: {meanState}")
print(f"mean estimate (burn-in {burnin}, thinned by {thinningStep}): {meanEst}")
print(f"autocorrelation of chain 0 at lag {thinningStep}: {[round(float(a[min(thinningStep, a.size - 1)]), 3) for a in acf]}")
print(f"mean estimate after burn-in, every state: {samples.mean(axis=0)}")
print(f"true covariance:\n{tgtCov}")
print(f"covariance estimate after burn-in:\n{np.cov(samples.T)}")
