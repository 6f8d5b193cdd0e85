"""The reference's example_inference_linearModel_twoLevel.py on the device: the same sequence of calls with the
builder imports switched and an ensemble size added --

    target MRW -> surrogate MRW -> MLDA burn-in (sub-chain 2) from where the surrogate chains stopped
    -> vanilla MLDA (sub-chain 5) from where the burn-in stopped -> MLDA with the adaptive error model from the same
    states; every stage thinned by its integrated autocorrelation time ('mean' for the random walks, 'max' for the
    delayed-acceptance chains) before its posterior mean is taken

(/root/reference: example_inference_linearModel_twoLevel.py:24-304; same maps, data seed, prior, noise, proposal,
minDataSize 100 with the noise heuristic).  K chains run each stage at once, a stage's chains start where THEIR
predecessors stopped, and the IAT is taken per chain on the device from the record in HBM; the thinning stride of a stage
is the median over its chains.  The plots of the reference's script need its host-side density grid and are left out.

    python examples/example_inference_linearModel_twoLevel.py [nChains [nSteps]]
"""
import os
import sys

import numpy as np
from numpy.random import seed, standard_normal

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from yagremcmc_amd.exampleSetup import ExampleLinearModelSolver
from yagremcmc_amd.parameter.vector import ParameterVector
from yagremcmc_amd.model.forwardModel import ForwardModel
from yagremcmc_amd.statistics.data import Data
from yagremcmc_amd.statistics.covariance import IIDCovarianceMatrix
from yagremcmc_amd.statistics.gaussian import Gaussian
from yagremcmc_amd.statistics.noise import CentredGaussianNoise
from yagremcmc_amd.statistics.likelihood import AdditiveGaussianNoiseLikelihood, AEMLikelihood
from yagremcmc_amd.statistics.bayesModel import BayesianRegressionModel
from yagremcmc_amd.statistics.modelHierarchy import BayesianRegressionModelHierarchy
from yagremcmc_amd.utility.hierarchy import SharedComponent, Hierarchy
from yagremcmc_amd.chain.method.mrw import MRWBuilder
from yagremcmc_amd.chain.method.mlda import MLDABuilder
from yagremcmc_amd.chain.method.aem import AEMBuilder
from yagremcmc_amd.chain.diagnostics import FullDiagnostics

nChains = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nStepsAll = int(sys.argv[2]) if len(sys.argv) > 2 else 50000      # the reference's 50 000 per stage
seed(2222)
DIM = 2

# -----------------------------------------------------------------------------
#                             PROBLEM SETUP
# -----------------------------------------------------------------------------
tgtMap = np.array([[1.4, -0.2], [-0.6, 0.7]])
tgtShift = np.zeros(DIM)
tgtSolver = ExampleLinearModelSolver(tgtMap, tgtShift)

surMapError = np.array([[-0.6, -0.2], [0.4, 1.1]])
surMap = tgtMap + surMapError
surShift = np.array([0.5, -0.9])
surSolver = ExampleLinearModelSolver(surMap, surShift)

tgtFwdModel = ForwardModel(tgtSolver)
surFwdModel = ForwardModel(surSolver)

trueParam = ParameterVector(np.array([1.5, 0.5]))
dataNoiseStdDev = np.sqrt(0.3)
nData = 5

tgtSolver.interpolate(trueParam)
tgtSolver.invoke()                                   # forward solve on the device
data = Data(np.array([tgtSolver.evaluation + dataNoiseStdDev * standard_normal(DIM) for _ in range(nData)]))
assert data.size == nData and data.dim == DIM

# -----------------------------------------------------------------------------
#                       BAYES MODEL DEFINITIONS
# -----------------------------------------------------------------------------
priorMean = ParameterVector(trueParam.coefficient + np.array([-0.2, 0.4]))
prior = Gaussian(priorMean, IIDCovarianceMatrix(DIM, 5.))

noiseModel = CentredGaussianNoise(IIDCovarianceMatrix(DIM, dataNoiseStdDev**2))

vanillaSurLikelihood = AdditiveGaussianNoiseLikelihood(data, surFwdModel, noiseModel)
vanillaTgtLikelihood = AdditiveGaussianNoiseLikelihood(data, tgtFwdModel, noiseModel)

minDataSize = 100
useNoiseHeuristic = True
aemSurLikelihood = AEMLikelihood(data, surFwdModel, noiseModel, minDataSize, useNoiseHeuristic)
aemTgtLikelihood = AEMLikelihood(data, tgtFwdModel, noiseModel, minDataSize, useNoiseHeuristic)

nLevels = 2
prior = SharedComponent(prior, nLevels)
vanillaLikelihood = Hierarchy([vanillaSurLikelihood, vanillaTgtLikelihood])
aemLikelihood = Hierarchy([aemSurLikelihood, aemTgtLikelihood])

tgtModel = BayesianRegressionModel(vanillaTgtLikelihood, prior.level(0))
surModel = BayesianRegressionModel(vanillaSurLikelihood, prior.level(0))
vanillaModel = BayesianRegressionModelHierarchy(vanillaLikelihood, prior)
aemModel = BayesianRegressionModelHierarchy(aemLikelihood, prior)

# -----------------------------------------------------------------------------
#                             MCMC SETUP
# -----------------------------------------------------------------------------
proposalCov = IIDCovarianceMatrix(DIM, 0.5)

tgtBuilder = MRWBuilder()
tgtBuilder.proposalCovariance = proposalCov
tgtBuilder.bayesModel = tgtModel
tgtBuilder.nChains = nChains
print("\nrequest building target mrw")
tgtMRW = tgtBuilder.build_method()

surBuilder = MRWBuilder()
surBuilder.proposalCovariance = proposalCov
surBuilder.bayesModel = surModel
surBuilder.nChains = nChains
print("\nrequest building surrogate mrw")
surMRW = surBuilder.build_method()

mldaBuilder = MLDABuilder()
mldaBuilder.baseProposalCovariance = proposalCov
mldaBuilder.subChainLengths = [2]
mldaBuilder.bayesModel = vanillaModel
mldaBuilder.nChains = nChains
print("\nrequest building burn-in mlda")
mldaBurnin = mldaBuilder.build_method()

mldaBuilder = MLDABuilder()
mldaBuilder.baseProposalCovariance = proposalCov
mldaBuilder.subChainLengths = [5]
mldaBuilder.bayesModel = vanillaModel
mldaBuilder.nChains = nChains
print("\nrequest building vanilla mlda")
vanillaMLDA = mldaBuilder.build_method()

aemBuilder = AEMBuilder()
aemBuilder.baseProposalCovariance = proposalCov
aemBuilder.subChainLengths = [5]
aemBuilder.bayesModel = aemModel
aemBuilder.targetDiagnostics = FullDiagnostics()
aemBuilder.nChains = nChains
print("\nrequest building aem mlda")
aemMLDA = aemBuilder.build_method()


# -----------------------------------------------------------------------------
#                                INFERENCE
# -----------------------------------------------------------------------------
def thinned_mean(mc, method, burnin):
    """the reference's `integrated_autocorrelation(states, method)` + `np.mean(states[burnin::thinning], axis=0)`,
    per chain on the device; the stage's stride is the median over its chains, the mean is over all of them"""
    iat = np.atleast_1d(mc.integrated_autocorrelation(method, allowUndefined=True))
    moved = iat != np.iinfo(np.int32).min          # a chain that never moved has no autocorrelation
    thinning = max(int(np.median(iat[moved])), 1) if moved.any() else 1
    states = mc.chain.ensemble_trajectory          # (n, K, d)
    return thinning, states[burnin::thinning].reshape(-1, DIM).mean(axis=0)


tgtNSteps = nStepsAll
tgtBurnin = min(500, tgtNSteps // 4)
initState = ParameterVector(np.zeros(DIM))

print(f"\nRunning {tgtNSteps} steps of the target MRW ({nChains} chains)")
tgtMRW.run(tgtNSteps, initState, verbose=False)
tgtThinning, tgtMean = thinned_mean(tgtMRW, 'mean', tgtBurnin)
tgtAccPr = tgtMRW.diagnostics.global_acceptance_rate()

surNSteps = nStepsAll
surBurnin = min(500, surNSteps // 4)
initState = ParameterVector(np.zeros(DIM))

print(f"\nRunning {surNSteps} steps of the surrogate MRW ({nChains} chains)")
surMRW.run(surNSteps, initState, verbose=False)
surThinning, surMean = thinned_mean(surMRW, 'mean', surBurnin)
surAccPr = surMRW.diagnostics.global_acceptance_rate()

# start the actual burn-in chains where the surrogate chains left off (every chain from its own last state)
initState = surMRW.chain.ensemble_trajectory[-1]
nSteps = min(500, nStepsAll)
print(f"\n\n\nBurning-in MLDA with {nSteps} steps, starting at the last surrogate chain positions")
mldaBurnin.run(nSteps, initState, verbose=False)

initState = mldaBurnin.chain.ensemble_trajectory[-1]
initStateCopy = np.copy(initState)

nSteps = nStepsAll
print(f"\n\n\nRunning {nSteps} steps of vanilla MLDA")
vanillaMLDA.run(nSteps, initState, verbose=False)
vMLDAThinning, vMLDAPostMean = thinned_mean(vanillaMLDA, 'max', 0)
vMLDAAccPr = vanillaMLDA.diagnostics.global_acceptance_rate()

print(f"\n\n\nRunning {nSteps} steps of MLDA with an adaptive error model")
assert np.array_equal(initState, initStateCopy)
aemMLDA.run(nSteps, initState, verbose=False)
aMLDAThinning, aMLDAPostMean = thinned_mean(aemMLDA, 'max', 0)
aMLDAAccPr = aemMLDA.diagnostics.global_acceptance_rate()

print("\n\nResults for target MRW")
print("----------------------")
print(f"\nacceptance rate: {tgtAccPr}")
print(f"IAT estimate: {tgtThinning}")
print(f"posterior mean: {tgtMean}")

print("\n\nResults for surrogate MRW")
print("----------------------")
print(f"\nacceptance rate: {surAccPr}")
print(f"IAT estimate: {surThinning}")
print(f"posterior mean: {surMean}")

print("\n\nResults for vanilla MLDA")
print("-------------------------")
print(f"acceptance rate: {vMLDAAccPr}")
print(f"IAT estimate: {vMLDAThinning}")
print(f"posterior mean: {vMLDAPostMean}")

print("\n\nResults for AEM MLDA")
print("---------------------")
print(f"acceptance rate: {aMLDAAccPr}")
print(f"IAT estimate: {aMLDAThinning}")
print(f"posterior mean: {aMLDAPostMean}")
em = aemMLDA.error_model()                          # per chain: what its coarse likelihood learned
print(f"estimated mean error (median over chains): {np.median(em['mean'], axis=0)}")

print("\nAEM DETAILS:")
print(f"Estimated Marginal Variance of posterior: {aemMLDA.diagnostics.marginal_variance()}")
print(f"chains past minDataSize = {minDataSize} accepted fine proposals: {int(em['has_precision'].sum())} of {nChains}")
print(f"accepted fine proposals per chain (the error model's sample count), median: {int(np.median(em['n']))}")
print(f"\ntrue parameter: {trueParam.coefficient}")
