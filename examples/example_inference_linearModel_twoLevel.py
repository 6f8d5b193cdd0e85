"""Linear forward model: random walk, two-level delayed acceptance over a model
hierarchy, the same with the adaptive error model, and preconditioned
Crank-Nicolson -- the ensemble version of the reference's
example_inference_linearModel_twoLevel.py (same maps, prior, noise, proposal,
minDataSize 100 with the noise heuristic)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from yagremcmc_amd.chain.method.aem import AEMBuilder
from yagremcmc_amd.chain.method.mlda import MLDABuilder
from yagremcmc_amd.chain.method.mrw import MRWBuilder
from yagremcmc_amd.chain.method.pcn import PCNBuilder
from yagremcmc_amd.exampleSetup import ExampleLinearModelSolver
from yagremcmc_amd.model.forwardModel import ForwardModel
from yagremcmc_amd.parameter.vector import ParameterVector
from yagremcmc_amd.statistics.bayesModel import BayesianRegressionModel
from yagremcmc_amd.statistics.covariance import IIDCovarianceMatrix
from yagremcmc_amd.statistics.data import Data
from yagremcmc_amd.statistics.gaussian import Gaussian
from yagremcmc_amd.statistics.likelihood import AdditiveGaussianNoiseLikelihood, AEMLikelihood
from yagremcmc_amd.statistics.modelHierarchy import BayesianRegressionModelHierarchy
from yagremcmc_amd.statistics.noise import CentredGaussianNoise
from yagremcmc_amd.utility.hierarchy import Hierarchy, SharedComponent

nChains = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nSteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
np.random.seed(2222)
DIM = 2

tgtMap = np.array([[1.4, -0.2], [-0.6, 0.7]])
surMap = tgtMap + np.array([[-0.6, -0.2], [0.4, 1.1]])
tgtSolver = ExampleLinearModelSolver(tgtMap, np.zeros(DIM))
surSolver = ExampleLinearModelSolver(surMap, np.array([0.5, -0.9]))

trueParam = ParameterVector(np.array([1.5, 0.5]))
noiseVar = 0.3
tgtSolver.interpolate(trueParam)
tgtSolver.invoke()                                   # forward solve on the device
data = Data(np.array([tgtSolver.evaluation + np.sqrt(noiseVar) * np.random.standard_normal(DIM) for _ in range(5)]))

prior = Gaussian(ParameterVector(trueParam.coefficient + np.array([-0.2, 0.4])), IIDCovarianceMatrix(DIM, 5.))
noise = CentredGaussianNoise(IIDCovarianceMatrix(DIM, noiseVar))
surLikelihood = AdditiveGaussianNoiseLikelihood(data, ForwardModel(surSolver), noise)
tgtLikelihood = AdditiveGaussianNoiseLikelihood(data, ForwardModel(tgtSolver), noise)
proposalCov = IIDCovarianceMatrix(DIM, 0.5)
init = ParameterVector(np.zeros(DIM))


def report(name, mc, burnin=None):
    burnin = min(500, nSteps // 4) if burnin is None else burnin
    states = mc.chain.ensemble_trajectory
    # a chain that did not move after the burn-in has no autocorrelation (short runs of a sampler with a poor
    # surrogate): ask for the marker instead of the error and say how many there were
    iat = np.atleast_1d(mc.integrated_autocorrelation('mean', burnin=burnin, allowUndefined=True))
    moved = iat != np.iinfo(np.int32).min
    stuck = "" if moved.all() else f"   ({int((~moved).sum())} chains did not move after the burn-in)"
    print(f"{name:28s} acceptance rate {mc.diagnostics.global_acceptance_rate():.3f}   posterior mean "
          f"{states[burnin:].reshape(-1, DIM).mean(axis=0)}   median IAT {np.median(iat[moved]):.0f}{stuck}")


mrw = MRWBuilder()
mrw.proposalCovariance, mrw.bayesModel, mrw.nChains = proposalCov, BayesianRegressionModel(tgtLikelihood, prior), nChains
tgtMRW = mrw.build_method()
tgtMRW.run(nSteps, init, verbose=False)
report("target MRW", tgtMRW)

mlda = MLDABuilder()
mlda.baseProposalCovariance, mlda.subChainLengths, mlda.nChains = proposalCov, [5], nChains
mlda.bayesModel = BayesianRegressionModelHierarchy(Hierarchy([surLikelihood, tgtLikelihood]), SharedComponent(prior, 2))
vanillaMLDA = mlda.build_method()
vanillaMLDA.run(nSteps, init, verbose=False)
report("vanilla MLDA (sub-chain 5)", vanillaMLDA)

aemSur = AEMLikelihood(data, ForwardModel(surSolver), noise, 100, True)
aemTgt = AEMLikelihood(data, ForwardModel(tgtSolver), noise, 100, True)
aem = AEMBuilder()
aem.baseProposalCovariance, aem.subChainLengths, aem.nChains = proposalCov, [5], nChains
aem.bayesModel = BayesianRegressionModelHierarchy(Hierarchy([aemSur, aemTgt]), SharedComponent(prior, 2))
aemMLDA = aem.build_method()
aemMLDA.run(nSteps, init, verbose=False)
report("MLDA + adaptive error model", aemMLDA)
em = aemMLDA.error_model()
print(f"{'':28s} chains past minDataSize: {int(em['has_precision'].sum())} of {nChains}; learned error mean "
      f"(median over chains) {np.median(em['mean'], axis=0)}")

centred = Gaussian(ParameterVector(np.zeros(DIM)), IIDCovarianceMatrix(DIM, 5.))
pcn = PCNBuilder()
pcn.stepSize, pcn.bayesModel, pcn.nChains = 0.05, BayesianRegressionModel(tgtLikelihood, centred), nChains
pcnChain = pcn.build_method()
pcnChain.run(nSteps, init, verbose=False)
report("pCN (centred prior, h=0.05)", pcnChain)
print(f"true parameter: {trueParam.coefficient}")
