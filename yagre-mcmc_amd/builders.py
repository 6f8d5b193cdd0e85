"""Builders: the drop-in boundary of the chain-ensemble path.

Same property names, validation and error messages as the reference's
`ChainBuilder` (yagremcmc/chain/builder.py:7-83), `MRWBuilder`
(chain/method/mrw.py:60-91) and `MLDABuilder` (chain/method/mlda.py:157-344);
`build_method()` returns an ensemble sampler with the reference's
`run / chain / diagnostics / target / clear` surface.  Ensemble extensions are
extra properties with defaults that reproduce the single-chain behaviour.
"""
from . import diagnostics as dg
from .ensemble import MLDA, AdaptiveErrorModel, MetropolisedRandomWalk, PreconditionedCrankNicolson
from .lowering import UnsupportedPlugin
from .plugins import UnnormalisedPosterior


class ChainBuilder:
    """chain/builder.py:7-83"""

    def __init__(self):
        self._bayesModel = None
        self._explicitTarget = None
        self._diagnostics = dg.AcceptanceRateDiagnostics()
        # ensemble extensions
        self.nChains = 1
        self.seed = 0
        self.device = 0
        self.chainOffset = 0
        self.recordTrajectory = True
        self.recordLogPosterior = False
        self.rk4Steps = None

    @property
    def bayesModel(self):
        return self._bayesModel

    @bayesModel.setter
    def bayesModel(self, model):
        self._bayesModel = model

    @property
    def explicitTarget(self):
        return self._explicitTarget

    @explicitTarget.setter
    def explicitTarget(self, density):
        self._explicitTarget = density

    @property
    def diagnostics(self):
        return self._diagnostics

    @diagnostics.setter
    def diagnostics(self, diagnostics):
        self._diagnostics = diagnostics

    def _ensemble_options(self):
        return dict(nChains=self.nChains, seed=self.seed, device=self.device, chainOffset=self.chainOffset,
                    recordTrajectory=self.recordTrajectory, recordLogPosterior=self.recordLogPosterior,
                    rk4Steps=self.rk4Steps)

    def validate_target_measure(self):
        if self._bayesModel is None and self._explicitTarget is None:
            raise ValueError("Either bayesian model or explicit target density must be provided for chain setup")
        if self._bayesModel is not None and self._explicitTarget is not None:
            raise ValueError("Only one of bayes model or explicit target density should be provided.")

    def target_is_posterior(self):
        return self._bayesModel is not None

    def target_is_explicit(self):
        return self._explicitTarget is not None

    def _validate_parameters(self):
        raise NotImplementedError

    def build_from_model(self):
        raise NotImplementedError

    def build_from_target(self):
        raise NotImplementedError

    def build_method(self):
        self._validate_parameters()
        self.validate_target_measure()
        if self.target_is_posterior():
            return self.build_from_model()
        if self.target_is_explicit():
            return self.build_from_target()
        raise ValueError("Invalid target distribution")


class MRWBuilder(ChainBuilder):
    """chain/method/mrw.py:60-91"""

    def __init__(self):
        super().__init__()
        self._proposalCov = None

    @property
    def proposalCovariance(self):
        return self._proposalCov

    @proposalCovariance.setter
    def proposalCovariance(self, covariance):
        self._proposalCov = covariance

    def build_from_model(self):
        target = UnnormalisedPosterior(self._bayesModel.likelihood, self._bayesModel.prior)
        return MetropolisedRandomWalk(target, self._proposalCov, self._diagnostics, **self._ensemble_options())

    def build_from_target(self):
        return MetropolisedRandomWalk(self._explicitTarget, self._proposalCov, self._diagnostics,
                                      **self._ensemble_options())

    def _validate_parameters(self):
        if self._proposalCov is None:
            raise ValueError("Proposal Covariance not set for MRW")


class MLDABuilder(ChainBuilder):
    """chain/method/mlda.py:157-344: one surrogate (two-level delayed acceptance) or an even number
    of them (the nest the reference executes, ensemble.MLDA)."""

    def __init__(self):
        super().__init__()
        self._basePropCov = None
        self._nSteps = None
        self._surrTgts = None
        self._surrDgnstList = None
        self._tgtDgnst = None
        self._biasCorrection = None
        self.recordSubChain = False

    @property
    def baseProposalCovariance(self):
        return self._basePropCov

    @baseProposalCovariance.setter
    def baseProposalCovariance(self, cov):
        self._basePropCov = cov

    @property
    def subChainLengths(self):
        return self._nSteps

    @subChainLengths.setter
    def subChainLengths(self, nSteps):
        self._nSteps = nSteps

    @property
    def surrogateTargets(self):
        return self._surrTgts

    @surrogateTargets.setter
    def surrogateTargets(self, tgts):
        self._surrTgts = tgts

    @property
    def targetDiagnostics(self):
        return self._tgtDgnst

    @targetDiagnostics.setter
    def targetDiagnostics(self, diagnostics):
        self._tgtDgnst = diagnostics

    @property
    def surrogateDiagnostics(self):
        return self._surrDgnstList

    @surrogateDiagnostics.setter
    def surrogateDiagnostics(self, surrDgnstList):
        self._surrDgnstList = surrDgnstList

    @property
    def biasCorrection(self):
        return self._biasCorrection

    @biasCorrection.setter
    def biasCorrection(self, bias):
        self._biasCorrection = bias

    def _validate_parameters(self):
        if self._basePropCov is None:
            raise ValueError("Coarse proposal covariance not set for MLDA")
        if self._nSteps is None:
            raise ValueError("Subchain lengths not set for MLDA")
        if self._bayesModel is not None:
            # mlda.py:228-249
            if not (hasattr(self._bayesModel, "level") and hasattr(self._bayesModel, "size")):
                raise ValueError("MLDA requires a hierarchy of models.")
            if self._surrTgts is not None:
                raise ValueError("Cannot set explicit surrogate targets for a hierarchy of Bayesian models.")
            if not len(self._nSteps) == self._bayesModel.size - 1:
                raise ValueError("Number of sub-chain lengths does not match the size of the model hierarchy.")
            if self._surrDgnstList is not None and not len(self._surrDgnstList) == self._bayesModel.size - 1:
                raise ValueError("Number of diagnostics does not match the size of the model hierarchy")
            if self._biasCorrection is not None:
                raise UnsupportedPlugin("bias correction has no device lowering")
        if self._explicitTarget is not None:
            if self._surrTgts is None:
                raise ValueError("Surrogate targets not set for MLDA")
            if not len(self._nSteps) == len(self._surrTgts):
                raise ValueError("Number of sub-chain lengths does not match number of surrogate targets")
            if self._surrDgnstList is not None and not len(self._surrDgnstList) == len(self._surrTgts):
                raise ValueError("Number of diagnostics does not match the size of the model hierarchy")
            if self._biasCorrection is not None:
                raise UnsupportedPlugin("bias correction has no device lowering")

    def build_from_model(self):
        """mlda.py:315-335: the finest level is the target, the coarser ones are the surrogates."""
        self._validate_parameters()
        model = self._bayesModel
        target = UnnormalisedPosterior(model.level(-1).likelihood, model.level(-1).prior)
        surrogates = [UnnormalisedPosterior(model.level(k).likelihood, model.level(k).prior)
                      for k in range(model.size - 1)]
        tgt_diag = self._tgtDgnst if self._tgtDgnst is not None else dg.AcceptanceRateDiagnostics()
        return self.build_mlda(target, surrogates, self._basePropCov, list(self._nSteps), tgt_diag,
                               self._surrDgnstList)

    def build_from_target(self):
        tgt_diag = self._tgtDgnst if self._tgtDgnst is not None else dg.AcceptanceRateDiagnostics()
        return self.build_mlda(self._explicitTarget, list(self._surrTgts), self._basePropCov, list(self._nSteps),
                               tgt_diag, self._surrDgnstList)

    def build_mlda(self, tgtPost, surPost, bpc, nS, tgtD, surD):
        """mlda.py:312-313"""
        return MLDA(tgtPost, surPost, bpc, nS, tgtD, surD, recordSubChain=self.recordSubChain,
                    **self._ensemble_options())


class AEMBuilder(MLDABuilder):
    """chain/method/aem.py:61-82"""

    def _validate_parameters(self):
        super()._validate_parameters()
        if self._bayesModel is None:
            raise NotImplementedError("Adaptive error correction only makes sense if the target emerges from a "
                                      "Bayesian model.")
        for i in range(self._bayesModel.size):
            if type(self._bayesModel.level(i).likelihood).__name__ != "AEMLikelihood":
                raise ValueError(f"Likelihood on level {i} is not adaptive.")

    def build_mlda(self, tgtPost, surPost, bpc, nS, tgtD, surD):
        return AdaptiveErrorModel(tgtPost, surPost, bpc, nS, tgtD, surD, recordSubChain=self.recordSubChain,
                                  **self._ensemble_options())


class PCNBuilder(ChainBuilder):
    """chain/method/pcn.py:60-88"""

    def __init__(self):
        super().__init__()
        self._stepSize = None

    @property
    def stepSize(self):
        return self._stepSize

    @stepSize.setter
    def stepSize(self, h):
        self._stepSize = h

    def build_from_model(self):
        return PreconditionedCrankNicolson(self._bayesModel.likelihood, self._bayesModel.prior, self._stepSize,
                                           self._diagnostics, **self._ensemble_options())

    def build_from_target(self):
        raise RuntimeError("PCN is only defined in relation to a Bayesian model")

    def _validate_parameters(self):
        if self._stepSize is None:
            raise ValueError("Step size not set in PCN.")
