"""Builders: the drop-in boundary of the chain-ensemble path.

The PUBLIC surface is the reference's: property names, `build_method()`,
`validate_target_measure()`, and the text of every misconfiguration error of
`ChainBuilder` (yagremcmc/chain/builder.py:7-83), `MRWBuilder`
(chain/method/mrw.py:60-91), `MLDABuilder` (chain/method/mlda.py:157-344),
`PCNBuilder` (chain/method/pcn.py:60-88) and `AEMBuilder`
(chain/method/aem.py:61-82) -- example scripts and the reference's builder
tests (test_builder.py matches the messages by regex) run unchanged.
Underneath, a builder is a bag of settings: every settable property is an
`Option` over one dictionary, and each class lists its consistency rules as
(predicate, exception, message) rows that `build_method()` walks in order.
Ensemble extensions (`nChains`, `seed`, `device`, `chainOffset`,
`recordTrajectory`, `recordLogPosterior`, `recordStride`, `rk4Steps`) are
further options whose defaults reproduce the single-chain behaviour.
"""
from . import diagnostics as dg
from .ensemble import MLDA, AdaptiveErrorModel, MetropolisedRandomWalk, PreconditionedCrankNicolson
from .lowering import UnsupportedPlugin
from .plugins import UnnormalisedPosterior


class Option:
    """A settable builder property stored in the builder's settings dictionary."""

    def __init__(self, default=None):
        self.default = default

    def __set_name__(self, owner, name):
        self.name = name

    def __get__(self, obj, owner=None):
        if obj is None:
            return self
        value = obj._settings.get(self.name, self)
        if value is self:  # first read: materialise the default (callables make per-builder objects)
            value = self.default() if callable(self.default) else self.default
            obj._settings[self.name] = value
        return value

    def __set__(self, obj, value):
        obj._settings[self.name] = value


ENSEMBLE_OPTIONS = ("nChains", "seed", "device", "chainOffset", "recordTrajectory", "recordLogPosterior",
                    "recordStride", "rk4Steps")


class ChainBuilder:
    """chain/builder.py:7-83"""

    bayesModel = Option()
    explicitTarget = Option()
    diagnostics = Option(dg.AcceptanceRateDiagnostics)
    # ensemble extensions
    nChains = Option(1)
    seed = Option(0)
    device = Option(0)
    chainOffset = Option(0)
    recordTrajectory = Option(True)
    recordLogPosterior = Option(False)
    recordStride = Option(1)
    rk4Steps = Option()

    #: consistency rules of the concrete builder: (predicate(builder) -> bool "violated", exception, message)
    rules = ()

    def __init__(self):
        self._settings = {}

    def _ensemble_options(self):
        return {name: getattr(self, name) for name in ENSEMBLE_OPTIONS}

    def target_is_posterior(self):
        return self.bayesModel is not None

    def target_is_explicit(self):
        return self.explicitTarget is not None

    def validate_target_measure(self):
        """builder.py:40-50"""
        given = self.target_is_posterior() + self.target_is_explicit()
        if given == 0:
            raise ValueError("Either bayesian model or explicit target density must be provided for chain setup")
        if given == 2:
            raise ValueError("Only one of bayes model or explicit target density should be provided.")

    def _validate_parameters(self):
        for klass in reversed(type(self).__mro__):
            for violated, exc, message in vars(klass).get("rules", ()):
                if violated(self):
                    raise exc(message(self) if callable(message) else message)

    def build_from_model(self):
        raise NotImplementedError

    def build_from_target(self):
        raise NotImplementedError

    def build_method(self):
        """builder.py:72-83"""
        self._validate_parameters()
        self.validate_target_measure()
        return self.build_from_model() if self.target_is_posterior() else self.build_from_target()


def _posterior_of(model):
    return UnnormalisedPosterior(model.likelihood, model.prior)


class MRWBuilder(ChainBuilder):
    """chain/method/mrw.py:60-91"""

    proposalCovariance = Option()
    rules = ((lambda b: b.proposalCovariance is None, ValueError, "Proposal Covariance not set for MRW"),)

    def _sampler(self, target):
        return MetropolisedRandomWalk(target, self.proposalCovariance, self.diagnostics, **self._ensemble_options())

    def build_from_model(self):
        return self._sampler(_posterior_of(self.bayesModel))

    def build_from_target(self):
        return self._sampler(self.explicitTarget)


def _is_hierarchy(model):
    return hasattr(model, "level") and hasattr(model, "size")


def _n_surrogates(b):
    """how many coarse levels the configured target has (mlda.py:228-262)"""
    return b.bayesModel.size - 1 if b.target_is_posterior() else len(b.surrogateTargets)


class MLDABuilder(ChainBuilder):
    """chain/method/mlda.py:157-344: one surrogate (two-level delayed acceptance) or an even number
    of them (the nest the reference executes, ensemble.MLDA)."""

    baseProposalCovariance = Option()
    subChainLengths = Option()
    surrogateTargets = Option()
    targetDiagnostics = Option()
    surrogateDiagnostics = Option()
    biasCorrection = Option()
    recordSubChain = Option(False)

    sampler = MLDA

    rules = (
        (lambda b: b.baseProposalCovariance is None, ValueError, "Coarse proposal covariance not set for MLDA"),
        (lambda b: b.subChainLengths is None, ValueError, "Subchain lengths not set for MLDA"),
        # a model hierarchy as target (mlda.py:228-249)
        (lambda b: b.target_is_posterior() and not _is_hierarchy(b.bayesModel), ValueError,
         "MLDA requires a hierarchy of models."),
        (lambda b: b.target_is_posterior() and b.surrogateTargets is not None, ValueError,
         "Cannot set explicit surrogate targets for a hierarchy of Bayesian models."),
        (lambda b: b.target_is_posterior() and len(b.subChainLengths) != _n_surrogates(b), ValueError,
         "Number of sub-chain lengths does not match the size of the model hierarchy."),
        # explicit densities as target (mlda.py:251-269)
        (lambda b: b.target_is_explicit() and b.surrogateTargets is None, ValueError,
         "Surrogate targets not set for MLDA"),
        (lambda b: b.target_is_explicit() and len(b.subChainLengths) != _n_surrogates(b), ValueError,
         "Number of sub-chain lengths does not match number of surrogate targets"),
        (lambda b: (b.target_is_posterior() or b.target_is_explicit()) and b.surrogateDiagnostics is not None
         and len(b.surrogateDiagnostics) != _n_surrogates(b), ValueError,
         "Number of diagnostics does not match the size of the model hierarchy"),
        (lambda b: (b.target_is_posterior() or b.target_is_explicit()) and b.biasCorrection is not None,
         UnsupportedPlugin, "bias correction has no device lowering"),
    )

    def _levels(self):
        """(target, [surrogates coarsest first]) -- mlda.py:315-344: for a model hierarchy the finest level is
        the target and the coarser ones are the surrogates."""
        if self.target_is_posterior():
            m = self.bayesModel
            return _posterior_of(m.level(-1)), [_posterior_of(m.level(k)) for k in range(m.size - 1)]
        return self.explicitTarget, list(self.surrogateTargets)

    def _build(self):
        target, surrogates = self._levels()
        top = self.targetDiagnostics if self.targetDiagnostics is not None else dg.AcceptanceRateDiagnostics()
        return self.build_mlda(target, surrogates, self.baseProposalCovariance, list(self.subChainLengths), top,
                               self.surrogateDiagnostics)

    def build_from_model(self):
        self._validate_parameters()
        return self._build()

    def build_from_target(self):
        return self._build()

    def build_mlda(self, target, surrogates, baseProposalCov, subChainLengths, targetDiag, surrogateDiags):
        """mlda.py:312-313"""
        return self.sampler(target, surrogates, baseProposalCov, subChainLengths, targetDiag, surrogateDiags,
                            recordSubChain=self.recordSubChain, **self._ensemble_options())


def _non_adaptive_level(b):
    """index of the first level whose likelihood is not an AEMLikelihood, or None (aem.py:66-79)"""
    if not b.target_is_posterior() or not _is_hierarchy(b.bayesModel):
        return None
    for i in range(b.bayesModel.size):
        if type(b.bayesModel.level(i).likelihood).__name__ != "AEMLikelihood":
            return i
    return None


class AEMBuilder(MLDABuilder):
    """chain/method/aem.py:61-82"""

    sampler = AdaptiveErrorModel

    rules = (
        (lambda b: not b.target_is_posterior(), NotImplementedError,
         "Adaptive error correction only makes sense if the target emerges from a Bayesian model."),
        (lambda b: _non_adaptive_level(b) is not None, ValueError,
         lambda b: f"Likelihood on level {_non_adaptive_level(b)} is not adaptive."),
    )


class PCNBuilder(ChainBuilder):
    """chain/method/pcn.py:60-88"""

    stepSize = Option()
    rules = ((lambda b: b.stepSize is None, ValueError, "Step size not set in PCN."),)

    def build_from_model(self):
        return PreconditionedCrankNicolson(self.bayesModel.likelihood, self.bayesModel.prior, self.stepSize,
                                           self.diagnostics, **self._ensemble_options())

    def build_from_target(self):
        raise RuntimeError("PCN is only defined in relation to a Bayesian model")
