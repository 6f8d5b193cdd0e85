"""The built sampler objects: ensembles of Metropolis-Hastings chains on the GPU.

`MetropolisedRandomWalk` and `MLDA` keep the surface of the reference's classes
of the same names (yagremcmc/chain/method/mrw.py:41-57, mlda.py:120-154, both
deriving MetropolisHastings, chain/metropolisHastings.py:12-125): `run`,
`.chain`, `.target`, `.diagnostics`, `.clear()`.  Underneath, `run` lowers the
plug-ins to a parameter block and advances `nChains` independent chains with
one HIP kernel launch per reporting interval.  Ensemble extensions are
keyword-only.
"""
import logging

import numpy as np

from . import _ffi, diagnostics as dg, lowering


class EnsembleChain:
    """chain/chain.py:4-21.  For one chain `trajectory` is the reference's list
    of coefficient arrays; for K chains it is the (N, K, d) record."""

    def __init__(self):
        self._x0 = None
        self._run = None
        self._cache = None

    def _attach(self, x0, run):
        self._x0, self._run, self._cache = x0, run, None

    @property
    def ensemble_trajectory(self):
        """(N, K, d): row 0 the initial states, row n the states after transition n."""
        if self._run is None:
            return np.zeros((0, 0, 0))
        if self._cache is None:
            if not (self._run.flags & _ffi.RECORD_TRAJECTORY):
                raise RuntimeError("this run did not record the trajectory")
            self._cache = np.concatenate([self._x0[None], self._run.trajectory()], axis=0)
        return self._cache

    @property
    def trajectory(self):
        full = self.ensemble_trajectory
        if full.shape[1] == 1:
            return [full[n, 0] for n in range(full.shape[0])]
        return full

    def trajectory_of(self, k):
        return self.ensemble_trajectory[:, k, :]

    @property
    def length(self):
        if self._run is None:
            return 0
        return self._run.steps_done()[1] + 1

    def clear(self):
        self._x0 = self._run = self._cache = None


class MetropolisHastingsEnsemble:
    """Template of the ensemble samplers (reference: MetropolisHastings,
    chain/metropolisHastings.py:12-125)."""

    _method = None
    _logger = None

    def __init__(self, targetDensity, proposalCov, diagnostics, *, nChains=1, seed=0, device=0,
                 chainOffset=0, recordTrajectory=True, recordLogPosterior=False, recordStride=1, rk4Steps=None):
        self._tgtDensity = targetDensity
        self._proposalCov = proposalCov
        self._diagnostics = dg.adopt(diagnostics)
        self._chain = EnsembleChain()
        self.nChains = int(nChains)
        self.seed = int(seed)
        self.device = int(device)
        self.chainOffset = int(chainOffset)
        self.recordTrajectory = bool(recordTrajectory)
        self.recordLogPosterior = bool(recordLogPosterior)
        self.recordStride = int(recordStride)
        self.rk4Steps = rk4Steps
        self._run = None
        self._problem = None
        if self.nChains < 1:
            raise ValueError("nChains must be positive")
        if self.recordStride < 1:
            raise ValueError("recordStride must be at least 1")

    # -- reference surface ---------------------------------------------------
    @property
    def chain(self):
        return self._chain

    @property
    def target(self):
        return self._tgtDensity

    @property
    def diagnostics(self):
        return self._diagnostics

    def clear(self):
        self._diagnostics.reset()
        self._chain.clear()
        self._run = None

    def run(self, chainLength, initialState, verbose=True):
        """metropolisHastings.py:103-120: chainLength states per chain including
        the initial one.  `initialState` is a parameter object (broadcast to all
        chains) or an array (nChains, d)."""
        chainLength = int(chainLength)
        if chainLength < 1:
            raise ValueError("chainLength must be at least 1")
        x0, state_eq = self._initial_states(initialState)
        ctx = _ffi.default_context(self.device)
        self._problem = self._make_problem(ctx, state_eq)
        nsteps = chainLength - 1
        run = _ffi.Run(self._problem, self.seed, self.nChains, x0, max(nsteps // self.recordStride, 1),
                       self._record_flags(), self.chainOffset, record_stride=self.recordStride)
        self._run = run
        self._chain._attach(np.broadcast_to(x0, (self.nChains, x0.shape[1])).copy(), run)
        if verbose and nsteps > 0:
            self._advance_verbosely(run, nsteps, chainLength)
        elif nsteps > 0:
            run.advance(nsteps)
        run.sync()
        self._ingest()

    # -- ensemble extensions -------------------------------------------------
    def advance(self, nSteps):
        """Continue every chain by nSteps transitions (the record must have room)."""
        if self._run is None:
            raise RuntimeError("run() has not been called")
        self._run.advance(int(nSteps))
        self._run.sync()
        self._chain._cache = None
        self._ingest()

    @property
    def accept_decisions(self):
        if not (self._run.flags & _ffi.RECORD_ACCEPT):
            raise RuntimeError("the per-transition decisions are not recorded with recordStride > 1 "
                               "(accept counts and rates are)")
        return self._run.accept()

    @property
    def log_posterior(self):
        """(N, K) log-density of every recorded state (needs recordLogPosterior)."""
        x0 = self._chain._x0
        lp0 = self._problem.logdensity(x0)[0]
        return np.concatenate([lp0[None], self._run.logpost()], axis=0)

    @property
    def final_state(self):
        return self._run.state()

    @property
    def chain_status(self):
        return self._run.summary()["status"]

    def kernel_time_ms(self):
        return self._run.kernel_time_ms()

    def integrated_autocorrelation(self, method='mean', burnin=0, sokalConst=5., allowUndefined=False):
        """postprocessing/autocorrelation.py:92-140 for every chain, computed on the device from the
        record: the equivalent of `ac.integrated_autocorrelation(states[burnin:], method)` per chain.
        Returns an int for one chain, an array of nChains ints otherwise.  A chain that never moved has no
        autocorrelation: ValueError, or IAT_UNDEFINED (INT32_MIN) in its place with allowUndefined."""
        methods = {"mean": 0, "max": 1}
        if method not in methods:
            raise ValueError(f"Invalid IAT - Type: {method}. Options are 'mean' and 'max'.")
        if burnin == 0:
            # row 0 of the reference's trajectory is the initial state, which the device record
            # does not hold: go through the fetched states so that it is included
            states = self._chain.ensemble_trajectory
            out = _ffi.default_context(self.device).iat_states(states, methods[method], sokalConst, allowUndefined)
        else:
            out = self._run.iat(burnin - 1, None, methods[method], sokalConst, allowUndefined)
        return int(out[0]) if self.nChains == 1 else out

    # -- internals -------------------------------------------------------------
    def _initial_states(self, initialState):
        if hasattr(initialState, "coefficient"):
            x0 = np.asarray(initialState.coefficient, dtype=np.float64).reshape(1, -1)
            return x0, lowering.state_eq_of(initialState)
        x0 = np.asarray(initialState, dtype=np.float64)
        if x0.ndim == 1:
            x0 = x0.reshape(1, -1)
        if x0.shape[0] not in (1, self.nChains):
            raise ValueError(f"initial states have {x0.shape[0]} rows for {self.nChains} chains")
        return x0, _ffi.EQ_EXACT

    def _record_flags(self):
        """recordStride = s > 1 (a thinned record: the states after transitions s, 2s, ... -- what the
        reference's examples keep after thinning by the IAT, example_mcmc_2d_singleLevel.py:71-79): the
        per-transition decision record is dropped, accept counts and moments still see every transition."""
        flags = _ffi.RECORD_ACCEPT if self.recordStride == 1 else 0
        if self.recordTrajectory:
            flags |= _ffi.RECORD_TRAJECTORY
        if self.recordLogPosterior:
            flags |= _ffi.RECORD_LOGPOST
        if self._diagnostics.kind == "full":
            flags |= _ffi.RECORD_MOMENTS
        return flags

    def _make_problem(self, ctx, state_eq):
        raise NotImplementedError

    def _ingest(self):
        run = self._run
        fetch = run.accept if (run.flags & _ffi.RECORD_ACCEPT) else None
        self._diagnostics._ingest(fetch, run.summary(), run.steps_done()[0])

    def _advance_verbosely(self, run, nsteps, chainLength):
        """VerbosityController (utility/verbosity.py:11-52): every
        max(chainLength // 20, 10) steps report the acceptance rate of the last
        interval (pooled over chains) and, for FullDiagnostics, the condition
        number of the marginal variances."""
        log = self._get_logger()
        interval = max(chainLength // 20, 10)
        log.info("Starting %d chain(s).", self.nChains)
        done = 0
        prev = np.zeros(self.nChains, dtype=np.uint64)
        while done < nsteps:
            step = min(interval, nsteps - done)
            run.advance(step)
            done += step
            if done < nsteps:
                s = run.summary()
                rate = float((s["accepted"] - prev).sum()) / (step * self.nChains)
                prev = s["accepted"]
                log.info("%d steps computed.  - Rolling acceptance rate: %.4f", done, rate)
                if self._diagnostics.kind == "full" and done >= 2:
                    n, _, m2 = dg.merge_welford(s["wf_n"], s["wf_mean"], s["wf_m2"])
                    mv = m2 / (n - 1)
                    if np.min(mv) > 1e-12:
                        log.info("  - Estimated condition number: %.4f", np.max(mv) / np.min(mv))

    @classmethod
    def _get_logger(cls):
        if cls._logger is None:
            logger = logging.getLogger("yagremcmc_amd.MH")
            if not logger.handlers:
                h = logging.StreamHandler()
                h.setFormatter(logging.Formatter("%(levelname)s: %(message)s"))
                logger.addHandler(h)
                logger.setLevel(logging.INFO)
            cls._logger = logger
        return cls._logger


class MetropolisedRandomWalk(MetropolisHastingsEnsemble):
    """chain/method/mrw.py:41-57: Gaussian random-walk proposal x + sigma*xi
    (mrw.py:9-38), symmetric acceptance ratio min(1, exp(logpi(y) - logpi(x)))."""

    def _make_problem(self, ctx, state_eq):
        dim = lowering.lower_density(self._tgtDensity, self.rk4Steps).desc.dim
        sigma = lowering.proposal_sigma(self._proposalCov, dim)
        return lowering.make_problem(ctx, _ffi.METHOD_MRW, self._tgtDensity, sigma, state_eq,
                                     rk4_steps=self.rk4Steps, chol=lowering.proposal_chol(self._proposalCov, dim))


class SurrogateLevel:
    """What `mcmc.surrogate(i)` returns (mlda.py:84-89 hands out the reference's per-level
    MetropolisHastings object): the level's target and its acceptance statistics."""

    def __init__(self, owner, index):
        self._owner, self._index = owner, index

    @property
    def target(self):
        return self._owner._surrogates[self._index]

    @property
    def diagnostics(self):
        return self._owner._level_diagnostics(self._index)


class LevelAcceptance:
    """AcceptanceRateDiagnostics surface (diagnostics.py:36-54) of a lower level: counts only."""

    def __init__(self, accepted, made):
        self.accepted, self.made = accepted, int(made)

    def global_acceptance_rate(self):
        if self.made == 0:
            return 0.
        return float(self.accepted.sum()) / (self.made * self.accepted.size)

    def per_chain_acceptance_rate(self):
        return self.accepted / max(self.made, 1)


class MLDA(MetropolisHastingsEnsemble):
    """chain/method/mlda.py:120-154.

    One surrogate: the proposal is the end point of a subChainLength-step random walk on the
    surrogate density (mlda.py:102-110), acceptance is mlda.py:146-154 (YMC_METHOD_MLDA2).

    An even number of surrogates: the nest the reference actually executes, in which only every
    second level applies its accept step (YMC_METHOD_MLDA_NESTED, include/ymc.h).  An odd number
    >= 3 fails in the reference when `run` reaches the base level (`MetropolisedRandomWalk` is
    asked for `set_state`); the same AttributeError is raised here."""

    def __init__(self, targetDensity, surrogateDensities, baseProposalCov, nSteps, targetDiagnostics,
                 surrogateDiagnosticsList=None, *, recordSubChain=False, **ensemble):
        if len(surrogateDensities) < 1:
            raise ValueError("Surrogate hierarchy must contain at least one surrogate.")
        if len(nSteps) != len(surrogateDensities):
            raise ValueError("Number of sub-chain lengths does not match number of surrogate targets")
        super().__init__(targetDensity, baseProposalCov, targetDiagnostics, **ensemble)
        self._surrogates = list(surrogateDensities)
        self._nSteps = [int(v) for v in nSteps]
        self._surrogateDiagnostics = surrogateDiagnosticsList
        self.recordSubChain = bool(recordSubChain)
        # diagnostics handed in for the one surrogate (mlda.py:84-89 reads them after the run) need its decisions
        if (surrogateDiagnosticsList is not None and len(self._surrogates) == 1 and self.recordStride == 1
                and self.recordTrajectory):
            self.recordSubChain = True
        if self.recordSubChain and len(self._surrogates) != 1:
            raise lowering.UnsupportedPlugin("sub-chain decisions are recorded for one surrogate only")
        if self.recordSubChain and self.recordStride > 1:
            raise ValueError("sub-chain decisions are a per-transition record: recordStride must be 1")

    @property
    def nSurrogates(self):
        return len(self._surrogates)

    def surrogate(self, sIdx):
        if sIdx < -1 or sIdx >= self.nSurrogates:
            raise IndexError(f"invalid surrogate index: {sIdx}")
        return SurrogateLevel(self, sIdx % self.nSurrogates)

    @property
    def sub_chain_decisions(self):
        """(N-1, S, K) outcomes of the coarse sub-steps (one surrogate, needs recordSubChain)."""
        return self._run.sub_accept()

    def level_acceptance_rates(self):
        """Acceptance rate of every level that takes decisions, coarsest first, the recorded chain
        last (pooled over chains)."""
        top = self._diagnostics.global_acceptance_rate()
        if self.nSurrogates == 1:
            return [top]
        acc, made = self._run.level_accept()
        return [LevelAcceptance(acc[e], made[e]).global_acceptance_rate() for e in range(acc.shape[0])] + [top]

    def _level_diagnostics(self, index):
        """Surrogate i of an even hierarchy decides iff i is even (it is level i/2 of the nest)."""
        n = self.nSurrogates
        if self._run is None or (n > 1 and index % 2 == 1):
            return LevelAcceptance(np.zeros(self.nChains, dtype=np.uint64), 0)
        if n == 1:
            # one surrogate: the coarse decisions are counted from the sub-chain record (the kernels keep no counter)
            if not self.recordSubChain:
                raise RuntimeError("the surrogate's acceptance statistics of a two-level run come from the record of its "
                                   "sub-chain decisions: set recordSubChain = True on the builder")
            sub = self.sub_chain_decisions
            return LevelAcceptance(sub.sum(axis=(0, 1), dtype=np.uint64), sub.shape[0] * sub.shape[1])
        acc, made = self._run.level_accept()
        return LevelAcceptance(acc[index // 2], made[index // 2])

    def run(self, chainLength, initialState, verbose=True):
        n = self.nSurrogates
        if n >= 3 and n % 2 == 1 and int(chainLength) > 1:
            # mlda.py:23-33 -> metropolisHastings.py:116 on the base level
            raise AttributeError("'MetropolisedRandomWalk' object has no attribute 'set_state'")
        super().run(chainLength, initialState, verbose)

    def _record_flags(self):
        flags = super()._record_flags()
        if self.recordSubChain:
            flags |= _ffi.RECORD_SUB_ACCEPT
        return flags

    def _make_problem(self, ctx, state_eq):
        dim = lowering.lower_density(self._tgtDensity, self.rk4Steps).desc.dim
        sigma = lowering.proposal_sigma(self._proposalCov, dim)
        if lowering.proposal_chol(self._proposalCov, dim) is not None:
            raise lowering.UnsupportedPlugin("a dense base proposal covariance is lowered for the random-walk "
                                             "sampler only")
        if self.nSurrogates == 1:
            return lowering.make_problem(ctx, _ffi.METHOD_MLDA2, self._tgtDensity, sigma, state_eq,
                                         surrogate=self._surrogates[0], sub_chain_length=self._nSteps[0],
                                         rk4_steps=self.rk4Steps)
        return lowering.make_nested_problem(ctx, self._tgtDensity, self._surrogates, self._nSteps, sigma,
                                            state_eq, rk4_steps=self.rk4Steps)


class AdaptiveErrorModel(MLDA):
    """chain/method/aem.py:6-58 with one surrogate: two-level delayed acceptance whose coarse
    AEMLikelihood learns the model error from the accepted fine proposals, per chain
    (YMC_METHOD_AEM2).  With more surrogates the reference fails at the first accepted step
    (`d_model_evaluation` does not exist); the same AttributeError is raised here."""

    def run(self, chainLength, initialState, verbose=True):
        if self.nSurrogates > 1 and int(chainLength) > 1:
            raise AttributeError("'AEMLikelihood' object has no attribute 'd_model_evaluation'")
        super().run(chainLength, initialState, verbose)

    def error_model(self):
        """dict n / mean / m2 / precision / has_precision, one row per chain."""
        return self._run.error_model()

    def _ingest(self):
        super()._ingest()
        lik = getattr(self._surrogates[0], "likelihood", None)
        acc = getattr(lik, "accumulator", None)
        if hasattr(acc, "_attach"):
            acc._attach(self._run.error_model())

    def _make_problem(self, ctx, state_eq):
        dim = lowering.lower_density(self._tgtDensity, self.rk4Steps).desc.dim
        sigma = lowering.proposal_sigma(self._proposalCov, dim)
        aem = lowering.aem_parameters(self._surrogates[0].likelihood)
        return lowering.make_problem(ctx, _ffi.METHOD_AEM2, self._tgtDensity, sigma, state_eq,
                                     surrogate=self._surrogates[0], sub_chain_length=self._nSteps[0], aem=aem)


class PreconditionedCrankNicolson(MetropolisHastingsEnsemble):
    """chain/method/pcn.py:38-57: proposal sqrt(1-2h) x + sqrt(2h) xi with xi drawn from the
    (centred, Gaussian) prior; the target of the accept/reject step is the likelihood."""

    def __init__(self, targetDensity, prior, stepSize, diagnostics, **ensemble):
        assert 0 < stepSize and stepSize <= 0.5
        if not lowering._has(prior, "mean", "covariance"):
            raise NotImplementedError("PCN only supports Gaussian priors")
        if np.any(np.asarray(prior.mean.coefficient) != 0.0):
            raise ValueError("Preconditioned Crank Nicholson requires centred prior")
        super().__init__(targetDensity, prior.covariance, diagnostics, **ensemble)
        self._prior = prior
        self._stepSize = float(stepSize)

    def _make_problem(self, ctx, state_eq):
        low = lowering.lower_likelihood(self._tgtDensity, self._prior, self.rk4Steps)
        sigma = lowering.proposal_sigma(self._prior.covariance, low.desc.dim)
        return lowering.make_problem(ctx, _ffi.METHOD_PCN, low, sigma, state_eq, pcn_step_size=self._stepSize)
