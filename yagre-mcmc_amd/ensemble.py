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
                 chainOffset=0, recordTrajectory=True, recordLogPosterior=False, rk4Steps=None):
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
        self.rk4Steps = rk4Steps
        self._run = None
        self._problem = None
        if self.nChains < 1:
            raise ValueError("nChains must be positive")

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
        run = _ffi.Run(self._problem, self.seed, self.nChains, x0, max(nsteps, 1), self._record_flags(),
                       self.chainOffset)
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

    def integrated_autocorrelation(self, method='mean', burnin=0, sokalConst=5.):
        """postprocessing/autocorrelation.py:92-140 for every chain, computed on the device from the
        record: the equivalent of `ac.integrated_autocorrelation(states[burnin:], method)` per chain.
        Returns an int for one chain, an array of nChains ints otherwise."""
        methods = {"mean": 0, "max": 1}
        if method not in methods:
            raise ValueError(f"Invalid IAT - Type: {method}. Options are 'mean' and 'max'.")
        if burnin == 0:
            # row 0 of the reference's trajectory is the initial state, which the device record
            # does not hold: go through the fetched states so that it is included
            states = self._chain.ensemble_trajectory
            out = _ffi.default_context(self.device).iat_states(states, methods[method], sokalConst)
        else:
            out = self._run.iat(burnin - 1, None, methods[method], sokalConst)
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
        flags = _ffi.RECORD_ACCEPT
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
        self._diagnostics._ingest(run.accept, run.summary(), run.steps_done()[0])

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
                                     rk4_steps=self.rk4Steps)


class MLDA(MetropolisHastingsEnsemble):
    """chain/method/mlda.py:120-154 restricted to ONE surrogate level: the
    proposal is the end point of a subChainLength-step random walk on the
    surrogate density (mlda.py:102-110); acceptance is mlda.py:146-154."""

    def __init__(self, targetDensity, surrogateDensities, baseProposalCov, nSteps, targetDiagnostics,
                 surrogateDiagnosticsList=None, *, recordSubChain=False, **ensemble):
        if len(surrogateDensities) != 1 or len(nSteps) != 1:
            raise lowering.UnsupportedPlugin(
                "the device path implements two-level delayed acceptance (one surrogate); "
                f"got {len(surrogateDensities)} surrogates")
        super().__init__(targetDensity, baseProposalCov, targetDiagnostics, **ensemble)
        self._surrogate = surrogateDensities[0]
        self._subChainLength = int(nSteps[0])
        self._surrogateDiagnostics = surrogateDiagnosticsList
        self.recordSubChain = bool(recordSubChain)

    @property
    def nSurrogates(self):
        return 1

    @property
    def sub_chain_decisions(self):
        """(N-1, S, K) outcomes of the coarse sub-steps (needs recordSubChain)."""
        return self._run.sub_accept()

    def _record_flags(self):
        flags = super()._record_flags()
        if self.recordSubChain:
            flags |= _ffi.RECORD_SUB_ACCEPT
        return flags

    def _make_problem(self, ctx, state_eq):
        dim = lowering.lower_density(self._tgtDensity, self.rk4Steps).desc.dim
        sigma = lowering.proposal_sigma(self._proposalCov, dim)
        return lowering.make_problem(ctx, _ffi.METHOD_MLDA2, self._tgtDensity, sigma, state_eq,
                                     surrogate=self._surrogate, sub_chain_length=self._subChainLength,
                                     rk4_steps=self.rk4Steps)


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
