"""Chain diagnostics of an ensemble (reference: yagremcmc/chain/diagnostics.py,
yagremcmc/statistics/estimation.py).

The numbers are produced on the device -- every accept decision, the per-chain
accept count and a per-chain Welford accumulator over the pre-transition
state -- and ingested here after a run.  For one chain the methods return
exactly what the reference's objects return; for K > 1 chains the same methods
give the pooled value and `per_chain_*` the individual ones.
"""
import numpy as np


def merge_welford(n, mean, m2):
    """Chan et al. pairwise merge of per-chain (n, mean, M2) into one pooled
    triple.  n: (K,), mean/m2: (K, d)."""
    n = np.asarray(n, dtype=np.float64)
    mean = np.asarray(mean, dtype=np.float64)
    m2 = np.asarray(m2, dtype=np.float64)
    total = n.sum()
    if total == 0:
        return 0, np.zeros(mean.shape[1:]), np.zeros(mean.shape[1:])
    pooled_mean = (n[:, None] * mean).sum(axis=0) / total
    pooled_m2 = m2.sum(axis=0) + (n[:, None] * np.square(mean - pooled_mean)).sum(axis=0)
    return int(total), pooled_mean, pooled_m2


class DummyDiagnostics:
    """diagnostics.py:8-16"""

    kind = "dummy"

    def process(self, transitionData):
        return

    def print_diagnostics(self, logger):
        return

    def reset(self):
        return

    def _ingest(self, fetch_decisions, summary, steps):
        return


class AcceptanceRateDiagnostics:
    """diagnostics.py:19-64.  The decision record stays on the device until it
    is asked for; the global rate comes from the per-chain accept counters."""

    kind = "acceptance"

    def __init__(self):
        self._fetch = None      # () -> (N-1, K) uint8
        self._matrix_cache = None
        self._accepted = None   # (K,) accepted transitions
        self._steps = 0
        self._lag = None

    @property
    def lag(self):
        return self._lag

    @lag.setter
    def lag(self, value):
        if value <= 0:
            raise ValueError("Lag must be a positive integer.")
        self._lag = value

    def _ingest(self, fetch_decisions, summary, steps):
        self._fetch = fetch_decisions
        self._matrix_cache = None
        self._accepted = np.asarray(summary["accepted"], dtype=np.uint64)
        self._steps = int(steps)

    @property
    def _matrix(self):
        if self._matrix_cache is None and self._fetch is not None:
            self._matrix_cache = self._fetch()
        return self._matrix_cache

    @property
    def _decisions(self):
        """One chain: the reference's list of 0/1 (diagnostics.py:22); K chains: (N-1, K)."""
        m = self._matrix
        if m is None:
            return []
        return [int(v) for v in m[:, 0]] if m.shape[1] == 1 else m

    def rolling_acceptance_rate(self):
        if not self._lag:
            raise RuntimeError("Lag for rolling acceptance rate not set.")
        if self._steps < self._lag:
            raise RuntimeError("Insufficient data for rolling acceptance rate.")
        m = self._matrix
        if m is None:
            raise RuntimeError("The per-transition decisions were not recorded (thinned record).")
        return np.sum(m[-self._lag:]) / (self._lag * m.shape[1])

    def global_acceptance_rate(self):
        if not self._steps:
            return 0.0
        return np.sum(self._accepted) / (self._steps * self._accepted.size)

    def per_chain_acceptance_rate(self):
        if not self._steps:
            return np.zeros(0)
        return self._accepted / self._steps

    def process(self, transitionData):
        """diagnostics.py:48-54 for stand-alone use on the host (one chain); a sampler run fills the
        object from the device record instead."""
        if self._fetch is not None:
            raise RuntimeError("these diagnostics are attached to a device run; reset() first")
        host = self._matrix_cache if self._matrix_cache is not None else np.zeros((0, 1), dtype=np.uint8)
        self._matrix_cache = np.concatenate([host, np.array([[int(transitionData.outcome)]], dtype=np.uint8)])
        self._steps += 1
        self._accepted = np.array([int(self._matrix_cache.sum())], dtype=np.uint64)

    def print_diagnostics(self, logger):
        try:
            logger.info(f"  - Rolling acceptance rate: {self.rolling_acceptance_rate():.4f}")
        except RuntimeError as e:
            logger.warning(f"  - Rolling acceptance rate unavailable: {e}")

    def reset(self):
        self._fetch = self._matrix_cache = self._accepted = None
        self._steps = 0


class FullDiagnostics:
    """diagnostics.py:67-107 with WelfordAccumulator (estimation.py:4-58)."""

    kind = "full"

    def __init__(self):
        self._diagnostics = AcceptanceRateDiagnostics()
        self._lag = None
        self._n = None      # (K,)
        self._mean = None   # (K, d)
        self._m2 = None     # (K, d)

    @property
    def lag(self):
        return self._lag

    @lag.setter
    def lag(self, lag):
        self._lag = lag

    def _ingest(self, fetch_decisions, summary, steps):
        self._diagnostics._ingest(fetch_decisions, summary, steps)
        self._n = np.asarray(summary["wf_n"], dtype=np.uint64)
        self._mean = np.asarray(summary["wf_mean"])
        self._m2 = np.asarray(summary["wf_m2"])

    def global_acceptance_rate(self):
        return self._diagnostics.global_acceptance_rate()

    def per_chain_acceptance_rate(self):
        return self._diagnostics.per_chain_acceptance_rate()

    def _pooled(self):
        if self._n is None:
            return 0, None, None
        if self._n.size == 1:
            return int(self._n[0]), self._mean[0], self._m2[0]
        return merge_welford(self._n, self._mean, self._m2)

    def mean(self):
        return self._pooled()[1]

    def marginal_variance(self):
        n, _, m2 = self._pooled()
        if n < 2:
            raise RuntimeError("Insufficient data for variance estimation.")
        return m2 / (n - 1)

    def condition_number(self):
        mv = self.marginal_variance()
        if np.min(mv) < 1e-12:
            raise RuntimeError("Singular marginal variance.")
        return np.max(mv) / np.min(mv)

    def per_chain_mean(self):
        return self._mean

    def per_chain_marginal_variance(self):
        if self._n is None or int(self._n.min()) < 2:
            raise RuntimeError("Insufficient data for variance estimation.")
        return self._m2 / (self._n[:, None].astype(np.float64) - 1.0)

    def process(self, transitionData):
        raise RuntimeError("ensemble diagnostics are filled from the device record, not per transition")

    def print_diagnostics(self, logger):
        self._diagnostics.lag = self._lag
        self._diagnostics.print_diagnostics(logger)
        logger.info(f"  - Estimated condition number: {self.condition_number():.4f}")

    def reset(self):
        self._diagnostics.reset()
        self._n = self._mean = self._m2 = None


def adopt(diagnostics):
    """Map a user-supplied diagnostics object (ours or the reference's, by class
    name) to the ensemble implementation of the same kind."""
    if diagnostics is None:
        return AcceptanceRateDiagnostics()
    if isinstance(diagnostics, (DummyDiagnostics, AcceptanceRateDiagnostics, FullDiagnostics)):
        return diagnostics
    name = type(diagnostics).__name__
    if name == "FullDiagnostics":
        return FullDiagnostics()
    if name == "AcceptanceRateDiagnostics":
        return AcceptanceRateDiagnostics()
    if name == "DummyDiagnostics":
        return DummyDiagnostics()
    raise ValueError(f"unknown diagnostics type {name}")
