"""The abstract protocols and small value types user code of the reference is written against
(yagremcmc/statistics/interface.py, parameter/interface.py, chain/interface.py, chain/transition.py,
chain/proposal.py, chain/chain.py, statistics/parameterLaw.py, statistics/estimation.py), so that
plug-ins defined against those module paths import unchanged.  A user-defined density or solver that
has no device lowering is refused by the builders (lowering.UnsupportedPlugin): these classes do not
add a CPU path."""
from abc import ABC, abstractmethod

import numpy as np


class ParameterInterface(ABC):
    """parameter/interface.py:4-37: a parameter with a fixed, read-only coefficient."""

    @property
    @abstractmethod
    def dimension(self):
        ...

    @property
    @abstractmethod
    def coefficient(self):
        ...

    @abstractmethod
    def evaluate(self):
        ...

    @abstractmethod
    def __eq__(self, other):
        ...

    @abstractmethod
    def clone_with(self, newCoefficient):
        ...


class DensityInterface(ABC):
    """statistics/interface.py:6-10"""

    @abstractmethod
    def evaluate_log(self, state):
        ...


class ParameterLawInterface(ABC):
    """statistics/interface.py:13-17"""

    @abstractmethod
    def generate_realisation(self):
        ...


class AbsolutelyContinuousParameterLaw(ParameterLawInterface):
    """statistics/parameterLaw.py:6-14"""

    @property
    @abstractmethod
    def density(self):
        ...


class CovarianceOperatorInterface(ABC):
    """statistics/interface.py:20-24"""

    @abstractmethod
    def apply_inverse(self, x):
        ...


class BayesianModelInterface(ABC):
    """statistics/interface.py:27-37"""

    @property
    @abstractmethod
    def likelihood(self):
        ...

    @property
    @abstractmethod
    def prior(self):
        ...


class NoiseModelInterface(ABC):
    """statistics/interface.py:40-44"""

    @abstractmethod
    def induced_norm_squared(self, vector):
        ...


class ChainDiagnostics:
    """chain/interface.py:4-16"""

    def process(self, transitionData):
        raise NotImplementedError

    def print_diagnostics(self, logger):
        raise NotImplementedError

    def clear(self):
        raise NotImplementedError


class TransitionData:
    """chain/transition.py:1-25: (state, proposal, outcome in {REJECTED, ACCEPTED})."""

    REJECTED = 0
    ACCEPTED = 1

    def __init__(self, state, proposal, outcome):
        if outcome not in (TransitionData.REJECTED, TransitionData.ACCEPTED):
            raise RuntimeError(f"invalid MC transition outcome: {outcome}")
        self._state, self._proposal, self._outcome = state, proposal, outcome

    @property
    def state(self):
        return self._state

    @property
    def proposal(self):
        return self._proposal

    @property
    def outcome(self):
        return self._outcome


class ProposalMethod(ABC):
    """chain/proposal.py:4-27"""

    def __init__(self):
        super().__init__()
        self._state = None
        self._stateType = None

    def get_state(self):
        return self._state

    def set_state(self, newState):
        self._state = newState
        self._stateType = type(newState)

    @property
    def stateType(self):
        return self._stateType

    @abstractmethod
    def generate_proposal(self):
        ...


class Chain:
    """chain/chain.py:4-21: the host-side list of states (the ensemble samplers use ensemble.EnsembleChain)."""

    def __init__(self):
        self._trajectory = []

    @property
    def trajectory(self):
        return self._trajectory

    @property
    def length(self):
        return len(self._trajectory)

    def append(self, stateVector):
        self._trajectory.append(stateVector)

    def clear(self):
        self._trajectory = []


class WelfordAccumulator:
    """statistics/estimation.py:4-58: streaming mean and unbiased marginal variance of vectors fed on
    the host (the chains' own moments are accumulated on the device, csrc/ymc_device.h ymc_welford)."""

    def __init__(self):
        self.reset()

    def mean(self):
        return self._mean

    def marginal_variance(self):
        if self._dataSize < 2:
            raise RuntimeError("Insufficient data for variance estimation.")
        return self._welfordM2 / (self._dataSize - 1)

    def condition_number(self):
        margVar = self.marginal_variance()
        if np.min(margVar) < 1e-12:
            raise RuntimeError("Singular marginal variance.")
        return np.max(margVar) / np.min(margVar)

    @property
    def nData(self):
        return self._dataSize

    def update(self, realisation):
        realisation = np.asarray(realisation, dtype=float)
        if self._mean is None:
            self._mean = np.zeros_like(realisation)
            self._welfordM2 = np.zeros_like(realisation)
        delta = realisation - self._mean
        self._dataSize += 1
        self._mean = self._mean + delta / self._dataSize
        self._welfordM2 = self._welfordM2 + delta * (realisation - self._mean)

    def reset(self):
        self._dataSize = 0
        self._mean = None
        self._welfordM2 = None
