"""Host-side plug-in objects of the chain-ensemble path.

These are the value objects a user hands to the builders: parameters,
covariances, the Gaussian law, noise, data, likelihood, Bayesian model,
forward model + solvers, and the explicit test targets.  They carry the same
names, constructor arguments, properties and error behaviour as the reference's
(file:line cited per class) so its example scripts keep working after an
import change -- but they are *descriptions*: evaluating a density goes to the
device through the C ABI (`lowering.py`), never through a NumPy restatement.
The builders also accept the reference's own instances (duck-typed in
`lowering.py`), so a user who has yagremcmc installed need not switch types.
"""
import math
from enum import Enum, unique

import numpy as np


def _view(attribute, doc=None):
    """A read-only property over a private attribute: the value objects below expose their constructor
    arguments under the reference's property names and nothing more."""
    return property(lambda self: getattr(self, attribute), doc=doc)


# --------------------------------------------------------------------------
# parameters (yagremcmc/parameter/{interface,scalar,vector}.py)
# --------------------------------------------------------------------------

class _Parameter:
    """ParameterInterface, parameter/interface.py:4-37: an immutable coefficient."""

    def __init__(self, coefficient):
        self.coefficient_ = coefficient
        self._ctype = type(coefficient)

    @classmethod
    def from_coefficient(cls, coefficient):
        return cls(coefficient)

    @classmethod
    def from_value(cls, value):
        return cls(value)

    @property
    def coefficient(self):
        return self.coefficient_

    def evaluate(self):
        return self.coefficient_

    def clone_with(self, newCoefficient):
        if not isinstance(newCoefficient, self._ctype):
            raise ValueError("Trying to change coefficient type in cloning.")
        return type(self)(newCoefficient)

    __hash__ = None


class ScalarParameter(_Parameter):
    """parameter/scalar.py:5-51.  Equality is math.isclose (rel_tol 1e-9),
    which the device reproduces as YMC_EQ_ISCLOSE."""

    def __init__(self, coefficient):
        if isinstance(coefficient, float):
            raise Exception("scalar parameters must be 1-dimensional array types")
        super().__init__(coefficient)

    @property
    def dimension(self):
        return 1

    def __eq__(self, other):
        if isinstance(other, ScalarParameter):
            return math.isclose(self.coefficient_[0], other.coefficient[0])
        return NotImplemented


class ParameterVector(_Parameter):
    """parameter/vector.py:5-52.  Equality is exact (numpy.array_equal)."""

    @property
    def dimension(self):
        return self.coefficient_.size

    def __eq__(self, other):
        if isinstance(other, type(self)):
            return bool(np.array_equal(self.coefficient_, other.coefficient))
        return NotImplemented


class LotkaVolterraParameter(ParameterVector):
    """test/testSetup.py:47-58: coefficient = log of (beta, delta)."""

    @classmethod
    def from_interpolation(cls, value):
        return cls(np.log(value))

    def evaluate(self):
        return np.exp(self.coefficient_)


# --------------------------------------------------------------------------
# covariance operators (yagremcmc/statistics/covariance.py)
# --------------------------------------------------------------------------

class CovarianceMatrix:
    """covariance.py:8-22"""

    def induced_norm_squared(self, x):
        return np.dot(x, self.apply_inverse(x))


class DiagonalCovarianceMatrix(CovarianceMatrix):
    """covariance.py:25-55: held as `_precision`, the reciprocals of the marginal variances; every other
    quantity is derived from it on demand.  The proposal's standard deviations are
    sqrt(reciprocal(precision)) -- formed exactly like that in lowering.proposal_sigma so that
    x + sigma*xi rounds as in the reference."""

    def __init__(self, marginalVariances):
        self.marginalVariance = marginalVariances

    def _get_variance(self):
        return np.reciprocal(self._precision)

    def _set_variance(self, mVar):
        self._precision = np.reciprocal(np.asarray(mVar, dtype=float))

    marginalVariance = property(_get_variance, _set_variance)
    dimension = property(lambda self: self._precision.size)

    def apply_chol_factor(self, x):
        return np.sqrt(self._get_variance()) * x

    def apply_inverse(self, x):
        return self._precision * x


class IIDCovarianceMatrix(DiagonalCovarianceMatrix):
    """covariance.py:58-66"""

    def __init__(self, dimension, variance):
        super().__init__(np.full(dimension, variance, dtype=float))


class DenseCovarianceMatrix(CovarianceMatrix):
    """covariance.py:69-94: a dense covariance through its lower Cholesky factor.  Lowered to the device as
    a random-walk proposal covariance (y = x + L xi) and as a noise or prior covariance (|L^-1 r|^2), all up to
    dimension 8; the host methods mirror the reference's."""

    def __init__(self, denseCovMat):
        from scipy.linalg import cholesky
        C = np.asarray(denseCovMat, dtype=float)
        assert C.ndim == 2 and C.shape[0] == C.shape[1]
        self.dim_, self.cholFactor_ = C.shape[0], cholesky(C, lower=True)

    dimension = _view("dim_")

    def apply_chol_factor(self, x):
        return self.cholFactor_ @ x

    def apply_inverse(self, x):
        from scipy.linalg import solve_triangular
        L = self.cholFactor_
        return solve_triangular(L.T, solve_triangular(L, x, lower=True), lower=False)

    def dense(self):
        return self.cholFactor_ @ self.cholFactor_.T


# --------------------------------------------------------------------------
# Gaussian law, noise, data (statistics/{gaussian,noise,data}.py)
# --------------------------------------------------------------------------

class _DeviceDensity:
    """DensityInterface (statistics/interface.py:6-10) whose evaluate_log runs
    on the device: the object lowers itself to a ymc_density_desc and calls
    ymc_logdensity_eval."""

    def _device_problem(self):
        from . import lowering
        prob = getattr(self, "_ymc_problem", None)
        if prob is None:
            prob = self._ymc_problem = lowering.density_only_problem(self)
        return prob

    def evaluate_log(self, parameter):
        x = np.asarray(parameter.coefficient, dtype=float).reshape(1, -1)
        return float(self._device_problem().logdensity(x)[0][0])

    def evaluate_log_batch(self, coefficients):
        return self._device_problem().logdensity(coefficients)[0]


class GaussianDensity(_DeviceDensity):
    """gaussian.py:8-24: -0.5 |x - mean|^2 in the precision norm, no constant."""

    def __init__(self, meanVector, covariance):
        self._mean, self._cov = meanVector, covariance

    covariance = _view("_cov")


class Gaussian:
    """gaussian.py:27-66 (the prior law / the proposal law)."""

    def __init__(self, mean, covariance):
        self._mean, self._cov = mean, covariance
        self._density = GaussianDensity(mean.coefficient, covariance)

    mean, covariance, density = _view("_mean"), _view("_cov"), _view("_density")


class CentredGaussianNoise:
    """noise.py:8-22"""

    def __init__(self, covariance):
        if not (isinstance(covariance, CovarianceMatrix) or hasattr(covariance, "induced_norm_squared")):
            raise ValueError("CentredGaussianNoise need to be instantiated "
                             f"with CovarianceMatrix. Got: {type(covariance)}")
        self._cov = covariance

    covariance = _view("_cov")

    def induced_norm_squared(self, vector):
        return self._cov.induced_norm_squared(vector)


class Data:
    """data.py:4-23: array of shape (nData, dataDim)."""

    def __init__(self, dataArray):
        self.array_ = np.array(dataArray)
        self.size_, self.dim_ = dataArray.shape[:2]

    size, dim, array = _view("size_", "number of observations"), _view("dim_"), _view("array_")


# --------------------------------------------------------------------------
# forward models (yagremcmc/model/*, exampleSetup.py, test/testSetup.py)
# --------------------------------------------------------------------------

@unique
class EvaluationStatus(Enum):
    """model/evaluation.py:4-9"""
    NONE = -1
    SUCCESS = 0
    FAILURE = 1


class SolverInterface:
    """model/interface.py:7-67.  Only solvers with a device lowering can drive
    an ensemble (lowering.lower_solver); a custom Python solver raises there."""

    @property
    def status(self):
        raise NotImplementedError

    @property
    def evaluation(self):
        raise NotImplementedError

    def interpolate(self, parameter):
        raise NotImplementedError

    def invoke(self):
        raise NotImplementedError


class _DeviceSolver(SolverInterface):
    def __init__(self):
        self._theta = None
        self._evaluation = None
        self._status = EvaluationStatus.NONE

    @property
    def status(self):
        return self._status

    @property
    def evaluation(self):
        if self._status == EvaluationStatus.FAILURE:
            raise RuntimeError("Trying to retrieve failed evaluation.")
        if self._status == EvaluationStatus.NONE:
            raise RuntimeError("Trying to retrieve evaluation before it was performed.")
        return self._evaluation

    def interpolate(self, parameter):
        self._theta = np.asarray(parameter.coefficient, dtype=float)

    def invoke(self):
        from . import lowering
        out = lowering.forward_only_problem(self).forward(self._theta.reshape(1, -1))[0]
        self._evaluation = self._shape_evaluation(out)
        self._status = (EvaluationStatus.SUCCESS if np.all(np.isfinite(out))
                        else EvaluationStatus.FAILURE)

    def _shape_evaluation(self, flat):
        return flat


class ExampleLinearModelSolver(_DeviceSolver):
    """exampleSetup.py:8-52: G(theta) = A theta + b."""

    def __init__(self, A, b):
        super().__init__()
        self._A = np.asarray(A, dtype=float)
        self._b = np.asarray(b, dtype=float)


class LotkaVolterraSolver(_DeviceSolver):
    """test/testSetup.py:61-141.  `config` takes the reference's keys (T, alpha, gamma, nData, dataDim,
    solver, rtol) plus the optional 'rk4Steps' and 'adaptive': the ensemble integrates with fixed-step classical
    RK4 instead of scipy's adaptive solve_ivp.  Without 'rk4Steps' the step count follows from 'rtol'
    (lowering.rk4_steps_for_rtol: 50 steps at the reference's default 1e-4, fewer for a looser, more for a
    tighter tolerance), so levels that differ only in 'rtol' stay different; 'solver' (the name of a scipy
    method) has no counterpart and is recorded only.

    'adaptive': True integrates with the device's adaptive Dormand-Prince 5(4) pair at the configured 'rtol'
    (and 'atol', scipy's default 1e-6) instead: the choice for chains that visit parameters whose dynamics outrun
    a fixed step -- at the starting point of the reference's example, exp(theta) = (9e-4, 16.4), RK4 in 50 steps
    overflows (the chain would be flagged and every proposal compared with a non-finite value)."""

    def __init__(self, design, config):
        super().__init__()
        self.x_ = np.asarray(design, dtype=float)
        self.tBoundary_ = (0.0, float(config['T']))
        self.fixedParam_ = [float(config['alpha']), float(config['gamma'])]
        self.dataShape_ = (int(config['nData']), int(config['dataDim']))
        self._solverMethod = config.get('solver')
        self._solverRTol = config.get('rtol')
        self.adaptive_ = bool(config.get('adaptive', False))
        self.atol_ = float(config.get('atol', 1e-6))
        if self.adaptive_ and config.get('rk4Steps'):
            raise ValueError("'adaptive' and 'rk4Steps' exclude each other")
        if config.get('rk4Steps'):
            self.rk4Steps_ = int(config['rk4Steps'])
        else:
            from .lowering import rk4_steps_for_rtol
            self.rk4Steps_ = rk4_steps_for_rtol(self._solverRTol) if self._solverRTol is not None else 50
        if self.x_.shape != (self.dataShape_[0], 2) or self.dataShape_[1] != 2:
            raise ValueError("design must have shape (nData, 2) and dataDim must be 2")

    @property
    def dataShape(self):
        return self.dataShape_

    @property
    def nData(self):
        return self.dataShape_[0]

    @property
    def dataDim(self):
        return self.dataShape_[1]

    def _shape_evaluation(self, flat):
        return flat.reshape(self.dataShape_)

    # the reference's attribute names (testSetup.py:75-78, read by test_solver_setup.py / test_solver_invoke.py)
    @property
    def param_(self):
        return [None, None] if self._theta is None else list(np.exp(self._theta))

    @property
    def evaluation_(self):
        return self._evaluation

    @property
    def status_(self):
        return self._status

    def full_solution(self, parameter, y0):
        """testSetup.py:142-162: the trajectory from y0 under `parameter`, here on the integrator's own
        grid t_i = i T / rk4Steps (100 points with the adaptive integrator); every state is a forward solve on
        the device up to t_i.  Returns (t (n+1,), y (2, n+1))."""
        n = 100 if self.adaptive_ else self.rk4Steps_
        T = self.tBoundary_[1]
        t = np.linspace(0.0, T, n + 1)
        y = np.empty((2, n + 1))
        y[:, 0] = np.asarray(y0, dtype=float)
        cfg = {'alpha': self.fixedParam_[0], 'gamma': self.fixedParam_[1], 'nData': 1, 'dataDim': 2}
        if self.adaptive_:
            cfg.update(adaptive=True, rtol=self._solverRTol, atol=self.atol_)
        else:
            cfg['rk4Steps'] = self.rk4Steps_
        for i in range(1, n + 1):
            probe = LotkaVolterraSolver(np.asarray(y0, dtype=float).reshape(1, 2), dict(cfg, T=float(t[i])))
            probe.interpolate(parameter)
            probe.invoke()
            y[:, i] = probe.evaluation[0]
        return t, y


class ForwardModel:
    """model/forwardModel.py:4-19"""

    def __init__(self, solver):
        self.solver_ = solver

    def evaluate(self, parameter):
        self.solver_.interpolate(parameter)
        self.solver_.invoke()
        if self.solver_.status == EvaluationStatus.SUCCESS:
            return self.solver_.evaluation
        raise Exception("Evaluation request failed.")


def generate_synthetic_data(parameter, solver, noiseVar):
    """test/testSetup.py:165-174 (host-side helper; draws from NumPy's global generator)."""
    solver.interpolate(parameter)
    solver.invoke()
    return Data(solver.evaluation + np.sqrt(noiseVar) * np.random.standard_normal(solver.dataShape))


# --------------------------------------------------------------------------
# likelihood, Bayesian model, posterior
# --------------------------------------------------------------------------

class AdditiveGaussianNoiseLikelihood(_DeviceDensity):
    """likelihood.py:49-87.  The reference's FIFO evaluation cache (:51,61-72)
    has no counterpart: on the device logpi(state) stays in a register."""

    def __init__(self, data, forwardModel, noiseModel):
        if not (isinstance(noiseModel, CentredGaussianNoise)
                or type(noiseModel).__name__ == "CentredGaussianNoise"):
            raise ValueError("AdditiveGaussianNoiseLikelihood requires centred Gaussian noise.")
        self._data = data
        self._fwdModel = forwardModel
        self._noiseModel = noiseModel


class AEMNoise(CentredGaussianNoise):
    """noise.py:25-61: the measurement noise plus, once learned, the model-error variance.  The
    learning happens per chain on the device; this object holds the configuration."""

    def __init__(self, measurementNoise, useHeuristic):
        if not hasattr(measurementNoise.covariance, "_precision"):
            raise NotImplementedError("Currently, AEM is only implemented for independent measurement noise.")
        self._dataNoise = measurementNoise
        self._aemNoise = None
        self._useHeuristic = useHeuristic

    @property
    def covariance(self):
        return self._dataNoise.covariance

    def induced_norm_squared(self, vector):
        return self._dataNoise.induced_norm_squared(vector)


class ErrorModelAccumulator:
    """WelfordAccumulator surface (estimation.py:4-58) of the model error learned by the chains of
    the last run: reference shapes for one chain, a leading chain axis otherwise."""

    def __init__(self):
        self._state = None

    def _attach(self, state):
        self._state = state

    def _pick(self, a):
        return a[0] if a.shape[0] == 1 else a

    @property
    def nData(self):
        if self._state is None:
            return 0
        n = self._state["n"]
        return int(n[0]) if n.size == 1 else n

    def mean(self):
        return None if self._state is None else self._pick(self._state["mean"])

    def marginal_variance(self):
        if self._state is None or np.any(self._state["n"] < 2):
            raise RuntimeError("Insufficient data for variance estimation.")
        return self._pick(self._state["m2"] / (self._state["n"][:, None].astype(np.float64) - 1.0))


class AEMLikelihood(AdditiveGaussianNoiseLikelihood):
    """likelihood.py:90-155.  evaluate_log is the likelihood BEFORE any adaptation; inside an
    AdaptiveErrorModel run every chain adapts its own copy on the device (csrc/ymc_aem.hip) and
    `accumulator` shows what was learned."""

    def __init__(self, data, forwardModel, noiseModel, minDataSize, useNoiseHeuristic=False):
        if minDataSize < 2:
            raise ValueError("Smallest senisible data size for AEM is 2.")
        self._minDataSize = minDataSize
        self._accumulator = ErrorModelAccumulator()
        super().__init__(data, forwardModel, AEMNoise(noiseModel, useNoiseHeuristic))

    @property
    def accumulator(self):
        return self._accumulator


class BayesianRegressionModel:
    """bayesModel.py:5-29 (non-hierarchical components only)."""

    def __init__(self, likelihood, prior):
        self._likelihood, self._prior = likelihood, prior

    prior, likelihood = _view("_prior"), _view("_likelihood")


class HierarchyBase:
    """utility/hierarchy.py:5-31: levels 0 .. size-1, coarsest first; -1 addresses the finest."""

    def __init__(self, nLevels):
        if nLevels < 1:
            raise ValueError(f"Trying to set up hierarchy with {nLevels} levels.")
        self._nLevels = nLevels

    size = _view("_nLevels")

    def validate_level_index(self, idx):
        if not -1 <= idx < self._nLevels:
            raise ValueError(f"invalid level index. Trying to access level {idx} in a hierarchy of "
                             f"{self._nLevels} levels.")

    def level(self, lvlIdx):
        self.validate_level_index(lvlIdx)
        return self._component(lvlIdx)


class SharedComponent(HierarchyBase):
    """utility/hierarchy.py:34-46: the same component at every level"""

    def __init__(self, sharedComponent, nLevels):
        super().__init__(nLevels)
        self._sharedComponent = sharedComponent

    def _component(self, lvlIdx):
        return self._sharedComponent


class Hierarchy(HierarchyBase):
    """utility/hierarchy.py:49-63: one component per level, coarsest first"""

    def __init__(self, hierarchy):
        super().__init__(len(hierarchy))
        self._hierarchy = list(hierarchy)

    def _component(self, lvlIdx):
        return self._hierarchy[lvlIdx]


class BayesianRegressionModelHierarchy(Hierarchy):
    """statistics/modelHierarchy.py:6-53"""

    def __init__(self, likelihood, prior):
        for name, inst in (("prior", prior), ("likelihood", likelihood)):
            if not (isinstance(inst, HierarchyBase) or (hasattr(inst, "level") and hasattr(inst, "size"))):
                raise ValueError(f"Argument '{name}' must be derived from the Hierarchy base class. "
                                 f"Received type: {type(inst).__name__}.")
        if prior.size != likelihood.size:
            raise ValueError("Hierarchies have mismatched sizes. The following mismatches were found:\n"
                             f" - prior: size {prior.size}\n - likelihood: size {likelihood.size}")
        super().__init__([BayesianRegressionModel(likelihood.level(ell), prior.level(ell))
                          for ell in range(likelihood.size)])


class UnnormalisedPosterior(_DeviceDensity):
    """chain/target.py:4-22: log-likelihood + log prior density."""

    def __init__(self, likelihood, prior):
        self._likelihood, self._prior = likelihood, prior

    likelihood, prior = _view("_likelihood"), _view("_prior")


# --------------------------------------------------------------------------
# explicit targets (yagremcmc/test/testSetup.py)
# --------------------------------------------------------------------------

class GaussianTargetDensity1d(_DeviceDensity):
    """test/testSetup.py:15-29: -0.5 (mean - x)^2 / var."""

    def __init__(self, mean, var):
        self.mean_ = mean
        self.var_ = var

    def evaluate_on_mesh(self, mesh):
        # plotting helper of the examples (:27-29), not part of the chain
        return np.exp(-0.5 * np.square((self.mean_.coefficient - mesh) / self.var_))


class GaussianTargetDensity2d(_DeviceDensity):
    """test/testSetup.py:32-44: scipy.stats.multivariate_normal(mean, cov).logpdf.
    The whitening matrix and log pseudo-determinant are formed the way scipy's
    covariance object does (symmetric eigendecomposition, U = V / sqrt(w))."""

    def __init__(self, mean, cov):
        self.mean_ = np.array(mean.coefficient, dtype=float)
        self.cov_ = np.array(cov, dtype=float)
        d = self.mean_.size
        if self.cov_.shape != (d, d):
            raise ValueError("covariance shape does not match the mean")
        w, v = np.linalg.eigh(self.cov_)
        if np.min(w) <= 0.0:
            raise ValueError("the input matrix must be symmetric positive definite")
        self.whiten_ = v * np.sqrt(1.0 / w)
        self.log_norm_ = d * np.log(2.0 * np.pi) + np.sum(np.log(w))

    def evaluate_on_mesh(self, mesh):
        mesh = np.asarray(mesh, dtype=float)
        flat = mesh.reshape(-1, self.mean_.size)
        return np.exp(self.evaluate_log_batch(flat)).reshape(mesh.shape[:-1])
