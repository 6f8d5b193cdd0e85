"""The worked example of an extension kind: a logistic-growth ODE solver as a second `SolverInterface`
(reference: model/interface.py:7-67; the shape of test/testSetup.py:61-141's LotkaVolterraSolver).

    dx/dt = r x (1 - x / K),  x(0) = x0_n for every design row n,  evaluation[n] = x(T)

The parameter's coefficient is theta = (log r, log K) (the exp parametrisation of LotkaVolterraParameter,
testSetup.py:47-58).  The device integrates with `rk4Steps` steps of classical RK4 (csrc/kinds/logistic.h; the kind's
line in csrc/ymc_kinds.def); this module is the Python half a contributor writes: the solver class the builders are
handed, and its lowering to the C ABI's descriptor.

    from yagremcmc_amd.kinds.logistic import LogisticGrowthSolver
    solver = LogisticGrowthSolver(x0, {'T': 3.0, 'rk4Steps': 40})
    likelihood = AdditiveGaussianNoiseLikelihood(data, ForwardModel(solver), noiseModel)     # data: (len(x0), 1)
"""
import numpy as np

from .. import _ffi, lowering
from ..plugins import _DeviceSolver


class LogisticGrowthSolver(_DeviceSolver):
    def __init__(self, design, config):
        super().__init__()
        self.x_ = np.asarray(design, dtype=float).reshape(-1)
        self.tBoundary_ = (0.0, float(config['T']))
        self.rk4Steps_ = int(config.get('rk4Steps', 50))
        if self.x_.size < 1 or not np.all(self.x_ > 0.0):
            raise ValueError("design: the initial populations must be positive")
        if not self.tBoundary_[1] > 0.0 or self.rk4Steps_ < 1:
            raise ValueError("T and rk4Steps must be positive")

    @property
    def dataShape(self):
        return (self.x_.size, 1)

    @property
    def nData(self):
        return self.x_.size

    @property
    def dataDim(self):
        return 1

    def _shape_evaluation(self, flat):
        return flat.reshape(self.x_.size, 1)

    def exact(self, theta):
        """the closed-form solution (a check of the integrator, not used by any sampler)"""
        r, k = np.exp(np.asarray(theta, dtype=float))
        return (k / (1.0 + (k / self.x_ - 1.0) * np.exp(-r * self.tBoundary_[1]))).reshape(-1, 1)


def _lower(solver, low, rk4_steps=None):
    low.desc.kind = _ffi.POSTERIOR_LOGISTIC
    low.set_array("design", solver.x_)
    low.desc.t_final = float(solver.tBoundary_[1])
    low.desc.rk4_steps = int(rk4_steps or solver.rk4Steps_)
    return 2, 1          # parameter dimension, forward outputs per design row


lowering.register_solver(LogisticGrowthSolver, _lower, design_rows=lambda s: s.x_.size)
