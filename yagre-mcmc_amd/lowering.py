"""Lowering of plug-in objects to the C ABI's parameter blocks.

The builders accept the plug-in objects of `plugins.py` AND the reference's own
instances (matched by class name and attributes, never imported).  A recognised
object becomes a `ymc_density_desc`; anything else raises -- there is no hidden
CPU path for an unknown `DensityInterface` or `SolverInterface`
(SURVEY.md 8b).
"""
import numpy as np

from . import _ffi


class UnsupportedPlugin(NotImplementedError):
    pass


def _name(obj):
    return type(obj).__name__


def _has(obj, *attrs):
    return all(hasattr(obj, a) for a in attrs)


# Classes are recognised by NAME (the reference is never imported here), so the name alone must not be
# trusted with the arithmetic: a user class that happens to be called GaussianTargetDensity2d -- or a
# subclass of the stock one -- which brings its own evaluate_log would otherwise be lowered as the stock
# density, silently.  A recognised object is lowered only if every method that defines its arithmetic is
# the one the reference's package or this package defines (or is absent: a data-only stand-in).
_STOCK_PACKAGES = ("yagremcmc", __package__ or "yagremcmc_amd")


def _require_stock(obj, what, *methods):
    for name in methods:
        for klass in type(obj).__mro__:
            if name in vars(klass):
                top = (klass.__module__ or "").split(".")[0]
                if top not in _STOCK_PACKAGES:
                    raise UnsupportedPlugin(
                        f"{what} {_name(obj)}: its {name}() is defined by {klass.__module__}.{klass.__qualname__}, not "
                        "by the stock class of that name; a custom implementation has no device lowering "
                        "(and there is no CPU fallback)")
                break


def proposal_sigma(cov, dim):
    """sqrt(reciprocal(precision)) exactly as DiagonalCovarianceMatrix.apply_chol_factor
    forms it (statistics/covariance.py:51-52) -- not sqrt(variance)."""
    if cov is None:
        raise ValueError("Proposal Covariance not set")
    if hasattr(cov, "cholFactor_"):
        return np.sqrt(np.diag(np.asarray(cov.cholFactor_) @ np.asarray(cov.cholFactor_).T))  # unused by the dense path
    if not hasattr(cov, "_precision"):
        raise UnsupportedPlugin(
            f"proposal covariance {_name(cov)}: only Diagonal/IID covariances have a device lowering")
    _require_stock(cov, "proposal covariance", "apply_chol_factor")
    prec = np.asarray(cov._precision, dtype=np.float64).reshape(-1)
    if prec.size != dim:
        raise ValueError(f"proposal covariance has dimension {prec.size}, the state has {dim}")
    return np.sqrt(np.reciprocal(prec))


def proposal_chol(cov, dim):
    """The lower Cholesky factor of a DenseCovarianceMatrix proposal (covariance.py:69-94), or None."""
    L = getattr(cov, "cholFactor_", None)
    if L is None:
        return None
    _require_stock(cov, "proposal covariance", "apply_chol_factor")
    L = np.tril(np.asarray(L, dtype=np.float64))
    if L.shape != (dim, dim):
        raise ValueError(f"proposal covariance has dimension {L.shape[0]}, the state has {dim}")
    return np.ascontiguousarray(L)


def _diag_precision(cov, n, what):
    if not hasattr(cov, "_precision"):
        raise UnsupportedPlugin(f"{what} covariance {_name(cov)}: only Diagonal/IID covariances are lowered")
    _require_stock(cov, f"{what} covariance", "apply_inverse", "induced_norm_squared")
    prec = np.asarray(cov._precision, dtype=np.float64).reshape(-1)
    if prec.size != n:
        raise ValueError(f"{what} covariance has dimension {prec.size}, expected {n}")
    return prec


def _set_covariance(low, cov, n, what, prec_field, linv_field):
    """A noise / prior covariance on the descriptor: Diagonal / IID as its precision vector; a
    DenseCovarianceMatrix (covariance.py:69-94, recognised by its lower Cholesky factor `cholFactor_`) as the
    inverse of that factor, so that the device forms |L^-1 r|^2 (ymc_whitened_norm_sq; beyond the lane
    kernels' sizes the library whitens a linear problem with it once, ymc_lower_times)."""
    L = getattr(cov, "cholFactor_", None)
    if L is None:
        low.set_array(prec_field, _diag_precision(cov, n, what))
        return
    _require_stock(cov, f"{what} covariance", "apply_inverse", "induced_norm_squared", "apply_chol_factor")
    L = np.tril(np.asarray(L, dtype=np.float64))
    if L.shape != (n, n):
        raise ValueError(f"{what} covariance has dimension {L.shape[0]}, expected {n}")
    # beyond the lane kernels' sizes the library whitens a dense noise covariance away and carries a dense prior
    # covariance as extra outputs of a LINEAR forward map (include/ymc.h); it refuses what it cannot lower
    from scipy.linalg import solve_triangular
    low.set_array(linv_field, np.tril(solve_triangular(L, np.eye(n), lower=True)))
    low.set_array(prec_field, np.reciprocal(np.diag(L @ L.T)))   # unused by the dense path; keeps the field valid


class Lowered:
    """A ymc_density_desc plus the arrays it points into."""

    def __init__(self, kind, dim):
        self.desc = _ffi.DensityDesc()
        self.desc.kind, self.desc.dim = kind, dim
        self.keep = []

    def set_array(self, field, arr):
        arr = _ffi.f64(arr)
        self.keep.append(arr)
        setattr(self.desc, field, _ffi.dptr(arr))
        return arr


def _whitening_of(target):
    """(mean, U, log_norm) of a GaussianTargetDensity2d: ours, or the reference's
    (test/testSetup.py:32-40, scipy frozen multivariate_normal in `dist_`)."""
    if _has(target, "mean_", "whiten_", "log_norm_"):
        return target.mean_, target.whiten_, float(target.log_norm_)
    dist = target.dist_
    mean = np.asarray(dist.mean, dtype=np.float64)
    co = getattr(dist, "cov_object", None)
    if co is not None and _has(co, "_LP", "_log_pdet", "rank"):
        return mean, np.asarray(co._LP, dtype=np.float64), float(co.rank * np.log(2.0 * np.pi) + co._log_pdet)
    cov = np.asarray(dist.cov, dtype=np.float64)
    w, v = np.linalg.eigh(cov)
    return mean, v * np.sqrt(1.0 / w), float(mean.size * np.log(2.0 * np.pi) + np.sum(np.log(w)))


def rk4_steps_for_rtol(rtol):
    """The reference configures the accuracy of a Lotka-Volterra level through scipy's `rtol`
    (test/testSetup.py:69-70,124-126); the device integrates with fixed-step classical RK4, whose global
    error scales with h^4.  Anchored at north_star's 50 steps for the reference's default rtol 1e-4:
    steps = ceil(50 (1e-4 / rtol)^(1/4)), clamped to [8, 400] -- 1e-2 -> 16, 1e-4 -> 50, 1e-5 -> 89.
    So a hierarchy defined only by rtol (example_inference_lotkaVolterra_twoLevel.py: RK23 at 1e-2 under
    DOP853 at 1e-5) lowers to DIFFERENT levels."""
    rtol = float(rtol)
    if not (rtol > 0.0) or not np.isfinite(rtol):
        raise ValueError(f"rtol must be positive and finite, got {rtol}")
    return int(min(400, max(8, int(np.ceil(50.0 * (1e-4 / rtol) ** 0.25)))))


def _lv_steps(solver, override):
    if override:
        return int(override)
    explicit = getattr(solver, "rk4Steps_", None)
    if explicit:
        return int(explicit)
    rtol = getattr(solver, "_solverRTol", None)
    return rk4_steps_for_rtol(rtol) if rtol is not None else 50


# The extension point for new forward models ("Adding a forward model"): a solver class whose device
# kind is registered in csrc/ymc_kinds.def announces its lowering here.  Registered classes are matched by IDENTITY
# (the class object itself, not its name), so nothing else can pass for them; a subclass is a different class and has
# no lowering unless it registers one.
_SOLVER_LOWERINGS = {}


def register_solver(cls, lower, design_rows=None):
    """`lower(solver, low, rk4_steps) -> (dim, data_dim)` fills the forward-model fields of `low.desc` (kind, design,
    t_final, rk4_steps, ...) for instances of exactly `cls`.  `design_rows(solver) -> int`: one forward solve per
    design row (the evaluation then has shape (rows, data_dim)); None: one evaluation for all data rows."""
    if not isinstance(cls, type):
        raise TypeError("register_solver takes the solver CLASS")
    _SOLVER_LOWERINGS[cls] = (lower, design_rows)


def solver_design_rows(solver):
    """rows of the forward evaluation of a per-design-row solver, or None"""
    if _name(solver) == "LotkaVolterraSolver" and hasattr(solver, "x_"):
        return int(np.asarray(solver.x_).shape[0])
    entry = _SOLVER_LOWERINGS.get(type(solver))
    return int(entry[1](solver)) if entry and entry[1] else None


def lower_solver(solver, low, rk4_steps=None):
    """SolverInterface -> forward-map fields of the descriptor."""
    n = _name(solver)
    entry = _SOLVER_LOWERINGS.get(type(solver))
    if entry is not None:
        return entry[0](solver, low, rk4_steps)
    _require_stock(solver, "solver", "invoke", "interpolate", "evaluation", "flow__")
    if n == "ExampleLinearModelSolver" and _has(solver, "_A", "_b"):
        A = np.asarray(solver._A, dtype=np.float64)
        if A.ndim != 2:
            raise ValueError("linear forward map: A must be a matrix")
        low.desc.kind = _ffi.POSTERIOR_LINEAR
        low.set_array("fwd_matrix", A)
        low.set_array("fwd_shift", np.asarray(solver._b, dtype=np.float64).reshape(-1))
        return A.shape[1], A.shape[0]
    if n == "LotkaVolterraSolver" and _has(solver, "x_", "tBoundary_", "fixedParam_", "dataShape_"):
        design = np.asarray(solver.x_, dtype=np.float64)
        low.desc.kind = _ffi.POSTERIOR_LV
        low.set_array("design", design)
        low.desc.t_final = float(solver.tBoundary_[1])
        low.desc.alpha = float(solver.fixedParam_[0])
        low.desc.gamma = float(solver.fixedParam_[1])
        low.desc.rk4_steps = _lv_steps(solver, rk4_steps)
        if getattr(solver, "adaptive_", False) and not rk4_steps:
            # the adaptive integrator at the reference's own `rtol` (ymc_lv_dp45_solve; include/ymc.h, lv_rtol)
            rtol = getattr(solver, "_solverRTol", None)
            low.desc.lv_rtol = 1e-4 if rtol is None else float(rtol)
            low.desc.lv_atol = float(getattr(solver, "atol_", 1e-6))
            if not (low.desc.lv_rtol > 0.0):
                raise ValueError(f"rtol must be positive, got {low.desc.lv_rtol}")
        return 2, 2
    raise UnsupportedPlugin(
        f"solver {n} has no device lowering (known: ExampleLinearModelSolver, LotkaVolterraSolver"
        + "".join(f", {c.__name__}" for c in _SOLVER_LOWERINGS) + "); there is no CPU fallback -- a new forward model "
        "needs a device kind and lowering.register_solver (INTEGRATION.md)")


def lower_posterior(likelihood, prior, rk4_steps=None):
    """UnnormalisedPosterior(likelihood, prior) (chain/target.py:4-22) with an
    AdditiveGaussianNoiseLikelihood (statistics/likelihood.py:49-87) and a Gaussian prior."""
    if _name(likelihood) not in ("AdditiveGaussianNoiseLikelihood", "AEMLikelihood") \
            or not _has(likelihood, "_data", "_fwdModel", "_noiseModel"):
        raise UnsupportedPlugin(f"likelihood {_name(likelihood)} has no device lowering")
    if not _has(prior, "mean", "covariance"):
        raise UnsupportedPlugin(f"prior {_name(prior)} has no device lowering (a Gaussian law is needed)")
    _require_stock(likelihood, "likelihood", "evaluate_log", "compute_log_likelihood", "compute_residual",
                   "compute_residual_norm_squared", "query_log_likelihood")
    _require_stock(likelihood._fwdModel, "forward model", "evaluate")
    _require_stock(likelihood._noiseModel, "noise model", "induced_norm_squared")
    _require_stock(prior, "prior", "density")
    if hasattr(prior, "density"):
        _require_stock(prior.density, "prior density", "evaluate_log")
    low = Lowered(0, 0)
    dim, data_dim = lower_solver(likelihood._fwdModel.solver_, low, rk4_steps)
    data = np.asarray(likelihood._data.array, dtype=np.float64)
    if data.ndim != 2 or data.shape[1] != data_dim:
        raise ValueError(f"data has shape {data.shape}; the forward map has {data_dim} outputs")
    rows = solver_design_rows(likelihood._fwdModel.solver_)
    if rows is not None and data.shape[0] != rows:
        raise ValueError(f"{_name(likelihood._fwdModel.solver_)}: one data row per design row is required "
                         f"({data.shape[0]} data rows, {rows} design rows)")
    low.desc.dim = dim
    low.desc.n_data, low.desc.data_dim = data.shape
    low.set_array("data", data)
    noise = likelihood._noiseModel
    if _name(noise) == "AEMNoise":  # noise.py:25-39: the measurement noise it wraps is the initial model
        noise = noise._dataNoise
    _set_covariance(low, noise.covariance, data_dim, "noise", "noise_precision", "noise_chol_inv")
    pm = np.asarray(prior.mean.coefficient, dtype=np.float64).reshape(-1)
    if pm.size != dim:
        raise ValueError(f"prior mean has dimension {pm.size}, the forward map takes {dim}")
    low.set_array("prior_mean", pm)
    _set_covariance(low, prior.covariance, dim, "prior", "prior_precision", "prior_chol_inv")
    return low


def lower_density(density, rk4_steps=None):
    """DensityInterface -> Lowered."""
    n = _name(density)
    _require_stock(density, "density", "evaluate_log")
    if n == "GaussianTargetDensity2d" and (_has(density, "dist_") or _has(density, "whiten_")):
        mean, U, log_norm = _whitening_of(density)
        low = Lowered(_ffi.GAUSS_WHITENED, int(np.size(mean)))
        low.set_array("mean", np.asarray(mean).reshape(-1))
        low.set_array("whiten", U)
        low.desc.log_norm = log_norm
        return low
    if n == "GaussianTargetDensity1d" and _has(density, "mean_", "var_"):
        low = Lowered(_ffi.GAUSS_ISO1D, 1)
        low.set_array("mean", np.asarray(density.mean_.coefficient, dtype=np.float64).reshape(-1)[:1])
        low.desc.variance = float(density.var_)
        return low
    if n == "UnnormalisedPosterior" and _has(density, "likelihood", "prior"):
        return lower_posterior(density.likelihood, density.prior, rk4_steps)
    raise UnsupportedPlugin(
        f"density {n} has no device lowering (known: GaussianTargetDensity1d/2d, UnnormalisedPosterior over "
        "AdditiveGaussianNoiseLikelihood + Gaussian prior); there is no CPU fallback")


def same_lowering(a, b):
    """Do two Lowered densities describe the same function (every scalar and every array equal)?"""
    for name, ctype in _ffi.DensityDesc._fields_:
        va, vb = getattr(a.desc, name), getattr(b.desc, name)
        if ctype is _ffi._dp:
            continue
        if va != vb:
            return False
    if len(a.keep) != len(b.keep):
        return False
    return all(x.shape == y.shape and np.array_equal(x, y) for x, y in zip(a.keep, b.keep))


def warn_if_degenerate(target_low, surrogate_lows, what="surrogate"):
    """A surrogate identical to its target makes delayed acceptance pointless (every proposal that the
    coarse walk accepted is accepted again, at S + 1 density evaluations per step): most often the levels
    were meant to differ in a setting that has no device counterpart."""
    import warnings
    for i, sur in enumerate(surrogate_lows):
        if same_lowering(target_low, sur):
            warnings.warn(f"{what} {i} lowers to exactly the target density: delayed acceptance degenerates (every "
                          "fine step repeats the coarse decision). If the levels were meant to differ in a solver "
                          "setting, give it a device counterpart (Lotka-Volterra: each level its own 'rk4Steps' or "
                          "'rtol').", RuntimeWarning, stacklevel=3)


def state_eq_of(state):
    """ScalarParameter compares with math.isclose (parameter/scalar.py:39-44),
    ParameterVector exactly (parameter/vector.py:37-45)."""
    return _ffi.EQ_ISCLOSE if _name(state) == "ScalarParameter" else _ffi.EQ_EXACT


def lower_likelihood(likelihood, prior, rk4_steps=None):
    """The likelihood alone as a density (the target of pCN, pcn.py:75-78): the posterior's
    descriptor with likelihood_only set; the prior fields still describe the Gaussian prior."""
    low = lower_posterior(likelihood, prior, rk4_steps)
    low.desc.likelihood_only = 1
    return low


def nested_levels(target, surrogates, sub_chain_lengths):
    """The effective levels of MLDA with an even number n of surrogates as the reference executes it
    (mlda.py:12-73,100-116; include/ymc.h, YMC_METHOD_MLDA_NESTED): only every second level's accept
    step runs.  Returns (P, Q, steps): P of levels 0..E-2, Q of levels 1..E-1, steps of levels 0..E-2."""
    n = len(surrogates)
    E = n // 2 + 1
    P = [surrogates[2 * e] for e in range(E - 1)]
    Q = [surrogates[2 * e - 1] for e in range(1, E)]
    steps = [int(sub_chain_lengths[2 * e + 1]) for e in range(E - 1)]
    return P, Q, steps


def make_nested_problem(ctx, target, surrogates, sub_chain_lengths, sigma, state_eq=_ffi.EQ_EXACT, rk4_steps=None):
    tgt = lower_density(target, rk4_steps)
    P, Q, steps = nested_levels(target, surrogates, sub_chain_lengths)
    E = len(P) + 1
    if E > _ffi.MAX_LEVELS:
        raise UnsupportedPlugin(f"{len(surrogates)} surrogates make {E} nested levels; the device nest holds "
                                f"{_ffi.MAX_LEVELS}")
    lowP = [lower_density(p, rk4_steps) for p in P]
    lowQ = [lower_density(q, rk4_steps) for q in Q]
    warn_if_degenerate(tgt, [lower_density(s, rk4_steps) for s in surrogates])
    desc = _ffi.ProblemDesc()
    desc.method, desc.state_eq = _ffi.METHOD_MLDA_NESTED, state_eq
    desc.target = tgt.desc
    desc.n_levels = E
    arrP = (_ffi.DensityDesc * (E - 1))(*[l.desc for l in lowP])
    arrQ = (_ffi.DensityDesc * (E - 1))(*[l.desc for l in lowQ])
    arrS = (_ffi.C.c_int32 * (E - 1))(*steps)
    desc.level_target, desc.level_correction, desc.level_steps = arrP, arrQ, arrS
    sigma = _ffi.f64(sigma).reshape(-1)
    if sigma.size != tgt.desc.dim:
        raise ValueError(f"proposal covariance has dimension {sigma.size}, the target has {tgt.desc.dim}")
    desc.proposal_sigma = _ffi.dptr(sigma)
    return _ffi.Problem(ctx, desc, [tgt, lowP, lowQ, arrP, arrQ, arrS, sigma])


def aem_parameters(likelihood):
    """(minDataSize, useNoiseHeuristic) of an AEMLikelihood (likelihood.py:92-105, noise.py:27-39)."""
    if _name(likelihood) != "AEMLikelihood" or not _has(likelihood, "_minDataSize", "_noiseModel"):
        raise ValueError("the likelihood is not adaptive")
    return int(likelihood._minDataSize), bool(getattr(likelihood._noiseModel, "_useHeuristic", False))


def make_problem(ctx, method, target, sigma, state_eq=_ffi.EQ_EXACT, surrogate=None, sub_chain_length=0,
                 rk4_steps=None, pcn_step_size=0.0, aem=None, chol=None):
    tgt = lower_density(target, rk4_steps) if not isinstance(target, Lowered) else target
    desc = _ffi.ProblemDesc()
    desc.method, desc.state_eq = method, state_eq
    desc.target = tgt.desc
    keep = [tgt]
    if method in (_ffi.METHOD_MLDA2, _ffi.METHOD_AEM2):
        sur = lower_density(surrogate, rk4_steps) if not isinstance(surrogate, Lowered) else surrogate
        desc.surrogate = sur.desc
        desc.sub_chain_length = int(sub_chain_length)
        keep.append(sur)
        if method == _ffi.METHOD_MLDA2:
            warn_if_degenerate(tgt, [sur])
    sigma = _ffi.f64(sigma).reshape(-1)
    if sigma.size != tgt.desc.dim:
        raise ValueError(f"proposal covariance has dimension {sigma.size}, the target has {tgt.desc.dim}")
    desc.proposal_sigma = _ffi.dptr(sigma)
    desc.pcn_step_size = float(pcn_step_size)
    if chol is not None:
        if method != _ffi.METHOD_MRW:
            raise UnsupportedPlugin("a dense proposal covariance is lowered for the random-walk sampler only")
        chol = _ffi.f64(chol)
        desc.proposal_chol = _ffi.dptr(chol)
        keep.append(chol)
    if method == _ffi.METHOD_AEM2:
        desc.aem_min_data_size, desc.aem_use_noise_heuristic = int(aem[0]), int(bool(aem[1]))
    keep.append(sigma)
    return _ffi.Problem(ctx, desc, keep)


def density_only_problem(density, device=0):
    """A problem used only for ymc_logdensity_eval (DensityInterface.evaluate_log)."""
    n = _name(density)
    if n == "GaussianDensity":
        raise UnsupportedPlugin("a bare GaussianDensity is evaluated as part of a posterior only")
    if n == "AdditiveGaussianNoiseLikelihood":
        # the likelihood alone: a flat stand-in prior of the right dimension, never evaluated
        from .plugins import Gaussian, IIDCovarianceMatrix, ParameterVector
        probe = Lowered(0, 0)
        dim, _ = lower_solver(density._fwdModel.solver_, probe)
        low = lower_likelihood(density, Gaussian(ParameterVector(np.zeros(dim)), IIDCovarianceMatrix(dim, 1.0)))
        return make_problem(_ffi.default_context(device), _ffi.METHOD_MRW, low, np.ones(dim))
    low = lower_density(density)
    return make_problem(_ffi.default_context(device), _ffi.METHOD_MRW, low, np.ones(low.desc.dim))


def forward_only_problem(solver, device=0):
    """A problem used only for ymc_forward_eval (ForwardModel.evaluate)."""
    prob = getattr(solver, "_ymc_forward_problem", None)
    if prob is None:
        low = Lowered(0, 0)
        dim, data_dim = lower_solver(solver, low)
        low.desc.dim = dim
        rows = solver_design_rows(solver) or 1
        low.desc.n_data, low.desc.data_dim = rows, data_dim
        low.set_array("data", np.zeros((rows, data_dim)))
        low.set_array("noise_precision", np.ones(data_dim))
        low.set_array("prior_mean", np.zeros(dim))
        low.set_array("prior_precision", np.ones(dim))
        prob = solver._ymc_forward_problem = make_problem(
            _ffi.default_context(device), _ffi.METHOD_MRW, low, np.ones(dim))
    return prob
