"""CPU-side checks: the C-ABI library loads and exports every symbol
include/ymc.h declares; host logic (lowering, builder validation, diagnostics
pooling) behaves; nothing here launches a kernel."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ffi():
    import __graft_entry__ as entry
    entry.build()
    from yagremcmc_amd import _ffi
    return _ffi


def test_library_exports_every_declared_symbol(ffi):
    header = open(os.path.join(ROOT, "include", "ymc.h")).read()
    declared = set(re.findall(r"^(?:int|const char \*)\s*\*?(ymc_[a-z0-9_]+)\(", header, flags=re.M))
    assert len(declared) >= 30
    handle = ffi.lib()
    for name in sorted(declared):
        assert hasattr(handle, name), f"libymc.so does not export {name}"
    assert declared == set(ffi.EXPORTED_SYMBOLS)
    assert handle.ymc_abi_version() == ffi.ABI_VERSION


def test_struct_layouts_match_the_header(ffi):
    import ctypes as C
    # sizes computed from the header's field lists (LP64)
    assert C.sizeof(ffi.DensityDesc) == 8 + 2 * 8 + 16 + 8 + 7 * 8 + 24 + 8 + 2 * 8 + 2 * 8
    assert C.sizeof(ffi.ProblemDesc) == 8 + 2 * C.sizeof(ffi.DensityDesc) + 8 + 8 + 8 + 8 + 3 * 8 + 8
    assert C.sizeof(ffi.RunDesc) == 7 * 8
    assert C.sizeof(ffi.DeviceInfo) == 128 + 64 + 8 + 8


def test_no_gpu_means_loud_failure_not_fallback(ffi):
    """On a box without a GPU the MH path must raise, never compute on the CPU."""
    try:
        n = ffi.device_count()
    except ffi.YmcError as e:
        assert e.code == ffi.ERR_DEVICE
        n = 0
    if n > 0:
        pytest.skip("a GPU is present")
    from yagremcmc_amd import MRWBuilder
    from yagremcmc_amd.parameter.vector import ParameterVector
    from yagremcmc_amd.statistics.covariance import IIDCovarianceMatrix
    from yagremcmc_amd.test.testSetup import GaussianTargetDensity2d
    b = MRWBuilder()
    b.explicitTarget = GaussianTargetDensity2d(ParameterVector(np.zeros(2)), np.eye(2))
    b.proposalCovariance = IIDCovarianceMatrix(2, 1.0)
    with pytest.raises(ffi.YmcError):
        b.build_method().run(10, ParameterVector(np.zeros(2)), verbose=False)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing in the package may import, load or include it."""
    pkg = os.path.join(ROOT, "yagre-mcmc_amd")
    bad = re.compile(r"(^\s*(from|import)\s+oracle\b)|libymc_oracle|#include\s*[\"<][^\">]*oracle", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                assert not bad.search(open(os.path.join(dirpath, f)).read()), f"{f} uses the oracle"


def test_builder_validation_messages(ffi):
    """test_builder.py:7-33 and builder.py:42-50 / mlda.py:218-269"""
    from yagremcmc_amd import MLDABuilder, MRWBuilder
    b = MRWBuilder()
    b.bayesModel = object()
    with pytest.raises(ValueError, match="Proposal Covariance not set for MRW"):
        b.build_method()
    b = MRWBuilder()
    b.proposalCovariance = object()
    with pytest.raises(ValueError, match="Either bayesian model or explicit target"):
        b.build_method()
    b.bayesModel, b.explicitTarget = object(), object()
    with pytest.raises(ValueError, match="Only one of bayes model or explicit target"):
        b.build_method()
    m = MLDABuilder()
    with pytest.raises(ValueError, match="Coarse proposal covariance not set for MLDA"):
        m.build_method()
    m.baseProposalCovariance = object()
    with pytest.raises(ValueError, match="Subchain lengths not set for MLDA"):
        m.build_method()
    m.subChainLengths = [6]
    m.explicitTarget = object()
    with pytest.raises(ValueError, match="Surrogate targets not set for MLDA"):
        m.build_method()
    m.surrogateTargets = [object(), object()]
    with pytest.raises(ValueError, match="Number of sub-chain lengths does not match"):
        m.build_method()


def test_lowering_of_plugins_and_duck_typed_reference_objects(ffi):
    from yagremcmc_amd import lowering
    from yagremcmc_amd.plugins import (AdditiveGaussianNoiseLikelihood, CentredGaussianNoise, Data,
                                       DiagonalCovarianceMatrix, ExampleLinearModelSolver, ForwardModel,
                                       Gaussian, GaussianTargetDensity1d, GaussianTargetDensity2d,
                                       IIDCovarianceMatrix, LotkaVolterraSolver, ParameterVector,
                                       ScalarParameter, UnnormalisedPosterior)
    cov = np.array([[2.4, -0.5], [-0.5, 0.7]])
    t = GaussianTargetDensity2d(ParameterVector(np.array([1.0, 1.5])), cov)
    low = lowering.lower_density(t)
    assert low.desc.kind == ffi.GAUSS_WHITENED and low.desc.dim == 2
    # U U^T = Sigma^-1 and log_norm = d log 2pi + log det
    assert np.allclose(t.whiten_ @ t.whiten_.T, np.linalg.inv(cov))
    assert np.isclose(low.desc.log_norm, 2 * np.log(2 * np.pi) + np.log(np.linalg.det(cov)))
    low = lowering.lower_density(GaussianTargetDensity1d(ScalarParameter(np.array([1.5])), 2.0))
    assert low.desc.kind == ffi.GAUSS_ISO1D and low.desc.variance == 2.0
    # sigma is sqrt(reciprocal(reciprocal(var))), not sqrt(var)
    var = np.array([0.1, 0.7, 3.3])
    assert np.array_equal(lowering.proposal_sigma(DiagonalCovarianceMatrix(var), 3),
                          np.sqrt(np.reciprocal(np.reciprocal(var))))
    with pytest.raises(ValueError):
        lowering.proposal_sigma(IIDCovarianceMatrix(2, 1.0), 3)
    # posterior over the linear solver
    A, b = np.arange(6.0).reshape(3, 2), np.ones(3)
    like = AdditiveGaussianNoiseLikelihood(Data(np.zeros((4, 3))), ForwardModel(ExampleLinearModelSolver(A, b)),
                                           CentredGaussianNoise(IIDCovarianceMatrix(3, 0.5)))
    prior = Gaussian(ParameterVector(np.zeros(2)), IIDCovarianceMatrix(2, 5.0))
    low = lowering.lower_density(UnnormalisedPosterior(like, prior))
    assert (low.desc.kind, low.desc.dim, low.desc.n_data, low.desc.data_dim) == (ffi.POSTERIOR_LINEAR, 2, 4, 3)
    # Lotka-Volterra
    cfg = {'T': 6., 'alpha': 0.8, 'gamma': 0.4, 'nData': 5, 'dataDim': 2, 'solver': 'LSODA', 'rtol': 1e-4}
    like = AdditiveGaussianNoiseLikelihood(Data(np.zeros((5, 2))), ForwardModel(LotkaVolterraSolver(np.ones((5, 2)), cfg)),
                                           CentredGaussianNoise(IIDCovarianceMatrix(2, 0.05)))
    low = lowering.lower_density(UnnormalisedPosterior(like, prior))
    assert (low.desc.kind, low.desc.rk4_steps, low.desc.t_final) == (ffi.POSTERIOR_LV, 50, 6.0)
    # objects of the reference's own classes are recognised by name + attributes
    RefCov = type("IIDCovarianceMatrix", (), {})
    rc = RefCov()
    rc._precision = np.reciprocal(np.full(2, 0.25))
    assert np.array_equal(lowering.proposal_sigma(rc, 2), np.full(2, 0.5))
    RefT1 = type("GaussianTargetDensity1d", (), {})
    r1 = RefT1()
    r1.mean_, r1.var_ = ScalarParameter(np.array([0.5])), 1.5
    assert lowering.lower_density(r1).desc.variance == 1.5
    # anything else raises: no hidden CPU path
    with pytest.raises(lowering.UnsupportedPlugin):
        lowering.lower_density(object())
    # a dense covariance (the reference's object, recognised by its Cholesky factor) lowers to L
    Dense = type("DenseCovarianceMatrix", (), {"cholFactor_": np.array([[2.0, 9.0], [0.5, 1.0]])})
    assert np.array_equal(lowering.proposal_chol(Dense(), 2), np.array([[2.0, 0.0], [0.5, 1.0]]))
    assert lowering.proposal_chol(rc, 2) is None
    with pytest.raises(ValueError):
        lowering.proposal_chol(Dense(), 3)
    Other = type("SomeCovariance", (), {})
    with pytest.raises(lowering.UnsupportedPlugin):
        lowering.proposal_sigma(Other(), 2)
    assert lowering.state_eq_of(ScalarParameter(np.array([1.0]))) == ffi.EQ_ISCLOSE
    assert lowering.state_eq_of(ParameterVector(np.array([1.0, 2.0]))) == ffi.EQ_EXACT


def test_recognised_names_with_custom_arithmetic_are_refused():
    """Plug-ins are matched by class name: a class of a known name -- or a subclass of the stock class -- that
    brings its own evaluate_log / invoke / apply_inverse must not be lowered as the stock object."""
    from yagremcmc_amd import lowering
    from yagremcmc_amd import plugins as P
    mean, cov = P.ParameterVector(np.array([1.0, 1.5])), np.array([[2.4, -0.5], [-0.5, 0.7]])

    class GaussianTargetDensity2d(P.GaussianTargetDensity2d):      # same name, tempered density
        def evaluate_log(self, parameter):
            return 0.5 * super().evaluate_log(parameter)

    with pytest.raises(lowering.UnsupportedPlugin, match="evaluate_log"):
        lowering.lower_density(GaussianTargetDensity2d(mean, cov))

    class Renamed(P.GaussianTargetDensity2d):                      # a plain subclass is not a known name
        pass

    with pytest.raises(lowering.UnsupportedPlugin):
        lowering.lower_density(Renamed(mean, cov))

    class ExampleLinearModelSolver(P.ExampleLinearModelSolver):    # same name, different forward map
        def invoke(self):
            raise RuntimeError("never called")

    A, b = np.arange(6.0).reshape(3, 2), np.ones(3)
    prior = P.Gaussian(P.ParameterVector(np.zeros(2)), P.IIDCovarianceMatrix(2, 5.0))
    noise = P.CentredGaussianNoise(P.IIDCovarianceMatrix(3, 0.5))
    like = P.AdditiveGaussianNoiseLikelihood(P.Data(np.zeros((4, 3))), P.ForwardModel(ExampleLinearModelSolver(A, b)), noise)
    with pytest.raises(lowering.UnsupportedPlugin, match="invoke"):
        lowering.lower_density(P.UnnormalisedPosterior(like, prior))

    class IIDCovarianceMatrix(P.IIDCovarianceMatrix):              # same name, different norm
        def apply_inverse(self, x):
            return 2.0 * x

    like = P.AdditiveGaussianNoiseLikelihood(P.Data(np.zeros((4, 3))), P.ForwardModel(P.ExampleLinearModelSolver(A, b)),
                                             P.CentredGaussianNoise(IIDCovarianceMatrix(3, 0.5)))
    with pytest.raises(lowering.UnsupportedPlugin, match="apply_inverse"):
        lowering.lower_density(P.UnnormalisedPosterior(like, prior))
    with pytest.raises(lowering.UnsupportedPlugin, match="apply_inverse"):
        lowering.lower_density(P.UnnormalisedPosterior(
            P.AdditiveGaussianNoiseLikelihood(P.Data(np.zeros((4, 3))), P.ForwardModel(P.ExampleLinearModelSolver(A, b)), noise),
            P.Gaussian(P.ParameterVector(np.zeros(2)), IIDCovarianceMatrix(2, 5.0))))
    # the stock classes themselves (and data-only stand-ins of the reference's) still lower
    assert lowering.lower_density(P.GaussianTargetDensity2d(mean, cov)).desc.dim == 2


def test_lotka_volterra_levels_defined_by_rtol_stay_different():
    """The reference's two-level Lotka-Volterra example defines its hierarchy ONLY through solver / rtol
    (RK23 at 1e-2 under DOP853 at 1e-5).  rtol maps to a fixed RK4 step count, so the levels lower to
    different densities; levels that do lower identically are reported."""
    import warnings
    from yagremcmc_amd import lowering
    from yagremcmc_amd import plugins as P
    assert [lowering.rk4_steps_for_rtol(r) for r in (1e-2, 1e-4, 1e-5, 1e-12, 10.0)] == [16, 50, 89, 400, 8]
    with pytest.raises(ValueError):
        lowering.rk4_steps_for_rtol(0.0)
    base = {'T': 10., 'alpha': 0.8, 'gamma': 0.4, 'nData': 4, 'dataDim': 2}
    design = np.linspace(0.6, 1.4, 8).reshape(4, 2)
    coarse = P.LotkaVolterraSolver(design, dict(base, solver='RK23', rtol=1e-2))
    fine = P.LotkaVolterraSolver(design, dict(base, solver='DOP853', rtol=1e-5))
    assert (coarse.rk4Steps_, fine.rk4Steps_) == (16, 89)
    assert P.LotkaVolterraSolver(design, dict(base, solver='LSODA', rtol=1e-2, rk4Steps=7)).rk4Steps_ == 7
    assert P.LotkaVolterraSolver(design, base).rk4Steps_ == 50
    prior = P.Gaussian(P.ParameterVector(np.zeros(2)), P.IIDCovarianceMatrix(2, 1.4))
    noise = P.CentredGaussianNoise(P.IIDCovarianceMatrix(2, 0.04))
    data = P.Data(np.ones((4, 2)))

    def posterior(solver):
        return P.UnnormalisedPosterior(P.AdditiveGaussianNoiseLikelihood(data, P.ForwardModel(solver), noise), prior)

    lo_c, lo_f = lowering.lower_density(posterior(coarse)), lowering.lower_density(posterior(fine))
    assert (lo_c.desc.rk4_steps, lo_f.desc.rk4_steps) == (16, 89) and not lowering.same_lowering(lo_c, lo_f)
    assert lowering.same_lowering(lo_f, lowering.lower_density(posterior(fine)))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        lowering.warn_if_degenerate(lo_f, [lo_c])                     # distinct levels: silent
    with pytest.warns(RuntimeWarning, match="degenerates"):
        lowering.warn_if_degenerate(lo_f, [lowering.lower_density(posterior(fine), None)])
    with pytest.warns(RuntimeWarning, match="degenerates"):           # a global rk4Steps override flattens the levels
        lowering.warn_if_degenerate(lowering.lower_density(posterior(fine), 30), [lowering.lower_density(posterior(coarse), 30)])
    # the reference's own solver object (matched by name) carries rtol as _solverRTol
    Ref = type("LotkaVolterraSolver", (), {})
    r = Ref()
    r.x_, r.tBoundary_, r.fixedParam_, r.dataShape_, r._solverRTol = design, (0., 10.), [0.8, 0.4], (4, 2), 1e-2
    low = lowering.Lowered(0, 0)
    lowering.lower_solver(r, low)
    assert low.desc.rk4_steps == 16


def test_parameter_objects_behave_like_the_reference():
    from yagremcmc_amd.plugins import ParameterVector, ScalarParameter
    a, b = ParameterVector(np.array([1.0, 2.0])), ParameterVector(np.array([1.0, 2.0]))
    assert a == b and a.dimension == 2 and not (a == ParameterVector(np.array([1.0, 2.0 + 1e-15])))
    assert a.clone_with(np.array([3.0, 4.0])).coefficient[0] == 3.0
    with pytest.raises(ValueError):
        a.clone_with([3.0, 4.0])
    s = ScalarParameter(np.array([1.0]))
    assert s == ScalarParameter(np.array([1.0 + 1e-12])) and s.dimension == 1
    with pytest.raises(Exception):
        ScalarParameter(1.0)


def test_welford_pooling_matches_numpy():
    """test_diagnostics.py:32-66 on the pooled (Chan) merge"""
    from yagremcmc_amd.diagnostics import merge_welford
    rng = np.random.default_rng(3)
    data = rng.normal(size=(7, 500, 3)) * np.array([1.0, 5.0, 0.1]) + np.array([0.0, 3.0, -2.0])
    n = np.full(7, 500)
    mean = data.mean(axis=1)
    m2 = ((data - mean[:, None]) ** 2).sum(axis=1)
    N, pm, pm2 = merge_welford(n, mean, m2)
    flat = data.reshape(-1, 3)
    assert N == 3500 and np.allclose(pm, flat.mean(axis=0)) and np.allclose(pm2 / (N - 1), flat.var(axis=0, ddof=1))


def test_module_paths_of_the_reference_resolve():
    """The module paths the reference's example scripts and inference tests import from exist under yagremcmc_amd."""
    import importlib
    for mod, names in {
        "chain.builder": ["ChainBuilder"], "chain.diagnostics": ["FullDiagnostics"],
        "chain.metropolisHastings": ["MetropolisHastings"], "chain.target": ["UnnormalisedPosterior"],
        "chain.method.mrw": ["MRWBuilder", "MetropolisedRandomWalk"],
        "chain.method.mlda": ["MLDABuilder", "MLDA"], "chain.method.pcn": ["PCNBuilder"],
        "chain.method.aem": ["AEMBuilder", "AdaptiveErrorModel"], "model.evaluation": ["EvaluationStatus"],
        "model.forwardModel": ["ForwardModel"], "model.interface": ["SolverInterface"],
        "parameter.scalar": ["ScalarParameter"],
        "parameter.vector": ["ParameterVector"], "postprocessing.autocorrelation": ["integrated_autocorrelation"],
        "statistics.bayesModel": ["BayesianRegressionModel"],
        "statistics.covariance": ["IIDCovarianceMatrix", "DiagonalCovarianceMatrix", "DenseCovarianceMatrix"],
        "statistics.data": ["Data"], "statistics.gaussian": ["Gaussian"],
        "statistics.likelihood": ["AdditiveGaussianNoiseLikelihood", "AEMLikelihood"],
        "statistics.modelHierarchy": ["BayesianRegressionModelHierarchy"],
        "statistics.noise": ["CentredGaussianNoise", "AEMNoise"],
        "test.testSetup": ["GaussianTargetDensity2d", "LotkaVolterraSolver"], "utility.hierarchy": ["Hierarchy"],
        "exampleSetup": ["ExampleLinearModelSolver"],
    }.items():
        m = importlib.import_module("yagremcmc_amd." + mod)
        for n in names:
            assert hasattr(m, n), f"{mod}.{n}"


def test_every_build_selection_switch_is_documented_and_swept():
    """The kernels' alternative builds are selected through YMC_* environment variables read in csrc/: each one is listed in
    INTEGRATION.md ("Which kernel runs") and -- unless it only exists for a test or an experiment harness -- appears in
    tools/env_sweep.sh, which runs the whole GPU suite under it (profiles/r04_env_sweep.txt)."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    for f in glob.glob(os.path.join(root, "yagre-mcmc_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "yagre-mcmc_amd", "csrc", "*.h")):
        names |= set(re.findall(r'getenv\("(YMC_[A-Z0-9_]+)"\)', open(f).read()))
    assert len(names) >= 15, names
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    sweep = open(os.path.join(root, "tools", "env_sweep.sh")).read()
    missing_doc = sorted(n for n in names if n not in doc)
    assert not missing_doc, f"not in INTEGRATION.md: {missing_doc}"
    not_swept_on_purpose = {"YMC_LV_POLL_LIMIT",      # test-only knob of the give-up paths
                            "YMC_SPLIT_NP"}           # swept together with YMC_SPLIT
    missing_sweep = sorted(n for n in names - not_swept_on_purpose if n not in sweep)
    assert not missing_sweep, f"not in tools/env_sweep.sh: {missing_sweep}"
