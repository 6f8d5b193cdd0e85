"""The CPU restatement (oracle/) against the golden vectors the LIVE reference
produced (tests/golden/make_golden.py): accept decisions and trajectories
bit-exact, log-densities to 1e-12 relative, diagnostics bit-exact."""
import hashlib

import numpy as np
import pytest

from oracle import oracle as o

LP_RTOL = 1e-12


def _check(out, g, n, lp_rtol=LP_RTOL):
    assert np.array_equal(out["acc"], g["acc"])
    if "traj" in g:
        assert np.array_equal(out["traj"], g["traj"][1:])
    want = g["logpost"][1:]
    assert np.abs(out["logpost"] - want).max() <= lp_rtol * np.abs(want).max()
    rel = np.abs(out["logpost"] - want) / np.maximum(np.abs(want), 1e-300)
    assert rel.max() <= 1e-9


def test_f1_mrw_1d_scalar_parameter(golden):
    g = golden("f1_mrw_1d.npz")
    n = int(g["n_states"])
    e = o.Ensemble(o.gauss_iso1d(float(g["mean"]), float(g["variance"])), g["sigma"], g["x0"],
                   int(g["seed"]), eq_mode=o.EQ_ISCLOSE)
    out = e.advance(n - 1)
    _check(out, g, n)
    # FullDiagnostics: rate, Welford mean and unbiased variance (estimation.py:11-53)
    assert e.acc_count[0] / (n - 1) == float(g["acc_rate"])
    assert np.array_equal(e.wf_mean[0], g["wf_mean"].reshape(-1))
    assert np.array_equal(e.wf_m2[0] / (float(e.wf_n[0]) - 1.0), g["wf_var"].reshape(-1))


@pytest.mark.parametrize("name", ["f2_mrw_2d.npz", "f2b_mrw_2d_indep.npz"])
def test_f2_mrw_2d(golden, name):
    g = golden(name)
    n = int(g["n_states"])
    p = o.gauss_whitened(g["mean"], g["whiten"], float(g["log_norm"]))
    e = o.Ensemble(p, g["sigma"], g["x0"], int(g["seed"]))
    out = e.advance(n - 1, want_margin=True)
    _check(out, g, n)
    assert np.array_equal(e.acc_count / (n - 1), g["acc_rate"])
    assert np.array_equal(e.wf_mean, g["wf_mean"])
    assert np.array_equal(e.wf_m2 / (e.wf_n[:, None] - 1.0), g["wf_var"])
    # no decision in the fixture is closer than 1e-9 to the threshold
    assert float(g["min_margin"]) > 1e-9 and out["margin"].min() > 1e-9
    # two half-length advances == one full one (state carried exactly)
    e2 = o.Ensemble(p, g["sigma"], g["x0"], int(g["seed"]))
    a = e2.advance((n - 1) // 2)
    b = e2.advance(n - 1 - (n - 1) // 2)
    assert np.array_equal(np.concatenate([a["traj"], b["traj"]]), out["traj"])
    # whitening recomputed from the covariance agrees with scipy's to rounding
    p2 = o.gauss_from_cov(g["mean"], g["cov"])
    lp2, _ = o.logdensity(p2, g["traj"][:, 0, :])
    assert np.abs(lp2 - g["logpost"][:, 0]).max() <= 1e-12 * np.abs(g["logpost"]).max()


def test_f3_linear_d128(golden):
    g = golden("f3_linear_d128.npz")
    n = int(g["n_states"])
    p = o.posterior_linear(g["A"], g["b"], g["data"], g["noise_precision"], g["prior_mean"],
                           g["prior_precision"])
    e = o.Ensemble(p, g["sigma"], g["x0"], int(g["seed"]))
    out = e.advance(n - 1)
    _check(out, g, n)
    assert np.array_equal(out["traj"][:, :2, :], g["traj_head"][1:])
    full = np.concatenate([g["x0"][None], out["traj"]])
    assert hashlib.sha256(np.ascontiguousarray(full).tobytes()).hexdigest() == str(g["traj_sha256"])


@pytest.mark.parametrize("name", ["f22_linear_d192_m160.npz", "f22b_linear_d320_m48.npz"])
def test_f22_linear_d192_m160(golden, name):
    """d = 192, m = 160: beyond what one wave's registers and the LDS held up to round 2; d = 320: beyond the matrix-core
    kernels (the reference is shape-agnostic, exampleSetup.py:42-52)."""
    g = golden(name)
    n = int(g["n_states"])
    p = o.posterior_linear(g["A"], g["b"], g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
    e = o.Ensemble(p, g["sigma"], g["x0"], int(g["seed"]))
    out = e.advance(n - 1)
    _check(out, g, n)
    assert np.array_equal(out["traj"][:, :2, :], g["traj_head"][1:])
    full = np.concatenate([g["x0"][None], out["traj"]])
    assert hashlib.sha256(np.ascontiguousarray(full).tobytes()).hexdigest() == str(g["traj_sha256"])


def test_f3b_linear_2x2_example(golden):
    g = golden("f3b_linear_2x2.npz")
    n = int(g["n_states"])
    p = o.posterior_linear(g["A"], g["b"], g["data"], g["noise_precision"], g["prior_mean"],
                           g["prior_precision"])
    lp, st = o.logdensity(p, g["grid"])
    assert (st == 0).all()
    assert np.abs(lp - g["grid_logpost"]).max() <= 1e-12 * np.abs(g["grid_logpost"]).max()
    e = o.Ensemble(p, g["sigma"], g["x0"], int(g["seed"]))
    _check(e.advance(n - 1), g, n)


def test_f4_lotka_volterra(golden):
    g = golden("f4_lotka_volterra.npz")
    n = int(g["n_states"])
    p = o.posterior_lv(g["design"], float(g["t_final"]), float(g["alpha"]), float(g["gamma"]),
                       int(g["rk4_steps"]), g["data"], g["noise_precision"], g["prior_mean"],
                       g["prior_precision"])
    # forward map: RK4-50 against the reference's own solver (LSODA rtol 1e-4,
    # test_solver_invoke.py:64-94 uses rtol 1e-3) and a tight DOP853 solution
    f = o.forward(p, g["thetas"]).reshape(-1, 5, 2)
    assert np.allclose(f, g["forward_lsoda_rtol1e4"], rtol=1e-3, atol=5e-3)
    assert np.abs(f - g["forward_dop853_rtol1e12"]).max() < 1e-4
    # loop parity: the reference loop ran with this RK4 as its solver plug-in
    e = o.Ensemble(p, g["sigma"], g["x0"], int(g["seed"]))
    out = e.advance(n - 1)
    _check(out, g, n)
    assert (e.status == 0).all()


def test_f4_nonfinite_forward_is_rejected_and_flagged(golden):
    g = golden("f4_lotka_volterra.npz")
    p = o.posterior_lv(g["design"], 10.0, 0.8, 0.4, 50, g["data"], g["noise_precision"],
                       g["prior_mean"], g["prior_precision"])
    # the example's stiff start (example_inference_lotkaVolterra_singleLevel.py:106)
    lp, st = o.logdensity(p, np.array([[-7.0, 2.8]]))
    assert st[0] & o.STATUS_NONFINITE and lp[0] == -np.inf


def _f24_problem(g, rtol=None):
    return o.posterior_lv(g["design"], float(g["t_final"]), float(g["alpha"]), float(g["gamma"]), 50, g["data"],
                          g["noise_precision"], g["prior_mean"], g["prior_precision"],
                          rtol=float(g["rtol"]) if rtol is None else rtol, atol=float(g["atol"]))


def test_f24_adaptive_lotka_volterra_from_the_examples_own_start(golden):
    """example_inference_lotkaVolterra_singleLevel.py:29-106 as shipped, from theta = (-7, 2.8): the adaptive
    integrator (ymc_lv_dp45_solve at the example's rtol 1e-4) is as close to a tight DOP853 solution as the
    reference's LSODA at that rtol -- also at the start, where RK4 in 50 steps overflows --, and the reference's LOOP
    run with it as its solver plug-in is reproduced decision for decision; no chain is flagged and all of them reach
    the posterior bulk."""
    g = golden("f24_lotka_volterra_adaptive.npz")
    n = int(g["n_states"])
    p = _f24_problem(g)
    f = o.forward(p, g["thetas"]).reshape(-1, 10, 2)
    tight, lsoda = g["forward_dop853_rtol1e12"], g["forward_lsoda_rtol1e4"]
    rel = lambda a: (np.abs(a - tight) / (1e-3 + np.abs(tight))).max()
    assert rel(f) < 1e-2 and rel(f) < 1.5 * rel(lsoda)
    # error follows the tolerance (down to atol = 1e-6 on the small components)
    assert rel(o.forward(_f24_problem(g, 1e-7), g["thetas"]).reshape(-1, 10, 2)) < 1e-3 * 1.0
    assert rel(o.forward(_f24_problem(g, 1e-10), g["thetas"]).reshape(-1, 10, 2)) < 1e-3
    e = o.Ensemble(p, g["sigma"], g["x0"], int(g["seed"]))
    out = e.advance(n - 1)
    _check(out, g, n)
    assert (e.status == 0).all()
    assert np.abs(out["traj"][-1] - np.log([0.4, 0.6])).max() < 0.5
    # the fixed-step integrator at the same start: non-finite, flagged
    rk4 = o.posterior_lv(g["design"], 10.0, 0.8, 0.4, 50, g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
    lp, st = o.logdensity(rk4, g["x0"][:1])
    assert st[0] & o.STATUS_NONFINITE and lp[0] == -np.inf


def test_adaptive_lotka_volterra_gives_up_loudly():
    """A right-hand side the step cannot follow (rates of e^40) ends the solve: NaN forward values, the density
    -inf and the evaluation flagged -- never a silently wrong number, never an endless loop."""
    design = np.array([[1.0, 1.0], [0.5, 1.5]])
    p = o.posterior_lv(design, 10.0, 0.8, 0.4, 50, np.ones((2, 2)), np.ones(2), np.zeros(2), np.ones(2), rtol=1e-6)
    f = o.forward(p, np.array([[40.0, 40.0]]))
    assert np.isnan(f).all()
    lp, st = o.logdensity(p, np.array([[40.0, 40.0], [0.0, 0.0]]))
    assert lp[0] == -np.inf and st[0] & o.STATUS_NONFINITE and np.isfinite(lp[1]) and st[1] == 0


def test_f5_two_level_delayed_acceptance(golden):
    g = golden("f5_mlda2_2d.npz")
    n, S = int(g["n_states"]), int(g["S"])
    fine = o.gauss_whitened(g["fine_mean"], g["fine_whiten"], float(g["fine_log_norm"]))
    coarse = o.gauss_whitened(g["coarse_mean"], g["coarse_whiten"], float(g["coarse_log_norm"]))
    e = o.Ensemble(fine, g["sigma"], g["x0"], int(g["seed"]), surrogate=coarse, sub_chain_length=S)
    out = e.advance(n - 1, want_sub=True)
    _check(out, g, n)
    assert np.array_equal(out["sub_acc"], g["sub_acc"])


def _nested_from_fixture(g):
    tgt = o.gauss_whitened(g["tgt_mean"], g["tgt_whiten"], float(g["tgt_log_norm"]))
    surs = [o.gauss_whitened(g["sur_mean"][i], g["sur_whiten"][i], float(g["sur_log_norm"][i]))
            for i in range(g["sur_mean"].shape[0])]
    return o.nested_levels(surs, [int(v) for v in g["sub_lengths"]], tgt)


@pytest.mark.parametrize("name", ["f10_mlda_nested2.npz", "f11_mlda_nested4.npz", "f23_mlda_nested8.npz",
                                  "f23b_mlda_nested14.npz"])
def test_f10_f11_even_surrogate_hierarchies_as_the_reference_runs_them(golden, name):
    """example_mcmc_2d_hierarchical.py (2 surrogates), test_mlda.py:189-256 (4 surrogates) and deeper than the
    reference's own cases (8 and 14 surrogates: 5 and 8 effective levels): only every second level decides;
    chains, decisions and per-level accept counts == the reference's."""
    g = golden(name)
    n = int(g["n_states"])
    P, Q, steps = _nested_from_fixture(g)
    e = o.NestedEnsemble(P, Q, steps, g["sigma"], g["x0"], int(g["seed"]))
    assert e.cost_top == int(g["cost_top"])
    out = e.advance(n - 1)
    _check(out, g, n)
    assert np.array_equal(e.level_acc, g["level_acc"])
    # resumable: two halves == one piece
    e2 = o.NestedEnsemble(P, Q, steps, g["sigma"], g["x0"], int(g["seed"]), nthreads=3)
    a, b = e2.advance(n // 3), e2.advance(n - 1 - n // 3)
    assert np.array_equal(np.concatenate([a["traj"], b["traj"]]), out["traj"])
    assert np.array_equal(e2.level_acc, e.level_acc)


def test_nested_with_one_level_pair_equals_two_level_scheme_when_q_is_p0(golden):
    """E = 2 with Q = P[0] is exactly the one-surrogate scheme (same draws by address)."""
    g = golden("f5_mlda2_2d.npz")
    fine = o.gauss_whitened(g["fine_mean"], g["fine_whiten"], float(g["fine_log_norm"]))
    coarse = o.gauss_whitened(g["coarse_mean"], g["coarse_whiten"], float(g["coarse_log_norm"]))
    S = int(g["S"])
    two = o.Ensemble(fine, g["sigma"], g["x0"][:8], 77, surrogate=coarse, sub_chain_length=S).advance(150)
    nest = o.NestedEnsemble([coarse, fine], [None, coarse], [S], g["sigma"], g["x0"][:8], 77).advance(150)
    assert np.array_equal(two["traj"], nest["traj"]) and np.array_equal(two["acc"], nest["acc"])


def _aem_from_fixture(g, **kw):
    common = dict(data=g["data"], noise_precision=g["noise_precision"], prior_mean=g["prior_mean"],
                  prior_precision=g["prior_precision"])
    fine = o.posterior_linear(g["A_fine"], g["b_fine"], **common)
    coarse = o.posterior_linear(g["A_coarse"], g["b_coarse"], **common)
    return o.AemEnsemble(fine, coarse, g["sigma"], g["x0"], int(g["seed"]), int(g["S"]), int(g["min_data_size"]),
                         int(g["use_noise_heuristic"]), **kw)


@pytest.mark.parametrize("name", ["f12_aem_linear.npz", "f12b_aem_linear_noheuristic.npz"])
def test_f12_adaptive_error_model(golden, name):
    """chain/method/aem.py on the linear two-level models: chains, decisions at both levels and the
    learned error model (count, mean, unbiased variance, inflated noise precision) == the reference's,
    including the reference's use of stale cached log-likelihoods (AEMCache)."""
    g = golden(name)
    n = int(g["n_states"])
    e = _aem_from_fixture(g)
    out = e.advance(n - 1, want_sub=True)
    _check(out, g, n)
    assert np.array_equal(out["sub_acc"], g["sub_acc"])
    assert np.array_equal(e.aem_n, g["err_n"]) and e.aem_n.min() > int(g["min_data_size"])
    assert np.allclose(e.aem_mean, g["err_mean"], rtol=1e-12, atol=0)
    assert np.allclose(e.aem_m2 / (e.aem_n[:, None] - 1.0), g["err_mvar"], rtol=1e-12, atol=0)
    assert np.allclose(e.aem_prec, g["aem_prec"], rtol=1e-12, atol=0)
    assert (e.aem_flags[:, 0] == 1).all() and (e.aem_flags[:, 1] == 3).all()
    # resumable
    e2 = _aem_from_fixture(g, nthreads=3)
    a, b = e2.advance(123), e2.advance(n - 124)
    assert np.array_equal(np.concatenate([a["traj"], b["traj"]]), out["traj"])
    assert np.array_equal(e2.aem_mean, e.aem_mean) and np.array_equal(e2.cache_val, e.cache_val)


def test_f17_dense_proposal_covariance_d16(golden):
    """DenseCovarianceMatrix proposal beyond the lane kernels' dimension: rows of L xi as k-ordered fma chains
    (the matrix-core order) against BLAS' dgemv in the reference: decisions equal, states to rounding."""
    g = golden("f17_mrw_dense_proposal_d16.npz")
    n = int(g["n_states"])
    p = o.posterior_linear(g["A"], g["b"], g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
    e = o.Ensemble(p, np.ones(16), g["x0"], int(g["seed"]), chol=g["chol"])
    out = e.advance(n - 1)
    assert np.array_equal(out["acc"], g["acc"])
    assert np.abs(out["traj"] - g["traj"][1:]).max() <= 1e-12
    want = g["logpost"][1:]
    assert np.abs(out["logpost"] - want).max() <= 1e-12 * np.abs(want).max()


@pytest.mark.parametrize("name", ["f16_aem_linear_d128.npz", "f16b_aem_linear_d48_rows12.npz"])
def test_f16_adaptive_error_model_d128(golden, name):
    """aem.py over two d = 128 linear models (the matrix-core kernel's size) and over two d = 48 models with twelve
    data rows: chains (sha256 of all states), decisions at both levels and the learned error model == the live
    reference's."""
    g = golden(name)
    n = int(g["n_states"])
    e = _aem_from_fixture(g)
    out = e.advance(n - 1, want_sub=True)
    assert np.array_equal(out["acc"], g["acc"]) and np.array_equal(out["sub_acc"], g["sub_acc"])
    full = np.concatenate([g["x0"][None], out["traj"]])
    assert hashlib.sha256(np.ascontiguousarray(full).tobytes()).hexdigest() == str(g["traj_sha256"])
    want = g["logpost"][1:]
    assert np.abs(out["logpost"] - want).max() <= LP_RTOL * np.abs(want).max()
    assert np.array_equal(e.aem_n, g["err_n"]) and e.aem_n.min() > int(g["min_data_size"])
    assert np.allclose(e.aem_mean, g["err_mean"], rtol=1e-11, atol=1e-14)
    assert np.allclose(e.aem_m2 / (e.aem_n[:, None] - 1.0), g["err_mvar"], rtol=1e-10, atol=0)
    assert np.allclose(e.aem_prec, g["aem_prec"], rtol=1e-11, atol=0)


def test_f13_dense_proposal_covariance(golden):
    """DenseCovarianceMatrix proposal (covariance.py:69-94): the reference sums L @ xi in BLAS' order,
    the ensemble row-wise left to right -- every decision equals the reference's, states agree to
    rounding (1e-12 allowed, 5e-15 observed)."""
    g = golden("f13_mrw_2d_dense_proposal.npz")
    n = int(g["n_states"])
    e = o.Ensemble(o.gauss_whitened(g["mean"], g["whiten"], float(g["log_norm"])), np.ones(2), g["x0"],
                   int(g["seed"]), chol=g["chol"])
    out = e.advance(n - 1)
    assert np.array_equal(out["acc"], g["acc"])
    assert np.abs(out["traj"] - g["traj"][1:]).max() <= 1e-12
    assert np.abs(out["logpost"] - g["logpost"][1:]).max() <= 1e-12 * np.abs(g["logpost"]).max()
    # a diagonal L is the diagonal proposal, bit for bit
    a = o.Ensemble(e.target, np.array([0.7, 1.3]), g["x0"], 3).advance(50)
    b = o.Ensemble(e.target, np.ones(2), g["x0"], 3, chol=np.diag([0.7, 1.3])).advance(50)
    assert np.array_equal(a["traj"], b["traj"])


def test_perfect_surrogate_accepts_everything(golden):
    """test_mlda.py:94-130: coarse == fine => every fine proposal that moved is accepted."""
    g = golden("f5_mlda2_2d.npz")
    fine = o.gauss_whitened(g["fine_mean"], g["fine_whiten"], float(g["fine_log_norm"]))
    e = o.Ensemble(fine, g["sigma"], g["x0"][:16], 99, surrogate=fine, sub_chain_length=4)
    out = e.advance(400, want_sub=True)
    moved = out["sub_acc"].any(axis=1)
    assert np.array_equal(out["acc"].astype(bool), moved)


def test_threads_do_not_change_results(golden):
    g = golden("f2_mrw_2d.npz")
    p = o.gauss_whitened(g["mean"], g["whiten"], float(g["log_norm"]))
    a = o.Ensemble(p, g["sigma"], g["x0"], 5, nthreads=1).advance(200)
    b = o.Ensemble(p, g["sigma"], g["x0"], 5, nthreads=4).advance(200)
    assert np.array_equal(a["traj"], b["traj"]) and np.array_equal(a["acc"], b["acc"])


def test_sharding_by_global_chain_id(golden):
    g = golden("f2_mrw_2d.npz")
    p = o.gauss_whitened(g["mean"], g["whiten"], float(g["log_norm"]))
    whole = o.Ensemble(p, g["sigma"], g["x0"], 5).advance(100)
    lo = o.Ensemble(p, g["sigma"], g["x0"][:40], 5, chain0=0).advance(100)
    hi = o.Ensemble(p, g["sigma"], g["x0"][40:], 5, chain0=40).advance(100)
    assert np.array_equal(np.concatenate([lo["traj"], hi["traj"]], axis=1), whole["traj"])


def test_f7_two_level_over_a_model_hierarchy(golden):
    """example_inference_linearModel_twoLevel.py: vanilla MLDA over [surrogate, target] linear models"""
    g = golden("f7_mlda2_linear_models.npz")
    n, S = int(g["n_states"]), int(g["S"])
    common = (g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
    fine = o.posterior_linear(g["A_fine"], g["b_fine"], *common)
    coarse = o.posterior_linear(g["A_coarse"], g["b_coarse"], *common)
    e = o.Ensemble(fine, g["sigma"], g["x0"], int(g["seed"]), surrogate=coarse, sub_chain_length=S)
    out = e.advance(n - 1, want_sub=True)
    _check(out, g, n)
    assert np.array_equal(out["sub_acc"], g["sub_acc"])


@pytest.mark.parametrize("name", ["f14_mlda2_linear_d128.npz", "f14b_mlda2_linear_d192.npz", "f14c_mlda2_linear_d300.npz"])
def test_f14_two_level_over_a_model_hierarchy_d128(golden, name):
    """mlda.py:315-335 with two d = 128 / d = 192 linear models (the matrix-core kernels' sizes): the reference's
    decisions at both levels, its chains (sha256 of all states) and log-posteriors"""
    g = golden(name)
    n, S = int(g["n_states"]), int(g["S"])
    common = (g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
    fine = o.posterior_linear(g["A_fine"], g["b_fine"], *common)
    coarse = o.posterior_linear(g["A_coarse"], g["b_coarse"], *common)
    e = o.Ensemble(fine, g["sigma"], g["x0"], int(g["seed"]), surrogate=coarse, sub_chain_length=S)
    out = e.advance(n - 1, want_sub=True)
    assert np.array_equal(out["acc"], g["acc"]) and np.array_equal(out["sub_acc"], g["sub_acc"])
    full = np.concatenate([g["x0"][None], out["traj"]])
    assert np.array_equal(full[:, :2, :], g["traj_head"])
    assert hashlib.sha256(np.ascontiguousarray(full).tobytes()).hexdigest() == str(g["traj_sha256"])
    want = g["logpost"][1:]
    assert np.abs(out["logpost"] - want).max() <= LP_RTOL * np.abs(want).max()
    assert 0.4 < g["acc"].mean() < 0.95 and 0.2 < g["sub_acc"].mean() < 0.8   # both levels decide both ways


@pytest.mark.parametrize("name", ["f18_mlda_nested2_linear_d128.npz", "f18b_mlda_nested4_linear_d24.npz",
                                  "f18c_mlda_nested8_linear_d24.npz", "f18d_mlda_nested2_linear_d192.npz",
                                  "f18e_mlda_nested2_linear_d300.npz"])
def test_f18_nested_levels_over_model_hierarchies_at_large_dimension(golden, name):
    """MLDABuilder.bayesModel = a hierarchy of three (d = 128) / five (d = 24) linear models: an even number of
    surrogates, run by the reference as a nest of effective levels (mlda.py:12-73,100-116,315-335).  The
    reference's top-level decisions, the accepted counts of the lower levels, its chains (sha256) and
    log-posteriors."""
    g = golden(name)
    n = int(g["n_states"])
    common = (g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
    models = [o.posterior_linear(A, b, *common) for A, b in zip(g["A"], g["b"])]
    P, Q, steps = o.nested_levels(models[:-1], [int(v) for v in g["sub_lengths"]], models[-1])
    e = o.NestedEnsemble(P, Q, steps, g["sigma"], g["x0"], int(g["seed"]))
    out = e.advance(n - 1)
    assert e.cost_top == int(g["cost_top"])
    assert np.array_equal(out["acc"], g["acc"]) and np.array_equal(e.level_acc, g["level_acc"])
    full = np.concatenate([g["x0"][None], out["traj"]])
    assert np.array_equal(full[:, :2, :], g["traj_head"])
    assert hashlib.sha256(np.ascontiguousarray(full).tobytes()).hexdigest() == str(g["traj_sha256"])
    want = g["logpost"][1:]
    assert np.abs(out["logpost"] - want).max() <= LP_RTOL * np.abs(want).max()
    assert 0.3 < g["acc"].mean() < 0.9                                        # the levels decide both ways


def _f15_problems(g):
    lin = o.dense_covariances(
        o.posterior_linear(g["A"], g["b"], g["data"], np.ones(4), g["prior_mean"], np.ones(3)),
        noise_cov=g["noise_cov"], prior_cov=g["prior_cov"])
    lv = o.dense_covariances(
        o.posterior_lv(g["lv_design"], float(g["lv_t_final"]), float(g["lv_alpha"]), float(g["lv_gamma"]),
                       int(g["lv_rk4_steps"]), g["lv_data"], np.ones(2), g["lv_prior_mean"], g["lv_prior_precision"]),
        noise_cov=g["lv_noise_cov"])
    return lin, lv


def test_f15_dense_noise_and_prior_covariances(golden):
    """CentredGaussianNoise / Gaussian over DenseCovarianceMatrix (covariance.py:69-94): the reference solves
    two triangular systems per norm with LAPACK (unspecified order), the ensemble forms |L^-1 r|^2 in its own
    order: every decision and every state equal, log-posteriors to rounding (measured 3e-15, bar 1e-12)."""
    g = golden("f15_dense_noise_prior.npz")
    lin, lv = _f15_problems(g)
    n = int(g["n_states"])
    e = o.Ensemble(lin, g["sigma"], g["x0"], int(g["seed"]))
    out = e.advance(n - 1)
    _check(out, g, n)
    nl = int(g["lv_n_states"])
    e = o.Ensemble(lv, g["lv_sigma"], g["lv_x0"], int(g["lv_seed"]))
    out = e.advance(nl - 1)
    _check(out, dict(acc=g["lv_acc"], traj=g["lv_traj"], logpost=g["lv_logpost"]), nl)
    # against NumPy on random residuals: |r|^2_Sigma = r . Sigma^-1 r
    rng = np.random.default_rng(0)
    x = rng.standard_normal((50, 3))
    want = np.array([-0.5 * (xi - g["prior_mean"]) @ np.linalg.solve(g["prior_cov"], xi - g["prior_mean"]) for xi in x])
    only_prior = dict(lin, data=np.zeros((1, 4)), n_data=1, fwd_matrix=np.zeros((4, 3)), fwd_shift=np.zeros(4))
    got, _ = o.logdensity(only_prior, x)
    assert np.allclose(got, want, rtol=1e-12, atol=0)


def _f19_problems(g):
    base = lambda A, b: o.posterior_linear(A, b, g["data"], np.ones(g["data"].shape[1]), g["prior_mean"], g["prior_precision"])
    fine = o.dense_covariances(base(g["A"], g["b"]), noise_cov=g["noise_cov"])
    coarse = o.dense_covariances(base(g["A_coarse"], g["b_coarse"]), noise_cov=g["noise_cov"])
    return fine, coarse


def test_f19_dense_noise_covariance_beyond_the_lane_sizes(golden):
    """A dense 24 x 24 noise covariance over a d = 24 linear model (noise.py:21-22, covariance.py:69-94): the
    problem is whitened once (ymc_lower_times) and runs as a diagonal one.  The live reference's decisions and
    chains of a random walk and of two-level delayed acceptance over two such models; log-posteriors to
    rounding; `forward` still returns A x + b."""
    g = golden("f19_dense_noise_d24.npz")
    fine, coarse = _f19_problems(g)
    assert "fwd_matrix_raw" in fine and np.array_equal(fine["noise_precision"], np.ones(24))
    n = int(g["n_states"])
    e = o.Ensemble(fine, g["sigma"], g["x0"], int(g["seed"]))
    out = e.advance(n - 1)
    _check(out, g, n)
    n2, S = int(g["ml_n_states"]), int(g["ml_S"])
    e = o.Ensemble(fine, g["sigma"], g["ml_x0"], int(g["ml_seed"]), surrogate=coarse, sub_chain_length=S)
    out = e.advance(n2 - 1, want_sub=True)
    _check(out, dict(acc=g["ml_acc"], traj=g["ml_traj"], logpost=g["ml_logpost"]), n2)
    assert np.array_equal(out["sub_acc"], g["ml_sub_acc"])
    assert 0.3 < g["acc"].mean() < 0.8 and 0.3 < g["ml_sub_acc"].mean() < 0.8
    x = np.random.default_rng(0).standard_normal((20, 24))
    assert np.allclose(o.forward(fine, x), x @ g["A"].T + g["b"], rtol=1e-13, atol=1e-13)
    # the whitened log-likelihood against NumPy: -0.5 sum_i (F - D_i) . Sigma^-1 (F - D_i)
    like = dict(fine, likelihood_only=1)
    got, _ = o.logdensity(like, x)
    F = x @ g["A"].T + g["b"]
    want = np.array([-0.5 * sum((f - D) @ np.linalg.solve(g["noise_cov"], f - D) for D in g["data"]) for f in F])
    assert np.allclose(got, want, rtol=1e-11, atol=0)


@pytest.mark.parametrize("name", ["f20_dense_prior_noise_d20.npz", "f20b_dense_prior_noise_d128.npz"])
def test_f20_dense_prior_and_noise_beyond_the_lane_sizes(golden, name):
    """Gaussian prior AND noise over DenseCovarianceMatrix at d = 20 and d = 128: the noise is whitened away, the prior
    becomes d more outputs of the forward map (oracle.dense_covariances, as the product's lowering).  The live
    reference's decisions and chains; log-posteriors to rounding; against NumPy on random points."""
    g = golden(name)
    d = g["A"].shape[1]
    p = o.dense_covariances(o.posterior_linear(g["A"], g["b"], g["data"], np.ones(d), g["prior_mean"], np.ones(d)),
                            noise_cov=g["noise_cov"], prior_cov=g["prior_cov"])
    assert p["data_dim"] == 2 * d and not p["prior_precision"].any() and "lin_tail" in p
    n = int(g["n_states"])
    e = o.Ensemble(p, g["sigma"], g["x0"], int(g["seed"]))
    _check(e.advance(n - 1), g, n)
    x = np.random.default_rng(1).standard_normal((25, d))
    F = x @ g["A"].T + g["b"]
    assert np.allclose(o.forward(p, x), F, rtol=1e-13, atol=1e-13)
    want = np.array([-0.5 * sum((f - D) @ np.linalg.solve(g["noise_cov"], f - D) for D in g["data"])
                     - 0.5 * (xi - g["prior_mean"]) @ np.linalg.solve(g["prior_cov"], xi - g["prior_mean"]) for f, xi in zip(F, x)])
    assert np.allclose(o.logdensity(p, x)[0], want, rtol=1e-12, atol=0)


def test_f8_preconditioned_crank_nicolson(golden):
    """pcn.py:9-57 on the Lotka-Volterra likelihood with a centred Gaussian prior"""
    g = golden("f8_pcn_lotka_volterra.npz")
    n = int(g["n_states"])
    p = o.posterior_lv(g["design"], float(g["t_final"]), float(g["alpha"]), float(g["gamma"]), int(g["rk4_steps"]),
                       g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
    p["likelihood_only"] = 1
    e = o.Ensemble(p, g["prior_sigma"], g["x0"], int(g["seed"]), pcn_step_size=float(g["step_size"]))
    out = e.advance(n - 1)
    assert np.array_equal(out["acc"], g["acc"]) and np.array_equal(out["traj"], g["traj"][1:])
    assert np.abs(out["logpost"] - g["loglik"][1:]).max() <= LP_RTOL * np.abs(g["loglik"]).max()


def test_f21_preconditioned_crank_nicolson_linear_d64(golden):
    """pcn.py:9-57 on a d = 64 linear model with a centred prior of non-uniform variances (the matrix-core
    kernel's pCN proposal): the live reference's decisions and chains, log-likelihoods to rounding"""
    g = golden("f21_pcn_linear_d64.npz")
    n = int(g["n_states"])
    p = o.posterior_linear(g["A"], g["b"], g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
    p["likelihood_only"] = 1
    e = o.Ensemble(p, g["prior_sigma"], g["x0"], int(g["seed"]), pcn_step_size=float(g["step_size"]))
    out = e.advance(n - 1)
    assert np.array_equal(out["acc"], g["acc"]) and np.array_equal(out["traj"], g["traj"][1:])
    assert np.abs(out["logpost"] - g["loglik"][1:]).max() <= LP_RTOL * np.abs(g["loglik"]).max()
    assert 0.1 < g["acc"].mean() < 0.6
