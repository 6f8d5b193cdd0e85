"""Parity of the HIP path (through the C ABI) with the CPU oracle and with the
golden vectors of the live reference.  Bar: accept masks, trajectories,
diagnostics AND log-densities bit-exact against the oracle (same numeric
spec); against the reference's own outputs masks/trajectories bit-exact and
log-densities within 1e-12 relative (north_star asks 1e-6)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as o  # noqa: E402


@pytest.fixture(scope="module")
def ffi():
    from yagremcmc_amd import _ffi
    assert _ffi.device_count() >= 1
    return _ffi


@pytest.fixture(scope="module")
def ctx(ffi):
    return ffi.default_context(0)


ALL = None


def _flags(ffi, moments=True, sub=False):
    f = ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT | ffi.RECORD_LOGPOST
    if moments:
        f |= ffi.RECORD_MOMENTS
    if sub:
        f |= ffi.RECORD_SUB_ACCEPT
    return f


def _gauss_low(ffi, mean, whiten, log_norm):
    from yagremcmc_amd import lowering
    low = lowering.Lowered(ffi.GAUSS_WHITENED, int(np.size(mean)))
    low.set_array("mean", np.asarray(mean).reshape(-1))
    low.set_array("whiten", whiten)
    low.desc.log_norm = float(log_norm)
    return low


def _posterior_low(ffi, kind, dim, g, **extra):
    from yagremcmc_amd import lowering
    low = lowering.Lowered(kind, dim)
    data = np.asarray(g["data"])
    low.desc.n_data, low.desc.data_dim = data.shape
    low.set_array("data", data)
    low.set_array("noise_precision", g["noise_precision"])
    low.set_array("prior_mean", g["prior_mean"])
    low.set_array("prior_precision", g["prior_precision"])
    for k, v in extra.items():
        if isinstance(v, np.ndarray):
            low.set_array(k, v)
        else:
            setattr(low.desc, k, v)
    return low


def _device_run(ffi, ctx, low, sigma, x0, seed, nsteps, method=None, sur=None, S=0, eq=None, chain0=0,
                moments=True, splits=1):
    from yagremcmc_amd import lowering
    method = method or ffi.METHOD_MRW
    prob = lowering.make_problem(ctx, method, low, sigma, eq if eq is not None else ffi.EQ_EXACT,
                                 surrogate=sur, sub_chain_length=S)
    K = x0.shape[0]
    run = ffi.Run(prob, seed, K, x0, nsteps, _flags(ffi, moments, sub=(method == ffi.METHOD_MLDA2)), chain0)
    base, rem = divmod(nsteps, splits)
    for i in range(splits):
        run.advance(base + (1 if i < rem else 0))
    run.sync()
    out = dict(traj=run.trajectory(), acc=run.accept(), logpost=run.logpost(), summary=run.summary(),
               state=run.state())
    if method == ffi.METHOD_MLDA2:
        out["sub_acc"] = run.sub_accept()
    return out


def _same_as_oracle(dev, e, out):
    assert np.array_equal(dev["acc"], out["acc"])
    assert np.array_equal(dev["traj"], out["traj"])
    assert np.array_equal(dev["logpost"], out["logpost"])
    assert np.array_equal(dev["summary"]["accepted"], e.acc_count)
    assert np.array_equal(dev["summary"]["wf_mean"], e.wf_mean)
    assert np.array_equal(dev["summary"]["wf_m2"], e.wf_m2)
    assert np.array_equal(dev["summary"]["status"], e.status)
    assert np.array_equal(dev["state"][0], e.x) and np.array_equal(dev["state"][1], e.lp)


def _same_as_reference(dev, g, lp_rtol=1e-12):
    assert np.array_equal(dev["acc"], g["acc"])
    if "traj" in g:
        assert np.array_equal(dev["traj"], g["traj"][1:])
    want = g["logpost"][1:]
    assert np.abs(dev["logpost"] - want).max() <= lp_rtol * np.abs(want).max()


# ---------------------------------------------------------------------------
# the numeric spec on the device == on the CPU, bit for bit
# ---------------------------------------------------------------------------

def test_elementary_functions_bit_exact(ctx):
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.integers(1, 2**53, size=400000) * 2.0**-53, np.exp(rng.uniform(-700, 700, 100000))])
    assert np.array_equal(ctx.numerics(0, x)[0], o.log(x))
    x = np.concatenate([rng.uniform(-750, 712, 400000), rng.uniform(-40, 2, 200000),
                        [0.0, -745.3, 709.79, np.inf, -np.inf, -1e-300]])
    assert np.array_equal(ctx.numerics(1, x)[0], o.exp(x))
    assert np.isnan(ctx.numerics(1, np.array([np.nan]))[0][0])
    t = rng.integers(0, 2**53, size=400000) * 2.0**-52
    s, c = ctx.numerics(2, t)
    so, co = o.sincospi(t)
    assert np.array_equal(s, so) and np.array_equal(c, co)


def test_sqrt_and_division_correctly_rounded_on_device(ctx):
    rng = np.random.default_rng(6)
    x = np.concatenate([rng.uniform(0, 80, 300000), np.exp(rng.uniform(-700, 700, 100000)), [0.0, 5e-324, 2.0]])
    assert np.array_equal(ctx.numerics(3, x)[0], np.sqrt(x))
    # the Box-Muller radius (bare Goldschmidt on v_rsq_f64): the host's sqrt over its whole domain and beyond
    w = np.concatenate([-2.0 * np.log((rng.integers(0, 2**53, 600000) + 1) * 2.0**-53), rng.uniform(0, 80, 300000),
                        np.exp(rng.uniform(-69, 60, 100000)), [0.0, 2.0**-52, 2.0**-100, 2.0**-101, 73.47, 1e300]])
    assert np.array_equal(ctx.numerics(6, w)[0], np.sqrt(np.maximum(w, 2.0**-100)))
    a = rng.normal(size=400000) * np.exp(rng.uniform(-30, 30, 400000))
    b = rng.integers(1, 10**7, size=400000).astype(np.float64)
    assert np.array_equal(ctx.numerics(4, a, divisor=b)[0], a / b)
    b2 = rng.normal(size=400000) * np.exp(rng.uniform(-300, 300, 400000))
    assert np.array_equal(ctx.numerics(4, a, divisor=b2)[0], a / b2)


def test_draw_stream_bit_exact(ctx):
    for d, K, nt, t0, chain0 in [(2, 200, 300, 0, 0), (1, 64, 100, 7, 1000), (5, 33, 50, 2**33 + 1, 2**31),
                                 (128, 8, 20, 3, 17)]:
        xi, u = ctx.rng_fill(20250103, chain0, K, t0, nt, d)
        xo, uo = o.rng_fill(20250103, chain0, K, t0, nt, d)
        assert np.array_equal(xi, xo) and np.array_equal(u, uo)


# ---------------------------------------------------------------------------
# golden fixtures
# ---------------------------------------------------------------------------

def test_f1_scalar_1d(ffi, ctx, golden):
    from yagremcmc_amd import lowering
    g = golden("f1_mrw_1d.npz")
    n = int(g["n_states"])
    low = lowering.Lowered(ffi.GAUSS_ISO1D, 1)
    low.set_array("mean", np.array([float(g["mean"])]))
    low.desc.variance = float(g["variance"])
    dev = _device_run(ffi, ctx, low, g["sigma"], g["x0"], int(g["seed"]), n - 1, eq=ffi.EQ_ISCLOSE)
    e = o.Ensemble(o.gauss_iso1d(float(g["mean"]), float(g["variance"])), g["sigma"], g["x0"], int(g["seed"]),
                   eq_mode=o.EQ_ISCLOSE)
    _same_as_oracle(dev, e, e.advance(n - 1))
    _same_as_reference(dev, g)
    assert np.array_equal(dev["summary"]["wf_mean"][0], g["wf_mean"].reshape(-1))
    assert np.array_equal(dev["summary"]["wf_m2"][0] / (n - 2.0), g["wf_var"].reshape(-1))


@pytest.mark.parametrize("name,splits", [("f2_mrw_2d.npz", 1), ("f2_mrw_2d.npz", 7), ("f2b_mrw_2d_indep.npz", 2)])
def test_f2_mrw_2d(ffi, ctx, golden, name, splits):
    g = golden(name)
    n = int(g["n_states"])
    low = _gauss_low(ffi, g["mean"], g["whiten"], g["log_norm"])
    dev = _device_run(ffi, ctx, low, g["sigma"], g["x0"], int(g["seed"]), n - 1, splits=splits)
    e = o.Ensemble(o.gauss_whitened(g["mean"], g["whiten"], float(g["log_norm"])), g["sigma"], g["x0"],
                   int(g["seed"]))
    _same_as_oracle(dev, e, e.advance(n - 1))
    _same_as_reference(dev, g)
    assert np.array_equal(dev["summary"]["accepted"] / (n - 1), g["acc_rate"])
    assert np.array_equal(dev["summary"]["wf_mean"], g["wf_mean"])
    assert np.array_equal(dev["summary"]["wf_m2"] / (n - 2.0), g["wf_var"])


def test_f3b_linear_2x2(ffi, ctx, golden):
    g = golden("f3b_linear_2x2.npz")
    n = int(g["n_states"])
    low = _posterior_low(ffi, ffi.POSTERIOR_LINEAR, 2, g, fwd_matrix=np.asarray(g["A"]), fwd_shift=np.asarray(g["b"]))
    dev = _device_run(ffi, ctx, low, g["sigma"], g["x0"], int(g["seed"]), n - 1)
    p = o.posterior_linear(g["A"], g["b"], g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
    e = o.Ensemble(p, g["sigma"], g["x0"], int(g["seed"]))
    _same_as_oracle(dev, e, e.advance(n - 1))
    _same_as_reference(dev, g)
    from yagremcmc_amd import lowering
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, g["sigma"])
    lp, st = prob.logdensity(g["grid"])
    assert np.array_equal(lp, o.logdensity(p, g["grid"])[0]) and (st == 0).all()
    assert np.abs(lp - g["grid_logpost"]).max() <= 1e-12 * np.abs(g["grid_logpost"]).max()
    assert np.array_equal(prob.forward(g["grid"]), o.forward(p, g["grid"]))


def test_f4_lotka_volterra(ffi, ctx, golden):
    g = golden("f4_lotka_volterra.npz")
    n = int(g["n_states"])
    low = _posterior_low(ffi, ffi.POSTERIOR_LV, 2, g, design=np.asarray(g["design"]), t_final=float(g["t_final"]),
                         alpha=float(g["alpha"]), gamma=float(g["gamma"]), rk4_steps=int(g["rk4_steps"]))
    dev = _device_run(ffi, ctx, low, g["sigma"], g["x0"], int(g["seed"]), n - 1)
    p = o.posterior_lv(g["design"], float(g["t_final"]), float(g["alpha"]), float(g["gamma"]), int(g["rk4_steps"]),
                       g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
    e = o.Ensemble(p, g["sigma"], g["x0"], int(g["seed"]))
    _same_as_oracle(dev, e, e.advance(n - 1))
    _same_as_reference(dev, g)
    from yagremcmc_amd import lowering
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, g["sigma"])
    f = prob.forward(g["thetas"])
    assert np.array_equal(f, o.forward(p, g["thetas"]))
    assert np.allclose(f.reshape(-1, 5, 2), g["forward_lsoda_rtol1e4"], rtol=1e-3, atol=5e-3)
    # stiff start of the example script (example_inference_lotkaVolterra_singleLevel.py:106):
    # same value and same failure flag as the oracle
    lp, st = prob.logdensity(np.array([[-7.0, 2.8], [-0.6, -0.3]]))
    lpo, sto = o.logdensity(p, np.array([[-7.0, 2.8], [-0.6, -0.3]]))
    assert np.array_equal(lp, lpo) and np.array_equal(st, sto)


def _f24_pair(ffi, g, rtol=None, steps=50):
    rtol = float(g["rtol"]) if rtol is None else rtol
    low = _posterior_low(ffi, ffi.POSTERIOR_LV, 2, g, design=np.asarray(g["design"]), t_final=float(g["t_final"]),
                         alpha=float(g["alpha"]), gamma=float(g["gamma"]), rk4_steps=steps, lv_rtol=rtol,
                         lv_atol=float(g["atol"]) if rtol > 0 else 0.0)
    p = o.posterior_lv(g["design"], float(g["t_final"]), float(g["alpha"]), float(g["gamma"]), steps, g["data"],
                       g["noise_precision"], g["prior_mean"], g["prior_precision"], rtol=rtol, atol=float(g["atol"]))
    return low, p


@pytest.mark.parametrize("rows", ["0", "1"])
def test_f24_adaptive_lotka_volterra_from_the_examples_own_start(ffi, ctx, golden, monkeypatch, rows):
    """The reference's Lotka-Volterra example from its own starting point (-7, 2.8), integrated adaptively
    (ymc_lv_dp45_solve), one lane per chain and one lane per design row: chains == the reference's loop with that
    integrator plugged in == oracle, bit for bit; the forward map == oracle's at every probe, including parameters
    where the solve gives up."""
    from yagremcmc_amd import lowering
    monkeypatch.setenv("YMC_LV_ROWS", rows)
    g = golden("f24_lotka_volterra_adaptive.npz")
    n = int(g["n_states"])
    low, p = _f24_pair(ffi, g)
    dev = _device_run(ffi, ctx, low, g["sigma"], g["x0"], int(g["seed"]), n - 1, splits=3)
    e = o.Ensemble(p, g["sigma"], g["x0"], int(g["seed"]))
    _same_as_oracle(dev, e, e.advance(n - 1))
    _same_as_reference(dev, g)
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, g["sigma"])
    probes = np.concatenate([g["thetas"], [[40.0, 40.0], [-30.0, 5.0], [5.0, -30.0]]])
    assert np.array_equal(prob.forward(probes), o.forward(p, probes), equal_nan=True)
    lp, st = prob.logdensity(probes)
    lpo, sto = o.logdensity(p, probes)
    assert np.array_equal(lp, lpo) and np.array_equal(st, sto) and st[-3] != 0 and (st[:len(g["thetas"])] == 0).all()
    for rtol in (1e-2, 1e-7, 1e-11):
        low2, p2 = _f24_pair(ffi, g, rtol)
        prob2 = lowering.make_problem(ctx, ffi.METHOD_MRW, low2, g["sigma"])
        assert np.array_equal(prob2.forward(g["thetas"]), o.forward(p2, g["thetas"]))


@pytest.mark.parametrize("method", ["mlda2", "nested", "pcn"])
def test_adaptive_lotka_volterra_levels_vs_oracle(ffi, ctx, golden, method):
    """Levels that differ in the tolerance of the adaptive integrator (the reference's two-level example separates its
    levels by scipy's rtol alone), and one adaptive level under a fixed-step one; pCN on the adaptive likelihood."""
    from yagremcmc_amd import lowering
    g = golden("f24_lotka_volterra_adaptive.npz")
    K = 80
    rng = np.random.default_rng(24)
    x0 = np.log([0.4, 0.6]) + 0.2 * rng.normal(size=(K, 2))
    sigma = np.array([0.08, 0.06])
    if method == "mlda2":
        (lf, pf), (lc, pc) = _f24_pair(ffi, g, 1e-6), _f24_pair(ffi, g, 1e-2)
        prob = lowering.make_problem(ctx, ffi.METHOD_MLDA2, lf, sigma, surrogate=lc, sub_chain_length=3)
        run = ffi.Run(prob, 5, K, x0, 30, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT | ffi.RECORD_LOGPOST, chain_offset=3)
        run.advance(13)
        run.advance(17)
        e = o.Ensemble(pf, sigma, x0, 5, chain0=3, surrogate=pc, sub_chain_length=3)
        out = e.advance(30)
        assert np.array_equal(run.trajectory(), out["traj"]) and np.array_equal(run.accept(), out["acc"])
        assert np.array_equal(run.logpost(), out["logpost"]) and 0.05 < out["acc"].mean() < 0.95
    elif method == "nested":
        lows = [_f24_pair(ffi, g, 1e-2), _f24_pair(ffi, g, 0.0, steps=12), _f24_pair(ffi, g, 1e-4), _f24_pair(ffi, g, 1e-3),
                _f24_pair(ffi, g, 1e-7)]
        subs = [2, 3, 2, 2]
        dev = _nested_device_run(ffi, ctx, lows[-1][0], [l for l, _ in lows[:-1]], subs, sigma, x0, 99, 12, splits=2, chain0=5)
        P, Q, steps = o.nested_levels([q for _, q in lows[:-1]], subs, lows[-1][1])
        e = o.NestedEnsemble(P, Q, steps, sigma, x0, 99, chain0=5)
        out = e.advance(12)
        _same_as_oracle(dev, e, out)
        assert np.array_equal(dev["level_acc"], e.level_acc)
    else:
        low, p = _f24_pair(ffi, g, 1e-5)
        low.desc.likelihood_only = 1
        p["likelihood_only"] = 1
        psig = 1.0 / np.sqrt(np.asarray(g["prior_precision"]))
        x0 = x0 * 0.3
        prob = lowering.make_problem(ctx, ffi.METHOD_PCN, low, psig, pcn_step_size=0.02)
        run = ffi.Run(prob, 8, K, x0, 40, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT, chain_offset=0)
        run.advance(40)
        e = o.Ensemble(p, psig, x0, 8, pcn_step_size=0.02)
        out = e.advance(40)
        assert np.array_equal(run.trajectory(), out["traj"]) and np.array_equal(run.accept(), out["acc"])


def test_f5_two_level(ffi, ctx, golden):
    g = golden("f5_mlda2_2d.npz")
    n, S = int(g["n_states"]), int(g["S"])
    fine = _gauss_low(ffi, g["fine_mean"], g["fine_whiten"], g["fine_log_norm"])
    coarse = _gauss_low(ffi, g["coarse_mean"], g["coarse_whiten"], g["coarse_log_norm"])
    dev = _device_run(ffi, ctx, fine, g["sigma"], g["x0"], int(g["seed"]), n - 1, method=ffi.METHOD_MLDA2,
                      sur=coarse, S=S, splits=3)
    of = o.gauss_whitened(g["fine_mean"], g["fine_whiten"], float(g["fine_log_norm"]))
    oc = o.gauss_whitened(g["coarse_mean"], g["coarse_whiten"], float(g["coarse_log_norm"]))
    e = o.Ensemble(of, g["sigma"], g["x0"], int(g["seed"]), surrogate=oc, sub_chain_length=S)
    out = e.advance(n - 1, want_sub=True)
    _same_as_oracle(dev, e, out)
    assert np.array_equal(dev["sub_acc"], out["sub_acc"])
    _same_as_reference(dev, g)
    assert np.array_equal(dev["sub_acc"], g["sub_acc"])


def test_perfect_surrogate_accepts_every_move(ffi, ctx, golden):
    """test_mlda.py:94-130"""
    g = golden("f5_mlda2_2d.npz")
    fine = _gauss_low(ffi, g["fine_mean"], g["fine_whiten"], g["fine_log_norm"])
    x0 = np.tile([-8.0, -7.0], (4096, 1))
    dev = _device_run(ffi, ctx, fine, g["sigma"], x0, 99, 300, method=ffi.METHOD_MLDA2, sur=fine, S=4)
    moved = dev["sub_acc"].any(axis=1)
    assert np.array_equal(dev["acc"].astype(bool), moved)


def _nested_device_run(ffi, ctx, g_tgt, g_surs, sub_lengths, sigma, x0, seed, nsteps, splits=1, chain0=0):
    """Through the C ABI: problem with YMC_METHOD_MLDA_NESTED built from lowered densities."""
    from yagremcmc_amd import lowering

    class _Pre:  # a density that is already lowered
        def __init__(self, low):
            self.low = low

    real = lowering.lower_density

    def passthrough(d, rk4_steps=None):
        return d.low if isinstance(d, _Pre) else real(d, rk4_steps)

    lowering.lower_density = passthrough
    try:
        prob = lowering.make_nested_problem(ctx, _Pre(g_tgt), [_Pre(s_) for s_ in g_surs], sub_lengths, sigma)
    finally:
        lowering.lower_density = real
    K = x0.shape[0]
    run = ffi.Run(prob, seed, K, x0, nsteps, _flags(ffi, True), chain0)
    base, rem = divmod(nsteps, splits)
    for i in range(splits):
        run.advance(base + (1 if i < rem else 0))
    run.sync()
    acc, made = run.level_accept()
    return dict(traj=run.trajectory(), acc=run.accept(), logpost=run.logpost(), summary=run.summary(),
                state=run.state(), level_acc=acc, level_made=made)


@pytest.mark.parametrize("name", ["f10_mlda_nested2.npz", "f11_mlda_nested4.npz", "f23_mlda_nested8.npz",
                                  "f23b_mlda_nested14.npz"])
def test_f10_f11_even_surrogate_hierarchies(ffi, ctx, golden, name):
    g = golden(name)
    n = int(g["n_states"])
    ns = g["sur_mean"].shape[0]
    subs = [int(v) for v in g["sub_lengths"]]
    tgt = _gauss_low(ffi, g["tgt_mean"], g["tgt_whiten"], g["tgt_log_norm"])
    surs = [_gauss_low(ffi, g["sur_mean"][i], g["sur_whiten"][i], g["sur_log_norm"][i]) for i in range(ns)]
    dev = _nested_device_run(ffi, ctx, tgt, surs, subs, g["sigma"], g["x0"], int(g["seed"]), n - 1, splits=3)
    ot = o.gauss_whitened(g["tgt_mean"], g["tgt_whiten"], float(g["tgt_log_norm"]))
    osur = [o.gauss_whitened(g["sur_mean"][i], g["sur_whiten"][i], float(g["sur_log_norm"][i])) for i in range(ns)]
    P, Q, steps = o.nested_levels(osur, subs, ot)
    e = o.NestedEnsemble(P, Q, steps, g["sigma"], g["x0"], int(g["seed"]))
    out = e.advance(n - 1)
    _same_as_oracle(dev, e, out)
    assert np.array_equal(dev["level_acc"], e.level_acc)
    _same_as_reference(dev, g)
    assert np.array_equal(dev["level_acc"], g["level_acc"])
    made = (n - 1)
    for e_ in range(len(steps) - 1, -1, -1):
        made *= steps[e_]
        assert dev["level_made"][e_] == made


@pytest.mark.parametrize("d,E", [(1, 2), (3, 3), (5, 4), (8, 4), (1, 5), (2, 7), (4, 6), (8, 8), (3, 5)])
def test_nested_other_dimensions_and_depths_vs_oracle(ffi, ctx, d, E):
    rng = np.random.default_rng(100 * d + E)
    ns = 2 * (E - 1)

    def rand_gauss(scale):
        A = rng.normal(size=(d, d))
        cov = scale * (A @ A.T / d + np.eye(d))
        return o.gauss_from_cov(rng.normal(size=d) * 0.2, cov)

    ot = rand_gauss(1.0)
    osur = [rand_gauss(1.5 + 0.5 * (ns - i)) for i in range(ns)]
    subs = [int(v) for v in rng.integers(1, 4 if E <= 5 else 3, size=ns)]
    low = lambda p: _gauss_low(ffi, p["mean"], p["whiten"], p["log_norm"])
    sigma = np.full(d, 0.9)
    n = 60 if E <= 4 else 14
    x0 = rng.normal(size=(300, d))
    dev = _nested_device_run(ffi, ctx, low(ot), [low(s_) for s_ in osur], subs, sigma, x0, 4242, n, splits=2,
                             chain0=1000)
    P, Q, steps = o.nested_levels(osur, subs, ot)
    e = o.NestedEnsemble(P, Q, steps, sigma, x0, 4242, chain0=1000)
    out = e.advance(n)
    _same_as_oracle(dev, e, out)
    assert np.array_equal(dev["level_acc"], e.level_acc)


def test_nested_full_size_perfect_surrogates_accept_every_move(ffi, ctx, golden):
    """test_mlda.py:94-130 at ensemble size: all densities equal => every top-level proposal that
    moved is accepted, and so is every decision of the lower levels."""
    g = golden("f10_mlda_nested2.npz")
    tgt = _gauss_low(ffi, g["tgt_mean"], g["tgt_whiten"], g["tgt_log_norm"])
    K, n = 65536, 50
    x0 = np.tile([-8.0, -7.0], (K, 1))
    dev = _nested_device_run(ffi, ctx, tgt, [tgt, tgt, tgt, tgt], [6, 4, 4, 3], g["sigma"], x0, 123, n)
    # level 1 decides on the end point of 4 random-walk steps on the same density: ratio == 1
    acc1, made1 = dev["level_acc"][1], int(dev["level_made"][1])
    moved_any = np.any(np.diff(np.concatenate([x0[None], dev["traj"]]), axis=0) != 0.0, axis=2)
    assert np.array_equal(dev["acc"].astype(bool), moved_any)
    assert made1 == n * 3 and np.all(acc1 <= made1) and acc1.mean() > 0.9 * made1


def _aem_device_run(ffi, ctx, g, fine_g, coarse_g, sigma, x0, seed, nsteps, S, min_data, heur, splits=1, chain0=0):
    from yagremcmc_amd import lowering
    d = x0.shape[1]
    fine = _posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, fine_g, fwd_matrix=np.asarray(fine_g["A"]),
                          fwd_shift=np.asarray(fine_g["b"]))
    coarse = _posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, coarse_g, fwd_matrix=np.asarray(coarse_g["A"]),
                            fwd_shift=np.asarray(coarse_g["b"]))
    prob = lowering.make_problem(ctx, ffi.METHOD_AEM2, fine, sigma, ffi.EQ_EXACT, surrogate=coarse,
                                 sub_chain_length=S, aem=(min_data, heur))
    run = ffi.Run(prob, seed, x0.shape[0], x0, nsteps, _flags(ffi, True, sub=True), chain0)
    base, rem = divmod(nsteps, splits)
    for i in range(splits):
        run.advance(base + (1 if i < rem else 0))
    run.sync()
    return dict(traj=run.trajectory(), acc=run.accept(), logpost=run.logpost(), summary=run.summary(),
                state=run.state(), sub_acc=run.sub_accept(), err=run.error_model())


def _same_error_model(dev, e):
    assert np.array_equal(dev["err"]["n"], e.aem_n)
    assert np.array_equal(dev["err"]["mean"], e.aem_mean) and np.array_equal(dev["err"]["m2"], e.aem_m2)
    assert np.array_equal(dev["err"]["precision"], e.aem_prec)
    assert np.array_equal(dev["err"]["has_precision"], e.aem_flags[:, 0] != 0)


@pytest.mark.parametrize("name", ["f12_aem_linear.npz", "f12b_aem_linear_noheuristic.npz"])
def test_f12_adaptive_error_model(ffi, ctx, golden, name):
    g = golden(name)
    n, S = int(g["n_states"]), int(g["S"])
    common = dict(data=g["data"], noise_precision=g["noise_precision"], prior_mean=g["prior_mean"],
                  prior_precision=g["prior_precision"])
    fg = dict(common, A=g["A_fine"], b=g["b_fine"])
    cg = dict(common, A=g["A_coarse"], b=g["b_coarse"])
    dev = _aem_device_run(ffi, ctx, g, fg, cg, g["sigma"], g["x0"], int(g["seed"]), n - 1, S,
                          int(g["min_data_size"]), int(g["use_noise_heuristic"]), splits=3)
    e = o.AemEnsemble(o.posterior_linear(g["A_fine"], g["b_fine"], **common),
                      o.posterior_linear(g["A_coarse"], g["b_coarse"], **common), g["sigma"], g["x0"],
                      int(g["seed"]), S, int(g["min_data_size"]), int(g["use_noise_heuristic"]))
    out = e.advance(n - 1, want_sub=True)
    _same_as_oracle(dev, e, out)
    assert np.array_equal(dev["sub_acc"], out["sub_acc"])
    _same_error_model(dev, e)
    _same_as_reference(dev, g)
    assert np.array_equal(dev["sub_acc"], g["sub_acc"])
    assert np.array_equal(dev["err"]["n"], g["err_n"])
    assert np.allclose(dev["err"]["mean"], g["err_mean"], rtol=1e-12, atol=0)
    assert np.allclose(dev["err"]["precision"], g["aem_prec"], rtol=1e-12, atol=0)


@pytest.mark.parametrize("d,m,nd", [(1, 1, 3), (3, 2, 4), (5, 8, 2), (8, 5, 6)])
def test_aem_other_shapes_vs_oracle(ffi, ctx, d, m, nd):
    rng = np.random.default_rng(7 * d + m)
    A = rng.normal(size=(m, d)) / np.sqrt(d)
    b = rng.normal(size=m) * 0.3
    As, bs = A + 0.2 * rng.normal(size=(m, d)), b + 0.2 * rng.normal(size=m)
    theta = rng.normal(size=d)
    data = np.array([A @ theta + b + 0.5 * rng.normal(size=m) for _ in range(nd)])
    common = dict(data=data, noise_precision=np.full(m, 4.0) * rng.uniform(0.8, 1.2, size=m),
                  prior_mean=np.zeros(d), prior_precision=np.full(d, 0.25))
    sigma = np.full(d, 0.35)
    x0 = rng.normal(size=(500, d)) * 0.5
    dev = _aem_device_run(ffi, ctx, None, dict(common, A=A, b=b), dict(common, A=As, b=bs), sigma, x0, 31337, 250, 3,
                          4, 1, splits=2, chain0=77)
    e = o.AemEnsemble(o.posterior_linear(A, b, **common), o.posterior_linear(As, bs, **common), sigma, x0, 31337, 3,
                      4, 1, chain0=77)
    out = e.advance(250, want_sub=True)
    _same_as_oracle(dev, e, out)
    assert np.array_equal(dev["sub_acc"], out["sub_acc"])
    _same_error_model(dev, e)
    assert dev["err"]["has_precision"].mean() > 0.5


def test_aem_with_a_dense_prior_at_lane_sizes_vs_oracle(ffi, ctx):
    """The adaptive error model takes diagonal noise only (noise.py:30-33) but any prior: a dense 3 x 3 prior
    covariance on both levels (lane kernel; at matrix-core sizes the combination is refused)."""
    from yagremcmc_amd import lowering
    rng = np.random.default_rng(515)
    d, m, nd = 3, 4, 5
    A = rng.normal(size=(m, d)) / np.sqrt(d)
    b = rng.normal(size=m) * 0.3
    As, bs = A + 0.2 * rng.normal(size=(m, d)), b + 0.2 * rng.normal(size=m)
    data = np.array([A @ rng.normal(size=d) + b + 0.5 * rng.normal(size=m) for _ in range(nd)])
    B = rng.normal(size=(d, d))
    pc = B @ B.T + 2.0 * np.eye(d)
    g = dict(data=data, noise_precision=np.full(m, 4.0), prior_mean=rng.normal(size=d) * 0.2, prior_precision=np.ones(d))
    fine = _dense_low(ffi, ffi.POSTERIOR_LINEAR, d, g, None, pc, fwd_matrix=A, fwd_shift=b)
    coarse = _dense_low(ffi, ffi.POSTERIOR_LINEAR, d, g, None, pc, fwd_matrix=As, fwd_shift=bs)
    sigma, x0 = np.full(d, 0.35), rng.normal(size=(200, d)) * 0.5
    prob = lowering.make_problem(ctx, ffi.METHOD_AEM2, fine, sigma, ffi.EQ_EXACT, surrogate=coarse, sub_chain_length=3,
                                 aem=(4, 1))
    run = ffi.Run(prob, 99, 200, x0, 150, _flags(ffi, True, sub=True), 11)
    run.advance(70)
    run.advance(80)
    dev = dict(traj=run.trajectory(), acc=run.accept(), logpost=run.logpost(), summary=run.summary(), state=run.state(),
               sub_acc=run.sub_accept(), err=run.error_model())
    of = o.dense_covariances(o.posterior_linear(A, b, **g), prior_cov=pc)
    oc = o.dense_covariances(o.posterior_linear(As, bs, **g), prior_cov=pc)
    e = o.AemEnsemble(of, oc, sigma, x0, 99, 3, 4, 1, chain0=11)
    out = e.advance(150, want_sub=True)
    _same_as_oracle(dev, e, out)
    assert np.array_equal(dev["sub_acc"], out["sub_acc"])
    _same_error_model(dev, e)


@pytest.mark.parametrize("build", ["matrix", "lds"])
@pytest.mark.parametrize("name", ["f16_aem_linear_d128.npz", "f16b_aem_linear_d48_rows12.npz"])
def test_f16_adaptive_error_model_d128_matrix_cores(ffi, ctx, golden, monkeypatch, name, build):
    """The adaptive error model on the matrix-core kernel (linear_tile_aem2_kernel: per-output error-model
    arrays in the tiled layout, cache keyed by the identity of the points): the live reference's chains at
    d = 128 (sha256) and at d = 48 with TWELVE data rows (two passes of eight rows over the coarse map), decisions of
    both levels, what every chain learned; == oracle bit for bit.  Both builds: the coarse matrix in LDS and the chains'
    error model in HBM (default), and the error model in LDS with both matrices streamed (linear_tile_aem2_lds_kernel;
    more than eight data rows keep the first)."""
    import hashlib
    monkeypatch.setenv("YMC_TILE_AEM", build)
    g = golden(name)
    n, S = int(g["n_states"]), int(g["S"])
    common = dict(data=g["data"], noise_precision=g["noise_precision"], prior_mean=g["prior_mean"],
                  prior_precision=g["prior_precision"])
    dev = _aem_device_run(ffi, ctx, g, dict(common, A=g["A_fine"], b=g["b_fine"]), dict(common, A=g["A_coarse"], b=g["b_coarse"]),
                          g["sigma"], g["x0"], int(g["seed"]), n - 1, S, int(g["min_data_size"]),
                          int(g["use_noise_heuristic"]), splits=3)
    e = o.AemEnsemble(o.posterior_linear(g["A_fine"], g["b_fine"], **common),
                      o.posterior_linear(g["A_coarse"], g["b_coarse"], **common), g["sigma"], g["x0"],
                      int(g["seed"]), S, int(g["min_data_size"]), int(g["use_noise_heuristic"]))
    out = e.advance(n - 1, want_sub=True)
    _same_as_oracle(dev, e, out)
    assert np.array_equal(dev["sub_acc"], out["sub_acc"])
    _same_error_model(dev, e)
    assert np.array_equal(dev["acc"], g["acc"]) and np.array_equal(dev["sub_acc"], g["sub_acc"])
    full = np.concatenate([g["x0"][None], dev["traj"]])
    assert hashlib.sha256(np.ascontiguousarray(full).tobytes()).hexdigest() == str(g["traj_sha256"])
    assert np.array_equal(dev["err"]["n"], g["err_n"])
    assert np.allclose(dev["err"]["precision"], g["aem_prec"], rtol=1e-11, atol=0)


@pytest.mark.parametrize("d,m,nd,K", [(128, 128, 5, 70), (9, 3, 2, 33), (16, 40, 1, 64), (40, 70, 8, 50), (5, 30, 6, 40),
                                      (40, 70, 9, 50), (24, 33, 21, 40), (128, 96, 14, 36)])
@pytest.mark.parametrize("build", ["matrix", "lds"])
def test_tile_kernel_adaptive_error_model_other_shapes(ffi, ctx, monkeypatch, d, m, nd, K, build):
    monkeypatch.setenv("YMC_TILE_AEM", build)
    rng = np.random.default_rng(40 + d + m)
    A = rng.normal(size=(m, d)) / np.sqrt(d)
    b = rng.normal(size=m) * 0.3
    As, bs = A + 0.05 * rng.normal(size=(m, d)), b + 0.1 * rng.normal(size=m)
    theta = rng.normal(size=d)
    data = np.array([A @ theta + b + 0.5 * rng.normal(size=m) for _ in range(nd)])
    common = dict(data=data, noise_precision=np.full(m, 4.0) * rng.uniform(0.8, 1.2, size=m),
                  prior_mean=np.zeros(d), prior_precision=np.full(d, 0.25))
    sigma = np.full(d, 0.5 / np.sqrt(d))
    x0 = rng.normal(size=(K, d)) * 0.3
    for heur in (1, 0):
        dev = _aem_device_run(ffi, ctx, None, dict(common, A=A, b=b), dict(common, A=As, b=bs), sigma, x0, 31338, 60, 3,
                              4, heur, splits=2, chain0=91)
        e = o.AemEnsemble(o.posterior_linear(A, b, **common), o.posterior_linear(As, bs, **common), sigma, x0, 31338, 3,
                          4, heur, chain0=91)
        out = e.advance(60, want_sub=True)
        _same_as_oracle(dev, e, out)
        assert np.array_equal(dev["sub_acc"], out["sub_acc"])
        _same_error_model(dev, e)
    assert dev["err"]["n"].max() > 4


def test_aem_full_size_invariants(ffi, ctx, golden):
    """65 536 chains: the error model has seen exactly the accepted fine proposals of its chain
    (aem.py:26-56), the inflated precision exists iff nData > minDataSize and is below the data
    precision, and a sample of chains equals the oracle."""
    g = golden("f12_aem_linear.npz")
    common = dict(data=g["data"], noise_precision=g["noise_precision"], prior_mean=g["prior_mean"],
                  prior_precision=g["prior_precision"])
    K, n, S, md = 65536, 120, 4, 10
    x0 = np.zeros((K, 2))
    dev = _aem_device_run(ffi, ctx, g, dict(common, A=g["A_fine"], b=g["b_fine"]),
                          dict(common, A=g["A_coarse"], b=g["b_coarse"]), g["sigma"], x0, 555, n, S, md, 1)
    err = dev["err"]
    assert np.array_equal(err["n"], dev["summary"]["accepted"])
    assert np.array_equal(err["n"], dev["acc"].sum(axis=0).astype(np.uint64))
    assert np.array_equal(err["has_precision"], err["n"] > md)
    assert err["has_precision"].mean() > 0.9
    assert np.all(err["precision"][err["has_precision"]] < g["noise_precision"][None, :])
    sel = slice(4096, 4096 + 64)
    e = o.AemEnsemble(o.posterior_linear(g["A_fine"], g["b_fine"], **common),
                      o.posterior_linear(g["A_coarse"], g["b_coarse"], **common), g["sigma"], x0[sel], 555, S, md, 1,
                      chain0=4096)
    out = e.advance(n)
    assert np.array_equal(dev["traj"][:, sel], out["traj"]) and np.array_equal(err["mean"][sel], e.aem_mean)


def test_f13_dense_proposal_covariance(ffi, ctx, golden):
    from yagremcmc_amd import lowering
    g = golden("f13_mrw_2d_dense_proposal.npz")
    n = int(g["n_states"])
    low = _gauss_low(ffi, g["mean"], g["whiten"], g["log_norm"])
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, np.ones(2), chol=g["chol"])
    run = ffi.Run(prob, int(g["seed"]), g["x0"].shape[0], g["x0"], n - 1, _flags(ffi, True))
    run.advance(200)
    run.advance(n - 201)
    run.sync()
    dev = dict(traj=run.trajectory(), acc=run.accept(), logpost=run.logpost(), summary=run.summary(), state=run.state())
    e = o.Ensemble(o.gauss_whitened(g["mean"], g["whiten"], float(g["log_norm"])), np.ones(2), g["x0"],
                   int(g["seed"]), chol=g["chol"])
    out = e.advance(n - 1)
    _same_as_oracle(dev, e, out)
    assert np.array_equal(dev["acc"], g["acc"])                       # the reference's decisions
    assert np.abs(dev["traj"] - g["traj"][1:]).max() <= 1e-12          # its states to rounding (BLAS order)


@pytest.mark.parametrize("d,kind", [(1, "gauss"), (3, "gauss"), (8, "gauss"), (5, "linear"), (2, "lv")])
def test_dense_proposal_other_shapes_vs_oracle(ffi, ctx, d, kind):
    from yagremcmc_amd import lowering
    from test_gpu_fuzz import _lowered
    rng = np.random.default_rng(40 + d)
    low, p = _lowered(ffi, kind, d, rng)
    A = rng.normal(size=(d, d))
    L = np.linalg.cholesky(0.05 * (A @ A.T + d * np.eye(d)))
    x0 = rng.normal(size=(300, d)) * 0.3 - (np.array([0.6, 0.3]) if kind == "lv" else 0.0)
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, np.ones(d), chol=L)
    run = ffi.Run(prob, 77, 300, x0, 40, _flags(ffi, True), 123)
    run.advance(40)
    run.sync()
    dev = dict(traj=run.trajectory(), acc=run.accept(), logpost=run.logpost(), summary=run.summary(), state=run.state())
    e = o.Ensemble(p, np.ones(d), x0, 77, chain0=123, chol=L)
    out = e.advance(40)
    assert 0.02 < out["acc"].mean() < 0.98
    _same_as_oracle(dev, e, out)


# ---------------------------------------------------------------------------
# other dimensions of the lane-per-chain kernels vs the oracle
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("d", [1, 3, 4, 5, 8])
def test_gauss_target_other_dimensions(ffi, ctx, d):
    rng = np.random.default_rng(100 + d)
    a = rng.normal(size=(d, d))
    cov = a @ a.T + d * np.eye(d)
    mean = rng.normal(size=d)
    p = o.gauss_from_cov(mean, cov)
    low = _gauss_low(ffi, p["mean"], p["whiten"], p["log_norm"])
    sigma = rng.uniform(0.5, 1.5, size=d)
    x0 = rng.normal(size=(300, d))
    dev = _device_run(ffi, ctx, low, sigma, x0, 777, 150, chain0=12345)
    e = o.Ensemble(p, sigma, x0, 777, chain0=12345)
    _same_as_oracle(dev, e, e.advance(150))


@pytest.mark.parametrize("d,m,nd", [(1, 1, 1), (3, 5, 4), (8, 8, 6), (5, 2, 9)])
def test_linear_posterior_other_shapes(ffi, ctx, d, m, nd):
    rng = np.random.default_rng(200 + d)
    g = dict(A=rng.normal(size=(m, d)), b=rng.normal(size=m), data=rng.normal(size=(nd, m)),
             noise_precision=rng.uniform(0.5, 3, size=m), prior_mean=rng.normal(size=d),
             prior_precision=rng.uniform(0.1, 2, size=d))
    low = _posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, g, fwd_matrix=g["A"], fwd_shift=g["b"])
    p = o.posterior_linear(g["A"], g["b"], g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
    sigma = np.full(d, 0.3)
    x0 = rng.normal(size=(257, d))
    dev = _device_run(ffi, ctx, low, sigma, x0, 31337, 120)
    e = o.Ensemble(p, sigma, x0, 31337)
    _same_as_oracle(dev, e, e.advance(120))


# ---------------------------------------------------------------------------
# BASELINE sizes: size-independent properties + oracle on slices of the ensemble
# ---------------------------------------------------------------------------

def test_c2_full_size_properties(ffi, ctx, golden):
    """65 536 chains on the 2-D Gaussian target (BASELINE configs[1])."""
    from yagremcmc_amd import lowering
    g = golden("f2_mrw_2d.npz")
    K, T = 65536, 400
    low = _gauss_low(ffi, g["mean"], g["whiten"], g["log_norm"])
    x0 = np.tile([-8.0, -7.0], (1, 1))
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, g["sigma"])
    run = ffi.Run(prob, 20250103, K, x0, T, _flags(ffi))
    run.advance(T)
    traj, acc = run.trajectory(), run.accept()
    # (i) slices of the big ensemble equal the oracle run on those global chain ids
    p = o.gauss_whitened(g["mean"], g["whiten"], float(g["log_norm"]))
    for c0 in (0, 32768 - 32, 65536 - 64):
        e = o.Ensemble(p, g["sigma"], np.tile([-8.0, -7.0], (64, 1)), 20250103, chain0=c0)
        out = e.advance(T)
        assert np.array_equal(traj[:, c0:c0 + 64], out["traj"]) and np.array_equal(acc[:, c0:c0 + 64], out["acc"])
    # (ii) sharding invariance: the upper half as its own run with a chain offset
    half = ffi.Run(prob, 20250103, K // 2, x0, T, _flags(ffi), K // 2)
    half.advance(T)
    assert np.array_equal(half.trajectory(), traj[:, K // 2:]) and np.array_equal(half.accept(), acc[:, K // 2:])
    # (iii) a rejected transition repeats the state, an accepted one moves it
    prev = np.concatenate([np.broadcast_to(x0, (1, K, 2)), traj[:-1]])
    same = (traj == prev).all(axis=2)
    assert np.array_equal(same, acc == 0)
    # (iv) accept counters == column sums of the mask; Welford mean == mean of pre-transition states
    s = run.summary()
    assert np.array_equal(s["accepted"], acc.sum(axis=0).astype(np.uint64))
    assert np.allclose(s["wf_mean"], prev.mean(axis=0), rtol=0, atol=1e-12)
    # (v) the ensemble has reached the target: moments over chains at the last step
    last = traj[-1]
    assert np.abs(last.mean(axis=0) - g["mean"]).max() < 0.03
    assert np.abs(np.cov(last.T) - g["cov"]).max() < 0.06
    assert (s["status"] == 0).all()


def test_record_capacity_and_rewind(ffi, ctx, golden):
    from yagremcmc_amd import lowering
    g = golden("f2_mrw_2d.npz")
    low = _gauss_low(ffi, g["mean"], g["whiten"], g["log_norm"])
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, g["sigma"])
    run = ffi.Run(prob, 1, 128, g["x0"][:1], 50, _flags(ffi))
    run.advance(50)
    first = run.trajectory()
    with pytest.raises(ffi.YmcError) as err:
        run.advance(1)
    assert err.value.code == ffi.ERR_CAPACITY
    run.rewind_record()
    run.advance(50)
    both = np.concatenate([first, run.trajectory()])
    e = o.Ensemble(o.gauss_whitened(g["mean"], g["whiten"], float(g["log_norm"])), g["sigma"],
                   np.tile(g["x0"][:1], (128, 1)), 1)
    assert np.array_equal(both, e.advance(100)["traj"])
    assert run.steps_done() == (100, 50)
    avg_ms, launches = run.kernel_time_ms()
    assert launches == 2 and avg_ms > 0


# ---------------------------------------------------------------------------
# large dimension: the fp64-MFMA tile kernel (BASELINE configs[2] shapes)
# ---------------------------------------------------------------------------

def test_f3_linear_d128_matrix_cores(ffi, ctx, golden):
    import hashlib
    g = golden("f3_linear_d128.npz")
    n = int(g["n_states"])
    low = _posterior_low(ffi, ffi.POSTERIOR_LINEAR, 128, g, fwd_matrix=np.asarray(g["A"]), fwd_shift=np.asarray(g["b"]))
    dev = _device_run(ffi, ctx, low, g["sigma"], g["x0"], int(g["seed"]), n - 1, moments=False, splits=2)
    p = o.posterior_linear(g["A"], g["b"], g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
    e = o.Ensemble(p, g["sigma"], g["x0"], int(g["seed"]))
    out = e.advance(n - 1)
    # bit-exact against the oracle: the f64 MFMA is a k-ordered fma chain
    assert np.array_equal(dev["acc"], out["acc"])
    assert np.array_equal(dev["traj"], out["traj"])
    assert np.array_equal(dev["logpost"], out["logpost"])
    assert np.array_equal(dev["summary"]["accepted"], e.acc_count)
    assert np.array_equal(dev["state"][0], e.x) and np.array_equal(dev["state"][1], e.lp)
    # and against the live reference's outputs
    _same_as_reference(dev, g)
    assert np.array_equal(dev["traj"][:, :2, :], g["traj_head"][1:])
    full = np.concatenate([g["x0"][None], dev["traj"]])
    assert hashlib.sha256(np.ascontiguousarray(full).tobytes()).hexdigest() == str(g["traj_sha256"])
    # batched evaluate_log / forward at d = 128
    from yagremcmc_amd import lowering
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, g["sigma"])
    xs = dev["traj"][-1]
    assert np.array_equal(prob.logdensity(xs)[0], o.logdensity(p, xs)[0])
    assert np.array_equal(prob.forward(xs), o.forward(p, xs))


@pytest.mark.parametrize("name", ["f22_linear_d192_m160.npz", "f22b_linear_d320_m48.npz"])
def test_f22_linear_d192_m160_streamed_matrix(ffi, ctx, golden, name):
    """The forward matrix streamed from L2 (it does not fit the LDS) and 192 parameter components per chain (KS = 64,
    one wave per SIMD); 320 components on the wide path: device == oracle == the live reference's chains (fixtures F22 /
    F22b, exampleSetup.py:42-52, likelihood.py:74-84), the IAT of the record included ('mean' sums the components in
    NumPy's pairwise order)."""
    import hashlib
    g = golden(name)
    n = int(g["n_states"])
    low = _posterior_low(ffi, ffi.POSTERIOR_LINEAR, g["A"].shape[1], g, fwd_matrix=np.asarray(g["A"]), fwd_shift=np.asarray(g["b"]))
    dev = _device_run(ffi, ctx, low, g["sigma"], g["x0"], int(g["seed"]), n - 1, moments=True, splits=3)
    p = o.posterior_linear(g["A"], g["b"], g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
    e = o.Ensemble(p, g["sigma"], g["x0"], int(g["seed"]))
    out = e.advance(n - 1)
    _same_as_oracle(dev, e, out)
    _same_as_reference(dev, g)
    assert np.array_equal(dev["traj"][:, :2, :], g["traj_head"][1:])
    full = np.concatenate([g["x0"][None], dev["traj"]])
    assert hashlib.sha256(np.ascontiguousarray(full).tobytes()).hexdigest() == str(g["traj_sha256"])
    from yagremcmc_amd import lowering
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, g["sigma"])
    xs = dev["traj"][-1]
    assert np.array_equal(prob.logdensity(xs)[0], o.logdensity(p, xs)[0])
    assert np.array_equal(prob.forward(xs), o.forward(p, xs))
    # post-processing on the d = 192 record, read in place in the tiled layout
    run = ffi.Run(prob, int(g["seed"]), g["x0"].shape[0], g["x0"], n - 1, ffi.RECORD_TRAJECTORY)
    run.advance(n - 1)
    states = run.trajectory()
    for method in (0, 1):
        assert np.array_equal(run.iat(0, n - 1, method), ctx.iat_states(states, method))
    series = np.ascontiguousarray(states.mean(axis=2).T)        # NumPy's own row means
    assert np.array_equal(run.iat(0, n - 1, 0), ctx.iat_series(series)[0])


@pytest.mark.parametrize("stream", ["0", "1"])
@pytest.mark.parametrize("d,m,nd,K,pcn", [(128, 160, 3, 150, False), (200, 48, 2, 70, False), (256, 256, 2, 40, False),
                                           (130, 1000, 1, 33, False), (96, 400, 4, 129, True), (256, 20, 3, 65, True),
                                           (40, 70, 8, 50, False), (20, 24, 2, 600, True)])
def test_tile_kernel_streamed_forward_matrix(ffi, ctx, monkeypatch, stream, d, m, nd, K, pcn):
    """Shapes beyond the LDS (any m whose constants fit, d <= 256) take the forward matrix from L2 through the
    workgroup's double buffer; shapes that fit can be forced onto that build (YMC_TILE_ASTREAM=1): == oracle for the
    random walk and pCN, ragged shapes, several workgroups, waves past the last tile, across launches."""
    fits = d <= 128 and (16 * ((m + 15) // 16 + ((m + 15) // 16) % 2)) * (4 * (1 << max(2, int(np.ceil(np.log2((d + 3) // 4)))))) * 8 < 150_000
    if stream == "0" and not fits:
        pytest.skip("this shape only exists streamed")
    monkeypatch.setenv("YMC_TILE_ASTREAM", stream)
    rng = np.random.default_rng(500 + d + m)
    g = dict(A=rng.normal(size=(m, d)) / np.sqrt(d), b=rng.normal(size=m), data=rng.normal(size=(nd, m)),
             noise_precision=rng.uniform(0.5, 3, size=m), prior_mean=rng.normal(size=d) * 0.1,
             prior_precision=rng.uniform(0.1, 2, size=d))
    p = o.posterior_linear(g["A"], g["b"], g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
    x0 = rng.normal(size=(K, d)) * 0.3
    if pcn:
        from yagremcmc_amd import lowering
        g["prior_mean"] = np.zeros(d)
        p = o.posterior_linear(g["A"], g["b"], g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
        low = _posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, g, fwd_matrix=g["A"], fwd_shift=g["b"], likelihood_only=1)
        sig = np.sqrt(np.reciprocal(g["prior_precision"]))
        prob = lowering.make_problem(ctx, ffi.METHOD_PCN, low, sig, pcn_step_size=0.002)
        run = ffi.Run(prob, 31, K, x0, 25, _flags(ffi, True), 400)
        run.advance(12)
        run.advance(13)
        run.sync()
        dev = dict(traj=run.trajectory(), acc=run.accept(), logpost=run.logpost(), summary=run.summary(), state=run.state())
        p["likelihood_only"] = 1
        e = o.Ensemble(p, sig, x0, 31, chain0=400, pcn_step_size=0.002)
    else:
        low = _posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, g, fwd_matrix=g["A"], fwd_shift=g["b"])
        sig = rng.uniform(0.01, 0.04, size=d)
        dev = _device_run(ffi, ctx, low, sig, x0, 31, 25, chain0=400, splits=2)
        e = o.Ensemble(p, sig, x0, 31, chain0=400)
    out = e.advance(25)
    _same_as_oracle(dev, e, out)
    assert 0 < dev["acc"].mean() < 1


@pytest.mark.parametrize("d,m,nd,K", [(128, 128, 5, 100), (9, 3, 2, 33), (16, 16, 1, 64), (40, 70, 8, 50),
                                      (64, 200, 3, 17), (100, 20, 4, 129), (7, 30, 5, 40)])
def test_tile_kernel_other_shapes(ffi, ctx, d, m, nd, K):
    rng = np.random.default_rng(300 + d + m)
    g = dict(A=rng.normal(size=(m, d)) / np.sqrt(d), b=rng.normal(size=m), data=rng.normal(size=(nd, m)),
             noise_precision=rng.uniform(0.5, 3, size=m), prior_mean=rng.normal(size=d),
             prior_precision=rng.uniform(0.1, 2, size=d))
    low = _posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, g, fwd_matrix=g["A"], fwd_shift=g["b"])
    p = o.posterior_linear(g["A"], g["b"], g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
    sigma = rng.uniform(0.02, 0.08, size=d)
    x0 = rng.normal(size=(K, d))
    dev = _device_run(ffi, ctx, low, sigma, x0, 4711, 60, chain0=1000, moments=True, splits=3)
    e = o.Ensemble(p, sigma, x0, 4711, chain0=1000)
    out = e.advance(60)
    assert 0.02 < out["acc"].mean() < 0.98
    _same_as_oracle(dev, e, out)   # incl. the Welford moments replayed from the record


@pytest.mark.parametrize("d,m,nd,K,h", [(128, 128, 5, 80, 0.002), (9, 3, 2, 33, 0.05), (40, 70, 8, 50, 0.01), (7, 30, 5, 40, 0.5)])
def test_tile_kernel_preconditioned_crank_nicolson(ffi, ctx, d, m, nd, K, h):
    """pCN (pcn.py:9-57) at dimensions the lane kernels do not cover: proposal sqrt(1-2h) x + sqrt(2h) xi,
    xi ~ prior, accept on the likelihood alone; bit-exact against the oracle incl. the replayed moments."""
    from yagremcmc_amd import lowering
    rng = np.random.default_rng(900 + d + m)
    g = dict(A=rng.normal(size=(m, d)) / np.sqrt(d), b=rng.normal(size=m), data=rng.normal(size=(nd, m)),
             noise_precision=rng.uniform(0.5, 3, size=m), prior_mean=np.zeros(d),
             prior_precision=rng.uniform(0.5, 2, size=d))
    low = _posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, g, fwd_matrix=g["A"], fwd_shift=g["b"], likelihood_only=1)
    prior_sigma = np.sqrt(np.reciprocal(g["prior_precision"]))
    x0 = rng.normal(size=(K, d)) * 0.3
    prob = lowering.make_problem(ctx, ffi.METHOD_PCN, low, prior_sigma, pcn_step_size=h)
    run = ffi.Run(prob, 99, K, x0, 50, _flags(ffi, True), 500)
    run.advance(20)
    run.advance(30)
    run.sync()
    dev = dict(traj=run.trajectory(), acc=run.accept(), logpost=run.logpost(), summary=run.summary(), state=run.state())
    p = o.posterior_linear(g["A"], g["b"], g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
    p["likelihood_only"] = 1
    e = o.Ensemble(p, prior_sigma, x0, 99, chain0=500, pcn_step_size=h)
    out = e.advance(50)
    assert 0.01 < out["acc"].mean() < 0.99
    _same_as_oracle(dev, e, out)


def test_f21_pcn_linear_d64_matrix_cores(ffi, ctx, golden):
    """pCN on the matrix-core kernel (`linear_tile_kernel<KS, kPropPcn>`): the LIVE reference's decisions and
    chains at d = 64, log-likelihoods to rounding; == oracle bit for bit."""
    from yagremcmc_amd import lowering
    g = golden("f21_pcn_linear_d64.npz")
    n, d, h = int(g["n_states"]), 64, float(g["step_size"])
    common = dict(data=g["data"], noise_precision=g["noise_precision"], prior_mean=g["prior_mean"], prior_precision=g["prior_precision"])
    low = _posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, common, fwd_matrix=g["A"], fwd_shift=g["b"], likelihood_only=1)
    prob = lowering.make_problem(ctx, ffi.METHOD_PCN, low, g["prior_sigma"], pcn_step_size=h)
    run = ffi.Run(prob, int(g["seed"]), g["x0"].shape[0], g["x0"], n - 1, _flags(ffi, True), 0)
    run.advance(100)
    run.advance(n - 1 - 100)
    dev = dict(traj=run.trajectory(), acc=run.accept(), logpost=run.logpost(), summary=run.summary(), state=run.state())
    assert np.array_equal(dev["acc"], g["acc"]) and np.array_equal(dev["traj"], g["traj"][1:])
    assert np.abs(dev["logpost"] - g["loglik"][1:]).max() <= 1e-12 * np.abs(g["loglik"]).max()
    p = o.posterior_linear(g["A"], g["b"], g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
    p["likelihood_only"] = 1
    e = o.Ensemble(p, g["prior_sigma"], g["x0"], int(g["seed"]), pcn_step_size=h)
    _same_as_oracle(dev, e, e.advance(n - 1))


def _dense_low(ffi, kind, dim, g, noise_cov=None, prior_cov=None, **extra):
    from scipy.linalg import solve_triangular
    low = _posterior_low(ffi, kind, dim, g, **extra)
    for field, cov in (("noise_chol_inv", noise_cov), ("prior_chol_inv", prior_cov)):
        if cov is not None:
            L = np.linalg.cholesky(np.asarray(cov, dtype=float))
            low.set_array(field, np.tril(solve_triangular(L, np.eye(L.shape[0]), lower=True)))
    return low


def test_f15_dense_noise_and_prior_covariances(ffi, ctx, golden):
    """DenseCovarianceMatrix as noise and prior model (noise.py:21-22, gaussian.py:19-24 over covariance.py:69-94):
    device == oracle bit for bit; decisions and chains == the live reference's, log-posteriors to rounding."""
    g = golden("f15_dense_noise_prior.npz")
    lin = o.dense_covariances(o.posterior_linear(g["A"], g["b"], g["data"], np.ones(4), g["prior_mean"], np.ones(3)),
                              noise_cov=g["noise_cov"], prior_cov=g["prior_cov"])
    lv = o.dense_covariances(
        o.posterior_lv(g["lv_design"], float(g["lv_t_final"]), float(g["lv_alpha"]), float(g["lv_gamma"]),
                       int(g["lv_rk4_steps"]), g["lv_data"], np.ones(2), g["lv_prior_mean"], g["lv_prior_precision"]),
        noise_cov=g["lv_noise_cov"])
    common = dict(data=g["data"], noise_precision=np.ones(4), prior_mean=g["prior_mean"], prior_precision=np.ones(3))
    low = _dense_low(ffi, ffi.POSTERIOR_LINEAR, 3, common, g["noise_cov"], g["prior_cov"], fwd_matrix=g["A"], fwd_shift=g["b"])
    n = int(g["n_states"])
    dev = _device_run(ffi, ctx, low, g["sigma"], g["x0"], int(g["seed"]), n - 1, splits=2)
    e = o.Ensemble(lin, g["sigma"], g["x0"], int(g["seed"]))
    _same_as_oracle(dev, e, e.advance(n - 1))
    _same_as_reference(dev, g)
    lvc = dict(data=g["lv_data"], noise_precision=np.ones(2), prior_mean=g["lv_prior_mean"], prior_precision=g["lv_prior_precision"])
    for rows in ("0", "1"):       # one lane per chain, and one lane per (chain, design row)
        import os
        os.environ["YMC_LV_ROWS"] = rows
        try:
            low = _dense_low(ffi, ffi.POSTERIOR_LV, 2, lvc, g["lv_noise_cov"], None, design=np.asarray(g["lv_design"]),
                             t_final=float(g["lv_t_final"]), alpha=float(g["lv_alpha"]), gamma=float(g["lv_gamma"]),
                             rk4_steps=int(g["lv_rk4_steps"]))
            nl = int(g["lv_n_states"])
            dev = _device_run(ffi, ctx, low, g["lv_sigma"], g["lv_x0"], int(g["lv_seed"]), nl - 1)
        finally:
            del os.environ["YMC_LV_ROWS"]
        e = o.Ensemble(lv, g["lv_sigma"], g["lv_x0"], int(g["lv_seed"]))
        _same_as_oracle(dev, e, e.advance(nl - 1))
        _same_as_reference(dev, dict(acc=g["lv_acc"], traj=g["lv_traj"], logpost=g["lv_logpost"]))


@pytest.mark.parametrize("d,m,nd,method", [(1, 1, 3, "mrw"), (5, 8, 2, "mrw"), (8, 3, 4, "mlda2"), (2, 6, 5, "mlda2")])
def test_dense_covariances_other_shapes_vs_oracle(ffi, ctx, d, m, nd, method):
    rng = np.random.default_rng(1000 + 10 * d + m)

    def spd(n):
        B = rng.normal(size=(n, n))
        return B @ B.T + n * 0.2 * np.eye(n)

    def level(scale):
        g = dict(data=rng.normal(size=(nd, m)), noise_precision=np.ones(m), prior_mean=rng.normal(size=d) * 0.2,
                 prior_precision=np.ones(d))
        A, b, nc, pc = scale * rng.normal(size=(m, d)), rng.normal(size=m), spd(m), spd(d)
        return (_dense_low(ffi, ffi.POSTERIOR_LINEAR, d, g, nc, pc, fwd_matrix=A, fwd_shift=b),
                o.dense_covariances(o.posterior_linear(A, b, g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"]),
                                    noise_cov=nc, prior_cov=pc))

    (lf, pf), (lc, pc_) = level(1.0), level(0.8)
    sigma, x0 = rng.uniform(0.1, 0.4, size=d), rng.normal(size=(45, d))
    if method == "mrw":
        dev = _device_run(ffi, ctx, lf, sigma, x0, 77, 40, chain0=5, splits=2)
        e = o.Ensemble(pf, sigma, x0, 77, chain0=5)
        _same_as_oracle(dev, e, e.advance(40))
    else:
        dev = _device_run(ffi, ctx, lf, sigma, x0, 77, 30, method=ffi.METHOD_MLDA2, sur=lc, S=3, chain0=5)
        e = o.Ensemble(pf, sigma, x0, 77, chain0=5, surrogate=pc_, sub_chain_length=3)
        out = e.advance(30, want_sub=True)
        _same_as_oracle(dev, e, out)


def test_f19_dense_noise_d24_matrix_cores(ffi, ctx, golden):
    """A dense 24 x 24 noise covariance on the matrix-core path: the library whitens the problem once
    (ymc_lower_times) and runs the diagonal kernels.  Random walk and two-level delayed acceptance: the LIVE
    reference's decisions and chains, log-posteriors to rounding, == oracle bit for bit; ymc_forward_eval still
    returns A x + b."""
    from yagremcmc_amd import lowering
    g = golden("f19_dense_noise_d24.npz")
    d = 24
    common = dict(data=g["data"], noise_precision=np.ones(d), prior_mean=g["prior_mean"], prior_precision=g["prior_precision"])
    fine = _dense_low(ffi, ffi.POSTERIOR_LINEAR, d, common, g["noise_cov"], None, fwd_matrix=g["A"], fwd_shift=g["b"])
    coarse = _dense_low(ffi, ffi.POSTERIOR_LINEAR, d, common, g["noise_cov"], None, fwd_matrix=g["A_coarse"], fwd_shift=g["b_coarse"])
    base = lambda A, b: o.posterior_linear(A, b, g["data"], np.ones(d), g["prior_mean"], g["prior_precision"])
    of = o.dense_covariances(base(g["A"], g["b"]), noise_cov=g["noise_cov"])
    oc = o.dense_covariances(base(g["A_coarse"], g["b_coarse"]), noise_cov=g["noise_cov"])
    n = int(g["n_states"])
    dev = _device_run(ffi, ctx, fine, g["sigma"], g["x0"], int(g["seed"]), n - 1, splits=2)
    _same_as_reference(dev, g)
    e = o.Ensemble(of, g["sigma"], g["x0"], int(g["seed"]))
    _same_as_oracle(dev, e, e.advance(n - 1))
    n2, S = int(g["ml_n_states"]), int(g["ml_S"])
    dev = _device_run(ffi, ctx, fine, g["sigma"], g["ml_x0"], int(g["ml_seed"]), n2 - 1, method=ffi.METHOD_MLDA2, sur=coarse,
                      S=S, splits=3)
    _same_as_reference(dev, dict(acc=g["ml_acc"], traj=g["ml_traj"], logpost=g["ml_logpost"]))
    assert np.array_equal(dev["sub_acc"], g["ml_sub_acc"])
    e = o.Ensemble(of, g["sigma"], g["ml_x0"], int(g["ml_seed"]), surrogate=oc, sub_chain_length=S)
    out = e.advance(n2 - 1, want_sub=True)
    _same_as_oracle(dev, e, out)
    assert np.array_equal(dev["sub_acc"], out["sub_acc"])
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW, fine, g["sigma"])
    x = np.random.default_rng(0).standard_normal((20, d))
    assert np.array_equal(prob.forward(x), o.forward(of, x))
    assert np.allclose(prob.forward(x), x @ g["A"].T + g["b"], rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("d,m,nd,K,method", [(128, 128, 3, 40, "mrw"), (9, 3, 2, 33, "pcn"), (16, 40, 1, 64, "mlda2"),
                                             (5, 30, 4, 50, "mrw"), (40, 20, 2, 129, "nested")])
def test_tile_kernel_dense_noise_other_shapes(ffi, ctx, d, m, nd, K, method):
    """Dense noise covariances on every matrix-core sampler that admits them (the adaptive error model refuses
    non-diagonal noise like the reference's AEMNoise), at ragged shapes; levels with their own covariances."""
    rng = np.random.default_rng(1900 + d + m)

    def model(scale):
        B = rng.normal(size=(m, m)) / np.sqrt(m)
        nc = B @ B.T + 0.4 * np.eye(m)
        g = dict(data=rng.normal(size=(nd, m)), noise_precision=np.ones(m), prior_mean=rng.normal(size=d) * 0.1,
                 prior_precision=rng.uniform(0.1, 2, size=d))
        A, b = scale * rng.normal(size=(m, d)) / np.sqrt(d), rng.normal(size=m)
        p = o.dense_covariances(o.posterior_linear(A, b, g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"]),
                                noise_cov=nc, whiten=True)
        return _dense_low(ffi, ffi.POSTERIOR_LINEAR, d, g, nc, None, fwd_matrix=A, fwd_shift=b), p

    sigma, x0 = rng.uniform(0.02, 0.08, size=d), rng.normal(size=(K, d))
    if method == "mrw":
        (lf, pf), = [model(1.0)]
        dev = _device_run(ffi, ctx, lf, sigma, x0, 78, 40, chain0=7, splits=2)
        e = o.Ensemble(pf, sigma, x0, 78, chain0=7)
        _same_as_oracle(dev, e, e.advance(40))
    elif method == "pcn":
        lf, pf = model(1.0)
        lf.desc.likelihood_only, pf["likelihood_only"] = 1, 1
        prior_sigma = np.sqrt(np.reciprocal(pf["prior_precision"]))
        from yagremcmc_amd import lowering
        prob = lowering.make_problem(ctx, ffi.METHOD_PCN, lf, prior_sigma, pcn_step_size=0.02)
        run = ffi.Run(prob, 78, K, x0, 40, _flags(ffi, True), 7)
        run.advance(40)
        dev = dict(traj=run.trajectory(), acc=run.accept(), logpost=run.logpost(), summary=run.summary(), state=run.state())
        e = o.Ensemble(pf, prior_sigma, x0, 78, chain0=7, pcn_step_size=0.02)
        _same_as_oracle(dev, e, e.advance(40))
    elif method == "mlda2":
        (lf, pf), (lc, pc_) = model(1.0), model(0.9)
        dev = _device_run(ffi, ctx, lf, sigma, x0, 78, 30, method=ffi.METHOD_MLDA2, sur=lc, S=3, chain0=7, splits=2)
        e = o.Ensemble(pf, sigma, x0, 78, chain0=7, surrogate=pc_, sub_chain_length=3)
        out = e.advance(30, want_sub=True)
        _same_as_oracle(dev, e, out)
        assert np.array_equal(dev["sub_acc"], out["sub_acc"])
    else:
        models = [model(1.0 - 0.03 * i) for i in range(3)]
        (lt, pt), surs = models[-1], models[:-1]
        dev = _nested_device_run(ffi, ctx, lt, [l for l, _ in surs], [2, 3], sigma, x0, 78, 20, splits=2, chain0=7)
        P, Q, steps = o.nested_levels([p for _, p in surs], [2, 3], pt)
        e = o.NestedEnsemble(P, Q, steps, sigma, x0, 78, chain0=7)
        out = e.advance(20)
        _same_as_oracle(dev, e, out)
        assert np.array_equal(dev["level_acc"], e.level_acc)


@pytest.mark.parametrize("name", ["f20_dense_prior_noise_d20.npz", "f20b_dense_prior_noise_d128.npz"])
def test_f20_dense_prior_and_noise_d20_matrix_cores(ffi, ctx, golden, name):
    """Dense prior and noise covariances on the matrix-core path (the prior as d more outputs of the forward
    map; at d = 128 that takes the map beyond the LDS: streamed): the LIVE reference's decisions and chains,
    log-posteriors to rounding, == oracle bit for bit; ymc_forward_eval returns the model's own outputs."""
    from yagremcmc_amd import lowering
    g = golden(name)
    d = g["A"].shape[1]
    common = dict(data=g["data"], noise_precision=np.ones(d), prior_mean=g["prior_mean"], prior_precision=np.ones(d))
    low = _dense_low(ffi, ffi.POSTERIOR_LINEAR, d, common, g["noise_cov"], g["prior_cov"], fwd_matrix=g["A"], fwd_shift=g["b"])
    p = o.dense_covariances(o.posterior_linear(g["A"], g["b"], g["data"], np.ones(d), g["prior_mean"], np.ones(d)),
                            noise_cov=g["noise_cov"], prior_cov=g["prior_cov"])
    n = int(g["n_states"])
    dev = _device_run(ffi, ctx, low, g["sigma"], g["x0"], int(g["seed"]), n - 1, splits=2)
    _same_as_reference(dev, g)
    e = o.Ensemble(p, g["sigma"], g["x0"], int(g["seed"]))
    _same_as_oracle(dev, e, e.advance(n - 1))
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, g["sigma"])
    x = np.random.default_rng(0).standard_normal((20, d))
    fwd = prob.forward(x)
    assert fwd.shape == (20, d) and np.array_equal(fwd, o.forward(p, x))
    assert np.array_equal(prob.logdensity(x)[0], o.logdensity(p, x)[0])


@pytest.mark.parametrize("d,m,nd,K,method", [(64, 64, 3, 40, "mrw"), (9, 3, 2, 33, "mrw"), (16, 40, 1, 64, "mlda2"),
                                             (5, 30, 4, 50, "mrw"), (24, 20, 2, 129, "nested")])
def test_tile_kernel_dense_prior_other_shapes(ffi, ctx, d, m, nd, K, method):
    """Dense prior (and noise) covariances on the matrix-core samplers at ragged shapes; levels with their own
    covariances and prior means."""
    rng = np.random.default_rng(2000 + d + m)

    def spd(n, ridge):
        B = rng.normal(size=(n, n)) / np.sqrt(n)
        return B @ B.T + ridge * np.eye(n)

    def model(scale, noise=True):
        nc, pc = (spd(m, 0.4) if noise else None), spd(d, 1.0)
        g = dict(data=rng.normal(size=(nd, m)), noise_precision=rng.uniform(0.5, 2, size=m) if not noise else np.ones(m),
                 prior_mean=rng.normal(size=d) * 0.3, prior_precision=np.ones(d))
        A, b = scale * rng.normal(size=(m, d)) / np.sqrt(d), rng.normal(size=m)
        p = o.dense_covariances(o.posterior_linear(A, b, g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"]),
                                noise_cov=nc, prior_cov=pc, whiten=True)
        return _dense_low(ffi, ffi.POSTERIOR_LINEAR, d, g, nc, pc, fwd_matrix=A, fwd_shift=b), p

    sigma, x0 = rng.uniform(0.02, 0.08, size=d), rng.normal(size=(K, d))
    if method == "mrw":
        lf, pf = model(1.0, noise=(d != 9))
        dev = _device_run(ffi, ctx, lf, sigma, x0, 79, 40, chain0=9, splits=2)
        e = o.Ensemble(pf, sigma, x0, 79, chain0=9)
        _same_as_oracle(dev, e, e.advance(40))
    elif method == "mlda2":
        (lf, pf), (lc, pc_) = model(1.0), model(0.9, noise=False)
        dev = _device_run(ffi, ctx, lf, sigma, x0, 79, 30, method=ffi.METHOD_MLDA2, sur=lc, S=3, chain0=9, splits=2)
        e = o.Ensemble(pf, sigma, x0, 79, chain0=9, surrogate=pc_, sub_chain_length=3)
        out = e.advance(30, want_sub=True)
        _same_as_oracle(dev, e, out)
        assert np.array_equal(dev["sub_acc"], out["sub_acc"])
    else:
        models = [model(1.0 - 0.03 * i) for i in range(3)]
        (lt, pt), surs = models[-1], models[:-1]
        dev = _nested_device_run(ffi, ctx, lt, [l for l, _ in surs], [2, 3], sigma, x0, 79, 20, splits=2, chain0=9)
        P, Q, steps = o.nested_levels([p for _, p in surs], [2, 3], pt)
        e = o.NestedEnsemble(P, Q, steps, sigma, x0, 79, chain0=9)
        out = e.advance(20)
        _same_as_oracle(dev, e, out)
        assert np.array_equal(dev["level_acc"], e.level_acc)


def test_tile_kernel_dense_prior_limits(ffi, ctx):
    """What the dense prior on the matrix-core path cannot be: larger than the LDS together with the forward map,
    the prior of pCN, the prior of the adaptive error model."""
    from yagremcmc_amd import lowering
    rng = np.random.default_rng(3)

    def low(d, m, likelihood_only=0):
        B = rng.normal(size=(d, d)) / np.sqrt(d)
        g = dict(data=rng.normal(size=(2, m)), noise_precision=np.ones(m), prior_mean=np.zeros(d), prior_precision=np.ones(d))
        return _dense_low(ffi, ffi.POSTERIOR_LINEAR, d, g, None, B @ B.T + np.eye(d), fwd_matrix=rng.normal(size=(m, d)),
                          fwd_shift=np.zeros(m), likelihood_only=likelihood_only)

    # delayed acceptance (two-level and nested) streams what does not fit the LDS ((128 + 128) outputs x 128 per level)
    lowering.make_problem(ctx, ffi.METHOD_MLDA2, low(128, 128), np.ones(128), surrogate=low(128, 128), sub_chain_length=2)
    _nested_device_run(ffi, ctx, low(128, 128), [low(128, 128), low(128, 128)], [2, 2], np.full(128, 0.01), np.zeros((4, 128)), 1, 2)
    lowering.make_problem(ctx, ffi.METHOD_MRW, low(96, 16), np.ones(96))        # (16 + 96) outputs x 96 fit
    # the random walk streams its matrix: a dense prior at d = 128 (and beyond) == oracle
    for d, m, K in ((128, 128, 40), (160, 30, 33)):
        B = rng.normal(size=(d, d)) / np.sqrt(d)
        pc = B @ B.T + np.eye(d)
        g = dict(data=rng.normal(size=(2, m)), noise_precision=rng.uniform(0.5, 2, m), prior_mean=rng.normal(size=d) * 0.1,
                 prior_precision=np.ones(d))
        A, b = rng.normal(size=(m, d)) / np.sqrt(d), rng.normal(size=m)
        lw = _dense_low(ffi, ffi.POSTERIOR_LINEAR, d, g, None, pc, fwd_matrix=A, fwd_shift=b)
        po = o.dense_covariances(o.posterior_linear(A, b, g["data"], g["noise_precision"], g["prior_mean"], np.ones(d)), prior_cov=pc)
        sig, x0 = np.full(d, 0.02), rng.normal(size=(K, d)) * 0.2
        dev = _device_run(ffi, ctx, lw, sig, x0, 8, 20, splits=2)
        e = o.Ensemble(po, sig, x0, 8)
        _same_as_oracle(dev, e, e.advance(20))
    with pytest.raises(ffi.YmcError, match="diagonal prior"):
        lowering.make_problem(ctx, ffi.METHOD_PCN, low(16, 16, likelihood_only=1), np.ones(16), pcn_step_size=0.1)
    with pytest.raises(ffi.YmcError, match="adaptive error model"):
        lowering.make_problem(ctx, ffi.METHOD_AEM2, low(16, 16), np.ones(16), surrogate=low(16, 16), sub_chain_length=2,
                              aem=(5, True))


def _dense_tile_run(ffi, ctx, low, L, x0, seed, nsteps, chain0=0, splits=2):
    from yagremcmc_amd import lowering
    d = x0.shape[1]
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, np.ones(d), chol=np.tril(L))
    run = ffi.Run(prob, seed, x0.shape[0], x0, nsteps, _flags(ffi, True), chain0)
    base, rem = divmod(nsteps, splits)
    for i in range(splits):
        run.advance(base + (1 if i < rem else 0))
    return dict(traj=run.trajectory(), acc=run.accept(), logpost=run.logpost(), summary=run.summary(), state=run.state())


def test_f17_dense_proposal_d16_matrix_cores(ffi, ctx, golden):
    """y = x + L xi on the matrix cores (L's fragments from L2, rows of L xi as MFMA fma chains): == oracle bit for
    bit; against the live reference (BLAS' dgemv order) every decision equal, states within 1e-12."""
    g = golden("f17_mrw_dense_proposal_d16.npz")
    n = int(g["n_states"])
    common = dict(data=g["data"], noise_precision=g["noise_precision"], prior_mean=g["prior_mean"], prior_precision=g["prior_precision"])
    low = _posterior_low(ffi, ffi.POSTERIOR_LINEAR, 16, common, fwd_matrix=g["A"], fwd_shift=g["b"])
    dev = _dense_tile_run(ffi, ctx, low, g["chol"], g["x0"], int(g["seed"]), n - 1)
    p = o.posterior_linear(g["A"], g["b"], g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
    e = o.Ensemble(p, np.ones(16), g["x0"], int(g["seed"]), chol=g["chol"])
    _same_as_oracle(dev, e, e.advance(n - 1))
    assert np.array_equal(dev["acc"], g["acc"]) and np.abs(dev["traj"] - g["traj"][1:]).max() <= 1e-12


@pytest.mark.parametrize("d,m,nd,K", [(128, 128, 2, 40), (9, 3, 2, 33), (20, 40, 1, 64), (100, 20, 4, 129),
                                      (128, 400, 2, 36), (192, 60, 2, 33), (256, 16, 1, 20), (40, 1000, 1, 17),
                                      (300, 24, 2, 33), (513, 100, 1, 17)])
def test_tile_kernel_dense_proposal_other_shapes(ffi, ctx, d, m, nd, K):
    """(shapes five to eight: a forward matrix beyond the LDS / d beyond 128 -- the proposal's factor and the forward matrix
    both through the shared double buffer, one after the other; the last two: beyond 256 parameters, the wide path)"""
    rng = np.random.default_rng(600 + d + m)
    g = dict(A=rng.normal(size=(m, d)) / np.sqrt(d), b=rng.normal(size=m), data=rng.normal(size=(nd, m)),
             noise_precision=rng.uniform(0.5, 3, size=m), prior_mean=rng.normal(size=d) * 0.1,
             prior_precision=rng.uniform(0.1, 2, size=d))
    low = _posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, g, fwd_matrix=g["A"], fwd_shift=g["b"])
    B = rng.normal(size=(d, d)) * 0.02 / np.sqrt(d)
    L = np.linalg.cholesky(B @ B.T + 0.002 * np.eye(d))
    x0 = rng.normal(size=(K, d))
    dev = _dense_tile_run(ffi, ctx, low, L, x0, 4713, 40, chain0=3000, splits=3)
    p = o.posterior_linear(g["A"], g["b"], g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
    e = o.Ensemble(p, np.ones(d), x0, 4713, chain0=3000, chol=L)
    out = e.advance(40)
    assert 0.02 < out["acc"].mean() < 0.98
    _same_as_oracle(dev, e, out)


@pytest.mark.parametrize("name", ["f14_mlda2_linear_d128.npz", "f14b_mlda2_linear_d192.npz", "f14c_mlda2_linear_d300.npz"])
def test_f14_two_level_d128_matrix_cores(ffi, ctx, golden, name):
    """Two-level delayed acceptance over two d = 128 (coarse matrix in LDS) / d = 192 (both streamed, one wave per SIMD)
    linear models (linear_tile_mlda2_kernel): the LIVE reference's decisions at both levels, its chains (sha256 of all
    states) and log-posteriors; == oracle."""
    import hashlib
    g = golden(name)
    n, S, d = int(g["n_states"]), int(g["S"]), g["A_fine"].shape[1]
    common = dict(data=g["data"], noise_precision=g["noise_precision"], prior_mean=g["prior_mean"],
                  prior_precision=g["prior_precision"])
    fine = _posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, common, fwd_matrix=g["A_fine"], fwd_shift=g["b_fine"])
    coarse = _posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, common, fwd_matrix=g["A_coarse"], fwd_shift=g["b_coarse"])
    dev = _device_run(ffi, ctx, fine, g["sigma"], g["x0"], int(g["seed"]), n - 1, method=ffi.METHOD_MLDA2, sur=coarse, S=S,
                      splits=3)
    assert np.array_equal(dev["acc"], g["acc"]) and np.array_equal(dev["sub_acc"], g["sub_acc"])
    full = np.concatenate([g["x0"][None], dev["traj"]])
    assert np.array_equal(full[:, :2, :], g["traj_head"])
    assert hashlib.sha256(np.ascontiguousarray(full).tobytes()).hexdigest() == str(g["traj_sha256"])
    want = g["logpost"][1:]
    assert np.abs(dev["logpost"] - want).max() <= 1e-12 * np.abs(want).max()
    of = o.posterior_linear(g["A_fine"], g["b_fine"], *common.values())
    oc = o.posterior_linear(g["A_coarse"], g["b_coarse"], *common.values())
    e = o.Ensemble(of, g["sigma"], g["x0"], int(g["seed"]), surrogate=oc, sub_chain_length=S)
    out = e.advance(n - 1, want_sub=True)
    _same_as_oracle(dev, e, out)
    assert np.array_equal(dev["sub_acc"], out["sub_acc"])


@pytest.mark.parametrize("d,mf,mc,nd,K,S", [(128, 128, 128, 5, 70, 5), (9, 3, 3, 2, 33, 2), (16, 40, 12, 1, 64, 3),
                                              (40, 70, 70, 8, 50, 4), (100, 20, 36, 4, 129, 1), (5, 30, 6, 5, 40, 6)])
def test_tile_kernel_two_level_other_shapes(ffi, ctx, d, mf, mc, nd, K, S):
    """Ragged shapes: dimensions that are not multiples of the MFMA tile, levels with different numbers of
    outputs (each with its own data), a level that alone would fit the lane kernels, a partly filled last wave."""
    rng = np.random.default_rng(500 + d + mf + mc)

    def level(m, scale):
        g = dict(data=rng.normal(size=(nd, m)), noise_precision=rng.uniform(0.5, 3, size=m), prior_mean=rng.normal(size=d) * 0.1,
                 prior_precision=rng.uniform(0.1, 2, size=d))
        A, b = scale * rng.normal(size=(m, d)) / np.sqrt(d), rng.normal(size=m)
        return (_posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, g, fwd_matrix=A, fwd_shift=b),
                o.posterior_linear(A, b, g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"]))

    (lf, pf), (lc, pc) = level(mf, 1.0), level(mc, 0.9)
    sigma = rng.uniform(0.02, 0.08, size=d)
    x0 = rng.normal(size=(K, d))
    dev = _device_run(ffi, ctx, lf, sigma, x0, 4712, 40, method=ffi.METHOD_MLDA2, sur=lc, S=S, chain0=2000, splits=3)
    e = o.Ensemble(pf, sigma, x0, 4712, chain0=2000, surrogate=pc, sub_chain_length=S)
    out = e.advance(40, want_sub=True)
    _same_as_oracle(dev, e, out)   # incl. the Welford moments replayed from the record
    assert np.array_equal(dev["sub_acc"], out["sub_acc"])


@pytest.mark.parametrize("name", ["f18_mlda_nested2_linear_d128.npz", "f18b_mlda_nested4_linear_d24.npz",
                                  "f18c_mlda_nested8_linear_d24.npz", "f18d_mlda_nested2_linear_d192.npz",
                                  "f18e_mlda_nested2_linear_d300.npz"])
def test_f18_nested_levels_matrix_cores(ffi, ctx, golden, name):
    """Hierarchies of three (d = 128), five and nine (d = 24) linear models (linear_tile_nested_kernel): the LIVE
    reference's top-level decisions, accepted counts of the lower levels, chains (sha256) and log-posteriors;
    == oracle."""
    import hashlib
    g = golden(name)
    n, d = int(g["n_states"]), g["A"].shape[2]
    common = dict(data=g["data"], noise_precision=g["noise_precision"], prior_mean=g["prior_mean"],
                  prior_precision=g["prior_precision"])
    lows = [_posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, common, fwd_matrix=A, fwd_shift=b) for A, b in zip(g["A"], g["b"])]
    subs = [int(v) for v in g["sub_lengths"]]
    dev = _nested_device_run(ffi, ctx, lows[-1], lows[:-1], subs, g["sigma"], g["x0"], int(g["seed"]), n - 1, splits=3)
    assert np.array_equal(dev["acc"], g["acc"]) and np.array_equal(dev["level_acc"], g["level_acc"])
    full = np.concatenate([g["x0"][None], dev["traj"]])
    assert np.array_equal(full[:, :2, :], g["traj_head"])
    assert hashlib.sha256(np.ascontiguousarray(full).tobytes()).hexdigest() == str(g["traj_sha256"])
    want = g["logpost"][1:]
    assert np.abs(dev["logpost"] - want).max() <= 1e-12 * np.abs(want).max()
    models = [o.posterior_linear(A, b, *common.values()) for A, b in zip(g["A"], g["b"])]
    P, Q, steps = o.nested_levels(models[:-1], subs, models[-1])
    e = o.NestedEnsemble(P, Q, steps, g["sigma"], g["x0"], int(g["seed"]))
    out = e.advance(n - 1)
    _same_as_oracle(dev, e, out)
    assert np.array_equal(dev["level_acc"], e.level_acc)


@pytest.mark.parametrize("d,ms,nd,K,E", [(128, (128, 128, 128), 5, 40, 2), (12, (5, 20, 9), 3, 33, 2),
                                         (24, (16, 40, 8, 30, 12), 2, 70, 3), (64, (64, 30, 64, 10, 48), 4, 17, 3),
                                         (9, (3, 3, 4, 2, 5, 3, 6), 1, 129, 4), (5, (30, 12, 6), 2, 20, 2),
                                         (20, (12, 9, 16, 5, 20, 7, 11, 18, 10), 2, 40, 5),
                                         (128, (16,) * 14 + (128,), 3, 20, 8), (10, (3, 9) * 5 + (12,), 1, 65, 6),
                                         (128, (256, 40, 300), 2, 36, 2), (192, (30, 64, 20, 50, 160), 3, 20, 3),
                                         (256, (16, 32, 16), 2, 17, 2)])
def test_tile_kernel_nested_levels_vs_oracle(ffi, ctx, d, ms, nd, K, E):
    """Delayed acceptance over an even number of surrogates at large dimension (ymc_tile_nested.hip): a
    hierarchy of 2 (E - 1) + 1 linear models, each with its own matrix, output count and data; ragged shapes, a
    partly filled last wave, a hierarchy one of whose models alone would fit the lane kernels; a base level whose
    matrix does not fit the LDS (256 x 128: streamed like the upper levels') and parameter dimensions up to 256.  Everything the
    lane kernel's test compares: chains, decisions of the recorded level, accepted counts of the lower levels,
    log-posteriors, Welford moments, carried state -- across three launches."""
    rng = np.random.default_rng(900 + d + sum(ms))
    assert len(ms) == 2 * (E - 1) + 1

    def model(m, scale):
        g = dict(data=rng.normal(size=(nd, m)), noise_precision=rng.uniform(0.5, 3, size=m), prior_mean=rng.normal(size=d) * 0.1,
                 prior_precision=rng.uniform(0.1, 2, size=d))
        A, b = scale * rng.normal(size=(m, d)) / np.sqrt(d), rng.normal(size=m)
        return (_posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, g, fwd_matrix=A, fwd_shift=b),
                o.posterior_linear(A, b, g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"]))

    models = [model(m, 1.0 - 0.03 * i) for i, m in enumerate(ms)]
    (lt, pt), surs = models[-1], models[:-1]
    subs = [int(v) for v in rng.integers(1, 4 if E <= 5 else 3, size=len(surs))]
    sigma = rng.uniform(0.02, 0.08, size=d)
    x0 = rng.normal(size=(K, d))
    n = 24 if E <= 4 else 9
    dev = _nested_device_run(ffi, ctx, lt, [l for l, _ in surs], subs, sigma, x0, 777, n, splits=3, chain0=3000)
    P, Q, steps = o.nested_levels([p for _, p in surs], subs, pt)
    e = o.NestedEnsemble(P, Q, steps, sigma, x0, 777, chain0=3000)
    out = e.advance(n)
    _same_as_oracle(dev, e, out)
    assert np.array_equal(dev["level_acc"], e.level_acc)
    assert 0 < dev["acc"].mean() < 1


@pytest.mark.parametrize("d,ms,K,E", [(128, (128, 128, 128), 40, 2), (24, (16, 40, 8, 30, 12), 70, 3), (9, (3, 3, 4, 2, 5, 3, 6), 129, 4)])
def test_tile_kernel_nested_base_level_streamed_agrees(ffi, ctx, monkeypatch, d, ms, K, E):
    """YMC_TILE_NESTED=stream: level 0's matrix through the shared double buffer at shapes that would fit the LDS == the
    resident build == oracle."""
    rng = np.random.default_rng(950 + d + sum(ms))

    def model(m, scale):
        g = dict(data=rng.normal(size=(2, m)), noise_precision=rng.uniform(0.5, 3, size=m), prior_mean=rng.normal(size=d) * 0.1,
                 prior_precision=rng.uniform(0.1, 2, size=d))
        A, b = scale * rng.normal(size=(m, d)) / np.sqrt(d), rng.normal(size=m)
        return (_posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, g, fwd_matrix=A, fwd_shift=b),
                o.posterior_linear(A, b, g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"]))

    models = [model(m, 1.0 - 0.03 * i) for i, m in enumerate(ms)]
    (lt, pt), surs = models[-1], models[:-1]
    subs = [int(v) for v in rng.integers(1, 4, size=len(surs))]
    sigma, x0 = rng.uniform(0.02, 0.08, size=d), rng.normal(size=(K, d))
    P, Q, steps = o.nested_levels([p for _, p in surs], subs, pt)
    e = o.NestedEnsemble(P, Q, steps, sigma, x0, 778, chain0=10)
    out = e.advance(15)
    for mode in ("stream", "resident"):
        monkeypatch.setenv("YMC_TILE_NESTED", mode)
        dev = _nested_device_run(ffi, ctx, lt, [l for l, _ in surs], subs, sigma, x0, 778, 15, splits=2, chain0=10)
        _same_as_oracle(dev, e, out)
        assert np.array_equal(dev["level_acc"], e.level_acc)


@pytest.mark.parametrize("stage", ["0", "1"])
def test_tile_kernels_with_and_without_the_shared_matrix_stream(ffi, ctx, monkeypatch, stage):
    """The two-level and nested matrix-core kernels bring the upper level's matrices from L2 once per workgroup
    through an LDS double buffer when there is room (YMC_TILE_STAGE=1, the default) or once per wave
    (YMC_TILE_STAGE=0): both forms == oracle, at a shape with waves past the last tile in the last workgroup."""
    monkeypatch.setenv("YMC_TILE_STAGE", stage)
    rng = np.random.default_rng(4321)
    d, nd, K = 96, 3, 37

    def model(m, scale):
        g = dict(data=rng.normal(size=(nd, m)), noise_precision=rng.uniform(0.5, 3, size=m), prior_mean=rng.normal(size=d) * 0.1,
                 prior_precision=rng.uniform(0.1, 2, size=d))
        A, b = scale * rng.normal(size=(m, d)) / np.sqrt(d), rng.normal(size=m)
        return (_posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, g, fwd_matrix=A, fwd_shift=b),
                o.posterior_linear(A, b, g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"]))

    models = [model(m, 1.0 - 0.03 * i) for i, m in enumerate((64, 100, 33, 96, 80))]
    sigma, x0 = rng.uniform(0.02, 0.06, size=d), rng.normal(size=(K, d))
    (lf, pf), (lc, pc) = models[4], models[0]
    dev = _device_run(ffi, ctx, lf, sigma, x0, 4712, 24, method=ffi.METHOD_MLDA2, sur=lc, S=3, chain0=2000, splits=2)
    e = o.Ensemble(pf, sigma, x0, 4712, chain0=2000, surrogate=pc, sub_chain_length=3)
    out = e.advance(24, want_sub=True)
    _same_as_oracle(dev, e, out)
    assert np.array_equal(dev["sub_acc"], out["sub_acc"])
    subs = [2, 2, 3, 2]
    dev = _nested_device_run(ffi, ctx, models[4][0], [l for l, _ in models[:4]], subs, sigma, x0, 777, 12, splits=2, chain0=3000)
    P, Q, steps = o.nested_levels([p for _, p in models[:4]], subs, models[4][1])
    e = o.NestedEnsemble(P, Q, steps, sigma, x0, 777, chain0=3000)
    out = e.advance(12)
    _same_as_oracle(dev, e, out)
    assert np.array_equal(dev["level_acc"], e.level_acc)
    # the random walk with a dense proposal covariance: its factor L through the same stream
    Bm = rng.normal(size=(d, d)) * 0.02 / np.sqrt(d)
    L = np.linalg.cholesky(Bm @ Bm.T + 0.002 * np.eye(d))
    dev = _dense_tile_run(ffi, ctx, models[4][0], L, x0, 4713, 30, chain0=3000, splits=3)
    e = o.Ensemble(models[4][1], np.ones(d), x0, 4713, chain0=3000, chol=L)
    _same_as_oracle(dev, e, e.advance(30))
    # the adaptive error model: both levels with the same outputs and data; units of KS/4 k-steps here, of KS/8
    # at d = m = 128 (test_f16, where the data rows leave 11 KB)
    m = 40
    A, b = rng.normal(size=(m, d)) / np.sqrt(d), rng.normal(size=m) * 0.3
    As, bs = A + 0.05 * rng.normal(size=(m, d)) / np.sqrt(d), b + 0.1 * rng.normal(size=m)
    data = np.array([A @ rng.normal(size=d) + b + 0.5 * rng.normal(size=m) for _ in range(nd)])
    common = dict(data=data, noise_precision=np.full(m, 4.0), prior_mean=np.zeros(d), prior_precision=np.full(d, 0.25))
    dev = _aem_device_run(ffi, ctx, None, dict(common, A=A, b=b), dict(common, A=As, b=bs), sigma, x0, 31337, 60, 3, 4, 1,
                          splits=2, chain0=77)
    e = o.AemEnsemble(o.posterior_linear(A, b, **common), o.posterior_linear(As, bs, **common), sigma, x0, 31337, 3, 4, 1,
                      chain0=77)
    out = e.advance(60, want_sub=True)
    _same_as_oracle(dev, e, out)
    assert np.array_equal(dev["sub_acc"], out["sub_acc"])
    _same_error_model(dev, e)


@pytest.mark.parametrize("mode", ["mem", "regs", "lds", "plain", "stream"])
@pytest.mark.parametrize("d,mc,mf,K,S", [(128, 128, 128, 150, 5), (128, 96, 112, 37, 2), (96, 64, 80, 200, 3), (20, 24, 17, 130, 4),
                                          (7, 5, 9, 300, 1)])
def test_tile_mlda2_builds_agree_with_the_oracle(ffi, ctx, monkeypatch, mode, d, mc, mf, K, S):
    """The three builds of the two-level matrix-core kernel (ymc_tile.hip: fine state in LDS with both matrices
    streamed / in registers with the coarse matrix resident / the latter with per-wave L2 reads of the fine
    matrix), forced through YMC_TILE_MLDA2: each == oracle -- chains, decisions of both levels, log-posteriors,
    carried state -- at the d = 128 shape of C3's two-level workload, at ragged shapes, with several workgroups
    and waves past the last tile, across launches."""
    monkeypatch.setenv("YMC_TILE_MLDA2", mode)
    rng = np.random.default_rng(77 + d + mc + K)
    nd = 3

    def model(m, scale):
        g = dict(data=rng.normal(size=(nd, m)), noise_precision=rng.uniform(0.5, 3, size=m), prior_mean=rng.normal(size=d) * 0.1,
                 prior_precision=rng.uniform(0.1, 2, size=d))
        A, b = scale * rng.normal(size=(m, d)) / np.sqrt(d), rng.normal(size=m)
        return (_posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, g, fwd_matrix=A, fwd_shift=b),
                o.posterior_linear(A, b, g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"]))

    (lf, pf), (lc, pc) = model(mf, 1.0), model(mc, 0.95)
    sigma, x0 = rng.uniform(0.02, 0.06, size=d), rng.normal(size=(K, d))
    dev = _device_run(ffi, ctx, lf, sigma, x0, 991, 18, method=ffi.METHOD_MLDA2, sur=lc, S=S, chain0=5000, splits=3)
    e = o.Ensemble(pf, sigma, x0, 991, chain0=5000, surrogate=pc, sub_chain_length=S)
    out = e.advance(18, want_sub=True)
    _same_as_oracle(dev, e, out)
    assert np.array_equal(dev["sub_acc"], out["sub_acc"])
    assert 0 < dev["acc"].mean() < 1


@pytest.mark.parametrize("d,mc,mf,K,S,dense", [(192, 40, 160, 40, 3, False), (128, 256, 64, 70, 2, False), (256, 16, 48, 33, 4, False),
                                                (130, 130, 130, 17, 2, False), (128, 128, 96, 36, 3, True), (144, 20, 30, 20, 2, True)])
def test_tile_mlda2_beyond_the_lds_and_d128_vs_oracle(ffi, ctx, d, mc, mf, K, S, dense):
    """Two-level delayed acceptance where the coarse matrix does not fit the LDS (256 outputs x 128; a dense prior's d more
    outputs per level) and at parameter dimensions up to 256 (one wave per SIMD, both matrices streamed through the shared
    double buffer, the fine state in memory): chains, both levels' decisions, log-posteriors, moments == oracle."""
    rng = np.random.default_rng(500 + d + mc)
    nd = 2

    def model(m, scale):
        g = dict(data=rng.normal(size=(nd, m)), noise_precision=rng.uniform(0.5, 3, size=m), prior_mean=rng.normal(size=d) * 0.1,
                 prior_precision=rng.uniform(0.1, 2, size=d))
        A, b = scale * rng.normal(size=(m, d)) / np.sqrt(d), rng.normal(size=m)
        if not dense:
            return (_posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, g, fwd_matrix=A, fwd_shift=b),
                    o.posterior_linear(A, b, g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"]))
        B = rng.normal(size=(d, d)) / np.sqrt(d)
        pc = B @ B.T + np.eye(d)
        g["prior_precision"] = np.ones(d)
        return (_dense_low(ffi, ffi.POSTERIOR_LINEAR, d, g, None, pc, fwd_matrix=A, fwd_shift=b),
                o.dense_covariances(o.posterior_linear(A, b, g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"]),
                                    prior_cov=pc))

    (lf, pf), (lc, pc_) = model(mf, 1.0), model(mc, 0.95)
    sigma, x0 = rng.uniform(0.02, 0.05, size=d), rng.normal(size=(K, d)) * 0.5
    dev = _device_run(ffi, ctx, lf, sigma, x0, 993, 14, method=ffi.METHOD_MLDA2, sur=lc, S=S, chain0=600, splits=3)
    e = o.Ensemble(pf, sigma, x0, 993, chain0=600, surrogate=pc_, sub_chain_length=S)
    out = e.advance(14, want_sub=True)
    _same_as_oracle(dev, e, out)
    assert np.array_equal(dev["sub_acc"], out["sub_acc"])
    assert 0 < dev["sub_acc"].mean() < 1


@pytest.mark.parametrize("d,m,nd,K,method,stride", [(300, 40, 2, 70, "mrw", 1), (513, 700, 3, 20, "pcn", 1), (1000, 64, 2, 17, "mrw", 4),
                                                    (257, 1, 1, 33, "mrw", 1), (320, 64, 2, 8229, "mrw", 1), (592, 48, 2, 8200, "pcn", 3),
                                                    (512, 520, 1, 8192, "mrw", 2), (1024, 256, 2, 8192, "mrw", 5), (1100, 30, 1, 40, "pcn", 1),
                                                    (544, 100, 2, 8210, "beyond-L2", 1)])
def test_wide_fused_kernel_vs_oracle_and_the_launch_per_stage_path(ffi, ctx, monkeypatch, d, m, nd, K, method, stride):
    """linear_wide_fused_kernel (round 4): the random walk and pCN beyond 256 parameters as ONE persistent launch per advance
    call -- the state in registers cut along k, the proposal in LDS, every wave its output tiles over all k, the sums handed
    from wave to wave behind tickets.  Small ensembles (sixteen waves a tile), large ones (eight waves, two tiles; one tile
    beyond 576 parameters; sixteen waves, also the build for matrices no L2 holds, forced on a small one),
    ragged tails, several rounds of output tiles, thinned records, launches cut in two: == oracle, and == the
    launch-per-stage path (YMC_WIDE_FUSED=0) bit for bit."""
    from yagremcmc_amd import lowering
    if method == "beyond-L2":
        monkeypatch.setenv("YMC_WIDE_W", "16")
        method = "mrw"
    rng = np.random.default_rng(4000 + d + m)
    g = dict(data=rng.normal(size=(nd, m)), noise_precision=rng.uniform(0.5, 3, size=m), prior_mean=rng.normal(size=d) * 0.1,
             prior_precision=rng.uniform(0.1, 2, size=d))
    A, b = rng.normal(size=(m, d)) / np.sqrt(d), rng.normal(size=m)
    x0 = rng.normal(size=(K, d)) * 0.3
    n = 10
    # FullDiagnostics' moments where the record holds every transition (the fused kernel's are replayed from it)
    flags = ffi.RECORD_TRAJECTORY | ffi.RECORD_LOGPOST | ((ffi.RECORD_ACCEPT | ffi.RECORD_MOMENTS) if stride == 1 else 0)
    if method == "pcn":
        g["prior_mean"] = np.zeros(d)
        low = _posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, g, fwd_matrix=A, fwd_shift=b)
        low.desc.likelihood_only = 1
        p = dict(o.posterior_linear(A, b, g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"]), likelihood_only=1)
        sigma = 1.0 / np.sqrt(g["prior_precision"])
        make = lambda: lowering.make_problem(ctx, ffi.METHOD_PCN, low, sigma, pcn_step_size=0.01)   # noqa: E731
        e = o.Ensemble(p, sigma, x0, 17, chain0=40, pcn_step_size=0.01)
    else:
        low = _posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, g, fwd_matrix=A, fwd_shift=b)
        p = o.posterior_linear(A, b, g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
        sigma = rng.uniform(0.01, 0.04, size=d)
        make = lambda: lowering.make_problem(ctx, ffi.METHOD_MRW, low, sigma)   # noqa: E731
        e = o.Ensemble(p, sigma, x0, 17, chain0=40)

    def device(fused, split=False):
        monkeypatch.setenv("YMC_WIDE_FUSED", "1" if fused else "0")
        monkeypatch.setenv("YMC_WIDE_SPLIT", "1" if split else "0")
        run = ffi.Run(make(), 17, K, x0, n, flags, 40, record_stride=stride)
        run.advance(3)
        run.advance(n - 3)
        run.sync()
        ms, launches = run.kernel_time_ms()
        s_ = run.summary()
        return dict(traj=run.trajectory(), lp=run.logpost(), acc=run.accept() if stride == 1 else None, state=run.state(),
                    accepted=s_["accepted"], status=s_["status"], launches=launches, wf=(s_["wf_mean"], s_["wf_m2"]))
    # small ensembles also as a tile over a group of workgroups (linear_wide_split_kernel; the same kernel where no group forms)
    runs = [device(True), device(False)] + ([device(True, split=True)] if K <= 2048 else [])
    out = e.advance(n)
    rows = slice(stride - 1, None, stride)
    for dev in runs:
        assert np.array_equal(dev["traj"], out["traj"][rows]) and np.array_equal(dev["lp"], out["logpost"][rows])
        if stride == 1:
            assert np.array_equal(dev["acc"], out["acc"])
        assert np.array_equal(dev["state"][0], e.x) and np.array_equal(dev["state"][1], e.lp)
        assert np.array_equal(dev["accepted"], e.acc_count) and np.array_equal(dev["status"], e.status)
        if stride == 1:
            assert np.array_equal(dev["wf"][0], e.wf_mean) and np.array_equal(dev["wf"][1], e.wf_m2)
    assert 0 < out["acc"].mean() < 1


@pytest.mark.parametrize("d,m,K,env", [(512, 512, 256, {}), (512, 512, 256, {"YMC_WIDE_SPLIT_KERNEL": "stream"}), (512, 520, 16, {}),
                                       (512, 520, 16, {"YMC_WIDE_SPLIT_KERNEL": "stream", "YMC_WIDE_SPLIT_W": "8"}),
                                       (320, 200, 300, {"YMC_WIDE_SPLIT_G": "5"}), (600, 90, 2048, {}), (264, 16, 40, {"YMC_WIDE_SPLIT": "f"}),
                                       (640, 2000, 20, {}), (300, 40, 1000, {"YMC_WIDE_SPLIT_KERNEL": "stream"}), (600, 700, 20, {}), (600, 460, 30, {}),
                                       (700, 330, 100, {"YMC_WIDE_SPLIT_G": "3", "YMC_WIDE_SPLIT_W": "8"}), (1100, 60, 50, {})])
def test_wide_split_kernel_vs_oracle(ffi, ctx, monkeypatch, d, m, K, env):
    """linear_wide_split_kernel (round 4): small ensembles beyond 256 parameters with every tile of 16 chains over a GROUP of
    workgroups -- each multiplies its share of the output tiles, the terms cross the chip once per transition (write-through
    stores, one arrival counter per tile, one acquire per workgroup), every member adds them in the canonical order and
    decides alike -- and linear_wide_systolic_kernel (up to 640 parameters, one or two output tiles per member): the member's
    rows of the matrix in registers cut along k, an output tile's accumulator travelling down the waves, the decision taken
    by the group's leader and published in one tagged word.  Groups of 32, 16, 5, 3, 2 and 1 (the protocol alone), output
    tiles that do not divide among the members, the leader's sum in several chunks (m = 2000), every build, thinned record
    with the launch cut in three, 40 transitions (the slots alternate): == oracle bit for bit."""
    from yagremcmc_amd import lowering
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    rng = np.random.default_rng(4100 + d + m)
    nd = 2
    g = dict(data=rng.normal(size=(nd, m)), noise_precision=rng.uniform(0.5, 3, size=m), prior_mean=rng.normal(size=d) * 0.1,
             prior_precision=rng.uniform(0.1, 2, size=d))
    A, b = rng.normal(size=(m, d)) / np.sqrt(d), rng.normal(size=m)
    x0 = rng.normal(size=(K, d)) * 0.3
    low = _posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, g, fwd_matrix=A, fwd_shift=b)
    p = o.posterior_linear(A, b, g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
    sigma = rng.uniform(0.01, 0.04, size=d)
    n, stride = 40, 2
    flags = ffi.RECORD_TRAJECTORY | ffi.RECORD_LOGPOST
    run = ffi.Run(lowering.make_problem(ctx, ffi.METHOD_MRW, low, sigma), 23, K, x0, n // stride, flags, 7, record_stride=stride)
    for part in (5, 14, 21):
        run.advance(part)
    run.sync()
    e = o.Ensemble(p, sigma, x0, 23, chain0=7)
    out = e.advance(n)
    rows = slice(stride - 1, None, stride)
    assert np.array_equal(run.trajectory(), out["traj"][rows]) and np.array_equal(run.logpost(), out["logpost"][rows])
    s_ = run.summary()
    assert np.array_equal(run.state()[0], e.x) and np.array_equal(run.state()[1], e.lp)
    assert np.array_equal(s_["accepted"], e.acc_count) and np.array_equal(s_["status"], e.status)
    assert 0 < out["acc"].mean() < 1
    # the record is tiled on the device (512-byte stores): the egress path untiles it like a fetch
    run.egress_begin(1, 3, 11, ffi.RECORD_TRAJECTORY | ffi.RECORD_LOGPOST)
    v = run.egress_wait(1)
    assert np.array_equal(v["trajectory"], out["traj"][rows][3:14]) and np.array_equal(v["logpost"], out["logpost"][rows][3:14])


@pytest.mark.parametrize("d,m,K,stride", [(300, 20, 30, 1), (512, 128, 4100, 1), (417, 300, 50, 3), (590, 40, 17, 1), (264, 1, 16, 2)])
def test_wide_fused_dense_proposal_vs_oracle_and_the_launch_per_stage_path(ffi, ctx, monkeypatch, d, m, K, stride):
    """The random walk with a dense proposal covariance beyond 256 parameters as ONE persistent launch (linear_wide_fused_kernel's
    dense build: the normals in a second LDS tile, L xi one more product on the matrix cores, written straight into the
    proposal tile's B layout): == oracle and == the launch-per-stage path (YMC_WIDE_FUSED=0) bit for bit; moments replayed
    from the record, thinned records, launches cut in two."""
    from yagremcmc_amd import lowering
    rng = np.random.default_rng(4300 + d + m)
    g = dict(data=rng.normal(size=(2, m)), noise_precision=rng.uniform(0.5, 3, size=m), prior_mean=rng.normal(size=d) * 0.1,
             prior_precision=rng.uniform(0.1, 2, size=d))
    A, b = rng.normal(size=(m, d)) / np.sqrt(d), rng.normal(size=m)
    low = _posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, g, fwd_matrix=A, fwd_shift=b)
    p = o.posterior_linear(A, b, g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
    B = rng.normal(size=(d, d)) * 0.02 / np.sqrt(d)
    L = np.linalg.cholesky(B @ B.T + 0.0005 * np.eye(d))
    x0 = rng.normal(size=(K, d)) * 0.3
    n = 9
    flags = ffi.RECORD_TRAJECTORY | ffi.RECORD_LOGPOST | ((ffi.RECORD_ACCEPT | ffi.RECORD_MOMENTS) if stride == 1 else 0)

    def device(fused):
        monkeypatch.setenv("YMC_WIDE_FUSED", "1" if fused else "0")
        run = ffi.Run(lowering.make_problem(ctx, ffi.METHOD_MRW, low, np.ones(d), chol=np.tril(L)), 31, K, x0, n, flags, 5, record_stride=stride)
        run.advance(4)
        run.advance(n - 4)
        run.sync()
        s_ = run.summary()
        return dict(traj=run.trajectory(), lp=run.logpost(), acc=run.accept() if stride == 1 else None, state=run.state(),
                    accepted=s_["accepted"], wf=(s_["wf_mean"], s_["wf_m2"]))
    fused, staged = device(True), device(False)
    e = o.Ensemble(p, np.ones(d), x0, 31, chain0=5, chol=L)
    out = e.advance(n)
    rows = slice(stride - 1, None, stride)
    for dev in (fused, staged):
        assert np.array_equal(dev["traj"], out["traj"][rows]) and np.array_equal(dev["lp"], out["logpost"][rows])
        assert np.array_equal(dev["state"][0], e.x) and np.array_equal(dev["state"][1], e.lp)
        assert np.array_equal(dev["accepted"], e.acc_count)
        if stride == 1:
            assert np.array_equal(dev["acc"], out["acc"])
            assert np.array_equal(dev["wf"][0], e.wf_mean) and np.array_equal(dev["wf"][1], e.wf_m2)
    assert 0 < out["acc"].mean() < 1


@pytest.mark.parametrize("kernel", ["systolic", "stream"])
def test_group_kernels_beside_another_launch(ffi, ctx, monkeypatch, kernel):
    """The group kernels' exchange under UNEVEN load: a second context (its own stream) keeps the chip busy with the d = 128
    matrix-core sampler while the group kernel runs -- its workgroups become resident at different times, every member's L1
    is warm with the slot's previous contents (the slots alternate for 60 transitions) -- and the chains still equal the
    oracle's bit for bit; then the other way round (the group kernel first)."""
    from yagremcmc_amd import lowering
    if kernel == "stream":
        monkeypatch.setenv("YMC_WIDE_SPLIT_KERNEL", "stream")
    rng = np.random.default_rng(91)

    def linear(d, m):
        g = dict(data=rng.normal(size=(2, m)), noise_precision=rng.uniform(0.5, 3, size=m), prior_mean=rng.normal(size=d) * 0.1,
                 prior_precision=rng.uniform(0.1, 2, size=d))
        A, b = rng.normal(size=(m, d)) / np.sqrt(d), rng.normal(size=m)
        return (_posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, g, fwd_matrix=A, fwd_shift=b),
                o.posterior_linear(A, b, g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"]))
    other = ffi.Context(ctx.device)
    low_b, _ = linear(128, 128)
    Kb = 65536
    busy = ffi.Run(lowering.make_problem(other, ffi.METHOD_MRW, low_b, np.full(128, 0.03)), 3, Kb, rng.normal(size=(Kb, 128)) * 0.1, 1, 0, 0)
    d, m, K, n = 400, 300, 128, 60
    low, p = linear(d, m)
    sigma, x0 = rng.uniform(0.01, 0.03, size=d), rng.normal(size=(K, d)) * 0.3
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, sigma)
    e = o.Ensemble(p, sigma, x0, 8, chain0=3)
    out = e.advance(n)
    for group_first in (False, True):
        run = ffi.Run(prob, 8, K, x0, n, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT | ffi.RECORD_LOGPOST, 3)
        if group_first:
            run.advance(n)
        for _ in range(6):
            busy.advance(40)      # asynchronous: ~3 ms each on the other stream
        if not group_first:
            run.advance(n)
        run.sync()
        busy.sync()
        assert np.array_equal(run.trajectory(), out["traj"]) and np.array_equal(run.accept(), out["acc"])
        assert np.array_equal(run.logpost(), out["logpost"]) and not run.summary()["status"].any()


@pytest.mark.parametrize("kernel", ["systolic", "stream"])
def test_an_abandoned_group_exchange_fails_the_run(ffi, ctx, monkeypatch, kernel):
    """The group kernels wait for other workgroups inside a launch (arrival counters, the leader's decision word): a wait
    that does not end within the poll limit gives up -- every wave still reaches the end of the launch -- and the run fails
    with YMC_ERR_DEVICE instead of delivering a record made from terms that never arrived (the limit is lowered through the
    test-only knob so that the give-up path really runs); a fresh run with the product's limit is fine."""
    from yagremcmc_amd import lowering
    if os.environ.get("YMC_WIDE_SPLIT") == "0" or os.environ.get("YMC_WIDE_FUSED") == "0":
        pytest.skip("the group kernels are switched off by the override under test (tools/env_sweep.sh)")
    if kernel == "stream":
        monkeypatch.setenv("YMC_WIDE_SPLIT_KERNEL", "stream")
    d, m, K = 400, 300, 64
    rng = np.random.default_rng(77)
    g = dict(data=rng.normal(size=(2, m)), noise_precision=rng.uniform(0.5, 3, size=m), prior_mean=rng.normal(size=d) * 0.1,
             prior_precision=rng.uniform(0.1, 2, size=d))
    A, b = rng.normal(size=(m, d)) / np.sqrt(d), rng.normal(size=m)
    low = _posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, g, fwd_matrix=A, fwd_shift=b)
    sigma, x0 = np.full(d, 0.02), rng.normal(size=(K, d)) * 0.3
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, sigma)
    monkeypatch.setenv("YMC_LV_POLL_LIMIT", "0")
    run = ffi.Run(prob, 5, K, x0, 30, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT, 0)
    run.advance(30)
    with pytest.raises(ffi.YmcError, match="internal device error") as err:
        run.sync()
    assert err.value.code == ffi.ERR_DEVICE
    with pytest.raises(ffi.YmcError, match="internal device error"):
        run.trajectory()
    monkeypatch.delenv("YMC_LV_POLL_LIMIT")
    good = ffi.Run(prob, 5, K, x0, 30, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT, 0)
    good.advance(30)
    good.sync()
    p = o.posterior_linear(A, b, g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
    e = o.Ensemble(p, sigma, x0, 5, chain0=0)
    out = e.advance(30)
    assert np.array_equal(good.trajectory(), out["traj"]) and not good.summary()["status"].any()


@pytest.mark.parametrize("d,m,nd,K,method,stride", [(300, 40, 2, 70, "mrw", 1), (257, 1, 1, 33, "mrw", 1), (513, 700, 3, 20, "pcn", 1),
                                                    (1000, 64, 2, 17, "mrw", 4), (384, 384, 2, 40, "pcn", 3), (300, 20, 2, 30, "dense", 1)])
def test_linear_posteriors_beyond_256_parameters_vs_oracle(ffi, ctx, d, m, nd, K, method, stride):
    """Beyond the matrix-core kernels' 256 parameters the random walk (diagonal proposal) and pCN run as a sequence of plain
    kernels per transition (launch_advance_wide): chains, decisions, log-posteriors, moments, carried state == oracle at any
    d and m; thinned records; dense noise and prior covariances folded in as at smaller sizes; the IAT of such records."""
    from yagremcmc_amd import lowering
    rng = np.random.default_rng(3000 + d + m)
    g = dict(data=rng.normal(size=(nd, m)), noise_precision=rng.uniform(0.5, 3, size=m), prior_mean=rng.normal(size=d) * 0.1,
             prior_precision=rng.uniform(0.1, 2, size=d))
    A, b = rng.normal(size=(m, d)) / np.sqrt(d), rng.normal(size=m)
    p = o.posterior_linear(A, b, g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
    if method == "dense":
        B = rng.normal(size=(m, m)) / np.sqrt(m)
        nc = B @ B.T + np.eye(m)
        Bp = rng.normal(size=(d, d)) / np.sqrt(d)
        pc = Bp @ Bp.T + np.eye(d)
        g["noise_precision"], g["prior_precision"] = np.ones(m), np.ones(d)
        low = _dense_low(ffi, ffi.POSTERIOR_LINEAR, d, g, nc, pc, fwd_matrix=A, fwd_shift=b)
        p = o.dense_covariances(o.posterior_linear(A, b, g["data"], np.ones(m), g["prior_mean"], np.ones(d)), noise_cov=nc, prior_cov=pc)
    else:
        low = _posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, g, fwd_matrix=A, fwd_shift=b)
    x0 = rng.normal(size=(K, d)) * 0.3
    n = 24
    flags = ffi.RECORD_TRAJECTORY | ffi.RECORD_LOGPOST | ffi.RECORD_MOMENTS | (ffi.RECORD_ACCEPT if stride == 1 else 0)
    if method == "pcn":
        g["prior_mean"] = np.zeros(d)
        low = _posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, g, fwd_matrix=A, fwd_shift=b)
        low.desc.likelihood_only = 1
        p = dict(o.posterior_linear(A, b, g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"]), likelihood_only=1)
        sigma = 1.0 / np.sqrt(g["prior_precision"])
        prob = lowering.make_problem(ctx, ffi.METHOD_PCN, low, sigma, pcn_step_size=0.01)
        e = o.Ensemble(p, sigma, x0, 17, chain0=40, pcn_step_size=0.01)
    else:
        sigma = rng.uniform(0.01, 0.04, size=d)
        prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, sigma)
        e = o.Ensemble(p, sigma, x0, 17, chain0=40)
    run = ffi.Run(prob, 17, K, x0, n, flags, 40, record_stride=stride)
    run.advance(7)
    run.advance(n - 7)
    out = e.advance(n)
    rows = slice(stride - 1, None, stride)
    assert np.array_equal(run.trajectory(), out["traj"][rows]) and np.array_equal(run.logpost(), out["logpost"][rows])
    if stride == 1:
        assert np.array_equal(run.accept(), out["acc"]) and 0 < out["acc"].mean() < 1
    x, lp = run.state()
    assert np.array_equal(x, e.x) and np.array_equal(lp, e.lp)
    s_ = run.summary()
    assert np.array_equal(s_["accepted"], e.acc_count) and np.array_equal(s_["status"], e.status)
    assert np.array_equal(s_["wf_mean"], e.wf_mean) and np.array_equal(s_["wf_m2"], e.wf_m2)
    probe = rng.normal(size=(9, d))
    assert np.array_equal(prob.logdensity(probe)[0], o.logdensity(p, probe)[0])
    assert np.array_equal(prob.forward(probe), o.forward(p, probe))


@pytest.mark.parametrize("d,mc,mf,K,S,dense", [(300, 20, 40, 40, 3, False), (513, 64, 700, 17, 2, False), (260, 30, 30, 70, 4, True)])
def test_two_level_beyond_256_parameters_vs_oracle(ffi, ctx, d, mc, mf, K, S, dense):
    """Two-level delayed acceptance beyond 256 parameters (the wide path: every sub-step and the fine decision as plain
    kernels, both levels evaluated on the matrix cores from global memory): chains, both levels' decisions, log-posteriors,
    moments, carried state == oracle."""
    from yagremcmc_amd import lowering
    rng = np.random.default_rng(3100 + d + mc)

    def model(m, scale):
        g = dict(data=rng.normal(size=(2, m)), noise_precision=rng.uniform(0.5, 3, size=m), prior_mean=rng.normal(size=d) * 0.1,
                 prior_precision=rng.uniform(0.1, 2, size=d))
        A, b = scale * rng.normal(size=(m, d)) / np.sqrt(d), rng.normal(size=m)
        if not dense:
            return (_posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, g, fwd_matrix=A, fwd_shift=b),
                    o.posterior_linear(A, b, g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"]))
        B = rng.normal(size=(d, d)) / np.sqrt(d)
        pc = B @ B.T + np.eye(d)
        g["prior_precision"] = np.ones(d)
        return (_dense_low(ffi, ffi.POSTERIOR_LINEAR, d, g, None, pc, fwd_matrix=A, fwd_shift=b),
                o.dense_covariances(o.posterior_linear(A, b, g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"]),
                                    prior_cov=pc))

    (lf, pf), (lc, pc_) = model(mf, 1.0), model(mc, 0.95)
    sigma, x0 = rng.uniform(0.01, 0.03, size=d), rng.normal(size=(K, d)) * 0.3
    dev = _device_run(ffi, ctx, lf, sigma, x0, 995, 12, method=ffi.METHOD_MLDA2, sur=lc, S=S, chain0=700, splits=3)
    e = o.Ensemble(pf, sigma, x0, 995, chain0=700, surrogate=pc_, sub_chain_length=S)
    out = e.advance(12, want_sub=True)
    _same_as_oracle(dev, e, out)
    assert np.array_equal(dev["sub_acc"], out["sub_acc"])
    assert 0 < dev["sub_acc"].mean() < 1 and 0 < dev["acc"].mean()


@pytest.mark.parametrize("d,mc,mf,K,S,stride", [(320, 32, 64, 4100, 3, 1), (600, 16, 48, 4096, 2, 2), (300, 40, 20, 50, 4, 1),
                                                  (1040, 16, 32, 20, 2, 1)])
def test_wide_fused_two_level_kernel_vs_oracle_and_the_launch_per_stage_path(ffi, ctx, monkeypatch, d, mc, mf, K, S, stride):
    """linear_wide_fused2_kernel (round 4): two-level delayed acceptance beyond 256 parameters as one persistent launch --
    x and z as slices of registers, the tile in LDS evaluated under the coarse posterior per sub-step and under the fine one
    for the decision.  Large ensembles (eight waves a tile, both slice sizes), small ones (sixteen), a coarse level with more
    outputs than the fine one, thinned records: == oracle and == the launch-per-stage path (YMC_WIDE_FUSED=0)."""
    from yagremcmc_amd import lowering
    rng = np.random.default_rng(5100 + d + mc)

    def model(m, scale):
        g = dict(data=rng.normal(size=(2, m)), noise_precision=rng.uniform(0.5, 3, size=m), prior_mean=rng.normal(size=d) * 0.1,
                 prior_precision=rng.uniform(0.1, 2, size=d))
        A, b = scale * rng.normal(size=(m, d)) / np.sqrt(d), rng.normal(size=m)
        return (_posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, g, fwd_matrix=A, fwd_shift=b),
                o.posterior_linear(A, b, g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"]))
    (lf, pf), (lc, pc_) = model(mf, 1.0), model(mc, 0.95)
    sigma, x0 = rng.uniform(0.01, 0.03, size=d), rng.normal(size=(K, d)) * 0.3
    n = 8
    flags = ffi.RECORD_TRAJECTORY | ffi.RECORD_LOGPOST | ((ffi.RECORD_ACCEPT | ffi.RECORD_SUB_ACCEPT | ffi.RECORD_MOMENTS) if stride == 1 else 0)

    def device(fused):
        monkeypatch.setenv("YMC_WIDE_FUSED", "1" if fused else "0")
        prob = lowering.make_problem(ctx, ffi.METHOD_MLDA2, lf, sigma, surrogate=lc, sub_chain_length=S)
        run = ffi.Run(prob, 995, K, x0, n, flags, 700, record_stride=stride)
        run.advance(3)
        run.advance(n - 3)
        run.sync()
        s_ = run.summary()
        return dict(traj=run.trajectory(), lp=run.logpost(), acc=run.accept() if stride == 1 else None,
                    sub=run.sub_accept() if stride == 1 else None, state=run.state(), accepted=s_["accepted"], status=s_["status"],
                    wf=(s_["wf_mean"], s_["wf_m2"]))
    fused, staged = device(True), device(False)
    e = o.Ensemble(pf, sigma, x0, 995, chain0=700, surrogate=pc_, sub_chain_length=S)
    out = e.advance(n, want_sub=True)
    rows = slice(stride - 1, None, stride)
    for dev in (fused, staged):
        assert np.array_equal(dev["traj"], out["traj"][rows]) and np.array_equal(dev["lp"], out["logpost"][rows])
        assert np.array_equal(dev["state"][0], e.x) and np.array_equal(dev["state"][1], e.lp)
        assert np.array_equal(dev["accepted"], e.acc_count) and np.array_equal(dev["status"], e.status)
        if stride == 1:
            assert np.array_equal(dev["acc"], out["acc"]) and np.array_equal(dev["sub"], out["sub_acc"])
            assert np.array_equal(dev["wf"][0], e.wf_mean) and np.array_equal(dev["wf"][1], e.wf_m2)
    assert 0 < out["sub_acc"].mean() < 1


@pytest.mark.parametrize("d,ms,K,E", [(300, (20, 30, 40), 33, 2), (260, (16, 8, 24, 12, 40), 20, 3), (400, (8,) * 8 + (16,), 17, 5)])
def test_nested_levels_beyond_256_parameters_vs_oracle(ffi, ctx, d, ms, K, E):
    """The nest of levels beyond 256 parameters (the wide path: the walk as host recursion over plain kernels, every level
    evaluated on the matrix cores from global memory): chains, top-level decisions, per-level accept counts, log-posteriors,
    moments == oracle."""
    rng = np.random.default_rng(3200 + d + sum(ms))

    def model(m, scale):
        g = dict(data=rng.normal(size=(2, m)), noise_precision=rng.uniform(0.5, 3, size=m), prior_mean=rng.normal(size=d) * 0.1,
                 prior_precision=rng.uniform(0.1, 2, size=d))
        A, b = scale * rng.normal(size=(m, d)) / np.sqrt(d), rng.normal(size=m)
        return (_posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, g, fwd_matrix=A, fwd_shift=b),
                o.posterior_linear(A, b, g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"]))

    models = [model(m, 1.0 - 0.02 * i) for i, m in enumerate(ms)]
    (lt, pt), surs = models[-1], models[:-1]
    subs = [int(v) for v in rng.integers(1, 3, size=len(surs))]
    sigma, x0 = rng.uniform(0.01, 0.03, size=d), rng.normal(size=(K, d)) * 0.3
    dev = _nested_device_run(ffi, ctx, lt, [l for l, _ in surs], subs, sigma, x0, 779, 8, splits=2, chain0=60)
    P, Q, steps = o.nested_levels([q for _, q in surs], subs, pt)
    e = o.NestedEnsemble(P, Q, steps, sigma, x0, 779, chain0=60)
    out = e.advance(8)
    _same_as_oracle(dev, e, out)
    assert np.array_equal(dev["level_acc"], e.level_acc)


def test_tile_kernel_limits(ffi, ctx):
    from yagremcmc_amd import lowering
    rng = np.random.default_rng(1)
    # any number of data rows: the likelihood goes through the row mean (ymc_linear_data_summary)
    g = dict(data=rng.normal(size=(37, 16)), noise_precision=rng.uniform(0.5, 2, 16), prior_mean=np.zeros(16),
             prior_precision=np.ones(16))
    A = rng.normal(size=(16, 16)) / 4
    low = _posterior_low(ffi, ffi.POSTERIOR_LINEAR, 16, g, fwd_matrix=A, fwd_shift=np.zeros(16))
    x0 = rng.normal(size=(40, 16))
    dev = _device_run(ffi, ctx, low, np.full(16, 0.05), x0, 5, 30, moments=False)
    e = o.Ensemble(o.posterior_linear(A, np.zeros(16), g["data"], g["noise_precision"], g["prior_mean"],
                                      g["prior_precision"]), np.full(16, 0.05), x0, 5)
    out = e.advance(30)
    assert np.array_equal(dev["traj"], out["traj"]) and np.array_equal(dev["logpost"], out["logpost"])
    # parameter dimensions: 256 for the single-level samplers (matrix streamed, one wave per SIMD), 128 for the others
    def lin(d, m):
        gg = dict(data=rng.normal(size=(2, m)), noise_precision=np.ones(m), prior_mean=np.zeros(d), prior_precision=np.ones(d))
        return _posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, gg, fwd_matrix=rng.normal(size=(m, d)), fwd_shift=np.zeros(m))
    lowering.make_problem(ctx, ffi.METHOD_MRW, lin(256, 4), np.ones(256))
    lowering.make_problem(ctx, ffi.METHOD_MRW, lin(257, 4), np.ones(257))     # beyond 256: the wide path (random walk, pCN)
    lowering.make_problem(ctx, ffi.METHOD_MLDA2, lin(256, 4), np.ones(256), surrogate=lin(256, 4), sub_chain_length=2)
    lowering.make_problem(ctx, ffi.METHOD_MLDA2, lin(260, 4), np.ones(260), surrogate=lin(260, 4), sub_chain_length=2)  # the wide path
    lowering.make_problem(ctx, ffi.METHOD_MLDA2, lin(128, 400), np.ones(128), surrogate=lin(128, 400), sub_chain_length=2)
    with pytest.raises(ffi.YmcError, match="exceeds the matrix-core kernels' 128") as err:   # adaptive error model: d <= 128
        lowering.make_problem(ctx, ffi.METHOD_AEM2, lin(200, 4), np.ones(200), surrogate=lin(200, 4), sub_chain_length=2, aem=(5, 0))
    assert err.value.code == ffi.ERR_UNSUPPORTED
    with pytest.raises(ffi.YmcError, match="not fit the 160 KB LDS") as err:                  # ... and an LDS-resident coarse matrix
        lowering.make_problem(ctx, ffi.METHOD_AEM2, lin(128, 400), np.ones(128), surrogate=lin(128, 400), sub_chain_length=2,
                              aem=(5, 0))
    assert err.value.code == ffi.ERR_UNSUPPORTED
    with pytest.raises(ffi.YmcError, match="too many outputs") as err:                        # constants alone beyond the LDS
        lowering.make_problem(ctx, ffi.METHOD_MRW, lin(64, 8000), np.ones(64))
    assert err.value.code == ffi.ERR_UNSUPPORTED


# ---------------------------------------------------------------------------
# the collective: RCCL communicator with one rank (all a 1-GPU box can host)
# ---------------------------------------------------------------------------

def test_rccl_summary_gather_single_rank(ffi, ctx, golden):
    from yagremcmc_amd import distributed, lowering
    g = golden("f2_mrw_2d.npz")
    low = _gauss_low(ffi, g["mean"], g["whiten"], g["log_norm"])
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, g["sigma"])
    run = ffi.Run(prob, 77, 1000, g["x0"][:1], 1, ffi.RECORD_MOMENTS)
    run.advance(250)
    comm = ffi.Communicator(ctx, ffi.Communicator.unique_id(), 1, 0)
    comm.barrier()
    assert comm.allreduce_max(3.25) == 3.25
    packed = comm.allgather_summary(run)
    assert packed.shape == (1, 1000, run.summary_words())
    s = distributed.unpack_summaries(packed, 2)
    loc, (x, lp) = run.summary(), run.state()
    assert np.array_equal(s["state"], x) and np.array_equal(s["logpost"], lp)
    assert np.array_equal(s["accepted"], loc["accepted"]) and np.array_equal(s["wf_n"], loc["wf_n"])
    assert np.array_equal(s["wf_mean"], loc["wf_mean"]) and np.array_equal(s["wf_m2"], loc["wf_m2"])
    assert np.array_equal(s["status"], loc["status"])
    pooled = distributed.pooled_diagnostics(s, 250)
    assert 0.3 < pooled["acceptance_rate"] < 0.8 and pooled["n"] == 250 * 1000
    comm.close()


# ---------------------------------------------------------------------------
# the other BASELINE configurations at full per-GPU size: properties + oracle slices
# ---------------------------------------------------------------------------

def _slices_match_oracle(ffi, traj, acc, make_oracle, K, T, width=32):
    for c0 in (0, K // 2 - width // 2, K - width):
        out = make_oracle(c0, width).advance(T)
        assert np.array_equal(traj[:, c0:c0 + width], out["traj"])
        assert np.array_equal(acc[:, c0:c0 + width], out["acc"])


def test_c3_full_size_linear_d128(ffi, ctx):
    """262 144 chains, d = m = 128, nData = 5 (BASELINE configs[2]); 6 transitions."""
    from yagremcmc_amd import lowering
    d, K, T = 128, 262144, 6
    rng = np.random.default_rng(7)
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    g = dict(A=A, b=rng.standard_normal(d), noise_precision=np.reciprocal(np.full(d, 0.3)), prior_mean=np.zeros(d),
             prior_precision=np.reciprocal(np.full(d, 5.0)))
    theta = rng.standard_normal(d)
    g["data"] = np.array([A @ theta + g["b"] + np.sqrt(0.3) * rng.standard_normal(d) for _ in range(5)])
    low = _posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, g, fwd_matrix=A, fwd_shift=g["b"])
    sigma = np.sqrt(np.reciprocal(np.reciprocal(np.full(d, 1.1e-3))))
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, sigma)
    x0 = np.zeros((1, d))
    run = ffi.Run(prob, 20250103, K, x0, T, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT)
    run.advance(T)
    traj, acc = run.trajectory(), run.accept()
    p = o.posterior_linear(A, g["b"], g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
    _slices_match_oracle(ffi, traj, acc, lambda c0, w: o.Ensemble(p, sigma, np.zeros((w, d)), 20250103, chain0=c0), K, T)
    prev = np.concatenate([np.broadcast_to(x0, (1, K, d)), traj[:-1]])
    assert np.array_equal((traj == prev).all(axis=2), acc == 0)
    assert np.array_equal(run.summary()["accepted"], acc.sum(axis=0).astype(np.uint64))
    assert 0.2 < acc.mean() < 0.8
    # sharding invariance on a quarter of the ensemble
    part = ffi.Run(prob, 20250103, K // 4, x0, T, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT, 3 * K // 4)
    part.advance(T)
    assert np.array_equal(part.trajectory(), traj[:, 3 * K // 4:])


def test_c3_full_size_two_level_d128(ffi, ctx):
    """BASELINE configs[2] is named after example_inference_linearModel_twoLevel.py: 262 144 chains, two d = m =
    128 linear models, S = 5 on the matrix-core kernel; slices of the ensemble against the oracle, the
    delayed-acceptance invariants on all of it, sharding invariance on the last quarter."""
    from yagremcmc_amd import lowering
    d, K, T, S = 128, 262144, 3, 5
    rng = np.random.default_rng(7)
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    b = rng.standard_normal(d)
    theta = rng.standard_normal(d)
    common = dict(data=np.array([A @ theta + b + np.sqrt(0.3) * rng.standard_normal(d) for _ in range(5)]),
                  noise_precision=np.reciprocal(np.full(d, 0.3)), prior_mean=np.zeros(d),
                  prior_precision=np.reciprocal(np.full(d, 5.0)))
    A_c, b_c = A + 0.02 * rng.standard_normal((d, d)) / np.sqrt(d), b + 0.05 * rng.standard_normal(d)
    fine = _posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, common, fwd_matrix=A, fwd_shift=b)
    coarse = _posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, common, fwd_matrix=A_c, fwd_shift=b_c)
    sigma = np.sqrt(np.reciprocal(np.reciprocal(np.full(d, 1.1e-3))))
    prob = lowering.make_problem(ctx, ffi.METHOD_MLDA2, fine, sigma, surrogate=coarse, sub_chain_length=S)
    x0 = np.zeros((1, d))
    flags = ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT | ffi.RECORD_SUB_ACCEPT
    run = ffi.Run(prob, 20250103, K, x0, T, flags)
    run.advance(T)
    traj, acc, sub = run.trajectory(), run.accept(), run.sub_accept()
    of = o.posterior_linear(A, b, *common.values())
    oc = o.posterior_linear(A_c, b_c, *common.values())
    _slices_match_oracle(ffi, traj, acc, lambda c0, w: o.Ensemble(of, sigma, np.zeros((w, d)), 20250103, chain0=c0,
                                                                  surrogate=oc, sub_chain_length=S), K, T)
    assert not (acc.astype(bool) & ~sub.any(axis=1)).any()          # a fine step is accepted only if a sub-step moved
    prev = np.concatenate([np.broadcast_to(x0, (1, K, d)), traj[:-1]])
    assert np.array_equal((traj == prev).all(axis=2), acc == 0)
    assert np.array_equal(run.summary()["accepted"], acc.sum(axis=0).astype(np.uint64))
    assert 0.3 < acc.mean() < 0.98 and 0.2 < sub.mean() < 0.8
    part = ffi.Run(prob, 20250103, K // 4, x0, T, flags, 3 * K // 4)
    part.advance(T)
    assert np.array_equal(part.trajectory(), traj[:, 3 * K // 4:]) and np.array_equal(part.sub_accept(), sub[:, :, 3 * K // 4:])


def test_c5_full_size_two_level(ffi, ctx, golden):
    """131 072 chains per GPU of the two-level sampler, S = 6 (BASELINE configs[4] per-GPU share)."""
    from yagremcmc_amd import lowering
    g = golden("f5_mlda2_2d.npz")
    K, T, S = 131072, 60, 6
    fine = _gauss_low(ffi, g["fine_mean"], g["fine_whiten"], g["fine_log_norm"])
    coarse = _gauss_low(ffi, g["coarse_mean"], g["coarse_whiten"], g["coarse_log_norm"])
    prob = lowering.make_problem(ctx, ffi.METHOD_MLDA2, fine, g["sigma"], surrogate=coarse, sub_chain_length=S)
    x0 = np.array([[-8.0, -7.0]])
    flags = ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT | ffi.RECORD_SUB_ACCEPT | ffi.RECORD_MOMENTS
    run = ffi.Run(prob, 5, K, x0, T, flags)
    run.advance(T)
    traj, acc, sub = run.trajectory(), run.accept(), run.sub_accept()
    of = o.gauss_whitened(g["fine_mean"], g["fine_whiten"], float(g["fine_log_norm"]))
    oc = o.gauss_whitened(g["coarse_mean"], g["coarse_whiten"], float(g["coarse_log_norm"]))
    _slices_match_oracle(ffi, traj, acc, lambda c0, w: o.Ensemble(of, g["sigma"], np.tile(x0, (w, 1)), 5, chain0=c0,
                                                                  surrogate=oc, sub_chain_length=S), K, T)
    # a fine step can only be accepted if some coarse sub-step moved
    assert not (acc.astype(bool) & ~sub.any(axis=1)).any()
    prev = np.concatenate([np.broadcast_to(x0, (1, K, 2)), traj[:-1]])
    assert np.array_equal((traj == prev).all(axis=2), acc == 0)
    s = run.summary()
    assert np.array_equal(s["accepted"], acc.sum(axis=0).astype(np.uint64))
    assert np.allclose(s["wf_mean"], prev.mean(axis=0), rtol=0, atol=1e-12)


def test_c4_full_size_lotka_volterra(ffi, ctx, golden):
    """16 384 chains per GPU, nData = 10, T = 10, RK4 x 50 (BASELINE configs[3] per-GPU share)."""
    from yagremcmc_amd import lowering
    K, T = 16384, 12
    rng = np.random.default_rng(1111)
    design = rng.uniform(0.5, 1.5, size=(10, 2))
    base = dict(noise_precision=np.reciprocal(np.full(2, 0.04)), prior_mean=np.zeros(2),
                prior_precision=np.reciprocal(np.full(2, 1.4)))
    p0 = o.posterior_lv(design, 10.0, 0.8, 0.4, 50, np.zeros((10, 2)), **base)
    data = o.forward(p0, np.log([0.4, 0.6])).reshape(10, 2) + 0.2 * rng.standard_normal((10, 2))
    p = o.posterior_lv(design, 10.0, 0.8, 0.4, 50, data, **base)
    low = _posterior_low(ffi, ffi.POSTERIOR_LV, 2, dict(base, data=data), design=design, t_final=10.0, alpha=0.8,
                         gamma=0.4, rk4_steps=50)
    sigma = np.sqrt(np.reciprocal(np.reciprocal(np.full(2, 0.15))))
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, sigma)
    x0 = np.array([[-0.6, -0.3]])
    run = ffi.Run(prob, 9, K, x0, T, _flags(ffi))
    run.advance(T)
    traj, acc = run.trajectory(), run.accept()
    _slices_match_oracle(ffi, traj, acc, lambda c0, w: o.Ensemble(p, sigma, np.tile(x0, (w, 1)), 9, chain0=c0), K, T)
    lp = run.logpost()
    assert np.array_equal(lp[-1], prob.logdensity(traj[-1])[0])      # recorded logpi == batch evaluate_log
    assert (run.summary()["status"] == 0).all()
    # a wide proposal from the stiff corner: non-finite solves are flagged and never accepted
    stiff = ffi.Run(prob, 9, 4096, np.array([[-7.0, 2.8]]), 4, _flags(ffi))
    stiff.advance(4)
    st = stiff.summary()["status"]
    assert np.isfinite(stiff.trajectory()).all()
    e = o.Ensemble(p, sigma, np.tile([[-7.0, 2.8]], (64, 1)), 9)
    out = e.advance(4)
    assert np.array_equal(stiff.trajectory()[:, :64], out["traj"]) and np.array_equal(st[:64], e.status)


@pytest.mark.parametrize("S", [2, 3, 4, 5, 6, 10, 7])
def test_two_level_compile_time_sub_chain_lengths(ffi, ctx, golden, S):
    """mlda2_static_kernel (csrc/ymc_mlda2.hip: the sub-chain unrolled, accept uniforms from whole Philox
    blocks on a fixed schedule) against the oracle for every instantiated length (7 is not one: run-time
    kernel), launches that start on odd and even fine steps and end inside a unit, a ragged ensemble."""
    from yagremcmc_amd import lowering
    g = golden("f5_mlda2_2d.npz")
    fine = _gauss_low(ffi, g["fine_mean"], g["fine_whiten"], g["fine_log_norm"])
    coarse = _gauss_low(ffi, g["coarse_mean"], g["coarse_whiten"], g["coarse_log_norm"])
    of = o.gauss_whitened(g["fine_mean"], g["fine_whiten"], float(g["fine_log_norm"]))
    oc = o.gauss_whitened(g["coarse_mean"], g["coarse_whiten"], float(g["coarse_log_norm"]))
    prob = lowering.make_problem(ctx, ffi.METHOD_MLDA2, fine, g["sigma"], surrogate=coarse, sub_chain_length=S)
    K = 200
    x0 = np.tile([-8.0, -7.0], (K, 1)) + np.random.default_rng(S).standard_normal((K, 2))
    for moments in (True, False):
        flags = ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT | (ffi.RECORD_MOMENTS if moments else 0)
        lengths = [1, 2, 3, 1, 8, 5, 4]
        run = ffi.Run(prob, 11, K, x0, sum(lengths), flags, chain_offset=1000)
        for n in lengths:
            run.advance(n)
        e = o.Ensemble(of, g["sigma"], x0, 11, chain0=1000, surrogate=oc, sub_chain_length=S)
        out = e.advance(sum(lengths))
        assert np.array_equal(run.trajectory(), out["traj"]) and np.array_equal(run.accept(), out["acc"])
        s, (x, lp) = run.summary(), run.state()
        assert np.array_equal(x, e.x) and np.array_equal(lp, e.lp) and np.array_equal(s["accepted"], e.acc_count)
        if moments:
            assert np.array_equal(s["wf_mean"], e.wf_mean) and np.array_equal(s["wf_m2"], e.wf_m2)
    if S == 6:   # the live reference's chains (fixture F5) through the compile-time kernel
        n = int(g["n_states"])
        run = ffi.Run(prob, int(g["seed"]), g["x0"].shape[0], g["x0"], n - 1, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT)
        run.advance(n - 1)
        assert np.array_equal(run.accept(), g["acc"]) and np.array_equal(run.trajectory(), g["traj"][1:])


@pytest.mark.parametrize("kind", ["linear", "lv", "gauss1"])
def test_two_level_compile_time_other_densities(ffi, ctx, kind):
    from yagremcmc_amd import lowering
    rng = np.random.default_rng(12)
    K, S, T = 96, 3, 11
    if kind == "linear":
        d, m, nd = 2, 3, 4
        def mk(seed):
            r = np.random.default_rng(seed)
            gg = dict(data=r.standard_normal((nd, m)), noise_precision=r.uniform(0.5, 2, m), prior_mean=r.standard_normal(d),
                      prior_precision=r.uniform(0.2, 1, d))
            A, b = r.standard_normal((m, d)), r.standard_normal(m)
            return (_posterior_low(ffi, ffi.POSTERIOR_LINEAR, d, gg, fwd_matrix=A, fwd_shift=b),
                    o.posterior_linear(A, b, gg["data"], gg["noise_precision"], gg["prior_mean"], gg["prior_precision"]))
        (lf, pf), (lc, pc) = mk(1), mk(2)
        sigma, x0 = np.array([0.3, 0.5]), rng.standard_normal((K, 2))
    elif kind == "lv":
        design = rng.uniform(0.5, 1.5, size=(4, 2))
        base = dict(noise_precision=np.array([25.0, 20.0]), prior_mean=np.zeros(2), prior_precision=np.array([0.7, 0.9]))
        data = rng.uniform(0.2, 2.0, size=(4, 2))
        def mk(steps):
            return (_posterior_low(ffi, ffi.POSTERIOR_LV, 2, dict(base, data=data), design=design, t_final=5.0, alpha=0.8,
                                   gamma=0.4, rk4_steps=steps),
                    o.posterior_lv(design, 5.0, 0.8, 0.4, steps, data, **base))
        (lf, pf), (lc, pc) = mk(40), mk(9)
        sigma, x0 = np.array([0.2, 0.2]), np.tile([[-0.6, -0.3]], (K, 1))
    else:
        d = 1
        def mk(mean, whiten):
            low = _gauss_low(ffi, np.array([mean]), np.array([[whiten]]), 1.0)
            return low, o.gauss_whitened(np.array([mean]), np.array([[whiten]]), 1.0)
        (lf, pf), (lc, pc) = mk(0.5, 1.2), mk(0.3, 0.8)
        sigma, x0 = np.array([0.9]), rng.standard_normal((K, 1))
    prob = lowering.make_problem(ctx, ffi.METHOD_MLDA2, lf, sigma, surrogate=lc, sub_chain_length=S)
    run = ffi.Run(prob, 5, K, x0, T, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT | ffi.RECORD_MOMENTS, chain_offset=3)
    run.advance(4)
    run.advance(7)
    e = o.Ensemble(pf, sigma, x0, 5, chain0=3, surrogate=pc, sub_chain_length=S)
    out = e.advance(T)
    assert np.array_equal(run.trajectory(), out["traj"]) and np.array_equal(run.accept(), out["acc"])
    s = run.summary()
    assert np.array_equal(s["wf_mean"], e.wf_mean) and np.array_equal(s["status"], e.status)


@pytest.mark.parametrize("split", ["0", "1"])
@pytest.mark.parametrize("d", [1, 2])
def test_random_walk_with_helper_waves_for_the_draws(ffi, ctx, golden, monkeypatch, split, d):
    """mrw_split_kernel (draws made by three helper waves per chain wave, handed over through LDS) against the
    one-lane-per-chain kernel and the oracle: ragged ensemble (300 chains: a partly filled workgroup and
    wave), launches of every length modulo the block of six, odd and even first transitions."""
    from yagremcmc_amd import lowering
    monkeypatch.setenv("YMC_SPLIT", split)
    g = golden("f2_mrw_2d.npz")
    if d == 2:
        low = _gauss_low(ffi, g["mean"], g["whiten"], g["log_norm"])
        p, sigma = o.gauss_whitened(g["mean"], g["whiten"], float(g["log_norm"])), g["sigma"]
        x0 = np.tile([-8.0, -7.0], (300, 1)) + np.random.default_rng(1).standard_normal((300, 2))
    else:
        low = lowering.Lowered(ffi.GAUSS_ISO1D, 1)
        low.set_array("mean", np.array([1.5]))
        low.desc.variance = 1.5
        p, sigma = o.gauss_iso1d(1.5, 1.5), np.array([1.3])
        x0 = np.random.default_rng(2).standard_normal((300, 1)) - 3.0
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, sigma)
    for moments in (True, False):
        flags = ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT | (ffi.RECORD_MOMENTS if moments else 0)
        lengths = [1, 5, 6, 7, 12, 13, 2, 25]
        run = ffi.Run(prob, 31, 300, x0, sum(lengths), flags, chain_offset=77)
        for n in lengths:
            run.advance(n)
        e = o.Ensemble(p, sigma, x0, 31, chain0=77)
        out = e.advance(sum(lengths))
        assert np.array_equal(run.trajectory(), out["traj"]) and np.array_equal(run.accept(), out["acc"])
        s, (x, lp) = run.summary(), run.state()
        assert np.array_equal(x, e.x) and np.array_equal(lp, e.lp) and np.array_equal(s["accepted"], e.acc_count)
        if moments:
            assert np.array_equal(s["wf_mean"], e.wf_mean) and np.array_equal(s["wf_m2"], e.wf_m2)
    # the reference's own chains (fixture F2, 64 chains x 2000 states) through the same kernel
    if d == 2:
        run = ffi.Run(prob, int(g["seed"]), g["x0"].shape[0], g["x0"], int(g["n_states"]) - 1,
                      ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT | ffi.RECORD_MOMENTS)
        run.advance(int(g["n_states"]) - 1)
        assert np.array_equal(run.trajectory(), g["traj"][1:]) and np.array_equal(run.accept(), g["acc"])
        assert np.array_equal(run.summary()["wf_mean"], g["wf_mean"])


@pytest.mark.parametrize("pieces,K,steps", [(12, 60, (300,)), (3, 11, (100, 200, 50)), (23, 131, (140,)), (16, 96, (60, 240)),
                                           (0, 131, (140,))])
def test_lotka_volterra_launch_in_handover_pieces(ffi, ctx, monkeypatch, pieces, K, steps):
    """mrw_lv_rows_balanced_kernel: the launch cut into equal pieces of the ribbon of all chain waves' transitions, every
    wave running the head of one chain wave and then the tail of its predecessor's (state handed over through the run's
    arrays behind a flag).  Forced here with piece counts that cut chain waves at various places (the product chooses a
    whole number of waves per SIMD): == oracle bit for bit -- chains, decisions, log-posteriors, moments, counters, the
    flagged stiff chains --, no chain flagged with YMC_CHAIN_HANDOFF, across launches, with and without a thinned record."""
    monkeypatch.setenv("YMC_LV_ROWS", "1")
    monkeypatch.setenv("YMC_LV_PIECES", str(pieces))
    from yagremcmc_amd import lowering
    rng = np.random.default_rng(80 + pieces)
    design = rng.uniform(0.5, 1.5, size=(10, 2))
    base = dict(noise_precision=np.array([25.0, 20.0]), prior_mean=np.array([0.1, -0.2]), prior_precision=np.array([0.7, 0.9]))
    data = rng.uniform(0.2, 2.0, size=(10, 2))
    p = o.posterior_lv(design, 10.0, 0.8, 0.4, 50, data, **base)
    low = _posterior_low(ffi, ffi.POSTERIOR_LV, 2, dict(base, data=data), design=design, t_final=10.0, alpha=0.8,
                         gamma=0.4, rk4_steps=50)
    x0 = np.tile([[-0.6, -0.3]], (K, 1)) + 0.05 * rng.normal(size=(K, 2))
    x0[-3:] = [-7.0, 2.8]                                   # stiff starts: flagged non-finite, never accepted
    sigma = np.array([0.4, 0.3])
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, sigma)
    total = sum(steps)
    run = ffi.Run(prob, 17, K, x0, total, _flags(ffi, True), 4000)
    for n in steps:
        run.advance(n)
    run.sync()
    dev = dict(traj=run.trajectory(), acc=run.accept(), logpost=run.logpost(), summary=run.summary(), state=run.state())
    e = o.Ensemble(p, sigma, x0, 17, chain0=4000)
    out = e.advance(total)
    _same_as_oracle(dev, e, out)
    assert not (dev["summary"]["status"] & ffi.CHAIN_HANDOFF).any()
    assert (dev["summary"]["status"][-3:] & ffi.CHAIN_NONFINITE).all()
    # thinned record: rows s-1::s of the full one, moments over every transition
    thin = ffi.Run(prob, 17, K, x0, total // 4 + 1, ffi.RECORD_TRAJECTORY | ffi.RECORD_MOMENTS, 4000, record_stride=4)
    for n in steps:
        thin.advance(n)
    assert np.array_equal(thin.trajectory(), dev["traj"][3::4])
    assert np.array_equal(thin.summary()["wf_mean"], e.wf_mean) and np.array_equal(thin.summary()["accepted"], e.acc_count)


def test_an_abandoned_handover_fails_the_run(ffi, ctx, monkeypatch):
    """a wave that gives up waiting for its predecessor's hand-over leaves transitions out of the
    record -- an internal error, never data.  The poll limit is lowered through the test-only knob so that the give-up
    path really runs: the chains carry YMC_CHAIN_HANDOFF, `ymc_run_sync` and every fetch of states return
    YMC_ERR_DEVICE with a message (so `ensemble.run`, which syncs, raises), the summary (the status bits) stays
    readable.  The reference aborts a chain whose evaluation failed with an exception (model/forwardModel.py:15-19)."""
    monkeypatch.setenv("YMC_LV_ROWS", "1")
    monkeypatch.setenv("YMC_LV_PIECES", "12")
    monkeypatch.setenv("YMC_LV_POLL_LIMIT", "0")
    from yagremcmc_amd import lowering
    rng = np.random.default_rng(81)
    design = rng.uniform(0.5, 1.5, size=(10, 2))
    base = dict(noise_precision=np.array([25.0, 20.0]), prior_mean=np.array([0.1, -0.2]), prior_precision=np.array([0.7, 0.9]))
    data = rng.uniform(0.2, 2.0, size=(10, 2))
    low = _posterior_low(ffi, ffi.POSTERIOR_LV, 2, dict(base, data=data), design=design, t_final=10.0, alpha=0.8,
                         gamma=0.4, rk4_steps=50)
    K = 60
    x0 = np.tile([[-0.6, -0.3]], (K, 1))
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, np.array([0.4, 0.3]))
    run = ffi.Run(prob, 17, K, x0, 300, _flags(ffi, True))
    run.advance(300)
    with pytest.raises(ffi.YmcError, match="internal device error.*YMC_CHAIN_HANDOFF") as err:
        run.sync()
    assert err.value.code == ffi.ERR_DEVICE
    for fetch in (run.trajectory, run.accept, run.logpost, run.state):
        with pytest.raises(ffi.YmcError, match="internal device error"):
            fetch()
    status = run.summary()["status"]
    assert (status & ffi.CHAIN_HANDOFF).any() and not (status & ffi.CHAIN_NONFINITE).any()
    # sticky: a later launch does not make the record whole again
    monkeypatch.setenv("YMC_LV_POLL_LIMIT", str(1 << 22))
    run.rewind_record()
    run.advance(300)
    with pytest.raises(ffi.YmcError, match="internal device error"):
        run.sync()
    # a fresh run of the same problem with the product's limit is fine
    good = ffi.Run(prob, 17, K, x0, 300, _flags(ffi, True))
    good.advance(300)
    good.sync()
    assert not good.summary()["status"].any()


@pytest.mark.parametrize("rows", ["0", "1"])
def test_lotka_volterra_row_parallel_variant(ffi, ctx, golden, monkeypatch, rows):
    """One lane per chain vs one lane per (chain, design row): a speed choice, same bits."""
    monkeypatch.setenv("YMC_LV_ROWS", rows)
    g = golden("f4_lotka_volterra.npz")
    n = int(g["n_states"])
    low = _posterior_low(ffi, ffi.POSTERIOR_LV, 2, g, design=np.asarray(g["design"]), t_final=float(g["t_final"]),
                         alpha=float(g["alpha"]), gamma=float(g["gamma"]), rk4_steps=int(g["rk4_steps"]))
    dev = _device_run(ffi, ctx, low, g["sigma"], g["x0"], int(g["seed"]), n - 1, splits=2)
    p = o.posterior_lv(g["design"], float(g["t_final"]), float(g["alpha"]), float(g["gamma"]), int(g["rk4_steps"]),
                       g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
    e = o.Ensemble(p, g["sigma"], g["x0"], int(g["seed"]))
    _same_as_oracle(dev, e, e.advance(n - 1))
    _same_as_reference(dev, g)
    # 10 rows, a chain count that leaves idle lanes and a ragged last wave; stiff start => flagged chains
    rng = np.random.default_rng(8)
    design = rng.uniform(0.5, 1.5, size=(10, 2))
    base = dict(noise_precision=np.array([25.0, 20.0]), prior_mean=np.array([0.1, -0.2]), prior_precision=np.array([0.7, 0.9]))
    data = rng.uniform(0.2, 2.0, size=(10, 2))
    p = o.posterior_lv(design, 10.0, 0.8, 0.4, 50, data, **base)
    low = _posterior_low(ffi, ffi.POSTERIOR_LV, 2, dict(base, data=data), design=design, t_final=10.0, alpha=0.8,
                         gamma=0.4, rk4_steps=50)
    x0 = np.concatenate([np.tile([[-0.6, -0.3]], (100, 1)), np.tile([[-7.0, 2.8]], (31, 1))])
    sigma = np.array([0.4, 0.3])
    dev = _device_run(ffi, ctx, low, sigma, x0, 3, 25, chain0=99)
    e = o.Ensemble(p, sigma, x0, 3, chain0=99)
    _same_as_oracle(dev, e, e.advance(25))
