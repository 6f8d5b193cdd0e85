"""Seeded fuzz of the HIP path against the oracle: random density kinds, dimensions, ensemble
sizes (ragged last wave), chain offsets, record-flag combinations, odd/even launch splits and
samplers.  Everything the device returns must equal the oracle bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# YMC_FUZZ_SHIFT=n moves every case to another seed (a hunt beyond the fixed cases: tools/fuzz_hunt.sh)
_SEED_SHIFT = 100003 * int(__import__("os").environ.get("YMC_FUZZ_SHIFT", "0"))

from oracle import oracle as o  # noqa: E402


def _lowered(ffi, kind, d, rng):
    from yagremcmc_amd import lowering
    if kind == "gauss":
        a = rng.normal(size=(d, d))
        p = o.gauss_from_cov(rng.normal(size=d), a @ a.T + d * np.eye(d))
        low = lowering.Lowered(ffi.GAUSS_WHITENED, d)
        low.set_array("mean", p["mean"])
        low.set_array("whiten", p["whiten"])
        low.desc.log_norm = p["log_norm"]
        return low, p
    if kind == "iso1d":
        p = o.gauss_iso1d(float(rng.normal()), float(rng.uniform(0.5, 2.0)))
        low = lowering.Lowered(ffi.GAUSS_ISO1D, 1)
        low.set_array("mean", p["mean"])
        low.desc.variance = p["variance"]
        return low, p
    if kind == "linear":
        m, nd = int(rng.integers(1, 9)), int(rng.integers(1, 7))
        g = dict(A=rng.normal(size=(m, d)), b=rng.normal(size=m), data=rng.normal(size=(nd, m)),
                 noise_precision=rng.uniform(0.5, 3, size=m), prior_mean=rng.normal(size=d) * 0.0,
                 prior_precision=rng.uniform(0.1, 2, size=d))
        p = o.posterior_linear(g["A"], g["b"], g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
        low = lowering.Lowered(ffi.POSTERIOR_LINEAR, d)
        low.desc.n_data, low.desc.data_dim = nd, m
        for k, v in (("data", g["data"]), ("noise_precision", g["noise_precision"]), ("prior_mean", g["prior_mean"]),
                     ("prior_precision", g["prior_precision"]), ("fwd_matrix", g["A"]), ("fwd_shift", g["b"])):
            low.set_array(k, v)
        return low, p
    nd = int(rng.integers(2, 12))
    design = rng.uniform(0.5, 1.5, size=(nd, 2))
    g = dict(data=rng.uniform(0.2, 2.0, size=(nd, 2)), noise_precision=rng.uniform(5, 30, size=2),
             prior_mean=np.zeros(2), prior_precision=rng.uniform(0.5, 2, size=2))
    steps, T = int(rng.integers(10, 40)), float(rng.uniform(2, 8))
    p = o.posterior_lv(design, T, 0.8, 0.4, steps, g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
    low = lowering.Lowered(ffi.POSTERIOR_LV, 2)
    low.desc.n_data, low.desc.data_dim = nd, 2
    for k, v in (("data", g["data"]), ("noise_precision", g["noise_precision"]), ("prior_mean", g["prior_mean"]),
                 ("prior_precision", g["prior_precision"]), ("design", design)):
        low.set_array(k, v)
    low.desc.t_final, low.desc.alpha, low.desc.gamma, low.desc.rk4_steps = T, 0.8, 0.4, steps
    return low, p


@pytest.mark.parametrize("case", range(48))
def test_random_configuration_matches_oracle(case, monkeypatch):
    from yagremcmc_amd import _ffi as ffi, lowering
    ctx = ffi.default_context(0)
    rng = np.random.default_rng(5000 + case + _SEED_SHIFT)
    kind = ["gauss", "gauss", "linear", "lv", "iso1d", "gauss"][case % 6]
    d = {"iso1d": 1, "lv": 2}.get(kind, int(rng.integers(1, 9)))
    low, p = _lowered(ffi, kind, d, rng)
    method = ["mrw", "mrw", "mlda", "pcn"][int(rng.integers(0, 4))]
    if method == "pcn" and kind not in ("linear", "lv"):
        method = "mrw"
    K = int(rng.choice([1, 7, 64, 65, 200, 513]))
    T = int(rng.integers(1, 60))
    chain0 = int(rng.choice([0, 3, 2**31, 2**32 - 1000]))
    seed = int(rng.integers(0, 2**63))
    flags = ffi.RECORD_ACCEPT | int(rng.choice([0, ffi.RECORD_TRAJECTORY])) | int(rng.choice([0, ffi.RECORD_LOGPOST])) \
        | int(rng.choice([0, ffi.RECORD_MOMENTS]))
    sigma = rng.uniform(0.05, 0.6, size=d)
    x0 = rng.normal(size=(K, d)) * 0.3
    if kind == "lv":
        x0 = x0 - np.array([0.6, 0.3])
        monkeypatch.setenv("YMC_LV_ROWS", str(case % 2))
    eq = ffi.EQ_ISCLOSE if kind == "iso1d" and case % 2 else ffi.EQ_EXACT
    kwargs = {}
    if method == "mlda":
        S = int(rng.integers(1, 5))
        low2, p2 = _lowered(ffi, kind, d, np.random.default_rng(9000 + case + _SEED_SHIFT)) if kind in ("gauss", "iso1d") else (low, p)
        prob = lowering.make_problem(ctx, ffi.METHOD_MLDA2, low, sigma, eq, surrogate=low2, sub_chain_length=S)
        e = o.Ensemble(p, sigma, x0, seed, chain0=chain0, eq_mode=eq, surrogate=p2, sub_chain_length=S)
        flags |= int(rng.choice([0, ffi.RECORD_SUB_ACCEPT]))
        kwargs["want_sub"] = True
    elif method == "pcn":
        low.desc.likelihood_only = 1
        p = dict(p, likelihood_only=1)
        h = float(rng.uniform(0.005, 0.3))
        prob = lowering.make_problem(ctx, ffi.METHOD_PCN, low, sigma, eq, pcn_step_size=h)
        e = o.Ensemble(p, sigma, x0, seed, chain0=chain0, eq_mode=eq, pcn_step_size=h)
    else:
        prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, sigma, eq)
        e = o.Ensemble(p, sigma, x0, seed, chain0=chain0, eq_mode=eq)
    run = ffi.Run(prob, seed, K, x0, T, flags, chain0)
    cut = int(rng.integers(0, T + 1))
    run.advance(cut)
    run.advance(T - cut)
    out = e.advance(T, **kwargs)
    assert np.array_equal(run.accept(), out["acc"])
    if flags & ffi.RECORD_TRAJECTORY:
        assert np.array_equal(run.trajectory(), out["traj"])
    if flags & ffi.RECORD_LOGPOST:
        assert np.array_equal(run.logpost(), out["logpost"], equal_nan=True)
    if flags & ffi.RECORD_SUB_ACCEPT:
        assert np.array_equal(run.sub_accept(), out["sub_acc"])
    x, lp = run.state()
    assert np.array_equal(x, e.x) and np.array_equal(lp, e.lp, equal_nan=True)
    s = run.summary()
    assert np.array_equal(s["accepted"], e.acc_count) and np.array_equal(s["status"], e.status)
    if flags & ffi.RECORD_MOMENTS:
        assert np.array_equal(s["wf_mean"], e.wf_mean) and np.array_equal(s["wf_m2"], e.wf_m2)
        assert (s["wf_n"] == T).all()


@pytest.mark.parametrize("case", range(40))
def test_random_round2_features_match_oracle(case, monkeypatch):
    """This round's paths, fuzzed against the oracle: thinned records (compile-time and run-time masks), the
    helper-wave random-walk kernel on / off, two-level kernels with compile-time and run-time sub-chain lengths,
    dense noise / prior covariances, the rows leaving through the egress buffers."""
    from scipy.linalg import solve_triangular
    from yagremcmc_amd import _ffi as ffi, lowering
    ctx = ffi.default_context(0)
    rng = np.random.default_rng(7000 + case + _SEED_SHIFT)
    kind = ["gauss", "linear", "gauss", "lv", "linear"][case % 5]
    d = 2 if kind == "lv" else int(rng.choice([1, 2, 2, 3, 5]))
    low, p = _lowered(ffi, kind, d, rng)
    if kind == "linear" and case % 2:       # dense noise and prior covariances
        m = int(low.desc.data_dim)

        def spd(n):
            B = rng.normal(size=(n, n))
            return B @ B.T + 0.3 * n * np.eye(n)
        nc, pc = spd(m), spd(d)
        for field, cov in (("noise_chol_inv", nc), ("prior_chol_inv", pc)):
            L = np.linalg.cholesky(cov)
            low.set_array(field, np.tril(solve_triangular(L, np.eye(L.shape[0]), lower=True)))
        p = o.dense_covariances(p, noise_cov=nc, prior_cov=pc)
    monkeypatch.setenv("YMC_SPLIT", str(case % 2))
    if kind == "lv":
        monkeypatch.setenv("YMC_LV_ROWS", str((case // 5) % 2))
    method = "mlda" if case % 3 == 0 else "mrw"
    K = int(rng.choice([5, 64, 130, 300]))
    stride = int(rng.choice([1, 1, 2, 3, 7]))
    T = int(rng.integers(stride, 70))
    chain0 = int(rng.choice([0, 11, 2**31 + 5]))
    seed = int(rng.integers(0, 2**63))
    flags = ffi.RECORD_TRAJECTORY | int(rng.choice([0, ffi.RECORD_MOMENTS])) | int(rng.choice([0, 0, ffi.RECORD_LOGPOST]))
    if stride == 1:
        flags |= ffi.RECORD_ACCEPT
    sigma = rng.uniform(0.05, 0.6, size=d)
    x0 = rng.normal(size=(K, d)) * 0.3 - (np.array([0.6, 0.3]) if kind == "lv" else 0.0)
    kwargs = {}
    if method == "mlda":
        S = int(rng.choice([1, 2, 3, 4, 6, 7]))        # 2, 3, 4, 6 have compile-time kernels at d <= 2
        low2, p2 = _lowered(ffi, kind, d, np.random.default_rng(9500 + case + _SEED_SHIFT)) if kind == "gauss" else (low, p)
        with np.errstate(all="ignore"):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                prob = lowering.make_problem(ctx, ffi.METHOD_MLDA2, low, sigma, surrogate=low2, sub_chain_length=S)
        e = o.Ensemble(p, sigma, x0, seed, chain0=chain0, surrogate=p2, sub_chain_length=S)
    else:
        prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, sigma)
        e = o.Ensemble(p, sigma, x0, seed, chain0=chain0)
    run = ffi.Run(prob, seed, K, x0, max(T // stride, 1), flags, chain0, record_stride=stride)
    cut = int(rng.integers(0, T + 1))
    run.advance(cut)
    run.advance(T - cut)
    out = e.advance(T, **kwargs)
    rows = T // stride
    want_traj, want_lp = out["traj"][stride - 1::stride][:rows], out["logpost"][stride - 1::stride][:rows]
    assert run.steps_done() == (T, rows)
    if rows:
        assert np.array_equal(run.trajectory(), want_traj)
        if flags & ffi.RECORD_LOGPOST:
            assert np.array_equal(run.logpost(), want_lp, equal_nan=True)
        first = int(rng.integers(0, rows))                       # the same rows through the egress buffers
        run.egress_begin(case % 2, first, rows - first, ffi.RECORD_TRAJECTORY)
        assert np.array_equal(run.egress_wait(case % 2)["trajectory"], want_traj[first:])
    if flags & ffi.RECORD_ACCEPT:
        assert np.array_equal(run.accept(), out["acc"])
    x, lp = run.state()
    assert np.array_equal(x, e.x) and np.array_equal(lp, e.lp, equal_nan=True)
    s = run.summary()
    assert np.array_equal(s["accepted"], e.acc_count) and np.array_equal(s["status"], e.status)
    if flags & ffi.RECORD_MOMENTS:
        assert np.array_equal(s["wf_mean"], e.wf_mean) and np.array_equal(s["wf_m2"], e.wf_m2)


class _Pre:
    """a density that is already lowered (lets the lowering layer pass it through)"""

    def __init__(self, low):
        self.low = low


def _nested_problem(ctx, lowering, tgt, surs, subs, sigma, eq):
    real = lowering.lower_density
    lowering.lower_density = lambda d, rk4_steps=None: d.low if isinstance(d, _Pre) else real(d, rk4_steps)
    try:
        return lowering.make_nested_problem(ctx, _Pre(tgt), [_Pre(s_) for s_ in surs], subs, sigma, eq)
    finally:
        lowering.lower_density = real


@pytest.mark.parametrize("case", range(24))
def test_random_nested_hierarchy_matches_oracle(case):
    """MLDA over 2, 4 or 6 surrogates of every density kind (the nest of csrc/ymc_nested.hip)."""
    from yagremcmc_amd import _ffi as ffi, lowering
    ctx = ffi.default_context(0)
    rng = np.random.default_rng(7000 + case + _SEED_SHIFT)
    kind = ["gauss", "linear", "lv", "iso1d"][case % 4]
    d = {"iso1d": 1, "lv": 2}.get(kind, int(rng.integers(1, 9)))
    E = int(rng.integers(2, 5))
    ns = 2 * (E - 1)
    tgt_low, tgt_p = _lowered(ffi, kind, d, rng)
    surs = [_lowered(ffi, kind, d, np.random.default_rng(8000 + 10 * case + i)) for i in range(ns)]
    if kind == "lv":  # keep the ODE cheap and the shapes of all levels equal (design rows may differ per level)
        pass
    subs = [int(v) for v in rng.integers(1, 4, size=ns)]
    K = int(rng.choice([1, 33, 64, 130]))
    T = int(rng.integers(1, 25))
    chain0 = int(rng.choice([0, 5, 2**31]))
    seed = int(rng.integers(0, 2**63))
    eq = ffi.EQ_ISCLOSE if kind == "iso1d" and case % 8 >= 4 else ffi.EQ_EXACT
    flags = ffi.RECORD_ACCEPT | ffi.RECORD_TRAJECTORY | int(rng.choice([0, ffi.RECORD_LOGPOST])) \
        | int(rng.choice([0, ffi.RECORD_MOMENTS]))
    sigma = rng.uniform(0.1, 0.5, size=d)
    x0 = rng.normal(size=(K, d)) * 0.3
    if kind == "lv":
        x0 = x0 - np.array([0.6, 0.3])
    prob = _nested_problem(ctx, lowering, tgt_low, [s_[0] for s_ in surs], subs, sigma, eq)
    run = ffi.Run(prob, seed, K, x0, T, flags, chain0)
    cut = int(rng.integers(0, T + 1))
    run.advance(cut)
    run.advance(T - cut)
    P, Q, steps = o.nested_levels([s_[1] for s_ in surs], subs, tgt_p)
    e = o.NestedEnsemble(P, Q, steps, sigma, x0, seed, chain0=chain0, eq_mode=eq)
    out = e.advance(T)
    assert np.array_equal(run.accept(), out["acc"]) and np.array_equal(run.trajectory(), out["traj"])
    if flags & ffi.RECORD_LOGPOST:
        assert np.array_equal(run.logpost(), out["logpost"], equal_nan=True)
    x, lp = run.state()
    assert np.array_equal(x, e.x) and np.array_equal(lp, e.lp, equal_nan=True)
    s = run.summary()
    assert np.array_equal(s["accepted"], e.acc_count) and np.array_equal(s["status"], e.status)
    assert np.array_equal(run.level_accept()[0], e.level_acc)
    if flags & ffi.RECORD_MOMENTS:
        assert np.array_equal(s["wf_mean"], e.wf_mean) and np.array_equal(s["wf_m2"], e.wf_m2)


@pytest.mark.parametrize("case", range(16))
def test_random_adaptive_error_model_matches_oracle(case):
    from yagremcmc_amd import _ffi as ffi, lowering
    ctx = ffi.default_context(0)
    rng = np.random.default_rng(11000 + case + _SEED_SHIFT)
    d, m, nd = int(rng.integers(1, 9)), int(rng.integers(1, 9)), int(rng.integers(1, 7))
    A = rng.normal(size=(m, d)) / np.sqrt(d)
    b = rng.normal(size=m) * 0.3
    As, bs = A + 0.3 * rng.normal(size=(m, d)), b + 0.3 * rng.normal(size=m)
    common = dict(data=rng.normal(size=(nd, m)), noise_precision=rng.uniform(1, 6, size=m), prior_mean=np.zeros(d),
                  prior_precision=rng.uniform(0.2, 1, size=d))

    def low_of(A_, b_):
        low = lowering.Lowered(ffi.POSTERIOR_LINEAR, d)
        low.desc.n_data, low.desc.data_dim = nd, m
        for k, v in dict(common, fwd_matrix=A_, fwd_shift=b_).items():
            low.set_array(k, v)
        return low

    S, md, heur = int(rng.integers(1, 5)), int(rng.integers(2, 7)), int(rng.integers(0, 2))
    K, T = int(rng.choice([1, 65, 200])), int(rng.integers(20, 150))
    chain0, seed = int(rng.choice([0, 9, 2**31])), int(rng.integers(0, 2**63))
    sigma = rng.uniform(0.1, 0.5, size=d)
    x0 = rng.normal(size=(K, d)) * 0.3
    flags = ffi.RECORD_ACCEPT | ffi.RECORD_TRAJECTORY | ffi.RECORD_SUB_ACCEPT | int(rng.choice([0, ffi.RECORD_MOMENTS])) \
        | int(rng.choice([0, ffi.RECORD_LOGPOST]))
    prob = lowering.make_problem(ctx, ffi.METHOD_AEM2, low_of(A, b), sigma, ffi.EQ_EXACT, surrogate=low_of(As, bs),
                                 sub_chain_length=S, aem=(md, heur))
    run = ffi.Run(prob, seed, K, x0, T, flags, chain0)
    cut = int(rng.integers(0, T + 1))
    run.advance(cut)
    run.advance(T - cut)
    e = o.AemEnsemble(o.posterior_linear(A, b, **common), o.posterior_linear(As, bs, **common), sigma, x0, seed, S, md,
                      heur, chain0=chain0)
    out = e.advance(T, want_sub=True)
    assert np.array_equal(run.accept(), out["acc"]) and np.array_equal(run.trajectory(), out["traj"])
    assert np.array_equal(run.sub_accept(), out["sub_acc"])
    err = run.error_model()
    assert np.array_equal(err["n"], e.aem_n) and np.array_equal(err["mean"], e.aem_mean)
    assert np.array_equal(err["m2"], e.aem_m2) and np.array_equal(err["precision"], e.aem_prec)
    x, lp = run.state()
    assert np.array_equal(x, e.x) and np.array_equal(lp, e.lp)
    if flags & ffi.RECORD_MOMENTS:
        s = run.summary()
        assert np.array_equal(s["wf_mean"], e.wf_mean) and np.array_equal(s["wf_m2"], e.wf_m2)


@pytest.mark.parametrize("case", range(96))
def test_random_matrix_core_configuration_matches_oracle(case):
    """The samplers of the matrix-core path (random walk with a diagonal or dense proposal, pCN, two-level, nested)
    over random shapes -- parameter dimension 9..128 or a small one with many outputs, per-level output counts,
    dense or diagonal noise, dense or diagonal prior where the LDS admits it --, ensemble sizes with a partly
    filled last wave, chain offsets, launch splits, record flags and a thinned record."""
    from scipy.linalg import solve_triangular
    from yagremcmc_amd import _ffi as ffi, lowering
    ctx = ffi.default_context(0)
    rng = np.random.default_rng(9000 + case + _SEED_SHIFT)
    sampler = ["mrw", "pcn", "mlda2", "nested", "mrw_chol", "mrw"][case % 6]
    d = int(rng.choice([5, 9, 12, 16, 17, 33, 64, 100, 128]))
    small = d <= 8
    nd = int(rng.integers(1, 6))
    dense_noise = bool(rng.integers(0, 2))
    dense_prior = bool(rng.integers(0, 2)) and sampler not in ("pcn",) and d <= 40

    def spd(n, ridge):
        B = rng.normal(size=(n, n)) / np.sqrt(n)
        return B @ B.T + ridge * np.eye(n)

    def inv_factor(cov):
        return np.tril(solve_triangular(np.linalg.cholesky(cov), np.eye(cov.shape[0]), lower=True))

    def model(scale):
        m = int(rng.integers(9, 41)) if small else int(rng.integers(1, min(d, 48) + 1))
        nc = spd(m, 0.4) if dense_noise else None
        pc = spd(d, 1.0) if dense_prior else None
        g = dict(data=rng.normal(size=(nd, m)), noise_precision=np.ones(m) if dense_noise else rng.uniform(0.5, 3, size=m),
                 prior_mean=rng.normal(size=d) * (0.0 if sampler == "pcn" else 0.2),
                 prior_precision=np.ones(d) if dense_prior else rng.uniform(0.2, 2, size=d))
        A, b = scale * rng.normal(size=(m, d)) / np.sqrt(d), rng.normal(size=m)
        low = lowering.Lowered(ffi.POSTERIOR_LINEAR, d)
        low.desc.n_data, low.desc.data_dim = nd, m
        for k_, v in (("data", g["data"]), ("noise_precision", g["noise_precision"]), ("prior_mean", g["prior_mean"]),
                      ("prior_precision", g["prior_precision"]), ("fwd_matrix", A), ("fwd_shift", b)):
            low.set_array(k_, v)
        if nc is not None:
            low.set_array("noise_chol_inv", inv_factor(nc))
        if pc is not None:
            low.set_array("prior_chol_inv", inv_factor(pc))
        p = o.posterior_linear(A, b, g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
        if nc is not None or pc is not None:
            p = o.dense_covariances(p, noise_cov=nc, prior_cov=pc, whiten=True)
        return low, p

    K = int(rng.choice([1, 15, 16, 33, 64, 130]))
    T = int(rng.integers(2, 22))
    chain0 = int(rng.choice([0, 7, 2**31]))
    seed = int(rng.integers(0, 2**63))
    stride = int(rng.choice([1, 1, 3]))
    flags = ffi.RECORD_TRAJECTORY | int(rng.choice([0, ffi.RECORD_LOGPOST]))
    if stride == 1:
        flags |= ffi.RECORD_ACCEPT | int(rng.choice([0, ffi.RECORD_MOMENTS]))
    sigma = rng.uniform(0.02, 0.1, size=d)
    x0 = rng.normal(size=(K, d)) * 0.5
    kw, okw = {}, {}
    if sampler == "pcn":
        low, p = model(1.0)
        low.desc.likelihood_only, p["likelihood_only"] = 1, 1
        sigma = np.sqrt(np.reciprocal(p["prior_precision"]))
        h = float(rng.uniform(0.005, 0.1))
        prob = lowering.make_problem(ctx, ffi.METHOD_PCN, low, sigma, pcn_step_size=h)
        e = o.Ensemble(p, sigma, x0, seed, chain0=chain0, pcn_step_size=h)
    elif sampler == "mlda2":
        (low, p), (lc, pc_) = model(1.0), model(0.92)
        S = int(rng.integers(1, 5))
        prob = lowering.make_problem(ctx, ffi.METHOD_MLDA2, low, sigma, surrogate=lc, sub_chain_length=S)
        e = o.Ensemble(p, sigma, x0, seed, chain0=chain0, surrogate=pc_, sub_chain_length=S)
    elif sampler == "nested":
        E = int(rng.integers(2, 4))
        models = [model(1.0 - 0.03 * i) for i in range(2 * (E - 1) + 1)]
        subs = [int(v) for v in rng.integers(1, 4, size=len(models) - 1)]
        prob = _nested_problem(ctx, lowering, models[-1][0], [l for l, _ in models[:-1]], subs, sigma, ffi.EQ_EXACT)
        P, Q, steps = o.nested_levels([q for _, q in models[:-1]], subs, models[-1][1])
        e = o.NestedEnsemble(P, Q, steps, sigma, x0, seed, chain0=chain0)
    elif sampler == "mrw_chol":
        low, p = model(1.0)
        L = np.linalg.cholesky(spd(d, 1.0) * 0.004)
        sigma = np.ones(d)
        prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, sigma, chol=np.tril(L))
        e = o.Ensemble(p, sigma, x0, seed, chain0=chain0, chol=L)
    else:
        low, p = model(1.0)
        prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, sigma)
        e = o.Ensemble(p, sigma, x0, seed, chain0=chain0)
    run = ffi.Run(prob, seed, K, x0, T // stride + 1, flags, chain0, record_stride=stride)
    cut = int(rng.integers(0, T + 1))
    run.advance(cut)
    run.advance(T - cut)
    out = e.advance(T)
    rows = slice(stride - 1, None, stride)
    assert np.array_equal(run.trajectory(), out["traj"][rows])
    if flags & ffi.RECORD_ACCEPT:
        assert np.array_equal(run.accept(), out["acc"])
    if flags & ffi.RECORD_LOGPOST:
        assert np.array_equal(run.logpost(), out["logpost"][rows])
    x, lp = run.state()
    assert np.array_equal(x, e.x) and np.array_equal(lp, e.lp)
    s = run.summary()
    assert np.array_equal(s["accepted"], e.acc_count) and np.array_equal(s["status"], e.status)
    if flags & ffi.RECORD_MOMENTS:
        assert np.array_equal(s["wf_mean"], e.wf_mean) and np.array_equal(s["wf_m2"], e.wf_m2)
    if sampler == "nested":
        assert np.array_equal(run.level_accept()[0], e.level_acc)


def _linear_pair(ffi, d, m, nd, rng, dense_prior=False, scale=1.0):
    from scipy.linalg import solve_triangular
    from yagremcmc_amd import lowering
    g = dict(A=scale * rng.normal(size=(m, d)) / np.sqrt(d), b=rng.normal(size=m), data=rng.normal(size=(nd, m)),
             noise_precision=rng.uniform(0.5, 3, size=m), prior_mean=rng.normal(size=d) * 0.1,
             prior_precision=np.ones(d) if dense_prior else rng.uniform(0.1, 2, size=d))
    p = o.posterior_linear(g["A"], g["b"], g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"])
    low = lowering.Lowered(ffi.POSTERIOR_LINEAR, d)
    low.desc.n_data, low.desc.data_dim = nd, m
    for k, v in (("data", g["data"]), ("noise_precision", g["noise_precision"]), ("prior_mean", g["prior_mean"]),
                 ("prior_precision", g["prior_precision"]), ("fwd_matrix", g["A"]), ("fwd_shift", g["b"])):
        low.set_array(k, v)
    if dense_prior:
        B = rng.normal(size=(d, d)) / np.sqrt(d)
        pc = B @ B.T + np.eye(d)
        L = np.linalg.cholesky(pc)
        low.set_array("prior_chol_inv", np.tril(solve_triangular(L, np.eye(d), lower=True)))
        p = o.dense_covariances(p, prior_cov=pc)
    return low, p


class _Pre:
    def __init__(self, low):
        self.low = low


@pytest.mark.parametrize("case", range(96))
def test_random_round3_features_match_oracle(case, monkeypatch):
    """Round 3's paths, fuzzed against the oracle: nests of up to eight effective levels (lane and matrix-core kernels),
    Lotka-Volterra levels integrated adaptively (lane per chain and lane per design row, two-level, nested), the
    adaptive error model with more than eight data rows, and the streamed builds of the matrix-core samplers (a base /
    coarse / forward matrix beyond the LDS, parameter dimensions beyond 128, dense proposals and priors there)."""
    from yagremcmc_amd import _ffi as ffi, lowering
    ctx = ffi.default_context(0)
    rng = np.random.default_rng(11000 + case + _SEED_SHIFT)
    feature = ["deep_nest", "lv_adaptive", "aem_rows", "stream_mlda2", "stream_nested", "stream_dense"][case % 6]
    K = int(rng.choice([5, 16, 33, 70]))
    seed, chain0 = int(rng.integers(0, 2**62)), int(rng.choice([0, 17, 2**31 + 5]))
    real = lowering.lower_density
    monkeypatch.setattr(lowering, "lower_density", lambda dd, rk4_steps=None: dd.low if isinstance(dd, _Pre) else real(dd, rk4_steps))

    def nested(models, subs, sigma, x0, n):
        (lt, pt), surs = models[-1], models[:-1]
        prob = lowering.make_nested_problem(ctx, _Pre(lt), [_Pre(l) for l, _ in surs], subs, sigma)
        run = ffi.Run(prob, seed, K, x0, n, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT | ffi.RECORD_LOGPOST | ffi.RECORD_MOMENTS, chain0)
        cut = int(rng.integers(0, n + 1))
        run.advance(cut)
        run.advance(n - cut)
        P, Q, steps = o.nested_levels([q for _, q in surs], subs, pt)
        e = o.NestedEnsemble(P, Q, steps, sigma, x0, seed, chain0=chain0)
        out = e.advance(n)
        assert np.array_equal(run.accept(), out["acc"]) and np.array_equal(run.trajectory(), out["traj"])
        assert np.array_equal(run.logpost(), out["logpost"])
        acc, _ = run.level_accept()
        assert np.array_equal(acc, e.level_acc)
        s = run.summary()
        assert np.array_equal(s["wf_mean"], e.wf_mean) and np.array_equal(s["wf_m2"], e.wf_m2)

    def two_level(lf, pf, lc, pc, sigma, x0, n, S):
        prob = lowering.make_problem(ctx, ffi.METHOD_MLDA2, lf, sigma, surrogate=lc, sub_chain_length=S)
        run = ffi.Run(prob, seed, K, x0, n, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT | ffi.RECORD_SUB_ACCEPT | ffi.RECORD_LOGPOST, chain0)
        cut = int(rng.integers(0, n + 1))
        run.advance(cut)
        run.advance(n - cut)
        e = o.Ensemble(pf, sigma, x0, seed, chain0=chain0, surrogate=pc, sub_chain_length=S)
        out = e.advance(n, want_sub=True)
        assert np.array_equal(run.accept(), out["acc"]) and np.array_equal(run.sub_accept(), out["sub_acc"])
        assert np.array_equal(run.trajectory(), out["traj"]) and np.array_equal(run.logpost(), out["logpost"])

    def lv(rtol):
        nd = int(rng.integers(2, 11))
        design = rng.uniform(0.5, 1.5, size=(nd, 2))
        g = dict(data=rng.uniform(0.2, 2.0, size=(nd, 2)), noise_precision=rng.uniform(5, 30, size=2), prior_mean=np.zeros(2),
                 prior_precision=rng.uniform(0.5, 2, size=2))
        lv.shape = lv.shape if hasattr(lv, "shape") else (design, g, float(rng.uniform(2, 9)))
        design, g, T = lv.shape
        steps = int(rng.integers(10, 40))
        p = o.posterior_lv(design, T, 0.8, 0.4, steps, g["data"], g["noise_precision"], g["prior_mean"], g["prior_precision"],
                           rtol=rtol)
        low = lowering.Lowered(ffi.POSTERIOR_LV, 2)
        low.desc.n_data, low.desc.data_dim = design.shape[0], 2
        for k, v in (("data", g["data"]), ("noise_precision", g["noise_precision"]), ("prior_mean", g["prior_mean"]),
                     ("prior_precision", g["prior_precision"]), ("design", design)):
            low.set_array(k, v)
        low.desc.t_final, low.desc.alpha, low.desc.gamma, low.desc.rk4_steps = T, 0.8, 0.4, steps
        low.desc.lv_rtol, low.desc.lv_atol = rtol, 1e-6 if rtol > 0 else 0.0
        return low, p

    if feature == "deep_nest":
        E = int(rng.integers(5, 9))
        tile = bool(case % 12 >= 6)
        d = int(rng.integers(9, 40)) if tile else int(rng.integers(1, 9))
        if tile:
            models = [_linear_pair(ffi, d, int(rng.integers(1, 40)), int(rng.integers(1, 4)), rng, scale=1.0 - 0.02 * i)
                      for i in range(2 * (E - 1) + 1)]
            sigma = rng.uniform(0.03, 0.1, size=d)
        else:
            models = [_lowered(ffi, "gauss", d, rng) for _ in range(2 * (E - 1) + 1)]
            sigma = rng.uniform(0.4, 1.0, size=d)
        subs = [int(v) for v in rng.integers(1, 3, size=2 * (E - 1))]
        nested(models, subs, sigma, rng.normal(size=(K, d)) * 0.5, int(rng.integers(2, 7)))
    elif feature == "lv_adaptive":
        monkeypatch.setenv("YMC_LV_ROWS", str((case // 6) % 2))
        rt = [1e-2, 1e-4, 1e-7]
        x0 = rng.normal(size=(K, 2)) * 0.25 - np.array([0.6, 0.3])
        sigma = rng.uniform(0.05, 0.3, size=2)
        mode = (case // 6) % 3
        if mode == 0:
            low, p = lv(rt[case % 3])
            prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, sigma)
            n = int(rng.integers(3, 40))
            run = ffi.Run(prob, seed, K, x0, n, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT | ffi.RECORD_LOGPOST, chain0)
            run.advance(n)
            out = o.Ensemble(p, sigma, x0, seed, chain0=chain0).advance(n)
            assert np.array_equal(run.accept(), out["acc"]) and np.array_equal(run.trajectory(), out["traj"])
            assert np.array_equal(run.logpost(), out["logpost"], equal_nan=True)
        elif mode == 1:
            (lf, pf), (lc, pc) = lv(1e-6), lv(float(rng.choice([1e-2, 0.0])))
            two_level(lf, pf, lc, pc, sigma, x0, int(rng.integers(3, 20)), int(rng.integers(1, 4)))
        else:
            models = [lv(float(r)) for r in rng.choice([1e-2, 1e-3, 0.0, 1e-5], size=5)]
            nested(models, [2, 1, 2, 2], sigma, x0, int(rng.integers(2, 8)))
    elif feature == "aem_rows":
        d, m, nd = int(rng.integers(9, 50)), int(rng.integers(3, 60)), int(rng.integers(9, 30))
        A, b = rng.normal(size=(m, d)) / np.sqrt(d), rng.normal(size=m) * 0.3
        As, bs = A + 0.05 * rng.normal(size=(m, d)), b + 0.1 * rng.normal(size=m)
        data = np.array([A @ rng.normal(size=d) * 0.2 + b + 0.5 * rng.normal(size=m) for _ in range(nd)])
        common = dict(data=data, noise_precision=rng.uniform(3, 5, size=m), prior_mean=np.zeros(d), prior_precision=np.full(d, 0.25))

        def low(A_, b_):
            l = lowering.Lowered(ffi.POSTERIOR_LINEAR, d)
            l.desc.n_data, l.desc.data_dim = nd, m
            for k, v in (("fwd_matrix", A_), ("fwd_shift", b_)) + tuple(common.items()):
                l.set_array(k, v)
            return l
        S, mind, heur = int(rng.integers(1, 4)), int(rng.integers(2, 6)), int(case % 2)
        sigma, x0 = np.full(d, 0.5 / np.sqrt(d)), rng.normal(size=(K, d)) * 0.3
        prob = lowering.make_problem(ctx, ffi.METHOD_AEM2, low(A, b), sigma, surrogate=low(As, bs), sub_chain_length=S, aem=(mind, heur))
        n = int(rng.integers(10, 40))
        run = ffi.Run(prob, seed, K, x0, n, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT | ffi.RECORD_SUB_ACCEPT, chain0)
        cut = int(rng.integers(0, n + 1))
        run.advance(cut)
        run.advance(n - cut)
        e = o.AemEnsemble(o.posterior_linear(A, b, **common), o.posterior_linear(As, bs, **common), sigma, x0, seed, S, mind, heur,
                          chain0=chain0)
        out = e.advance(n, want_sub=True)
        assert np.array_equal(run.accept(), out["acc"]) and np.array_equal(run.sub_accept(), out["sub_acc"])
        assert np.array_equal(run.trajectory(), out["traj"])
        err = run.error_model()
        assert np.array_equal(err["n"], e.aem_n) and np.array_equal(err["mean"], e.aem_mean) and np.array_equal(err["m2"], e.aem_m2)
    elif feature == "stream_mlda2":
        d = int(rng.choice([100, 128, 130, 200, 256]))
        mc, mf = int(rng.integers(10, 300)), int(rng.integers(10, 200))
        dense = bool(case % 12 >= 6) and d <= 160
        (lf, pf), (lc, pc) = _linear_pair(ffi, d, mf, 2, rng, dense), _linear_pair(ffi, d, mc, 2, rng, dense, scale=0.95)
        if case % 4 == 0:
            monkeypatch.setenv("YMC_TILE_MLDA2", "stream")
        two_level(lf, pf, lc, pc, rng.uniform(0.02, 0.05, size=d), rng.normal(size=(K, d)) * 0.5, int(rng.integers(3, 12)),
                  int(rng.integers(1, 4)))
    elif feature == "stream_nested":
        d = int(rng.choice([64, 128, 140, 256]))
        E = int(rng.integers(2, 5))
        ms = [int(rng.integers(8, 280 if i == 0 else 80)) for i in range(2 * (E - 1) + 1)]
        models = [_linear_pair(ffi, d, m, 2, rng, scale=1.0 - 0.02 * i) for i, m in enumerate(ms)]
        if case % 4 == 0:
            monkeypatch.setenv("YMC_TILE_NESTED", "stream")
        nested(models, [int(v) for v in rng.integers(1, 3, size=2 * (E - 1))], rng.uniform(0.02, 0.06, size=d),
               rng.normal(size=(K, d)) * 0.5, int(rng.integers(2, 8)))
    else:
        d, m = int(rng.choice([40, 128, 150, 256])), int(rng.choice([20, 200, 600]))
        low, p = _linear_pair(ffi, d, m, 2, rng, dense_prior=bool(case % 12 >= 6) and d <= 150)
        B = rng.normal(size=(d, d)) * 0.02 / np.sqrt(d)
        L = np.linalg.cholesky(B @ B.T + 0.002 * np.eye(d))
        x0 = rng.normal(size=(K, d)) * 0.5
        prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, np.ones(d), chol=np.tril(L))
        n = int(rng.integers(3, 30))
        run = ffi.Run(prob, seed, K, x0, n, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT | ffi.RECORD_LOGPOST, chain0)
        run.advance(n)
        out = o.Ensemble(p, np.ones(d), x0, seed, chain0=chain0, chol=L).advance(n)
        assert np.array_equal(run.accept(), out["acc"]) and np.array_equal(run.trajectory(), out["traj"])
        assert np.array_equal(run.logpost(), out["logpost"])


@pytest.mark.parametrize("case", range(30))
def test_random_wide_path_matches_oracle(case, monkeypatch):
    """The wide path (more than 256 parameters), fuzzed against the oracle: random walk with a diagonal or dense proposal, pCN,
    two-level and nested delayed acceptance; ragged ensembles, chain offsets, thinned records, launches cut at random."""
    from yagremcmc_amd import _ffi as ffi, lowering
    ctx = ffi.default_context(0)
    rng = np.random.default_rng(13000 + case + _SEED_SHIFT)
    method = ["mrw", "pcn", "mlda2", "nested", "dense"][case % 5]
    d = int(rng.integers(257, 700))
    K = int(rng.choice([1, 15, 16, 33, 70]))
    seed, chain0 = int(rng.integers(0, 2**62)), int(rng.choice([0, 9, 2**31 + 7]))
    n = int(rng.integers(2, 12))
    cut = int(rng.integers(0, n + 1))
    real = lowering.lower_density
    monkeypatch.setattr(lowering, "lower_density", lambda dd, rk4_steps=None: dd.low if isinstance(dd, _Pre) else real(dd, rk4_steps))
    x0 = rng.normal(size=(K, d)) * 0.3
    sigma = rng.uniform(0.01, 0.03, size=d)
    if method in ("mrw", "pcn", "dense"):
        low, p = _linear_pair(ffi, d, int(rng.integers(1, 300)), int(rng.integers(1, 4)), rng, dense_prior=(method == "mrw" and case % 2 == 1))
        stride = int(rng.choice([1, 1, 3]))
        flags = ffi.RECORD_TRAJECTORY | ffi.RECORD_LOGPOST | ffi.RECORD_MOMENTS | (ffi.RECORD_ACCEPT if stride == 1 else 0)
        if method == "pcn":
            low.set_array("prior_mean", np.zeros(d))
            low.desc.likelihood_only = 1
            p = dict(p, prior_mean=np.zeros(d), likelihood_only=1)
            sigma = 1.0 / np.sqrt(p["prior_precision"])
            h = float(rng.uniform(0.005, 0.05))
            prob = lowering.make_problem(ctx, ffi.METHOD_PCN, low, sigma, pcn_step_size=h)
            e = o.Ensemble(p, sigma, x0, seed, chain0=chain0, pcn_step_size=h)
        elif method == "dense":
            B = rng.normal(size=(d, d)) * 0.02 / np.sqrt(d)
            L = np.linalg.cholesky(B @ B.T + 0.0005 * np.eye(d))
            prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, np.ones(d), chol=np.tril(L))
            e = o.Ensemble(p, np.ones(d), x0, seed, chain0=chain0, chol=L)
        else:
            prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, sigma)
            e = o.Ensemble(p, sigma, x0, seed, chain0=chain0)
        run = ffi.Run(prob, seed, K, x0, n, flags, chain0, record_stride=stride)
        run.advance(cut)
        run.advance(n - cut)
        out = e.advance(n)
        rows = slice(stride - 1, None, stride)
        assert np.array_equal(run.trajectory(), out["traj"][rows]) and np.array_equal(run.logpost(), out["logpost"][rows])
        if stride == 1:
            assert np.array_equal(run.accept(), out["acc"])
        s_ = run.summary()
        assert np.array_equal(s_["wf_mean"], e.wf_mean) and np.array_equal(s_["wf_m2"], e.wf_m2)
        assert np.array_equal(s_["accepted"], e.acc_count)
    elif method == "mlda2":
        (lf, pf), (lc, pc) = _linear_pair(ffi, d, int(rng.integers(1, 200)), 2, rng), _linear_pair(ffi, d, int(rng.integers(1, 100)), 2, rng, scale=0.95)
        S = int(rng.integers(1, 4))
        prob = lowering.make_problem(ctx, ffi.METHOD_MLDA2, lf, sigma, surrogate=lc, sub_chain_length=S)
        run = ffi.Run(prob, seed, K, x0, n, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT | ffi.RECORD_SUB_ACCEPT | ffi.RECORD_LOGPOST, chain0)
        run.advance(cut)
        run.advance(n - cut)
        e = o.Ensemble(pf, sigma, x0, seed, chain0=chain0, surrogate=pc, sub_chain_length=S)
        out = e.advance(n, want_sub=True)
        assert np.array_equal(run.accept(), out["acc"]) and np.array_equal(run.sub_accept(), out["sub_acc"])
        assert np.array_equal(run.trajectory(), out["traj"]) and np.array_equal(run.logpost(), out["logpost"])
    else:
        E = int(rng.integers(2, 5))
        models = [_linear_pair(ffi, d, int(rng.integers(1, 60)), 2, rng, scale=1.0 - 0.02 * i) for i in range(2 * (E - 1) + 1)]
        (lt, pt), surs = models[-1], models[:-1]
        subs = [int(v) for v in rng.integers(1, 3, size=len(surs))]
        prob = lowering.make_nested_problem(ctx, _Pre(lt), [_Pre(l) for l, _ in surs], subs, sigma)
        run = ffi.Run(prob, seed, K, x0, n, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT | ffi.RECORD_LOGPOST | ffi.RECORD_MOMENTS, chain0)
        run.advance(cut)
        run.advance(n - cut)
        P, Q, steps = o.nested_levels([q for _, q in surs], subs, pt)
        e = o.NestedEnsemble(P, Q, steps, sigma, x0, seed, chain0=chain0)
        out = e.advance(n)
        assert np.array_equal(run.accept(), out["acc"]) and np.array_equal(run.trajectory(), out["traj"])
        assert np.array_equal(run.logpost(), out["logpost"])
        acc, _ = run.level_accept()
        assert np.array_equal(acc, e.level_acc)
    x, lp = run.state()
    assert np.array_equal(x, e.x) and np.array_equal(lp, e.lp)


@pytest.mark.parametrize("case", range(32))
def test_random_group_kernels_match_oracle(case, monkeypatch):
    """Round 4's group kernels (a tile of 16 chains over several workgroups: linear_wide_systolic_kernel, linear_wide_split_kernel),
    fuzzed against the oracle: random walk and pCN beyond 256 parameters on small ensembles; group sizes forced to odd values,
    both builds, both workgroup sizes of the streaming build, output counts from 1 to beyond one staging chunk, ragged
    ensembles, chain offsets, thinned records, launches cut at random (the exchange slots alternate across the cut)."""
    from yagremcmc_amd import _ffi as ffi, lowering
    ctx = ffi.default_context(0)
    rng = np.random.default_rng(17000 + case + _SEED_SHIFT)
    method = ["mrw", "pcn"][case % 2]
    d = int(rng.integers(257, 1100))
    K = int(rng.choice([1, 16, 40, 130, 600, 2048]))
    n = int(rng.integers(2, 15))
    m_cap = max(1, min(2100, int(2.5e9 / (K * n * d))))
    m = int(rng.integers(1, m_cap + 1)) if case % 4 else min(m_cap, int(rng.integers(500, 2100)))
    seed, chain0 = int(rng.integers(0, 2**62)), int(rng.choice([0, 9, 2**31 + 7]))
    cut = int(rng.integers(0, n + 1))
    g_force = rng.choice(["", "2", "3", "7", "13"])
    if g_force:
        monkeypatch.setenv("YMC_WIDE_SPLIT_G", str(g_force))
    if rng.random() < 0.4:
        monkeypatch.setenv("YMC_WIDE_SPLIT_KERNEL", "stream")
    if rng.random() < 0.5:
        monkeypatch.setenv("YMC_WIDE_SPLIT_W", "8")
    x0 = rng.normal(size=(K, d)) * 0.3
    sigma = rng.uniform(0.01, 0.03, size=d)
    low, p = _linear_pair(ffi, d, m, int(rng.integers(1, 4)), rng)
    stride = int(rng.choice([1, 1, 3]))
    flags = ffi.RECORD_TRAJECTORY | ffi.RECORD_LOGPOST | ((ffi.RECORD_ACCEPT | ffi.RECORD_MOMENTS) if stride == 1 else 0)
    if method == "pcn":
        low.set_array("prior_mean", np.zeros(d))
        low.desc.likelihood_only = 1
        p = dict(p, prior_mean=np.zeros(d), likelihood_only=1)
        sigma = 1.0 / np.sqrt(p["prior_precision"])
        h = float(rng.uniform(0.005, 0.05))
        prob = lowering.make_problem(ctx, ffi.METHOD_PCN, low, sigma, pcn_step_size=h)
        e = o.Ensemble(p, sigma, x0, seed, chain0=chain0, pcn_step_size=h)
    else:
        prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, sigma)
        e = o.Ensemble(p, sigma, x0, seed, chain0=chain0)
    run = ffi.Run(prob, seed, K, x0, n, flags, chain0, record_stride=stride)
    run.advance(cut)
    run.advance(n - cut)
    out = e.advance(n)
    rows = slice(stride - 1, None, stride)
    assert np.array_equal(run.trajectory(), out["traj"][rows]) and np.array_equal(run.logpost(), out["logpost"][rows])
    s_ = run.summary()
    if stride == 1:
        assert np.array_equal(run.accept(), out["acc"])
        assert np.array_equal(s_["wf_mean"], e.wf_mean) and np.array_equal(s_["wf_m2"], e.wf_m2)
    assert np.array_equal(s_["accepted"], e.acc_count) and np.array_equal(s_["status"], e.status)
    x, lp = run.state()
    assert np.array_equal(x, e.x) and np.array_equal(lp, e.lp)
