"""Thinned record (ymc_run_desc.record_stride) and the egress path (ymc_run_egress_*): a stride-s record is
rows s-1::s of the full record of the same run, for every kernel family and however the run is cut into
launches; rows staged through the page-locked double buffer equal the rows fetched the slow way."""
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


def _gauss(ffi, mean, whiten, log_norm):
    from yagremcmc_amd import lowering
    low = lowering.Lowered(ffi.GAUSS_WHITENED, int(np.size(mean)))
    low.set_array("mean", np.asarray(mean, dtype=float).reshape(-1))
    low.set_array("whiten", np.asarray(whiten, dtype=float))
    low.desc.log_norm = float(log_norm)
    return low


def _linear(ffi, d, m, nd, seed):
    from yagremcmc_amd import lowering
    r = np.random.default_rng(seed)
    low = lowering.Lowered(ffi.POSTERIOR_LINEAR, d)
    low.desc.n_data, low.desc.data_dim = nd, m
    for k, v in dict(data=r.standard_normal((nd, m)), noise_precision=r.uniform(0.5, 2, m), prior_mean=r.standard_normal(d) * 0.1,
                     prior_precision=r.uniform(0.2, 1, d), fwd_matrix=r.standard_normal((m, d)) / np.sqrt(d),
                     fwd_shift=r.standard_normal(m)).items():
        low.set_array(k, v)
    return low


def _problems(ffi, ctx, golden):
    """(name, problem, x0, flags that make sense with a stride)"""
    from yagremcmc_amd import lowering
    g, g5 = golden("f2_mrw_2d.npz"), golden("f5_mlda2_2d.npz")
    rng = np.random.default_rng(0)
    TLM = ffi.RECORD_TRAJECTORY | ffi.RECORD_LOGPOST | ffi.RECORD_MOMENTS
    out = []
    low = _gauss(ffi, g["mean"], g["whiten"], g["log_norm"])
    out.append(("mrw", lowering.make_problem(ctx, ffi.METHOD_MRW, low, g["sigma"]), rng.standard_normal((70, 2)) - 3, TLM))
    fine = _gauss(ffi, g5["fine_mean"], g5["fine_whiten"], g5["fine_log_norm"])
    coarse = _gauss(ffi, g5["coarse_mean"], g5["coarse_whiten"], g5["coarse_log_norm"])
    out.append(("mlda2", lowering.make_problem(ctx, ffi.METHOD_MLDA2, fine, g5["sigma"], surrogate=coarse, sub_chain_length=4),
                rng.standard_normal((70, 2)) - 3, TLM))
    out.append(("pcn", lowering.make_problem(ctx, ffi.METHOD_PCN, _lik_only(_linear(ffi, 3, 4, 5, 1)), np.full(3, 1.5), pcn_step_size=0.1),
                rng.standard_normal((40, 3)), TLM))
    out.append(("tile", lowering.make_problem(ctx, ffi.METHOD_MRW, _linear(ffi, 20, 24, 3, 2), np.full(20, 0.05)),
                rng.standard_normal((50, 20)) * 0.1, ffi.RECORD_TRAJECTORY | ffi.RECORD_LOGPOST))
    return out


def _lik_only(low):
    low.desc.likelihood_only = 1
    return low


@pytest.mark.parametrize("stride", [2, 3, 7])
def test_thinned_record_is_every_sth_row_of_the_full_record(ffi, ctx, golden, stride):
    T = 45
    for name, prob, x0, flags in _problems(ffi, ctx, golden):
        K = x0.shape[0]
        full = ffi.Run(prob, 9, K, x0, T, flags)
        full.advance(T)
        thin = ffi.Run(prob, 9, K, x0, T // stride, flags, record_stride=stride)
        for n in (1, 4, stride, 2 * stride + 1, 11):          # launches that start anywhere inside a stride
            thin.advance(n)
        thin.advance(T - thin.steps_done()[0])
        assert thin.steps_done() == (T, T // stride), name
        assert np.array_equal(thin.trajectory(), full.trajectory()[stride - 1::stride]), name
        assert np.array_equal(thin.logpost(), full.logpost()[stride - 1::stride]), name
        sf, st = full.summary(), thin.summary()
        for key in ("accepted", "wf_n", "wf_mean", "wf_m2", "status"):        # these still see every transition
            assert np.array_equal(sf[key], st[key]), (name, key)
        assert np.array_equal(full.state()[0], thin.state()[0]), name
        with pytest.raises(ffi.YmcError, match="rows"):
            thin.advance(stride)                                                # the thinned record is full


@pytest.mark.parametrize("split", ["0", "1"])
@pytest.mark.parametrize("stride", [2, 5])
def test_thinned_record_specialised_random_walk_kernels(ffi, ctx, golden, monkeypatch, split, stride):
    """The builders' thinned default (trajectory, with or without moments, no decision record) runs compile-time
    record masks in the one-lane-per-chain kernel and in the helper-wave kernel: same rows as the full record."""
    monkeypatch.setenv("YMC_SPLIT", split)
    _, prob, x0, _ = _problems(ffi, ctx, golden)[0]
    K, T = x0.shape[0], 53
    full = ffi.Run(prob, 9, K, x0, T, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT | ffi.RECORD_MOMENTS)
    full.advance(T)
    for flags in (ffi.RECORD_TRAJECTORY, ffi.RECORD_TRAJECTORY | ffi.RECORD_MOMENTS):
        thin = ffi.Run(prob, 9, K, x0, T // stride, flags, record_stride=stride)
        for n in (1, 6, 7, 12, 2):
            thin.advance(n)
        thin.advance(T - thin.steps_done()[0])
        assert thin.steps_done() == (T, T // stride)
        assert np.array_equal(thin.trajectory(), full.trajectory()[stride - 1::stride])
        sf, st = full.summary(), thin.summary()
        assert np.array_equal(sf["accepted"], st["accepted"]) and np.array_equal(full.state()[0], thin.state()[0])
        if flags & ffi.RECORD_MOMENTS:
            assert np.array_equal(sf["wf_mean"], st["wf_mean"]) and np.array_equal(sf["wf_m2"], st["wf_m2"])


@pytest.mark.parametrize("stride", [2, 5])
def test_thinned_record_specialised_two_level_kernels(ffi, ctx, golden, stride):
    """Two-level with a compile-time sub-chain length (S = 4) and the thinned default record."""
    _, prob, x0, _ = _problems(ffi, ctx, golden)[1]
    K, T = x0.shape[0], 37
    full = ffi.Run(prob, 9, K, x0, T, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT | ffi.RECORD_MOMENTS)
    full.advance(T)
    for flags in (ffi.RECORD_TRAJECTORY, ffi.RECORD_TRAJECTORY | ffi.RECORD_MOMENTS):
        thin = ffi.Run(prob, 9, K, x0, T // stride, flags, record_stride=stride)
        for n in (1, 2, 3, 8):
            thin.advance(n)
        thin.advance(T - thin.steps_done()[0])
        assert np.array_equal(thin.trajectory(), full.trajectory()[stride - 1::stride])
        sf, st = full.summary(), thin.summary()
        assert np.array_equal(sf["accepted"], st["accepted"]) and np.array_equal(full.state()[1], thin.state()[1])
        if flags & ffi.RECORD_MOMENTS:
            assert np.array_equal(sf["wf_mean"], st["wf_mean"])


def test_thinned_record_misuse_is_refused(ffi, ctx, golden):
    name, prob, x0, _ = _problems(ffi, ctx, golden)[0]
    with pytest.raises(ffi.YmcError, match="full-rate"):
        ffi.Run(prob, 1, 8, x0[:8], 10, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT, record_stride=2)
    tile = _problems(ffi, ctx, golden)[3]
    with pytest.raises(ffi.YmcError, match="FULL record"):
        ffi.Run(tile[1], 1, 8, tile[2][:8], 10, ffi.RECORD_TRAJECTORY | ffi.RECORD_MOMENTS, record_stride=2)


def test_builder_record_stride(golden):
    """recordStride on the builders: the chain holds x0 and every s-th state, rates and moments are unchanged."""
    from yagremcmc_amd import FullDiagnostics, MRWBuilder
    from yagremcmc_amd.parameter.vector import ParameterVector
    from yagremcmc_amd.statistics.covariance import IIDCovarianceMatrix
    from yagremcmc_amd.test.testSetup import GaussianTargetDensity2d
    target = GaussianTargetDensity2d(ParameterVector(np.array([1.0, 1.5])), np.array([[2.4, -0.5], [-0.5, 0.7]]))

    def sampler(stride):
        b = MRWBuilder()
        b.explicitTarget, b.proposalCovariance, b.diagnostics = target, IIDCovarianceMatrix(2, 1.0), FullDiagnostics()
        b.nChains, b.seed, b.recordStride = 32, 3, stride
        mc = b.build_method()
        mc.run(61, ParameterVector(np.array([-8.0, -7.0])), verbose=False)
        return mc

    full, thin = sampler(1), sampler(5)
    assert np.array_equal(thin.chain.ensemble_trajectory, full.chain.ensemble_trajectory[::5])
    assert thin.diagnostics.global_acceptance_rate() == full.diagnostics.global_acceptance_rate()
    assert np.array_equal(thin.diagnostics.mean(), full.diagnostics.mean())
    with pytest.raises(RuntimeError, match="recordStride"):
        thin.accept_decisions


def test_egress_double_buffer_equals_fetch(ffi, ctx, golden):
    """The loop of include/ymc.h: two halves of the record alternate between 'being written' and 'on its way to
    the host'; every staged block equals the block fetched synchronously from an identical run."""
    from yagremcmc_amd import lowering
    g = golden("f2_mrw_2d.npz")
    low = _gauss(ffi, g["mean"], g["whiten"], g["log_norm"])
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, g["sigma"])
    K, n, blocks = 4096, 50, 7
    x0 = np.tile([-8.0, -7.0], (K, 1))
    flags = ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT | ffi.RECORD_LOGPOST | ffi.RECORD_MOMENTS
    ref = ffi.Run(prob, 21, K, x0, n * blocks, flags)
    ref.advance(n * blocks)
    want_t, want_a, want_l = ref.trajectory(), ref.accept(), ref.logpost()
    run = ffi.Run(prob, 21, K, x0, 2 * n, flags)
    what = ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT | ffi.RECORD_LOGPOST
    got = []
    for b in range(blocks):
        half = b & 1
        if half == 0 and b > 0:
            run.rewind_record()
        run.advance(n)                       # writes rows [half n, half n + n); waits on the device for that half's copy
        run.egress_begin(half, half * n, n, what)
        if b >= 1:
            v = run.egress_wait(1 - half)
            got.append({k: np.array(a) for k, a in v.items() if k != "first"})
    v = run.egress_wait((blocks - 1) & 1)
    got.append({k: np.array(a) for k, a in v.items() if k != "first"})
    for b, blk in enumerate(got):
        assert np.array_equal(blk["trajectory"], want_t[b * n:(b + 1) * n]), b
        assert np.array_equal(blk["accept"], want_a[b * n:(b + 1) * n]), b
        assert np.array_equal(blk["logpost"], want_l[b * n:(b + 1) * n]), b
    assert np.array_equal(run.summary()["wf_mean"], ref.summary()["wf_mean"])
    # misuse
    with pytest.raises(ffi.YmcError, match="slot"):
        run.egress_begin(2, 0, 1)
    # a view points into the run's page-locked staging memory: it keeps the run alive, and the run refuses to
    # re-allocate (grow) the slot or to be closed under it
    with pytest.raises(RuntimeError, match="must grow"):
        run.egress_begin(0, 0, 2 * n)
    with pytest.raises(RuntimeError, match="live egress views"):
        run.close()
    keep = v["trajectory"][1:3]          # a derived view is enough to hold the memory
    del v
    with pytest.raises(RuntimeError, match="live egress views"):
        run.close()
    assert np.array_equal(keep, want_t[(blocks - 1) * n + 1:(blocks - 1) * n + 3])
    del keep
    with pytest.raises(ffi.YmcError, match="recorded"):
        run.egress_begin(0, 0, 10 * n)
    bare = ffi.Run(prob, 1, 8, x0[:8], 4, ffi.RECORD_TRAJECTORY)
    bare.advance(4)
    with pytest.raises(ffi.YmcError, match="does not record"):
        bare.egress_begin(0, 0, 4, ffi.RECORD_ACCEPT)
    with pytest.raises(ffi.YmcError, match="nothing was begun"):
        bare.egress_wait(1)


def test_egress_of_the_tile_kernels_record(ffi, ctx):
    """d > 8: the record is in the matrix-core kernel's tiled layout and is untiled on the device on its way out."""
    from yagremcmc_amd import lowering
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW, _linear(ffi, 20, 24, 3, 2), np.full(20, 0.05))
    x0 = np.random.default_rng(5).standard_normal((100, 20)) * 0.1
    run = ffi.Run(prob, 2, 100, x0, 12, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT)
    run.advance(12)
    run.egress_begin(0, 3, 8, ffi.RECORD_TRAJECTORY)
    v = run.egress_wait(0)
    assert v["first"] == 3 and np.array_equal(v["trajectory"], run.trajectory(3, 8))
