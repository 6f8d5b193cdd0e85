"""The boundary is a C ABI: a plain-C99 client compiled with gcc against include/ymc.h drives the
same ensemble as the ctypes layer and must produce the same bytes."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fnv1a(data, h):
    # vectorised FNV-1a is not possible; hash in chunks with Python ints (small inputs only)
    for b in data:
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def _build(tmp_path):
    exe = str(tmp_path / "ymc_c_smoke")
    libdir = os.path.join(ROOT, "yagre-mcmc_amd")
    cmd = ["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_abi", "smoke.c"), "-o", exe, "-L", libdir, "-lymc", f"-Wl,-rpath,{libdir}"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_header_is_plain_c(tmp_path):
    """include/ymc.h compiles as strict C99 and as C++ (no HIP, no torch types in the signatures)."""
    src = tmp_path / "inc.c"
    src.write_text('#include "ymc.h"\nint main(void) { return YMC_ABI_VERSION > 0 ? 0 : 1; }\n')
    for cc, std in (("gcc", "-std=c99"), ("g++", "-std=c++11")):
        subprocess.run([cc, std, "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-x",
                        "c" if cc == "gcc" else "c++", "-I", os.path.join(ROOT, "include"), str(src)],
                       check=True, capture_output=True, text=True)


@pytest.mark.gpu
def test_c_client_equals_ctypes_client(tmp_path):
    from yagremcmc_amd import _ffi, lowering
    exe = _build(tmp_path)
    K, T = 512, 60
    g = np.load(os.path.join(ROOT, "tests", "golden", "f2_mrw_2d.npz"))
    U, c = g["whiten"], float(g["log_norm"])
    args = [exe, str(K), str(T)] + [repr(float(v)) for v in U.reshape(-1)] + [repr(c)]
    out = subprocess.run(args, check=True, capture_output=True, text=True).stdout.split()
    low = lowering.Lowered(_ffi.GAUSS_WHITENED, 2)
    low.set_array("mean", np.array([1.0, 1.5]))
    low.set_array("whiten", U)
    low.desc.log_norm = c
    ctx = _ffi.default_context(0)
    prob = lowering.make_problem(ctx, _ffi.METHOD_MRW, low, np.ones(2))
    run = _ffi.Run(prob, 20250103, K, np.array([[-8.0, -7.0]]), T,
                   _ffi.RECORD_TRAJECTORY | _ffi.RECORD_ACCEPT | _ffi.RECORD_MOMENTS)
    run.advance(T)
    run.sync()
    h = _fnv1a(run.trajectory().tobytes(), 14695981039346656037)
    h = _fnv1a(run.accept().tobytes(), h)
    assert out[0] == f"{h:016x}"
    assert float(out[1]) == run.summary()["accepted"].sum() / (K * T)


@pytest.mark.gpu
def test_handles_may_be_destroyed_in_any_order():
    """A garbage-collecting host gives no destruction order: a context released before its problems and
    runs must stay usable by them and be freed by the last one (no use-after-free, no abort at exit)."""
    import ctypes as C
    from yagremcmc_amd import _ffi, lowering
    lib = _ffi.lib()
    ctx = _ffi.Context(0)
    low = lowering.Lowered(_ffi.GAUSS_WHITENED, 2)
    low.set_array("mean", np.zeros(2))
    low.set_array("whiten", np.eye(2))
    prob = lowering.make_problem(ctx, _ffi.METHOD_MRW, low, np.ones(2))
    run = _ffi.Run(prob, 1, 64, np.zeros((1, 2)), 10, _ffi.RECORD_TRAJECTORY | _ffi.RECORD_ACCEPT)
    raw_ctx, raw_prob, raw_run = ctx.raw, prob.raw, run.raw
    for h in (ctx, prob, run):   # take the raw handles over from the Python wrappers
        h.raw = None
    assert lib.ymc_ctx_destroy(raw_ctx) == 0        # released, still alive: the problem uses it
    assert lib.ymc_problem_destroy(raw_prob) == 0   # released, still alive: the run uses it
    assert lib.ymc_run_advance(raw_run, C.c_uint64(10)) == 0 and lib.ymc_run_sync(raw_run) == 0
    traj = np.empty((10, 64, 2))
    assert lib.ymc_run_fetch_trajectory(raw_run, C.c_uint64(0), C.c_uint64(10), traj.ctypes.data_as(C.c_void_p),
                                        C.c_size_t(traj.nbytes)) == 0
    assert np.isfinite(traj).all()
    assert lib.ymc_run_destroy(raw_run) == 0        # frees the run, then the problem, then the context


@pytest.mark.gpu
def test_invalid_descriptions_are_refused_with_messages():
    """Every misconfiguration the new samplers can meet returns a negative status and a message (no abort)."""
    from yagremcmc_amd import _ffi, lowering
    ctx = _ffi.default_context(0)

    def gauss(d=2):
        low = lowering.Lowered(_ffi.GAUSS_WHITENED, d)
        low.set_array("mean", np.zeros(d))
        low.set_array("whiten", np.eye(d))
        return low

    def linear(d=2, m=2, nd=3):
        low = lowering.Lowered(_ffi.POSTERIOR_LINEAR, d)
        low.desc.n_data, low.desc.data_dim = nd, m
        for k, v in dict(data=np.zeros((nd, m)), noise_precision=np.ones(m), prior_mean=np.zeros(d),
                         prior_precision=np.ones(d), fwd_matrix=np.ones((m, d)), fwd_shift=np.zeros(m)).items():
            low.set_array(k, v)
        return low

    class Pre:
        def __init__(self, low):
            self.low = low

    real = lowering.lower_density
    lowering.lower_density = lambda d, rk4_steps=None: d.low if isinstance(d, Pre) else real(d, rk4_steps)
    try:
        with pytest.raises(lowering.UnsupportedPlugin, match="nested levels"):     # 16 surrogates = 9 levels
            lowering.make_nested_problem(ctx, Pre(gauss()), [Pre(gauss())] * 16, [1] * 16, np.ones(2))
        with pytest.warns(RuntimeWarning, match="degenerates"):                    # 14 surrogates = 8 levels
            lowering.make_nested_problem(ctx, Pre(gauss()), [Pre(gauss())] * 14, [1] * 14, np.ones(2))
        with pytest.raises(_ffi.YmcError, match="Subchain lengths"):
            lowering.make_nested_problem(ctx, Pre(gauss()), [Pre(gauss())] * 2, [1, 0], np.ones(2))
        with pytest.raises(_ffi.YmcError, match="kinds"):
            lowering.make_nested_problem(ctx, Pre(gauss()), [Pre(gauss()), Pre(linear())], [2, 2], np.ones(2))
        with pytest.raises(_ffi.YmcError, match="dimension"):
            lowering.make_nested_problem(ctx, Pre(gauss()), [Pre(gauss(3)), Pre(gauss())], [2, 2], np.ones(2))
    finally:
        lowering.lower_density = real
    with pytest.raises(_ffi.YmcError, match="Smallest senisible data size"):
        lowering.make_problem(ctx, _ffi.METHOD_AEM2, linear(), np.ones(2), surrogate=linear(), sub_chain_length=3, aem=(1, 0))
    with pytest.raises(_ffi.YmcError, match="Bayesian model"):
        lowering.make_problem(ctx, _ffi.METHOD_AEM2, gauss(), np.ones(2), surrogate=gauss(), sub_chain_length=3, aem=(5, 0))
    with pytest.raises(_ffi.YmcError, match="output dimension"):
        lowering.make_problem(ctx, _ffi.METHOD_AEM2, linear(m=2), np.ones(2), surrogate=linear(m=3), sub_chain_length=3,
                              aem=(5, 0))
    with pytest.raises(lowering.UnsupportedPlugin):
        lowering.make_problem(ctx, _ffi.METHOD_MLDA2, gauss(), np.ones(2), surrogate=gauss(), sub_chain_length=2,
                              chol=np.eye(2))
    # a surrogate whose forward map is too wide for the lane kernels takes BOTH levels to the matrix-core
    # kernel (it must never be evaluated truncated next to a lane-kernel target); there the adaptive error
    # model takes its data rows eight to a pass over the forward map, as many as the LDS holds
    lowering.make_problem(ctx, _ffi.METHOD_MLDA2, linear(m=2), np.ones(2), surrogate=linear(m=12), sub_chain_length=2)
    lowering.make_problem(ctx, _ffi.METHOD_AEM2, linear(m=12), np.ones(2), surrogate=linear(m=12), sub_chain_length=2, aem=(5, 0))
    lowering.make_problem(ctx, _ffi.METHOD_AEM2, linear(m=12, nd=9), np.ones(2), surrogate=linear(m=12, nd=9),
                          sub_chain_length=2, aem=(5, 0))
    with pytest.raises(_ffi.YmcError, match="not finite"):
        lowering.make_problem(ctx, _ffi.METHOD_MRW, gauss(), np.ones(2), chol=np.array([[1.0, 0.0], [np.nan, 1.0]]))
    # fetches that do not belong to the method
    prob = lowering.make_problem(ctx, _ffi.METHOD_MRW, gauss(), np.ones(2))
    run = _ffi.Run(prob, 1, 8, np.zeros((1, 2)), 4, _ffi.RECORD_ACCEPT)
    run.advance(4)
    with pytest.raises(_ffi.YmcError, match="not an adaptive-error-model run"):
        run.error_model()
    with pytest.raises(_ffi.YmcError, match="not a two-level run"):
        run.sub_accept()
    with pytest.raises(_ffi.YmcError, match="4 of 4 rows"):
        run.advance(1)
    # a burn-in beyond the record: a negative row count must not wrap around into a huge allocation
    full = _ffi.Run(prob, 1, 8, np.zeros((1, 2)), 6, _ffi.RECORD_TRAJECTORY | _ffi.RECORD_ACCEPT)
    full.advance(6)
    with pytest.raises(ValueError, match="burn-in"):
        full.iat(first=10)
    with pytest.raises(ValueError, match="recorded"):
        full.trajectory(first=10)
    out = np.zeros(8, dtype=np.int32)
    rc = _ffi.lib().ymc_run_iat(full.raw, 10, (1 << 64) - 4, 0, 5.0, _ffi.vptr(out))   # what -4 looks like through the ABI
    assert rc == _ffi.ERR_INVALID and b"burn-in" in _ffi.lib().ymc_last_error()
    buf = np.zeros((2, 8, 2))
    assert _ffi.lib().ymc_run_fetch_trajectory(full.raw, 5, (1 << 64) - 3, _ffi.vptr(buf), buf.nbytes) == _ffi.ERR_INVALID


@pytest.mark.gpu
def test_device_memory_returns_when_problems_and_runs_are_destroyed():
    """Every sampler's problem and run, created, advanced and destroyed forty times over: the device's free memory
    comes back (hipMemGetInfo before and after; the context keeps a few scratch buffers, bounded)."""
    import ctypes as C
    from yagremcmc_amd import _ffi, lowering
    hip = C.CDLL("libamdhip64.so")

    def free_bytes():
        free, total = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return free.value

    ctx = _ffi.default_context(0)
    rng = np.random.default_rng(3)

    def lin(d, m, nd=2):
        low = lowering.Lowered(_ffi.POSTERIOR_LINEAR, d)
        low.desc.n_data, low.desc.data_dim = nd, m
        for k, v in dict(data=rng.normal(size=(nd, m)), noise_precision=np.ones(m), prior_mean=np.zeros(d), prior_precision=np.ones(d),
                         fwd_matrix=rng.normal(size=(m, d)) / np.sqrt(d), fwd_shift=np.zeros(m)).items():
            low.set_array(k, v)
        return low

    def gauss(d=2):
        low = lowering.Lowered(_ffi.GAUSS_WHITENED, d)
        low.set_array("mean", np.zeros(d))
        low.set_array("whiten", np.eye(d))
        return low

    class Pre:
        def __init__(self, low):
            self.low = low

    real = lowering.lower_density
    lowering.lower_density = lambda dd, rk4_steps=None: dd.low if isinstance(dd, Pre) else real(dd, rk4_steps)
    flags = _ffi.RECORD_TRAJECTORY | _ffi.RECORD_ACCEPT | _ffi.RECORD_MOMENTS
    K = 4096

    def cycle():
        made = [
            (lowering.make_problem(ctx, _ffi.METHOD_MRW, gauss(), np.ones(2)), 2),
            (lowering.make_problem(ctx, _ffi.METHOD_MRW, lin(64, 64), np.full(64, 0.05)), 64),
            (lowering.make_problem(ctx, _ffi.METHOD_MRW, lin(200, 300), np.full(200, 0.03)), 200),
            (lowering.make_problem(ctx, _ffi.METHOD_MRW, lin(300, 24), np.full(300, 0.03)), 300),        # the wide path
            (lowering.make_problem(ctx, _ffi.METHOD_MLDA2, lin(128, 200), np.full(128, 0.03), surrogate=lin(128, 200), sub_chain_length=2), 128),
            (lowering.make_problem(ctx, _ffi.METHOD_AEM2, lin(32, 40, 11), np.full(32, 0.05), surrogate=lin(32, 40, 11), sub_chain_length=2,
                                   aem=(4, 1)), 32),
            (lowering.make_nested_problem(ctx, Pre(lin(48, 20)), [Pre(lin(48, 30)), Pre(lin(48, 16))], [2, 2], np.full(48, 0.05)), 48),
            (lowering.make_nested_problem(ctx, Pre(gauss()), [Pre(gauss())] * 10, [1] * 10, np.ones(2)), 2),
        ]
        for prob, d in made:
            run = _ffi.Run(prob, 1, K, np.zeros((K, d)), 6, flags)
            run.advance(6)
            run.trajectory()
            run.summary()
            run.close()
            prob.close()

    import warnings
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            cycle()                                  # the context's own scratch buffers are in place after one cycle
            before = free_bytes()
            for _ in range(40):
                cycle()
            after = free_bytes()
    finally:
        lowering.lower_density = real
    assert before - after < 64 << 20, f"{(before - after) / 2**20:.1f} MiB of device memory did not come back"
    # a record beyond the card's memory: refused with the runtime's message, and what had been allocated is released
    prob = lowering.make_problem(ctx, _ffi.METHOD_MRW, gauss(), np.ones(2))
    with pytest.raises(_ffi.YmcError, match="(?i)memory") as err:
        _ffi.Run(prob, 1, 1 << 24, np.zeros((1, 2)), 2000, _ffi.RECORD_TRAJECTORY | _ffi.RECORD_ACCEPT)
    assert err.value.code == _ffi.ERR_DEVICE
    assert after - free_bytes() < 64 << 20
    run = _ffi.Run(prob, 1, 1024, np.zeros((1, 2)), 10, _ffi.RECORD_TRAJECTORY)      # the context is still usable
    run.advance(10)
    assert np.isfinite(run.trajectory()).all()
