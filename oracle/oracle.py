"""ctypes front end of the CPU checker (oracle/ymc_oracle.c).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this module.  The product (yagre-mcmc_amd/) never
does; its MH path fails loudly when the HIP library is missing.

A "problem" here is a plain dict (see `gauss_whitened`, `gauss_iso1d`,
`posterior_linear`, `posterior_lv`), deliberately independent of the product's
ctypes structures.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# YMC_ORACLE_LIB: another build of the same source (the sanitizer build of tests/test_sanitizers.py)
_LIB_PATH = os.environ.get("YMC_ORACLE_LIB") or os.path.join(_HERE, "libymc_oracle.so")

GAUSS_WHITENED, GAUSS_ISO1D, POST_LINEAR, POST_LV, POST_LOGISTIC = 1, 2, 3, 4, 5
EQ_EXACT, EQ_ISCLOSE = 0, 1
STATUS_NONFINITE = 1

_dp = C.POINTER(C.c_double)


class _Density(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("dim", C.c_int32),
        ("mean", _dp), ("whiten", _dp), ("log_norm", C.c_double),
        ("variance", C.c_double),
        ("n_data", C.c_int32), ("data_dim", C.c_int32),
        ("data", _dp), ("noise_precision", _dp), ("prior_mean", _dp), ("prior_precision", _dp),
        ("fwd_matrix", _dp), ("fwd_shift", _dp),
        ("design", _dp), ("t_final", C.c_double), ("alpha", C.c_double), ("gamma", C.c_double),
        ("rk4_steps", C.c_int32), ("likelihood_only", C.c_int32),
        ("lin_dbar", _dp), ("lin_nprec", _dp), ("lin_const", C.c_double),
        ("noise_linv", _dp), ("prior_linv", _dp),
        ("lv_rtol", C.c_double), ("lv_atol", C.c_double),
    ]


def build(force=False):
    """Compile oracle/libymc_oracle.so with gcc (recipe: oracle/Makefile)."""
    if force and os.path.exists(_LIB_PATH):
        os.remove(_LIB_PATH)
    subprocess.check_call(["make", "-s", "-C", _HERE, "libymc_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.ymo_max_threads.restype = C.c_int32
        _lib.ymo_forward.restype = C.c_int
    return _lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _vp(a):
    return C.c_void_p(a.ctypes.data) if a is not None else None


# ---------------------------------------------------------------------------
# problem dictionaries
# ---------------------------------------------------------------------------

def gauss_whitened(mean, whiten, log_norm):
    mean = _f64(mean).reshape(-1)
    return dict(kind=GAUSS_WHITENED, dim=mean.size, mean=mean,
                whiten=_f64(whiten).reshape(mean.size, mean.size), log_norm=float(log_norm))


def gauss_from_cov(mean, cov):
    """Whitening data of scipy.stats.multivariate_normal(mean, cov) the way
    scipy's CovViaPSD forms it (eigh; U = V / sqrt(w); log_pdet = sum log w)."""
    mean = _f64(mean).reshape(-1)
    cov = _f64(cov).reshape(mean.size, mean.size)
    w, v = np.linalg.eigh(cov)
    whiten = v * np.sqrt(1.0 / w)
    log_norm = mean.size * np.log(2.0 * np.pi) + np.sum(np.log(w))
    return gauss_whitened(mean, whiten, log_norm)


def gauss_iso1d(mean, variance):
    return dict(kind=GAUSS_ISO1D, dim=1, mean=_f64([mean]).reshape(-1), variance=float(variance))


def _posterior_common(kind, dim, data, noise_precision, prior_mean, prior_precision):
    data = _f64(data)
    assert data.ndim == 2
    return dict(kind=kind, dim=int(dim), data=data, n_data=data.shape[0], data_dim=data.shape[1],
                noise_precision=_f64(noise_precision).reshape(-1),
                prior_mean=_f64(prior_mean).reshape(-1),
                prior_precision=_f64(prior_precision).reshape(-1))


LANE_DMAX = LANE_MMAX = 8   # include/ymc.h: the lane kernels' largest parameter / output dimension


def dense_covariances(p, noise_cov=None, prior_cov=None, whiten=None):
    """Replace the diagonal noise / prior covariance of a posterior by a dense one (DenseCovarianceMatrix,
    covariance.py:69-94): the inverse of its lower Cholesky factor (ymc_whitened_norm_sq).  Beyond the lane
    kernels' sizes (`whiten`: None = by size, as the product's lowering decides for a single density) a linear
    problem with a dense NOISE covariance is whitened instead: A, b and the data rows are multiplied by L^-1
    (ymc_lower_times) and the noise becomes the identity; a dense PRIOR covariance becomes d more outputs of the
    forward map there (rows Lp^-1, shift 0, "data" Lp^-1 mu, weight 1; the diagonal prior is switched off).
    `forward` still returns A x + b."""
    p = dict(p)
    from scipy.linalg import solve_triangular

    def inverse_factor(cov):
        L = np.linalg.cholesky(_f64(cov))
        return np.ascontiguousarray(np.tril(solve_triangular(L, np.eye(L.shape[0]), lower=True)))

    big = p["kind"] == POST_LINEAR and (p["dim"] > LANE_DMAX or p["data_dim"] > LANE_MMAX)
    big = big if whiten is None else whiten

    def times(linv, src, cols):
        n = linv.shape[0]
        src = np.ascontiguousarray(_f64(src).reshape(n, cols))
        out = np.empty_like(src)
        lib().ymo_lower_times(_ptr(linv), C.c_int32(n), _ptr(src), C.c_int32(cols), _ptr(out))
        return out

    if big and (noise_cov is not None or prior_cov is not None):
        assert p["kind"] == POST_LINEAR
        p["fwd_matrix_raw"], p["fwd_shift_raw"], p["data_dim_raw"] = p["fwd_matrix"], p["fwd_shift"], p["data_dim"]
    if noise_cov is not None:
        linv = inverse_factor(noise_cov)
        if big:
            m = p["data_dim"]
            p["fwd_matrix"] = times(linv, p["fwd_matrix"], p["dim"])
            p["fwd_shift"] = times(linv, p["fwd_shift"], 1).reshape(-1)
            p["data"] = np.stack([times(linv, row, 1).reshape(-1) for row in _f64(p["data"])])
            p["noise_precision"] = np.ones(m)
        else:
            p["noise_linv"] = linv
    if prior_cov is not None:
        linv = inverse_factor(prior_cov)
        if big:
            d, nd = p["dim"], p["n_data"]
            c = times(linv, p["prior_mean"], 1).reshape(-1)
            p["fwd_matrix"] = np.vstack([_f64(p["fwd_matrix"]), linv])
            p["fwd_shift"] = np.concatenate([_f64(p["fwd_shift"]), np.zeros(d)])
            p["data"] = np.hstack([_f64(p["data"]), np.tile(c, (nd, 1))])
            p["noise_precision"] = np.concatenate([_f64(p["noise_precision"]), np.full(d, 1.0 / nd)])
            p["prior_precision"] = np.zeros(d)
            p["data_dim"] = p["data_dim"] + d
            p["lin_tail"] = (c, np.ones(d))   # the summary of the pseudo-observations, set exactly (_lower)
        else:
            p["prior_linv"] = linv
    return p


def posterior_linear(A, b, data, noise_precision, prior_mean, prior_precision):
    A = _f64(A)
    p = _posterior_common(POST_LINEAR, A.shape[1], data, noise_precision, prior_mean, prior_precision)
    assert A.shape[0] == p["data_dim"]
    p.update(fwd_matrix=A, fwd_shift=_f64(b).reshape(-1))
    return p


def posterior_lv(design, t_final, alpha, gamma, rk4_steps, data, noise_precision, prior_mean,
                 prior_precision, rtol=0.0, atol=1e-6):
    """rtol > 0: the adaptive integrator (ymc_lv_dp45_solve) instead of rk4_steps steps of RK4."""
    design = _f64(design)
    p = _posterior_common(POST_LV, 2, data, noise_precision, prior_mean, prior_precision)
    assert design.shape == (p["n_data"], 2) and p["data_dim"] == 2
    p.update(design=design, t_final=float(t_final), alpha=float(alpha), gamma=float(gamma),
             rk4_steps=int(rk4_steps), lv_rtol=float(rtol), lv_atol=float(atol) if rtol > 0 else 0.0)
    return p


def posterior_logistic(design, t_final, rk4_steps, data, noise_precision, prior_mean, prior_precision):
    """The worked extension kind (csrc/kinds/logistic.h): logistic growth from the initial populations `design`
    (n_data,), theta = (log r, log K), one forward output per design row."""
    design = _f64(design).reshape(-1)
    p = _posterior_common(POST_LOGISTIC, 2, data, noise_precision, prior_mean, prior_precision)
    assert design.shape == (p["n_data"],) and p["data_dim"] == 1
    p.update(design=design, t_final=float(t_final), rk4_steps=int(rk4_steps))
    return p


DMAX, MMAX = 2048, 8192     # YMO_DMAX, YMO_MMAX of ymc_oracle.c (its work arrays live on the stack)


def _lower(p):
    """dict -> (_Density, keepalive)"""
    if p["dim"] > DMAX or p.get("data_dim", 0) > MMAX:
        raise ValueError(f"the checker handles dimensions up to {DMAX} and {MMAX} forward outputs, got {p['dim']} and "
                         f"{p.get('data_dim', 0)}")
    d = _Density()
    d.kind, d.dim = p["kind"], p["dim"]
    keep = []
    for name in ("mean", "whiten", "data", "noise_precision", "prior_mean", "prior_precision",
                 "fwd_matrix", "fwd_shift", "design", "noise_linv", "prior_linv"):
        if name in p:
            arr = _f64(p[name])
            keep.append(arr)
            setattr(d, name, _ptr(arr))
    for name in ("log_norm", "variance", "t_final", "alpha", "gamma", "lv_rtol", "lv_atol"):
        if name in p:
            setattr(d, name, p[name])
    for name in ("n_data", "data_dim", "rk4_steps", "likelihood_only"):
        if name in p:
            setattr(d, name, p[name])
    if p["kind"] == POST_LINEAR:
        # the data-only parts of the likelihood (ymc_linear_data_summary), once per lowering
        m = int(p["data_dim"])
        dbar, nprec, cst = np.zeros(m), np.zeros(m), C.c_double(0.0)
        lib().ymo_linear_summary(d.data, d.noise_precision, C.c_int32(int(p["n_data"])), C.c_int32(m), _ptr(dbar),
                                 _ptr(nprec), C.byref(cst))
        if "lin_tail" in p:   # a dense prior as pseudo-observations (dense_covariances): weight 1, "data" c, exactly
            c, w = p["lin_tail"]
            dbar[m - c.size:], nprec[m - c.size:] = c, w
        keep += [dbar, nprec]
        d.lin_dbar, d.lin_nprec, d.lin_const = _ptr(dbar), _ptr(nprec), cst.value
    return d, keep


# ---------------------------------------------------------------------------
# numeric-spec hooks
# ---------------------------------------------------------------------------

def philox(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().ymo_philox(_vp(c), _vp(k), _vp(out))
    return out


def _vec1(fn, x):
    x = _f64(x).reshape(-1)
    y = np.empty_like(x)
    fn(C.c_int64(x.size), _ptr(x), _ptr(y))
    return y


def log(x):
    return _vec1(lib().ymo_log_v, x)


def exp(x):
    return _vec1(lib().ymo_exp_v, x)


def accept_prob(x):
    return _vec1(lib().ymo_accept_prob_v, x)


def div_by_recip(a, b):
    a, b = _f64(a).reshape(-1), _f64(b).reshape(-1)
    q = np.empty_like(a)
    lib().ymo_div_by_recip_v(C.c_int64(a.size), _ptr(a), _ptr(b), _ptr(q))
    return q


def sincospi(x):
    x = _f64(x).reshape(-1)
    s, c = np.empty_like(x), np.empty_like(x)
    lib().ymo_sincospi_v(C.c_int64(x.size), _ptr(x), _ptr(s), _ptr(c))
    return s, c


def rng_fill(seed, chain0, K, t0, nt, d, normals=True, uniforms=True):
    """xi[nt][K][d], u[nt][K] -- the draws of chains chain0..chain0+K-1."""
    xi = np.empty((nt, K, d)) if normals else None
    u = np.empty((nt, K)) if uniforms else None
    lib().ymo_rng_fill(C.c_uint64(seed), C.c_uint64(chain0), C.c_int64(K), C.c_uint64(t0),
                       C.c_int64(nt), C.c_int32(d), _ptr(xi), _ptr(u))
    return xi, u


# ---------------------------------------------------------------------------
# densities
# ---------------------------------------------------------------------------

def logdensity(problem, x):
    d, keep = _lower(problem)
    x = _f64(x).reshape(-1, problem["dim"])
    out = np.empty(x.shape[0])
    status = np.zeros(x.shape[0], dtype=np.uint32)
    lib().ymo_logdensity(C.byref(d), C.c_int64(x.shape[0]), _ptr(x), _ptr(out), _vp(status))
    return out, status


def forward(problem, x):
    if "fwd_matrix_raw" in problem:   # a whitened / extended problem (dense_covariances): the map the user described
        m = problem["data_dim_raw"]
        problem = dict(problem, fwd_matrix=problem["fwd_matrix_raw"], fwd_shift=problem["fwd_shift_raw"], data_dim=m,
                       data=_f64(problem["data"])[:, :m], noise_precision=_f64(problem["noise_precision"])[:m])
        problem.pop("lin_tail", None)
    d, keep = _lower(problem)
    x = _f64(x).reshape(-1, problem["dim"])
    m = problem["n_data"] * 2 if problem["kind"] == POST_LV else problem["n_data"] if problem["kind"] == POST_LOGISTIC \
        else problem["data_dim"]
    out = np.empty((x.shape[0], m))
    bad = lib().ymo_forward(C.byref(d), C.c_int64(x.shape[0]), _ptr(x), _ptr(out))
    if bad < 0:
        raise ValueError("problem has no forward map")
    return out


# ---------------------------------------------------------------------------
# chain ensembles
# ---------------------------------------------------------------------------

class Ensemble:
    """K chains advanced by the oracle; mirrors the state a device run carries."""

    def __init__(self, target, sigma, x0, seed, chain0=0, eq_mode=EQ_EXACT, surrogate=None,
                 sub_chain_length=0, nthreads=1, pcn_step_size=None, chol=None):
        """`pcn_step_size`: run preconditioned Crank-Nicolson instead of the random walk; `target`
        must then carry likelihood_only=1 and `sigma` is the PRIOR's standard deviation."""
        self.pcn_step_size = pcn_step_size
        self.chol = None if chol is None else np.tril(_f64(chol))  # dense proposal covariance: lower Cholesky factor
        self.target, self.surrogate = target, surrogate
        self.S = int(sub_chain_length)
        self.d = target["dim"]
        x0 = _f64(x0)
        if x0.ndim == 1:
            x0 = x0.reshape(1, -1)
        self.K = x0.shape[0]
        self.x = x0.copy()
        self.sigma = _f64(sigma).reshape(-1)
        assert self.sigma.size == self.d and x0.shape[1] == self.d
        self.seed, self.chain0, self.eq_mode = int(seed), int(chain0), int(eq_mode)
        self.nthreads = int(nthreads)
        self.t = 0  # MRW: transitions done; MLDA2: fine steps done
        self.lp, self.status0 = logdensity(target, self.x)
        self.lpc = logdensity(surrogate, self.x)[0] if surrogate is not None else None
        self.acc_count = np.zeros(self.K, dtype=np.uint64)
        self.wf_n = np.zeros(self.K, dtype=np.uint64)
        self.wf_mean = np.zeros((self.K, self.d))
        self.wf_m2 = np.zeros((self.K, self.d))
        self.status = np.zeros(self.K, dtype=np.uint32)

    def advance(self, nsteps, record=True, want_margin=False, want_sub=False, reuse=None):
        """`reuse`: the dict a previous advance of the same length returned; its
        arrays are overwritten (lets a timing run write into pages that are
        already resident)."""
        K, d = self.K, self.d
        if reuse is not None:
            traj, acc, lp_out = reuse["traj"], reuse["acc"], reuse["logpost"]
            assert traj.shape == (nsteps, K, d)
        else:
            traj = np.empty((nsteps, K, d)) if record else None
            acc = np.empty((nsteps, K), dtype=np.uint8) if record else None
            lp_out = np.empty((nsteps, K)) if record else None
        margin = np.empty((nsteps, K)) if want_margin else None
        tgt, keep = _lower(self.target)
        out = dict(traj=traj, acc=acc, logpost=lp_out, margin=margin)
        if self.pcn_step_size is not None:
            lib().ymo_pcn_advance(
                C.byref(tgt), _ptr(self.sigma), C.c_double(self.pcn_step_size), C.c_int32(self.eq_mode),
                C.c_uint64(self.seed), C.c_uint64(self.chain0), C.c_int64(K), C.c_uint64(self.t), C.c_int64(nsteps),
                _ptr(self.x), _ptr(self.lp), _vp(self.acc_count), _vp(self.wf_n), _ptr(self.wf_mean),
                _ptr(self.wf_m2), _vp(self.status), _ptr(traj), _vp(acc), _ptr(lp_out), C.c_int32(self.nthreads))
        elif self.surrogate is None:
            lib().ymo_mrw_advance(
                C.byref(tgt), _ptr(self.sigma), C.c_int32(self.eq_mode), C.c_uint64(self.seed),
                C.c_uint64(self.chain0), C.c_int64(K), C.c_uint64(self.t), C.c_int64(nsteps),
                _ptr(self.x), _ptr(self.lp), _vp(self.acc_count), _vp(self.wf_n),
                _ptr(self.wf_mean), _ptr(self.wf_m2), _vp(self.status), _ptr(traj), _vp(acc),
                _ptr(lp_out), _ptr(margin), C.c_int32(self.nthreads), _ptr(self.chol))
        else:
            sur, keep2 = _lower(self.surrogate)
            sub = np.empty((nsteps, self.S, K), dtype=np.uint8) if want_sub else None
            out["sub_acc"] = sub
            lib().ymo_mlda2_advance(
                C.byref(tgt), C.byref(sur), _ptr(self.sigma), C.c_int32(self.S),
                C.c_int32(self.eq_mode), C.c_uint64(self.seed), C.c_uint64(self.chain0),
                C.c_int64(K), C.c_uint64(self.t), C.c_int64(nsteps), _ptr(self.x), _ptr(self.lp),
                _ptr(self.lpc), _vp(self.acc_count), _vp(self.wf_n), _ptr(self.wf_mean),
                _ptr(self.wf_m2), _vp(self.status), _ptr(traj), _vp(acc), _vp(sub), _ptr(lp_out),
                _ptr(margin), C.c_int32(self.nthreads))
        self.t += nsteps
        return out


def nested_levels(surrogates, sub_chain_lengths, target):
    """Effective levels of MLDA with an even number n of surrogates as the reference runs it
    (mlda.py:12-73,100-116): returns (P, Q, steps) with P[e], Q[e] for e = 0..E-1 (Q[0] None),
    steps[e] for e < E-1, E = n/2 + 1."""
    n = len(surrogates)
    assert n >= 2 and n % 2 == 0 and len(sub_chain_lengths) == n
    E = n // 2 + 1
    P = [surrogates[2 * e] for e in range(E - 1)] + [target]
    Q = [None] + [surrogates[2 * e - 1] for e in range(1, E)]
    steps = [int(sub_chain_lengths[2 * e + 1]) for e in range(E - 1)]
    return P, Q, steps


class NestedEnsemble:
    """K chains of the nested delayed-acceptance scheme (ymo_mlda_nested_advance)."""

    def __init__(self, P, Q, steps, sigma, x0, seed, chain0=0, eq_mode=EQ_EXACT, nthreads=1):
        self.P, self.Q, self.steps = list(P), list(Q), [int(v) for v in steps]
        self.E = len(self.P)
        assert len(self.Q) == self.E and len(self.steps) == self.E - 1 and self.E >= 2
        self.d = self.P[-1]["dim"]
        x0 = _f64(x0)
        if x0.ndim == 1:
            x0 = x0.reshape(1, -1)
        self.K = x0.shape[0]
        self.x = x0.copy()
        self.sigma = _f64(sigma).reshape(-1)
        self.seed, self.chain0, self.eq_mode, self.nthreads = int(seed), int(chain0), int(eq_mode), int(nthreads)
        self.t = 0
        cost = 1
        for s_ in self.steps:
            cost = s_ * cost + 1
        self.cost_top = cost
        self.lp = logdensity(self.P[-1], self.x)[0]
        self.lpq = logdensity(self.Q[-1], self.x)[0]
        self.acc_count = np.zeros(self.K, dtype=np.uint64)
        self.level_acc = np.zeros((self.E - 1, self.K), dtype=np.uint64)
        self.wf_n = np.zeros(self.K, dtype=np.uint64)
        self.wf_mean = np.zeros((self.K, self.d))
        self.wf_m2 = np.zeros((self.K, self.d))
        self.status = np.zeros(self.K, dtype=np.uint32)

    def advance(self, nsteps):
        K, d, E = self.K, self.d, self.E
        traj = np.empty((nsteps, K, d))
        acc = np.empty((nsteps, K), dtype=np.uint8)
        lp_out = np.empty((nsteps, K))
        keep = []
        Parr = (C.c_void_p * E)()
        Qarr = (C.c_void_p * E)()
        for e in range(E):
            dp, kp = _lower(self.P[e])
            keep += [dp, kp]
            Parr[e] = C.addressof(dp)
            if self.Q[e] is not None:
                dq, kq = _lower(self.Q[e])
                keep += [dq, kq]
                Qarr[e] = C.addressof(dq)
        steps = (C.c_int32 * max(E - 1, 1))(*self.steps)
        lib().ymo_mlda_nested_advance(
            C.c_int32(E), Parr, Qarr, steps, _ptr(self.sigma), C.c_int32(self.eq_mode), C.c_uint64(self.seed),
            C.c_uint64(self.chain0), C.c_int64(K), C.c_uint64(self.t), C.c_uint64(self.cost_top), C.c_int64(nsteps),
            _ptr(self.x), _ptr(self.lp), _ptr(self.lpq), _vp(self.acc_count), _vp(self.level_acc), _vp(self.wf_n),
            _ptr(self.wf_mean), _ptr(self.wf_m2), _vp(self.status), _ptr(traj), _vp(acc), _ptr(lp_out),
            C.c_int32(self.nthreads))
        self.t += nsteps
        return dict(traj=traj, acc=acc, logpost=lp_out)


class AemEnsemble:
    """K chains of two-level delayed acceptance with the adaptive error model on the coarse
    (linear) likelihood (ymo_aem_advance); `fine`, `coarse` are posterior_linear dicts."""

    def __init__(self, fine, coarse, sigma, x0, seed, sub_chain_length, min_data_size, use_noise_heuristic,
                 chain0=0, eq_mode=EQ_EXACT, nthreads=1):
        self.fine, self.coarse = fine, coarse
        self.S, self.min_data, self.heur = int(sub_chain_length), int(min_data_size), int(bool(use_noise_heuristic))
        self.d, self.m = fine["dim"], coarse["data_dim"]
        x0 = _f64(x0)
        if x0.ndim == 1:
            x0 = x0.reshape(1, -1)
        self.K = K = x0.shape[0]
        self.x = x0.copy()
        self.sigma = _f64(sigma).reshape(-1)
        self.seed, self.chain0, self.eq_mode, self.nthreads = int(seed), int(chain0), int(eq_mode), int(nthreads)
        self.t = 0
        self.lp = logdensity(fine, self.x)[0]
        self.acc_count = np.zeros(K, dtype=np.uint64)
        self.wf_n = np.zeros(K, dtype=np.uint64)
        self.wf_mean = np.zeros((K, self.d))
        self.wf_m2 = np.zeros((K, self.d))
        self.status = np.zeros(K, dtype=np.uint32)
        self.aem_n = np.zeros(K, dtype=np.uint64)
        self.aem_mean = np.zeros((K, self.m))
        self.aem_m2 = np.zeros((K, self.m))
        self.aem_prec = np.zeros((K, self.m))
        self.aem_flags = np.zeros((K, 2), dtype=np.int32)   # has_prec, cache fill
        self.cache_key = np.zeros((K, 3, self.d))
        self.cache_val = np.zeros((K, 3))

    def advance(self, nsteps, want_sub=False):
        K, d = self.K, self.d
        traj = np.empty((nsteps, K, d))
        acc = np.empty((nsteps, K), dtype=np.uint8)
        lp_out = np.empty((nsteps, K))
        sub = np.empty((nsteps, self.S, K), dtype=np.uint8) if want_sub else None
        fine, k1 = _lower(self.fine)
        coarse, k2 = _lower(self.coarse)
        lib().ymo_aem_advance(
            C.byref(fine), C.byref(coarse), _ptr(self.sigma), C.c_int32(self.S), C.c_int32(self.min_data),
            C.c_int32(self.heur), C.c_int32(self.eq_mode), C.c_uint64(self.seed), C.c_uint64(self.chain0),
            C.c_int64(K), C.c_uint64(self.t), C.c_int64(nsteps), _ptr(self.x), _ptr(self.lp), _vp(self.acc_count),
            _vp(self.wf_n), _ptr(self.wf_mean), _ptr(self.wf_m2), _vp(self.status), _vp(self.aem_n),
            _ptr(self.aem_mean), _ptr(self.aem_m2), _ptr(self.aem_prec), _vp(self.aem_flags), _ptr(self.cache_key),
            _ptr(self.cache_val), _ptr(traj), _vp(acc), _vp(sub), _ptr(lp_out), C.c_int32(self.nthreads))
        self.t += nsteps
        return dict(traj=traj, acc=acc, logpost=lp_out, sub_acc=sub)


def max_threads():
    return int(lib().ymo_max_threads())
