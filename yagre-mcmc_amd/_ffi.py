"""ctypes binding of libymc.so (include/ymc.h).

This is the whole host<->device boundary: plain pointers and sizes, no PyTorch.
There is no CPU fallback: if the library is missing or a call fails, an
exception is raised.
"""
import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# YMC_LIBRARY: another build of the same library (the host-sanitizer build of tests/test_sanitizers.py)
LIB_PATH = os.environ.get("YMC_LIBRARY") or os.path.join(_HERE, "libymc.so")

ABI_VERSION = 4

# ymc_status
OK, ERR_INVALID, ERR_DEVICE, ERR_UNSUPPORTED, ERR_CAPACITY, ERR_COMM = 0, -1, -2, -3, -4, -5
# ymc_density_kind
GAUSS_WHITENED, GAUSS_ISO1D, POSTERIOR_LINEAR, POSTERIOR_LV, POSTERIOR_LOGISTIC = 1, 2, 3, 4, 5
# ymc_method / ymc_state_eq
METHOD_MRW, METHOD_MLDA2, METHOD_PCN, METHOD_MLDA_NESTED, METHOD_AEM2 = 1, 2, 3, 4, 5
MAX_LEVELS = 8
EQ_EXACT, EQ_ISCLOSE = 0, 1
# record flags
RECORD_TRAJECTORY, RECORD_ACCEPT, RECORD_LOGPOST, RECORD_SUB_ACCEPT, RECORD_MOMENTS = 1, 2, 4, 8, 16
CHAIN_NONFINITE = 1
CHAIN_HANDOFF = 2
IAT_UNDEFINED = -2147483648
UNIQUE_ID_BYTES = 128

LANE_DMAX = 8
LANE_MMAX = 8

_dp = C.POINTER(C.c_double)


class YmcError(RuntimeError):
    """A libymc call failed (status code + the library's message)."""

    def __init__(self, code, message):
        super().__init__(f"libymc error {code}: {message}")
        self.code = code


class DensityDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("dim", C.c_int32),
        ("mean", _dp), ("whiten", _dp), ("log_norm", C.c_double), ("variance", C.c_double),
        ("n_data", C.c_int32), ("data_dim", C.c_int32),
        ("data", _dp), ("noise_precision", _dp), ("prior_mean", _dp), ("prior_precision", _dp),
        ("fwd_matrix", _dp), ("fwd_shift", _dp), ("design", _dp),
        ("t_final", C.c_double), ("alpha", C.c_double), ("gamma", C.c_double),
        ("rk4_steps", C.c_int32), ("likelihood_only", C.c_int32),
        ("noise_chol_inv", _dp), ("prior_chol_inv", _dp),
        ("lv_rtol", C.c_double), ("lv_atol", C.c_double),
    ]


class ProblemDesc(C.Structure):
    _fields_ = [
        ("method", C.c_int32), ("state_eq", C.c_int32),
        ("target", DensityDesc), ("surrogate", DensityDesc),
        ("sub_chain_length", C.c_int32), ("aem_min_data_size", C.c_int32),
        ("proposal_sigma", _dp), ("pcn_step_size", C.c_double),
        ("n_levels", C.c_int32), ("aem_use_noise_heuristic", C.c_int32),
        ("level_target", C.POINTER(DensityDesc)), ("level_correction", C.POINTER(DensityDesc)),
        ("level_steps", C.POINTER(C.c_int32)),
        ("proposal_chol", _dp),
    ]


class RunDesc(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64), ("chain_offset", C.c_uint64), ("n_chains", C.c_uint64),
        ("initial_state", _dp), ("x0_rows", C.c_uint64), ("record_capacity", C.c_uint64),
        ("record_flags", C.c_uint32), ("record_stride", C.c_uint32),
    ]


class EgressView(C.Structure):
    _fields_ = [("trajectory", _dp), ("accept", C.POINTER(C.c_uint8)), ("logpost", _dp),
                ("first", C.c_uint64), ("rows", C.c_uint64)]


class DeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("arch", C.c_char * 64), ("compute_units", C.c_int32),
                ("clock_mhz", C.c_int32), ("hbm_bytes", C.c_uint64)]


_vp = C.c_void_p
_u64 = C.c_uint64
_PROTOTYPES = {
    "ymc_abi_version": (C.c_int, []),
    "ymc_last_error": (C.c_char_p, []),
    "ymc_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "ymc_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "ymc_ctx_destroy": (C.c_int, [_vp]),
    "ymc_ctx_device_info": (C.c_int, [_vp, C.POINTER(DeviceInfo)]),
    "ymc_ctx_clock_probe_start": (C.c_int, [_vp, C.c_double]),
    "ymc_ctx_clock_probe_read": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ymc_problem_create": (C.c_int, [_vp, C.POINTER(ProblemDesc), C.POINTER(_vp)]),
    "ymc_problem_destroy": (C.c_int, [_vp]),
    "ymc_run_create": (C.c_int, [_vp, _vp, C.POINTER(RunDesc), C.POINTER(_vp)]),
    "ymc_run_destroy": (C.c_int, [_vp]),
    "ymc_run_advance": (C.c_int, [_vp, _u64]),
    "ymc_run_sync": (C.c_int, [_vp]),
    "ymc_run_kernel_time_ms": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(_u64)]),
    "ymc_run_rewind_record": (C.c_int, [_vp]),
    "ymc_run_steps_done": (C.c_int, [_vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "ymc_run_fetch_trajectory": (C.c_int, [_vp, _u64, _u64, _vp, C.c_size_t]),
    "ymc_run_fetch_accept": (C.c_int, [_vp, _u64, _u64, _vp, C.c_size_t]),
    "ymc_run_fetch_logpost": (C.c_int, [_vp, _u64, _u64, _vp, C.c_size_t]),
    "ymc_run_fetch_sub_accept": (C.c_int, [_vp, _u64, _u64, _vp, C.c_size_t]),
    "ymc_run_fetch_level_accept": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "ymc_run_fetch_error_model": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "ymc_run_fetch_state": (C.c_int, [_vp, _vp, _vp]),
    "ymc_run_fetch_summary": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "ymc_run_fetch_packed_summary": (C.c_int, [_vp, _vp, C.c_size_t]),
    "ymc_run_egress_begin": (C.c_int, [_vp, C.c_int, _u64, _u64, C.c_uint32]),
    "ymc_run_egress_wait": (C.c_int, [_vp, C.c_int, C.POINTER(EgressView)]),
    "ymc_logdensity_eval": (C.c_int, [_vp, _vp, C.c_int, _u64, _vp, _vp, _vp]),
    "ymc_forward_eval": (C.c_int, [_vp, _vp, C.c_int, _u64, _vp, _vp]),
    "ymc_run_iat": (C.c_int, [_vp, _u64, _u64, C.c_int, C.c_double, _vp]),
    "ymc_iat_states": (C.c_int, [_vp, _vp, _u64, _u64, C.c_int32, C.c_int, C.c_double, _vp]),
    "ymc_iat_series": (C.c_int, [_vp, _vp, _u64, _u64, C.c_double, _vp, _vp, _u64]),
    "ymc_rng_fill": (C.c_int, [_vp, _u64, _u64, _u64, _u64, _u64, C.c_int32, _vp, _vp]),
    "ymc_numerics_eval": (C.c_int, [_vp, C.c_int, _u64, _vp, _vp, _vp]),
    "ymc_comm_preload": (C.c_int, []),
    "ymc_comm_unique_id": (C.c_int, [_vp, C.c_size_t]),
    "ymc_comm_create": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, C.c_int, C.POINTER(_vp)]),
    "ymc_comm_destroy": (C.c_int, [_vp]),
    "ymc_comm_count": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ymc_comm_barrier": (C.c_int, [_vp]),
    "ymc_comm_allreduce_max": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "ymc_summary_words": (C.c_int, [_vp, C.POINTER(_u64)]),
    "ymc_comm_allgather_summary": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
}

EXPORTED_SYMBOLS = tuple(_PROTOTYPES)

_lib = None


def lib():
    """Load libymc.so once.  Raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C yagre-mcmc_amd/csrc`. The MH path has no CPU fallback.")
        # the host driver of this pool only supports dmabuf IPC; RCCL / cross-process GPU memory
        # sharing needs this before the HIP runtime initialises
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.ymc_abi_version() != ABI_VERSION:
            raise ImportError("libymc.so ABI version mismatch; rebuild it")
        _lib = handle
    return _lib


def check(code):
    if code != OK:
        msg = lib().ymc_last_error()
        raise YmcError(code, msg.decode("utf-8", "replace") if msg else "")


def checked_iat(iat, allow_undefined=False):
    """A constant series has no autocorrelation (0 / 0; the reference fails on it): raise unless the
    caller asked for the IAT_UNDEFINED markers."""
    bad = int((iat == IAT_UNDEFINED).sum())
    if bad and not allow_undefined:
        raise ValueError(f"{bad} of {iat.size} series are constant (zero variance): their integrated autocorrelation "
                         "time is undefined. Pass allowUndefined=True to get IAT_UNDEFINED (INT32_MIN) for them.")
    return iat


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def dptr(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def vptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None else None


class Handle:
    """Owns one opaque library handle and destroys it exactly once."""

    _destroy = None

    def __init__(self, raw):
        self.raw = raw

    def close(self):
        if self.raw is not None and _lib is not None:
            getattr(_lib, self._destroy)(self.raw)
        self.raw = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context(Handle):
    _destroy = "ymc_ctx_destroy"

    def __init__(self, device=0):
        out = _vp()
        check(lib().ymc_ctx_create(int(device), C.byref(out)))
        super().__init__(out)
        self.device = int(device)

    def info(self):
        di = DeviceInfo()
        check(lib().ymc_ctx_device_info(self.raw, C.byref(di)))
        return dict(name=di.name.decode(), arch=di.arch.decode(), compute_units=di.compute_units,
                    clock_mhz=di.clock_mhz, hbm_bytes=di.hbm_bytes)

    def clock_probe_start(self, duration_ms):
        """Start measuring the shader clock the device holds for the next duration_ms (next to whatever runs)."""
        check(lib().ymc_ctx_clock_probe_start(self.raw, float(duration_ms)))

    def clock_probe_read(self):
        """-> (MHz, measured ms) of the probe started last."""
        mhz, ms = C.c_double(0.0), C.c_double(0.0)
        check(lib().ymc_ctx_clock_probe_read(self.raw, C.byref(mhz), C.byref(ms)))
        return mhz.value, ms.value

    def rng_fill(self, seed, chain0, K, t0, nt, d, normals=True, uniforms=True):
        xi = np.empty((nt, K, d)) if normals else None
        u = np.empty((nt, K)) if uniforms else None
        check(lib().ymc_rng_fill(self.raw, seed, chain0, K, t0, nt, d, vptr(xi), vptr(u)))
        return xi, u

    def iat_states(self, states, method=0, sokal_const=5.0, allow_undefined=False):
        """states (n, K, d) -> IAT per chain"""
        states = f64(states)
        n, K, d = states.shape
        out = np.empty(K, dtype=np.int32)
        check(lib().ymc_iat_states(self.raw, vptr(states), n, K, d, method, float(sokal_const), vptr(out)))
        return checked_iat(out, allow_undefined)

    def iat_series(self, series, sokal_const=5.0, acf_lags=0, allow_undefined=False):
        """series (S, N) -> (IAT per series, acf (S, acf_lags) or None)"""
        series = f64(series)
        S, N = series.shape
        out = np.empty(S, dtype=np.int32)
        acf = np.empty((S, acf_lags)) if acf_lags else None
        check(lib().ymc_iat_series(self.raw, vptr(series), S, N, float(sokal_const), vptr(out), vptr(acf), acf_lags))
        return checked_iat(out, allow_undefined), acf

    def numerics(self, fn, x, divisor=None):
        x = f64(x).reshape(-1)
        y0 = np.empty_like(x)
        y1 = f64(divisor).reshape(-1).copy() if divisor is not None else np.empty_like(x)
        check(lib().ymc_numerics_eval(self.raw, fn, x.size, vptr(x), vptr(y0), vptr(y1)))
        return y0, y1


_default_ctx = {}


def default_context(device=0):
    ctx = _default_ctx.get(device)
    if ctx is None:
        ctx = _default_ctx[device] = Context(device)
    return ctx


def device_count():
    n = C.c_int(0)
    check(lib().ymc_device_count(C.byref(n)))
    return n.value


class Problem(Handle):
    _destroy = "ymc_problem_destroy"

    def __init__(self, ctx, desc, keepalive):
        out = _vp()
        check(lib().ymc_problem_create(ctx.raw, C.byref(desc), C.byref(out)))
        super().__init__(out)
        self.ctx = ctx
        self.dim = desc.target.dim
        self.method = desc.method
        self.S = desc.sub_chain_length if desc.method in (METHOD_MLDA2, METHOD_AEM2) else 0
        self.data_dim = desc.target.data_dim
        self.n_levels = desc.n_levels if desc.method == METHOD_MLDA_NESTED else 0
        self.kind = desc.target.kind
        self._n_data = (desc.target.n_data, desc.target.data_dim)
        del keepalive  # the library copied everything

    def logdensity(self, x, which=0):
        x = f64(x).reshape(-1, self.dim)
        out = np.empty(x.shape[0])
        st = np.zeros(x.shape[0], dtype=np.uint32)
        check(lib().ymc_logdensity_eval(self.ctx.raw, self.raw, which, x.shape[0], vptr(x), vptr(out), vptr(st)))
        return out, st

    def forward(self, x, which=0):
        x = f64(x).reshape(-1, self.dim)
        # one evaluation for all data rows (linear map), or one solve per design row (Lotka-Volterra, extension kinds)
        m = self._n_data[1] if self.kind == POSTERIOR_LINEAR else self._n_data[0] * self._n_data[1]
        out = np.empty((x.shape[0], m))
        check(lib().ymc_forward_eval(self.ctx.raw, self.raw, which, x.shape[0], vptr(x), vptr(out)))
        return out


class EgressArray(np.ndarray):
    """A view into a run's page-locked staging memory (Run.egress_wait).  It keeps its run alive -- the memory is
    freed with the run -- and the run keeps count of the views it has handed out: it refuses to be closed, or to
    grow the slot, under a live one."""

    _owner = None

    def __array_finalize__(self, obj):
        self._owner = getattr(obj, "_owner", None)


class Run(Handle):
    _destroy = "ymc_run_destroy"

    def __init__(self, problem, seed, n_chains, x0, record_capacity, record_flags, chain_offset=0, record_stride=1):
        x0 = f64(x0)
        if x0.ndim == 1:
            x0 = x0.reshape(1, -1)
        if x0.shape[1] != problem.dim or x0.shape[0] not in (1, n_chains):
            raise ValueError(f"initial state has shape {x0.shape}; expected (1|{n_chains}, {problem.dim})")
        desc = RunDesc(seed=seed, chain_offset=chain_offset, n_chains=n_chains, initial_state=dptr(x0),
                       x0_rows=x0.shape[0], record_capacity=record_capacity, record_flags=record_flags,
                       record_stride=int(record_stride))
        out = _vp()
        check(lib().ymc_run_create(problem.ctx.raw, problem.raw, C.byref(desc), C.byref(out)))
        super().__init__(out)
        self.problem = problem
        self.K, self.d, self.S = int(n_chains), problem.dim, problem.S
        self.flags = record_flags
        self.stride = max(int(record_stride), 1)
        self._egress_views = [[], []]     # weak references to the staging views handed out, per slot
        self._egress_rows = [0, 0]        # the largest request a slot has been sized for

    def _live_egress_views(self, slot):
        self._egress_views[slot] = [w for w in self._egress_views[slot] if w() is not None]
        return len(self._egress_views[slot])

    def close(self):
        if self.raw is not None and any(self._live_egress_views(s) for s in (0, 1)):
            raise RuntimeError("this run still has live egress views (Run.egress_wait): drop them (or copy them) "
                               "before closing it -- they point into its page-locked staging memory")
        super().close()

    def advance(self, nsteps):
        check(lib().ymc_run_advance(self.raw, int(nsteps)))

    def sync(self):
        check(lib().ymc_run_sync(self.raw))

    def kernel_time_ms(self):
        avg, n = C.c_double(0.0), _u64(0)
        check(lib().ymc_run_kernel_time_ms(self.raw, C.byref(avg), C.byref(n)))
        return avg.value, n.value

    def rewind_record(self):
        check(lib().ymc_run_rewind_record(self.raw))

    def steps_done(self):
        total, rec = _u64(0), _u64(0)
        check(lib().ymc_run_steps_done(self.raw, C.byref(total), C.byref(rec)))
        return total.value, rec.value

    def _fetch(self, fn, first, n, shape, dtype):
        if n is None:
            n = self.steps_done()[1] - first
        if first < 0 or n < 0:
            raise ValueError(f"rows [{first}, {first} + {n}) requested; {self.steps_done()[1]} are recorded")
        out = np.empty((n,) + shape, dtype=dtype)
        check(getattr(lib(), fn)(self.raw, first, n, vptr(out), out.nbytes))
        return out

    def trajectory(self, first=0, n=None):
        return self._fetch("ymc_run_fetch_trajectory", first, n, (self.K, self.d), np.float64)

    def accept(self, first=0, n=None):
        return self._fetch("ymc_run_fetch_accept", first, n, (self.K,), np.uint8)

    def logpost(self, first=0, n=None):
        return self._fetch("ymc_run_fetch_logpost", first, n, (self.K,), np.float64)

    def sub_accept(self, first=0, n=None):
        return self._fetch("ymc_run_fetch_sub_accept", first, n, (self.S, self.K), np.uint8)

    def error_model(self):
        """AEM2: the coarse likelihood's learned error model per chain."""
        m = self.problem.data_dim
        n = np.zeros(self.K, dtype=np.uint64)
        mean, m2, prec = np.zeros((self.K, m)), np.zeros((self.K, m)), np.zeros((self.K, m))
        has = np.zeros(self.K, dtype=np.int32)
        check(lib().ymc_run_fetch_error_model(self.raw, vptr(n), vptr(mean), vptr(m2), vptr(prec), vptr(has)))
        return dict(n=n, mean=mean, m2=m2, precision=prec, has_precision=has.astype(bool))

    def level_accept(self):
        """Nested delayed acceptance: (accepted[E-1][K], made[E-1]) of the lower levels."""
        E = self.problem.n_levels
        acc = np.zeros((E - 1, self.K), dtype=np.uint64)
        made = np.zeros(E - 1, dtype=np.uint64)
        check(lib().ymc_run_fetch_level_accept(self.raw, vptr(acc), acc.nbytes, vptr(made)))
        return acc, made

    def state(self):
        x = np.empty((self.K, self.d))
        lp = np.empty(self.K)
        check(lib().ymc_run_fetch_state(self.raw, vptr(x), vptr(lp)))
        return x, lp

    def summary(self):
        acc = np.empty(self.K, dtype=np.uint64)
        n = np.empty(self.K, dtype=np.uint64)
        mean = np.empty((self.K, self.d))
        m2 = np.empty((self.K, self.d))
        st = np.empty(self.K, dtype=np.uint32)
        check(lib().ymc_run_fetch_summary(self.raw, vptr(acc), vptr(n), vptr(mean), vptr(m2), vptr(st)))
        return dict(accepted=acc, wf_n=n, wf_mean=mean, wf_m2=m2, status=st)

    def iat(self, first=0, n=None, method=0, sokal_const=5.0, allow_undefined=False):
        """Integrated autocorrelation time of every chain from record rows [first, first+n)."""
        if n is None:
            n = self.steps_done()[1] - first
        if first < 0 or n < 2:
            raise ValueError(f"the burn-in leaves {n} of {self.steps_done()[1]} recorded states: at least 2 are needed")
        out = np.empty(self.K, dtype=np.int32)
        check(lib().ymc_run_iat(self.raw, first, n, method, float(sokal_const), vptr(out)))
        return checked_iat(out, allow_undefined)

    def egress_begin(self, slot, first, n, what=RECORD_TRAJECTORY):
        """Start copying record rows [first, first + n) to the run's page-locked staging slot 0 | 1 on the copy
        stream; returns at once and the copy runs beside later launches (include/ymc.h, egress)."""
        slot = int(slot)
        if slot in (0, 1) and int(n) > self._egress_rows[slot]:
            # the library re-allocates a slot that has to grow: nothing may still point into the old block
            if self._live_egress_views(slot):
                raise RuntimeError(f"egress slot {slot} must grow from {self._egress_rows[slot]} to {int(n)} rows while "
                                   "views of it are alive: drop or copy them first (or begin with the largest block)")
        check(lib().ymc_run_egress_begin(self.raw, slot, int(first), int(n), int(what)))
        if slot in (0, 1):
            self._egress_rows[slot] = max(self._egress_rows[slot], int(n))

    def egress_wait(self, slot):
        """Block until the slot's copy is done; -> dict of NumPy VIEWS into the page-locked staging memory
        (trajectory (n, K, d), accept (n, K), logpost (n, K) as requested; `first`); their CONTENT is valid until
        the next egress_begin on that slot, the memory for as long as a view exists (each holds its run)."""
        v = EgressView()
        check(lib().ymc_run_egress_wait(self.raw, int(slot), C.byref(v)))
        n = int(v.rows)
        out = dict(first=int(v.first))

        def view(ptr, shape):
            a = np.ctypeslib.as_array(ptr, shape=shape).view(EgressArray)
            a._owner = self           # the staging memory lives exactly as long as the run
            self._egress_views[int(slot)].append(weakref.ref(a))
            return a
        if v.trajectory:
            out["trajectory"] = view(v.trajectory, (n, self.K, self.d))
        if v.accept:
            out["accept"] = view(v.accept, (n, self.K))
        if v.logpost:
            out["logpost"] = view(v.logpost, (n, self.K))
        return out

    def packed_summary(self):
        """(K, 3d+5) u64: the records pack_summary_kernel writes (the payload of the multi-GPU gather)."""
        out = np.empty((self.K, self.summary_words()), dtype=np.uint64)
        check(lib().ymc_run_fetch_packed_summary(self.raw, vptr(out), out.nbytes))
        return out

    def summary_words(self):
        w = _u64(0)
        check(lib().ymc_summary_words(self.raw, C.byref(w)))
        return w.value


class Communicator(Handle):
    """One rank of the RCCL communicator (one process per GPU)."""

    _destroy = "ymc_comm_destroy"
    backend = "rccl"

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(UNIQUE_ID_BYTES)
        check(lib().ymc_comm_unique_id(buf, UNIQUE_ID_BYTES))
        return bytes(buf.raw)

    @staticmethod
    def preload():
        """Load librccl now (raises YmcError if it cannot be found)."""
        check(lib().ymc_comm_preload())

    def count(self):
        """The communicator's size as RCCL reports it (ncclCommCount)."""
        n = C.c_int(0)
        check(lib().ymc_comm_count(self.raw, C.byref(n), None))
        return n.value

    def close(self):
        super().close()
        store, self.store = getattr(self, "store", None), None
        if store is not None:
            store.close()

    def __init__(self, ctx, unique_id, nranks, rank):
        out = _vp()
        buf = C.create_string_buffer(unique_id, UNIQUE_ID_BYTES)
        check(lib().ymc_comm_create(ctx.raw, buf, UNIQUE_ID_BYTES, nranks, rank, C.byref(out)))
        super().__init__(out)
        self.nranks, self.rank = nranks, rank

    def barrier(self):
        check(lib().ymc_comm_barrier(self.raw))

    def allreduce_max(self, value):
        v = C.c_double(float(value))
        check(lib().ymc_comm_allreduce_max(self.raw, C.byref(v)))
        return v.value

    def allgather_summary(self, run):
        words = run.summary_words()
        out = np.empty((self.nranks, run.K, words), dtype=np.uint64)
        check(lib().ymc_comm_allgather_summary(self.raw, run.raw, vptr(out), out.nbytes))
        return out
