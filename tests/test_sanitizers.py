"""AddressSanitizer + UndefinedBehaviorSanitizer passes on the CPU (GPU ASan is not available for gfx950
without xnack): the checker (oracle/ymc_oracle.c, gcc) over the golden vectors, and the host side of the C
ABI (csrc/ymc_api.hip + ymc_comm.hip, hipcc host compilation) over every entry point that can run without
a device.  Each pass is a subprocess with the matching sanitizer runtime preloaded; any report aborts it."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(cmd, name):
    try:
        path = subprocess.check_output(cmd + [f"-print-file-name={name}"], text=True).strip()
    except (OSError, subprocess.CalledProcessError):
        return None
    return path if os.path.isabs(path) and os.path.exists(path) else None


def _env(preload, **extra):
    env = dict(os.environ, LD_PRELOAD=preload, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", PYTHONDONTWRITEBYTECODE="1")
    env.update(extra)
    return env


def test_oracle_under_asan_ubsan():
    rt = _runtime(["gcc"], "libasan.so")
    if rt is None:
        pytest.skip("gcc's libasan.so not found")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libymc_oracle_san.so"])
    lib = os.path.join(ROOT, "oracle", "libymc_oracle_san.so")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_golden.py"), os.path.join(ROOT, "tests", "test_numerics.py")],
                       env=_env(rt, YMC_ORACLE_LIB=lib), cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "passed" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


_HOST_SNIPPET = r"""
import ctypes as C, numpy as np, sys
sys.path.insert(0, %(root)r)
from yagremcmc_amd import _ffi
lib = _ffi.lib()
assert lib.ymc_abi_version() == _ffi.ABI_VERSION
# argument validation and the error channel of every entry point that needs no device
n = C.c_int(-1)
rc = lib.ymc_device_count(C.byref(n))
assert rc in (_ffi.OK, _ffi.ERR_DEVICE)
assert lib.ymc_device_count(None) == _ffi.ERR_INVALID and b"null" in lib.ymc_last_error()
out = C.c_void_p()
assert lib.ymc_ctx_create(0, None) == _ffi.ERR_INVALID
if rc != _ffi.OK or n.value == 0:
    assert lib.ymc_ctx_create(0, C.byref(out)) < 0 and len(lib.ymc_last_error()) > 0
for fn, args in (("ymc_ctx_destroy", (None,)), ("ymc_problem_destroy", (None,)), ("ymc_run_destroy", (None,)),
                 ("ymc_comm_destroy", (None,))):
    assert getattr(lib, fn)(*args) == _ffi.OK
desc = _ffi.ProblemDesc()
assert lib.ymc_problem_create(None, C.byref(desc), C.byref(out)) == _ffi.ERR_INVALID
assert lib.ymc_run_create(None, None, None, C.byref(out)) == _ffi.ERR_INVALID
for fn in ("ymc_run_sync", "ymc_run_rewind_record"):
    assert getattr(lib, fn)(None) == _ffi.ERR_INVALID
assert lib.ymc_run_advance(None, 5) == _ffi.ERR_INVALID
assert lib.ymc_run_fetch_trajectory(None, 0, 1, None, 0) == _ffi.ERR_INVALID
assert lib.ymc_run_fetch_packed_summary(None, None, 0) == _ffi.ERR_INVALID
assert lib.ymc_ctx_clock_probe_start(None, 1.0) == _ffi.ERR_INVALID
assert lib.ymc_comm_count(None, None, None) == _ffi.ERR_INVALID
buf = C.create_string_buffer(8)
assert lib.ymc_comm_unique_id(buf, 8) == _ffi.ERR_INVALID          # too small a buffer
assert lib.ymc_comm_create(None, buf, 8, 2, 0, C.byref(out)) == _ffi.ERR_INVALID
assert lib.ymc_iat_series(None, None, 1, 10, 5.0, None, None, 0) == _ffi.ERR_INVALID
msg = lib.ymc_last_error()
assert isinstance(msg, bytes) and len(msg) > 0
print("host entry points ok")
"""


def test_c_abi_host_code_under_asan_ubsan():
    rt = _runtime(["/opt/rocm/lib/llvm/bin/clang"], "libclang_rt.asan-x86_64.so")
    if rt is None:
        pytest.skip("clang's ASan runtime not found")
    subprocess.check_call(["make", "-s", "-j", "4", "-C", os.path.join(ROOT, "yagre-mcmc_amd", "csrc"), "san"])
    lib = os.path.join(ROOT, "yagre-mcmc_amd", "libymc_san.so")
    assert os.path.exists(lib)
    r = subprocess.run([sys.executable, "-c", _HOST_SNIPPET % dict(root=ROOT)], env=_env(rt, YMC_LIBRARY=lib), cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "host entry points ok" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
