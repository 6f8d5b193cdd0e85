#!/usr/bin/env python3
"""bench.py with the device taken out, for the CPU tests of its LAUNCH paths (tests/test_bench_launch.py):
the real bench.main() -- argument handling, self-spawn of N ranks, rendezvous, barrier, max-over-ranks
timing, the gather of summaries, the one JSON line -- over stand-ins for `_ffi.Context` / `_ffi.Run` that
advance a small ensemble with the CPU checker.  Test infrastructure: nothing in the product imports it, and
it only ever runs with YMC_COMM=tcp (RCCL needs the GPUs this stands in for).

Children started by bench.py's own launcher re-run `sys.argv`, i.e. this file, so every rank is stubbed."""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if os.environ.get("YMC_STUB_FAIL_RANK") is not None and os.environ.get("RANK") == os.environ["YMC_STUB_FAIL_RANK"]:
    sys.exit(9)         # a rank that dies before the rendezvous

from oracle import oracle  # noqa: E402
from yagremcmc_amd import _ffi  # noqa: E402

spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


class StubContext:
    def __init__(self, device=0):
        self.device = device

    def info(self):
        return dict(name="stub", arch="none", compute_units=256, clock_mhz=2400, hbm_bytes=0)

    def clock_probe_start(self, ms):
        pass

    def clock_probe_read(self):
        return 2400.0, 1.0


class StubRun:
    def __init__(self, problem, seed, n_chains, x0, cap, flags, chain_offset=0, record_stride=1):
        self.K, self.d = int(n_chains), 2
        self.e = oracle.Ensemble(problem, np.ones(2), np.tile(x0, (self.K, 1)), seed, chain0=chain_offset)
        self.launches = self.total = 0

    def rewind_record(self):
        pass

    def advance(self, T):
        self.e.advance(T, record=False)
        self.launches += 1
        self.total += T

    def sync(self):
        pass

    def kernel_time_ms(self):
        n, self.launches = self.launches, 0
        return 1.0, n

    def steps_done(self):
        return self.total, 0

    def summary(self):
        e = self.e
        status = e.status | np.uint32(int(os.environ.get("YMC_STUB_STATUS_BITS", "0")))
        return dict(accepted=e.acc_count, wf_n=e.wf_n, wf_mean=e.wf_mean, wf_m2=e.wf_m2, status=status)

    def state(self):
        return self.e.x, self.e.lp


def stub_workload(name, ffi, ctx, lowering):
    U, c = bench.whitening(bench.TGT_MEAN, bench.TGT_COV)
    return dict(problem=oracle.gauss_whitened(bench.TGT_MEAN, U, c), x0=np.array([[-8.0, -7.0]]), K=32, T=40, L=1,
                d=2, desc=f"stub ({name}): C2's target on the CPU checker",
                spec=dict(target=dict(kind="gauss", mean=bench.TGT_MEAN, whiten=U, log_norm=c), sigma=np.ones(2)))


_ffi.Context, _ffi.Run, _ffi.device_count = StubContext, StubRun, lambda: 8
bench.make_workload = stub_workload
if "--no-cpu-baseline" not in sys.argv and os.environ.get("YMC_STUB_CPU_BASELINE") != "1":
    sys.argv.append("--no-cpu-baseline")
bench.main()
