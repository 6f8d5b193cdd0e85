#!/usr/bin/env python3
"""Where the cycles of the two-level matrix-core kernel go (GPU box; `make -C yagre-mcmc_amd/csrc prof` first):
    YMC_TILE_MLDA2=regs|lds python tools/tile_sections.py [aem] [S] [chains]
runs C3's two-level workload on libymc_prof.so (section timers compiled into linear_tile_mlda2_kernel) and prints
the cycles wave 0 of workgroup 0 spent per section, per sub-step / fine step."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["YMC_LIBRARY"] = os.path.join(ROOT, "yagre-mcmc_amd", os.environ.get("YMC_PROF_LIB", "libymc_prof.so"))
AEM = len(sys.argv) > 1 and sys.argv[1] == "aem"
if AEM:
    sys.argv.pop(1)
S = int(sys.argv[1]) if len(sys.argv) > 1 else (4 if AEM else 5)
os.environ["YMC_C3M_S"] = str(S)
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from yagremcmc_amd import _ffi as ffi, lowering  # noqa: E402

ctx = ffi.Context(0)
wl = bench.make_workload("c3a" if AEM else "c3m", ffi, ctx, lowering)
K = int(sys.argv[2]) if len(sys.argv) > 2 else wl["K"]
T = 20
run = ffi.Run(wl["problem"], 20250103, K, wl["x0"], T + 1, ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT)
for _ in range(3):
    run.rewind_record()
    run.advance(T)
run.sync()
ms, n = run.kernel_time_ms()
out = (C.c_ulonglong * 16)()
lib = ffi.lib()
reader = lib.ymc_debug_tile_aem_sections if AEM else lib.ymc_debug_tile_sections
reader.argtypes = [C.POINTER(C.c_ulonglong)]
assert reader(out) == 0
if AEM:
    names = ["draws + proposal + moved", "query(proposal): likelihood now", "query(state)", "accept + select",
             "fine: load state + moved", "fine likelihood", "fine: query(x), query(z)", "error-model update", "select + store + record"]
    per = [T * S] * 4 + [T] * 5
else:
    names = ["draws + proposal", "coarse forward map + likelihood", "coarse prior + accept + select", "fine: moved test",
             "fine forward map + likelihood", "fine prior + accept + select", "record"]
    per = [T * S, T * S, T * S, T, T, T, T]
print(f"mode {os.environ.get('YMC_TILE_MLDA2', 'default')}  S {S}  chains {K}  kernel {ms:.3f} ms / {T} fine steps")
tot = 0.0
for nm, v, p in zip(names, out, per):
    print(f"  {nm:36s} {v / p:10.0f} cycles each   {v / T:10.0f} per fine step")
    tot += v / T
print(f"  {'sum':36s} {'':10s}              {tot:10.0f} per fine step (100 MHz ticks x clock ratio if s_memtime is the shader clock)")
