#!/usr/bin/env python3
"""Headline benchmark: MH steps/s of the chain-ensemble hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5|iat]

One bench "step" = one pass of the hot path over this rank's batch of chains:
`launches_per_step` launches of the fused propose / evaluate / accept / record
kernel, each advancing every chain by `transitions_per_launch` MH transitions
into the record in HBM (full trajectory + accept mask + FullDiagnostics'
Welford moments, i.e. what example_mcmc_2d_singleLevel.py records; the record
is refilled by every launch).  The defaults make a step ~0.5 s, so the
driver's 20 timed steps are a ~10 s region at the clock the chip SUSTAINS (long
enough for a once-per-few-seconds utilisation sampler to see the load), and
the W warm-up steps bring it there (no other pre-heating).  Default workload =
BASELINE.json configs[1]: the 2-D Gaussian target, random-walk proposal,
65 536 chains per GPU.  For N > 1: one process per GPU, either started by a
launcher (torchrun env: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*) or, when
`python bench.py --gpus N` is run as one plain process, by bench.py itself
(N fresh children before anything touches the GPU); chains shard by
global chain id, nothing is exchanged while sampling (weak scaling), and the
single collective is an RCCL all-gather of the per-chain summaries after the
timed region.  RCCL failing to come up is an error (exit code 1), never a
silent switch of transport.  No PyTorch anywhere: the host drives libymc.so
through ctypes.

Rank 0 prints ONE JSON line.  `roofline.achieved` = algorithmic bytes (or
flops) per launch / average kernel time measured with HIP
events on the kernel's stream in THIS run; fields copied from committed
profiles carry a `source`.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6  # AMD MI355X data sheet: FP64 vector = FP64 matrix = 78.6 TFLOP/s (the guide lists no f64 rate)
STEP_TARGET_S = 0.5      # a bench step is sized to about this much device time
# seconds of CPU work of a cpu_baseline leg: (all visible threads, 16 threads, 1 thread); YMC_BENCH_CPU_SCALE
# shortens them for the launch tests
_CPU_SCALE = float(os.environ.get("YMC_BENCH_CPU_SCALE", "1"))
CPU_SECONDS = tuple(_CPU_SCALE * v for v in (8.0, 4.0, 4.0))
LEG_CPU_SECONDS = tuple(_CPU_SCALE * v for v in (2.5, 2.5, 1.5))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c5", "c4", "iat", "c3m", "c3a", "wide", "widek"])
    ap.add_argument("--chains", type=int, default=0, help="chains per GPU (default: the workload's)")
    ap.add_argument("--transitions", type=int, default=0, help="MH transitions per chain per launch")
    ap.add_argument("--launches-per-step", type=int, default=0, help="kernel launches per bench step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-moments", action="store_true")
    ap.add_argument("--record-stride", type=int, default=1,
                    help="thinned record: keep every s-th state (drops the per-transition decision record)")
    ap.add_argument("--legs", default="auto",
                    help="workloads measured after the headline and reported under \"workloads\": auto (next to the "
                         "default C2 headline: c3,c3m,c4,c5 on one GPU, c4,c5 on N > 1), none, or a comma list")
    return ap.parse_args()


# ---------------------------------------------------------------------------
# synthetic workloads: plain arrays; the device problem is built
# from them here, the CPU checker's (cpu_baseline only) from the same arrays
# ---------------------------------------------------------------------------

TGT_MEAN = np.array([1.0, 1.5])
TGT_COV = np.array([[2.4, -0.5], [-0.5, 0.7]])


def whitening(mean, cov):
    w, v = np.linalg.eigh(cov)
    return v * np.sqrt(1.0 / w), float(mean.size * np.log(2.0 * np.pi) + np.sum(np.log(w)))


def _gauss(ffi, lowering, mean, cov):
    U, c = whitening(mean, cov)
    low = lowering.Lowered(ffi.GAUSS_WHITENED, mean.size)
    low.set_array("mean", mean)
    low.set_array("whiten", U)
    low.desc.log_norm = c
    return low, dict(kind="gauss", mean=mean, whiten=U, log_norm=c)


def _lv_lowered(ffi, lowering, design, data, noise_prec, prior_mean, prior_prec):
    low = lowering.Lowered(ffi.POSTERIOR_LV, 2)
    low.desc.n_data, low.desc.data_dim = design.shape[0], 2
    for k, v in (("data", data), ("noise_precision", noise_prec), ("prior_mean", prior_mean),
                 ("prior_precision", prior_prec), ("design", design)):
        low.set_array(k, v)
    low.desc.t_final, low.desc.alpha, low.desc.gamma, low.desc.rk4_steps = 10.0, 0.8, 0.4, 50
    return low


MFMA_WORKLOADS = ("c3", "c3m", "c3a", "wide", "widek")   # linear models on the matrix cores: roofline = fp64 MFMA


def make_workload(name, ffi, ctx, lowering):
    """-> dict(problem, x0 (1,d), K, T, L, desc, d, spec): defaults of chains / GPU, transitions / launch and
    launches / step; `spec` holds the arrays that describe the same problem to the cpu_baseline leg."""
    if name in ("c2", "c5", "iat"):
        fine, s_fine = _gauss(ffi, lowering, TGT_MEAN, TGT_COV)
        sigma = np.sqrt(np.reciprocal(np.reciprocal(np.full(2, 1.0))))
        x0 = np.array([[-8.0, -7.0]])
        if name in ("c2", "iat"):
            prob = lowering.make_problem(ctx, ffi.METHOD_MRW, fine, sigma)
            return dict(problem=prob, x0=x0, K=65536, T=10000, L=100, d=2,
                        desc="C2: 2-D Gaussian target (example_mcmc_2d_singleLevel.py), random-walk proposal "
                             "var 1.0, x0=(-8,-7)", spec=dict(target=s_fine, sigma=sigma))
        cm = TGT_MEAN + np.array([0.15, -0.2])
        cc = 2.0 * TGT_COV + np.array([[0.1, 1.2], [1.2, 0.05]])
        coarse, s_coarse = _gauss(ffi, lowering, cm, cc)
        prob = lowering.make_problem(ctx, ffi.METHOD_MLDA2, fine, sigma, surrogate=coarse, sub_chain_length=6)
        return dict(problem=prob, x0=x0, K=131072, T=1000, L=100, d=2,
                    desc="C5: two-level delayed acceptance (example_mcmc_2d_twoLevel.py), S=6",
                    spec=dict(target=s_fine, sigma=sigma, surrogate=s_coarse, S=6))
    if name in MFMA_WORKLOADS:
        # linear forward model, d = m = 128, nData = 5; "wide" / "widek": the same construction at d = m = 512
        # (beyond the BASELINE configs: the persistent wide kernels on 65 536 chains, the group kernels on 256)
        d = 512 if name in ("wide", "widek") else 128
        rng = np.random.default_rng(7)
        A = rng.standard_normal((d, d)) / np.sqrt(d)
        b = rng.standard_normal(d)
        theta = rng.standard_normal(d)
        data = np.array([A @ theta + b + np.sqrt(0.3) * rng.standard_normal(d)
                         for _ in range(int(os.environ.get("YMC_C3_ND", "5")))])
        noise_prec, prior_mean, prior_prec = np.reciprocal(np.full(d, 0.3)), np.zeros(d), np.reciprocal(np.full(d, 5.0))
        low = lowering.Lowered(ffi.POSTERIOR_LINEAR, d)
        low.desc.n_data, low.desc.data_dim = data.shape[0], d
        for k, v in (("data", data), ("noise_precision", noise_prec), ("prior_mean", prior_mean),
                     ("prior_precision", prior_prec), ("fwd_matrix", A), ("fwd_shift", b)):
            low.set_array(k, v)
        sigma = np.sqrt(np.reciprocal(np.reciprocal(np.full(d, 1.1e-3 * 128 / d))))
        if name == "c3a":
            # the adaptive error model over two models of C3's size: S = 4 coarse transitions per fine step,
            # the coarse likelihood learning the model error F_fine - F_coarse (minDataSize 8, noise heuristic on)
            A_c, b_c = A + 0.02 * rng.standard_normal((d, d)) / np.sqrt(d), b + 0.05 * rng.standard_normal(d)
            coarse = lowering.Lowered(ffi.POSTERIOR_LINEAR, d)
            coarse.desc.n_data, coarse.desc.data_dim = data.shape[0], d
            for k, v in (("data", data), ("noise_precision", noise_prec), ("prior_mean", prior_mean),
                         ("prior_precision", prior_prec), ("fwd_matrix", A_c), ("fwd_shift", b_c)):
                coarse.set_array(k, v)
            prob = lowering.make_problem(ctx, ffi.METHOD_AEM2, low, sigma, surrogate=coarse, sub_chain_length=4, aem=(8, True))
            return dict(problem=prob, x0=np.zeros((1, d)), K=65536, T=20, L=30, d=d,
                        desc="C3 adaptive error model: AEM delayed acceptance over two linear models, d = m = 128, nData = 5, "
                             "S = 4 (fp64 MFMA tile kernel; steps = FINE steps)",
                        spec=dict(S=4, fine_maps=2, aem=True))
        if name == "c3m":
            # two-level delayed acceptance over a model hierarchy at C3's size (configs[2] is named after
            # example_inference_linearModel_twoLevel.py): the coarse model is a perturbation of the fine one
            A_c, b_c = A + 0.02 * rng.standard_normal((d, d)) / np.sqrt(d), b + 0.05 * rng.standard_normal(d)
            coarse = lowering.Lowered(ffi.POSTERIOR_LINEAR, d)
            coarse.desc.n_data, coarse.desc.data_dim = data.shape[0], d
            for k, v in (("data", data), ("noise_precision", noise_prec), ("prior_mean", prior_mean),
                         ("prior_precision", prior_prec), ("fwd_matrix", A_c), ("fwd_shift", b_c)):
                coarse.set_array(k, v)
            S = int(os.environ.get("YMC_C3M_S", "5"))     # the workload is S = 5; other values for kernel experiments
            prob = lowering.make_problem(ctx, ffi.METHOD_MLDA2, low, sigma, surrogate=coarse, sub_chain_length=S)
            lin = lambda A_, b_: dict(kind="linear", A=A_, b=b_, data=data, noise_prec=noise_prec,  # noqa: E731
                                      prior_mean=prior_mean, prior_prec=prior_prec)
            return dict(problem=prob, x0=np.zeros((1, d)), K=262144, T=20, L=18, d=d,
                        desc="C3 two-level: delayed acceptance over two linear models, d = m = 128, nData = 5, S = 5 "
                             "(fp64 MFMA tile kernel; steps = FINE steps, each with 5 coarse transitions)",
                        spec=dict(target=lin(A, b), surrogate=lin(A_c, b_c), S=S, sigma=sigma))
        prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, sigma)
        if name in ("wide", "widek"):
            small = name == "widek"
            return dict(problem=prob, x0=np.zeros((1, d)), K=256 if small else 65536, T=1000 if small else 30, L=25 if small else 24, d=d,
                        desc="linear forward model posterior d = m = 512, nData = 5 (" +
                             ("a tile of 16 chains over a group of 16 workgroups: linear_wide_systolic_kernel)" if small
                              else "one persistent launch per advance call: linear_wide_fused_kernel)"),
                        spec=dict(target=dict(kind="linear", A=A, b=b, data=data, noise_prec=noise_prec,
                                              prior_mean=prior_mean, prior_prec=prior_prec), sigma=sigma))
        return dict(problem=prob, x0=np.zeros((1, d)), K=262144, T=100, L=25, d=d,
                    desc="C3: linear forward model posterior d = m = 128, nData = 5 (fp64 MFMA tile kernel)",
                    spec=dict(target=dict(kind="linear", A=A, b=b, data=data, noise_prec=noise_prec,
                                          prior_mean=prior_mean, prior_prec=prior_prec), sigma=sigma))
    # c4: Lotka-Volterra, nData=10, T=10, RK4 x 50.  The synthetic observations are the PRODUCT's own forward
    # map (ymc_forward_eval) at the true parameter plus noise.
    design = np.random.default_rng(1111).uniform(0.5, 1.5, size=(10, 2))
    truth = np.log(np.array([0.4, 0.6]))
    noise_prec, prior_mean, prior_prec = np.reciprocal(np.full(2, 0.04)), np.zeros(2), np.reciprocal(np.full(2, 1.4))
    sigma = np.sqrt(np.reciprocal(np.reciprocal(np.full(2, 0.15))))
    probe = lowering.make_problem(ctx, ffi.METHOD_MRW,
                                  _lv_lowered(ffi, lowering, design, np.zeros((10, 2)), noise_prec, prior_mean, prior_prec),
                                  sigma)
    clean = probe.forward(truth).reshape(10, 2)
    data = clean + np.sqrt(0.04) * np.random.default_rng(2222).standard_normal((10, 2))
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW,
                                 _lv_lowered(ffi, lowering, design, data, noise_prec, prior_mean, prior_prec), sigma)
    return dict(problem=prob, x0=np.array([[-0.6, -0.3]]), K=16384, T=1000, L=40, d=2,
                desc="C4: Lotka-Volterra posterior, nData=10, T=10, RK4 x 50",
                spec=dict(target=dict(kind="lv", design=design, data=data, noise_prec=noise_prec,
                                      prior_mean=prior_mean, prior_prec=prior_prec), sigma=sigma))


def cpu_baseline(spec, x0, seconds=CPU_SECONDS):
    """The oracle (CPU restatement, kind "port") timed on the host cores on a bounded sample of the same
    workload: same target, same recording, fewer chains and transitions (~16 s of CPU work for the headline,
    ~6 s for a leg): all hardware threads the process may use, 16 threads (one GPU's share of an 8-GPU host)
    and one thread."""
    from oracle import oracle  # the ONLY place the benchmark touches the checker

    def lower(s):
        if s["kind"] == "gauss":
            return oracle.gauss_whitened(s["mean"], s["whiten"], s["log_norm"])
        if s["kind"] == "linear":
            return oracle.posterior_linear(s["A"], s["b"], s["data"], s["noise_prec"], s["prior_mean"], s["prior_prec"])
        return oracle.posterior_lv(s["design"], 10.0, 0.8, 0.4, 50, s["data"], s["noise_prec"], s["prior_mean"],
                                   s["prior_prec"])

    target = lower(spec["target"])
    surrogate = lower(spec["surrogate"]) if "surrogate" in spec else None
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    visible = max(1, min(visible, oracle.max_threads()))

    def timed(K, T, threads, min_seconds):
        e = oracle.Ensemble(target, spec["sigma"], np.tile(x0, (K, 1)), 20250103, surrogate=surrogate,
                            sub_chain_length=spec.get("S", 0), nthreads=threads)
        bufs = e.advance(T)          # untimed: faults the record buffers in, like the GPU warm-up
        reps, t = 0, time.perf_counter()
        while True:
            e.advance(T, reuse=bufs)
            reps += 1
            dt = time.perf_counter() - t
            if dt >= min_seconds:
                return K * T * reps / dt, reps, dt

    heavy = spec["target"]["kind"] != "gauss"
    T = 50 if heavy else 500
    per_thread = 16 if heavy else 1024    # ~25 B of record per transition (d = 2)
    rate_all, reps, dt = timed(per_thread * visible, T, visible, seconds[0])
    n16 = min(16, visible)
    rate16 = timed(per_thread * n16, T, n16, seconds[1])[0] if n16 != visible else rate_all
    rate1 = timed(per_thread, T, 1, seconds[2])[0]
    # the GPU box gives a one-GPU job a share of the host's CPU time: every visible hardware thread at once can
    # be SLOWER than 16 threads.  The headline is the better of the two, with the thread count that produced it.
    best, cores = (rate_all, visible) if rate_all >= rate16 else (rate16, n16)
    return {"value": best, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"{per_thread} chains per thread x {T} transitions of the same workload (trajectory, mask, "
                      f"log-density and moments recorded, record buffers resident), passes repeated for {seconds[0]:g} s "
                      f"({visible} threads), {seconds[1]:g} s (16 threads) and {seconds[2]:g} s (1 thread)",
            "host_threads_visible": visible, "all_visible_threads_steps_per_s": rate_all,
            "threads_16_steps_per_s": rate16, "single_thread_steps_per_s": rate1,
            "reference_single_core_steps_per_s_survey_container": 2.2e4}


def profiled(workload, K, T):
    """What the committed rocprofv3 passes of this configuration say (profiles/pmc_traffic.json): HBM bytes
    per launch and the SQ instruction counts.  Copied, not measured in this run -- hence the `source`."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(path)).get(f"{workload}_K{K}_T{T}")
    except (OSError, ValueError):
        rec = None
    return rec or {}


def roofline_block(name, stride, wl, K, T, kernel_ms, launches, clock_mhz, info):
    d = wl["d"]
    ksec = kernel_ms * 1e-3
    bytes_launch = K * T * (8 * d + 1) if stride <= 1 else K * (T // stride) * 8 * d
    hbm = bytes_launch / ksec / 1e9 if ksec > 0 else 0.0
    rec = profiled(name, K, T)
    traffic = rec.get("hbm_bytes_per_launch")
    common = {"traffic": traffic, "traffic_source": rec.get("source") if traffic else None,
              "kernel_ms_avg": kernel_ms, "launches_timed": launches,
              "algorithmic_bytes_per_launch": bytes_launch, "hbm_gb_per_s": hbm, "hbm_frac": hbm / HBM_PEAK_GBPS}
    # VALU issue: what bounds the lane-per-chain kernels.  One wave64 VALU instruction occupies its SIMD's
    # issue port for 4 cycles; the chip has 4 SIMDs per CU.
    sq = rec.get("sq_per_launch") or {}
    valu = None
    if sq.get("SQ_INSTS_VALU") and clock_mhz and ksec > 0:
        waves = math.ceil(K / (16 if name in MFMA_WORKLOADS else 64))   # waves' worth of chains
        simds = 4 * info["compute_units"]
        avail = ksec * clock_mhz * 1e6 * simds          # SIMD issue cycles in one launch
        mfma = sq.get("SQ_INSTS_MFMA", 0.0)
        ipt = (sq["SQ_INSTS_VALU"] - mfma) / (waves * T)
        busy = 4.0 * (sq["SQ_INSTS_VALU"] - mfma) + sq.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        valu = {"instr_per_transition": ipt, "salu_instr_per_transition": sq.get("SQ_INSTS_SALU", 0.0) / (waves * T),
                "mfma_instr_per_transition": mfma / (waves * T) if mfma else None,
                "mfma_coexec_cycles": sq.get("SQ_VALU_MFMA_COEXEC_CYCLES"),
                "cycles_per_transition": avail / (waves * T), "frac": busy / avail,
                "clock_mhz": clock_mhz, "chain_waves_per_simd": waves / simds,
                "note": "frac = (4 cycles x wave-level VALU instructions + matrix-pipe busy cycles) per launch / (kernel "
                        "time x sustained clock x SIMDs): the share of the SIMDs' fp64 datapath time in use (an f64 MFMA "
                        "and a VALU instruction never execute together on gfx950: SQ_VALU_MFMA_COEXEC_CYCLES = 0); "
                        "instruction counts per wave of chains and transition",
                "source": {"instr": rec.get("source"), "kernel_time_and_clock": "this run"}}
    if name in MFMA_WORKLOADS:
        # c3m: S = 5 coarse + 1 fine forward map per fine step; c3a: S = 4 coarse + the fine matrix twice (likelihood, model error)
        maps = wl["spec"].get("fine_maps", 1) + wl["spec"].get("S", 0)
        flops = 2.0 * d * d * K * T * maps
        tf = flops / ksec / 1e12 if ksec > 0 else 0.0
        return dict(common, bound="mfma", achieved=tf, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s", frac=tf / FP64_PEAK_TFLOPS,
                    algorithmic_flops_per_launch=flops, valu=valu,
                    note="per GPU (rank 0); v_mfma_f64_16x16x4_f64 GEMV (2 d m flop per transition); the d normal "
                         "draws per transition are VALU work on top of it -- DESIGN.md section 4")
    if name == "c4":
        flops = 10 * 50 * 50.0 * K * T   # n_data x rk4_steps x 50 flop (30 fp64 instructions, 20 of them fma)
        tf = flops / ksec / 1e12 if ksec > 0 else 0.0
        return dict(common, bound="valu", achieved=tf, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s", frac=tf / FP64_PEAK_TFLOPS,
                    algorithmic_flops_per_launch=flops, valu=valu,
                    note="per GPU (rank 0); fp64 vector flops of the RK4 integrator alone (10 design rows x 50 steps x "
                         "50 flop per transition) against the 78.6 TFLOP/s vector peak; HBM figures alongside")
    return dict(common, bound="hbm", achieved=hbm, peak=HBM_PEAK_GBPS, unit="GB/s", frac=hbm / HBM_PEAK_GBPS, valu=valu,
                note="per GPU (rank 0); bytes = (8 d + 1) per transition per chain"
                     + (" (per FINE step; every fine step also makes 6 coarse transitions)" if name == "c5" else "")
                     + "; the kernel is VALU-issue bound (Philox + Box-Muller + exp per transition) -- see `valu` "
                       "and DESIGN.md section 4")


def bench_iat(args, ffi, ctx, wl, K, info):
    """Post-processing workload: integrated autocorrelation time of every chain from the record
    in HBM (autocorrelation.py:92-140), 'mean' and 'max'.  One step = both modes over all chains."""
    N = args.transitions or 10000
    run = ffi.Run(wl["problem"], 20250103, K, wl["x0"], N, ffi.RECORD_TRAJECTORY)
    run.advance(N)
    run.sync()
    burn = 200
    n = N - burn
    for _ in range(max(args.warmup, 1)):
        iat_mean = run.iat(burn, n, 0)
        iat_max = run.iat(burn, n, 1)
    steps = max(1, args.steps)
    t0 = time.perf_counter()
    for _ in range(steps):
        iat_mean = run.iat(burn, n, 0)
        iat_max = run.iat(burn, n, 1)
    elapsed = time.perf_counter() - t0
    bytes_step = 2.0 * K * n * 2 * 8          # both modes read the d = 2 record once
    out = {"metric": "IAT post-processing, chains/s", "value": 2.0 * K * steps / elapsed, "unit": "chain-IATs/s",
           "n_gpus": 1, "steps": steps, "warmup": args.warmup, "ms_per_step": elapsed / steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"IAT ('mean' + 'max') of {K} chains x {n} states x d=2 from the HBM record of C2",
                      "chains_per_gpu": K, "states": n},
           "timed_region_s": elapsed,
           "roofline": {"bound": "hbm", "achieved": bytes_step * steps / elapsed / 1e9, "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": bytes_step * steps / elapsed / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                        "note": "algorithmic bytes = one read of the record per mode; end-to-end time of ymc_run_iat "
                                "(series extraction + lag sums + result copy)"},
           "iat_mean_median": float(np.median(iat_mean)), "iat_max_median": float(np.median(iat_max)), "device": info}
    print(json.dumps(out))


# legs run after the headline: BASELINE.json's other configs at their own per-GPU sizes, ~3 s of device time each.
# One GPU: all of them, each with its own bounded cpu_baseline.  N > 1: the two configs BASELINE defines as
# sharded over the node (configs[3], configs[4]); the CPU leg stays on the N = 1 line.
LEGS_ONE_GPU = ("c3", "c3m", "c4", "c5")
LEGS_SHARDED = ("c4", "c5")
LEG_STEPS, LEG_WARMUP = 6, 1
LAST_CLOCK_MHZ = [None]     # the clock this rank's GPU held in the last measured region (every rank, not only rank 0)
INTERNAL_BITS = 2          # YMC_CHAIN_HANDOFF (include/ymc.h): an internal error, never a property of the problem


def flagged_by_bit(status):
    status = np.asarray(status)
    return {"nonfinite": int(((status & 1) != 0).sum()), "internal": int(((status & INTERNAL_BITS) != 0).sum()),
            "other": int(((status & ~np.uint32(3)) != 0).sum())}


def measure(name, args, ffi, ctx, lowering, info, rank, world, comm, steps, warmup, headline):
    """One workload on this rank's GPU: W warm-up steps, then exactly `steps` timed steps between barriers
    (max over ranks), the kernel time from HIP events on the kernel's stream, the gather of the per-chain
    summaries.  -> (result dict on rank 0 / None elsewhere, workload)"""
    wl = make_workload(name, ffi, ctx, lowering)
    K = (args.chains if headline else 0) or wl["K"]
    T = (args.transitions if headline else 0) or wl["T"]
    L = (args.launches_per_step if headline else 0) or wl["L"]
    d, x0 = wl["d"], wl["x0"]
    no_moments = args.no_moments or name in MFMA_WORKLOADS   # the tile kernels keep no Welford state (the
    #                                                                 reference's default diagnostics)
    stride = max(args.record_stride, 1) if headline else 1
    flags = ffi.RECORD_TRAJECTORY | (ffi.RECORD_ACCEPT if stride == 1 else 0) | (0 if no_moments else ffi.RECORD_MOMENTS)
    # rows per launch: T // stride, or one more when the launch boundaries do not fall on multiples of the stride
    run = ffi.Run(wl["problem"], 20250103, K, x0, T // stride + 1, flags, chain_offset=rank * K, record_stride=stride)

    def step():
        for _ in range(L):
            run.rewind_record()
            run.advance(T)

    def barrier():
        run.sync()
        if comm is not None:
            comm.barrier()

    t_w = time.perf_counter()
    for _ in range(warmup):
        step()
    run.sync()
    warm_s = time.perf_counter() - t_w
    run.kernel_time_ms()  # drop the warm-up launches from the kernel-time average

    # the clock the chip holds DURING the timed region: a one-wave probe next to the chain kernels for
    # ~80 % of the region's expected length (from the warm-up; without one, a calibration step)
    if warmup > 0:
        expect_s = warm_s / warmup * steps
    else:
        t_c = time.perf_counter()
        step()
        run.sync()
        expect_s = (time.perf_counter() - t_c) * steps
        run.kernel_time_ms()
    probe_ms = min(max(0.8 * expect_s * 1e3, 1.0), 20000.0)

    # "widek" keeps every CU busy with ONE resident workgroup of a group that meets once per transition: the probe's wave
    # would share a CU with one of them and hold its whole group back (24 against 18 us per transition) -- not probed
    probed = name != "widek"
    barrier()
    t0 = time.perf_counter()
    if probed:
        ctx.clock_probe_start(probe_ms)
    for _ in range(steps):
        step()
    run.sync()
    elapsed = time.perf_counter() - t0
    if comm is not None:
        elapsed = comm.allreduce_max(elapsed)
    kernel_ms, launches = run.kernel_time_ms()
    clock_mhz, probe_ran_ms = ctx.clock_probe_read() if probed else (None, None)
    LAST_CLOCK_MHZ[0] = clock_mhz

    # the single collective of a run: gather the per-chain summaries
    gather_ms = None
    if comm is not None:
        tg = time.perf_counter()
        packed = comm.allgather_summary(run)  # (world, K, 3d+5) u64
        gather_ms = (time.perf_counter() - tg) * 1e3
        summary = distributed_mod().unpack_summaries(packed, d)
    else:
        summary = run.summary()
    total_steps = run.steps_done()[0]
    close = getattr(run, "close", None)
    if close is not None:
        close()            # the record (C3: 27 GB) goes back before the next workload allocates its own
    if rank != 0:
        return None, wl
    pooled = distributed_mod().pooled_diagnostics(summary, total_steps)
    value = float(world) * K * T * L * steps / elapsed
    roofline = roofline_block(name, stride, wl, K, T, kernel_ms, launches, clock_mhz, info)
    res = dict(value=value, elapsed=elapsed, K=K, T=T, L=L, d=d, stride=stride, no_moments=no_moments,
               clock_mhz=clock_mhz, probe_ran_ms=probe_ran_ms, roofline=roofline, kernel_ms=kernel_ms,
               pooled=pooled, gather_ms=gather_ms, flagged=flagged_by_bit(summary["status"]), steps=steps, warmup=warmup)
    return res, wl


def distributed_mod():
    from yagremcmc_amd import distributed
    return distributed


def leg_block(name, res, wl, cpu):
    """The compact record of a leg inside the one JSON line."""
    r = res["roofline"]
    out = {"metric": "MH steps/sec (whole node)", "value": res["value"], "unit": "steps/s",
           "ms_per_step": res["elapsed"] / res["steps"] * 1e3, "steps": res["steps"], "warmup": res["warmup"],
           "timed_region_s": res["elapsed"], "dtype": "f64",
           "config": {"workload": wl["desc"], "chains_per_gpu": res["K"], "transitions_per_launch": res["T"],
                      "launches_per_step": res["L"]},
           "roofline": {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms_avg",
                                              "launches_timed", "hbm_gb_per_s")},
           "sustained_clock_mhz": res["clock_mhz"], "acceptance_rate": res["pooled"]["acceptance_rate"],
           "chains_flagged": res["flagged"], "gather_ms": res["gather_ms"], "cpu_baseline": cpu}
    if name == "c5":
        out["coarse_sub_steps_per_s"] = res["value"] * 6
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as ONE plain process: become the launcher of N fresh ranks (nothing here has touched the GPU yet)
        from yagremcmc_amd import distributed
        sys.exit(distributed.spawn_local_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world           # under a launcher the launch decides

    from yagremcmc_amd import _ffi as ffi, distributed, lowering
    device = local_rank if world > 1 else 0
    if world > 1 and ffi.device_count() == 1 and any(os.environ.get(v) for v in
                                                       ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")):
        device = 0  # a launcher that gives every rank its own single visible GPU (RCCL refuses a shared one loudly)
    if os.environ.get("YMC_BENCH_ALLOW_SHARED_GPU") == "1":
        device %= ffi.device_count()  # rehearsal of the N > 1 flow on a box with fewer GPUs than ranks
    # N > 1: this rank's host process onto the cores of ITS GPU's NUMA node, before anything touches the GPU; the
    # launch says once per rank which device / node / cores / power cap it got (stderr)
    placement = None
    if world > 1:
        placement = distributed.pin_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))), device)
        placement["rank"] = rank
        print("bench.py: " + distributed.describe_rank(placement), file=sys.stderr, flush=True)
    ctx = ffi.Context(device)
    info = ctx.info()
    if args.workload == "iat":
        wl = make_workload("iat", ffi, ctx, lowering)
        return bench_iat(args, ffi, ctx, wl, args.chains or wl["K"], info)

    # YMC_BENCH_FORCE_COMM=1 exercises the RCCL path with a single rank (1-GPU rehearsal of the N > 1 flow)
    use_comm = world > 1 or os.environ.get("YMC_BENCH_FORCE_COMM") == "1"
    comm = None
    if use_comm:
        try:
            comm = distributed.make_communicator(ctx, rank, world)
        except distributed.CommError as e:
            print(f"bench.py: rank {rank}: {e}", file=sys.stderr, flush=True)
            sys.exit(1)

    res, wl = measure(args.workload, args, ffi, ctx, lowering, info, rank, world, comm, args.steps, args.warmup, True)
    local_clock_mhz = LAST_CLOCK_MHZ[0]
    # every rank's placement and the clock ITS GPU held during the headline region: a drop under the node's power
    # envelope shows per GPU (rank 0 alone would hide it)
    per_rank = None
    if comm is not None and getattr(comm, "store", None) is not None and placement is not None:
        mine = dict(placement, sustained_clock_mhz=local_clock_mhz)
        per_rank = [json.loads(p.decode()) for p in comm.store.allgather(json.dumps(mine).encode())]
    want_cpu = world == 1 and not args.no_cpu_baseline
    cpu = None
    if rank == 0 and want_cpu and not wl["spec"].get("aem"):
        cpu = cpu_baseline(wl["spec"], wl["x0"], CPU_SECONDS)

    # the other BASELINE configs, driver-timed in the same line (only next to the default headline)
    legs = {}
    leg_names = ()
    if args.legs == "auto":
        untouched = args.workload == "c2" and not (args.chains or args.transitions or args.launches_per_step
                                                   or args.record_stride > 1)
        leg_names = (LEGS_ONE_GPU if world == 1 else LEGS_SHARDED) if untouched else ()
    elif args.legs != "none":
        leg_names = tuple(n for n in args.legs.split(",") if n)
    for name in leg_names:
        lres, lwl = measure(name, args, ffi, ctx, lowering, info, rank, world, comm, LEG_STEPS, LEG_WARMUP, False)
        if rank == 0:
            lcpu = cpu_baseline(lwl["spec"], lwl["x0"], LEG_CPU_SECONDS) if want_cpu and not lwl["spec"].get("aem") else None
            legs[name] = leg_block(name, lres, lwl, lcpu)

    nranks_seen = backend = None
    if comm is not None:
        nranks_seen, backend = comm.count(), comm.backend
        comm.close()
    if rank != 0:
        return

    roofline = res["roofline"]
    K, T, L, stride = res["K"], res["T"], res["L"], res["stride"]
    out = {
        "metric": "MH steps/sec (whole node)", "value": res["value"], "unit": "steps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["elapsed"] / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": wl["desc"], "chains_per_gpu": K, "chains_total": world * K,
                   "transitions_per_launch": T, "launches_per_step": L, "transitions_per_step": T * L,
                   "record_stride": stride,
                   "record": "trajectory f64 [T][K][d] + accept u8 [T][K]"
                   + ("" if res["no_moments"] else " + Welford moments") + ", refilled by every launch",
                   "parallelism": f"chains sharded over {world} GPU(s)",
                   "rng": "Philox4x32-10, seed 20250103, keyed by global chain id"},
        "timed_region_s": res["elapsed"],
        "sustained_clock_mhz": res["clock_mhz"],
        "clock_probe_ms": res["probe_ran_ms"],
        "roofline": roofline,
        "hbm_gb_per_s_whole_node": world * roofline["hbm_gb_per_s"],
        "kernel_steps_per_s_per_gpu": K * T / (res["kernel_ms"] * 1e-3) if res["kernel_ms"] > 0 else 0.0,
        "acceptance_rate": res["pooled"]["acceptance_rate"],
        "pooled_mean": None if res["no_moments"] else [float(v) for v in res["pooled"]["mean"]],
        "chains_flagged_nonfinite": res["flagged"]["nonfinite"],
        "chains_flagged": res["flagged"],
        "gather_ms": res["gather_ms"],
        "collective": backend,
        "nranks_seen": nranks_seen,
        "device": info,
    }
    if per_rank is not None:
        out["ranks"] = per_rank
    if args.workload == "c5":
        out["coarse_sub_steps_per_s"] = res["value"] * 6
    if cpu is not None:
        out["cpu_baseline"] = cpu
    if legs:
        out["workloads"] = legs
    print(json.dumps(out))
    internal = res["flagged"]["internal"] + sum(v["chains_flagged"]["internal"] for v in legs.values())
    if internal:
        print(f"bench.py: {internal} chains carry an internal device error (YMC_CHAIN_HANDOFF): the trajectories are "
              f"incomplete", file=sys.stderr, flush=True)
        sys.exit(1)


if __name__ == "__main__":
    main()
