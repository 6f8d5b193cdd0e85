#!/usr/bin/env python3
"""Headline benchmark: MH steps/s of the chain-ensemble hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One bench "step" = one launch of the fused propose/evaluate/accept/record
kernel advancing every chain of this rank by --transitions MH transitions
(full trajectory + accept mask + FullDiagnostics' Welford moments, i.e. what
example_mcmc_2d_singleLevel.py records).  Default workload = BASELINE.json
configs[1]: the 2-D Gaussian target, random-walk proposal, 65 536 chains per
GPU.  For N > 1 the driver launches one process per GPU (torchrun env:
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*); chains shard by global chain id,
nothing is exchanged while sampling (weak scaling), and the single collective
is an RCCL all-gather of the per-chain summaries after the timed region.
No PyTorch anywhere: the host drives libymc.so through ctypes.

Rank 0 prints ONE JSON line.  `roofline.achieved` = algorithmic bytes per
launch ((8 d + 1) B per transition per chain) / average kernel
time measured with HIP events on the kernel's stream.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_MFMA_PEAK_TFLOPS = 78.6  # AMD MI355X data sheet, FP64 matrix (the micro-architecture guide lists no f64 MFMA rate)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c5", "c4"])
    ap.add_argument("--chains", type=int, default=0, help="chains per GPU (default: the workload's)")
    ap.add_argument("--transitions", type=int, default=0, help="MH transitions per chain per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-moments", action="store_true")
    return ap.parse_args()


# ---------------------------------------------------------------------------
# synthetic workloads
# ---------------------------------------------------------------------------

TGT_MEAN = np.array([1.0, 1.5])
TGT_COV = np.array([[2.4, -0.5], [-0.5, 0.7]])


def whitening(mean, cov):
    w, v = np.linalg.eigh(cov)
    return v * np.sqrt(1.0 / w), float(mean.size * np.log(2.0 * np.pi) + np.sum(np.log(w)))


def lv_problem_arrays():
    rng = np.random.default_rng(1111)
    design = rng.uniform(0.5, 1.5, size=(10, 2))
    return design


def make_workload(name, ffi, ctx, lowering):
    """-> (problem, x0 (1,d), default chains/GPU, default transitions/step, description, d, oracle problem)"""
    from oracle import oracle  # only to describe the same problem to the cpu_baseline leg
    if name in ("c2", "c5"):
        U, c = whitening(TGT_MEAN, TGT_COV)
        fine = lowering.Lowered(ffi.GAUSS_WHITENED, 2)
        fine.set_array("mean", TGT_MEAN)
        fine.set_array("whiten", U)
        fine.desc.log_norm = c
        sigma = np.sqrt(np.reciprocal(np.reciprocal(np.full(2, 1.0))))
        x0 = np.array([[-8.0, -7.0]])
        o_fine = oracle.gauss_whitened(TGT_MEAN, U, c)
        if name == "c2":
            prob = lowering.make_problem(ctx, ffi.METHOD_MRW, fine, sigma)
            return (prob, x0, 65536, 1000, "C2: 2-D Gaussian target (example_mcmc_2d_singleLevel.py), "
                    "random-walk proposal var 1.0, x0=(-8,-7)", 2, dict(target=o_fine, sigma=sigma))
        cm = TGT_MEAN + np.array([0.15, -0.2])
        cc = 2.0 * TGT_COV + np.array([[0.1, 1.2], [1.2, 0.05]])
        Uc, c_c = whitening(cm, cc)
        coarse = lowering.Lowered(ffi.GAUSS_WHITENED, 2)
        coarse.set_array("mean", cm)
        coarse.set_array("whiten", Uc)
        coarse.desc.log_norm = c_c
        prob = lowering.make_problem(ctx, ffi.METHOD_MLDA2, fine, sigma, surrogate=coarse, sub_chain_length=6)
        return (prob, x0, 131072, 200, "C5: two-level delayed acceptance (example_mcmc_2d_twoLevel.py), S=6",
                2, dict(target=o_fine, sigma=sigma, surrogate=oracle.gauss_whitened(cm, Uc, c_c), S=6))
    if name == "c3":
        # linear forward model, d = m = 128, nData = 5
        d = 128
        rng = np.random.default_rng(7)
        A = rng.standard_normal((d, d)) / np.sqrt(d)
        b = rng.standard_normal(d)
        theta = rng.standard_normal(d)
        data = np.array([A @ theta + b + np.sqrt(0.3) * rng.standard_normal(d) for _ in range(int(os.environ.get("YMC_C3_ND", "5")))])
        noise_prec, prior_mean, prior_prec = np.reciprocal(np.full(d, 0.3)), np.zeros(d), np.reciprocal(np.full(d, 5.0))
        low = lowering.Lowered(ffi.POSTERIOR_LINEAR, d)
        low.desc.n_data, low.desc.data_dim = data.shape[0], d
        for k, v in (("data", data), ("noise_precision", noise_prec), ("prior_mean", prior_mean),
                     ("prior_precision", prior_prec), ("fwd_matrix", A), ("fwd_shift", b)):
            low.set_array(k, v)
        sigma = np.sqrt(np.reciprocal(np.reciprocal(np.full(d, 1.1e-3))))
        prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, sigma)
        o_prob = oracle.posterior_linear(A, b, data, noise_prec, prior_mean, prior_prec)
        return (prob, np.zeros((1, d)), 262144, 20, "C3: linear forward model posterior d = m = 128, nData = 5 "
                "(fp64 MFMA tile kernel)", d, dict(target=o_prob, sigma=sigma))
    # c4: Lotka-Volterra, nData=10, T=10, RK4 x 50
    design = lv_problem_arrays()
    truth = np.log(np.array([0.4, 0.6]))
    noise_prec, prior_mean, prior_prec = np.reciprocal(np.full(2, 0.04)), np.zeros(2), np.reciprocal(np.full(2, 1.4))
    o_prob = oracle.posterior_lv(design, 10.0, 0.8, 0.4, 50, np.zeros((10, 2)), noise_prec, prior_mean, prior_prec)
    clean = oracle.forward(o_prob, truth).reshape(10, 2)
    data = clean + np.sqrt(0.04) * np.random.default_rng(2222).standard_normal((10, 2))
    o_prob = oracle.posterior_lv(design, 10.0, 0.8, 0.4, 50, data, noise_prec, prior_mean, prior_prec)
    low = lowering.Lowered(ffi.POSTERIOR_LV, 2)
    low.desc.n_data, low.desc.data_dim = 10, 2
    for k, v in (("data", data), ("noise_precision", noise_prec), ("prior_mean", prior_mean),
                 ("prior_precision", prior_prec), ("design", design)):
        low.set_array(k, v)
    low.desc.t_final, low.desc.alpha, low.desc.gamma, low.desc.rk4_steps = 10.0, 0.8, 0.4, 50
    sigma = np.sqrt(np.reciprocal(np.reciprocal(np.full(2, 0.15))))
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW, low, sigma)
    return (prob, np.array([[-0.6, -0.3]]), 16384, 100,
            "C4: Lotka-Volterra posterior, nData=10, T=10, RK4 x 50", 2, dict(target=o_prob, sigma=sigma))


def cpu_baseline(wl, d):
    """The oracle (CPU restatement, kind "port") timed on the host cores on a
    bounded sample of the same workload: same target, same recording, fewer
    chains and transitions (~10-20 s of CPU work)."""
    from oracle import oracle
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(visible, 16)     # one GPU's share of the host (the box exposes all 256 hardware threads)

    x0 = wl.get("x0")

    def ensemble(K, threads):
        return oracle.Ensemble(wl["target"], wl["sigma"], np.tile(x0, (K, 1)), 20250103,
                               surrogate=wl.get("surrogate"), sub_chain_length=wl.get("S", 0), nthreads=threads)

    def timed(K, T, threads, min_seconds):
        e = ensemble(K, threads)
        bufs = e.advance(T)          # untimed: faults the record buffers in, like the GPU warm-up
        reps, t = 0, time.perf_counter()
        while True:
            e.advance(T, reuse=bufs)
            reps += 1
            dt = time.perf_counter() - t
            if dt >= min_seconds:
                return K * T * reps / dt, reps, dt

    T = 500
    K = 1024 * cores             # 25 B per transition of record: ~100 MB per 8 cores
    rate, reps, dt = timed(K, T, cores, 10.0)
    K1 = 1024
    rate1, reps1, dt1 = timed(K1, T, 1, 4.0)
    return {"value": rate, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"{K} chains x {T} transitions x {reps} passes of the same workload (trajectory, mask, log-density and "
                      f"moments recorded, record buffers resident), {cores} OpenMP threads, {dt:.1f} s",
            "host_threads_visible": visible, "single_thread_steps_per_s": rate1,
            "reference_single_core_steps_per_s_survey_container": 2.2e4}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched as N processes (torch.distributed.run / torchrun)")
        args.gpus = world

    from yagremcmc_amd import _ffi as ffi, distributed, lowering
    device = local_rank if world > 1 else 0
    if os.environ.get("YMC_BENCH_ALLOW_SHARED_GPU") == "1":
        device %= ffi.device_count()  # rehearsal of the N > 1 flow on a box with fewer GPUs than ranks
    ctx = ffi.Context(device)
    info = ctx.info()
    prob, x0, def_chains, def_T, desc, d, wl = make_workload(args.workload, ffi, ctx, lowering)
    wl["x0"] = x0
    K = args.chains or def_chains
    T = args.transitions or def_T
    if args.workload == "c3":
        args.no_moments = True   # the reference's default diagnostics; the tile kernel keeps no Welford state
    flags = ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT | (0 if args.no_moments else ffi.RECORD_MOMENTS)
    run = ffi.Run(prob, 20250103, K, x0, T, flags, chain_offset=rank * K)

    # YMC_BENCH_FORCE_COMM=1 exercises the RCCL path with a single rank (1-GPU rehearsal of the N > 1 flow)
    use_comm = world > 1 or os.environ.get("YMC_BENCH_FORCE_COMM") == "1"
    comm = distributed.make_communicator(ctx, rank, world) if use_comm else None

    def barrier():
        run.sync()
        if comm is not None:
            comm.barrier()

    # Untimed pre-heat: a cold MI355X takes some 50-100 ms of load to reach its sustained clock (measured:
    # the same kernel runs 14 % slower during the first milliseconds of a process).  The W warm-up steps
    # of the contract follow; with small W they alone would leave the ramp inside the timed region.
    preheat_ms = float(os.environ.get("YMC_BENCH_PREHEAT_MS", "400"))
    t_heat = time.perf_counter()
    while (time.perf_counter() - t_heat) * 1e3 < preheat_ms:
        for _ in range(8):
            run.rewind_record()
            run.advance(T)
        run.sync()
    for _ in range(args.warmup):
        run.rewind_record()
        run.advance(T)
    run.sync()
    run.kernel_time_ms()  # drop pre-heat and warm-up launches from the kernel-time average

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run.rewind_record()
        run.advance(T)
    run.sync()
    elapsed = time.perf_counter() - t0
    if comm is not None:
        elapsed = comm.allreduce_max(elapsed)
    kernel_ms, launches = run.kernel_time_ms()

    # the single collective of the run: gather the per-chain summaries
    gather_ms = None
    if comm is not None:
        tg = time.perf_counter()
        packed = comm.allgather_summary(run)  # (world, K, 3d+5) u64
        gather_ms = (time.perf_counter() - tg) * 1e3
        summary = distributed.unpack_summaries(packed, d)
    else:
        summary = run.summary()
    if rank != 0:
        return
    total_steps = run.steps_done()[0]
    pooled = distributed.pooled_diagnostics(summary, total_steps)

    transitions = float(world) * K * T * args.steps
    value = transitions / elapsed
    bytes_per_transition = 8 * d + 1
    launch_bytes = K * T * bytes_per_transition
    achieved = launch_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    traffic = valu_busy = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            rec = json.load(open(tpath)).get(f"{args.workload}_K{K}_T{T}")
            traffic = rec["hbm_bytes_per_launch"] if rec else None
            sq = (rec or {}).get("sq_per_launch") or {}
            if sq.get("SQ_WAVE_CYCLES"):
                # what actually bounds these kernels: the share of wave-cycles with the VALU busy
                valu_busy = sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"]
        except Exception:
            traffic = valu_busy = None

    if args.workload == "c3":
        # the forward map is a batched 128 x 128 GEMV on the fp64 matrix cores: 2 d m flop per transition
        flops = 2.0 * d * d * K * T
        tf = flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
        roofline = {"bound": "mfma", "achieved": tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": tf / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic, "kernel_ms_avg": kernel_ms,
                    "launches_timed": launches, "algorithmic_flops_per_launch": flops,
                    "algorithmic_bytes_per_launch": launch_bytes, "hbm_gb_per_s": achieved,
                    "note": "per GPU (rank 0); v_mfma_f64_16x16x4_f64 GEMV (2 d m flop per transition); on gfx950 the "
                            "f64 MFMA shares the SIMD's fp64 datapath with the VALU, so the 128 normal draws per "
                            "transition (VALU) add to, rather than hide behind, the matrix time -- DESIGN.md"}
    else:
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "kernel_ms_avg": kernel_ms,
                    "launches_timed": launches, "algorithmic_bytes_per_launch": launch_bytes,
                    "note": "per GPU (rank 0); bytes = (8 d + 1) per transition per chain, traffic = PMC WRITE_SIZE + "
                            "2 x FETCH_SIZE of the same configuration (profiles/); the kernel is VALU-issue bound "
                            "(Philox + Box-Muller + exp per transition, one wave per SIMD at 65 536 chains), "
                            "see DESIGN.md"}
    out = {
        "metric": "MH steps/sec (whole node)", "value": value, "unit": "steps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": desc, "chains_per_gpu": K, "chains_total": world * K,
                   "transitions_per_step": T, "record": "trajectory f64 [T][K][d] + accept u8 [T][K]"
                   + ("" if args.no_moments else " + Welford moments"), "parallelism": f"chains sharded over {world} GPU(s)",
                   "rng": "Philox4x32-10, seed 20250103, keyed by global chain id"},
        "roofline": dict(roofline, valu_busy_frac_profiled=valu_busy),
        "hbm_gb_per_s_whole_node": world * achieved,
        "kernel_steps_per_s_per_gpu": K * T / (kernel_ms * 1e-3) if kernel_ms > 0 else 0.0,
        "acceptance_rate": pooled["acceptance_rate"],
        "pooled_mean": None if args.no_moments else [float(v) for v in pooled["mean"]],
        "chains_flagged_nonfinite": pooled["flagged"],
        "gather_ms": gather_ms,
        "preheat_ms": preheat_ms,
        "collective": None if comm is None else comm.backend,
        "device": info,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(wl, d)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
