"""The multi-GPU partition at BASELINE's FULL totals, on one GPU: the ensemble of configs[3]
(Lotka-Volterra, 131 072 chains) and configs[4] (two-level, 1 048 576 chains) is run once as ONE run and
once as the 8 shards the 8-GPU launch makes (chain_offset = r K / 8, same seed), and everything a rank
would contribute must be identical: trajectories, accept masks, the per-chain summaries as
pack_summary_kernel packs them (the payload of the RCCL gather), and the pooled diagnostics computed by
`distributed.unpack_summaries` / `pooled_diagnostics` from 8 packed shards vs from the one run.
(contiguous blocks of global chain ids, RNG keyed by the global id.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ffi():
    from yagremcmc_amd import _ffi
    assert _ffi.device_count() >= 1
    return _ffi


@pytest.fixture(scope="module")
def ctx(ffi):
    return ffi.default_context(0)


def _whitening(mean, cov):
    w, v = np.linalg.eigh(cov)
    return v * np.sqrt(1.0 / w), float(mean.size * np.log(2.0 * np.pi) + np.sum(np.log(w)))


def _gauss(ffi, mean, cov):
    from yagremcmc_amd import lowering
    U, c = _whitening(mean, cov)
    low = lowering.Lowered(ffi.GAUSS_WHITENED, mean.size)
    low.set_array("mean", mean)
    low.set_array("whiten", U)
    low.desc.log_norm = c
    return low


def _one_vs_shards(ffi, prob, K, T, x0, flags, seed, world=8, sub=False):
    """-> nothing; asserts that `world` shard runs reproduce the single run piece by piece."""
    from yagremcmc_amd import distributed
    d = prob.dim
    per_chain = x0.shape[0] == K
    one = ffi.Run(prob, seed, K, x0, T, flags)
    one.advance(T)
    traj, acc = one.trajectory(), one.accept()
    sub_acc = one.sub_accept() if sub else None
    packed_one = one.packed_summary()
    s_one, (x_one, lp_one) = one.summary(), one.state()
    # the packed record IS the summary (word layout of pack_summary_kernel)
    u = distributed.unpack_summaries(packed_one[None], d)
    assert np.array_equal(u["state"], x_one) and np.array_equal(u["logpost"], lp_one)
    assert np.array_equal(u["accepted"], s_one["accepted"]) and np.array_equal(u["wf_mean"], s_one["wf_mean"])
    assert np.array_equal(u["wf_m2"], s_one["wf_m2"]) and np.array_equal(u["status"], s_one["status"])
    del one

    packed = []
    for r in range(world):
        off, k = distributed.shard(K, r, world)
        part = ffi.Run(prob, seed, k, x0[off:off + k] if per_chain else x0, T, flags, chain_offset=off)
        part.advance(T)
        assert np.array_equal(part.trajectory(), traj[:, off:off + k]), f"trajectory of shard {r}"
        assert np.array_equal(part.accept(), acc[:, off:off + k]), f"accept mask of shard {r}"
        if sub:
            assert np.array_equal(part.sub_accept(), sub_acc[:, :, off:off + k]), f"sub-chain decisions of shard {r}"
        packed.append(part.packed_summary())
        del part
    packed = np.stack(packed)                                   # what ncclAllGather delivers: (W, K/W, 3d+5)
    assert np.array_equal(packed.reshape(K, -1), packed_one)
    pooled8 = distributed.pooled_diagnostics(distributed.unpack_summaries(packed, d), T)
    pooled1 = distributed.pooled_diagnostics(distributed.unpack_summaries(packed_one[None], d), T)
    assert pooled8["acceptance_rate"] == pooled1["acceptance_rate"] and pooled8["n"] == pooled1["n"] == K * T
    assert np.array_equal(pooled8["mean"], pooled1["mean"])
    assert np.array_equal(pooled8["marginal_variance"], pooled1["marginal_variance"])
    assert pooled8["flagged"] == pooled1["flagged"]
    # and the pooled moments are the moments of the recorded pre-transition states
    prev = np.concatenate([np.broadcast_to(x0, (K, d))[None], traj[:-1]])
    assert np.allclose(pooled1["mean"], prev.reshape(-1, d).mean(axis=0), rtol=0, atol=1e-9)
    assert abs(pooled1["acceptance_rate"] - acc.mean()) < 1e-12
    return pooled1


def test_c4_total_131072_chains_one_run_equals_8_shards(ffi, ctx):
    """BASELINE configs[3]: Lotka-Volterra, nData 10, RK4 x 50, 131 072 chains = 8 x 16 384.  The single run
    uses the lane-per-chain kernel, the shards the row-parallel one (a speed choice by ensemble size): the
    comparison therefore also pins the two kernels against each other at full size."""
    from yagremcmc_amd import lowering
    K, T = 131072, 24
    design = np.random.default_rng(1111).uniform(0.5, 1.5, size=(10, 2))
    noise_prec, prior_mean, prior_prec = np.reciprocal(np.full(2, 0.04)), np.zeros(2), np.reciprocal(np.full(2, 1.4))

    def lowered(data):
        low = lowering.Lowered(ffi.POSTERIOR_LV, 2)
        low.desc.n_data, low.desc.data_dim = 10, 2
        for k, v in (("data", data), ("noise_precision", noise_prec), ("prior_mean", prior_mean),
                     ("prior_precision", prior_prec), ("design", design)):
            low.set_array(k, v)
        low.desc.t_final, low.desc.alpha, low.desc.gamma, low.desc.rk4_steps = 10.0, 0.8, 0.4, 50
        return low

    sigma = np.sqrt(np.reciprocal(np.reciprocal(np.full(2, 0.15))))
    probe = lowering.make_problem(ctx, ffi.METHOD_MRW, lowered(np.zeros((10, 2))), sigma)
    data = probe.forward(np.log([0.4, 0.6])).reshape(10, 2) + 0.2 * np.random.default_rng(2222).standard_normal((10, 2))
    prob = lowering.make_problem(ctx, ffi.METHOD_MRW, lowered(data), sigma)
    flags = ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT | ffi.RECORD_MOMENTS
    pooled = _one_vs_shards(ffi, prob, K, T, np.array([[-0.6, -0.3]]), flags, 20250103)
    assert pooled["flagged"] == 0 and 0.05 < pooled["acceptance_rate"] < 0.9


def test_c5_total_1048576_chains_one_run_equals_8_shards(ffi, ctx):
    """BASELINE configs[4]: two-level delayed acceptance, S = 6, 1 048 576 chains = 8 x 131 072, dispersed
    initial states (one row per chain, so the shards also slice x0)."""
    from yagremcmc_amd import lowering
    K, T, S = 1048576, 16, 6
    mean, cov = np.array([1.0, 1.5]), np.array([[2.4, -0.5], [-0.5, 0.7]])
    fine = _gauss(ffi, mean, cov)
    coarse = _gauss(ffi, mean + np.array([0.15, -0.2]), 2.0 * cov + np.array([[0.1, 1.2], [1.2, 0.05]]))
    sigma = np.sqrt(np.reciprocal(np.reciprocal(np.full(2, 1.0))))
    prob = lowering.make_problem(ctx, ffi.METHOD_MLDA2, fine, sigma, surrogate=coarse, sub_chain_length=S)
    x0 = np.array([-8.0, -7.0]) + np.random.default_rng(3).standard_normal((K, 2))
    flags = ffi.RECORD_TRAJECTORY | ffi.RECORD_ACCEPT | ffi.RECORD_SUB_ACCEPT | ffi.RECORD_MOMENTS
    pooled = _one_vs_shards(ffi, prob, K, T, x0, flags, 20250103, sub=True)
    assert pooled["flagged"] == 0 and 0.2 < pooled["acceptance_rate"] < 0.95


def _launch_bench(world, port, extra_env, *args):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, YMC_BENCH_ALLOW_SHARED_GPU="1", **extra_env)
    env.pop("YMC_COMM", None) if "YMC_COMM" not in extra_env else None
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(world),
           "--steps", "2", "--warmup", "1", "--chains", "8192", "--transitions", "200", "--launches-per-step", "2", *args]
    return subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)


def test_bench_under_the_drivers_launcher_two_ranks(ffi):
    """bench.py exactly as the driver starts it for N > 1 (`python -m torch.distributed.run ...`), two ranks on
    this box's one GPU.  With the explicit rehearsal transport the ranks meet over the TCP store (next to the
    launcher's own store on MASTER_PORT), shard by global chain id and gather; WITHOUT it RCCL is mandatory,
    RCCL refuses two ranks on one device, and every rank exits non-zero with the reason -- no line is printed."""
    import json
    import socket
    def free_port():
        # the store probes MASTER_PORT + 1 .. + 32: keep all of them valid port numbers
        for _ in range(64):
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                p = s_.getsockname()[1]
            if p + 33 <= 65535:
                return p
        return 29500
    port = free_port()
    r = _launch_bench(2, port, {"YMC_COMM": "tcp"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["collective"] == "tcp" and line["nranks_seen"] == 2
    assert line["config"]["chains_total"] == 16384 and line["value"] > 0 and line["scaling"] == "weak"
    assert 0.3 < line["acceptance_rate"] < 0.8 and line["chains_flagged_nonfinite"] == 0
    r = _launch_bench(2, free_port(), {"YMC_COMM_TIMEOUT": "60"})
    assert r.returncode != 0
    assert "RCCL is required for more than one rank" in r.stderr and "ncclCommInitRank" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_clock_probe_reports_a_plausible_shader_clock(ffi, ctx):
    ctx.clock_probe_start(30.0)
    mhz, ms = ctx.clock_probe_read()
    assert 25.0 <= ms < 200.0
    assert 100.0 < mhz < 3000.0, mhz
    with pytest.raises(ffi.YmcError):
        ctx.clock_probe_read()          # nothing pending
    with pytest.raises(ffi.YmcError):
        ctx.clock_probe_start(0.0)
