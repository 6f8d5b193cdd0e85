"""The N > 1 path on the CPU: two gloo ranks each advance their shard of the
ensemble (with the oracle standing in for the device, which this container
lacks), pack per-chain summaries in the device's word layout, all-gather them
and pool the diagnostics with the product's host code (`distributed.py`).  The
result must equal the single-rank run over all chains."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pack_like_device(e, d):
    """what pack_summary_kernel writes: state, logpost, accepted, wf_n, mean, M2, status, 0"""
    K = e.K
    out = np.zeros((K, 3 * d + 5), dtype=np.uint64)
    out[:, 0:d] = e.x.view(np.uint64)
    out[:, d] = e.lp.view(np.uint64)
    out[:, d + 1] = e.acc_count
    out[:, d + 2] = e.wf_n
    out[:, d + 3:2 * d + 3] = e.wf_mean.view(np.uint64)
    out[:, 2 * d + 3:3 * d + 3] = e.wf_m2.view(np.uint64)
    out[:, 3 * d + 3] = e.status
    return out


def _worker(rank, world, port, total, steps, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import oracle as o
    from yagremcmc_amd import distributed
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(ROOT, "tests", "golden", "f2_mrw_2d.npz"))
    p = o.gauss_whitened(g["mean"], g["whiten"], float(g["log_norm"]))
    assert distributed.launch_env() == (rank, rank, world)
    off, k = distributed.shard(total, rank, world)
    e = o.Ensemble(p, g["sigma"], np.tile([-8.0, -7.0], (k, 1)), 4242, chain0=off)
    e.advance(steps, record=False)
    mine = torch.from_numpy(_pack_like_device(e, 2).astype(np.int64))
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    packed = np.stack([t.numpy().astype(np.uint64) for t in gathered])
    summary = distributed.unpack_summaries(packed, 2)
    pooled = distributed.pooled_diagnostics(summary, steps)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the bench's max-over-ranks timing reduction
    if rank == 0:
        q.put((summary, pooled, float(t[0])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one(tmp_path):
    import torch.multiprocessing as mp
    from oracle import oracle as o
    from yagremcmc_amd import distributed

    total, steps, world = 96, 300, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, steps, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    summary, pooled, tmax = q.get(timeout=120)
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    assert tmax == 2.0

    g = np.load(os.path.join(ROOT, "tests", "golden", "f2_mrw_2d.npz"))
    p = o.gauss_whitened(g["mean"], g["whiten"], float(g["log_norm"]))
    e = o.Ensemble(p, g["sigma"], np.tile([-8.0, -7.0], (total, 1)), 4242)
    e.advance(steps, record=False)
    assert np.array_equal(summary["state"], e.x) and np.array_equal(summary["logpost"], e.lp)
    assert np.array_equal(summary["accepted"], e.acc_count)
    assert np.array_equal(summary["wf_mean"], e.wf_mean) and np.array_equal(summary["wf_m2"], e.wf_m2)
    assert (summary["status"] == 0).all()
    one = distributed.pooled_diagnostics(
        dict(accepted=e.acc_count, wf_n=e.wf_n, wf_mean=e.wf_mean, wf_m2=e.wf_m2, status=e.status), steps)
    assert pooled["acceptance_rate"] == one["acceptance_rate"]
    assert np.array_equal(pooled["mean"], one["mean"]) and np.array_equal(pooled["marginal_variance"], one["marginal_variance"])


def test_shard_and_rendezvous(tmp_path, monkeypatch):
    from yagremcmc_amd import distributed
    assert [distributed.shard(1024, r, 4) for r in range(4)] == [(0, 256), (256, 256), (512, 256), (768, 256)]
    with pytest.raises(ValueError):
        distributed.shard(10, 0, 3)
    with pytest.raises(ValueError):
        distributed.shard(8, 2, 2)
    monkeypatch.setenv("MASTER_PORT", "29517")
    a, b = distributed.FileRendezvous(str(tmp_path)), distributed.FileRendezvous(str(tmp_path))
    assert a.path == b.path
    with pytest.raises(RuntimeError):
        b.wait(128, timeout=0.2)
    a.publish(b"x" * 128)
    assert b.wait(128, timeout=1.0) == b"x" * 128
    a.cleanup()
    assert not os.path.exists(a.path)


class _FakeRun:
    """summary()/state() of a device run, served by the oracle ensemble"""

    def __init__(self, e):
        self.e, self.K, self.d = e, e.K, e.d

    def summary(self):
        e = self.e
        return dict(accepted=e.acc_count, wf_n=e.wf_n, wf_mean=e.wf_mean, wf_m2=e.wf_m2, status=e.status)

    def state(self):
        return self.e.x, self.e.lp


def _file_comm_worker(rank, world, port, total, steps, q):
    sys.path.insert(0, ROOT)
    from oracle import oracle as o
    from yagremcmc_amd import distributed
    os.environ.update(MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), YMC_COMM="file")
    comm = distributed.make_communicator(None, rank, world)
    assert comm.backend == "file"
    g = np.load(os.path.join(ROOT, "tests", "golden", "f2_mrw_2d.npz"))
    p = o.gauss_whitened(g["mean"], g["whiten"], float(g["log_norm"]))
    off, k = distributed.shard(total, rank, world)
    e = o.Ensemble(p, g["sigma"], np.tile([-8.0, -7.0], (k, 1)), 4242, chain0=off)
    e.advance(steps, record=False)
    comm.barrier()
    tmax = comm.allreduce_max(float(rank + 1))
    packed = comm.allgather_summary(_FakeRun(e))
    for _ in range(3):
        comm.barrier()  # exercises the clean-up of earlier collectives
    if rank == 0:
        q.put((packed, tmax))


def test_file_communicator_two_ranks(tmp_path):
    """The stand-in used when RCCL cannot come up: same gather result as the device layout."""
    import multiprocessing as mp
    from oracle import oracle as o
    from yagremcmc_amd import distributed
    total, steps, world = 64, 100, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_file_comm_worker, args=(r, world, port, total, steps, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    packed, tmax = q.get(timeout=120)
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    assert tmax == 2.0 and packed.shape == (2, 32, 11)
    g = np.load(os.path.join(ROOT, "tests", "golden", "f2_mrw_2d.npz"))
    p = o.gauss_whitened(g["mean"], g["whiten"], float(g["log_norm"]))
    e = o.Ensemble(p, g["sigma"], np.tile([-8.0, -7.0], (total, 1)), 4242)
    e.advance(steps, record=False)
    assert np.array_equal(packed.reshape(-1, 11), _pack_like_device(e, 2))
    summary = distributed.unpack_summaries(packed, 2)
    assert np.array_equal(summary["state"], e.x) and np.array_equal(summary["accepted"], e.acc_count)
