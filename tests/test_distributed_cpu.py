"""The N > 1 path on the CPU: two gloo ranks each advance their shard of the
ensemble (with the oracle standing in for the device, which this container
lacks), pack per-chain summaries in the device's word layout, all-gather them
and pool the diagnostics with the product's host code (`distributed.py`).  The
result must equal the single-rank run over all chains."""
import os
import socket
import sys
import time

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


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard():
    from yagremcmc_amd import distributed
    assert [distributed.shard(1024, r, 4) for r in range(4)] == [(0, 256), (256, 256), (512, 256), (768, 256)]
    with pytest.raises(ValueError):
        distributed.shard(10, 0, 3)
    with pytest.raises(ValueError):
        distributed.shard(8, 2, 2)


def _store_ranks(world, port, body, **kw):
    """run body(store) on `world` ranks (threads; every TcpStore is its own client, rank 0 also serves)"""
    import threading
    from yagremcmc_amd import distributed
    out, errs = [None] * world, []

    def one(r):
        try:
            st = distributed.TcpStore(r, world, "127.0.0.1", port, **kw)
            try:
                out[r] = body(st)
            finally:
                st.close()
        except Exception as e:  # noqa: BLE001
            errs.append((r, e))

    ts = [threading.Thread(target=one, args=(r,)) for r in range(world)]
    for t in reversed(ts):  # rank 0 last: the others must wait for it
        t.start()
        time.sleep(0.05)
    for t in ts:
        t.join(60)
    assert not errs, errs
    return out


def test_tcp_store_allgather_and_broadcast():
    """The store that carries the RCCL unique id: rank 0 serves next to MASTER_PORT (which the
    launcher's own store may hold), every rank gets every part in rank order, large payloads included."""
    port = _free_port()
    with socket.socket() as holder:  # somebody (torchrun's c10d store) owns MASTER_PORT itself
        holder.bind(("127.0.0.1", port))
        holder.listen(1)

        def body(st):
            a = st.allgather(bytes([st.rank]) * (st.rank + 1))
            b = st.broadcast(b"I" * 128 if st.rank == 0 else b"")
            big = st.allgather(np.full(200_000, st.rank, dtype=np.uint64).tobytes())
            return a, b, [np.frombuffer(x, dtype=np.uint64)[[0, -1]].tolist() for x in big], st.nonce, st.port
        res = _store_ranks(3, port, body, timeout=20)
    for a, b, big, nonce, p in res:
        assert a == [b"\0", b"\1\1", b"\2\2\2"] and b == b"I" * 128
        assert big == [[0, 0], [1, 1], [2, 2]]
        assert nonce == res[0][3] and p == res[0][4] and p != port


def test_tcp_store_eight_ranks():
    """The size of the 8-GPU launch: unique-id broadcast, outcome agreement, max-over-ranks, gather of
    per-rank payloads of the C5 summary size (131 072 chains x 11 words), in rank order."""
    import struct
    port = _free_port()

    def body(st):
        uid = st.broadcast(bytes(range(128)) if st.rank == 0 else b"")
        ok = st.allgather(b"ok")
        tmax = max(struct.unpack("<d", p)[0] for p in st.allgather(struct.pack("<d", 1.0 + st.rank)))
        mine = np.full(131072 * 11, st.rank, dtype=np.uint64)
        parts = st.allgather(mine.tobytes())
        return uid, ok, tmax, [int(np.frombuffer(p, dtype=np.uint64)[-1]) for p in parts], len(parts[3])
    res = _store_ranks(8, port, body, timeout=60)
    for uid, ok, tmax, last, nbytes in res:
        assert uid == bytes(range(128)) and ok == [b"ok"] * 8 and tmax == 8.0
        assert last == list(range(8)) and nbytes == 131072 * 11 * 8


def test_tcp_store_back_to_back_same_key():
    """Two launches with the same MASTER_PORT and run id, one after the other (a torchrun restart, a
    second launch from the same shell): nothing of the first is visible to the second."""
    port = _free_port()
    first = _store_ranks(2, port, lambda st: (st.allgather(b"first%d" % st.rank), st.nonce), timeout=20)
    second = _store_ranks(2, port, lambda st: (st.allgather(b"second%d" % st.rank), st.nonce), timeout=20)
    assert first[0][0] == first[1][0] == [b"first0", b"first1"]
    assert second[0][0] == second[1][0] == [b"second0", b"second1"]
    assert first[0][1] != second[0][1]  # a fresh nonce per launch


def test_tcp_store_times_out_without_rank0():
    from yagremcmc_amd import distributed
    with pytest.raises(distributed.CommError, match="no store of this launch answered"):
        distributed.TcpStore(1, 2, "127.0.0.1", _free_port(), timeout=0.5)


def test_tcp_store_is_not_fooled_by_an_echo():
    """A peer that echoes the hello back -- what a TCP self-connection on a candidate port does (the client's
    ephemeral source port can equal the port it probes) -- is not a store."""
    import threading
    from yagremcmc_amd import distributed
    srv = socket.socket()
    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    srv.bind(("127.0.0.1", 0))          # the echo sits on the FIRST candidate port of the store (MASTER_PORT + 1)
    port = srv.getsockname()[1] - 1
    srv.listen(4)
    srv.settimeout(0.2)
    stop = threading.Event()

    def echo():
        while not stop.is_set():
            try:
                c, _ = srv.accept()
            except OSError:
                continue
            try:
                c.settimeout(1.0)
                c.sendall(c.recv(4096))
            except OSError:
                pass
            c.close()

    t = threading.Thread(target=echo, daemon=True)
    t.start()
    try:
        with pytest.raises(distributed.CommError, match="no store of this launch answered"):
            distributed.TcpStore(1, 2, "127.0.0.1", port, timeout=1.0)
    finally:
        stop.set()
        t.join(2)
        srv.close()


def test_tcp_store_rejects_other_launches(monkeypatch):
    """A store of ANOTHER launch on a candidate port is passed over (handshake key mismatch)."""
    from yagremcmc_amd import distributed
    port = _free_port()
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "job-a")
    a = distributed.TcpStore(0, 2, "127.0.0.1", port, timeout=5)
    try:
        monkeypatch.setenv("TORCHELASTIC_RUN_ID", "job-b")
        with pytest.raises(distributed.CommError):
            distributed.TcpStore(1, 2, "127.0.0.1", port, timeout=0.6)
    finally:
        a.sock.close()
        a.server.stop.set()


class _FakeRun:
    """summary()/state() of a device run, served by the oracle ensemble"""

    def __init__(self, e):
        self.e, self.K, self.d = e, e.K, e.d

    def summary(self):
        e = self.e
        return dict(accepted=e.acc_count, wf_n=e.wf_n, wf_mean=e.wf_mean, wf_m2=e.wf_m2, status=e.status)

    def state(self):
        return self.e.x, self.e.lp


def _tcp_comm_worker(rank, world, port, total, steps, q):
    sys.path.insert(0, ROOT)
    from oracle import oracle as o
    from yagremcmc_amd import distributed
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      YMC_COMM="tcp")
    comm = distributed.make_communicator(None, rank, world)
    assert comm.backend == "tcp" and comm.count() == world
    g = np.load(os.path.join(ROOT, "tests", "golden", "f2_mrw_2d.npz"))
    p = o.gauss_whitened(g["mean"], g["whiten"], float(g["log_norm"]))
    off, k = distributed.shard(total, rank, world)
    e = o.Ensemble(p, g["sigma"], np.tile([-8.0, -7.0], (k, 1)), 4242, chain0=off)
    e.advance(steps, record=False)
    comm.barrier()
    tmax = comm.allreduce_max(float(rank + 1))
    packed = comm.allgather_summary(_FakeRun(e))
    for _ in range(3):
        comm.barrier()
    comm.close()
    if rank == 0:
        q.put((packed, tmax))


@pytest.mark.parametrize("world", [2, 8])
def test_tcp_communicator_processes(world):
    """YMC_COMM=tcp (explicit only): the rehearsal communicator gives the device layout's gather; with 8
    processes it is the shape of the 8-GPU launch (rendezvous, barrier, max-reduction, gather, goodbye)."""
    import multiprocessing as mp
    from oracle import oracle as o
    from yagremcmc_amd import distributed
    total, steps = 64, 100
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tcp_comm_worker, args=(r, world, port, total, steps, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    packed, tmax = q.get(timeout=120)
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    assert tmax == float(world) and packed.shape == (world, total // world, 11)
    g = np.load(os.path.join(ROOT, "tests", "golden", "f2_mrw_2d.npz"))
    p = o.gauss_whitened(g["mean"], g["whiten"], float(g["log_norm"]))
    e = o.Ensemble(p, g["sigma"], np.tile([-8.0, -7.0], (total, 1)), 4242)
    e.advance(steps, record=False)
    assert np.array_equal(packed.reshape(-1, 11), _pack_like_device(e, 2))
    summary = distributed.unpack_summaries(packed, 2)
    assert np.array_equal(summary["state"], e.x) and np.array_equal(summary["accepted"], e.acc_count)


def _rccl_must_fail_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from yagremcmc_amd import distributed
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.pop("YMC_COMM", None)
    try:
        # no GPU context here: the RCCL bring-up cannot succeed on any rank
        distributed.make_communicator(None, rank, world)
    except distributed.CommError as e:
        q.put((rank, str(e)))
        sys.exit(7)
    q.put((rank, "a communicator came back"))


def test_rccl_failure_is_an_error_on_every_rank():
    """No stand-in: when RCCL cannot be brought up, every rank raises (and the bench exits non-zero)
    instead of switching to another transport."""
    import multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_must_fail_worker, args=(r, world, port, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 7
    for r in range(world):
        assert "failed" in got[r] and "RCCL is required" in got[r], got[r]
    from yagremcmc_amd import distributed
    with pytest.raises(distributed.CommError, match="'rccl' and 'tcp'"):
        os.environ["YMC_COMM"] = "file"
        try:
            distributed.make_communicator(None, 0, 1)
        finally:
            del os.environ["YMC_COMM"]


# ---------------------------------------------------------------------------
# rank -> GPU -> NUMA node -> cores: a faked sysfs
# ---------------------------------------------------------------------------

def _fake_node(root, nid, simd, minor=None, numa=None, cpulist=None, cap_uw=None, node_cpulist=None):
    d = root / "class" / "kfd" / "kfd" / "topology" / "nodes" / str(nid)
    d.mkdir(parents=True)
    (d / "properties").write_text(f"cpu_cores_count {0 if simd else 48}\nsimd_count {simd}\ndrm_render_minor {minor if minor is not None else 0}\n")
    if minor is None:
        return
    dev = root / "class" / "drm" / f"renderD{minor}" / "device"
    dev.mkdir(parents=True)
    if numa is not None:
        (dev / "numa_node").write_text(f"{numa}\n")
    if cpulist is not None:
        (dev / "local_cpulist").write_text(cpulist + "\n")
    if cap_uw is not None:
        hw = dev / "hwmon" / "hwmon3"
        hw.mkdir(parents=True)
        (hw / "power1_cap").write_text(f"{cap_uw}\n")
    if node_cpulist is not None:
        nd = root / "devices" / "system" / "node" / f"node{numa}"
        nd.mkdir(parents=True, exist_ok=True)
        (nd / "cpulist").write_text(node_cpulist + "\n")


def _eight_gpu_box(tmp_path):
    """two CPU nodes (KFD nodes 0, 1: no SIMDs), eight GPUs on render minors 128..135: four per socket, 96 cores each
    (hyper-threads as a second range), power cap 1400 W"""
    _fake_node(tmp_path, 0, 0)
    _fake_node(tmp_path, 1, 0)
    for g in range(8):
        sock = g // 4
        cl = f"{96 * sock}-{96 * sock + 95},{192 + 96 * sock}-{192 + 96 * sock + 95}"
        _fake_node(tmp_path, 2 + g, 1024, minor=128 + g, numa=sock, cpulist=cl, cap_uw=1400000000)
    return str(tmp_path)


def test_gpu_topology_from_sysfs(tmp_path, monkeypatch):
    from yagremcmc_amd import distributed as d
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    root = _eight_gpu_box(tmp_path)
    topo = d.gpu_topology(root)
    assert [g["index"] for g in topo] == list(range(8)) and [g["kfd_node"] for g in topo] == list(range(2, 10))
    assert [g["numa_node"] for g in topo] == [0] * 4 + [1] * 4
    assert topo[0]["cpus"] == set(range(0, 96)) | set(range(192, 288)) and topo[5]["power_cap_w"] == 1400.0
    # a launcher that hands the job devices 6, 2: HIP's device 0 is the box's GPU 6
    assert [g["kfd_node"] for g in d.gpu_topology(root, visible="6,2")] == [8, 4]
    assert d.gpu_topology(str(tmp_path / "nowhere")) == []


def test_ranks_are_pinned_to_their_gpus_socket_in_equal_shares(tmp_path, monkeypatch):
    from yagremcmc_amd import distributed as d
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "YMC_PIN_RANKS"):
        monkeypatch.delenv(v, raising=False)
    root = _eight_gpu_box(tmp_path)
    topo = d.gpu_topology(root)
    allowed = set(range(384))
    sets = d.rank_cpu_sets(list(range(8)), topo, allowed)
    assert all(len(s) == 48 for s in sets)                              # 192 hardware threads per socket, four ranks
    assert set().union(*sets[:4]) == topo[0]["cpus"] and set().union(*sets[4:]) == topo[4]["cpus"]
    assert all(sets[i].isdisjoint(sets[j]) for i in range(8) for j in range(i))
    # a cgroup that allows the job 64 cores of socket 0 only: socket 0's ranks share those, socket 1's stay unpinned
    sets = d.rank_cpu_sets(list(range(8)), topo, set(range(64)))
    assert [len(s) if s else None for s in sets] == [16, 16, 16, 16, None, None, None, None]
    # two ranks: one share each of the WHOLE socket's cores
    assert [len(s) for s in d.rank_cpu_sets([0, 1], topo, allowed)] == [96, 96]
    # the report of a rank (nothing applied: the test process keeps its affinity)
    monkeypatch.setattr(d.os, "sched_getaffinity", lambda pid: allowed, raising=False)
    info = d.pin_rank(5, 8, sysfs=root, apply=False)
    assert info["device"] == 5 and info["numa_node"] == 1 and info["cpus"] == "144-191" and info["power_cap_w"] == 1400.0
    assert d.describe_rank(info) == "rank 5 -> device 5 -> NUMA node 1 -> cores 144-191; power cap 1400 W"
    # a box without the topology: nothing is pinned and the line says so
    info = d.pin_rank(1, 2, sysfs=str(tmp_path / "nowhere"), apply=False)
    assert info["cpus"] is None and "not pinned" in d.describe_rank(info)
    assert d._format_cpulist(d._parse_cpulist("0-3,8,10-11")) == "0-3,8,10-11"
