"""How `bench.py --gpus N` gets its N ranks (VERDICT r2 "multi-GPU first contact"): started as one
plain process it spawns them itself, under a launcher it joins them; rank 0's store never depends on MASTER_ADDR
resolving.  The device is stubbed (tests/helpers/bench_stub.py runs the real bench.main() over the CPU checker)
and the communicator is the explicit YMC_COMM=tcp rehearsal one -- RCCL needs the GPUs."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "helpers", "bench_stub.py")
LAUNCH_VARS = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "YMC_LAUNCH_SECRET",
               "TORCHELASTIC_RUN_ID", "YMC_STORE_BIND", "YMC_STORE_PORT")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in LAUNCH_VARS}
    env.update(YMC_COMM="tcp", YMC_COMM_TIMEOUT="60")
    env.update(kw)
    return env


def _expected(total, steps):
    """the pooled acceptance rate of `total` chains after `steps` transitions, one process"""
    from oracle import oracle as o
    sys.path.insert(0, ROOT)
    import bench
    U, c = bench.whitening(bench.TGT_MEAN, bench.TGT_COV)
    e = o.Ensemble(o.gauss_whitened(bench.TGT_MEAN, U, c), np.ones(2), np.tile([-8.0, -7.0], (total, 1)), 20250103)
    e.advance(steps, record=False)
    return float(e.acc_count.astype(np.float64).sum() / (total * steps))


def _check_line(stdout, world, steps=2, warmup=1):
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, stdout            # ONE line, from rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["collective"] == "tcp" and out["nranks_seen"] == world
    assert out["config"]["chains_total"] == 32 * world and out["steps"] == steps and out["warmup"] == warmup
    assert out["value"] > 0 and out["scaling"] == "weak" and out["gather_ms"] is not None
    # the gather really carried every rank's chains: pooled over W shards == one run over all chains
    assert out["acceptance_rate"] == _expected(32 * world, 40 * (steps + warmup))
    # every rank reports where it ran and the clock ITS device held (here: the stub's 2400 MHz, no topology to pin to)
    assert [r["rank"] for r in out["ranks"]] == list(range(world))
    assert all(r["sustained_clock_mhz"] == 2400.0 and "numa_node" in r and "cpus" in r for r in out["ranks"])
    return out


@pytest.mark.parametrize("world", [2, 4])
def test_plain_process_spawns_its_ranks(world):
    """`python bench.py --gpus N` with no launcher environment: N fresh children, one JSON line, exit 0."""
    r = subprocess.run([sys.executable, STUB, "--gpus", str(world), "--steps", "2", "--warmup", "1"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    _check_line(r.stdout, world)


def test_one_gpu_needs_no_launch():
    r = subprocess.run([sys.executable, STUB, "--gpus", "1", "--steps", "2", "--warmup", "1"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip())
    assert out["n_gpus"] == 1 and out["collective"] is None and out["acceptance_rate"] == _expected(32, 120)


LEG_KEYS = {"value", "ms_per_step", "roofline", "cpu_baseline", "config", "chains_flagged", "steps", "warmup",
            "timed_region_s", "acceptance_rate"}


def test_one_gpu_line_carries_every_baseline_config():
    """after the unchanged C2 headline the one line holds a driver-timed block per other
    BASELINE config, each with its roofline and its own bounded cpu_baseline."""
    r = subprocess.run([sys.executable, STUB, "--gpus", "1", "--steps", "2", "--warmup", "1"],
                       env=_env(YMC_STUB_CPU_BASELINE="1", YMC_BENCH_CPU_SCALE="0.02"), capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip())
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["value"] > 0
    assert list(out["workloads"]) == ["c3", "c3m", "c4", "c5"]
    for name, blk in out["workloads"].items():
        assert LEG_KEYS <= set(blk), (name, sorted(blk))
        assert blk["value"] > 0 and blk["steps"] == 6 and blk["warmup"] == 1
        assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(blk["roofline"])
        assert blk["cpu_baseline"]["value"] > 0 and blk["cpu_baseline"]["cores"] >= 1
        assert blk["chains_flagged"] == {"nonfinite": 0, "internal": 0, "other": 0}
    assert out["chains_flagged"] == {"nonfinite": 0, "internal": 0, "other": 0}
    # a non-default headline is a kernel experiment: no legs ride along; --legs picks them by name
    r = subprocess.run([sys.executable, STUB, "--gpus", "1", "--steps", "1", "--warmup", "0", "--chains", "16"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "workloads" not in json.loads(r.stdout.strip())
    r = subprocess.run([sys.executable, STUB, "--gpus", "1", "--steps", "1", "--warmup", "0", "--legs", "c5"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and list(json.loads(r.stdout.strip())["workloads"]) == ["c5"]


def test_sharded_launch_carries_the_node_wide_configs():
    r = subprocess.run([sys.executable, STUB, "--gpus", "2", "--steps", "2", "--warmup", "1"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _check_line(r.stdout, 2)
    assert list(out["workloads"]) == ["c4", "c5"]
    for blk in out["workloads"].values():
        assert blk["value"] > 0 and blk["gather_ms"] is not None and blk["cpu_baseline"] is None
        assert blk["acceptance_rate"] == _expected(64, 40 * 7)


def test_an_internal_device_error_fails_the_bench():
    """YMC_CHAIN_HANDOFF on any chain: the line says so by bit and the exit code is 1."""
    r = subprocess.run([sys.executable, STUB, "--gpus", "1", "--steps", "1", "--warmup", "0", "--legs", "none"],
                       env=_env(YMC_STUB_STATUS_BITS="2"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "internal device error" in r.stderr
    out = json.loads(r.stdout.strip())
    assert out["chains_flagged"]["internal"] == 32 and out["chains_flagged_nonfinite"] == 0


def test_a_failing_rank_fails_the_launch():
    """Rank 1 dies before the rendezvous: rank 0 gives up at the store's timeout, the parent exits non-zero and
    prints no result line."""
    r = subprocess.run([sys.executable, STUB, "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=_env(YMC_STUB_FAIL_RANK="1", YMC_COMM_TIMEOUT="3"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not r.stdout.strip()
    assert "rank 1 exited 9" in r.stderr


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        p = s.getsockname()[1]
    return p if p + 33 <= 65535 else 29500


def _launch_by_env(world, master_addr, extra=None):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = _env(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR=master_addr,
                   MASTER_PORT=str(port), **(extra or {}))
        procs.append(subprocess.Popen([sys.executable, STUB, "--gpus", str(world), "--steps", "2", "--warmup", "1"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    return [p.returncode for p in procs], outs


def test_unresolvable_master_addr():
    """The GPU boxes' hostname (`runc`) does not resolve: a launcher that exports it as MASTER_ADDR must not end
    the run (round 2: `[Errno -2] Name or service not known` on rank 0, a timeout on rank 1)."""
    assert not socket.gethostname() == "no-such-host.invalid"
    codes, outs = _launch_by_env(2, "no-such-host.invalid")
    assert codes == [0, 0], [o[1][-1500:] for o in outs]
    _check_line(outs[0][0], 2)
    assert not outs[1][0].strip()


def test_master_addr_localhost_and_all_interfaces():
    """MASTER_ADDR=localhost (what --standalone exports), and a store bound to every interface reached through
    loopback."""
    codes, outs = _launch_by_env(2, "localhost")
    assert codes == [0, 0], [o[1][-1500:] for o in outs]
    _check_line(outs[0][0], 2)
    codes, outs = _launch_by_env(2, "127.0.0.1", extra=dict(YMC_STORE_BIND="0.0.0.0"))
    assert codes == [0, 0], [o[1][-1500:] for o in outs]
    _check_line(outs[0][0], 2)


def test_store_addresses(monkeypatch):
    from yagremcmc_amd import distributed as d
    for v in ("LOCAL_WORLD_SIZE", "YMC_STORE_BIND"):
        monkeypatch.delenv(v, raising=False)
    assert d.store_addresses("no-such-host.invalid", 8) == ("127.0.0.1", ["127.0.0.1"])
    assert d.store_addresses("127.0.0.1", 8) == ("127.0.0.1", ["127.0.0.1"])
    bind, tries = d.store_addresses("localhost", 8)
    assert bind == "127.0.0.1" and tries[0] == "127.0.0.1"
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")          # torchrun on one node, whatever the master address is
    assert d.store_addresses("10.1.2.3", 8) == ("127.0.0.1", ["127.0.0.1", "10.1.2.3"])
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")          # two nodes: every interface, the master first
    assert d.store_addresses("10.1.2.3", 8) == ("0.0.0.0", ["10.1.2.3", "127.0.0.1"])
    p = d.free_master_port()
    assert 1024 <= p and p + 32 <= 65535


def test_torchrun_standalone():
    """The driver's launch line shape (`python -m torch.distributed.run ... bench.py --gpus N`), with --standalone
    (MASTER_ADDR = localhost, a random run id) as the verdict's rehearsal did."""
    pytest.importorskip("torch")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--standalone", "--nnodes=1", "--nproc-per-node", "2",
                        STUB, "--gpus", "2", "--steps", "2", "--warmup", "1"], env=_env(), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    _check_line(r.stdout, 2)


def test_store_survives_strangers():
    """A port scan / a stray connect / a bogus header must not stall or end the launch's collectives:
    silent connections, garbage hellos and an absurd length header while two ranks run collectives."""
    import struct
    import threading
    import time
    from yagremcmc_amd import distributed as d
    port = _free_port()
    res, errs = [None, None], []

    def rank(r):
        try:
            st = d.TcpStore(r, 2, "127.0.0.1", port, timeout=20)
            try:
                t0 = time.monotonic()
                res[r] = ([st.allgather(b"x%d" % r) for _ in range(20)], time.monotonic() - t0)
            finally:
                st.close()
        except Exception as e:  # noqa: BLE001
            errs.append((r, e))

    t0 = threading.Thread(target=rank, args=(0,))
    t0.start()
    time.sleep(0.3)
    strangers = []
    for payload in (b"", b"GET / HTTP/1.0\r\n\r\n", d._HELLO.pack(d._MAGIC, b"k" * 16, 1, 2), b"\xff" * 7):
        s = socket.create_connection(("127.0.0.1", port + 1), timeout=2)
        if payload:
            s.sendall(payload)
        strangers.append(s)
    t1 = threading.Thread(target=rank, args=(1,))
    t1.start()
    t0.join(60)
    t1.join(60)
    for s in strangers:
        s.close()
    assert not errs, errs
    assert res[0][0] == res[1][0] == [[b"x0", b"x1"]] * 20
    assert res[0][1] < 5.0          # nobody waited on a stranger's hello
    # a greeted rank announcing more than the cap is dropped, the server stays up
    key = d._launch_key(2, port)
    srv = d.TcpStore(0, 2, "127.0.0.1", port, timeout=5)
    try:
        s = socket.create_connection(("127.0.0.1", srv.port), timeout=2)
        s.sendall(d._HELLO.pack(d._MAGIC, key, 1, 2))
        assert d._recv_exact(s, d._WELCOME.size)[:8] == d._MAGIC_REPLY
        s.sendall(struct.pack("<QQ", 0, 1 << 40))
        s.settimeout(3)
        assert s.recv(1) == b""     # closed by the server
        s.close()
        assert srv.server.is_alive()
    finally:
        srv.sock.close()
        srv.server.stop.set()
