"""Sharding of an ensemble over GPUs (one process per GPU) and the host side of
the single collective: the gather of per-chain summaries.

Nothing is exchanged while sampling: rank r of W owns the contiguous block of
global chain ids [r*K, (r+1)*K) and every chain draws from the stream of its
GLOBAL id, so per-chain results do not depend on W.  After sampling each rank
contributes its packed summaries (3d+5 eight-byte words per chain, written by
pack_summary_kernel) to one RCCL all-gather (`_ffi.Communicator`).

Bring-up (no PyTorch in these processes): the RCCL unique id travels over a
small TCP store that rank 0 serves next to MASTER_ADDR:MASTER_PORT
(`TcpStore`), so any launcher that exports RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT works -- torchrun, mpirun wrappers, a shell loop.  A failure to
bring RCCL up is an ERROR on every rank (`CommError`): there is no stand-in
that a benchmark could silently fall back to.  `YMC_COMM=tcp` selects, on
request only, a communicator over the same store (CPU rehearsals of the N > 1
flow and ranks sharing one GPU, where RCCL refuses to initialise).
"""
import hashlib
import os
import selectors
import socket
import struct
import sys
import threading
import time

import numpy as np

from .diagnostics import merge_welford


class CommError(RuntimeError):
    """The inter-rank communicator could not be brought up, or the ranks disagree."""


def launch_env():
    """(rank, local_rank, world) from the torchrun-style environment."""
    rank = int(os.environ.get("RANK", "0"))
    return rank, int(os.environ.get("LOCAL_RANK", str(rank))), int(os.environ.get("WORLD_SIZE", "1"))


def shard(total_chains, rank, world):
    """Contiguous equal blocks of global chain ids; -> (chain_offset, n_chains)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank / world size")
    if total_chains % world:
        raise ValueError(f"{total_chains} chains do not split evenly over {world} ranks")
    k = total_chains // world
    return rank * k, k


def summary_words(d):
    return 3 * d + 5


def unpack_summaries(packed, d):
    """(W, K, 3d+5) u64 -> dict of arrays over all W*K chains in global id order."""
    flat = np.ascontiguousarray(packed, dtype=np.uint64).reshape(-1, summary_words(d))
    return dict(
        state=flat[:, 0:d].view(np.float64),
        logpost=flat[:, d].view(np.float64),
        accepted=flat[:, d + 1].copy(),
        wf_n=flat[:, d + 2].copy(),
        wf_mean=flat[:, d + 3:2 * d + 3].view(np.float64),
        wf_m2=flat[:, 2 * d + 3:3 * d + 3].view(np.float64),
        status=flat[:, 3 * d + 3].astype(np.uint32),
    )


def pooled_diagnostics(summary, steps):
    """Ensemble-wide acceptance rate and Welford moments from per-chain summaries."""
    n_chains = summary["accepted"].size
    rate = float(summary["accepted"].astype(np.float64).sum() / (n_chains * steps)) if steps else 0.0
    n, mean, m2 = merge_welford(summary["wf_n"], summary["wf_mean"], summary["wf_m2"])
    var = m2 / (n - 1) if n > 1 else None
    return dict(acceptance_rate=rate, n=n, mean=mean, marginal_variance=var,
                flagged=int((summary["status"] != 0).sum()))


# ---------------------------------------------------------------------------
# the TCP store
# ---------------------------------------------------------------------------

_MAGIC = b"YMCSTOR3"           # client -> server
_MAGIC_REPLY = b"YMCSTOK3"     # server -> client: differs, so a TCP self-connection (a client whose ephemeral
#                                source port equals the candidate port it probes, which echoes its own hello
#                                back) cannot pass for a store
_PORT_SPAN = 32          # candidate ports after MASTER_PORT
_HELLO = struct.Struct("<8s16sII")     # magic, key, rank, world
_WELCOME = struct.Struct("<8s16s")     # magic, nonce of this launch
_MSG = struct.Struct("<QQ")            # collective sequence number, payload bytes
_MAX_PAYLOAD = 1 << 28   # one rank's part of a collective (C5's summaries are 11.5 MB per rank)
_HELLO_PATIENCE_S = 2.0  # an accepted connection that has not said a full hello by then is dropped
_LOOPBACK = "127.0.0.1"


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(n - len(buf), 1 << 20))
        if not chunk:
            raise ConnectionError("peer closed the store connection")
        buf += chunk
    return bytes(buf)


def _launch_key(world, port):
    """Identifies ONE launch: ranks of another job that shares the machine (and possibly a
    candidate port) never pass each other's handshake.  YMC_LAUNCH_SECRET (set to random bytes by
    `spawn_local_ranks`; torchrun's --standalone run id is random as well) makes it unguessable."""
    run_id = os.environ.get("TORCHELASTIC_RUN_ID", "none")
    restart = os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
    secret = os.environ.get("YMC_LAUNCH_SECRET", "")
    return hashlib.sha256(f"{run_id}|{restart}|{world}|{port}|{secret}".encode()).digest()[:16]


def candidate_ports(master_port):
    """Where rank 0 may serve the store.  MASTER_PORT itself is NOT used: under
    `python -m torch.distributed.run --master-port P` (static rendezvous) the launcher's own c10d
    store already listens on P, so rank 0 takes the first free port of P+1 .. P+32 and the other
    ranks find it by the handshake."""
    override = os.environ.get("YMC_STORE_PORT")
    if override:
        return [int(override)]
    out = []
    for i in range(1, _PORT_SPAN + 1):
        p = master_port + i
        if p > 65535:
            p = 1024 + (p - 65536)
        out.append(p)
    return out


def _resolve(addr):
    """IPv4 addresses of a host name ([] when it does not resolve -- the GPU boxes' own hostname does not)."""
    try:
        infos = socket.getaddrinfo(addr, None, socket.AF_INET, socket.SOCK_STREAM)
    except (socket.gaierror, UnicodeError, OSError):
        return []
    out = []
    for info in infos:
        if info[4][0] not in out:
            out.append(info[4][0])
    return out


def _single_node(world, resolved):
    """All ranks of this launch on one machine?  The launcher says so (LOCAL_WORLD_SIZE == WORLD_SIZE);
    without that, a master address that is loopback or does not resolve at all can only be local."""
    local = os.environ.get("LOCAL_WORLD_SIZE")
    if local is not None:
        try:
            return int(local) == int(world)
        except ValueError:
            pass
    return not resolved or all(a.startswith("127.") for a in resolved)


def store_addresses(addr, world):
    """-> (address rank 0 binds, addresses a client tries in turn).  Rank 0 never binds MASTER_ADDR *by name*
    (a name that does not resolve on the box -- `runc` -- used to end the launch there): loopback when every
    rank is on this machine, every interface otherwise.  Clients try MASTER_ADDR as resolved and loopback,
    loopback first on one node; a peer that is not this launch's store fails the handshake and is passed over."""
    override = os.environ.get("YMC_STORE_BIND")
    resolved = _resolve(addr)
    single = _single_node(world, resolved)
    bind = override or (_LOOPBACK if single else "0.0.0.0")
    tries = [_LOOPBACK] + [a for a in resolved if a != _LOOPBACK] if single else resolved + [_LOOPBACK]
    seen = []
    for a in tries:
        if a not in seen:
            seen.append(a)
    return bind, seen


class _Peer:
    __slots__ = ("sock", "rank", "buf", "since")

    def __init__(self, sock):
        self.sock, self.rank, self.buf, self.since = sock, None, bytearray(), time.monotonic()


class _StoreServer(threading.Thread):
    """Rank 0's side: W persistent connections; collective number `seq` completes when every rank has
    sent its part, and every rank then receives all parts in rank order.  One selector thread that never
    blocks on a peer: every readable socket gets ONE recv per wake-up into its own buffer, the handshake and
    the messages are parsed from there; a connection that has not completed its hello within
    _HELLO_PATIENCE_S, a header announcing more than _MAX_PAYLOAD bytes and a message that stays partial for
    longer than `stall_s` end that connection only."""

    def __init__(self, listener, world, key, nonce, stall_s=60.0):
        super().__init__(name="ymc-store", daemon=True)
        self.listener, self.world, self.key, self.nonce, self.stall_s = listener, world, key, nonce, stall_s
        self.sel = selectors.DefaultSelector()
        self.peers = {}          # sock -> _Peer
        self.parts = {}          # seq -> {rank: payload}
        self.stop = threading.Event()
        listener.setblocking(False)
        self.sel.register(listener, selectors.EVENT_READ, None)

    def run(self):
        try:
            while not self.stop.is_set():
                for key, _ in self.sel.select(timeout=0.2):
                    if key.fileobj is self.listener:
                        self._accept()
                    else:
                        self._readable(self.peers.get(key.fileobj))
                self._reap()
        finally:
            for s in list(self.peers) + [self.listener]:
                try:
                    s.close()
                except OSError:
                    pass

    def _accept(self):
        try:
            sock, _ = self.listener.accept()
        except OSError:
            return
        sock.settimeout(10.0)      # bounds the sends to a peer that has stopped reading; recv is only called when readable
        self.peers[sock] = _Peer(sock)
        self.sel.register(sock, selectors.EVENT_READ, None)

    def _drop(self, peer):
        try:
            self.sel.unregister(peer.sock)
        except (KeyError, ValueError):
            pass
        self.peers.pop(peer.sock, None)
        try:
            peer.sock.close()
        except OSError:
            pass

    def _reap(self):
        now = time.monotonic()
        for peer in list(self.peers.values()):
            if peer.rank is None and now - peer.since > _HELLO_PATIENCE_S:
                self._drop(peer)
            elif peer.rank is not None and peer.buf and now - peer.since > self.stall_s:
                self._drop(peer)

    def _readable(self, peer):
        if peer is None:
            return
        try:
            chunk = peer.sock.recv(1 << 20)
        except (BlockingIOError, InterruptedError, socket.timeout):
            return
        except OSError:
            chunk = b""
        if not chunk:
            self._drop(peer)
            return
        if not peer.buf:
            peer.since = time.monotonic()     # a new message starts now
        peer.buf += chunk
        if peer.rank is None and not self._hello(peer):
            return
        while peer.sock in self.peers and len(peer.buf) >= _MSG.size:
            seq, nbytes = _MSG.unpack_from(peer.buf)
            if nbytes > _MAX_PAYLOAD:
                self._drop(peer)
                return
            if len(peer.buf) < _MSG.size + nbytes:
                return
            payload = bytes(peer.buf[_MSG.size:_MSG.size + nbytes])
            del peer.buf[:_MSG.size + nbytes]
            peer.since = time.monotonic()
            self._deliver(peer, seq, payload)

    def _hello(self, peer):
        """-> True once the peer is a greeted rank (its buffer then holds whatever followed the hello)."""
        if len(peer.buf) < _HELLO.size:
            return False
        magic, key, rank, world = _HELLO.unpack_from(peer.buf)
        taken = any(p.rank == rank for p in self.peers.values())
        if magic != _MAGIC or key != self.key or world != self.world or not 0 <= rank < world or taken:
            self._drop(peer)
            return False
        del peer.buf[:_HELLO.size]
        try:
            peer.sock.sendall(_WELCOME.pack(_MAGIC_REPLY, self.nonce))
            peer.sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        except OSError:
            self._drop(peer)
            return False
        peer.rank = rank
        return True

    def _deliver(self, peer, seq, payload):
        got = self.parts.setdefault(seq, {})
        got[peer.rank] = payload
        if len(got) == self.world:
            body = b"".join(struct.pack("<Q", len(got[r])) + got[r] for r in range(self.world))
            del self.parts[seq]
            for p in list(self.peers.values()):
                if p.rank is None:
                    continue
                try:
                    p.sock.sendall(_MSG.pack(seq, len(body)) + body)
                except OSError:
                    self._drop(p)


class TcpStore:
    """All-gather of byte strings among the ranks of one launch over loopback / the node's network.
    Rank 0 serves (a daemon thread in its process); every rank, 0 included, is a client."""

    def __init__(self, rank, world, addr=None, port=None, timeout=None):
        self.rank, self.world = int(rank), int(world)
        self.addr = addr or os.environ.get("MASTER_ADDR", _LOOPBACK)
        self.master_port = int(port if port is not None else os.environ.get("MASTER_PORT", "29500"))
        self.timeout = float(timeout if timeout is not None else os.environ.get("YMC_COMM_TIMEOUT", "300"))
        self.key = _launch_key(self.world, self.master_port)
        self.bind_addr, self.try_addrs = store_addresses(self.addr, self.world)
        self.seq = 0
        self.server = None
        self.nonce = None
        self.port = None
        self.peer_addr = None
        if self.rank == 0:
            self._serve()
        self.sock = self._connect()

    def _serve(self):
        last = None
        for p in candidate_ports(self.master_port):
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                s.bind((self.bind_addr, p))
                s.listen(self.world + 8)
            except OSError as e:
                last = e
                s.close()
                continue
            self.server = _StoreServer(s, self.world, self.key, os.urandom(16), stall_s=max(self.timeout, 5.0))
            self.server.start()
            self.port = p
            return
        raise CommError(f"rank 0 cannot serve the store on any of {candidate_ports(self.master_port)} "
                        f"at {self.bind_addr}: {last}")

    def _connect(self):
        deadline = time.time() + self.timeout
        ports = [self.port] if self.port else candidate_ports(self.master_port)
        if self.rank == 0:       # its own server, in this process
            addrs = [_LOOPBACK if self.bind_addr == "0.0.0.0" else self.bind_addr]
        else:
            addrs = self.try_addrs
        while True:
            for a in addrs:
                for p in ports:
                    try:
                        s = socket.create_connection((a, p), timeout=2.0)
                    except OSError:
                        continue
                    try:
                        if s.getsockname() == s.getpeername():
                            raise ConnectionError("connected to itself")
                        s.settimeout(5.0)
                        s.sendall(_HELLO.pack(_MAGIC, self.key, self.rank, self.world))
                        magic, nonce = _WELCOME.unpack(_recv_exact(s, _WELCOME.size))
                        if magic != _MAGIC_REPLY:
                            raise ConnectionError("not this launch's store")
                    except (OSError, ConnectionError, struct.error):
                        s.close()
                        continue
                    s.settimeout(self.timeout)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    self.nonce, self.port, self.peer_addr = nonce, p, a
                    return s
            if time.time() > deadline:
                raise CommError(f"rank {self.rank}: no store of this launch answered on {addrs} (MASTER_ADDR "
                                f"{self.addr!r}) ports {ports} within {self.timeout:.0f} s (is rank 0 running with "
                                "the same MASTER_ADDR / MASTER_PORT / WORLD_SIZE?)")
            time.sleep(0.05)

    def allgather(self, payload):
        """-> [payload of rank 0, ..., payload of rank W-1]; blocks until every rank has called."""
        seq, self.seq = self.seq, self.seq + 1
        payload = bytes(payload)
        try:
            self.sock.sendall(_MSG.pack(seq, len(payload)) + payload)
            rseq, nbytes = _MSG.unpack(_recv_exact(self.sock, _MSG.size))
            body = _recv_exact(self.sock, nbytes)
        except (OSError, ConnectionError) as e:
            raise CommError(f"rank {self.rank}: store collective {seq} failed ({e}); another rank has "
                            "probably exited") from e
        if rseq != seq:
            raise CommError(f"rank {self.rank}: store answered collective {rseq}, expected {seq}")
        out, off = [], 0
        for _ in range(self.world):
            (n,) = struct.unpack_from("<Q", body, off)
            out.append(body[off + 8:off + 8 + n])
            off += 8 + n
        return out

    def broadcast(self, payload, root=0):
        return self.allgather(payload if self.rank == root else b"")[root]

    def close(self):
        try:
            self.sock.settimeout(5.0)      # a peer that died without saying goodbye is not waited for
            self.allgather(b"bye")         # nobody tears the server down under a peer's last collective
        except CommError:
            pass
        try:
            self.sock.close()
        except OSError:
            pass
        if self.server is not None:
            self.server.stop.set()
            self.server.join(timeout=2.0)
            self.server = None


# ---------------------------------------------------------------------------
# launching the ranks of one node without a launcher
# ---------------------------------------------------------------------------

def free_master_port():
    """A port P that is free now and leaves P+1 .. P+32 (the store's candidates) below 65536."""
    for _ in range(64):
        with socket.socket() as s:
            s.bind((_LOOPBACK, 0))
            p = s.getsockname()[1]
        if p + _PORT_SPAN <= 65535:
            return p
    return 29500


def spawn_local_ranks(n, argv=None, grace_s=10.0):
    """`python prog.py --gpus N` started as ONE plain process: run this command line as N fresh child
    processes, one per GPU of this node, with the environment a launcher would export (RANK, LOCAL_RANK,
    WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR = 127.0.0.1, a free MASTER_PORT, a random YMC_LAUNCH_SECRET).
    Rank 0 inherits this process's stdout (its single result line), the other ranks' stdout goes to stderr.
    -> the exit code: 0 only if every rank returned 0; the first failure ends the others (by their PIDs).

    Must be called BEFORE anything in this process touches the GPU: the children are fresh processes
    (fork + exec from a process without a HIP context), never a re-exec of an initialised one."""
    import secrets
    import subprocess
    argv = list(sys.argv if argv is None else argv)
    n = int(n)
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR=_LOOPBACK,
                MASTER_PORT=str(free_master_port()), YMC_LAUNCH_SECRET=secrets.token_hex(16))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.stdout.flush()
    procs = []
    try:
        for r in range(n):
            env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen([sys.executable] + argv, env=env,
                                          stdout=None if r == 0 else sys.stderr.fileno()))
        codes = [None] * n
        failed_at = None
        while any(c is None for c in codes):
            for r, pr in enumerate(procs):
                if codes[r] is None:
                    codes[r] = pr.poll()
            bad = [c for c in codes if c not in (None, 0)]
            if bad and failed_at is None:
                failed_at = time.monotonic()
            if failed_at is not None and time.monotonic() - failed_at > grace_s:
                break       # the others had their chance to notice (store collectives fail, watchdogs fire)
            time.sleep(0.05)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.terminate()
        for pr in procs:
            try:
                pr.wait(timeout=5.0)
            except subprocess.TimeoutExpired:
                pr.kill()
                pr.wait()
    codes = [pr.returncode for pr in procs]
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print("spawn_local_ranks: " + ", ".join(f"rank {r} exited {c}" for r, c in bad), file=sys.stderr, flush=True)
        first = bad[0][1]
        return first if 0 < first < 256 else 1
    return 0


# ---------------------------------------------------------------------------
# rank -> GPU -> NUMA node -> host cores
# ---------------------------------------------------------------------------
# Eight host processes drive eight GPUs of one node: each is pinned to the cores of ITS GPU's NUMA node (the ctypes
# calls, the page-locked egress buffers and the store's sockets then stay on the socket the GPU hangs off), and the
# launch says once, on stderr, which rank got which device, node, cores and power cap -- so that a clock that drops
# under the node's power envelope, or two ranks on one socket's cores, is visible in the first real run.  Everything
# is read from sysfs (no GPU call: this runs before the rank creates its context); a box without that information
# is left unpinned and says so.

def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def _format_cpulist(cpus):
    cpus = sorted(cpus)
    out, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(out)


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def gpu_topology(sysfs="/sys", visible=None):
    """The GPUs in the order HIP numbers them -- the KFD topology nodes that have SIMDs, by node id, filtered and
    re-ordered by ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES when those are plain index lists -- each with its NUMA
    node, that node's cores and its power cap: [{index, kfd_node, render_minor, numa_node, cpus (set), power_cap_w}].
    -> [] when the box does not expose it."""
    base = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
    try:
        node_ids = sorted(int(n) for n in os.listdir(base) if n.isdigit())
    except OSError:
        return []
    gpus = []
    for nid in node_ids:
        props = {}
        for line in (_read(os.path.join(base, str(nid), "properties")) or "").splitlines():
            k, _, v = line.partition(" ")
            props[k] = v.strip()
        if int(props.get("simd_count", "0") or 0) <= 0:
            continue
        minor = int(props.get("drm_render_minor", "-1") or -1)
        dev = os.path.join(sysfs, "class", "drm", f"renderD{minor}", "device")
        numa = _read(os.path.join(dev, "numa_node"))
        cpulist = _read(os.path.join(dev, "local_cpulist"))
        numa = int(numa) if numa not in (None, "") else -1
        if (not cpulist) and numa >= 0:
            cpulist = _read(os.path.join(sysfs, "devices", "system", "node", f"node{numa}", "cpulist"))
        cap = None
        hw = os.path.join(dev, "hwmon")
        try:
            for h in sorted(os.listdir(hw)):
                v = _read(os.path.join(hw, h, "power1_cap"))
                if v:
                    cap = int(v) / 1e6          # microwatts
                    break
        except OSError:
            pass
        gpus.append(dict(kfd_node=nid, render_minor=minor, numa_node=numa,
                         cpus=_parse_cpulist(cpulist) if cpulist else set(), power_cap_w=cap))
    if visible is None:
        visible = os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("HIP_VISIBLE_DEVICES") or ""
    picks = [p.strip() for p in visible.split(",") if p.strip()]
    if picks and all(p.isdigit() for p in picks):
        gpus = [gpus[int(p)] for p in picks if int(p) < len(gpus)]
    for i, g in enumerate(gpus):
        g["index"] = i
    return gpus


def rank_cpu_sets(devices, topo, allowed):
    """cores for the ranks that drive `devices` (device index per local rank): the rank's GPU's NUMA node's cores that
    this process may use, cut into equal contiguous shares where several ranks' GPUs hang off one node.  A rank
    whose device has no known node, or whose share would be empty, gets None (stays unpinned)."""
    by_node = {}
    for r, dev in enumerate(devices):
        g = topo[dev] if 0 <= dev < len(topo) else None
        if g is not None and g["cpus"]:
            by_node.setdefault((g["numa_node"], frozenset(g["cpus"])), []).append(r)
    out = [None] * len(devices)
    for (_, cpus), ranks in by_node.items():
        usable = sorted(set(cpus) & set(allowed))
        share = len(usable) // len(ranks)
        if share == 0:
            continue
        for i, r in enumerate(ranks):
            out[r] = set(usable[i * share:(i + 1) * share])
    return out


def pin_rank(local_rank, local_world, device=None, sysfs="/sys", apply=True):
    """Pin the calling process (a rank, before it touches the GPU) to its share of its GPU's NUMA node.
    -> dict(rank, device, numa_node, cpus (cpulist string or None), power_cap_w): what the launch reports."""
    topo = gpu_topology(sysfs)
    devices = list(range(local_world))
    if device is not None:
        devices[local_rank] = device
    allowed = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else set(range(os.cpu_count() or 1))
    sets = rank_cpu_sets(devices, topo, allowed) if topo else [None] * local_world
    mine = sets[local_rank]
    dev = devices[local_rank]
    g = topo[dev] if topo and dev < len(topo) else None
    if apply and mine and os.environ.get("YMC_PIN_RANKS", "1") != "0" and hasattr(os, "sched_setaffinity"):
        try:
            os.sched_setaffinity(0, mine)
        except OSError:
            mine = None
    elif os.environ.get("YMC_PIN_RANKS", "1") == "0":
        mine = None
    return dict(rank=local_rank, device=dev, numa_node=g["numa_node"] if g else None,
                cpus=_format_cpulist(mine) if mine else None, power_cap_w=g["power_cap_w"] if g else None)


def describe_rank(info):
    cap = f"{info['power_cap_w']:.0f} W" if info.get("power_cap_w") else "unknown"
    where = f"NUMA node {info['numa_node']}" if info.get("numa_node") is not None and info["numa_node"] >= 0 else "NUMA node unknown"
    cpus = f"cores {info['cpus']}" if info.get("cpus") else "not pinned"
    return f"rank {info['rank']} -> device {info['device']} -> {where} -> {cpus}; power cap {cap}"


# ---------------------------------------------------------------------------
# communicators
# ---------------------------------------------------------------------------

class _StdoutToStderr:
    """RCCL prints a version banner on the C stdout while a communicator is
    initialised; a benchmark that promises ONE line on stdout routes it to
    stderr (fd level, flushing the C buffer before switching back)."""

    def __enter__(self):
        import ctypes
        sys.stdout.flush()
        self._libc = ctypes.CDLL(None)
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        self._libc.fflush(None)
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def pack_summary_host(run, d):
    """The word layout of pack_summary_kernel from a host-side summary fetch: (K, 3d+5) u64."""
    s, (x, lp) = run.summary(), run.state()
    out = np.zeros((run.K, summary_words(d)), dtype=np.uint64)
    out[:, 0:d] = np.ascontiguousarray(x).view(np.uint64)
    out[:, d] = np.ascontiguousarray(lp).view(np.uint64)
    out[:, d + 1] = s["accepted"]
    out[:, d + 2] = s["wf_n"]
    out[:, d + 3:2 * d + 3] = np.ascontiguousarray(s["wf_mean"]).view(np.uint64)
    out[:, 2 * d + 3:3 * d + 3] = np.ascontiguousarray(s["wf_m2"]).view(np.uint64)
    out[:, 3 * d + 3] = s["status"]
    return out


class TcpCommunicator:
    """The communicator's interface over the TCP store.  Selected ONLY by YMC_COMM=tcp: rehearsals of
    the N > 1 flow without N GPUs.  Nothing on the sampling path uses it; it carries the barrier, the
    max-over-ranks timing and the final gather of summaries."""

    backend = "tcp"

    def __init__(self, store):
        self.store, self.rank, self.nranks = store, store.rank, store.world

    def count(self):
        return self.nranks

    def barrier(self):
        self.store.allgather(b"")

    def allreduce_max(self, value):
        return max(struct.unpack("<d", p)[0] for p in self.store.allgather(struct.pack("<d", float(value))))

    def allgather_summary(self, run):
        mine = run.packed_summary() if hasattr(run, "packed_summary") else pack_summary_host(run, run.d)
        parts = self.store.allgather(np.ascontiguousarray(mine, dtype=np.uint64).tobytes())
        return np.stack([np.frombuffer(p, dtype=np.uint64).reshape(mine.shape) for p in parts])

    def close(self):
        self.store.close()


class _Watchdog:
    """ncclCommInitRank blocks inside C for as long as a peer is missing.  If it has not returned after
    `seconds`, say so and end the process: a hung rank must not outlive its launcher's patience."""

    def __init__(self, seconds, what, rank):
        self.done = threading.Event()
        self.t = threading.Thread(target=self._run, args=(seconds, what, rank), daemon=True)

    def _run(self, seconds, what, rank):
        if not self.done.wait(seconds):
            print(f"ymc: rank {rank}: {what} did not finish within {seconds:.0f} s; exiting", file=sys.stderr, flush=True)
            os._exit(3)

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *exc):
        self.done.set()
        return False


def _agree(store, ok, message, what):
    """Every rank reports ok / its error; all of them raise if any failed."""
    answers = store.allgather((b"ok" if ok else b"E:" + message.encode("utf-8", "replace")))
    bad = [(r, a[2:].decode("utf-8", "replace")) for r, a in enumerate(answers) if a != b"ok"]
    if bad:
        detail = "; ".join(f"rank {r}: {m}" for r, m in bad)
        raise CommError(f"{what} failed ({detail}). RCCL is required for more than one rank; "
                        "set YMC_COMM=tcp only to rehearse the flow without it.")


def make_communicator(ctx, rank, world):
    """The communicator of this rank: RCCL (default, `backend == "rccl"`), or the TCP stand-in when
    YMC_COMM=tcp is set explicitly.  RCCL failing to come up raises CommError on EVERY rank -- the
    caller exits non-zero; nothing falls back."""
    from . import _ffi
    mode = os.environ.get("YMC_COMM", "rccl")
    if mode not in ("rccl", "tcp"):
        raise CommError(f"YMC_COMM={mode!r}: the communicators are 'rccl' and 'tcp'")
    store = TcpStore(rank, world)
    if mode == "tcp":
        return TcpCommunicator(store)
    # one node: RCCL's bootstrap sockets go over loopback (the container hostname may not resolve)
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    timeout = store.timeout
    with _StdoutToStderr():
        # 1. what can fail on ONE rank (librccl missing, unique id) is settled before anyone blocks
        uid, err = b"", ""
        try:
            if rank == 0:
                uid = _ffi.Communicator.unique_id()
            else:
                _ffi.Communicator.preload()
        except (OSError, _ffi.YmcError) as e:
            err = str(e)
        _agree(store, not err, err, "loading RCCL")
        uid = store.broadcast(uid)
        # 2. the collective bring-up, under a watchdog
        comm, err = None, ""
        try:
            with _Watchdog(timeout, "ncclCommInitRank", rank):
                comm = _ffi.Communicator(ctx, uid, world, rank)
                comm.barrier()
                seen = comm.count()
            if seen != world:
                err = f"RCCL reports {seen} ranks, the launch has {world}"
        except _ffi.YmcError as e:
            err = str(e)
        _agree(store, not err, err, "ncclCommInitRank")
    comm.store = store
    return comm
