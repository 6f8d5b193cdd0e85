"""Sharding of an ensemble over GPUs (one process per GPU) and the host side of
the single collective: the gather of per-chain summaries.

Nothing is exchanged while sampling: rank r of W owns the contiguous block of
global chain ids [r*K, (r+1)*K) and every chain draws from the stream of its
GLOBAL id, so per-chain results do not depend on W.  After sampling each rank
contributes its packed summaries (3d+5 eight-byte words per chain, written by
pack_summary_kernel) to one RCCL all-gather (`_ffi.Communicator`).
"""
import os
import sys
import time

import numpy as np

from .diagnostics import merge_welford


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


class FileRendezvous:
    """Exchange of the RCCL unique id without PyTorch: rank 0 publishes the id in
    a file keyed by the launcher's pid, MASTER_PORT and run id; the others poll."""

    def __init__(self, directory="/tmp"):
        key = f"{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}"
        self.path = os.path.join(directory, f"ymc_rdzv_{key}")

    def publish(self, payload):
        tmp = f"{self.path}.{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(payload)
        os.replace(tmp, self.path)

    def wait(self, nbytes, timeout=300.0):
        deadline = time.time() + timeout
        while time.time() < deadline:
            try:
                with open(self.path, "rb") as f:
                    payload = f.read()
                if len(payload) == nbytes:
                    return payload
            except FileNotFoundError:
                pass
            time.sleep(0.05)
        raise RuntimeError("rank 0 never published the RCCL unique id")

    def cleanup(self):
        try:
            os.remove(self.path)
        except OSError:
            pass


class _StdoutToStderr:
    """RCCL prints a version banner on the C stdout while a communicator is
    initialised; a benchmark that promises ONE line on stdout routes it to
    stderr (fd level, flushing the C buffer before switching back)."""

    def __enter__(self):
        import ctypes
        import sys
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


class FileCommunicator:
    """The communicator's interface over files of one node's /tmp: the stand-in when RCCL cannot be
    brought up (library missing, bootstrap refused).  Nothing on the sampling path uses it; it only
    keeps the barrier, the max-over-ranks timing and the final gather of summaries working."""

    backend = "file"

    def __init__(self, rank, world, directory="/tmp", timeout=600.0):
        self.rank, self.nranks, self.timeout = rank, world, timeout
        self.dir = FileRendezvous(directory).path + "_comm"
        os.makedirs(self.dir, exist_ok=True)
        self.seq = 0

    def _path(self, seq, rank):
        return os.path.join(self.dir, f"op{seq}_r{rank}.npy")

    def _exchange(self, array):
        seq, self.seq = self.seq, self.seq + 1
        if seq >= 2:
            try:
                os.remove(self._path(seq - 2, self.rank))  # every rank has read it (it wrote op seq-1 since)
            except OSError:
                pass
        tmp = self._path(seq, self.rank) + f".{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            np.save(f, np.ascontiguousarray(array))
        os.replace(tmp, self._path(seq, self.rank))
        out, deadline = [], time.time() + self.timeout
        for r in range(self.nranks):
            while True:
                try:
                    out.append(np.load(self._path(seq, r)))
                    break
                except (FileNotFoundError, ValueError, EOFError):
                    if time.time() > deadline:
                        raise RuntimeError(f"rank {r} did not reach collective {seq}")
                    time.sleep(0.002)
        return out

    def barrier(self):
        """All ranks leave within microseconds of each other: the file exchange alone is only as tight as
        its polling interval, so the ranks agree on a wall-clock instant shortly after the last arrival
        (one node, one clock) and spin until it."""
        stamps = self._exchange(np.array([time.time()]))
        release = max(float(v[0]) for v in stamps) + 0.05
        while time.time() < release:
            pass

    def allreduce_max(self, value):
        return float(max(v[0] for v in self._exchange(np.array([float(value)]))))

    def allgather_summary(self, run):
        return np.stack(self._exchange(pack_summary_host(run, run.d)))


_FALLBACK_MARK = b"YMC-NO-RCCL"


def make_communicator(ctx, rank, world):
    """The RCCL communicator of this rank; a FileCommunicator if YMC_COMM=file is set or RCCL fails to
    come up (rank 0 then tells the others through the rendezvous file, so all ranks agree)."""
    from . import _ffi
    if os.environ.get("YMC_COMM", "rccl") == "file":
        return FileCommunicator(rank, world)
    # one node: RCCL's bootstrap sockets go over loopback (the container hostname may not resolve)
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    rdzv = FileRendezvous()
    mark = _FALLBACK_MARK.ljust(_ffi.UNIQUE_ID_BYTES, b"\0")
    with _StdoutToStderr():
        if rank == 0:
            try:
                uid = _ffi.Communicator.unique_id()
            except (OSError, _ffi.YmcError) as e:
                print(f"ymc: RCCL unavailable ({e}); using the file communicator", file=sys.stderr)
                uid = mark
            rdzv.publish(uid)
        else:
            uid = rdzv.wait(_ffi.UNIQUE_ID_BYTES)
        if uid == mark:
            return FileCommunicator(rank, world)
        try:
            comm = _ffi.Communicator(ctx, uid, world, rank)
            comm.barrier()
        except _ffi.YmcError as e:
            # ncclCommInitRank fails on every rank or on none (it is itself a collective)
            print(f"ymc: RCCL communicator failed on rank {rank} ({e}); using the file communicator", file=sys.stderr)
            return FileCommunicator(rank, world)
    if rank == 0:
        rdzv.cleanup()
    return comm
