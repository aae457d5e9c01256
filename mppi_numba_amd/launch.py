"""One process per GPU without any framework: rank discovery, a rendezvous file and a tiny
TCP hub for the handful of host-side exchanges a multi-GPU run needs (the RCCL unique id, a
barrier, a max over ranks for the clock, and -- only as a debugging fallback -- the 2T+2
doubles of each rank's update packet).  The data path proper is RCCL on the GPUs
(mppi_planner_comm_init); nothing here touches device memory.

Two ways in, same protocol:
  * `python bench.py --gpus N` launched plainly: `spawn_ranks()` starts the N ranks itself
    (RANK / LOCAL_RANK / WORLD_SIZE / MPPI_RDZV_FILE in their environment);
  * launched by `python -m torch.distributed.run --nproc-per-node N ...`: the launcher has set
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT; the rendezvous file name is derived from the
    launcher's pid and port, which all ranks share.
Rank 0 listens on an ephemeral port of 127.0.0.1 and publishes it through the rendezvous file
(written atomically); the other ranks connect.  Single node only, as is the sharding itself.
"""
import json
import os
import pickle
import socket
import struct
import subprocess
import sys
import tempfile
import time


def rank_from_env():
    """(rank, local_rank, world) as the launcher exported them, (0, 0, 1) when run plainly."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def launched_by_a_launcher():
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def rendezvous_path():
    path = os.environ.get("MPPI_RDZV_FILE")
    if path:
        return path
    # torch.distributed.run: every worker has the same parent (the elastic agent) and port
    return os.path.join(tempfile.gettempdir(), "mppi_rdzv_%d_%s.json" % (os.getppid(), os.environ.get("MASTER_PORT", "0")))


def spawn_ranks(world, argv, extra_env=None, timeout=None):
    """Start `world` copies of `argv` (a full command line), one per rank; returns rank 0's
    exit code after all have ended.  Rank 0 inherits stdout; every rank inherits stderr."""
    fd, path = tempfile.mkstemp(prefix="mppi_rdzv_", suffix=".json")
    os.close(fd)
    os.unlink(path)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MPPI_RDZV_FILE=path,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.update(extra_env or {})
        procs.append(subprocess.Popen(argv, env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    deadline = None if timeout is None else time.time() + timeout
    codes = [None] * world
    try:
        while any(c is None for c in codes):
            for r, p in enumerate(procs):
                if codes[r] is None:
                    codes[r] = p.poll()
            failed = [r for r, c in enumerate(codes) if c not in (None, 0)]
            if failed or (deadline is not None and time.time() > deadline):
                # one rank died (or the run overran): the others would wait for it forever
                time.sleep(1.0)
                for r, p in enumerate(procs):
                    if p.poll() is None:
                        p.kill()
                for r, p in enumerate(procs):
                    codes[r] = p.wait()
                if failed:
                    print("rank(s) %s failed with exit code(s) %s" % (failed, [codes[r] for r in failed]), file=sys.stderr)
                return codes[failed[0]] if failed else 124
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        if os.path.exists(path):
            os.unlink(path)
    return codes[0]


def _send(sock, obj):
    blob = pickle.dumps(obj)
    sock.sendall(struct.pack("<Q", len(blob)) + blob)


def _recv(sock):
    def exactly(n):
        buf = b""
        while len(buf) < n:
            chunk = sock.recv(n - len(buf))
            if not chunk:
                raise ConnectionError("peer closed the rendezvous socket")
            buf += chunk
        return buf
    (n,) = struct.unpack("<Q", exactly(8))
    return pickle.loads(exactly(n))


class Hub:
    """Star-shaped exchanges through rank 0 (a few hundred bytes each; used a handful of times per
    run, never inside the timed region's data path)."""

    def __init__(self, rank, world, path=None, timeout=300.0):
        self.rank, self.world = rank, world
        self.peers = []   # rank 0: sockets of ranks 1..world-1, in rank order
        self.sock = None  # other ranks: socket to rank 0
        if world == 1:
            return
        path = path or rendezvous_path()
        if rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(("127.0.0.1", 0))
            srv.listen(world)
            tmp = path + ".tmp%d" % os.getpid()
            with open(tmp, "w") as fh:
                json.dump({"port": srv.getsockname()[1], "pid": os.getpid(), "time": time.time()}, fh)
            os.replace(tmp, path)
            srv.settimeout(timeout)
            by_rank = {}
            while len(by_rank) < world - 1:
                conn, _ = srv.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                by_rank[_recv(conn)] = conn
            self.peers = [by_rank[r] for r in range(1, world)]
            srv.close()
            os.unlink(path)
        else:
            started = time.time()
            info = None
            while info is None:
                try:
                    with open(path) as fh:
                        cand = json.load(fh)
                    if cand["time"] > started - 600.0:  # not a leftover of an earlier run
                        info = cand
                except (OSError, ValueError, KeyError):
                    pass
                if info is None:
                    if time.time() - started > timeout:
                        raise TimeoutError("no rendezvous file %s from rank 0" % path)
                    time.sleep(0.02)
            while True:
                try:
                    self.sock = socket.create_connection(("127.0.0.1", info["port"]), timeout=timeout)
                    break
                except OSError:
                    if time.time() - started > timeout:
                        raise
                    time.sleep(0.05)
            self.sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            _send(self.sock, rank)

    def gather(self, value):
        """Rank 0 gets [value of rank 0, ..., value of rank world-1]; the others get None."""
        if self.world == 1:
            return [value]
        if self.rank == 0:
            return [value] + [_recv(s) for s in self.peers]
        _send(self.sock, value)
        return None

    def broadcast(self, value):
        """Everyone gets rank 0's value."""
        if self.world == 1:
            return value
        if self.rank == 0:
            for s in self.peers:
                _send(s, value)
            return value
        return _recv(self.sock)

    def all_gather(self, value):
        return self.broadcast(self.gather(value))

    def barrier(self):
        self.all_gather(None)

    def all_max(self, value):
        return max(self.all_gather(value))

    def close(self):
        for s in self.peers + ([self.sock] if self.sock else []):
            try:
                s.close()
            except OSError:
                pass
        self.peers, self.sock = [], None
