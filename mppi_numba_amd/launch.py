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
Rank 0 listens on an ephemeral port of 127.0.0.1 and publishes it, with a random token, through the
rendezvous file (created exclusively, mode 0600, in a private directory when bench.py starts the
ranks itself); the other ranks check that the file is their own user's, connect and present the
token.  Messages are a tagged encoding of bytes / ints / floats / float arrays -- nothing that
executes on decoding.  Single node only, as is the sharding itself.
"""
import json
import os
import hmac
import secrets
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


def spawn_ranks(world, argv, extra_env=None, timeout=None, on_failure=None):
    """Start `world` copies of `argv` (a full command line), one per rank; returns rank 0's
    exit code after all have ended.  Rank 0 inherits stdout; every rank inherits stderr."""
    # a directory of our own (mode 0700): the file rank 0 creates in it cannot be anticipated by anyone
    rdzv_dir = tempfile.mkdtemp(prefix="mppi_rdzv_")
    path = os.path.join(rdzv_dir, "rendezvous.json")
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
                    # ranks that give up in an orderly way (exit code 2) have said why on rank 0's stdout already; a rank
                    # that crashed has not, and rank 0 was killed waiting for it: the launcher says it, in the same
                    # one-line form, so that whoever parses the run's stdout finds an error instead of nothing
                    if on_failure is not None and any(codes[r] != 2 for r in failed):
                        on_failure({str(r): "exit code %s" % codes[r] for r in failed})
                return codes[failed[0]] if failed else 124
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        if os.path.exists(path):
            os.unlink(path)
        try:
            os.rmdir(rdzv_dir)
        except OSError:
            pass
    return codes[0]


# ---- wire format -------------------------------------------------------------------------------
# What crosses the hub is a 128-byte communicator id, a few ints / floats and float arrays: a small
# tagged encoding of exactly those (nothing that executes on decoding; the previous version used
# pickle, which does).
_DTYPES = ["float32", "float64", "int32", "int64", "int8", "uint8"]


def _encode(obj, out):
    import numpy as np
    if obj is None:
        out.append(b"N")
    elif isinstance(obj, bool):
        out.append(b"T" if obj else b"f")
    elif isinstance(obj, (int, np.integer)):
        out.append(b"I" + struct.pack("<q", int(obj)))
    elif isinstance(obj, (float, np.floating)):
        out.append(b"F" + struct.pack("<d", float(obj)))
    elif isinstance(obj, (bytes, bytearray)):
        out.append(b"B" + struct.pack("<I", len(obj)) + bytes(obj))
    elif isinstance(obj, str):
        raw = obj.encode("utf-8")
        out.append(b"S" + struct.pack("<I", len(raw)) + raw)
    elif isinstance(obj, (list, tuple)):
        out.append(b"L" + struct.pack("<I", len(obj)))
        for item in obj:
            _encode(item, out)
    elif isinstance(obj, dict):
        out.append(b"D" + struct.pack("<I", len(obj)))
        for key, item in obj.items():
            _encode(str(key), out)
            _encode(item, out)
    elif isinstance(obj, np.ndarray):
        arr = np.ascontiguousarray(obj)
        if arr.dtype.name not in _DTYPES:
            raise TypeError("hub: arrays of dtype %s are not exchanged" % arr.dtype)
        out.append(b"A" + struct.pack("<BB", _DTYPES.index(arr.dtype.name), arr.ndim) +
                   struct.pack("<%dI" % arr.ndim, *arr.shape) + arr.tobytes())
    else:
        raise TypeError("hub: values of type %s are not exchanged" % type(obj).__name__)


_MAX_DEPTH = 8  # what crosses the hub is at most a list of lists of arrays


def _decode(buf, at=0, depth=0):
    import numpy as np
    tag = buf[at:at + 1]
    at += 1
    if tag == b"N":
        return None, at
    if tag in (b"T", b"f"):
        return tag == b"T", at
    if tag == b"I":
        return struct.unpack_from("<q", buf, at)[0], at + 8
    if tag == b"F":
        return struct.unpack_from("<d", buf, at)[0], at + 8
    if tag in (b"B", b"S"):
        (n,) = struct.unpack_from("<I", buf, at)
        raw = bytes(buf[at + 4:at + 4 + n])
        if len(raw) != n:
            raise ValueError("hub: truncated message")
        return (raw if tag == b"B" else raw.decode("utf-8")), at + 4 + n
    if tag in (b"L", b"D"):
        if depth >= _MAX_DEPTH:
            raise ValueError("hub: message nested too deeply")
        (n,) = struct.unpack_from("<I", buf, at)
        at += 4
        if n > len(buf) - at:  # (every item takes at least one byte)
            raise ValueError("hub: malformed container")
    if tag == b"L":
        items = []
        for _ in range(n):
            item, at = _decode(buf, at, depth + 1)
            items.append(item)
        return items, at
    if tag == b"D":
        items = {}
        for _ in range(n):
            key, at = _decode(buf, at, depth + 1)
            items[key], at = _decode(buf, at, depth + 1)
        return items, at
    if tag == b"A":
        code, ndim = struct.unpack_from("<BB", buf, at)
        at += 2
        if code >= len(_DTYPES) or ndim > 8:
            raise ValueError("hub: malformed array header")
        shape = struct.unpack_from("<%dI" % ndim, buf, at)
        at += 4 * ndim
        dtype = np.dtype(_DTYPES[code])
        count = 1
        for d in shape:
            count *= d
        nbytes = count * dtype.itemsize
        if at + nbytes > len(buf):
            raise ValueError("hub: truncated array")
        arr = np.frombuffer(buf, dtype=dtype, count=count, offset=at).reshape(shape).copy()
        return arr, at + nbytes
    raise ValueError("hub: unknown tag %r" % tag)


_MAX_MESSAGE = 1 << 28


def _send(sock, obj):
    parts = []
    _encode(obj, parts)
    blob = b"".join(parts)
    sock.sendall(struct.pack("<Q", len(blob)) + blob)


def _recv(sock, max_bytes=_MAX_MESSAGE):
    def exactly(n):
        buf = bytearray()
        while len(buf) < n:
            chunk = sock.recv(n - len(buf))
            if not chunk:
                raise ConnectionError("peer closed the rendezvous socket")
            buf += chunk
        return bytes(buf)
    (n,) = struct.unpack("<Q", exactly(8))
    if n > max_bytes:
        raise ValueError("hub: message of %d bytes refused" % n)
    value, end = _decode(exactly(n))
    if end != n:
        raise ValueError("hub: trailing bytes in message")
    return value


def _write_rendezvous(path, port, token):
    """Created exclusively and owner-only: nobody else can pre-create it, point it elsewhere through a
    symlink, or read the token the ranks have to present."""
    try:
        st = os.lstat(path)
        if st.st_uid == os.getuid() and time.time() - st.st_mtime > 600.0:
            os.unlink(path)  # a leftover of an earlier run of ours
    except OSError:
        pass
    fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_WRONLY | getattr(os, "O_NOFOLLOW", 0), 0o600)
    with os.fdopen(fd, "w") as fh:
        json.dump({"port": port, "pid": os.getpid(), "time": time.time(), "token": token}, fh)


def _read_rendezvous(path):
    """The file as rank 0 wrote it, or None: it must be a regular file of OUR user that nobody else can write."""
    fd = os.open(path, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
    try:
        st = os.fstat(fd)
        import stat as _stat
        if not _stat.S_ISREG(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o022):
            return None
        with os.fdopen(os.dup(fd)) as fh:
            return json.load(fh)
    finally:
        os.close(fd)


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
            token = secrets.token_hex(32)
            _write_rendezvous(path, srv.getsockname()[1], token)
            srv.settimeout(timeout)
            by_rank = {}
            while len(by_rank) < world - 1:
                conn, _ = srv.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(timeout)
                try:
                    # the first message must be [token, rank] -- a couple of hundred bytes; whoever connects without the
                    # token gets nothing decoded beyond that (no large allocation, no deep nesting) and is dropped
                    hello = _recv(conn, max_bytes=256)
                    ok = (isinstance(hello, list) and len(hello) == 2 and isinstance(hello[0], str) and
                          hmac.compare_digest(hello[0], token) and type(hello[1]) is int and
                          1 <= hello[1] < world and hello[1] not in by_rank)
                except (ValueError, ConnectionError, OSError, struct.error, RecursionError, MemoryError):
                    ok = False
                if not ok:
                    conn.close()
                    continue
                conn.settimeout(None)
                by_rank[hello[1]] = conn
            self.peers = [by_rank[r] for r in range(1, world)]
            srv.close()
            os.unlink(path)
        else:
            started = time.time()
            info = None
            while info is None:
                try:
                    cand = _read_rendezvous(path)
                    if cand is not None and cand["time"] > started - 600.0:  # not a leftover of an earlier run
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
            _send(self.sock, [info["token"], rank])

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
