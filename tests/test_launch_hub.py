"""The framework-free launcher bench.py uses for `--gpus N`: rank spawning, the rendezvous
file and the TCP hub (mppi_numba_amd/launch.py), with real processes on CPU; and the sharded
update arithmetic carried over it (the exchange RCCL's all-gather does on the GPUs)."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %r)
    import numpy as np
    from mppi_numba_amd import launch
    rank, local_rank, world = launch.rank_from_env()
    hub = launch.Hub(rank, world)
    uid = hub.broadcast(b"id-from-rank-0" if rank == 0 else None)
    got = hub.all_gather({"rank": rank, "pid": os.getpid()})
    hub.barrier()
    slowest = hub.all_max(0.25 * (rank + 1))
    # the sharded update (update_kernels.h): packets {beta_g, den_g, num_g[T][2]} combined in rank order
    rng = np.random.default_rng(7)
    n, t, lam = 96 * world, 5, 0.7
    costs, eps = rng.uniform(3, 40, n), rng.normal(size=(n, t, 2))
    mine = slice(rank * n // world, (rank + 1) * n // world)
    beta_g = costs[mine].min()
    w_g = np.exp(-(costs[mine] - beta_g) / lam)
    packet = np.concatenate(([beta_g, w_g.sum()], np.einsum("n,ntc->tc", w_g, eps[mine]).ravel()))
    packets = np.stack(hub.all_gather(packet))
    beta = packets[:, 0].min()
    scale = np.exp(-(packets[:, 0] - beta) / lam)
    du = (scale[:, None] * packets[:, 2:]).sum(0) / (scale * packets[:, 1]).sum()
    w = np.exp(-(costs - costs.min()) / lam)
    want = np.einsum("n,ntc->tc", w / w.sum(), eps).ravel()
    if rank == 0:
        print(json.dumps({"uid": uid.decode(), "ranks": [g["rank"] for g in got], "pids": len({g["pid"] for g in got}),
                          "slowest": slowest, "err": float(np.abs(du - want).max())}))
    hub.close()
    """) % ROOT


@pytest.mark.parametrize("world", [2, 4])
def test_spawned_ranks_meet_at_the_hub(world, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    driver = ("import sys; sys.path.insert(0, %r)\n"
              "from mppi_numba_amd import launch\n"
              "sys.exit(launch.spawn_ranks(%d, [sys.executable, %r], timeout=120))\n" % (ROOT, world, str(script)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MPPI_RDZV_FILE")}
    out = subprocess.run([sys.executable, "-c", driver], capture_output=True, text=True, timeout=180, env=env)
    assert out.returncode == 0, out.stderr
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["uid"] == "id-from-rank-0" and res["ranks"] == list(range(world)) and res["pids"] == world
    assert res["slowest"] == 0.25 * world
    assert res["err"] < 1e-12


def test_launcher_environment_of_torch_distributed_run(tmp_path):
    """The contract's launch line (`python -m torch.distributed.run --nproc-per-node N ...`) sets
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT and makes all workers children of one agent
    process: the rendezvous file name derived from (parent pid, port) is shared and the hub
    forms without MPPI_RDZV_FILE."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MPPI_RDZV_FILE")}
    agent = textwrap.dedent("""
        import os, subprocess, sys
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29431")
            procs.append(subprocess.Popen([sys.executable, %r], env=env))
        sys.exit(max(p.wait() for p in procs))
        """) % str(script)
    out = subprocess.run([sys.executable, "-c", agent], capture_output=True, text=True, timeout=180, env=env)
    assert out.returncode == 0, out.stderr
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["ranks"] == [0, 1] and res["err"] < 1e-12


def test_a_dying_rank_takes_the_job_down(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(3)\ntime.sleep(60)\n")
    driver = ("import sys; sys.path.insert(0, %r)\n"
              "from mppi_numba_amd import launch\n"
              "sys.exit(launch.spawn_ranks(2, [sys.executable, %r], timeout=120))\n" % (ROOT, str(script)))
    out = subprocess.run([sys.executable, "-c", driver], capture_output=True, text=True, timeout=60)
    assert out.returncode == 3
    assert "rank(s) [1] failed" in out.stderr


def test_wire_format_round_trip_and_refusals():
    """The hub's messages are a tagged encoding of bytes / numbers / float arrays: nothing that runs
    code on decoding (the first version used pickle)."""
    import socket
    from mppi_numba_amd import launch
    a, b = socket.socketpair()
    try:
        msg = [b"\x00" * 128, None, 3, -2.5, True, "text", {"rank": 1, "packet": np.arange(6, dtype=np.float64).reshape(3, 2)},
               [np.float32(1.5), np.zeros((2, 0, 3), dtype=np.float32)]]
        launch._send(a, msg)
        got = launch._recv(b)
        assert got[0] == msg[0] and got[1] is None and got[2] == 3 and got[3] == -2.5 and got[4] is True and got[5] == "text"
        assert got[6]["rank"] == 1 and np.array_equal(got[6]["packet"], msg[6]["packet"]) and got[6]["packet"].dtype == np.float64
        assert got[7][0] == 1.5 and got[7][1].shape == (2, 0, 3)
        with pytest.raises(TypeError):
            launch._send(a, object())
        with pytest.raises(TypeError):
            launch._send(a, np.array(["x"]))
        import pickle, struct
        blob = pickle.dumps({"x": 1})
        a.sendall(struct.pack("<Q", len(blob)) + blob)  # what the old protocol would have sent
        with pytest.raises(ValueError):
            launch._recv(b)
        # a first message is read with a cap of a few hundred bytes, and nothing nests deeper than the hub's own messages
        launch._send(a, ["x" * 64, 1])
        assert launch._recv(b, max_bytes=256) == ["x" * 64, 1]
        launch._send(a, ["x" * 4096, 1])
        with pytest.raises(ValueError):
            launch._recv(b, max_bytes=256)
    finally:
        a.close()
        b.close()
    a, b = socket.socketpair()
    try:
        import struct
        deep = b"L" + struct.pack("<I", 1)
        blob = deep * 50 + b"N"
        a.sendall(struct.pack("<Q", len(blob)) + blob)
        with pytest.raises(ValueError, match="nested"):
            launch._recv(b)
        blob = b"L" + struct.pack("<I", 0xffffffff)  # a container that claims four billion items
        a.sendall(struct.pack("<Q", len(blob)) + blob)
        with pytest.raises(ValueError):
            launch._recv(b)
    finally:
        a.close()
        b.close()


def test_rendezvous_file_is_exclusive_and_owner_only(tmp_path):
    from mppi_numba_amd import launch
    path = str(tmp_path / "r.json")
    launch._write_rendezvous(path, 1234, "tok")
    assert (os.stat(path).st_mode & 0o777) == 0o600
    assert launch._read_rendezvous(path)["port"] == 1234
    with pytest.raises(FileExistsError):
        launch._write_rendezvous(path, 1, "other")  # (a fresh file of somebody else's is never overwritten)
    os.chmod(path, 0o666)
    assert launch._read_rendezvous(path) is None  # writable by others: not trusted
    os.unlink(path)
    os.symlink(str(tmp_path / "elsewhere"), path)
    with pytest.raises(OSError):
        launch._write_rendezvous(path, 1, "t")  # never through a symlink


def test_hub_turns_away_a_peer_without_the_token(tmp_path):
    import socket, threading, time as _time
    from mppi_numba_amd import launch
    path = str(tmp_path / "r.json")
    result = {}

    def rank0():
        hub = launch.Hub(0, 2, path=path, timeout=30)
        result["gathered"] = hub.all_gather("zero")
        hub.close()

    th = threading.Thread(target=rank0)
    th.start()
    info = None
    for _ in range(500):
        try:
            info = launch._read_rendezvous(path)
            if info:
                break
        except OSError:
            pass
        _time.sleep(0.01)
    assert info
    intruder = socket.create_connection(("127.0.0.1", info["port"]))
    launch._send(intruder, ["not-the-token", 1])
    _time.sleep(0.1)
    big = socket.create_connection(("127.0.0.1", info["port"]))
    launch._send(big, [[["x" * 100000]], True])  # oversized, nested, a bool for the rank: dropped before any of it is decoded
    _time.sleep(0.1)
    hub1 = launch.Hub(1, 2, path=path, timeout=30)
    got = hub1.all_gather("one")
    hub1.close()
    th.join(30)
    intruder.close()
    big.close()
    assert got == ["zero", "one"] and result["gathered"] == ["zero", "one"]


# ---- bench.py: a run that cannot start is ONE parsed JSON line with an `error`, never a hang ----
def _bench(args, env=None, timeout=120):
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                         env=dict(os.environ, MPPI_HUB_TIMEOUT="20", **(env or {})), cwd=root)
    lines = [ln for ln in run.stdout.splitlines() if ln.strip()]
    return run.returncode, [json.loads(ln) for ln in lines], run.stderr


def _no_gpu():
    from mppi_numba_amd import _lib
    try:
        return _lib.device_count() == 0
    except Exception:
        return True


@pytest.mark.skipif(not _no_gpu(), reason="start-up failure of every rank: a box without a GPU")
@pytest.mark.parametrize("gpus", [1, 2])
def test_ranks_that_cannot_open_a_device_report_one_json_error(gpus):
    code, lines, err = _bench(["--gpus", str(gpus), "--steps", "2", "--warmup", "1"])
    assert code == 2, (code, err[-500:])
    assert len(lines) == 1 and lines[0]["value"] is None and lines[0]["n_gpus"] == gpus
    assert "failed to start" in lines[0]["error"] and set(lines[0]["errors_per_rank"]) == {str(r) for r in range(gpus)}


def test_a_rank_that_dies_before_the_rendezvous_is_reported_by_the_launcher():
    code, lines, err = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1"], env={"MPPI_BENCH_DIE_RANK": "1"})
    assert code != 0
    assert len(lines) == 1 and lines[0]["value"] is None and "ended abnormally" in lines[0]["error"], (lines, err[-500:])
    assert lines[0]["errors_per_rank"] == {"1": "exit code 3"}


def test_a_rank_missing_under_an_external_launcher_times_out_into_a_json_error():
    """Launched like the driver does (RANK / WORLD_SIZE from the environment), one rank never shows up: rank 0 gives the
    rendezvous up after MPPI_HUB_TIMEOUT and says so in the line."""
    import json
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MPPI_RDZV_FILE=os.path.join(d, "r.json"), MPPI_HUB_TIMEOUT="3")
        run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                             capture_output=True, text=True, timeout=60, env=env, cwd=root)
    assert run.returncode == 2
    line = json.loads(run.stdout.strip().splitlines()[-1])
    assert line["value"] is None and "rendezvous of 2 ranks failed" in line["error"]
