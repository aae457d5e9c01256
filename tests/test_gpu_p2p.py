"""The peer exchange (include/mppi_hip.h: mppi_planner_p2p_*; update_kernels.h: PeerExchange): control samples
sharded over ranks WITHOUT a collective -- every rank writes its numbers for step t of an update straight into its
peers' inboxes (IPC-mapped fine-grained device memory) from inside the rollout launch that applies the update, so a
sharded iteration is one launch.  On this box the ranks are processes that share the one GPU and reach each other
through IPC handles: the one-process-per-GPU set-up minus xGMI.  The control sequence must have the BITS of the
all-gather + k_apply path (here: packets staged through the host hub), on every rank and between the ranks
(update_useq_numba mppi.py:1113-1191)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_ranks(*args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MPPI_RDZV_FILE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "p2p_ranks.py")] + list(args), capture_output=True,
                         text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("P2P_")]
    assert out.returncode == 0 and len(lines) == 1, (out.stdout[-1500:], out.stderr[-1500:])
    return lines[0]


@pytest.mark.parametrize("ranks,n,t,iterations", [(2, 1024, 100, 6), (2, 4096, 100, 5), (3, 2048, 64, 4), (8, 512, 60, 7)])
def test_ranks_sharing_the_device_through_ipc_have_the_bits_of_the_k_apply_loop(ranks, n, t, iterations):
    line = run_ranks("--ranks", str(ranks), "--n", str(n), "--t", str(t), "--iterations", str(iterations))
    print("\n" + line)
    assert line.startswith("P2P_OK") and "world=%d" % ranks in line, line
    assert "k_rollout_scan_exact+reduces_tiles" in line, line   # (the exchange ran inside the rollout launches)
    assert "max|du|=0.000e+00" in line, line


@pytest.mark.parametrize("workload", ["c2s", "c2c"])
def test_ranks_on_a_map_they_stop_speculating_on_keep_the_exchange(workload):
    """which tiles fail their traction vote differs from rank to rank (own noise), so the ranks
    stop speculating at their own synchronisations; round 4 then left k_rollout_scan_exact for k_rollout_pipe -- and
    with it the peer exchange, which its peers kept waiting in.  Round 5: the kernel stays (direct: its exact
    three-wave schedule), the packets and the exchange with it; the bits are those of the host-staged k_apply loop."""
    line = run_ranks("--ranks", "2", "--n", "2048", "--t", "100", "--iterations", "6", "--calls", "4", "--workload", workload)
    print("\n" + line)
    assert line.startswith("P2P_OK") and "world=2" in line, line
    assert "k_rollout_scan_exact+direct+reduces_tiles" in line, line
    assert "max|du|=0.000e+00" in line, line


@pytest.mark.parametrize("ranks,workload", [(2, "c2"), (2, "c2s")])
def test_device_group_in_one_process_uses_the_peer_exchange(ranks, workload):
    """mppi_group_p2p_connect (peer access before the inboxes are allocated, a ping of all ranks at once) and
    mppi_group_iterate_async's enqueue -- a few iterations per device in turn: every launch of one device waits inside
    the kernel for the other devices' numbers -- with the shard planners of one process.
    (Two planners: on ONE device the runtime maps a process's streams onto four hardware queues, and launches that
    share a queue run one after the other -- with three planners the connect's own ping reports "heard 2 of 3" and
    refuses, as it should; devices of their own have queues of their own.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MPPI_RDZV_FILE")}
    env["MPPI_P2P_MAX_POLLS"] = str(1 << 21)  # (a second, not five, should the launches ever fail to run side by side)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "p2p_group.py"), "--ranks", str(ranks), "--workload", workload],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("GROUP_P2P_")]
    assert out.returncode == 0 and len(lines) == 1, (out.stdout[-1500:], out.stderr[-1500:])
    print("\n" + lines[0])
    assert lines[0].startswith("GROUP_P2P_OK") and "max|du|=0.000e+00" in lines[0], lines[0]
    assert "kernel=k_rollout_scan_exact" in lines[0] and "exchanges=0" not in lines[0], lines[0]


@pytest.mark.parametrize("ranks,n,t,iterations", [(2, 2048, 100, 6), (3, 1024, 64, 5)])
def test_ranks_in_the_tolerance_mode_have_the_bits_of_the_k_apply_loop(ranks, n, t, iterations):
    """math="fast": k_rollout_scan carries the same exchange (publish_step is shared)."""
    line = run_ranks("--ranks", str(ranks), "--n", str(n), "--t", str(t), "--iterations", str(iterations), "--math", "fast")
    print("\n" + line)
    assert line.startswith("P2P_OK") and "world=%d" % ranks in line, line
    assert "k_rollout_scan+reduces_tiles" in line, line
    assert "max|du|=0.000e+00" in line, line


def test_bench_line_with_the_peer_exchange():
    """bench.py --gpus 2: both exchanges are tried; two ranks on one device cannot have an RCCL communicator, so the
    line is the peer exchange's and says so."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MPPI_RDZV_FILE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--n", "2048", "--steps", "20",
                          "--warmup", "5", "--regions", "2", "--no-cpu-baseline"], capture_output=True, text=True,
                         timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    cfg = res["config"]
    assert res["n_gpus"] == 2 and cfg["global_rollouts"] == 4096
    assert cfg["exchange"].startswith("peer exchange"), cfg["exchange"]
    assert set(cfg["exchange_us_per_step"]) == {"p2p"} and cfg["exchange_us_per_step"]["p2p"] > 0
    assert "reduces_tiles=1" in cfg["rollout_kernel"] and res["roofline"]["frac"] > 0
    assert res["ms_per_step"] < 0.2   # (the host-staged exchange is ~0.2 ms per step, the peer exchange ~0.02)


@pytest.mark.parametrize("max_polls", [1 << 20, None])
def test_a_dead_peer_is_an_error_code_not_a_hang(max_polls):
    """Rank 1 leaves before the loop: rank 0's launches wait for its numbers for a bounded time, raise the fault word
    and run on; the call that synchronises returns MPPI_ERR_COMM -- no hung device, no aborted process (the workgroups
    that wait for the published controls outlast the exchange's own limit) -- and with the exchange switched off the
    handle serves stage-level calls again."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MPPI_RDZV_FILE", "MPPI_P2P_MAX_POLLS")}
    if max_polls:
        env["MPPI_P2P_MAX_POLLS"] = str(max_polls)  # (a fraction of a second instead of the default: a few seconds)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "p2p_dead_peer.py")], capture_output=True, text=True,
                         timeout=300, env=env, cwd=ROOT)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("DEAD_PEER_")]
    assert out.returncode == 0 and len(lines) == 1 and lines[0].startswith("DEAD_PEER_OK"), (out.stdout[-800:], out.stderr[-800:])
