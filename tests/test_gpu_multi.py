"""Multi-GPU plumbing on a ONE-GPU box: everything except more than one physical device.
  * bench.py --gpus 2 launched plainly starts its two ranks itself, they rendezvous over the hub
    and (sharing the only device, where RCCL refuses two ranks) exchange their packets through
    the host: the launcher, the sharding and the JSON contract run end to end;
  * the one-process group API with a group of one device (RCCL groups, deferred exchange);
  * hipGraph replay of a handle that HAS a communicator: the all-gather is captured."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from test_gpu_scale import build

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MPPI_RDZV_FILE")}
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True,
                         text=True, timeout=timeout, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0]), out.stderr


def test_bench_self_launches_two_ranks_on_one_device():
    res, err = run_bench("--gpus", "2", "--steps", "10", "--warmup", "2", "--n", "2048", "--no-cpu-baseline",
                         "--exchange", "host")
    assert res["n_gpus"] == 2 and res["config"]["global_rollouts"] == 4096 and res["config"]["rollouts_per_gpu"] == 2048
    assert "started by bench.py itself" in res["config"]["launcher"]
    assert "host" in res["config"]["exchange"] and res["config"]["n_ranks_seen_by_rccl"] == 0
    assert res["value"] > 0 and res["scaling"] == "weak"


def test_bench_under_an_external_launcher_environment():
    """The contract's launch line exports RANK / LOCAL_RANK / WORLD_SIZE: with world 1 that must
    simply be the single-GPU run, with the JSON's rccl rank count reported."""
    res, _ = run_bench("--gpus", "1", "--steps", "10", "--warmup", "2", "--n", "1024", "--no-cpu-baseline",
                       env_extra=dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517"))
    assert res["n_gpus"] == 1 and res["config"]["exchange"] == "none" and res["config"]["n_ranks_seen_by_rccl"] == 0


def test_group_of_one_device_equals_the_plain_handle():
    from mppi_numba_amd.mppi import MPPI_Group
    _, cfg, lin, ang, plain, params = build("c2", 2048)
    group = MPPI_Group([cfg])
    group.setup(params, [lin], [ang])
    assert group.planners[0].comm_count() == 1
    want = plain.solve()
    got = group.solve()
    assert np.array_equal(got, want)
    plain.iterate_async(6)
    plain.synchronize()
    group.iterate_async(6)
    group.synchronize()
    assert np.array_equal(group.planners[0].u_cur_d.copy_to_host(), plain.u_cur_d.copy_to_host())
    assert np.array_equal(group.planners[0].costs_d.copy_to_host(), plain.costs_d.copy_to_host())


@pytest.mark.parametrize("workload,n", [("c2", 2048), ("c4", 65536)])
def test_graph_replay_captures_the_rccl_all_gather(workload, n):
    from mppi_numba_amd.mppi import comm_unique_id
    _, _, _, _, direct, _ = build(workload, n)
    _, _, _, _, graph, _ = build(workload, n)
    for planner in (direct, graph):
        planner.comm_init(comm_unique_id())
        assert planner.comm_count() == 1
    graph.set_graph_replay(True, 2)
    for planner in (direct, graph):
        planner.solve()
        planner.iterate_async(9)
        planner.synchronize()
    stats = graph.graph_stats()
    assert stats["captures"] >= 1 and stats["replays"] >= 3, stats
    assert np.array_equal(direct.u_cur_d.copy_to_host(), graph.u_cur_d.copy_to_host())
    assert np.array_equal(direct.costs_d.copy_to_host(), graph.costs_d.copy_to_host())
