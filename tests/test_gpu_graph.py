"""hipGraph replay of the iteration loop (mppi_planner_set_graph_replay): identical results to
the direct loop in every regime, replays actually happen, and a captured graph is dropped as
soon as anything it froze changes (parameters, maps, start state)."""
import numpy as np
import pytest

import bench
from test_gpu_scale import build

pytestmark = pytest.mark.gpu


def pair(workload, n, rng="philox"):
    _, _, lin_a, ang_a, direct, params = build(workload, n, rng=rng)
    _, _, lin_b, ang_b, graphed, _ = build(workload, n, rng=rng)
    graphed.set_graph_replay(True)
    return direct, graphed, params


@pytest.mark.parametrize("workload,n,token", [
    ("c2", 2048, "k_rollout_scan_exact"),  # noise computed in the rollout launch
    ("c2", 12288, "k_rollout_pipe"),     # in-launch generation of the next noise by spare workgroups
    ("c4", 65536, "k_rollout_fused"),    # generation on the second stream, fork / join inside the graph
    ("c3", 192, "k_rollout_tdm_fast"),   # generator in line: captured as part of the iteration
])
def test_graph_replay_equals_the_direct_loop(workload, n, token):
    direct, graphed, _ = pair(workload, n)
    for planner in (direct, graphed):
        planner.solve()
    for chunk in (1, 2, 7, 12, 3):  # odd and even chunk lengths, leftovers before and after the graph
        direct.iterate_async(chunk)
        graphed.iterate_async(chunk)
        direct.synchronize()
        graphed.synchronize()
        assert np.array_equal(direct.u_cur_d.copy_to_host(), graphed.u_cur_d.copy_to_host()), chunk
        assert np.array_equal(direct.costs_d.copy_to_host(), graphed.costs_d.copy_to_host()), chunk
    assert token in graphed.last_rollout_kernel()
    stats = graphed.graph_stats()
    assert stats["captures"] >= 1 and stats["replays"] >= 8, stats
    # switching it off again: still the same sequence
    graphed.set_graph_replay(False)
    direct.iterate_async(4)
    graphed.iterate_async(4)
    direct.synchronize()
    graphed.synchronize()
    assert np.array_equal(direct.costs_d.copy_to_host(), graphed.costs_d.copy_to_host())


def test_graph_replay_with_the_numba_compatible_generator():
    direct, graphed, _ = pair("c2", 1024, rng="xoroshiro")
    for planner in (direct, graphed):
        planner.solve()
        planner.iterate_async(9)
        planner.synchronize()
    assert np.array_equal(direct.u_cur_d.copy_to_host(), graphed.u_cur_d.copy_to_host())
    assert graphed.graph_stats()["replays"] >= 3


def test_longer_graphs():
    """Eight iterations per graph: leftovers of every length run directly."""
    direct, graphed, _ = pair("c2", 2048)
    graphed.set_graph_replay(True, iterations_per_graph=8)
    for planner in (direct, graphed):
        planner.solve()
    for chunk in (3, 8, 21, 16, 1):
        direct.iterate_async(chunk)
        graphed.iterate_async(chunk)
        direct.synchronize()
        graphed.synchronize()
        assert np.array_equal(direct.u_cur_d.copy_to_host(), graphed.u_cur_d.copy_to_host()), chunk
    stats = graphed.graph_stats()
    assert stats["replays"] >= 4 and stats["captures"] <= 4, stats  # (one graph per parity of the control and tile-packet buffers)
    from mppi_numba_amd import _lib
    with pytest.raises(_lib.MppiError, match="even"):
        graphed.set_graph_replay(True, iterations_per_graph=3)


def test_graph_is_recaptured_when_frozen_arguments_change():
    direct, graphed, params = pair("c2", 2048)
    for planner in (direct, graphed):
        planner.params["num_opt"] = 6
        planner.solve()
    before = graphed.graph_stats()
    assert before["replays"] >= 2
    # closed loop: a new start state and goal, a different temperature, shifted controls
    for step in range(4):
        for planner in (direct, graphed):
            planner.params["x0"] = np.array([6.0 + step, 5.0 + 0.5 * step, 0.3 * step])
            planner.params["lambda_weight"] = 1.0 + 0.5 * step
            planner.shift_and_update_on_device(planner.params["x0"], 1)
        u_d, u_g = direct.solve(), graphed.solve()
        assert np.array_equal(u_d, u_g), step
    after = graphed.graph_stats()
    assert after["captures"] >= before["captures"] + 4
    # unchanged parameters: no new capture
    for planner in (direct, graphed):
        planner.solve()
    assert graphed.graph_stats()["captures"] == after["captures"]
    assert np.array_equal(direct.u_cur_d.copy_to_host(), graphed.u_cur_d.copy_to_host())


def test_batched_handle_keeps_its_graph_across_new_start_states():
    """Start / goal of a batched handle live in device memory: new states, same graph."""
    from mppi_numba_amd.batch import MPPI_Batch
    from test_gpu_batch import make_world, problems
    cfg, lin, ang, params = make_world("c2", 1024, 50)
    params = dict(params, num_opt=5)
    rng = np.random.default_rng(4)
    handles = []
    for graph in (False, True):
        b = MPPI_Batch(cfg, 4)
        x0s, goals = problems(lin, 4, np.random.default_rng(4))
        b.setup(params, lin, ang, x0s, goals)
        if graph:
            b.set_graph_replay(True)
        handles.append(b)
    direct, graphed = handles
    x0s, goals = problems(lin, 4, np.random.default_rng(4))
    for step in range(5):
        u_d, u_g = direct.solve(), graphed.solve()
        assert np.array_equal(u_d, u_g), step
        x0s = (x0s + rng.uniform(-0.3, 0.3, x0s.shape)).astype(np.float32)
        x0s[:, 0] = np.clip(x0s[:, 0], 1.0, 60.0)
        x0s[:, 1] = np.clip(x0s[:, 1], 1.0, 60.0)
        direct.shift_and_update(x0s, u_d, 1)
        graphed.shift_and_update(x0s, u_g, 1)
    stats = graphed.graph_stats()
    assert stats["captures"] <= 2 and stats["replays"] >= 8, stats


def test_graph_replay_of_a_sharded_handle_needs_its_communicator():
    """A host-staged exchange cannot be captured; with an RCCL communicator the all-gather is
    part of the graph (tests/test_gpu_multi.py)."""
    from mppi_numba_amd import _lib
    _, _, _, _, half, _ = build("c2", 1024, rank=0, world=2)
    with pytest.raises(_lib.MppiError, match="communicator"):
        half.set_graph_replay(True)
