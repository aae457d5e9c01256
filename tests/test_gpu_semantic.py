"""The C2 shape (N = 8192, T = 100) on maps where the time-parallel kernel's assumption -- one
traction value along every tile's path -- does NOT hold: bench.py's `c2s`
(a semantic map as the reference builds it in deterministic-dynamics mode, terrain.py:183-342: four
terrain types in patches of 4-16 m, traction piecewise constant) and `c2c` (CVaR-bin traction that
changes from cell to cell, as C4).

A tile whose vote fails is re-executed inside the same launch on the exact three-wave pipelined
schedule (rollout_scan_exact_kernel.h: scan_exact_reexecute -- round 3 rolled it out with one wave,
~300 us per launch); the planner counts failed tiles per LAUNCH and, at its next synchronisation,
stops speculating on such a map: round 5 keeps the kernel and launches it `direct` (every tile on the
exact schedule at once: one launch per iteration), round 4 took k_rollout_pipe + k_update_rows (still
there behind MPPI_DEBUG_NO_SCAN_DIRECT).  Costs are the oracle's bits either way (mppi.py:916-1009)."""
import numpy as np
import pytest

import bench
from helpers import ulp_diff_f32
from mppi_numba_amd import _lib
from test_gpu_scale import oracle_costs

pytestmark = pytest.mark.gpu


def stage_level_iteration(planner, w, params, lin, ang):
    planner.sample_noise()
    noise, u_in = planner.noise_samples_d.copy_to_host(), planner.u_cur_d.copy_to_host()
    planner.rollout()
    name = planner.last_rollout_kernel()
    got = planner.costs_d.copy_to_host()
    want = oracle_costs(w, params, lin, ang, noise, u_in)
    ulps = ulp_diff_f32(got, want)
    assert (ulps == 0).mean() >= 0.999, (name, (ulps == 0).mean(), ulps.max())
    assert (np.abs(got - want) / np.abs(want)).max() < 1e-6, name
    planner.update()
    return name


@pytest.mark.parametrize("workload", ["c2s", "c2c"])
def test_failed_votes_are_reexecuted_pipelined_and_the_planner_then_takes_the_exact_kernel(workload):
    w, cfg, lin, ang, planner, params = bench.build_planner(workload)
    lin.sample_grids()
    ang.sample_grids()
    # nothing is known about the map yet: the time-parallel kernel, most of its tiles fail their vote
    name = stage_level_iteration(planner, w, params, lin, ang)
    assert name.startswith("k_rollout_scan_exact") and "failed_tiles=pipelined" in name, name
    planner.solve()  # the host synchronises and sees the failed tiles
    # round 5: the same kernel without the time-parallel attempt -- every tile on the exact three-wave schedule at
    # once, noise from the counters, previous update folded in, tile packets out: one launch per iteration
    name = stage_level_iteration(planner, w, params, lin, ang)
    assert name.startswith("k_rollout_scan_exact") and "direct=1" in name, name
    planner.iterate_async(4)
    planner.synchronize()
    name = planner.last_rollout_kernel()
    assert name.startswith("k_rollout_scan_exact") and "direct=1" in name and "reduces_tiles=1" in name, name
    # ... and round 4's choice behind a developer switch: the bits of the oracle as well
    planner.set_debug_flags(_lib.DEBUG_NO_SCAN_DIRECT)
    name = stage_level_iteration(planner, w, params, lin, ang)
    assert name.startswith("k_rollout_pipe"), name


@pytest.mark.parametrize("workload", ["c2s", "c2c"])
def test_loop_that_keeps_speculating_has_the_bits_of_the_exact_kernel(workload):
    """MPPI_DEBUG_KEEP_SPECULATING: every launch of the loop re-executes its failed tiles (and applies the
    previous update itself); the loop that runs k_rollout_pipe + k_update_rows is the yardstick.  Costs
    are bit-identical launch by launch; u follows another summation tree (tiles of 32 vs 64 rollouts)."""
    _, _, _, _, spec, params = bench.build_planner(workload, 4096)
    _, _, _, _, exact, _ = bench.build_planner(workload, 4096)
    spec.set_debug_flags(_lib.DEBUG_KEEP_SPECULATING)
    exact.set_debug_flags(_lib.DEBUG_NO_SCAN_KERNEL | _lib.DEBUG_NO_DEEP_KERNEL | _lib.DEBUG_NO_SPEC_KERNEL)
    for planner in (spec, exact):
        planner.solve()
        planner.iterate_async(1)
        planner.synchronize()
    assert spec.last_rollout_kernel().startswith("k_rollout_scan_exact"), spec.last_rollout_kernel()
    assert exact.last_rollout_kernel().startswith("k_rollout_pipe"), exact.last_rollout_kernel()
    span = np.array([params["vrange"][1] - params["vrange"][0], params["wrange"][1] - params["wrange"][0]])
    du = float((np.abs(spec.u_cur_d.copy_to_host() - exact.u_cur_d.copy_to_host()) / span).max())
    assert du <= 2e-6, du


@pytest.mark.parametrize("workload", ["c2s", "c2c"])
def test_direct_exact_schedule_has_the_bits_of_the_loop_that_keeps_speculating(workload):
    """Round 5: once the planner has stopped speculating the kernel runs `direct`.  Same noise counters, same
    arithmetic per rollout, same tiles of 32 rollouts, same packets, same fold: the control sequence of a loop that
    switches (default) and of one that never does (MPPI_DEBUG_KEEP_SPECULATING) is the same BITS after every call."""
    _, _, _, _, spec, _ = bench.build_planner(workload)
    _, _, _, _, auto, _ = bench.build_planner(workload)
    spec.set_debug_flags(_lib.DEBUG_KEEP_SPECULATING)
    for call in range(3):
        for planner in (spec, auto):
            planner.solve()
            planner.iterate_async(5)
            planner.synchronize()
        assert np.array_equal(spec.u_cur_d.copy_to_host().view(np.uint32), auto.u_cur_d.copy_to_host().view(np.uint32)), call
        assert np.array_equal(spec.costs_d.copy_to_host().view(np.uint32), auto.costs_d.copy_to_host().view(np.uint32)), call
    assert "direct=1" in auto.last_rollout_kernel() and "direct=1" not in spec.last_rollout_kernel()


@pytest.mark.parametrize("workload", ["c2s"])
def test_graph_replay_of_the_direct_loop_has_the_bits_of_the_direct_launches(workload):
    """hipGraph replay (opt-in) over a map the planner has stopped speculating on: the captured launches are the
    direct ones (fold, buffer parities and all); the same bits as the loop launched directly."""
    _, _, _, _, plain, _ = bench.build_planner(workload, 4096)
    _, _, _, _, graph, _ = bench.build_planner(workload, 4096)
    graph.set_graph_replay(True, 4)
    for planner in (plain, graph):
        planner.solve()          # (speculative launch, failed tiles; the host sees them)
        planner.iterate_async(3)
        planner.synchronize()
    for call in range(3):
        for planner in (plain, graph):
            planner.iterate_async(9)   # warm-up iteration + two graphs of four
            planner.synchronize()
        assert "direct=1" in graph.last_rollout_kernel(), graph.last_rollout_kernel()
        assert np.array_equal(plain.u_cur_d.copy_to_host().view(np.uint32), graph.u_cur_d.copy_to_host().view(np.uint32)), call
        assert np.array_equal(plain.costs_d.copy_to_host().view(np.uint32), graph.costs_d.copy_to_host().view(np.uint32)), call
