"""k_rollout_scan (math="fast"): the rollout parallel over the horizon, with the noise computed in
the launch and the update reduced per tile, against the oracle -- BASELINE configs[1] at full
size, every variant pinned by name, the vote's fallback on a map whose traction changes from cell
to cell, frozen rollouts, goal breaks, ragged sizes, shards.

Tolerance mode: prefix sums over the horizon cannot reproduce the
reference's T sequential float32 roundings of heading and position; the COST accumulation is
walked in the reference's order.  Gates: 99.9 % of the costs within 1e-6 relative (most of them
bit-identical); the update, GIVEN the costs, within 1e-5 of the control range; and end to end u
within 1e-5 of the control range at BASELINE configs[1] (north_star's bar).  End to end elsewhere
the distance is printed and bounded loosely: the weights are exp(-cost / lambda), so ONE ulp of a
cost of 1e4 (1e-3) moves a weight by 0.1 % when lambda = 1, and u by up to that times the weight's
share -- any kernel that is not bit-identical on every rollout that carries weight sits there."""
import numpy as np
import pytest

import bench
from mppi_numba_amd import _lib
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def build(workload="c2", n=None, math="fast", **kw):
    return bench.build_planner(workload, n, math=math, **kw)


def oracle_costs(params, lin, ang, noise, u):
    p = O.make_params(params, lin.res, lin.padded_xlimits, lin.padded_ylimits,
                      lin.bin_values_bounds_d.copy_to_host(), ang.bin_values_bounds_d.copy_to_host())
    return O.rollout_det(p, lin.sample_grid_batch_d.copy_to_host(), ang.sample_grid_batch_d.copy_to_host(),
                         lin.obstacle_map_d.copy_to_host(), lin.unknown_map_d.copy_to_host(), noise, u)


def span(params):
    return np.array([params["vrange"][1] - params["vrange"][0], params["wrange"][1] - params["wrange"][0]])


def one_stage_level_iteration(planner, params, lin, ang):
    planner.sample_noise()
    noise = planner.noise_samples_d.copy_to_host()
    u_in = planner.u_cur_d.copy_to_host()
    planner.rollout()
    got = planner.costs_d.copy_to_host()
    want = oracle_costs(params, lin, ang, noise, u_in)
    planner.update()
    u_out = planner.u_cur_d.copy_to_host()
    _, u_ref, _ = O.update_useq(params["lambda_weight"], want, noise, params["vrange"], params["wrange"], u_in)
    # the update alone: the reference's update applied to THIS kernel's costs
    _, u_own, _ = O.update_useq(params["lambda_weight"], got, noise, params["vrange"], params["wrange"], u_in)
    given_costs = float((np.abs(u_out - u_own) / span(params)).max())
    assert given_costs <= 1e-5, given_costs
    rel = np.abs(got - want) / np.maximum(np.abs(want), 30.0)
    return rel, float((np.abs(u_out - u_ref) / span(params)).max()), got, want


VARIANTS = {
    "default": (0, "tile=32"),
    "full_tiles": (_lib.DEBUG_SCAN_FULL_TILES, "tile=64"),
}


@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_c2_costs_and_update_vs_oracle(variant):
    flags, tag = VARIANTS[variant]
    w, cfg, lin, ang, planner, params = build("c2")
    planner.set_debug_flags(flags)
    planner.solve()
    planner.iterate_async(5)
    planner.synchronize()
    rel, margin, got, want = one_stage_level_iteration(planner, params, lin, ang)
    name = planner.last_rollout_kernel()
    assert name.startswith("k_rollout_scan") and "noise=read" in name and tag in name, name
    q = np.quantile(rel, [0.5, 0.99, 0.999, 1.0])
    print("\n%s: bit-identical %.4f, rel quantiles %s, max |du|/range %.2e" % (variant, (got == want).mean(), q, margin))
    assert np.quantile(rel, 0.999) < 1e-6, q
    assert (rel < 1e-5).mean() >= 0.9995
    assert (got == want).mean() > 0.6
    assert margin <= 1e-5
    assert abs(planner.weights_d.copy_to_host().sum() - 1.0) < 1e-5


def test_loop_generates_its_noise_and_matches_the_stage_level_sequence():
    """iterate_async() runs the kernel with in-launch Philox noise and never stores it; the same
    iterations driven stage by stage (sample_noise -> rollout -> update) read stored noise of the
    same counters: same u bit for bit, and the noise handed out afterwards is the last iteration's."""
    _, _, lin_a, ang_a, a, params = build("c2", 4096)
    _, _, lin_b, ang_b, b, _ = build("c2", 4096)
    a.solve()
    b.solve()
    assert "noise=in-kernel" in a.last_rollout_kernel(), a.last_rollout_kernel()
    np.testing.assert_array_equal(a.u_cur_d.copy_to_host(), b.u_cur_d.copy_to_host())
    a.iterate_async(4)
    a.synchronize()
    for _ in range(4):
        b.sample_noise()
        b.rollout()
        b.update()
    assert "noise=read" in b.last_rollout_kernel()
    np.testing.assert_array_equal(a.noise_samples_d.copy_to_host(), b.noise_samples_d.copy_to_host())
    np.testing.assert_array_equal(a.costs_d.copy_to_host(), b.costs_d.copy_to_host())
    np.testing.assert_array_equal(a.u_cur_d.copy_to_host(), b.u_cur_d.copy_to_host())
    u_ref = a.u_cur_d.copy_to_host()
    # and the loops that store their noise (64-rollout tiles: the spare CUs write the next block)
    for flags in (_lib.DEBUG_SCAN_READ_NOISE, _lib.DEBUG_SCAN_READ_NOISE | _lib.DEBUG_SCAN_FULL_TILES):
        _, _, _, _, c, _ = build("c2", 4096)
        c.set_debug_flags(flags)
        c.solve()
        c.iterate_async(4)
        c.synchronize()
        assert "noise=read" in c.last_rollout_kernel(), c.last_rollout_kernel()
        if flags & _lib.DEBUG_SCAN_FULL_TILES:
            assert "noise_blocks=0" not in c.last_rollout_kernel(), c.last_rollout_kernel()
            # (other tiles, other partial sums: float32 resolution)
            assert (np.abs(u_ref - c.u_cur_d.copy_to_host()) / span(params)).max() <= 2e-6
        else:
            np.testing.assert_array_equal(u_ref, c.u_cur_d.copy_to_host())


def test_batched_handle_and_solve_results():
    """Per problem of a batched handle (own tiles, own controls): every problem equals a
    single-problem handle given the same noise and controls; solve() hands out the u of its last
    iteration (the host-mapped mirror is written by that iteration only)."""
    from mppi_numba_amd.batch import MPPI_Batch
    w, cfg, lin, ang, _, params = build("c2", 1024)
    planner = MPPI_Batch(cfg, 6)
    x0s, goals = bench.batch_problems(6, np.random.default_rng(5))
    planner.setup(params, lin, ang, x0s, goals)
    planner.solve()
    planner.iterate_async(3)
    planner.synchronize()
    name = planner.last_rollout_kernel()
    assert name.startswith("k_rollout_scan") and "problems=6" in name, name
    planner.sample_noise()
    noise = planner.noise_samples_d.copy_to_host().reshape(6, 1024, 100, 2)
    u_in = planner.u_cur_d.copy_to_host().reshape(6, 100, 2)
    planner.rollout()
    costs = planner.costs_d.copy_to_host().reshape(6, 1024)
    planner.update()
    u_out = planner.u_cur_d.copy_to_host().reshape(6, 100, 2)
    for b in (0, 3, 5):
        _, _, lin1, ang1, single, _ = build("c2", 1024)
        p1 = dict(params, x0=x0s[b], xgoal=goals[b])
        single.setup(p1, lin1, ang1)
        single.lin_tdm.sample_grids()
        single.ang_tdm.sample_grids()
        single.set_u(u_in[b])
        single.set_noise(noise[b])
        single.rollout()
        np.testing.assert_array_equal(single.costs_d.copy_to_host(), costs[b])
        single.update()
        np.testing.assert_array_equal(single.u_cur_d.copy_to_host(), u_out[b])
    # solve(): the returned sequence is the device's
    for num_opt in (1, 3):
        _, _, lin2, ang2, s2, p2 = build("c2", 2048)
        p2 = dict(p2, num_opt=num_opt)
        s2.setup(p2, lin2, ang2)
        got = s2.solve()
        np.testing.assert_array_equal(got, s2.u_cur_d.copy_to_host())


def planner_on(workload, n, t, seed=3):
    """A fast-math planner of `n` rollouts and `t` steps over the world of a bench workload."""
    import copy
    from mppi_numba_amd.config import Config
    from mppi_numba_amd.mppi import MPPI_Numba
    w, cfg, lin, ang, _, params = build(workload, 256)
    cfg2 = Config(T=t * 0.1, dt=0.1, num_grid_samples=1, num_control_rollouts=n, max_speed_padding=5.0,
                  num_vis_state_rollouts=1, max_map_dim=(260, 260), seed=seed, enforce_recommended_limits=False,
                  math="fast", use_det_dynamics=True)
    assert cfg2.num_steps == t
    planner = MPPI_Numba(cfg2)
    planner.setup(copy.deepcopy(params), lin, ang)
    return lin, ang, planner, params


def test_vote_fails_on_a_cellwise_map_and_the_tile_is_rerun():
    """BASELINE configs[3]'s map (CVaR bin of a random 16-bin PMF per cell): the traction changes
    from cell to cell, every tile fails its vote and is rolled out sequentially."""
    lin, ang, planner, params = planner_on("c4", 8192, 100)
    planner.set_debug_flags(_lib.DEBUG_KEEP_SPECULATING)
    planner.solve()
    rel, margin, got, want = one_stage_level_iteration(planner, params, lin, ang)
    assert planner.last_rollout_kernel().startswith("k_rollout_scan"), planner.last_rollout_kernel()
    q = np.quantile(rel, [0.5, 0.99, 0.999, 1.0])
    print("\ncell-wise map: bit-identical %.4f, rel quantiles %s, max |du|/range %.2e" % ((got == want).mean(), q, margin))
    assert (rel < 1e-5).mean() >= 0.995, q
    assert margin <= 2e-3
    # left to itself the planner stops launching the kernel on such a map
    planner.set_debug_flags(0)
    planner.iterate_async(3)
    planner.synchronize()
    planner.iterate_async(1)
    planner.synchronize()
    assert not planner.last_rollout_kernel().startswith("k_rollout_scan"), planner.last_rollout_kernel()


@pytest.mark.parametrize("x0,goal,quantile", [
    ((1.0, 30.0, np.pi), (60.0, 60.0), 0.995),    # most rollouts end frozen in the padding ring
    ((4.0, 4.0, np.pi / 4), (9.0, 9.0), 0.995),   # goal reached within the first chunks
    ((4.0, 4.0, np.pi / 4), (4.2, 4.1), 0.99),    # start inside the goal circle
    ((32.0, 32.0, -2.0), (10.0, 50.0), 0.999),    # the middle of the map
])
def test_events_vs_oracle(x0, goal, quantile):
    w, cfg, lin, ang, planner, params = build("c2", 4096)
    params = dict(params, x0=np.array(x0), xgoal=np.array(goal))
    planner.setup(params, lin, ang)
    planner.solve()
    planner.iterate_async(2)
    planner.synchronize()
    rel, margin, got, want = one_stage_level_iteration(planner, params, lin, ang)
    assert planner.last_rollout_kernel().startswith("k_rollout_scan"), planner.last_rollout_kernel()
    q = np.quantile(rel, [0.5, 0.99, 0.999, 1.0])
    print("\nx0=%s goal=%s: bit-identical %.4f, rel quantiles %s, max |du|/range %.2e" % (x0, goal, (got == want).mean(), q, margin))
    assert np.quantile(rel, quantile) < 1e-6, q
    assert margin <= 2e-3


@pytest.mark.parametrize("n,t,flags", [(1000, 100, 0), (64, 37, 0), (8192, 128, 0), (70, 8, 0), (4096, 2, 0), (33, 99, 0),
                                       (1000, 100, 128), (8192, 120, 128), (100, 5, 128)])
def test_ragged_sizes(n, t, flags):
    lin, ang, planner, params = planner_on("c2", n, t)
    planner.set_debug_flags(flags)
    planner.solve()
    rel, margin, got, want = one_stage_level_iteration(planner, params, lin, ang)
    assert planner.last_rollout_kernel().startswith("k_rollout_scan"), planner.last_rollout_kernel()
    print("\nn=%d t=%d: bit-identical %.4f, max |du|/range %.2e" % (n, t, (got == want).mean(), margin))
    assert np.quantile(rel, 0.99) < 1e-6 and margin <= 2e-3, (np.quantile(rel, [0.5, 0.99, 1.0]), margin)


def test_horizon_beyond_16_chunks_takes_another_kernel():
    lin, ang, planner, params = planner_on("c2", 1024, 200)
    planner.solve()
    assert not planner.last_rollout_kernel().startswith("k_rollout_scan")


def test_sharded_packets_equal_the_unsharded_update():
    """Control samples over 2 / 8 ranks (one GPU, packets gathered by hand): k_combine_tiles writes
    the rank packet, k_apply combines them -- the u of the unsharded handle to float32 resolution."""
    w, cfg, lin, ang, whole, params = build("c2", 8192)
    whole.solve()
    whole.iterate_async(3)
    whole.synchronize()
    u_whole = whole.u_cur_d.copy_to_host()
    for world in (2, 8):
        shards = [bench.build_planner("c2", 8192 // world, rank=r, world=world, math="fast")[4] for r in range(world)]
        for s in shards:
            s.lin_tdm.sample_grids()  # (solve() does this; the stage-level calls do not)
            s.ang_tdm.sample_grids()
        for _ in range(4):
            packets = []
            for s in shards:
                s.sample_noise()
                s.rollout()
                assert s.last_rollout_kernel().startswith("k_rollout_scan")
                packets.append(s.update_local())
            for s in shards:
                s.update_apply(np.stack(packets))
        for s in shards:
            assert (np.abs(s.u_cur_d.copy_to_host() - u_whole) / span(params)).max() <= 2e-6
        np.testing.assert_array_equal(shards[0].u_cur_d.copy_to_host(), shards[-1].u_cur_d.copy_to_host())


def test_graph_replay_of_the_generating_loop():
    _, _, _, _, a, _ = build("c2", 4096)
    _, _, _, _, b, _ = build("c2", 4096)
    a.solve()
    b.solve()
    b.set_graph_replay(True, 2)
    a.iterate_async(9)
    a.synchronize()
    b.iterate_async(9)
    b.synchronize()
    assert b.graph_stats()["replays"] >= 3
    np.testing.assert_array_equal(a.u_cur_d.copy_to_host(), b.u_cur_d.copy_to_host())
    np.testing.assert_array_equal(a.noise_samples_d.copy_to_host(), b.noise_samples_d.copy_to_host())
