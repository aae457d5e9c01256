"""One GPU, iteration loop of the time-parallel exact kernel: an iteration is ONE launch.  The
rollout launch leaves one packet per tile of 32 rollouts; workgroup t of the NEXT launch combines
them for step t -- the very function block t of k_combine_tiles runs -- and publishes u[t] to all
workgroups of that launch (update_kernels.h: PendingApply::reduce_tiles; update_useq_numba
mppi.py:1113-1191).  Only the last iteration of a call runs k_combine_tiles.

Checked here: the loop with the fold against the loop with an update launch per iteration
(MPPI_DEBUG_NO_REDUCE_FOLD): the SAME BITS, u and costs, after any number of iterations and calls;
ragged tile counts, fewer workgroups than steps, horizons on both sides of 64 steps; run-to-run
determinism; graph replay (bits of the direct loop); and the oracle end to end -- a stage-level
iteration placed behind a folded loop, and the costs of a folded launch itself."""
import numpy as np
import pytest

import bench
from mppi_numba_amd import _lib
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def build(n, t=None, seed=1, math="exact"):
    if t is not None:
        saved = dict(bench.WORKLOADS["c2"])
        bench.WORKLOADS["c2"] = dict(saved, t=t)
    try:
        return bench.build_planner("c2", n, seed=seed, math=math)
    finally:
        if t is not None:
            bench.WORKLOADS["c2"] = saved


def span(params):
    return np.array([params["vrange"][1] - params["vrange"][0], params["wrange"][1] - params["wrange"][0]])


def run(planner, iterations, calls=1):
    for _ in range(calls):
        planner.iterate_async(iterations)
        planner.synchronize()
    return planner.u_cur_d.copy_to_host(), planner.costs_d.copy_to_host()


@pytest.mark.parametrize("n,t", [(8192, 100), (8192, 64), (4096, 30), (1000, 100), (200, 100), (40, 17), (32, 100), (256, 104)])
def test_loop_with_the_fold_equals_the_loop_with_update_launches(n, t):
    _, _, _, _, folded, params = build(n, t)
    _, _, _, _, plain, _ = build(n, t)
    plain.set_debug_flags(_lib.DEBUG_NO_REDUCE_FOLD)
    for planner in (folded, plain):
        planner.solve()
    can_fold = 4 * ((n + 31) // 32) >= t
    for iterations in (1, 2, 5, 8):
        u_f, c_f = run(folded, iterations)
        u_p, c_p = run(plain, iterations)
        name = folded.last_rollout_kernel()
        assert name.startswith("k_rollout_scan_exact"), name
        assert ("reduces_tiles=1" in name) == (iterations > 1 and can_fold), (iterations, name)
        assert "reduces_tiles" not in plain.last_rollout_kernel()
        assert np.array_equal(u_f, u_p), (iterations, float((np.abs(u_f - u_p) / span(params)).max()))
        assert np.array_equal(c_f, c_p), iterations


def test_the_fold_is_deterministic():
    got = []
    for _ in range(3):
        _, _, _, _, planner, _ = build(8192)
        planner.solve()
        got.append(run(planner, 7, calls=3))
    for u, c in got[1:]:
        assert np.array_equal(u, got[0][0]) and np.array_equal(c, got[0][1])


def test_graph_replay_of_the_folded_loop_has_the_bits_of_the_direct_loop():
    for per_graph in (2, 4):
        _, _, _, _, direct, _ = build(4096)
        _, _, _, _, graphed, _ = build(4096)
        graphed.set_graph_replay(True, iterations_per_graph=per_graph)
        for planner in (direct, graphed):
            planner.solve()
        for chunk in (1, 2, 7, 12, 3, 8):
            u_d, c_d = run(direct, chunk)
            u_g, c_g = run(graphed, chunk)
            assert np.array_equal(u_d, u_g), (per_graph, chunk)
            assert np.array_equal(c_d, c_g), (per_graph, chunk)
        stats = graphed.graph_stats()
        assert stats["replays"] >= 6, stats


def test_stage_level_iteration_behind_a_folded_loop_vs_oracle():
    """The folded loop leaves the planner where the stage-level calls can take over: noise from the
    generator, costs against the oracle (bits), the update against the oracle's (1e-5 of the range)."""
    w, cfg, lin, ang, planner, params = bench.build_planner("c2", 8192)
    planner.solve()
    run(planner, 6)
    assert "reduces_tiles=1" in planner.last_rollout_kernel()
    planner.sample_noise()
    noise = planner.noise_samples_d.copy_to_host()
    u_in = planner.u_cur_d.copy_to_host()
    planner.rollout()
    got = planner.costs_d.copy_to_host()
    p = O.make_params(params, lin.res, lin.padded_xlimits, lin.padded_ylimits,
                      lin.bin_values_bounds_d.copy_to_host(), ang.bin_values_bounds_d.copy_to_host())
    want = O.rollout_det(p, lin.sample_grid_batch_d.copy_to_host(), ang.sample_grid_batch_d.copy_to_host(),
                         lin.obstacle_map_d.copy_to_host(), lin.unknown_map_d.copy_to_host(), noise, u_in)
    assert (got == want).mean() >= 0.999
    planner.update()
    _, u_ref, _ = O.update_useq(params["lambda_weight"], want, noise, params["vrange"], params["wrange"], u_in)
    assert float((np.abs(planner.u_cur_d.copy_to_host() - u_ref) / span(params)).max()) <= 1e-5


def test_the_loop_noise_of_a_folded_iteration_against_the_oracle():
    """Two iterations in one call: the second launch reduces and applies the first one's update.  The
    first call of a twin handle stops after one iteration (k_combine_tiles): its u is what the fold
    must have formed (to the summation order), and the second launch's costs must be the oracle's for
    that u and the noise of its Philox block -- bit for bit, as the fold forms the twin's u exactly."""
    _, _, lin, ang, two, params = bench.build_planner("c2", 2048)
    _, _, _, _, one, _ = bench.build_planner("c2", 2048)
    for planner in (one, two):
        planner.solve()
    run(one, 1)
    u1 = one.u_cur_d.copy_to_host()
    run(two, 2)
    assert "reduces_tiles=1" in two.last_rollout_kernel()
    noise = two.noise_samples_d.copy_to_host()  # (regenerated from the counters of the last launch)
    costs = two.costs_d.copy_to_host()
    p = O.make_params(params, lin.res, lin.padded_xlimits, lin.padded_ylimits,
                      lin.bin_values_bounds_d.copy_to_host(), ang.bin_values_bounds_d.copy_to_host())
    want = O.rollout_det(p, lin.sample_grid_batch_d.copy_to_host(), ang.sample_grid_batch_d.copy_to_host(),
                         lin.obstacle_map_d.copy_to_host(), lin.unknown_map_d.copy_to_host(), noise, u1)
    rel = np.abs(costs - want) / np.maximum(np.abs(want), 30.0)
    print("\nfolded second iteration vs oracle on the twin's u: bit-identical %.4f, max rel %.2e" % ((costs == want).mean(), rel.max()))
    assert (costs == want).mean() >= 0.999 and rel.max() <= 1e-6


@pytest.mark.parametrize("n,t", [(8192, 100), (8192, 128), (4096, 30), (1000, 100), (200, 100), (40, 17), (64, 100), (40, 5), (100, 8)])
def test_fast_mode_loop_with_the_fold_equals_the_loop_with_update_launches(n, t):
    """math="fast" (k_rollout_scan, the tolerance kernel): the same fold, the same statement -- u and costs of the loop
    with one launch per iteration are the bits of the loop with an update launch per iteration."""
    _, _, _, _, folded, params = build(n, t, math="fast")
    _, _, _, _, plain, _ = build(n, t, math="fast")
    plain.set_debug_flags(_lib.DEBUG_NO_REDUCE_FOLD)
    for planner in (folded, plain):
        planner.solve()
    for iterations in (1, 2, 5, 8):
        u_f, c_f = run(folded, iterations)
        u_p, c_p = run(plain, iterations)
        name = folded.last_rollout_kernel()
        assert name.startswith("k_rollout_scan ") or name.startswith("k_rollout_scan_exact"), name
        if name.startswith("k_rollout_scan "):
            tile = int(name.split("tile=")[1].split()[0])
            can_fold = 4 * ((n + tile - 1) // tile) >= t
            assert ("reduces_tiles=1" in name) == (iterations > 1 and can_fold), (iterations, name)
        assert "reduces_tiles" not in plain.last_rollout_kernel()
        assert np.array_equal(u_f, u_p), (iterations, float((np.abs(u_f - u_p) / span(params)).max()))
        assert np.array_equal(c_f, c_p), iterations


def test_fast_mode_graph_replay_of_the_folded_loop_has_the_bits_of_the_direct_loop():
    _, _, _, _, direct, _ = build(4096, math="fast")
    _, _, _, _, graphed, _ = build(4096, math="fast")
    graphed.set_graph_replay(True, iterations_per_graph=2)
    for planner in (direct, graphed):
        planner.solve()
    for chunk in (1, 2, 7, 12, 3):
        u_d, c_d = run(direct, chunk)
        u_g, c_g = run(graphed, chunk)
        assert np.array_equal(u_d, u_g) and np.array_equal(c_d, c_g), chunk
