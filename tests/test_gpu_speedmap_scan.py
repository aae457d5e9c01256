"""Round 6: the reference's third planner mode -- nominal dynamics, the time of a step charged by
the risk speed map, rollout_det_dyn_w_speed_map_numba mppi.py:1013-1111 -- on the time-parallel kernel of the latency
regime, k_rollout_scan_exact<.., SPEED>: one launch per iteration (noise from the counters, the previous update folded
in, tile packets out).  Its dynamics run on NOMINAL traction (terrain.py:455-463), i.e. the assumption the three walks
rest on holds by construction; the risk byte travels with the 32-bit cell and only enters the stage cost.

Bits of the reference on the seven `speedmap_*` fixtures with the kernel's name asserted, the full-size workloads
against the oracle (tests/test_gpu_scale.py: c2m, c2m1k), the loop with the folded update, rollouts that freeze in the
padding ring, and a map on which the vote fails (grids injected through the C API): re-executed exactly, counted, and
the planner leaves for k_rollout_fused<SPEED>."""
import numpy as np
import pytest

import bench
from gpu_helpers import build_from_golden
from helpers import golden, iterations, ulp_diff_f32
from mppi_numba_amd import _lib
from oracle import oracle as O
from test_gpu_scale import oracle_costs, oracle_params

pytestmark = pytest.mark.gpu

# (the seventh, semantic_speedmap, is an end-to-end fixture -- seed -> sampled grids -> u through set_TDM_from_semantic_grid:
#  tests/test_gpu_parity.py::test_semantic_grid_end_to_end_xoroshiro runs it on this kernel; asserted below by name)
SPEEDMAP_FIXTURES = ["speedmap_cvar", "speedmap_mean", "speedmap_mean_bounds", "speedmap_odd_units",
                     "speedmap_odd_units_w101", "speedmap_odd_units_w202"]
KERNEL = "k_rollout_scan_exact speed_map"


@pytest.mark.parametrize("name", SPEEDMAP_FIXTURES)
def test_reference_fixtures_on_the_time_parallel_kernel(name):
    from gpu_helpers import solve_of_iteration
    g = golden(name)
    _, lin, ang, planner, P = build_from_golden(name, g)
    span = np.array([P["vrange"][1] - P["vrange"][0], P["wrange"][1] - P["wrange"][0]])
    for k, it in enumerate(iterations(g)):
        sol = solve_of_iteration(g, k)
        lin.set_sampled_grids(g["solve%d_lin_sample_grid" % sol])
        ang.set_sampled_grids(g["solve%d_ang_sample_grid" % sol])
        planner.params["x0"] = g["solve%d_x0" % sol]
        planner.set_noise(it["noise"])
        planner.set_u(it["u_in"])
        planner.rollout()
        assert planner.last_rollout_kernel().startswith(KERNEL), planner.last_rollout_kernel()
        got = planner.costs_d.copy_to_host()
        assert ulp_diff_f32(got, it["costs"]).max() == 0, (name, k, int(ulp_diff_f32(got, it["costs"]).max()))
        planner.update()
        assert (np.abs(planner.u_cur_d.copy_to_host().astype(np.float64) - it["u_out"]) / span).max() <= 1e-5


def test_semantic_grid_entry_point_runs_the_time_parallel_kernel():
    """set_TDM_from_semantic_grid in speed-map mode, from the seed (the reference's streams): the end-to-end fixture's u,
    computed by this kernel."""
    from test_host_and_abi import semantic_inputs
    from gpu_helpers import config_from_golden
    from mppi_numba_amd.mppi import MPPI_Numba
    from mppi_numba_amd.terrain import TDM_Numba
    from helpers import params_from_golden
    g = golden("semantic_speedmap")
    values, id2name, name2terrain, terrain2pmf, alpha = semantic_inputs(g)
    cfg = config_from_golden("speedmap", g, rng="xoroshiro")
    res = float(g["in_res"])
    rows, cols = g["in_semantic_grid"].shape
    lin, ang = TDM_Numba(cfg), TDM_Numba(cfg)
    for tdm in (lin, ang):
        tdm.set_TDM_from_semantic_grid(g["in_semantic_grid"], res, len(values), values, np.array([0.0, 1.0]),
                                       (0.0, cols * res), (0.0, rows * res), id2name, name2terrain, terrain2pmf,
                                       det_dynamics_cvar_alpha=alpha, obstacle_map=g["in_obstacle_map"],
                                       unknown_map=g["in_unknown_map"])
    planner = MPPI_Numba(cfg)
    P = params_from_golden(g)
    planner.setup(P, lin, ang)
    useq = planner.solve()
    assert planner.last_rollout_kernel().startswith(KERNEL), planner.last_rollout_kernel()
    span = np.array([P["vrange"][1] - P["vrange"][0], P["wrange"][1] - P["wrange"][0]])
    assert ulp_diff_f32(planner.costs_d.copy_to_host(), iterations(g)[0]["costs"]).max() == 0
    assert (np.abs(useq.astype(np.float64) - g["solve0_useq"]) / span).max() <= 1e-5


def test_loop_is_one_launch_per_iteration_and_matches_the_stage_level_path():
    """iterate_async: the update of iteration k is combined by the rollout launch of iteration k + 1 (reduces_tiles),
    the noise never exists in memory -- and u equals what the same iterations give through rollout() + update()."""
    w, cfg, lin, ang, planner, params = bench.build_planner("c2m")
    planner.solve()
    planner.iterate_async(6)
    planner.synchronize()
    name = planner.last_rollout_kernel()
    assert name.startswith(KERNEL) and "noise=in-kernel" in name and "reduces_tiles=1" in name, name
    u_loop = planner.u_cur_d.copy_to_host()
    # the same seven iterations stage by stage on a second handle (same seed -> same Philox blocks)
    w2, cfg2, lin2, ang2, ref, params2 = bench.build_planner("c2m")
    lin2.sample_grids(1.0)
    ang2.sample_grids(1.0)
    for _ in range(7):
        ref.sample_noise()
        ref.rollout()
        ref.update()
    assert np.array_equal(ref.u_cur_d.copy_to_host(), u_loop)
    # ... and the last iteration's costs are the oracle's
    planner.sample_noise()
    noise, u_in = planner.noise_samples_d.copy_to_host(), planner.u_cur_d.copy_to_host()
    planner.rollout()
    got = planner.costs_d.copy_to_host()
    want = oracle_costs(w, params, lin, ang, noise, u_in)
    assert (ulp_diff_f32(got, want) == 0).mean() >= 0.999 and (np.abs(got - want) / np.abs(want)).max() < 1e-6


def test_rollouts_that_freeze_in_the_padding_ring_pay_that_cells_time():
    """Start two cells from the zero-traction ring, heading out: most rollouts stop there and go on paying the ring
    cell's stage cost (its risk byte is 0: dt / 1e-6 per step) -- the frozen-rollout records of the kernel."""
    w, cfg, lin, ang, planner, params = bench.build_planner("c2m", n=2048)
    params = dict(params, x0=np.array([0.3, 30.0, np.pi]), xgoal=np.array([40.0, 30.0]))
    planner.set_params(params)
    planner.solve()
    planner.sample_noise()
    noise, u_in = planner.noise_samples_d.copy_to_host(), planner.u_cur_d.copy_to_host()
    planner.rollout()
    assert planner.last_rollout_kernel().startswith(KERNEL), planner.last_rollout_kernel()
    got = planner.costs_d.copy_to_host()
    want = oracle_costs(w, params, lin, ang, noise, u_in)
    assert (want > 1e6).mean() > 0.2, "the case must freeze rollouts in the ring"
    assert np.array_equal(got, want), int(ulp_diff_f32(got, want).max())


def test_a_failed_vote_is_reexecuted_exactly_and_the_planner_leaves_for_the_fused_kernel():
    """Traction that changes from cell to cell in speed-map mode (not something the reference's TDM produces: injected
    through the C API): every tile's vote fails, the cost wave re-runs it on the general arithmetic -- same bits --,
    the failures are counted and after the next synchronisation the planner runs k_rollout_fused<SPEED>."""
    w, cfg, lin, ang, planner, params = bench.build_planner("c2m", n=1024)
    planner.solve()
    rng = np.random.default_rng(5)
    grid = lin.sample_grid_batch_d.copy_to_host()
    rows, cols = lin.obstacle_map_d.copy_to_host().shape
    noisy = grid.copy()
    noisy[:, 2:rows - 2, 2:cols - 2] = rng.integers(40, 101, size=(grid.shape[0], rows - 4, cols - 4)).astype(np.int8)
    lin.set_sampled_grids(noisy[:, :rows, :cols])
    ang.set_sampled_grids(noisy[:, :rows, :cols])
    planner.sample_noise()
    noise, u_in = planner.noise_samples_d.copy_to_host(), planner.u_cur_d.copy_to_host()
    planner.rollout()
    assert planner.last_rollout_kernel().startswith(KERNEL), planner.last_rollout_kernel()
    got = planner.costs_d.copy_to_host()
    p = oracle_params(params, lin, ang)
    want = O.rollout_det(p, noisy[:, :rows, :cols], noisy[:, :rows, :cols], lin.obstacle_map_d.copy_to_host(),
                         lin.unknown_map_d.copy_to_host(), noise, u_in, risk=lin.risk_traction_map_d.copy_to_host())
    assert (ulp_diff_f32(got, want) == 0).mean() >= 0.999, int(ulp_diff_f32(got, want).max())
    planner.update()
    planner.iterate_async(2)  # (no TDM sampling: the injected grids stay)
    planner.synchronize()     # the host looks at the failure count here
    planner.set_noise(noise)
    planner.set_u(u_in)
    planner.rollout()
    assert planner.last_rollout_kernel().startswith("k_rollout_fused speed_map"), planner.last_rollout_kernel()
    assert (ulp_diff_f32(planner.costs_d.copy_to_host(), want) == 0).mean() >= 0.999
