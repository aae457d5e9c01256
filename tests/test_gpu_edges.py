"""Edge sizes of the hot path against the oracle: single rollouts, ragged tiles, horizons
shorter than a pipeline chunk, one traction sample, sample counts around the wave size and
beyond a workgroup (the reference's 'oversized' path), costs and the update."""
import numpy as np
import pytest

import bench
from helpers import ulp_diff_f32
from oracle import oracle as O
from test_gpu_scale import custom_world, oracle_costs

pytestmark = pytest.mark.gpu


def build(mode, n, t_steps, m):
    from mppi_numba_amd.config import Config
    from mppi_numba_amd.mppi import MPPI_Numba
    from mppi_numba_amd.terrain import TDM_Numba
    pmf, obstacle, unknown, td = custom_world(60, 50, 0.25, seed=n + 7 * t_steps + m)
    cfg = Config(T=t_steps * 0.1 + 0.05, dt=0.1, num_grid_samples=m, num_control_rollouts=n, max_speed_padding=4.0,
                 num_vis_state_rollouts=1, max_map_dim=(64, 54), seed=5, enforce_recommended_limits=False, **mode)
    assert cfg.num_steps == t_steps, cfg.num_steps
    lin, ang = TDM_Numba(cfg), TDM_Numba(cfg)
    lin.set_TDM_from_PMF_grid(pmf, td, obstacle, unknown)
    ang.set_TDM_from_PMF_grid(pmf[:, :, ::-1].copy(), td, obstacle, unknown)
    planner = MPPI_Numba(cfg)
    params = bench.make_params("c3" if m > 1 else "c2")
    params.update(x0=np.array([6.1, 7.3, 0.4]), xgoal=np.array([7.0, 8.2]), lambda_weight=3.0, goal_tolerance=0.4)
    planner.setup(params, lin, ang)
    return cfg, lin, ang, planner, params


@pytest.mark.parametrize("n,t_steps", [(1, 1), (1, 9), (2, 2), (63, 7), (64, 8), (65, 17), (130, 1), (257, 33),
                                       (130, 1500),   # long horizon, still inside the incremental trig's proof
                                       (70, 2500)])   # beyond it (T > 2000): the full-sincos kernel
def test_det_edge_sizes(n, t_steps):
    cfg, lin, ang, planner, params = build(dict(use_det_dynamics=True), n, t_steps, 1)
    u0 = planner.solve()
    assert u0.shape == (t_steps, 2) and np.isfinite(u0).all()
    planner.sample_noise()
    noise, u_in = planner.noise_samples_d.copy_to_host(), planner.u_cur_d.copy_to_host()
    assert noise.shape == (n, t_steps, 2)
    planner.rollout()
    got = planner.costs_d.copy_to_host()
    want = oracle_costs(dict(m=1), params, lin, ang, noise, u_in)
    assert np.array_equal(got, want), (planner.last_rollout_kernel(), ulp_diff_f32(got, want).max())
    planner.update()
    _, u_ref, _ = O.update_useq(params["lambda_weight"], want, noise, params["vrange"], params["wrange"], u_in)
    assert (np.abs(planner.u_cur_d.copy_to_host() - u_ref) / np.array([3.0, np.pi])).max() <= 1e-5
    assert abs(planner.weights_d.copy_to_host().sum() - 1.0) < 1e-5


@pytest.mark.parametrize("n,t_steps,m,alpha", [(1, 3, 1, 1.0), (3, 5, 2, 0.5), (5, 12, 63, 0.3), (4, 9, 65, 1.0),
                                               (2, 6, 1500, 0.2)])
def test_cvar_edge_sizes(n, t_steps, m, alpha):
    cfg, lin, ang, planner, params = build(dict(use_tdm=True), n, t_steps, m)
    params["cvar_alpha"] = alpha
    planner.set_params(params)
    u0 = planner.solve()
    assert u0.shape == (t_steps, 2) and np.isfinite(u0).all()
    assert lin.sample_grid_batch_d.shape[0] == m
    planner.sample_noise()
    noise, u_in = planner.noise_samples_d.copy_to_host(), planner.u_cur_d.copy_to_host()
    planner.rollout()
    got = planner.costs_d.copy_to_host()
    # (the CVaR kernel also for M = 1: its cost order differs from the deterministic kernel's)
    p = O.make_params(params, lin.res, lin.padded_xlimits, lin.padded_ylimits,
                      lin.bin_values_bounds_d.copy_to_host(), ang.bin_values_bounds_d.copy_to_host())
    want = O.rollout_tdm(p, lin.sample_grid_batch_d.copy_to_host(), ang.sample_grid_batch_d.copy_to_host(),
                         lin.obstacle_map_d.copy_to_host(), lin.unknown_map_d.copy_to_host(), noise, u_in)
    assert np.array_equal(got, want), (planner.last_rollout_kernel(), ulp_diff_f32(got, want).max())


@pytest.mark.parametrize("n,t_steps", [(1, 2), (65, 5)])
def test_speed_map_edge_sizes(n, t_steps):
    cfg, lin, ang, planner, params = build(dict(use_nom_dynamics_with_speed_map=True), n, t_steps, 1)
    planner.solve()
    planner.sample_noise()
    noise, u_in = planner.noise_samples_d.copy_to_host(), planner.u_cur_d.copy_to_host()
    planner.rollout()
    got = planner.costs_d.copy_to_host()
    p = O.make_params(params, lin.res, lin.padded_xlimits, lin.padded_ylimits,
                      lin.bin_values_bounds_d.copy_to_host(), ang.bin_values_bounds_d.copy_to_host())
    want = O.rollout_det(p, lin.sample_grid_batch_d.copy_to_host(), ang.sample_grid_batch_d.copy_to_host(),
                         lin.obstacle_map_d.copy_to_host(), lin.unknown_map_d.copy_to_host(), noise, u_in,
                         risk=lin.risk_traction_map_d.copy_to_host())
    assert np.array_equal(got, want)
    # (round 6: the latency regime of this mode is the time-parallel kernel; the fused kernel behind the developer switch)
    assert "k_rollout_scan_exact speed_map" in planner.last_rollout_kernel()
    from mppi_numba_amd import _lib
    planner.set_debug_flags(_lib.DEBUG_NO_SCAN_KERNEL)
    planner.rollout()
    assert np.array_equal(planner.costs_d.copy_to_host(), want)
    assert "k_rollout_fused speed_map" in planner.last_rollout_kernel()


def test_speed_map_large_reach_uses_global_cells_and_matches_oracle():
    """Reach window of 32-bit cells (traction bits + risk byte) larger than LDS: the general
    kernel on the global cell array; same costs."""
    from mppi_numba_amd.config import Config
    from mppi_numba_amd.mppi import MPPI_Numba
    from mppi_numba_amd.terrain import TDM_Numba
    pmf, obstacle, unknown, td = bench.synthetic_world("c3", np.random.default_rng(0))
    cfg = Config(T=10.0, dt=0.1, num_grid_samples=1, num_control_rollouts=2048, max_speed_padding=5.0,
                 num_vis_state_rollouts=1, max_map_dim=(260, 260), seed=2, enforce_recommended_limits=False,
                 use_nom_dynamics_with_speed_map=True)
    lin, ang = TDM_Numba(cfg), TDM_Numba(cfg)
    lin.set_TDM_from_PMF_grid(pmf, td, obstacle, unknown)
    ang.set_TDM_from_PMF_grid(pmf, td, obstacle, unknown)
    planner = MPPI_Numba(cfg)
    params = bench.make_params("c2")
    params.update(x0=np.array([32.0, 32.0, 0.5]), xgoal=np.array([50.0, 40.0]))
    planner.setup(params, lin, ang)
    from mppi_numba_amd import _lib
    planner.set_debug_flags(_lib.DEBUG_NO_SCAN_KERNEL)  # (the time-parallel kernel needs no window: tests/test_gpu_speedmap_scan.py)
    planner.solve()
    planner.sample_noise()
    noise, u_in = planner.noise_samples_d.copy_to_host(), planner.u_cur_d.copy_to_host()
    planner.rollout()
    assert "k_rollout_map speed_map global_cells" in planner.last_rollout_kernel()
    p = O.make_params(params, lin.res, lin.padded_xlimits, lin.padded_ylimits,
                      lin.bin_values_bounds_d.copy_to_host(), ang.bin_values_bounds_d.copy_to_host())
    want = O.rollout_det(p, lin.sample_grid_batch_d.copy_to_host(), ang.sample_grid_batch_d.copy_to_host(),
                         lin.obstacle_map_d.copy_to_host(), lin.unknown_map_d.copy_to_host(), noise, u_in,
                         risk=lin.risk_traction_map_d.copy_to_host())
    got = planner.costs_d.copy_to_host()
    ulps = ulp_diff_f32(got, want)
    assert (ulps == 0).mean() >= 0.999
    # the same problem from the map corner: the window fits, the fused kernel runs, same bar
    params.update(x0=np.array([4.0, 4.0, 0.5]))
    planner.set_params(params)
    planner.rollout()
    assert "k_rollout_fused speed_map" in planner.last_rollout_kernel()
    p = O.make_params(params, lin.res, lin.padded_xlimits, lin.padded_ylimits,
                      lin.bin_values_bounds_d.copy_to_host(), ang.bin_values_bounds_d.copy_to_host())
    want = O.rollout_det(p, lin.sample_grid_batch_d.copy_to_host(), ang.sample_grid_batch_d.copy_to_host(),
                         lin.obstacle_map_d.copy_to_host(), lin.unknown_map_d.copy_to_host(), noise, u_in,
                         risk=lin.risk_traction_map_d.copy_to_host())
    ulps = ulp_diff_f32(planner.costs_d.copy_to_host(), want)
    assert (ulps == 0).mean() >= 0.999
