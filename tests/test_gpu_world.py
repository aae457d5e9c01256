"""the simulated world on the device.

* DeviceWorld.get == TractionGrid.get (terrain.py:776-782), cell rule included;
* the device draw of sample_grids_true_dist (terrain.py:586-608) follows the terrains' sample pools;
* MPPI_Numba.closed_loop == the notebook's loop (test.ipynb cell 4) driven from the host with the
  same seeds: same controls, same trajectory, same number of steps."""
import numpy as np
import pytest

from test_gpu_notebook_flow import notebook_world

pytestmark = pytest.mark.gpu


def test_device_world_get_equals_traction_grid_get():
    from mppi_numba_amd.terrain import DeviceWorld, TractionGrid
    rng = np.random.default_rng(5)
    for res, xlim, ylim, shape in ((1.0, None, None, (9, 13)), (0.3, (-2.1, 1.8), (0.7, 3.4), (9, 13)),
                                   (0.25, (0.0, 64.0), (0.0, 32.0), (128, 256))):
        lin, ang = rng.random(shape), rng.random(shape)
        host = TractionGrid(lin, ang, res=res, xlimits=xlim, ylimits=ylim)
        dev = DeviceWorld.from_traction_grid(host)
        x = rng.uniform(host.xlimits[0] - 2 * res, host.xlimits[1] + 2 * res, 4000)
        y = rng.uniform(host.ylimits[0] - 2 * res, host.ylimits[1] + 2 * res, 4000)
        # cell borders, where floor(a / b) and a // b part ways
        kx = rng.integers(-2, shape[1] + 2, 500)
        x[:500] = host.xlimits[0] + kx * res
        y[:500] = host.ylimits[0] + rng.integers(-2, shape[0] + 2, 500) * res
        x[500:1000] = np.nextafter(x[:500], -np.inf)
        y[500:1000] = np.nextafter(y[:500], np.inf)
        got_l, got_a = dev.get(x, y)
        want = np.array([host.get(a, b) for a, b in zip(x, y)], dtype=np.float64)
        np.testing.assert_array_equal(got_l, want[:, 0])
        np.testing.assert_array_equal(got_a, want[:, 1])
        assert (want[:, 0] == 0).any() and (want[:, 0] != 0).any()
        l0, a0 = dev.get(x[3], y[3])
        assert (l0, a0) == tuple(want[3])
        gl, ga = dev.get_grids()
        np.testing.assert_array_equal(gl, lin)
        np.testing.assert_array_equal(ga, ang)


def test_true_distribution_draw_on_device_follows_the_pools():
    from scipy import stats
    from mppi_numba_amd.config import Config
    from mppi_numba_amd.terrain import TDM_Numba, TractionGrid
    W = notebook_world(seed=3)
    big = np.zeros((96, 128), dtype=np.int8)
    big[np.random.default_rng(1).random(big.shape) < 0.4] = 1
    cfg = Config(T=1.0, dt=0.1, num_grid_samples=8, num_control_rollouts=64, max_speed_padding=1.0,
                 max_map_dim=(140, 140), seed=7, use_tdm=True)
    tdm = TDM_Numba(cfg)
    bounds = (np.min(W["bin_values"]), np.max(W["bin_values"]))
    tdm.set_TDM_from_semantic_grid(big, 1.0, W["num_pmf_bins"], W["bin_values"], bounds, (0, 128.0), (0, 96.0),
                                   W["id2name"], W["name2terrain"], W["lin_pmf"], det_dynamics_cvar_alpha=1.0)
    grid = tdm.sample_grids_true_dist_on_device(seed=11)
    assert isinstance(grid, TractionGrid) and grid.lin_traction.shape == big.shape
    again = tdm.sample_grids_true_dist_on_device(seed=11)
    np.testing.assert_array_equal(grid.lin_traction, again.lin_traction)  # same seed, fresh world: same draw
    other = tdm.sample_grids_true_dist_on_device(seed=12)
    assert (other.lin_traction != grid.lin_traction).mean() > 0.9
    for sid, terrain in ((0, W["dirt"]), (1, W["bush"])):
        mask = big == sid
        for drawn, pool in ((grid.lin_traction[mask], terrain.lin_saved_samples),
                            (grid.ang_traction[mask], terrain.ang_saved_samples)):
            pool = np.asarray(pool, dtype=np.float64)
            assert np.isin(drawn, pool).all()
            assert stats.ks_2samp(drawn, pool).pvalue > 1e-3
            fresh = np.asarray(terrain.lin_density.sample(20000), dtype=np.float64)
            assert abs(drawn.mean() - fresh.mean()) < 0.02
        # linear and angular draws of a cell are independent picks
        assert abs(np.corrcoef(grid.lin_traction[mask], grid.ang_traction[mask])[0, 1]) < 0.05
    # the host sampler of the reference signature is still there
    host = tdm.sample_grids_true_dist()
    assert host.lin_traction.shape == big.shape


def _planner(mode, seed, num_instances=1):
    from mppi_numba_amd.batch import MPPI_Batch
    from mppi_numba_amd.config import Config
    from mppi_numba_amd.mppi import MPPI_Numba
    from mppi_numba_amd.terrain import TDM_Numba
    W = notebook_world(seed=seed)
    grid, res = W["semantic_grid"], W["res"]
    xlimits, ylimits = (0, grid.shape[1] * res), (0, grid.shape[0] * res)
    bounds = (np.min(W["bin_values"]), np.max(W["bin_values"]))
    use_tdm = mode == "use_tdm"
    cfg = Config(T=5.0, dt=0.1, num_grid_samples=128 if use_tdm else 1, num_control_rollouts=1024,
                 max_speed_padding=5.0, num_vis_state_rollouts=1, max_map_dim=(15, 15), seed=1, **{mode: True})
    x0 = np.array([0.5, 0.5, np.pi / 4])
    xgoal = np.array([8.5, 8.5])
    params = dict(dt=cfg.dt, x0=x0, xgoal=xgoal, goal_tolerance=0.5, v_post_rollout=0.01,
                  cvar_alpha=0.2 if use_tdm else 1.0, alpha_dyn=1.0 if use_tdm else 0.2, dist_weight=1,
                  lambda_weight=1.0, num_opt=1, u_std=np.array([2.0, 3.0]),
                  vrange=np.array([0.0, 3.0]), wrange=np.array([-np.pi, np.pi]))
    lin, ang = TDM_Numba(cfg), TDM_Numba(cfg)
    for tdm, table in ((lin, W["lin_pmf"]), (ang, W["ang_pmf"])):
        tdm.set_TDM_from_semantic_grid(grid, res, W["num_pmf_bins"], W["bin_values"], bounds, xlimits, ylimits,
                                       W["id2name"], W["name2terrain"], table,
                                       det_dynamics_cvar_alpha=params['alpha_dyn'])
    planner = MPPI_Numba(cfg) if num_instances == 1 else MPPI_Batch(cfg, num_instances)
    return W, cfg, params, lin, ang, planner


def _host_loop(planner, cfg, traction_grid, x0, xgoal, tol, max_steps):
    """test.ipynb cell 4, verbatim arithmetic."""
    xhist = np.full((max_steps + 1, 3), np.nan)
    uhist = np.full((max_steps, 2), np.nan, dtype=np.float32)
    xhist[0] = x0
    steps = max_steps
    for t in range(max_steps):
        useq = planner.solve()
        u_curr = uhist[t] = useq[0]
        lt, at = traction_grid.get(xhist[t, 0], xhist[t, 1])
        xhist[t + 1, 0] = xhist[t, 0] + cfg.dt * lt * np.cos(xhist[t, 2]) * u_curr[0]
        xhist[t + 1, 1] = xhist[t, 1] + cfg.dt * lt * np.sin(xhist[t, 2]) * u_curr[0]
        xhist[t + 1, 2] = xhist[t, 2] + cfg.dt * at * u_curr[1]
        planner.shift_and_update(xhist[t + 1], useq, num_shifts=1)
        if np.linalg.norm(xhist[t + 1, :2] - xgoal) <= tol:
            steps = t + 1
            break
    return xhist, uhist, steps


@pytest.mark.parametrize("mode", ["use_det_dynamics", "use_tdm", "use_nom_dynamics_with_speed_map"])
def test_closed_loop_on_device_equals_the_notebook_loop(mode):
    from mppi_numba_amd.terrain import TractionGrid
    max_steps = 60
    W, cfg, params, lin, ang, host_planner = _planner(mode, seed=11)
    np.random.seed(4)
    world = lin.sample_grids_true_dist()
    host_planner.setup(params, lin, ang)
    want_x, want_u, want_steps = _host_loop(host_planner, cfg, world, params["x0"], params["xgoal"],
                                            params["goal_tolerance"], max_steps)
    _, cfg2, params2, lin2, ang2, dev_planner = _planner(mode, seed=11)
    dev_planner.setup(params2, lin2, ang2)
    twin = TractionGrid(world.lin_traction.copy(), world.ang_traction.copy())
    got_x, got_u, got_steps = dev_planner.closed_loop(twin, max_steps)
    assert got_steps == want_steps
    ran = want_steps
    np.testing.assert_array_equal(np.isnan(got_x), np.isnan(want_x))
    np.testing.assert_allclose(got_u[:ran], want_u[:ran], rtol=0, atol=2e-6)
    np.testing.assert_allclose(got_x[:ran + 1], want_x[:ran + 1], rtol=0, atol=1e-5)
    assert np.linalg.norm(got_x[ran, :2] - params["xgoal"]) < np.linalg.norm(params["x0"][:2] - params["xgoal"])
    # the handle is where the notebook's loop leaves it: same next solution from the same state
    np.testing.assert_allclose(dev_planner.params["x0"], got_x[ran])
    a, b = dev_planner.solve(), host_planner.solve()
    np.testing.assert_allclose(a, b, rtol=0, atol=2e-5)


def test_closed_loop_batched_problems_each_follow_their_own_loop():
    from mppi_numba_amd.terrain import DeviceWorld, TractionGrid
    B, max_steps = 3, 40
    W, cfg, params, lin, ang, batch = _planner("use_det_dynamics", seed=5, num_instances=B)
    np.random.seed(9)
    world = lin.sample_grids_true_dist()
    x0s = np.array([[0.5, 0.5, np.pi / 4], [8.5, 0.5, np.pi / 2], [0.5, 8.4, 0.0]])
    goals = np.array([[8.5, 8.5], [0.5, 8.5], [8.5, 0.6]])
    batch.setup(params, lin, ang, x0s, goals)
    dev = DeviceWorld.from_traction_grid(world)
    xh, uh, steps = batch.closed_loop(dev, max_steps, x_init=x0s)
    assert xh.shape == (B, max_steps + 1, 3) and uh.shape == (B, max_steps, 2) and steps.shape == (B,)
    for b in range(B):
        n = int(steps[b])
        assert np.isfinite(xh[b, :n + 1]).all() and np.isnan(xh[b, n + 1:]).all()
        np.testing.assert_array_equal(xh[b, 0], x0s[b])
        # every logged transition is the world's Euler step of the logged control
        for t in range(n):
            lt, at = world.get(xh[b, t, 0], xh[b, t, 1])
            step = np.array([cfg.dt * lt * np.cos(xh[b, t, 2]) * uh[b, t, 0],
                             cfg.dt * lt * np.sin(xh[b, t, 2]) * uh[b, t, 0], cfg.dt * at * uh[b, t, 1]])
            np.testing.assert_allclose(xh[b, t + 1], xh[b, t] + step, rtol=0, atol=1e-12)
        d0, d1 = np.linalg.norm(x0s[b, :2] - goals[b]), np.linalg.norm(xh[b, n, :2] - goals[b])
        assert d1 < 0.7 * d0, (b, d0, d1)
        if n < max_steps:
            assert d1 <= params["goal_tolerance"]
    np.testing.assert_allclose(batch.x0s, np.stack([xh[b, steps[b]] for b in range(B)]).astype(np.float32))


def test_device_world_of_an_int8_traction_grid():
    """TractionGrid(use_int8=True) keeps 0..100 integers (terrain.py:757-759); its device twin answers
    with the same numbers."""
    from mppi_numba_amd.terrain import DeviceWorld, TractionGrid
    rng = np.random.default_rng(8)
    host = TractionGrid(rng.random((20, 30)), rng.random((20, 30)), res=0.5, use_int8=True)
    assert host.lin_traction.dtype == np.int8
    dev = DeviceWorld.from_traction_grid(host)
    x, y = rng.uniform(-1, 16, 500), rng.uniform(-1, 11, 500)
    got_l, got_a = dev.get(x, y)
    want = np.array([host.get(a, b) for a, b in zip(x, y)], dtype=np.float64)
    np.testing.assert_array_equal(got_l, want[:, 0])
    np.testing.assert_array_equal(got_a, want[:, 1])
