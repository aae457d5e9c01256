"""The reference's acceptance demo as a test: the closed loop of /root/reference/test.ipynb
(cells 2-4: terrain models -> semantic grid -> TDMs -> planner -> solve / simulate /
shift_and_update until the goal, with the notebook's plots drawn headless), written against
the `mppi_numba` ALIAS exactly as the notebook imports it.  The GPU box has no reference
checkout, so the host helpers are the test suite's stand-ins (tests/standins/, found through MPPI_NUMBA_REFERENCE) here; with a checkout beside it
`tools/run_reference_notebook.py` runs the notebook's own cells (tests/test_alias_notebooks.py
does that for the cells that need no GPU)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def notebook_world(seed):
    """test.ipynb cell 2 + the map part of cell 3 (9 x 9 cells of 1 m, vegetation patches)."""
    from mppi_numba.density import GaussianMixture
    from mppi_numba.terrain import Terrain
    np.random.seed(seed)
    pmf_bounds = [0, 1.0]
    mk = lambda w, m, s: GaussianMixture(sample_bounds=pmf_bounds, pmf_bounds=pmf_bounds, weights=w, means=m, stds=s)
    bush = Terrain(name='Vegetation', lin_density=mk([0.6, 0.4], [0, 0.8], [0.15, 0.1]),
                   ang_density=mk([0.6, 0.4], [0, 0.8], [0.15, 0.1]), rgb=np.array((0, 250, 0)) / 255.0)
    dirt = Terrain(name='Dirt', lin_density=mk([1], [0.65], [0.1]), ang_density=mk([1], [0.65], [0.1]),
                   rgb=np.array((200, 190, 160)) / 255.0)
    num_bins = 20
    lin_pmf = {t: t.lin_density.get_pmf(num_bins=num_bins) for t in (bush, dirt)}
    ang_pmf = {t: t.ang_density.get_pmf(num_bins=num_bins) for t in (bush, dirt)}
    grid_shape, margin = (9, 9), 1
    semantic_grid = np.zeros(grid_shape, dtype=np.int8)
    rand = np.random.rand(grid_shape[0] - 2 * margin, grid_shape[1] - 2 * margin)
    semantic_grid[margin:-margin, margin:-margin][rand < 0.4] = 1
    return dict(bush=bush, dirt=dirt, id2name={0: dirt.name, 1: bush.name},
                name2terrain={bush.name: bush, dirt.name: dirt}, lin_pmf=lin_pmf, ang_pmf=ang_pmf,
                semantic_grid=semantic_grid, res=1.0, margin=margin, bin_values=lin_pmf[bush][0],
                num_pmf_bins=len(lin_pmf[dirt][1]))


@pytest.mark.parametrize("mode", ["use_tdm", "use_det_dynamics", "use_nom_dynamics_with_speed_map"])
def test_notebook_closed_loop(mode):
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    from mppi_numba.config import Config
    from mppi_numba.mppi import MPPI_Numba
    from mppi_numba.terrain import TDM_Numba, TractionGrid
    from mppi_numba.visualization import TDM_Visualizer, vis_density, vis_density_as_pmf

    W = notebook_world(seed=11)
    fig, axes = plt.subplots(1, 2)
    vis_density(axes[0], W["bush"].lin_density, W["bush"], color=W["bush"].rgb)
    vis_density_as_pmf(axes[1], W["dirt"].lin_density, W["dirt"], num_bins=20, color=W["dirt"].rgb)
    plt.close(fig)

    grid, res = W["semantic_grid"], W["res"]
    xlimits, ylimits = (0, grid.shape[1] * res), (0, grid.shape[0] * res)
    bounds = (np.min(W["bin_values"]), np.max(W["bin_values"]))
    max_speed = 3.0
    use_tdm = mode == "use_tdm"
    cfg = Config(T=10.0, dt=0.1, num_grid_samples=1024, num_control_rollouts=1024,
                 max_speed_padding=max_speed + 2.0, num_vis_state_rollouts=100 if use_tdm else 1,
                 max_map_dim=(15, 15), seed=1, **{mode: True})
    x0 = np.array([W["margin"] / 2, W["margin"] / 2, np.pi / 4])
    xgoal = np.array([grid.shape[0] - W["margin"] / 2, grid.shape[0] - W["margin"] / 2])
    mppi_params = dict(dt=cfg.dt, x0=x0, xgoal=xgoal, goal_tolerance=0.5, v_post_rollout=0.01,
                       cvar_alpha=0.2 if use_tdm else 1.0, alpha_dyn=1.0 if use_tdm else 0.2, dist_weight=1,
                       lambda_weight=1.0, num_opt=1, u_std=np.array([2.0, 3.0]),
                       vrange=np.array([0.0, max_speed]), wrange=np.array([-np.pi, np.pi]))
    mppi_planner, lin_tdm, ang_tdm = MPPI_Numba(cfg), TDM_Numba(cfg), TDM_Numba(cfg)
    for tdm, table in ((lin_tdm, W["lin_pmf"]), (ang_tdm, W["ang_pmf"])):
        tdm.reset()
        tdm.set_TDM_from_semantic_grid(grid, res, W["num_pmf_bins"], W["bin_values"], bounds, xlimits, ylimits,
                                       W["id2name"], W["name2terrain"], table,
                                       det_dynamics_cvar_alpha=mppi_params['alpha_dyn'],
                                       obstacle_map=None, unknown_map=None)
    mppi_planner.reset()
    mppi_planner.setup(mppi_params, lin_tdm, ang_tdm)

    traction_grid = lin_tdm.sample_grids_true_dist()
    assert isinstance(traction_grid, TractionGrid) and traction_grid.lin_traction.shape == grid.shape
    fig, ax = TDM_Visualizer(lin_tdm).draw(figsize=(5, 5))
    assert len(ax.collections) == 2
    plt.close(fig)

    max_steps = 151
    xhist = np.full((max_steps + 1, 3), np.nan)
    uhist = np.full((max_steps, 2), np.nan)
    xhist[0] = x0
    reached = None
    for t in range(max_steps):
        useq = mppi_planner.solve()
        assert useq.shape == (cfg.num_steps, 2) and np.isfinite(useq).all()
        u_curr = uhist[t] = useq[0]
        lt, at = traction_grid.get(xhist[t, 0], xhist[t, 1])
        xhist[t + 1, 0] = xhist[t, 0] + cfg.dt * lt * np.cos(xhist[t, 2]) * u_curr[0]
        xhist[t + 1, 1] = xhist[t, 1] + cfg.dt * lt * np.sin(xhist[t, 2]) * u_curr[0]
        xhist[t + 1, 2] = xhist[t, 2] + cfg.dt * at * u_curr[1]
        if t % 30 == 0:
            fig, ax = TDM_Visualizer(lin_tdm).draw(figsize=(5, 5))
            states = mppi_planner.get_state_rollout()
            assert states.shape == (cfg.num_vis_state_rollouts, cfg.num_steps + 1, 3)
            np.testing.assert_allclose(states[:, 0, :], np.broadcast_to(xhist[t].astype(np.float32), states[:, 0, :].shape))
            ax.plot(states[:, :, 0].T, states[:, :, 1].T, 'b')
            plt.close(fig)
        mppi_planner.shift_and_update(xhist[t + 1], useq, num_shifts=1)
        if np.linalg.norm(xhist[t + 1, :2] - xgoal) <= mppi_params['goal_tolerance']:
            reached = t
            break
    # the controls respect their ranges and the vehicle makes progress towards the goal
    done = ~np.isnan(uhist[:, 0])
    assert (uhist[done, 0] >= 0).all() and (uhist[done, 0] <= max_speed).all()
    assert (np.abs(uhist[done, 1]) <= np.pi + 1e-6).all()
    last = xhist[np.flatnonzero(~np.isnan(xhist[:, 0]))[-1], :2]
    start_dist = np.linalg.norm(x0[:2] - xgoal)
    assert reached is not None or np.linalg.norm(last - xgoal) < 0.5 * start_dist, \
        "no progress: %.2f of %.2f m left after %d steps" % (np.linalg.norm(last - xgoal), start_dist, max_steps)
