"""GPU parity tests: the HIP path, called through the C ABI (via the Python
mirror of the reference's classes), against
  (a) the golden fixtures recorded from the reference's own code, and
  (b) the C restatement (oracle/) on larger seeded inputs.

Tolerances.  With MPPI_MATH_EXACT the kernels keep the rounding points of the
reference's CPU path, so costs are expected BIT-IDENTICAL; the tests allow a
stray last-bit difference in at most 0.1% of rollouts (float64 libm vs device
polynomial differences need a ~1e-9 coincidence per step to show).  The control
update is a deterministic float64 tree here and unordered float32 atomics in the
reference: u is compared to 1e-5 of the control range (north_star's bound),
weights to 1e-5 relative.
"""
import numpy as np
import pytest

from helpers import golden, iterations, params_from_golden, ulp_diff_f32
from gpu_helpers import build_from_golden, solve_of_iteration

pytestmark = pytest.mark.gpu

MAP_FIXTURES = ["det_cvar", "det_mean", "speedmap_cvar", "speedmap_mean", "speedmap_mean_bounds", "tdm_cvar",
                "tdm_mean_alpha_dyn", "tdm_cvar_odd", "tdm_oversized_mean",
                # res = 0.3, dt = 0.05, reverse driving, |theta0| of several turns, bounds (0, 1.2)
                "det_odd_units", "speedmap_odd_units", "tdm_odd_units",
                # the same set-ups on other random worlds (GOLDEN_SEED_OFFSET=101 / 202)
                "det_odd_units_w101", "speedmap_odd_units_w101", "tdm_odd_units_w101",
                "det_odd_units_w202", "speedmap_odd_units_w202", "tdm_odd_units_w202"]


def control_scale(P):
    return np.array([max(abs(P["vrange"][0]), abs(P["vrange"][1])),
                     max(abs(P["wrange"][0]), abs(P["wrange"][1]))], dtype=np.float64)


def assert_costs_match(got, want, what):
    ulps = ulp_diff_f32(got, want)
    frac_exact = float((ulps == 0).mean())
    assert ulps.max() <= 2 and frac_exact >= 0.999 or len(want) < 1000 and ulps.max() == 0, \
        "%s: max ulp %d, exact fraction %.4f" % (what, ulps.max(), frac_exact)


@pytest.mark.parametrize("name", MAP_FIXTURES)
def test_host_preprocessing_matches_reference(name):
    g = golden(name)
    _, lin, ang, _, _ = build_from_golden(name, g)
    for tag, tdm in (("lin", lin), ("ang", ang)):
        assert (tdm.pmf_grid == g[tag + "_pmf_grid_unpadded"]).all()
        assert (tdm.pmf_grid_d.copy_to_host() == g[tag + "_pmf_grid_padded"]).all()
        assert (tdm.obstacle_map_d.copy_to_host() == g[tag + "_obstacle_map_padded"]).all()
        assert (tdm.unknown_map_d.copy_to_host() == g[tag + "_unknown_map_padded"]).all()
        assert np.array_equal(np.asarray(tdm.padded_xlimits, float), g[tag + "_padded_xlimits"])
        assert np.array_equal(np.asarray(tdm.padded_ylimits, float), g[tag + "_padded_ylimits"])
        assert tdm.pad_cells == int(g[tag + "_pad_cells"])
        assert tdm.bin_values_bounds_d.copy_to_host().dtype == g[tag + "_bin_values_bounds"].dtype
        if (tag + "_risk_traction_map_padded") in g:
            assert (tdm.risk_traction_map_d.copy_to_host() == g[tag + "_risk_traction_map_padded"]).all()


@pytest.mark.parametrize("name", MAP_FIXTURES)
def test_rollout_costs_vs_golden(name):
    """Inject the reference's sampled grids, noise and u; compare costs."""
    g = golden(name)
    _, lin, ang, planner, P = build_from_golden(name, g)
    for k, it in enumerate(iterations(g)):
        s = solve_of_iteration(g, k)
        lin.set_sampled_grids(g["solve%d_lin_sample_grid" % s])
        ang.set_sampled_grids(g["solve%d_ang_sample_grid" % s])
        planner.params["x0"] = g["solve%d_x0" % s]
        planner.set_noise(it["noise"])
        planner.set_u(it["u_in"])
        planner.rollout()
        assert_costs_match(planner.costs_d.copy_to_host(), it["costs"], "%s it%d" % (name, k))
        # noise survives the layout round trip
        assert (planner.noise_samples_d.copy_to_host() == it["noise"]).all()


@pytest.mark.parametrize("name", MAP_FIXTURES)
def test_update_vs_golden(name):
    g = golden(name)
    _, lin, ang, planner, P = build_from_golden(name, g)
    scale = control_scale(P)
    for k, it in enumerate(iterations(g)):
        planner.set_noise(it["noise"])
        planner.set_u(it["u_in"])
        planner.set_costs(it["costs"])
        planner.update()
        w = planner.weights_d.copy_to_host()
        u = planner.u_cur_d.copy_to_host()
        want_w = g["it%d_serial_weights" % k].astype(np.float64)
        assert np.abs(w - want_w).max() <= 1e-5 * want_w.max() + 1e-30
        big = want_w > 1e-6
        assert (np.abs(w[big] - want_w[big]) / want_w[big]).max() <= 1e-5
        for want_u in (it["u_out"], g["it%d_serial_u_out" % k]):
            assert (np.abs(u.astype(np.float64) - want_u) / scale).max() <= 1e-5
        assert (planner.u_prev_d.copy_to_host() == u).all()


@pytest.mark.parametrize("name", ["det_cvar", "speedmap_cvar", "tdm_cvar", "tdm_mean_alpha_dyn",
                                  "tdm_cvar_odd", "tdm_oversized_mean", "det_odd_units", "speedmap_odd_units",
                                  "tdm_odd_units", "det_odd_units_w101", "tdm_odd_units_w202"])
def test_end_to_end_from_seed_xoroshiro(name):
    """Level L3: with the numba-compatible generator the whole closed loop
    (sample grids -> noise -> rollout -> update -> shift) reproduces the
    reference from the seed."""
    g = golden(name)
    _, lin, ang, planner, P = build_from_golden(name, g, rng="xoroshiro")
    scale = control_scale(P)
    its = iterations(g)
    s = 0
    while ("solve%d_useq" % s) in g:
        planner.params["x0"] = g["solve%d_x0" % s]
        useq = planner.solve()
        for tag, tdm in (("lin", lin), ("ang", ang)):
            want = g["solve%d_%s_sample_grid" % (s, tag)]
            got = tdm.sample_grid_batch_d.copy_to_host()[:, :want.shape[1], :want.shape[2]]
            assert (got == want).all(), "sampled %s grids differ in solve %d" % (tag, s)
        last = (int(g["solve%d_first_iteration" % (s + 1)]) if ("solve%d_first_iteration" % (s + 1)) in g
                else len(its)) - 1
        assert ulp_diff_f32(planner.noise_samples_d.copy_to_host(), its[last]["noise"]).max() == 0
        # the chain runs through the update, whose summation order differs -> tolerance on u
        assert (np.abs(useq.astype(np.float64) - g["solve%d_useq" % s]) / scale).max() <= 1e-5
        if s == 0:
            want_sr = g["state_rollout_after_solve0"]
            got_sr = planner.get_state_rollout()
            assert np.abs(got_sr - want_sr).max() <= 1e-4
        # closed loop exactly as the generator did it
        x = np.asarray(g["solve%d_x0" % s], dtype=float)
        u0 = g["solve%d_useq" % s][0]
        x = x + float(g["cfg_dt"]) * np.array([u0[0] * np.cos(x[2]), u0[0] * np.sin(x[2]), u0[1]])
        planner.shift_and_update(x, g["solve%d_useq" % s].copy(), num_shifts=1)
        s += 1


@pytest.mark.parametrize("name", ["det_cvar", "speedmap_cvar", "tdm_cvar", "tdm_mean_alpha_dyn", "det_odd_units",
                                  "speedmap_odd_units_w101", "tdm_odd_units_w202"])
def test_state_rollout_vs_golden(name):
    g = golden(name)
    _, lin, ang, planner, P = build_from_golden(name, g)
    its = iterations(g)
    last = int(g["solve1_first_iteration"]) - 1 if "solve1_first_iteration" in g else len(its) - 1
    lin.set_sampled_grids(g["solve0_lin_sample_grid"])
    ang.set_sampled_grids(g["solve0_ang_sample_grid"])
    planner.params["x0"] = g["solve0_x0"]
    planner.set_noise(its[last]["noise"])
    # reproduce "u_prev aliases u_cur after the update": run the update from its inputs
    planner.set_u(its[last]["u_in"])
    planner.set_costs(its[last]["costs"])
    planner.update()
    planner.set_u(its[last]["u_out"])
    got = planner.get_state_rollout()
    want = g["state_rollout_after_solve0"]
    # row 0 (and all rows in use_tdm) depend only on u_cur == the fixture's u_out: bit-exact
    if name.startswith("tdm"):
        assert ulp_diff_f32(got, want).max() == 0
    else:
        assert ulp_diff_f32(got[0], want[0]).max() == 0
        assert np.abs(got - want).max() <= 1e-4  # rows b>0 use u_prev (update order tolerance)


@pytest.mark.parametrize("name", ["barebone_flat", "barebone_obstacles"])
def test_barebone_vs_golden(name):
    from mppi_numba_amd.barebone import Config, MPPI_Numba
    g = golden(name)
    P = params_from_golden(g)
    cfg = Config(T=float(g["cfg_T"]), dt=float(g["cfg_dt"]),
                 num_control_rollouts=int(g["cfg_num_control_rollouts"]),
                 num_vis_state_rollouts=int(g["cfg_num_vis_state_rollouts"]), seed=int(g["cfg_seed"]),
                 enforce_recommended_limits=False)
    planner = MPPI_Numba(cfg)
    planner.setup(P)
    scale = control_scale(P)
    for k, it in enumerate(iterations(g)):
        planner.params["x0"] = g["solve%d_x0" % k]
        planner.set_noise(it["noise"])
        planner.set_u(it["u_in"])
        planner.rollout()
        got = planner.costs_d.copy_to_host()
        assert ulp_diff_f32(got, it["costs"]).max() == 0
        planner.update()
        u = planner.u_cur_d.copy_to_host()
        assert (np.abs(u.astype(np.float64) - it["u_out"]) / scale).max() <= 1e-5
        if k == 0:
            planner.set_u(it["u_out"])
            sr = planner.get_state_rollout()
            assert ulp_diff_f32(sr[0], g["state_rollout_after_solve0"][0]).max() == 0
            assert np.abs(sr - g["state_rollout_after_solve0"]).max() <= 1e-4


def test_barebone_end_to_end_xoroshiro():
    """BASELINE.json configs[0]: barebone unicycle N=64, T=30, from the seed."""
    from mppi_numba_amd.barebone import Config, MPPI_Numba
    g = golden("barebone_flat")
    P = params_from_golden(g)
    cfg = Config(T=float(g["cfg_T"]), dt=float(g["cfg_dt"]), num_control_rollouts=64,
                 num_vis_state_rollouts=6, seed=1, enforce_recommended_limits=False, rng="xoroshiro")
    planner = MPPI_Numba(cfg)
    planner.setup(P)
    scale = control_scale(P)
    x = P["x0"].astype(float).copy()
    for s in range(2):
        useq = planner.solve()
        assert ulp_diff_f32(planner.noise_samples_d.copy_to_host(), g["it%d_noise" % s]).max() == 0
        assert (np.abs(useq.astype(np.float64) - g["solve%d_useq" % s]) / scale).max() <= 1e-5
        u0 = g["solve%d_useq" % s][0]
        x = x + cfg.dt * np.array([u0[0] * np.cos(x[2]), u0[0] * np.sin(x[2]), u0[1]])
        planner.shift_and_update(x, g["solve%d_useq" % s].copy(), num_shifts=1)


@pytest.mark.parametrize("name", ["semantic_tdm", "semantic_det", "semantic_det_mean", "semantic_speedmap"])
def test_semantic_grid_end_to_end_xoroshiro(name):
    """set_TDM_from_semantic_grid -> solve(), from the seed, against the reference."""
    from test_host_and_abi import semantic_inputs
    from gpu_helpers import config_from_golden
    from mppi_numba_amd.mppi import MPPI_Numba
    from mppi_numba_amd.terrain import TDM_Numba
    g = golden(name)
    values, id2name, name2terrain, terrain2pmf, alpha = semantic_inputs(g)
    mode_name = {"semantic_tdm": "tdm", "semantic_det": "det", "semantic_det_mean": "det",
                 "semantic_speedmap": "speedmap"}[name]
    cfg = config_from_golden(mode_name, g, rng="xoroshiro")
    rows, cols = g["in_semantic_grid"].shape
    res = float(g["in_res"])
    lin, ang = TDM_Numba(cfg), TDM_Numba(cfg)
    for tdm in (lin, ang):
        tdm.set_TDM_from_semantic_grid(g["in_semantic_grid"], res, len(values), values, np.array([0.0, 1.0]),
                                       (0.0, cols * res), (0.0, rows * res), id2name, name2terrain, terrain2pmf,
                                       det_dynamics_cvar_alpha=alpha, obstacle_map=g["in_obstacle_map"],
                                       unknown_map=g["in_unknown_map"])
    assert (lin.pmf_grid_d.copy_to_host() == g["lin_pmf_grid_padded"]).all()
    assert (np.asarray(lin.semantic_grid) == g["lin_semantic_grid_after"]).all()
    planner = MPPI_Numba(cfg)
    P = params_from_golden(g)
    planner.setup(P, lin, ang)
    useq = planner.solve()
    want = g["solve0_lin_sample_grid"]
    got = lin.sample_grid_batch_d.copy_to_host()[:, :want.shape[1], :want.shape[2]]
    assert (got == want).all()
    it = iterations(g)[0]
    assert ulp_diff_f32(planner.noise_samples_d.copy_to_host(), it["noise"]).max() == 0
    assert ulp_diff_f32(planner.costs_d.copy_to_host(), it["costs"]).max() == 0
    assert (np.abs(useq.astype(np.float64) - g["solve0_useq"]) / control_scale(P)).max() <= 1e-5


def test_closed_loop_reaches_goal_like_test_ipynb():
    """The acceptance demo of the reference (test.ipynb:381-433): 9x9 semantic world,
    closed loop solve -> simulate on a sampled ground-truth traction grid ->
    shift_and_update, until the goal is reached.  Default generator (Philox)."""
    from mppi_numba_amd.config import Config
    from mppi_numba_amd.mppi import MPPI_Numba
    from mppi_numba_amd.terrain import TDM_Numba, TractionGrid
    rng = np.random.default_rng(7)
    rows = cols = 9
    res = 1.0
    bins = 12
    sg = (rng.random((rows, cols)) < 0.4).astype(int)  # 0 dirt, 1 vegetation
    sg[0, 0] = sg[-1, -1] = 0
    values = np.concatenate([[0.0], (np.arange(bins - 2) + 0.5) / (bins - 2), [1.0]])

    def pmf_of(center, width):
        p = np.exp(-0.5 * ((values - center) / width) ** 2)
        p[0] = p[-1] = 0.0
        return p / p.sum()

    id2name = {0: "dirt", 1: "veg"}
    name2terrain = {"dirt": "T_DIRT", "veg": "T_VEG"}
    terrain2pmf = {"T_DIRT": (values, pmf_of(0.8, 0.08)), "T_VEG": (values, pmf_of(0.45, 0.2))}
    cfg = Config(T=8.0, dt=0.1, num_grid_samples=256, num_control_rollouts=1024, max_speed_padding=3.0,
                 num_vis_state_rollouts=8, max_map_dim=(15, 15), seed=1, use_tdm=True)
    lin, ang = TDM_Numba(cfg), TDM_Numba(cfg)
    for tdm in (lin, ang):
        tdm.set_TDM_from_semantic_grid(sg, res, bins, values, np.array([0.0, 1.0]), (0.0, cols * res),
                                       (0.0, rows * res), id2name, name2terrain, terrain2pmf)
    x0 = np.array([0.5, 0.5, np.pi / 4])
    xgoal = np.array([8.5, 8.5])
    params = dict(x0=x0, xgoal=xgoal, dt=cfg.dt, goal_tolerance=0.5, v_post_rollout=0.01, lambda_weight=1.0,
                  cvar_alpha=0.8, alpha_dyn=1.0, num_opt=1, u_std=np.array([1.0, 1.0]),
                  vrange=np.array([0.0, 3.0]), wrange=np.array([-np.pi, np.pi]))
    planner = MPPI_Numba(cfg)
    planner.setup(params, lin, ang)
    # ground truth: each cell at its terrain's mean traction
    mean_tr = {0: float((values * terrain2pmf["T_DIRT"][1]).sum()), 1: float((values * terrain2pmf["T_VEG"][1]).sum())}
    truth = np.vectorize(mean_tr.get)(sg).astype(float)
    world = TractionGrid(truth, truth, res=res)
    x = x0.copy()
    reached_at = None
    for step in range(250):
        useq = planner.solve()
        assert useq is not None and np.isfinite(useq).all()
        l, a = world.get(x[0], x[1])
        x = x + cfg.dt * np.array([l * useq[0, 0] * np.cos(x[2]), l * useq[0, 0] * np.sin(x[2]), a * useq[0, 1]])
        planner.shift_and_update(x, useq, num_shifts=1)
        if np.linalg.norm(x[:2] - xgoal) <= params["goal_tolerance"]:
            reached_at = step
            break
    assert reached_at is not None, "robot at %s after 250 steps" % x
    sr = planner.get_state_rollout()
    assert sr.shape == (cfg.num_vis_state_rollouts, cfg.num_steps + 1, 3) and np.isfinite(sr).all()


def test_sampling_straight_into_cell_words_equals_sample_then_pack():
    """solve() of a CVaR planner with many samples draws both TDMs directly into the planner's
    cell words; the int8 grids follow on demand from the same Philox counters.  Same costs as the
    ordinary path (grids sampled, then transposed), same grids."""
    import bench
    from mppi_numba_amd.config import Config
    from mppi_numba_amd.mppi import MPPI_Numba
    from mppi_numba_amd.terrain import TDM_Numba
    m, n, t_steps = 256, 96, 20
    pmf, obstacle, unknown, td = bench.synthetic_world("c3", np.random.default_rng(3))
    pmf, obstacle, unknown = pmf[:, :70, :90].copy(), obstacle[:70, :90].copy(), unknown[:70, :90].copy()
    td = dict(td, xlimits=(0.0, 90 * 0.25), ylimits=(0.0, 70 * 0.25))
    handles = []
    for _ in range(2):
        cfg = Config(T=t_steps * 0.1, dt=0.1, num_grid_samples=m, num_control_rollouts=n, max_speed_padding=4.0,
                     num_vis_state_rollouts=1, max_map_dim=(80, 100), seed=9, enforce_recommended_limits=False,
                     use_tdm=True)
        lin, ang = TDM_Numba(cfg), TDM_Numba(cfg)
        lin.set_TDM_from_PMF_grid(pmf, td, obstacle, unknown)
        ang.set_TDM_from_PMF_grid(pmf[::-1].copy(), td, obstacle, unknown)
        planner = MPPI_Numba(cfg)
        params = bench.make_params("c3")
        params.update(x0=np.array([5.0, 6.0, 0.4]), xgoal=np.array([9.0, 9.0]), alpha_dyn=0.8)
        planner.setup(params, lin, ang)
        handles.append((lin, ang, planner, params))
    (lin_a, ang_a, fused, _), (lin_b, ang_b, staged, params) = handles
    for _ in range(2):  # two solves: the epochs advance alike
        u_fused = fused.solve()
        # the ordinary path by hand: sample both TDMs, then one staged iteration
        lin_b.sample_grids(params["alpha_dyn"])
        ang_b.sample_grids(params["alpha_dyn"])
        staged.sample_noise()
        staged.rollout()
        staged.update()
        assert np.array_equal(fused.costs_d.copy_to_host(), staged.costs_d.copy_to_host())
        assert np.array_equal(u_fused, staged.u_cur_d.copy_to_host())
        assert np.array_equal(lin_a.sample_grid_batch_d.copy_to_host(), lin_b.sample_grid_batch_d.copy_to_host())
        assert np.array_equal(ang_a.sample_grid_batch_d.copy_to_host(), ang_b.sample_grid_batch_d.copy_to_host())
