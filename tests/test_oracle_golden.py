"""Pins the CPU restatement (oracle/mppi_oracle.c) against fixtures produced by
the reference's own code on its CPU path (oracle/gen_golden.py, numba CUDA
simulator).  Every comparison here is BIT-EXACT unless stated: the restatement
follows the simulator's arithmetic rounding for rounding.

The reference has no tests of its own; these fixtures are
the golden vectors of this repository.
"""
import numpy as np
import pytest

from helpers import golden, iterations, params_from_golden, ulp_diff_f32
from oracle import oracle as O

MAP_FIXTURES = ["det_cvar", "det_mean", "speedmap_cvar", "speedmap_mean", "speedmap_mean_bounds", "tdm_cvar",
                "tdm_mean_alpha_dyn", "tdm_cvar_odd", "tdm_oversized_mean",
                # res = 0.3, dt = 0.05, reverse driving, |theta0| of several turns, bounds (0, 1.2)
                "det_odd_units", "speedmap_odd_units", "tdm_odd_units",
                # the same set-ups on other random worlds (GOLDEN_SEED_OFFSET=101 / 202)
                "det_odd_units_w101", "speedmap_odd_units_w101", "tdm_odd_units_w101",
                "det_odd_units_w202", "speedmap_odd_units_w202", "tdm_odd_units_w202"]
BAREBONE_FIXTURES = ["barebone_flat", "barebone_obstacles"]


def solve_of_iteration(g, k):
    s = 0
    while ("solve%d_first_iteration" % (s + 1)) in g and k >= int(g["solve%d_first_iteration" % (s + 1)]):
        s += 1
    return s


def map_params(g, P):
    return O.make_params(P, g["lin_res"], g["lin_padded_xlimits"], g["lin_padded_ylimits"],
                         g["lin_bin_values_bounds"], g["ang_bin_values_bounds"])


def barebone_params(P):
    # barebone notebook defaults: DEFAULT_OBS_COST = 1e3, DEFAULT_DIST_WEIGHT = 10
    return O.make_params(P, 1.0, [0, 0], [0, 0], [0.0, 1.0], [0.0, 1.0],
                         default_obs_cost=1e3, default_dist_weight=10)


def barebone_obstacles(P):
    if "obstacle_positions" in P:
        return P["obstacle_positions"], P["obstacle_radius"]
    # the notebook's dummy obstacle when none is given (cell 3, move_mppi_task_vars_to_device)
    return np.array([[1e5, 1e5]]), np.array([0.0])


def test_xoroshiro_known_answers():
    g = golden("rng_xoroshiro")
    st = O.xoroshiro_init(6, int(g["seed"]))
    assert (st[:, 0] == g["initial_s0"]).all() and (st[:, 1] == g["initial_s1"]).all()
    normals = np.array([[O.xoroshiro_normal(st, s) for _ in range(5)] for s in range(6)])
    assert (normals == g["normals_f64"]).all()
    st = O.xoroshiro_init(6, int(g["seed"]))
    uniforms = np.array([[O.xoroshiro_uniform(st, s) for _ in range(7)] for s in range(6)],
                        dtype=np.float32)
    assert (uniforms == g["uniforms_f32"]).all()
    st = O.xoroshiro_init(3, 12345)
    assert (st[:, 0] == g["seed12345_s0"]).all() and (st[:, 1] == g["seed12345_s1"]).all()
    # values quoted in SURVEY.md section 8c (seed 1)
    assert int(g["initial_s0"][0]) == 10451216379200822465
    assert abs(g["normals_f64"][0, 0] - (-1.4470639)) < 1e-6


@pytest.mark.parametrize("name", MAP_FIXTURES)
def test_rollout_costs_bit_exact(name):
    g = golden(name)
    P = params_from_golden(g)
    for k, it in enumerate(iterations(g)):
        s = solve_of_iteration(g, k)
        P["x0"] = g["solve%d_x0" % s]
        p = map_params(g, P)
        lin, ang = g["solve%d_lin_sample_grid" % s], g["solve%d_ang_sample_grid" % s]
        obs, unk = g["lin_obstacle_map_padded"], g["lin_unknown_map_padded"]
        if name.startswith("tdm"):
            c = O.rollout_tdm(p, lin, ang, obs, unk, it["noise"], it["u_in"])
        else:
            c = O.rollout_det(p, lin, ang, obs, unk, it["noise"], it["u_in"],
                              risk=g.get("lin_risk_traction_map_padded"))
        assert ulp_diff_f32(c, it["costs"]).max() == 0, (name, k)


@pytest.mark.parametrize("name", BAREBONE_FIXTURES)
def test_barebone_costs_bit_exact(name):
    g = golden(name)
    P = params_from_golden(g)
    pos, rad = barebone_obstacles(P)
    for k, it in enumerate(iterations(g)):
        P["x0"] = g["solve%d_x0" % k]
        c = O.rollout_barebone(barebone_params(P), pos, rad, it["noise"], it["u_in"])
        assert ulp_diff_f32(c, it["costs"]).max() == 0, (name, k)


@pytest.mark.parametrize("name", MAP_FIXTURES + BAREBONE_FIXTURES)
def test_update_serial_launch_bit_exact(name):
    """update_useq_numba launched [1,1] is deterministic: weights AND u match
    bit for bit."""
    g = golden(name)
    P = params_from_golden(g)
    for k, it in enumerate(iterations(g)):
        w, u, _ = O.update_useq(P["lambda_weight"], it["costs"], it["noise"], P["vrange"],
                                P["wrange"], it["u_in"], num_threads=1)
        assert ulp_diff_f32(w, g["it%d_serial_weights" % k]).max() == 0
        assert ulp_diff_f32(u, g["it%d_serial_u_out" % k]).max() == 0


@pytest.mark.parametrize("name", MAP_FIXTURES + BAREBONE_FIXTURES)
def test_update_reference_launch(name):
    """update_useq_numba[1,32]: the min / exp / sum trees are deterministic ->
    weights bit-exact (denormal-range weights excepted: the fixture generator
    accepts a simulator run when weights agree to 1e-4 relative, see
    gen_golden._UpdateRecorder); u is accumulated with unordered float32
    atomics in the reference -> compared to a few ulp of the control range."""
    g = golden(name)
    P = params_from_golden(g)
    for k, it in enumerate(iterations(g)):
        w, u, _ = O.update_useq(P["lambda_weight"], it["costs"], it["noise"], P["vrange"],
                                P["wrange"], it["u_in"])
        big = it["weights"] > 1e-30
        assert ulp_diff_f32(w[big], it["weights"][big]).max() == 0
        assert np.abs(w - it["weights"]).max() < 1e-30 or ulp_diff_f32(w, it["weights"])[big].max() == 0
        assert np.abs(u - it["u_out"]).max() <= 1e-6


@pytest.mark.parametrize("name", ["det_cvar", "speedmap_cvar", "tdm_cvar", "tdm_mean_alpha_dyn",
                                  "tdm_cvar_odd", "tdm_oversized_mean", "det_odd_units", "speedmap_odd_units",
                                  "tdm_odd_units", "det_odd_units_w101", "tdm_odd_units_w202"])
def test_noise_and_grids_from_seed(name):
    """xoroshiro streams persist across sample_noise / sample_grids calls."""
    g = golden(name)
    P = params_from_golden(g)
    n, t = int(g["cfg_num_control_rollouts"]), int(g["cfg_num_steps"])
    st = O.xoroshiro_init(n * t, int(g["cfg_seed"]))
    for it in iterations(g):
        assert ulp_diff_f32(O.sample_noise(st, P["u_std"], n, t), it["noise"]).max() == 0
    m = int(g["cfg_num_grid_samples"]) if name.startswith("tdm") else 1
    tx, ty = (int(v) for v in g["cfg_tdm_sample_thread_dim"])
    rows, cols = (int(v) for v in g["cfg_max_map_dim"])
    for tag in ("lin", "ang"):
        stg = O.xoroshiro_init(m * tx * ty, int(g["cfg_seed"]))
        table = O.bin_table(g[tag + "_bin_values"], g[tag + "_bin_values_bounds"])
        s = 0
        while ("solve%d_%s_sample_grid" % (s, tag)) in g:
            out = np.full((m, rows, cols), -7, dtype=np.int8)
            O.sample_grids(g[tag + "_pmf_grid_padded"], stg, m, (tx, ty), table,
                           float(P.get("alpha_dyn", 1.0)), out)
            want = g["solve%d_%s_sample_grid" % (s, tag)]
            rp, cp = want.shape[1:]
            assert (out[:, :rp, :cp] == want).all()
            assert (out[:, rp:, :] == -7).all() and (out[:, :, cp:] == -7).all()
            s += 1


@pytest.mark.parametrize("name", ["det_cvar", "speedmap_cvar", "tdm_cvar", "tdm_mean_alpha_dyn"])
def test_state_rollouts_bit_exact(name):
    g = golden(name)
    P = params_from_golden(g)
    its = iterations(g)
    P["x0"] = g["solve0_x0"]
    p = map_params(g, P)
    last = int(g["solve1_first_iteration"]) - 1 if "solve1_first_iteration" in g else len(its) - 1
    want = g["state_rollout_after_solve0"]
    lin, ang = g["solve0_lin_sample_grid"], g["solve0_ang_sample_grid"]
    if name.startswith("tdm"):
        got = O.state_rollout_envs(p, lin, ang, its[last]["u_out"], want.shape[0])
    else:
        # after solve(), u_prev_d aliases u_cur_d (mppi.py:362)
        got = O.state_rollout_noise(p, lin, ang, its[last]["noise"], its[last]["u_out"],
                                    its[last]["u_out"], want.shape[0])
    assert ulp_diff_f32(got, want).max() == 0


@pytest.mark.parametrize("name", BAREBONE_FIXTURES)
def test_barebone_state_rollouts_bit_exact(name):
    g = golden(name)
    P = params_from_golden(g)
    its = iterations(g)
    P["x0"] = g["solve0_x0"]
    want = g["state_rollout_after_solve0"]
    got = O.state_rollout_barebone(barebone_params(P), its[0]["noise"], its[0]["u_out"],
                                   its[0]["u_out"], want.shape[0])
    assert ulp_diff_f32(got, want).max() == 0


def test_fixture_coverage():
    """The fixtures exercise the branches the reference has: obstacle hits,
    early goal break, CVaR with float32-rounded alpha, alpha_dyn < 1."""
    for name in ("det_cvar", "speedmap_cvar", "tdm_cvar", "barebone_obstacles"):
        costs = np.concatenate([it["costs"] for it in iterations(golden(name))])
        assert (costs > 1e4).any(), name + ": some rollouts must hit an obstacle"
    for name in ("speedmap_cvar", "tdm_mean_alpha_dyn", "barebone_flat"):
        costs = np.concatenate([it["costs"] for it in iterations(golden(name))])
        assert (costs < 100).any(), name + ": some rollouts must reach the goal"
    g = golden("tdm_cvar_odd")
    m, alpha = int(g["cfg_num_grid_samples"]), np.float32(g["param_cvar_alpha"])
    assert int(np.ceil(float(m) * float(alpha))) == 4  # 10 * float32(0.3) -> 4, not 3
    assert float(golden("tdm_mean_alpha_dyn")["param_alpha_dyn"]) < 1.0
