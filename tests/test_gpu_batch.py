"""Batched multi-query handle (BASELINE.json configs[4]): B
problems that share the maps, one launch over (problem, rollout).

The batch is not in the reference, so its oracle is two-fold: every problem must be
bit-identical to a single-problem handle fed the same noise and controls, and its costs
must match the CPU restatement of the reference's rollout kernels (oracle/)."""
import numpy as np
import pytest

import bench
from helpers import ulp_diff_f32
from oracle import oracle as O
from test_gpu_scale import oracle_params

pytestmark = pytest.mark.gpu


def make_world(workload, n, t_steps, m=1):
    from mppi_numba_amd.config import Config
    from mppi_numba_amd.terrain import TDM_Numba
    w = dict(bench.WORKLOADS[workload])
    cfg = Config(T=t_steps * 0.1, dt=0.1, num_grid_samples=m, num_control_rollouts=n,
                 max_speed_padding=5.0, num_vis_state_rollouts=4, max_map_dim=(260, 260), seed=3,
                 enforce_recommended_limits=False, **w["mode"])
    assert cfg.num_steps == t_steps
    pmf, obstacle, unknown, tdm_dict = bench.synthetic_world(workload, np.random.default_rng(0))
    lin, ang = TDM_Numba(cfg), TDM_Numba(cfg)
    lin.set_TDM_from_PMF_grid(pmf, tdm_dict, obstacle, unknown)
    ang.set_TDM_from_PMF_grid(pmf, tdm_dict, obstacle, unknown)
    return cfg, lin, ang, bench.make_params(workload)


def problems(lin, count, rng):
    """Start states spread over the map, including the corners (window clipped at the border)."""
    (x_lo, x_hi), (y_lo, y_hi) = lin.xlimits, lin.ylimits
    x0s = np.stack([rng.uniform(x_lo + 0.3, x_hi - 0.3, count), rng.uniform(y_lo + 0.3, y_hi - 0.3, count),
                    rng.uniform(-np.pi, np.pi, count)], axis=1).astype(np.float32)
    x0s[0, :2] = (x_lo + 0.05, y_lo + 0.05)
    x0s[1, :2] = (x_hi - 0.05, y_hi - 0.05)
    if count > 2:
        x0s[2, :2] = (x_lo + 0.05, y_hi - 0.05)
    goals = np.stack([rng.uniform(x_lo + 1, x_hi - 1, count), rng.uniform(y_lo + 1, y_hi - 1, count)],
                     axis=1).astype(np.float32)
    goals[-1] = x0s[-1, :2] + 0.5  # one goal close enough to be reached within the horizon
    return x0s, goals


def single_problem(cfg, lin, ang, params, x0, goal):
    from mppi_numba_amd.mppi import MPPI_Numba
    planner = MPPI_Numba(cfg)
    p = dict(params)
    p["x0"], p["xgoal"] = np.asarray(x0, dtype=np.float32), np.asarray(goal, dtype=np.float32)
    planner.setup(p, lin, ang)
    return planner, p


@pytest.mark.parametrize("workload,n,t_steps,m,count,token", [
    ("c2", 1024, 60, 1, 5, "k_rollout_scan_exact"),     # deterministic traction: 160 tiles of 32, one round
    ("c2s", 1024, 60, 1, 5, "k_rollout_scan_exact"),    # semantic map: after solve() the exact schedule inside that kernel (direct), one window origin per problem
    ("c2", 2048, 60, 1, 6, "k_rollout_pipe"),           # 192 tiles of 64: one tile per CU, LDS reach windows
    ("c2", 256, 250, 1, 3, "k_rollout_"),               # long horizon: whole map or global cells
    ("c2", 4096, 30, 1, 12, "k_rollout_fused"),  # throughput regime: fused kernel, LDS windows
    ("c3", 128, 40, 64, 3, "k_rollout_tdm"),            # CVaR over M sampled maps
])
def test_batch_matches_single_problem_handles_and_oracle(workload, n, t_steps, m, count, token):
    from mppi_numba_amd.batch import MPPI_Batch
    cfg, lin, ang, params = make_world(workload, n, t_steps, m)
    rng = np.random.default_rng(5)
    x0s, goals = problems(lin, count, rng)
    batch = MPPI_Batch(cfg, count)
    batch.setup(params, lin, ang, x0s, goals)
    useqs = batch.solve()  # samples the traction maps; the stage-level calls below reuse them
    if workload == "c2s":  # (from zero controls nobody leaves its start patch; a few control steps later tiles fail their vote
        for _ in range(3):  #  and the planner stops speculating at the synchronisation that follows)
            useqs = batch.solve()
    assert useqs.shape == (count, t_steps, 2) and np.isfinite(useqs).all()
    assert token in batch.last_rollout_kernel()
    if token in ("k_rollout_pipe", "k_rollout_scan_exact"):
        assert "problems=%d" % count in batch.last_rollout_kernel()
    # a different warm start per problem, then injected noise
    u_in = (useqs + rng.normal(0, 0.05, useqs.shape)).astype(np.float32)
    batch.set_u(u_in)
    batch.sample_noise()
    noise = batch.noise_samples_d.copy_to_host().reshape(count, n, t_steps, 2)
    batch.rollout()
    if workload == "c2s":
        assert "direct=1" in batch.last_rollout_kernel(), batch.last_rollout_kernel()
    costs = batch.costs_d.copy_to_host()
    assert costs.shape == (count, n)
    batch.update()
    u_out = batch.u_cur_d.copy_to_host()
    weights = batch.weights_d.copy_to_host()
    assert np.allclose(weights.sum(axis=1), 1.0, atol=1e-5)
    grids = (lin.sample_grid_batch_d.copy_to_host(), ang.sample_grid_batch_d.copy_to_host(),
             lin.obstacle_map_d.copy_to_host(), lin.unknown_map_d.copy_to_host())
    for b in range(count):
        single, p = single_problem(cfg, lin, ang, params, x0s[b], goals[b])
        if token != "k_rollout_scan_exact":
            # alone, a problem of this size can be one round of the time-parallel kernel, whose update sums the
            # weighted noise per 32-rollout tile: same costs, another float64 summation tree for u.  Bit
            # equality of u is a property of one kernel family: keep the single handle on the batch's.
            from mppi_numba_amd import _lib
            single.set_debug_flags(_lib.DEBUG_NO_SCAN_KERNEL)
        single.set_u(u_in[b])
        single.set_noise(noise[b])
        single.rollout()
        if token in ("k_rollout_pipe", "k_rollout_scan_exact"):
            assert token in single.last_rollout_kernel(), single.last_rollout_kernel()
        want = single.costs_d.copy_to_host()
        assert np.array_equal(costs[b], want), "problem %d: costs differ from the single-problem handle" % b
        single.update()
        assert np.array_equal(u_out[b], single.u_cur_d.copy_to_host()), "problem %d: update differs" % b
        assert np.array_equal(weights[b], single.weights_d.copy_to_host())
        # and against the CPU restatement of the reference
        op = oracle_params(p, lin, ang)
        ref = O.rollout_tdm(op, *grids, noise[b], u_in[b]) if m > 1 else O.rollout_det(op, *grids, noise[b], u_in[b])
        ulps = ulp_diff_f32(costs[b], ref)
        assert (ulps == 0).mean() >= 0.999, "problem %d: exact fraction %.5f" % (b, (ulps == 0).mean())
        _, u_ref, _ = O.update_useq(p["lambda_weight"], ref, noise[b], p["vrange"], p["wrange"], u_in[b])
        assert (np.abs(u_out[b] - u_ref) / np.array([3.0, np.pi])).max() <= 1e-5


def test_c5_full_size_vs_oracle():
    """BASELINE configs[4] at full size: 64 problems x N=4096, T=100 in one launch (the objects
    bench.py --workload c5 times), every problem against the CPU restatement of
    mppi.py:916-1009 and 1113-1191."""
    from mppi_numba_amd.batch import MPPI_Batch
    w = bench.WORKLOADS["c5"]
    n, t_steps, count = w["n"], w["t"], w["problems"]
    cfg, lin, ang, params = make_world("c5", n, t_steps)
    x0s, goals = bench.batch_problems(count, np.random.default_rng(100))
    batch = MPPI_Batch(cfg, count)
    batch.setup(params, lin, ang, x0s, goals)
    batch.solve()
    batch.iterate_async(3)
    batch.synchronize()
    assert batch.last_rollout_kernel().startswith("k_rollout_fused"), batch.last_rollout_kernel()
    assert "problems=%d" % count in batch.last_rollout_kernel()
    u_in = batch.u_cur_d.copy_to_host()
    batch.sample_noise()
    noise = batch.noise_samples_d.copy_to_host().reshape(count, n, t_steps, 2)
    batch.rollout()
    costs = batch.costs_d.copy_to_host()
    batch.update()
    u_out = batch.u_cur_d.copy_to_host()
    grids = (lin.sample_grid_batch_d.copy_to_host(), ang.sample_grid_batch_d.copy_to_host(),
             lin.obstacle_map_d.copy_to_host(), lin.unknown_map_d.copy_to_host())
    exact, worst_u = [], 0.0
    for b in range(count):
        p = dict(params)
        p["x0"], p["xgoal"] = x0s[b], goals[b]
        ref = O.rollout_det(oracle_params(p, lin, ang), *grids, noise[b], u_in[b])
        ulps = ulp_diff_f32(costs[b], ref)
        exact.append((ulps == 0).mean())
        assert exact[-1] >= 0.999, "problem %d: exact fraction %.5f" % (b, exact[-1])
        assert (np.abs(costs[b] - ref) / np.abs(ref)).max() < 1e-6
        _, u_ref, _ = O.update_useq(p["lambda_weight"], ref, noise[b], p["vrange"], p["wrange"], u_in[b])
        worst_u = max(worst_u, float((np.abs(u_out[b] - u_ref) / np.array([3.0, np.pi])).max()))
    print("\nc5 64 x 4096: exact costs min %.5f, max |du|/range %.3e" % (min(exact), worst_u))
    assert worst_u <= 1e-5


def test_one_problem_through_the_instance_path_equals_the_classic_path():
    """count = 1 takes the per-problem device parameters; params['x0'] the by-value ones."""
    from mppi_numba_amd.batch import MPPI_Batch
    cfg, lin, ang, params = make_world("c2", 2048, 80)
    x0 = np.array([[7.3, 41.0, 2.2]], dtype=np.float32)
    goal = np.array([[30.0, 20.0]], dtype=np.float32)
    batch = MPPI_Batch(cfg, 1)
    batch.setup(params, lin, ang, x0, goal)
    classic, _ = single_problem(cfg, lin, ang, params, x0[0], goal[0])
    lin.sample_grids()
    ang.sample_grids()
    batch.sample_noise()
    noise = batch.noise_samples_d.copy_to_host()
    classic.set_noise(noise)
    batch.rollout()
    classic.rollout()
    assert "problems=1" in batch.last_rollout_kernel() and "problems=0" in classic.last_rollout_kernel()
    assert np.array_equal(batch.costs_d.copy_to_host(), classic.costs_d.copy_to_host())
    batch.update()
    classic.update()
    assert np.array_equal(batch.u_cur_d.copy_to_host().reshape(-1, 2), classic.u_cur_d.copy_to_host())


def test_speed_map_batch_matches_single():
    from mppi_numba_amd.batch import MPPI_Batch
    from mppi_numba_amd.config import Config
    from mppi_numba_amd.terrain import TDM_Numba
    n, t_steps, count = 512, 50, 3
    cfg = Config(T=t_steps * 0.1, dt=0.1, num_grid_samples=1, num_control_rollouts=n, max_speed_padding=5.0,
                 num_vis_state_rollouts=2, max_map_dim=(260, 260), seed=4, enforce_recommended_limits=False,
                 use_nom_dynamics_with_speed_map=True)
    pmf, obstacle, unknown, tdm_dict = bench.synthetic_world("c2", np.random.default_rng(0))
    lin, ang = TDM_Numba(cfg), TDM_Numba(cfg)
    lin.set_TDM_from_PMF_grid(pmf, tdm_dict, obstacle, unknown)
    ang.set_TDM_from_PMF_grid(pmf, tdm_dict, obstacle, unknown)
    params = bench.make_params("c2")
    rng = np.random.default_rng(9)
    x0s, goals = problems(lin, count, rng)
    batch = MPPI_Batch(cfg, count)
    batch.setup(params, lin, ang, x0s, goals)
    batch.solve()
    batch.sample_noise()
    noise = batch.noise_samples_d.copy_to_host().reshape(count, n, t_steps, 2)
    u_in = batch.u_cur_d.copy_to_host()
    batch.rollout()
    costs = batch.costs_d.copy_to_host()
    for b in range(count):
        single, _ = single_problem(cfg, lin, ang, params, x0s[b], goals[b])
        single.set_u(u_in[b])
        single.set_noise(noise[b])
        single.rollout()
        assert np.array_equal(costs[b], single.costs_d.copy_to_host())


def test_batch_closed_loop_every_problem_reaches_its_goal():
    """Receding-horizon loop on the nominal traction: all problems converge, each to ITS goal."""
    from mppi_numba_amd.batch import MPPI_Batch
    cfg, lin, ang, params = make_world("c2", 1024, 50)
    params = dict(params, num_opt=2)
    count = 6
    rng = np.random.default_rng(11)
    (x_lo, x_hi), (y_lo, y_hi) = lin.xlimits, lin.ylimits
    x0s = np.stack([rng.uniform(x_lo + 10, x_hi - 10, count), rng.uniform(y_lo + 10, y_hi - 10, count),
                    rng.uniform(-3, 3, count)], axis=1).astype(np.float32)
    goals = (x0s[:, :2] + rng.uniform(-6, 6, (count, 2))).astype(np.float32)
    batch = MPPI_Batch(cfg, count)
    batch.setup(params, lin, ang, x0s, goals)
    lin_grid = lin.sample_grid_batch_d.copy_to_host()
    state = x0s.astype(np.float64)
    start_dist = np.linalg.norm(state[:, :2] - goals, axis=1)
    dt = params["dt"]
    for _ in range(120):
        useqs = batch.solve()
        v, w = useqs[:, 0, 0].astype(np.float64), useqs[:, 0, 1].astype(np.float64)
        # nominal (traction 1) unicycle step on the host; the planner sees reduced traction
        state[:, 0] += dt * v * np.cos(state[:, 2]) * 0.8
        state[:, 1] += dt * v * np.sin(state[:, 2]) * 0.8
        state[:, 2] += dt * w * 0.8
        batch.shift_and_update(state.astype(np.float32), useqs, 1)
    end_dist = np.linalg.norm(state[:, :2] - goals, axis=1)
    assert lin_grid.size > 0
    assert (end_dist < np.maximum(1.0, 0.25 * start_dist)).all(), (start_dist, end_dist)


def test_batch_state_rollouts_follow_each_problems_start():
    from mppi_numba_amd.batch import MPPI_Batch
    cfg, lin, ang, params = make_world("c2", 256, 30)
    rng = np.random.default_rng(2)
    x0s, goals = problems(lin, 3, rng)
    batch = MPPI_Batch(cfg, 3)
    batch.setup(params, lin, ang, x0s, goals)
    useqs = batch.solve()
    for b in range(3):
        got = batch.get_state_rollout(b)
        assert got.shape == (cfg.num_vis_state_rollouts, 31, 3)
        assert np.allclose(got[:, 0, :], x0s[b])
        single, _ = single_problem(cfg, lin, ang, params, x0s[b], goals[b])
        single.set_u(useqs[b])
        assert np.array_equal(single.get_state_rollout()[0], got[0])


def test_batch_sharded_over_two_handles_matches_one():
    """world_size 2 on one GPU through the packet interface: B*(2T+2) doubles per rank."""
    from mppi_numba_amd.batch import MPPI_Batch
    cfg, lin, ang, params = make_world("c2", 1024, 40)
    count = 3
    rng = np.random.default_rng(6)
    x0s, goals = problems(lin, count, rng)
    whole = MPPI_Batch(cfg, count)
    whole.setup(params, lin, ang, x0s, goals)
    lin.sample_grids()
    ang.sample_grids()
    whole.sample_noise()
    noise = whole.noise_samples_d.copy_to_host().reshape(count, 1024, 40, 2)
    whole.rollout()
    whole.update()
    want = whole.u_cur_d.copy_to_host()
    halves = [MPPI_Batch(cfg, count, rank=r, world_size=2) for r in range(2)]
    packets = []
    for r, h in enumerate(halves):
        h.setup(params, lin, ang, x0s, goals)
        h.set_noise(np.ascontiguousarray(noise[:, r * 512:(r + 1) * 512]).reshape(count * 512, 40, 2))
        h.rollout()
        packets.append(h.update_local())
    assert packets[0].shape == (count * 82,)
    for h in halves:
        h.update_apply(np.stack(packets))
        got = h.u_cur_d.copy_to_host()
        assert np.abs(got - want).max() <= 2e-6


def test_batch_argument_errors():
    from mppi_numba_amd.batch import MPPI_Batch
    from mppi_numba_amd.config import Config
    cfg, lin, ang, params = make_world("c2", 256, 20)
    batch = MPPI_Batch(cfg, 2)
    batch.set_tdm(lin, ang)
    batch.set_params(params)
    assert batch.solve() is None  # instances not set: the mirror prints and returns None
    bad = Config(T=2.0, dt=0.1, num_grid_samples=1, num_control_rollouts=200, max_speed_padding=5.0,
                 num_vis_state_rollouts=1, max_map_dim=(260, 260), seed=1, enforce_recommended_limits=False,
                 use_det_dynamics=True)
    with pytest.raises(RuntimeError, match="multiple of 64"):
        MPPI_Batch(bad, 2)
