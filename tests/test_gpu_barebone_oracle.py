"""The barebone rollout (k_rollout_barebone: heading by rotation where the host can bound the increment, full float64 sincos
otherwise) against the C restatement of the notebook's rollout kernel (barebone_mppi_numba.ipynb cell 3; oracle/mppi_oracle.c:
oracle_rollout_barebone), bit for bit, through the C ABI: random problems -- sizes incl. the reference's published
configuration's shape, disc counts, goals inside and outside the reach, reverse driving, turn rates beyond the rotation's
range.  (Round 6 also built a time-parallel form of this kernel; it was not faster -- DESIGN.md section 10 -- and these
cases, written for it, stay for the kernel that runs.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def problem(rng, n, t, n_discs, goal_near, reverse):
    dt = float(rng.choice([0.05, 0.1, 0.2]))
    x0 = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-np.pi, np.pi)])
    reach = t * dt * 2.0
    goal = x0[:2] + (rng.uniform(0.05, 0.4) if goal_near else rng.uniform(0.8, 1.5)) * reach * np.array([np.cos(x0[2]), np.sin(x0[2])])
    pos = rng.uniform(-0.5, 0.5, (max(n_discs, 1), 2)) * reach + x0[:2]
    rad = rng.uniform(0.1, 0.3, max(n_discs, 1)) * reach
    if n_discs == 0:
        rad[:] = 0.0
        pos[:] = 1e6
    params = dict(dt=dt, x0=x0, xgoal=goal, goal_tolerance=float(rng.uniform(0.2, 0.6)), dist_weight=float(rng.choice([1, 10, 3.5])),
                  lambda_weight=float(rng.choice([1.0, 0.3])), num_opt=1, u_std=np.array([rng.uniform(0.3, 1.5), rng.uniform(0.3, 1.5)]),
                  vrange=np.array([-1.0 if reverse else 0.0, 2.0]), wrange=np.array([-np.pi, np.pi]) * rng.choice([0.5, 1.0, 3.0]),
                  obstacle_positions=pos, obstacle_radius=rad, obs_penalty=float(rng.choice([1e6, 50.0])))
    return dict(T=(t + 0.5) * dt, dt=dt, num_control_rollouts=n, num_vis_state_rollouts=1, seed=int(rng.integers(1, 1000))), params


def run_case(seed, n, t, n_discs, goal_near=False, reverse=False, debug_flags=0):
    from mppi_numba_amd.barebone import Config, MPPI_Numba
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    cfg_kwargs, params = problem(rng, n, t, n_discs, goal_near, reverse)
    cfg = Config(enforce_recommended_limits=False, **cfg_kwargs)
    assert cfg.num_steps == t
    planner = MPPI_Numba(cfg)
    planner.setup(params)
    if debug_flags:
        planner.set_debug_flags(debug_flags)
    p = O.make_params(params, 1.0, [0, 0], [0, 0], [0.0, 1.0], [0.0, 1.0], default_obs_cost=1e3, default_dist_weight=10)
    out = []
    for it in range(2):
        planner.sample_noise()
        noise, u_in = planner.noise_samples_d.copy_to_host(), planner.u_cur_d.copy_to_host()
        planner.rollout()
        got = planner.costs_d.copy_to_host()
        want = O.rollout_barebone(p, params["obstacle_positions"], params["obstacle_radius"], noise, u_in)
        out.append((got, want, planner.last_rollout_kernel()))
        planner.update()
    return out


CASES = [  # (seed, N, T, discs, goal near the start, reverse driving allowed)
    (1, 1000, 50, 2, False, False),   # the reference's published configuration's shape
    (2, 64, 30, 0, False, False),     # BASELINE configs[0]'s shape
    (3, 1, 2, 0, False, False), (4, 63, 7, 1, True, False), (5, 65, 8, 3, True, True), (6, 4097, 9, 5, False, True),
    (7, 1000, 56, 32, True, False), (11, 300, 200, 0, False, False), (12, 300, 57, 33, True, False),
    (8, 777, 33, 2, True, True), (9, 2048, 50, 2, True, False), (10, 500, 17, 7, False, False),
]


@pytest.mark.parametrize("seed,n,t,discs,near,reverse", CASES)
def test_barebone_vs_oracle_bit_for_bit(seed, n, t, discs, near, reverse):
    kernels = set()
    for got, want, kernel in run_case(seed, n, t, discs, near, reverse):
        assert kernel.startswith("k_rollout_barebone"), kernel
        kernels.add(kernel)
        assert np.isfinite(want).all()
        assert (got.view(np.int32) == want.view(np.int32)).all(), \
            "%d of %d costs differ, max rel %.3g" % ((got != want).sum(), n, (np.abs(got - want) / np.abs(want)).max())


def test_both_forms_of_the_heading_are_covered():
    """wrange * dt beyond 0.36 rad selects the full sincos, below it the rotation: the cases above contain both."""
    seen = set()
    for seed, n, t, discs, near, reverse in CASES:
        seen.add(run_case(seed, min(n, 128), t, discs, near, reverse)[0][2].split("rotation=")[-1][:1])
    assert seen == {"0", "1"}, seen


def test_goal_cases_really_reach_the_goal():
    """At least one of the `goal near` cases has rollouts that stop inside the goal circle and rollouts that do not:
    their costs differ in kind (no terminal cost), which the oracle's agreement above then covers."""
    mixed = 0
    for seed, n, t, discs, near, reverse in CASES:
        if not near:
            continue
        got, want, _ = run_case(seed, n, t, discs, near, reverse)[0]
        mixed += int(want.min() < 0.5 * np.median(want) or want.max() > 2.0 * np.median(want))
    assert mixed >= 1
