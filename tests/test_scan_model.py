"""The time-parallel rollout ALGORITHM (tests/scan_model.py, the numpy statement of
csrc/rollout_scan_kernel.h) against the oracle on the CPU: a nominal-traction world of
BASELINE configs[1]'s shape, where every tile keeps its assumption, and a world with a slow
patch, where the tiles that reach it must report a failed vote (the kernel then re-runs them
sequentially)."""
import numpy as np
import pytest

from oracle import oracle as O
from scan_model import scan_rollout


def world(patch=False, seed=0):
    rng = np.random.default_rng(seed)
    rows = cols = 256
    pad = 2
    obstacle = (rng.random((rows, cols)) < 0.02).astype(np.int8)
    unknown = (rng.random((rows, cols)) < 0.02).astype(np.int8)
    for m in (obstacle, unknown):
        m[12:20, 12:20] = 0
        m[236:244, 236:244] = 0
    lin = np.zeros((1, rows + 2 * pad, cols + 2 * pad), dtype=np.int8)
    lin[0, pad:-pad, pad:-pad] = 100
    ang = lin.copy()
    if patch:
        lin[0, 60:120, 60:120] = 40
        ang[0, 60:120, 60:120] = 55
    obs = np.zeros((rows + 2 * pad, cols + 2 * pad), dtype=np.int8)
    unk = obs.copy()
    obs[pad:-pad, pad:-pad] = obstacle
    unk[pad:-pad, pad:-pad] = unknown
    limits = np.array([-pad * 0.25, (cols + pad) * 0.25])
    return lin, ang, obs, unk, limits


def params(goal=(60.0, 60.0), x0=(4.0, 4.0, np.pi / 4)):
    return dict(x0=np.array(x0), xgoal=np.array(goal), dt=0.1, goal_tolerance=0.5, v_post_rollout=0.01,
                lambda_weight=1.0, cvar_alpha=1.0, alpha_dyn=1.0, num_opt=1, u_std=np.array([2.0, 3.0]),
                vrange=np.array([0.0, 3.0]), wrange=np.array([-np.pi, np.pi]), dist_weight=1.0,
                obs_penalty=1e5, unknown_penalty=1e2)


def inputs(n, t, seed=1):
    rng = np.random.default_rng(seed)
    noise = (rng.standard_normal((n, t, 2)) * np.array([2.0, 3.0])).astype(np.float32)
    u = np.stack([np.full(t, 2.2), 0.3 * np.sin(np.arange(t) / 9.0)], axis=1).astype(np.float32)
    return noise, u


@pytest.mark.parametrize("t,ch", [(100, 8), (100, 4), (37, 8), (120, 8), (99, 4)])
def test_model_matches_the_oracle_on_a_nominal_map(t, ch):
    lin, ang, obs, unk, limits = world()
    P = params()
    p = O.make_params(P, 0.25, limits, limits, [0.0, 1.0], [0.0, 1.0])
    noise, u = inputs(4096, t)
    want = O.rollout_det(p, lin, ang, obs, unk, noise, u)
    got, failed = scan_rollout(p, lin, ang, obs, unk, noise, u, ch=ch)
    assert not failed.any()
    rel = np.abs(got - want) / np.abs(want)
    # rollouts that graze a cell border or the goal circle may land on the other side.  (A rollout
    # frozen in the padding ring adds the SAME stage cost for the rest of the horizon: with a
    # float32 addend every one of those additions can round the other way -- 0.13 % of the costs
    # beyond 1e-6; the frozen steps are therefore taken with the float64 addend, see frozen_block)
    assert np.quantile(rel, 0.999) < 1e-6, np.quantile(rel, [0.5, 0.99, 0.999, 1.0])
    assert (rel < 1e-5).mean() >= 0.9995
    assert (rel == 0).mean() > 0.7  # most costs come out bit-identical


def test_goal_break_and_start_inside_the_goal_circle():
    lin, ang, obs, unk, limits = world()
    noise, u = inputs(1024, 100, seed=3)
    for goal in [(9.0, 9.0), (4.2, 4.1)]:  # reached within the first chunks; reached by step 0
        p = O.make_params(params(goal=goal), 0.25, limits, limits, [0.0, 1.0], [0.0, 1.0])
        want = O.rollout_det(p, lin, ang, obs, unk, noise, u)
        got, failed = scan_rollout(p, lin, ang, obs, unk, noise, u)
        # (these costs are a few control-cost terms of either sign around zero: measured against the
        #  size of the terms, not of their sum)
        rel = np.abs(got - want) / np.maximum(np.abs(want), 30.0)
        assert not failed.any() and np.quantile(rel, 0.99) < 1e-6, (goal, np.quantile(rel, [0.5, 0.99, 1.0]))


def test_rollouts_frozen_in_the_zero_traction_ring():
    """Start next to the map border heading out: most rollouts enter the padding ring (traction 0)
    and stay there for the rest of the horizon, which the reference computes step by step."""
    lin, ang, obs, unk, limits = world()
    p = O.make_params(params(x0=(1.0, 30.0, np.pi)), 0.25, limits, limits, [0.0, 1.0], [0.0, 1.0])
    noise, u = inputs(2048, 100, seed=5)
    want = O.rollout_det(p, lin, ang, obs, unk, noise, u)
    got, failed = scan_rollout(p, lin, ang, obs, unk, noise, u)
    rel = np.abs(got - want) / np.abs(want)
    assert not failed.any()
    assert np.quantile(rel, 0.995) < 1e-6, np.quantile(rel, [0.5, 0.99, 0.999, 1.0])


def test_vote_fails_where_the_traction_changes():
    lin, ang, obs, unk, limits = world(patch=True)
    p = O.make_params(params(x0=(10.0, 10.0, np.pi / 4)), 0.25, limits, limits, [0.0, 1.0], [0.0, 1.0])
    noise, u = inputs(2048, 100, seed=7)
    want = O.rollout_det(p, lin, ang, obs, unk, noise, u)
    got, failed = scan_rollout(p, lin, ang, obs, unk, noise, u)
    assert failed.any()  # the patch at 15..30 m is within reach
    ok = ~np.repeat(failed, 64)
    if ok.any():
        rel = np.abs(got[ok] - want[ok]) / np.abs(want[ok])
        assert np.quantile(rel, 0.99) < 1e-6
    # and a tile that never fails must not have been touched by the patch: its costs match (checked
    # above); a tile that fails is re-run by the kernel -- nothing of the model's result is used


def test_frozen_block_equals_the_additions_one_by_one():
    from scan_model import frozen_block
    rng = np.random.default_rng(11)
    n = 4000
    acc = (10.0 ** rng.uniform(-1, 6, n)).astype(np.float32)
    k = 10.0 ** rng.uniform(-2, 2.5, n)
    pen = np.where(rng.random(n) < 0.1, np.float32(100.0), np.float32(0.0)).astype(np.float32)
    count = rng.integers(0, 130, n)
    want = acc.copy()
    for i in range(130):
        on = i < count
        want = np.where(on, ((want.astype(np.float64) + k).astype(np.float32) + pen).astype(np.float32), want)
    got = frozen_block(acc, k, pen, count)
    # (an exact tie of the float64 addend would need the parity of the running sum: measure zero here)
    assert (got == want).all(), np.abs(got - want).max()
