"""The ALGORITHM of the time-parallel rollout with the reference's rounding points (tests/scan_exact_model.py,
the numpy statement of csrc/rollout_scan_exact_kernel.h) against the oracle on the CPU: three walked sums,
everything else side by side, goal breaks, rollouts that stop in the zero-traction ring and keep paying
through ordinary records, the vote.  Bits of the oracle wherever the vote holds."""
import numpy as np
import pytest

from oracle import oracle as O
from scan_exact_model import scan_exact_rollout
from test_scan_model import inputs, params, world


def run(P, t, n=256, patch=False, seed=1):
    lin, ang, obs, unk, limits = world(patch=patch)
    p = O.make_params(P, 0.25, limits, limits, [0.0, 1.0], [0.0, 1.0])
    noise, u = inputs(n, t, seed=seed)
    want = O.rollout_det(p, lin, ang, obs, unk, noise, u)
    got, failed = scan_exact_rollout(p, lin, ang, obs, unk, noise, u)
    return got, want, failed


@pytest.mark.parametrize("t", [100, 37, 8, 3, 104])
def test_bits_of_the_oracle_on_a_nominal_map(t):
    got, want, failed = run(params(), t)
    assert not failed.any()
    assert np.array_equal(got, want), np.abs(got - want).max()


@pytest.mark.parametrize("goal", [(9.0, 9.0), (4.2, 4.1), (5.0, 4.5)])
def test_goal_breaks(goal):
    """the goal within the first chunks, inside the start cell, a few steps away"""
    got, want, failed = run(params(goal=goal), 60, seed=3)
    assert not failed.any()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("x0", [(0.3, 30.0, np.pi), (30.0, 0.2, -np.pi / 2), (0.1, 0.1, 3.9)])
def test_rollouts_that_stop_in_the_zero_traction_ring(x0):
    """start one cell from the padding ring, heading into it: most rollouts stop within a few steps and
    pay the stage cost of the place where they stand for the rest of the horizon (some of them inside
    obstacle / unknown cells: penalties every step)"""
    got, want, failed = run(params(x0=x0), 90, seed=5)
    assert not failed.any()
    assert np.array_equal(got, want)
    lin, ang, obs, unk, limits = world()
    # (the case is what it says: the oracle's costs are those of rollouts that stopped -- far above T * dt)
    assert np.median(want) > 100.0


def test_stop_inside_the_goal_circle():
    """a goal on the edge of the map: rollouts that enter the ring within the goal tolerance pay one more
    stage cost and are done"""
    got, want, failed = run(dict(params(goal=(0.1, 30.0), x0=(0.6, 30.0, np.pi)), goal_tolerance=0.9), 50, seed=7)
    assert not failed.any()
    assert np.array_equal(got, want)


def test_vote_fails_where_the_traction_changes():
    """a slow patch in the way: tiles that reach it report a failed vote (the kernel re-runs them step by
    step); the others keep the oracle's bits"""
    got, want, failed = run(params(), 100, n=512, patch=True)
    assert failed.any()
    ok = ~np.repeat(failed, 32)[:512]
    assert ok.any()
    assert np.array_equal(got[ok], want[ok])
