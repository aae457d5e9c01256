"""Soak of the flag-synchronised time-parallel exact kernel (rollout_scan_exact_kernel.h): its waves hand
groups of 8 steps to each other through LDS flags, without workgroup barriers, relying on the LDS executing a
wave's instructions in order -- an ordering slip would show up rarely and as wrong costs, not as a crash.
20 000 iterations of BASELINE configs[1] (C2: N = 8192, T = 100) in the ordinary loop (noise computed in the
launch, updates applied by the next launch); every 100th iteration is re-computed by k_rollout_pipe (barriers
between its stages) from the same Philox counters and the same controls and compared bit for bit, and the whole
run is repeated with an update launch per iteration: the control sequences must come out identical.
(mppi.py:916-1009.)"""
import numpy as np
import pytest

import bench
from mppi_numba_amd import _lib

pytestmark = pytest.mark.gpu

ROUNDS, PER_ROUND = 200, 100


def test_twenty_thousand_iterations_against_the_barrier_synchronised_kernel():
    _, _, _, _, loop, params = bench.build_planner("c2")
    _, _, _, _, plain, _ = bench.build_planner("c2")
    _, _, _, _, check, _ = bench.build_planner("c2")
    plain.set_debug_flags(_lib.DEBUG_NO_REDUCE_FOLD)
    check.set_debug_flags(_lib.DEBUG_NO_SCAN_KERNEL)
    for planner in (loop, plain, check):
        planner.solve()
    worst = 0
    for rnd in range(ROUNDS):
        loop.iterate_async(PER_ROUND - 1)
        loop.synchronize()
        assert "reduces_tiles=1" in loop.last_rollout_kernel()
        u_in = loop.u_cur_d.copy_to_host()
        loop.iterate_async(1)
        loop.synchronize()
        name = loop.last_rollout_kernel()
        assert name.startswith("k_rollout_scan_exact") and "noise=in-kernel" in name, name
        costs = loop.costs_d.copy_to_host()
        noise = loop.noise_samples_d.copy_to_host()  # (regenerated from the counters the launch used)
        check.set_noise(noise)
        check.set_u(u_in)
        check.rollout()
        assert check.last_rollout_kernel().startswith("k_rollout_pipe"), check.last_rollout_kernel()
        want = check.costs_d.copy_to_host()
        differing = int((costs != want).sum())
        worst = max(worst, differing)
        assert differing == 0, (rnd, differing, np.abs(costs - want).max())
        plain.iterate_async(PER_ROUND - 1)
        plain.iterate_async(1)
    plain.synchronize()
    assert np.array_equal(loop.u_cur_d.copy_to_host(), plain.u_cur_d.copy_to_host())
    assert np.array_equal(loop.costs_d.copy_to_host(), plain.costs_d.copy_to_host())
    print("\n%d iterations, %d checked bit for bit against k_rollout_pipe: 0 costs differ" % (ROUNDS * PER_ROUND, ROUNDS))


def test_ten_thousand_iterations_of_the_tolerance_kernel_with_and_without_the_fold():
    """math="fast": 10 000 iterations in calls of uneven length, one launch per iteration against an update launch per
    iteration: the same control sequence and costs at every check, finite throughout."""
    _, _, _, _, loop, _ = bench.build_planner("c2", math="fast")
    _, _, _, _, plain, _ = bench.build_planner("c2", math="fast")
    plain.set_debug_flags(_lib.DEBUG_NO_REDUCE_FOLD)
    for planner in (loop, plain):
        planner.solve()
    done = 0
    for call in range(400):
        k = (7, 25, 1, 64, 28)[call % 5]
        loop.iterate_async(k)
        plain.iterate_async(k)
        done += k
        if call % 40 == 39:
            loop.synchronize()
            plain.synchronize()
            u, c = loop.u_cur_d.copy_to_host(), loop.costs_d.copy_to_host()
            assert np.isfinite(u).all() and np.isfinite(c).all()
            assert np.array_equal(u, plain.u_cur_d.copy_to_host()), done
            assert np.array_equal(c, plain.costs_d.copy_to_host()), done
    assert "reduces_tiles=1" in loop.last_rollout_kernel() and done == 10000, (done, loop.last_rollout_kernel())


def test_five_thousand_iterations_of_the_direct_loop_against_k_rollout_pipe():
    """Round 5: the exact three-wave schedule INSIDE k_rollout_scan_exact (direct: maps the planner has stopped
    speculating on) shares the fold, the Philox noise in LDS and the tile packets with the speculative launch.  5 000
    iterations on the semantic map `c2s` in calls of uneven length; every 50th iteration is re-computed by
    k_rollout_pipe (a kernel of its own, noise from memory) from the same counters and controls and compared bit for
    bit, and the run is repeated with an update launch per iteration: the same control sequence."""
    _, _, _, _, loop, _ = bench.build_planner("c2s")
    _, _, _, _, plain, _ = bench.build_planner("c2s")
    _, _, _, _, check, _ = bench.build_planner("c2s")
    plain.set_debug_flags(_lib.DEBUG_NO_REDUCE_FOLD)
    check.set_debug_flags(_lib.DEBUG_NO_SCAN_DIRECT | _lib.DEBUG_NO_SCAN_KERNEL | _lib.DEBUG_NO_DEEP_KERNEL | _lib.DEBUG_NO_SPEC_KERNEL)
    for planner in (loop, plain, check):
        planner.solve()
        planner.solve()   # (the first synchronisation sees the failed tiles)
    done = 0
    for rnd in range(100):
        k = (49, 7, 31, 63, 95)[rnd % 5]
        loop.iterate_async(k)
        loop.synchronize()
        u_in = loop.u_cur_d.copy_to_host()
        loop.iterate_async(1)
        loop.synchronize()
        name = loop.last_rollout_kernel()
        assert name.startswith("k_rollout_scan_exact") and "direct=1" in name, name
        costs = loop.costs_d.copy_to_host()
        noise = loop.noise_samples_d.copy_to_host()
        check.set_noise(noise)
        check.set_u(u_in)
        check.rollout()
        assert check.last_rollout_kernel().startswith("k_rollout_pipe"), check.last_rollout_kernel()
        differing = int((costs != check.costs_d.copy_to_host()).sum())
        assert differing == 0, (rnd, differing)
        plain.iterate_async(k)
        plain.iterate_async(1)
        done += k + 1
    plain.synchronize()
    assert "direct=1" in plain.last_rollout_kernel() and "reduces_tiles" not in plain.last_rollout_kernel()
    assert np.array_equal(loop.u_cur_d.copy_to_host(), plain.u_cur_d.copy_to_host())
    assert np.array_equal(loop.costs_d.copy_to_host(), plain.costs_d.copy_to_host())
    print("\n%d iterations of the direct loop, 100 checked bit for bit against k_rollout_pipe: 0 costs differ" % done)
