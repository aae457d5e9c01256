"""Round 6: in the throughput regime the next iteration's noise is generated on the planner's second stream beside the
rollout, and the two streams are ordered by words in device memory instead of cross-stream events (rollout_kernels.h:
DevParams::noise_flag / progress -- the events cost the loop ~6 us each at N = 65536).  The launch that waits for the
generator's word is bounded and fails SOFT, like the hand-over of the folded update: a host-mapped fault word, the next
draining call returns MPPI_ERR_BUSY, the handle orders its streams with events from then on (what it takes in practice:
a tool that executes one kernel at a time -- rocprofv3 --pmc).  Here the generator is silenced on purpose."""
import numpy as np
import pytest

import bench
from helpers import ulp_diff_f32
from mppi_numba_amd import _lib
from test_gpu_scale import oracle_costs

pytestmark = pytest.mark.gpu

ERR_BUSY = -6


def test_flag_ordered_loop_equals_the_stage_level_iterations():
    """N = 65536, T = 100 (the second stream is in use): the loop's controls are those of the same iterations driven
    stage by stage (noise in line), bit for bit -- with the flags and with MPPI_NO_NOISE_FLAG's events alike."""
    w, cfg, lin, ang, loop, params = bench.build_planner("ns")
    _, _, _, _, staged, _ = bench.build_planner("ns")
    loop.solve()
    loop.iterate_async(5)
    loop.synchronize()
    assert loop.last_rollout_kernel().startswith("k_rollout_fused"), loop.last_rollout_kernel()
    staged.solve()
    for _ in range(5):
        staged.sample_noise()
        staged.rollout()
        staged.update()
    assert np.array_equal(loop.u_cur_d.copy_to_host(), staged.u_cur_d.copy_to_host())


def test_a_generator_that_never_announces_itself_fails_soft_and_the_handle_falls_back_to_events():
    w, cfg, lin, ang, planner, params = bench.build_planner("ns")
    planner.solve()
    planner.iterate_async(3)
    planner.synchronize()
    u_good = planner.u_cur_d.copy_to_host()
    planner.set_debug_flags(_lib.DEBUG_DROP_NOISE_FLAG)
    planner.iterate_async(3)
    with pytest.raises(_lib.MppiError) as err:
        planner.synchronize()
    assert err.value.code == ERR_BUSY and "second stream" in str(err.value), err.value
    # the process is alive, the handle works (events from now on, whatever the test hook says), the costs are the oracle's
    planner.set_u(u_good)
    planner.iterate_async(4)
    planner.synchronize()
    assert np.isfinite(planner.u_cur_d.copy_to_host()).all()
    planner.set_debug_flags(0)
    planner.sample_noise()
    noise, u_in = planner.noise_samples_d.copy_to_host(), planner.u_cur_d.copy_to_host()
    planner.rollout()
    got = planner.costs_d.copy_to_host()
    want = oracle_costs(w, params, lin, ang, noise, u_in)
    assert (ulp_diff_f32(got, want) == 0).mean() >= 0.999


def test_event_ordered_loop_behind_the_developer_switch_gives_the_same_controls():
    """MPPI_NO_NOISE_FLAG=1 (read once per process: a process of its own): the two streams ordered by events as in rounds
    1-5 -- what a handle falls back to after a flag fault, and what tools/r06_profiles.sh uses for the counter passes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import numpy as np, bench\n"
            "w, cfg, lin, ang, p, params = bench.build_planner('ns')\n"
            "p.solve(); p.iterate_async(5); p.synchronize()\n"
            "assert p.last_rollout_kernel().startswith('k_rollout_fused')\n"
            "np.save(sys.argv[1], p.u_cur_d.copy_to_host())\n") % root
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        outs = []
        for tag, env in (("flags", {}), ("events", {"MPPI_NO_NOISE_FLAG": "1"})):
            path = os.path.join(d, tag + ".npy")
            run = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, **env), capture_output=True, text=True,
                                 timeout=180, cwd=root)
            assert run.returncode == 0, run.stderr[-800:]
            outs.append(np.load(path))
    assert np.isfinite(outs[0]).all() and np.array_equal(outs[0], outs[1])
