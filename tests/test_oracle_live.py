"""The oracle against the REFERENCE ITSELF on worlds that are not among the committed
fixtures: where /root/reference and the numba interpreter exist (the build container, not the
GPU box), a few fixtures are regenerated with other random maps by running the reference's
kernels under the CUDA simulator (oracle/gen_golden.py, GOLDEN_SEED_OFFSET), and the C
restatement has to reproduce them bit for bit like the committed ones."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import iterations, params_from_golden, ulp_diff_f32
from oracle import oracle as O
from test_oracle_golden import map_params, solve_of_iteration

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NUMBA_PYTHON = "/opt/conda/bin/python3.9"
NAMES = ["det_odd_units", "speedmap_odd_units", "tdm_odd_units"]  # (the quick ones: ~5 s each under the simulator)

pytestmark = pytest.mark.skipif(
    not (os.path.isdir("/root/reference/mppi_numba") and os.path.exists(NUMBA_PYTHON)),
    reason="needs the reference and its numba interpreter (build container only)")


@pytest.fixture(scope="module", params=[303])
def live_fixtures(request, tmp_path_factory):
    out = tmp_path_factory.mktemp("live_golden_%d" % request.param)
    cmd = [NUMBA_PYTHON, os.path.join(ROOT, "oracle", "gen_golden.py"), "--out", str(out)]
    for name in NAMES:
        cmd += ["--only", name]
    env = dict(os.environ, GOLDEN_SEED_OFFSET=str(request.param))
    try:
        subprocess.run(cmd, check=True, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    except (subprocess.SubprocessError, OSError) as e:  # the simulator environment, not the oracle, failed
        pytest.skip("could not run the reference under the simulator here: %r" % (e,))
    loaded = {}
    for name in NAMES:
        with np.load(os.path.join(str(out), name + ".npz")) as z:
            loaded[name] = {k: z[k] for k in z.files}
    return loaded


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_the_reference_on_other_worlds(live_fixtures, name):
    g = live_fixtures[name]
    P = params_from_golden(g)
    n_iter = 0
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
        w, u, _ = O.update_useq(P["lambda_weight"], it["costs"], it["noise"], P["vrange"], P["wrange"],
                                it["u_in"], num_threads=1)
        assert ulp_diff_f32(w, g["it%d_serial_weights" % k]).max() == 0
        assert ulp_diff_f32(u, g["it%d_serial_u_out" % k]).max() == 0
        n_iter += 1
    assert n_iter >= 1
