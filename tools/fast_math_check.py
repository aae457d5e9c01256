"""math="fast" against the oracle on the bench's worlds: quantiles of the relative cost error, the
fraction within 1e-5 and the achieved |du| / range (developer tool, GPU box):
    python tools/fast_math_check.py"""
import sys, os, numpy as np, contextlib, io
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from test_gpu_scale import build, oracle_costs, oracle_params
from oracle import oracle as O
for wl, n in (("c2", 8192), ("c4", 8192)):
    with contextlib.redirect_stdout(io.StringIO()):
        w, cfg, lin, ang, planner, params = build(wl, n, math="fast")
        planner.solve()
        planner.sample_noise()
        noise = planner.noise_samples_d.copy_to_host()
        u_in = planner.u_cur_d.copy_to_host()
        planner.rollout()
        got = planner.costs_d.copy_to_host()
        planner.update()
        u_out = planner.u_cur_d.copy_to_host()
        want = oracle_costs(w, params, lin, ang, noise, u_in)
        _, u_ref, _ = O.update_useq(params["lambda_weight"], want, noise, params["vrange"], params["wrange"], u_in)
    rel = np.abs(got - want) / np.abs(want)
    span = np.array([params["vrange"][1] - params["vrange"][0], params["wrange"][1] - params["wrange"][0]])
    print(wl, planner.last_rollout_kernel().split(" ")[0], "rel quantiles 50/99/99.9/max", np.quantile(rel, [0.5, 0.99, 0.999, 1.0]),
          "frac<1e-5", (rel < 1e-5).mean(), "u margin", (np.abs(u_out - u_ref) / span).max())
