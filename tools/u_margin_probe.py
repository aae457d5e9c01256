"""How far the device's control update is from the oracle's on the bench's C2 problem, iteration by
iteration: max |du| / range, how many of the 2T entries differ at all, how many sit on a clip bound.
(With lambda = 1 one rollout often carries all the weight: the update is then u + that rollout's
noise in both implementations and the sequences are bit-identical; otherwise they differ by one
float32 ulp in some entries -- 8e-8 of the range, against north_star's 1e-5.)  Developer tool, GPU box:
    python tools/u_margin_probe.py"""
import sys, os, numpy as np, contextlib, io
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from test_gpu_scale import build, oracle_costs
from oracle import oracle as O
with contextlib.redirect_stdout(io.StringIO()):
    w, cfg, lin, ang, planner, params = build("c2", 8192)
    planner.solve()
res = []
for it in range(5):
    with contextlib.redirect_stdout(io.StringIO()):
        planner.sample_noise()
        noise = planner.noise_samples_d.copy_to_host(); u_in = planner.u_cur_d.copy_to_host()
        planner.rollout(); got = planner.costs_d.copy_to_host(); planner.update(); u_out = planner.u_cur_d.copy_to_host()
        want = oracle_costs(w, params, lin, ang, noise, u_in)
        wts, u_ref, _ = O.update_useq(params["lambda_weight"], want, noise, params["vrange"], params["wrange"], u_in)
    span = np.array([3.0, 2 * np.pi])
    d = np.abs(u_out - u_ref) / span
    clipped = ((u_out[:, 0] == 0) | (u_out[:, 0] == 3.0)).sum(), (np.abs(u_out[:, 1]) >= np.float32(np.pi)).sum()
    print(it, "max margin", d.max(), "entries differing", (u_out != u_ref).sum(), "of", u_out.size, "clipped v/w", clipped,
          "du max", np.abs(u_out - u_in).max())
