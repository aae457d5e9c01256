"""Host-visible latency of one closed-loop control step (solve + shift), per mode.

The reference publishes 2.74 ms per solve (barebone, N=1000, T=50, RTX 3070) and
about 3.3 ms per control step in det mode at N=1024, T=100.
This measures the same host-side sequence on this machine:
    useq = planner.solve(); planner.shift_and_update(x, useq, 1)
Usage: python tools/control_step_latency.py [--n 1024] [--t 100] [--num-opt 1]
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_world, make_params  # noqa: E402
from mppi_numba_amd.config import Config  # noqa: E402
from mppi_numba_amd.mppi import MPPI_Numba  # noqa: E402
from mppi_numba_amd.terrain import TDM_Numba  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--t", type=int, default=100)
    ap.add_argument("--num-opt", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--m", type=int, default=1, help="> 1: use_tdm with that many traction samples (CVaR)")
    args = ap.parse_args()
    with contextlib.redirect_stdout(io.StringIO()):
        mode = dict(use_tdm=True) if args.m > 1 else dict(use_det_dynamics=True)
        cfg = Config(T=args.t * 0.1, dt=0.1, num_grid_samples=args.m, num_control_rollouts=args.n,
                     max_speed_padding=5.0, num_vis_state_rollouts=1, max_map_dim=(260, 260), seed=1,
                     enforce_recommended_limits=False, **mode)
        pmf, obstacle, unknown, tdm_dict = synthetic_world("c3" if args.m > 1 else "c2", np.random.default_rng(0))
        lin, ang = TDM_Numba(cfg), TDM_Numba(cfg)
        lin.set_TDM_from_PMF_grid(pmf, tdm_dict, obstacle, unknown)
        ang.set_TDM_from_PMF_grid(pmf, tdm_dict, obstacle, unknown)
        planner = MPPI_Numba(cfg)
        params = make_params("c3" if args.m > 1 else "c2")
        params["num_opt"] = args.num_opt
        planner.setup(params, lin, ang)
    x = np.array(params["x0"], dtype=np.float32)
    for _ in range(20):
        useq = planner.solve()
        planner.shift_and_update(x, useq, 1)
    t_solve, t_shift = [], []
    for _ in range(args.steps):
        t0 = time.perf_counter()
        useq = planner.solve()
        t1 = time.perf_counter()
        planner.shift_and_update(x, useq, 1)
        t2 = time.perf_counter()
        t_solve.append(t1 - t0)
        t_shift.append(t2 - t1)
    us = lambda v: round(1e6 * float(np.median(v)), 1)
    print(json.dumps(dict(n=args.n, t=args.t, m=args.m, num_opt=args.num_opt, solve_us=us(t_solve),
                          shift_and_update_us=us(t_shift),
                          control_step_us=us(np.add(t_solve, t_shift)),
                          kernel=planner.last_rollout_kernel())))


if __name__ == "__main__":
    main()
