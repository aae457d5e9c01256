"""Closed loop of the notebooks (solve -> world lookup -> Euler step -> shift_and_update), driven
from the host as the notebook does versus MPPI_Numba.closed_loop on the device.

Usage: python tools/closed_loop_latency.py [--n 1024] [--t 100] [--steps 300]
Prints one JSON line: microseconds per control step for both."""
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
from mppi_numba_amd.terrain import TDM_Numba, TractionGrid  # noqa: E402


def build(args):
    cfg = Config(T=args.t * 0.1, dt=0.1, num_grid_samples=1, num_control_rollouts=args.n, max_speed_padding=5.0,
                 num_vis_state_rollouts=1, max_map_dim=(260, 260), seed=1, enforce_recommended_limits=False,
                 use_det_dynamics=True)
    pmf, obstacle, unknown, tdm_dict = synthetic_world("c2", np.random.default_rng(0))
    lin, ang = TDM_Numba(cfg), TDM_Numba(cfg)
    lin.set_TDM_from_PMF_grid(pmf, tdm_dict, obstacle, unknown)
    ang.set_TDM_from_PMF_grid(pmf, tdm_dict, obstacle, unknown)
    planner = MPPI_Numba(cfg)
    params = make_params("c2")
    params["goal_tolerance"] = 0.5
    planner.setup(params, lin, ang)
    return cfg, planner, params


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--t", type=int, default=100)
    ap.add_argument("--steps", type=int, default=300)
    args = ap.parse_args()
    rng = np.random.default_rng(3)
    world = TractionGrid(rng.uniform(0.4, 0.9, (256, 256)), rng.uniform(0.4, 0.9, (256, 256)), res=0.25)
    with contextlib.redirect_stdout(io.StringIO()):
        cfg, host_planner, params = build(args)
        _, dev_planner, _ = build(args)
    # the host loop (test.ipynb cell 4 without the plots)
    x = np.array(params["x0"], dtype=np.float64)
    for warm in (True, False):
        x = np.array(params["x0"], dtype=np.float64)
        host_planner.params["x0"] = x.copy()
        t0 = time.perf_counter()
        for _ in range(20 if warm else args.steps):
            useq = host_planner.solve()
            lt, at = world.get(x[0], x[1])
            x = x + cfg.dt * np.array([lt * np.cos(x[2]) * useq[0, 0], lt * np.sin(x[2]) * useq[0, 0], at * useq[0, 1]])
            host_planner.shift_and_update(x, useq, num_shifts=1)
        host_s = time.perf_counter() - t0
    # the device loop (never stops early: tolerance 0)
    dev_planner.closed_loop(world, 20, goal_tolerance=0.0)
    dev_planner.params["x0"] = np.array(params["x0"], dtype=np.float64)
    t0 = time.perf_counter()
    xh, uh, steps = dev_planner.closed_loop(world, args.steps, goal_tolerance=0.0)
    dev_s = time.perf_counter() - t0
    assert steps == args.steps and np.isfinite(xh).all()
    print(json.dumps({"n": args.n, "t": args.t, "steps": args.steps,
                      "host_loop_us_per_control_step": 1e6 * host_s / args.steps,
                      "device_loop_us_per_control_step": 1e6 * dev_s / args.steps,
                      "rollout_kernel": dev_planner.last_rollout_kernel().split(" ")[0]}))


if __name__ == "__main__":
    main()
