"""Time the REFERENCE'S OWN CPU path (its @cuda.jit kernels under NUMBA_ENABLE_CUDASIM=1) on
BASELINE.json configs[0]: barebone unicycle MPPI, N=64 control samples, T=30 steps, flat terrain
(barebone_mppi_numba.ipynb cells 1-3; SURVEY.md section 8d "CPU reference timing (1)").

TEST / MEASUREMENT INFRASTRUCTURE ONLY (needs /root/reference and /opt/conda/bin/python3.9; the
simulator runs the kernels as Python threads, effectively one core):

    /opt/conda/bin/python3.9 oracle/time_reference_cudasim.py [--solves K] [--json OUT]

bench.py runs this as a subprocess when both exist on the box and reports it as
`cpu_baseline_reference`; on a box without the reference it reports "unavailable" and quotes the
number measured in the build container (profiles/r02_reference_cudasim.json).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
if "--reference" in sys.argv:  # (before the shim is imported: it reads MPPI_REFERENCE_ROOT at import time)
    os.environ["MPPI_REFERENCE_ROOT"] = sys.argv[sys.argv.index("--reference") + 1]
import cudasim_shim  # noqa: E402  (must precede anything that imports numba)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--solves", type=int, default=3)
    ap.add_argument("--json", default=None)
    ap.add_argument("--reference", default=None, help="checkout root of mit-acl/mppi_numba (default /root/reference)")
    args = ap.parse_args()
    ns = cudasim_shim.load_barebone_namespace()
    cfg = ns["Config"](T=3.0, dt=0.1, num_control_rollouts=100, num_vis_state_rollouts=1, seed=1)
    cfg.num_control_rollouts = 64   # below Config's clamp, set after construction
    cfg.num_vis_state_rollouts = 1
    params = dict(dt=cfg.dt, x0=np.array([0, 0, np.pi / 4]), xgoal=np.array([0.9, 0.7]), goal_tolerance=0.5,
                  dist_weight=10, lambda_weight=1.0, num_opt=1, u_std=np.array([1.0, 1.0]),
                  vrange=np.array([0.0, 2.0]), wrange=np.array([-np.pi, np.pi]))
    t0 = time.perf_counter()
    planner = ns["MPPI_Numba"](cfg)   # creates the 64*30 xoroshiro states
    planner.setup(params)
    setup_s = time.perf_counter() - t0
    planner.solve()                   # warm-up
    t0 = time.perf_counter()
    for _ in range(args.solves):
        planner.solve()               # num_opt = 1: noise + rollout + update
    per_solve = (time.perf_counter() - t0) / args.solves
    out = dict(value=cfg.num_control_rollouts / per_solve, unit="rollouts/s", cores=1, kind="reference-cudasim",
               sample="%d solve() calls (num_opt=1) of the reference's barebone MPPI_Numba, N=64, T=30, flat "
                      "terrain, kernels run by numba's CUDA simulator: %.2f s per solve (setup %.1f s)"
                      % (args.solves, per_solve, setup_s),
               seconds_per_iteration=per_solve, host_cpus=os.cpu_count())
    text = json.dumps(out)
    if args.json:
        with open(args.json, "w") as fh:
            fh.write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
