"""hipGraph replay vs direct launches of the MPPI iteration loop (developer measurement).
Usage: python tools/graph_probe.py"""
import contextlib
import ctypes as C
import io
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mppi_numba_amd import _lib  # noqa: E402
from mppi_numba_amd.config import Config  # noqa: E402
from mppi_numba_amd.mppi import MPPI_Numba  # noqa: E402
from mppi_numba_amd.terrain import TDM_Numba  # noqa: E402


def main():
    out = []
    for label, workload, n, iterations in (("N=1024", "c2", 1024, 2), ("C2 N=8192", "c2", 8192, 2),
                                           ("C2 N=8192", "c2", 8192, 8), ("C3", "c3", 4096, 2),
                                           ("C4 N=65536", "c4", 65536, 2)):
        w = bench.WORKLOADS[workload]
        with contextlib.redirect_stdout(io.StringIO()):
            cfg = Config(T=w["t"] * 0.1, dt=0.1, num_grid_samples=w["m"], num_control_rollouts=n,
                         max_speed_padding=5.0, num_vis_state_rollouts=1, max_map_dim=(260, 260), seed=1,
                         enforce_recommended_limits=False, **w["mode"])
            pmf, obstacle, unknown, td = bench.synthetic_world(workload, np.random.default_rng(0))
            lin, ang = TDM_Numba(cfg), TDM_Numba(cfg)
            lin.set_TDM_from_PMF_grid(pmf, td, obstacle, unknown)
            ang.set_TDM_from_PMF_grid(pmf, td, obstacle, unknown)
            planner = MPPI_Numba(cfg)
            planner.setup(bench.make_params(workload), lin, ang)
            planner.solve()
        direct, graph = C.c_float(0), C.c_float(0)
        _lib.call("mppi_planner_graph_probe", planner._handle, lin._handle, ang._handle, iterations, 200,
                  C.byref(direct), C.byref(graph))
        out.append(dict(case=label, iterations_per_graph=iterations, direct_us_per_iteration=round(direct.value, 2),
                        graph_us_per_iteration=round(graph.value, 2)))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
