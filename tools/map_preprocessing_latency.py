"""Time of TDM_Numba.set_TDM_from_PMF_grid (map change -> planner-ready maps on the GPU),
numpy host path (as the reference) vs the HIP preprocessing kernel.
Usage: python tools/map_preprocessing_latency.py"""
import contextlib
import io
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_world  # noqa: E402
from mppi_numba_amd.config import Config  # noqa: E402
from mppi_numba_amd.terrain import TDM_Numba  # noqa: E402

MODES = dict(det=dict(use_det_dynamics=True), speed=dict(use_nom_dynamics_with_speed_map=True),
             tdm=dict(use_tdm=True))


def main():
    pmf, obstacle, unknown, td = synthetic_world("c3", np.random.default_rng(0))  # 16 bins, 256x256
    out = {}
    for mode, flags in MODES.items():
        for where in ("host", "device"):
            with contextlib.redirect_stdout(io.StringIO()):
                cfg = Config(T=10.0, dt=0.1, num_grid_samples=8 if mode == "tdm" else 1, num_control_rollouts=1024,
                             max_speed_padding=5.0, num_vis_state_rollouts=1, max_map_dim=(260, 260), seed=1,
                             enforce_recommended_limits=False, map_preprocessing=where, **flags)
                tdm = TDM_Numba(cfg)
                for _ in range(3):
                    tdm.set_TDM_from_PMF_grid(pmf, td, obstacle, unknown)
                ts = []
                for _ in range(20):
                    t0 = time.perf_counter()
                    tdm.set_TDM_from_PMF_grid(pmf, td, obstacle, unknown)
                    ts.append(time.perf_counter() - t0)
            out["%s_%s_ms" % (mode, where)] = round(1e3 * float(np.median(ts)), 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
