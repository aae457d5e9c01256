#!/usr/bin/env python3
"""How far the minimum cost (beta of the update, mppi.py:1147-1154) moves from one iteration to the next, in units of
lambda -- the window a fixed-point accumulation of the update sums relative to the PREVIOUS iteration's beta would
have to cover (profiles/r05_notes.md).  GPU box only.

    python tools/r05_beta_fluct.py > profiles/r05_beta_fluct.txt
"""
import contextlib
import io
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402

for wl, n, shift in (("c2", 8192, False), ("c2", 8192, True), ("c2s", 8192, True), ("c2", 1024, True)):
    with contextlib.redirect_stdout(io.StringIO()):
        w, cfg, lin, ang, planner, params = bench.build_planner(wl, n)
        planner.solve()
    lam = float(params["lambda_weight"])
    mins = []
    for i in range(200):
        planner.iterate_async(1)
        planner.synchronize()
        mins.append(float(planner.costs_d.copy_to_host().min()))
        if shift and i % 4 == 3:  # a control step every fourth iteration: the sequence moves on by one step
            u = planner.u_cur_d.copy_to_host()
            planner.shift_and_update(params["x0"], u, num_shifts=1)
    m = np.array(mins)
    d = np.abs(np.diff(m)) / lam
    print("%-4s N=%5d %-22s beta range [%.1f, %.1f]  |beta_k+1 - beta_k| / lambda: median %.2f  p90 %.2f  p99 %.2f  max %.2f" % (
        wl, n, "with control steps" if shift else "one start state", m.min(), m.max(), np.median(d), np.quantile(d, 0.9), np.quantile(d, 0.99), d.max()))
