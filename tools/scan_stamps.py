#!/usr/bin/env python3
"""In-kernel clock stamps of k_rollout_scan (developer tool; GPU box only).

    make -C mppi_numba_amd/csrc stamps
    MPPI_HIP_LIB=$PWD/build/libmppi_stamps.so python tools/scan_stamps.py [--n 8192] [--flags N]

Workgroup 5: per wave (chunk) the cycles, from the workgroup's first stamp, at which it entered the
kernel (0), had its noise and controls (1), reached barrier 1 (2), barrier 2 (3), had its stage costs
(4), reached barrier 3 (5), barrier 4 (6); wave 0 also: costs written (7); every wave: end (8)."""
import argparse
import contextlib
import ctypes as C
import io
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--flags", type=int, default=0)
    args = ap.parse_args()
    from mppi_numba_amd import _lib
    with contextlib.redirect_stdout(io.StringIO()):
        from bench import build_planner as build
        w, cfg, lin, ang, planner, params = build("c2", args.n, math="fast")
        planner.set_debug_flags(args.flags)
        planner.solve()
        planner.iterate_async(20)
        planner.synchronize()
    buf = (C.c_ulonglong * 4096)()
    _lib.call("mppi_debug_read_stamps", buf, 4096, 1)
    for rep in range(3):
        planner.iterate_async(1)
        planner.synchronize()
        _lib.call("mppi_debug_read_stamps", buf, 4096, 1)
        st = np.array(buf[:], dtype=np.uint64).astype(np.int64)
    print(planner.last_rollout_kernel())
    rows = [st[64 + 16 * c: 64 + 16 * c + 9] for c in range(16)]
    t0 = min(int(r[0]) for r in rows if r[0])
    print("chunk  " + "".join("%8d" % k for k in range(9)))
    for c, r in enumerate(rows):
        if r[0]:
            print("%5d  " % c + "".join("%8s" % (int(v - t0) if v else "-") for v in r))


if __name__ == "__main__":
    main()
