#!/usr/bin/env python3
"""In-kernel clock stamps of k_rollout_scan (developer tool; GPU box only).

    make -C mppi_numba_amd/csrc stamps
    MPPI_HIP_LIB=$PWD/build/libmppi_stamps.so python tools/scan_stamps.py [--n 8192] [--flags N]

Workgroup 5, cycles from the workgroup's first stamp, one row per wave.
k_rollout_scan (math fast): per wave (chunk): entered (0), had its noise and controls (1), reached
barrier 1 (2), barrier 2 (3), had its stage costs (4), reached barrier 3 (5), barrier 4 (6); wave 0 also:
costs written (7); every wave: end (8); wave 0: stage additions done (9), frozen steps done (10), weights (11).
k_rollout_scan_exact: wave 0 = heading walk, wave 1 = position walk: entered (0), walk done (1); wave 2 =
cost walk: stage walk done (1), frozen steps (2), control-cost walk (3), weights written (4), left group 0 (5), 6 (6),
9 (7), 10 .. 12 (10 .. 12); waves 3.. =
chunk waves: increments stored (1), sin / cos + position increments (3),
positions arrived (4), distances done and lookups requested (7), cells there (11), before the wait for the
earlier groups' events (12), events published (5), own events looked up and terminal cost (13), stopped rollouts
(14), selects (15), records stored (6); every wave: end (8), first barrier passed
(9); chunk waves: noise generated (10), control-cost products (2: after the records)."""
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
    ap.add_argument("--math", default="fast", choices=["fast", "exact"])
    ap.add_argument("--workload", default="c2", help="c2s / c2c: the exact three-wave schedule inside the kernel (direct mode; slots 11 .. 15 of waves 0 state, 1 cost, 2 producer: entered, first barrier passed, chunk 0 done, chunk 6 done, loop end)")
    ap.add_argument("--iterations", type=int, default=1, help="iterations per call (>1: the last launch folds the previous update)")
    args = ap.parse_args()
    from mppi_numba_amd import _lib
    with contextlib.redirect_stdout(io.StringIO()):
        from bench import build_planner as build
        w, cfg, lin, ang, planner, params = build(args.workload, args.n, math=args.math)
        planner.set_debug_flags(args.flags)
        planner.solve()
        planner.iterate_async(20)
        planner.synchronize()
    buf = (C.c_ulonglong * 4096)()
    _lib.call("mppi_debug_read_stamps", buf, 4096, 1)
    for rep in range(3):
        planner.iterate_async(args.iterations)  # (the stamps of the last launch survive)
        planner.synchronize()
        _lib.call("mppi_debug_read_stamps", buf, 4096, 1)
        st = np.array(buf[:], dtype=np.uint64).astype(np.int64)
    print(planner.last_rollout_kernel())
    # workgroup 5 and (exact kernel) workgroup 200; walkers of a launch that applies the previous update itself:
    # 5 combine begins, 6 published, 7 sequence collected
    for title, first in (("workgroup 5", 64), ("workgroup 200", 1024)):
        rows = [st[first + 16 * c: first + 16 * c + 16] for c in range(16)]
        if not any(r[0] for r in rows):
            continue
        t0 = min(int(r[0]) for r in rows if r[0])
        print(title)
        print(" wave  " + "".join("%8d" % k for k in range(16)))
        for c, r in enumerate(rows):
            if r[0]:
                print("%5d  " % c + "".join("%8s" % (int(v - t0) if v else "-") for v in r))
    rows = [st[64 + 16 * c: 64 + 16 * c + 12] for c in range(16)]
    t0 = min(int(r[0]) for r in rows if r[0])
    # every workgroup's entry / exit (slots 2048 + 2 b, 2049 + 2 b) when the kernel records them
    ent, ext = st[2048:2048 + 1024:2], st[2049:2049 + 1024:2]
    if ent.any():
        e0 = ent[ent > 0].min()
        print("workgroup entries: first %d last %d; exits: first %d last %d (cycles from the first entry)" % (
            0, int(ent.max() - e0), int(ext[ext > 0].min() - e0), int(ext.max() - e0)))
        t0 = e0
    print("k_combine_tiles, first / last workgroup: entry, loads back, minimum, sums, end (cycles from the rollout workgroup's first stamp)")
    for base in (520, 528):
        print("   ", [int(v - t0) if v else None for v in st[base:base + 5]])


if __name__ == "__main__":
    main()
