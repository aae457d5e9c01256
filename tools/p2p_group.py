#!/usr/bin/env python3
"""The peer exchange inside ONE process: a device group (mppi_group_p2p_connect: peer access, ping;
mppi_group_iterate_async: the loops of all devices enqueued a few iterations at a time, round robin).  On a box with
one GPU all shard planners sit on device 0 (their streams are different hardware queues, so their launches run side
by side like those of different devices).  Checked against twin handles whose packets are staged through the host and
applied by k_apply.  Prints GROUP_P2P_OK ... or GROUP_P2P_MISMATCH ...  (tests/test_gpu_p2p.py)"""
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
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--iterations", type=int, default=10)
    ap.add_argument("--workload", default="c2")
    args = ap.parse_args()
    import bench
    from mppi_numba_amd import _lib
    G = args.ranks
    with contextlib.redirect_stdout(io.StringIO()):
        peers = [bench.build_planner(args.workload, args.n, rank=g, world=G) for g in range(G)]
        twins = [bench.build_planner(args.workload, args.n, rank=g, world=G) for g in range(G)]
    planners = [b[4] for b in peers]
    mk = lambda hs: (C.c_void_p * G)(*hs)
    ps, lins, angs = mk([p._handle for p in planners]), mk([b[2]._handle for b in peers]), mk([b[3]._handle for b in peers])
    _lib.call("mppi_group_p2p_connect", ps, G)

    def group_iterate(k):
        for p in planners:
            p.move_mppi_task_vars_to_device()
        _lib.call("mppi_group_iterate_async", ps, lins, angs, G, int(k))
        for p in planners:
            p.synchronize()

    def staged_iterate(k):
        for _ in range(k):
            packets = []
            for b in twins:
                b[4].sample_noise()
                b[4].rollout()
                packets.append(b[4].update_local())
            for b in twins:
                b[4].update_apply(np.stack(packets))

    for b in peers + twins:
        b[2].sample_grids(1.0)
        b[3].sample_grids(1.0)
    ok, worst = True, 0.0
    for call, k in enumerate((1, args.iterations, args.iterations + 3)):
        group_iterate(k)
        staged_iterate(k)
        for g in range(G):
            a, b = planners[g].u_cur_d.copy_to_host(), twins[g][4].u_cur_d.copy_to_host()
            ok = ok and bool(np.array_equal(a, b)) and bool(np.array_equal(planners[g].costs_d.copy_to_host(), twins[g][4].costs_d.copy_to_host()))
            worst = max(worst, float(np.abs(a - b).max()))
        ok = ok and all(np.array_equal(planners[0].u_cur_d.copy_to_host(), p.u_cur_d.copy_to_host()) for p in planners)
    name = planners[0].last_rollout_kernel()
    print("%s ranks=%d n_per_rank=%d kernel=%s max|du|=%.3e exchanges=%d" % (
        "GROUP_P2P_OK" if ok else "GROUP_P2P_MISMATCH", G, args.n,
        name.split(" ")[0] + ("+direct" if "direct=1" in name else "") + ("+reduces_tiles" if "reduces_tiles=1" in name else ""),
        worst, planners[0].p2p_stats()["exchanges"]))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
