#!/usr/bin/env python3
"""The peer exchange end to end with several processes (ranks) -- on a box with ONE GPU they share the device and
reach each other's inboxes through IPC handles, which is exactly the one-process-per-GPU set-up minus xGMI.

    python tools/p2p_ranks.py --ranks 2 --n 1024 --t 100 --iterations 6

Every rank runs the same sequence of calls twice: with the peer exchange (a sharded iteration is ONE launch, the
numbers of each step cross inside the rollout launch or inside the closing update launch) and, on a twin handle,
with the packets staged through the host hub and applied by k_apply (mppi_planner_update_local / update_apply).
The control sequences must agree bit for bit, on every rank and between the ranks.  Rank 0 prints one line:
P2P_OK ... or P2P_MISMATCH ...  (tests/test_gpu_p2p.py runs this)"""
import argparse
import contextlib
import io
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--n", type=int, default=1024, help="control samples per rank")
    ap.add_argument("--t", type=int, default=100)
    ap.add_argument("--iterations", type=int, default=6)
    ap.add_argument("--calls", type=int, default=3)
    ap.add_argument("--time", type=int, default=0, help="also time this many iterations of each exchange")
    ap.add_argument("--math", default="exact", choices=["exact", "fast"])
    ap.add_argument("--workload", default="c2", choices=["c2", "c2s", "c2c"],
                    help="c2s / c2c: maps on which tiles fail their traction vote and the planners stop speculating -- every "
                         "rank at its own synchronisation: the kernel family (and with it the exchange) must survive that")
    args = ap.parse_args()
    from mppi_numba_amd import launch
    if not launch.launched_by_a_launcher():
        sys.exit(launch.spawn_ranks(args.ranks, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], timeout=600))
    rank, local_rank, world = launch.rank_from_env()
    hub = launch.Hub(rank, world)
    import bench
    from mppi_numba_amd import _lib
    saved = dict(bench.WORKLOADS[args.workload])
    bench.WORKLOADS[args.workload] = dict(saved, t=args.t)
    with contextlib.redirect_stdout(io.StringIO()):
        _, _, lin, ang, peer, params = bench.build_planner(args.workload, args.n, rank=rank, world=world, math=args.math)
        _, _, lin2, ang2, staged, _ = bench.build_planner(args.workload, args.n, rank=rank, world=world, math=args.math)
    handles = hub.all_gather(peer.p2p_export())
    peer.p2p_connect(handles)
    hub.barrier()
    heard = peer.p2p_ping(0xabc0 + 1, timeout_ms=2000)
    assert heard == world, "ping: %d of %d ranks heard" % (heard, world)

    def staged_iterations(k):
        for _ in range(k):
            staged.sample_noise()
            staged.rollout()
            staged.update_apply(np.stack(hub.all_gather(staged.update_local())))

    def staged_solve():
        lin2.sample_grids(1.0)
        ang2.sample_grids(1.0)
        staged_iterations(1)

    ok, worst = True, 0.0
    hub.barrier()
    peer.solve()  # (one iteration: the exchange inside the closing update launch)
    staged_solve()
    for call in range(args.calls):
        hub.barrier()  # (the ranks enter their loops together: a late rank would look like a dead one)
        peer.iterate_async(args.iterations)
        peer.synchronize()
        staged_iterations(args.iterations)
        u_p, u_s = peer.u_cur_d.copy_to_host(), staged.u_cur_d.copy_to_host()
        same = bool(np.array_equal(u_p, u_s)) and bool(np.array_equal(peer.costs_d.copy_to_host(), staged.costs_d.copy_to_host()))
        worst = max(worst, float(np.abs(u_p - u_s).max()))
        ok = ok and same
    name = peer.last_rollout_kernel()
    all_u = hub.all_gather(peer.u_cur_d.copy_to_host().ravel())
    ok = ok and all(np.array_equal(all_u[0], v) for v in all_u)
    timing = ""
    if args.time:
        for label, fn in (("p2p", lambda: (peer.iterate_async(args.time), peer.synchronize())), ("host-staged", lambda: staged_iterations(args.time))):
            hub.barrier()
            t0 = time.perf_counter()
            fn()
            dt = hub.all_max(time.perf_counter() - t0)
            timing += " %s_us_per_iteration=%.2f" % (label, 1e6 * dt / args.time)
    oks = hub.all_gather(1 if ok else 0)
    worsts = hub.all_gather(worst)
    stats = peer.p2p_stats()
    if rank == 0:
        print("%s world=%d n_per_rank=%d T=%d exchanges=%d inbox=%s kernel=%s max|du|=%.3e%s" % (
            "P2P_OK" if all(oks) else "P2P_MISMATCH", world, args.n, args.t, stats["exchanges"], stats["inbox"],
            name.split(" ")[0] + ("+direct" if "direct=1" in name else "") + ("+reduces_tiles" if "reduces_tiles=1" in name else ""), max(worsts), timing))
        sys.stdout.flush()
    hub.barrier()
    hub.close()
    sys.exit(0 if all(oks) else 1)


if __name__ == "__main__":
    main()
