"""How long a call of K iterations takes after the GPU has sat idle for a while (C2): iterate_async(K) + synchronize()
behind time.sleep(gap).  python tools/idle_gap_probe.py [--k 20]"""
import argparse, os, sys, time, contextlib, io
import numpy as np
sys.path.insert(0, os.getcwd())
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--k", type=int, default=20)
ap.add_argument("--reps", type=int, default=30)
args = ap.parse_args()
with contextlib.redirect_stdout(io.StringIO()):
    _, _, lin, ang, planner, params = bench.build_planner("c2", 8192)
planner.solve()
planner.iterate_async(50); planner.synchronize()
for gap_us in (0, 20, 100, 1000, 10000, 100000):
    for warm in (0, 5):
        tot = []
        for _ in range(args.reps):
            if gap_us:
                t = time.perf_counter()
                if gap_us >= 1000:
                    time.sleep(gap_us * 1e-6)
                else:
                    while time.perf_counter() - t < gap_us * 1e-6:
                        pass
            if warm:
                planner.iterate_async(warm); planner.synchronize()
            t0 = time.perf_counter()
            planner.iterate_async(args.k)
            planner.synchronize()
            tot.append(time.perf_counter() - t0)
        print("idle %7d us, then %d warm-up iterations: call of %d median %7.1f us  min %7.1f  -> %.2f us per iteration" %
              (gap_us, warm, args.k, 1e6 * np.median(tot), 1e6 * np.min(tot), 1e6 * np.median(tot) / args.k))
