"""What a CALL costs beside its iterations (C2, exact): iterate_async(K) + synchronize() for several K, the host's
share (time until iterate_async returns) and an idle synchronize.  t(K) = a + b K: b is the iteration, a the call.
  python tools/call_overhead.py [--n 8192]"""
import argparse, os, sys, time, contextlib, io
import numpy as np
sys.path.insert(0, os.getcwd())
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=8192)
ap.add_argument("--reps", type=int, default=40)
args = ap.parse_args()
with contextlib.redirect_stdout(io.StringIO()):
    _, _, lin, ang, planner, params = bench.build_planner("c2", args.n)
planner.solve()
planner.iterate_async(50); planner.synchronize()
t = []
for _ in range(200):
    t0 = time.perf_counter(); planner.synchronize(); t.append(time.perf_counter() - t0)
print("synchronize() on an idle stream: median %.2f us" % (1e6 * np.median(t)))
rows = []
for K in (1, 2, 4, 8, 20, 40, 100):
    tot, host = [], []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        planner.iterate_async(K)
        t1 = time.perf_counter()
        planner.synchronize()
        t2 = time.perf_counter()
        tot.append(t2 - t0); host.append(t1 - t0)
    rows.append((K, 1e6 * np.median(tot), 1e6 * np.min(tot), 1e6 * np.median(host), planner.last_elapsed_ms() * 1e3))
    print("K=%3d  call median %8.1f us  min %8.1f  host enqueue %7.1f  events %8.1f   per iteration %.2f" %
          (rows[-1] + (rows[-1][1] / K,)))
K = np.array([r[0] for r in rows], float); T = np.array([r[1] for r in rows])
b, a = np.polyfit(K, T, 1)
print("fit over all K: call = %.1f us + %.2f us per iteration" % (a, b))
print("kernel of the loop:", planner.last_rollout_kernel()[:100])
