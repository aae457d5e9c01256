import sys, numpy as np
sys.path.insert(0, '.')
import bench
for wl, n in (("c2", 8192), ("c2", 1024)):
    w, cfg, lin, ang, planner, params = bench.build_planner(wl, n)
    planner.solve()
    mins = []
    for i in range(60):
        planner.iterate_async(1); planner.synchronize()
        mins.append(float(planner.costs_d.copy_to_host().min()))
    m = np.array(mins)
    print(wl, n, "min cost per iteration: first", m[:8].round(2), "range", m.min(), m.max(), "max |step|", np.abs(np.diff(m)).max(), "steps q", np.quantile(np.abs(np.diff(m)), [0.5, 0.9, 0.99]).round(2))
