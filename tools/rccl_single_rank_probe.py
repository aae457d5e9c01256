"""What the sharded update path costs on ONE GPU: a handle with a 1-rank RCCL communicator runs
rollout -> k_update_rows<packet> -> ncclAllGather (1 rank) -> k_apply instead of rollout ->
k_update_rows<apply>.  Prints microseconds per iteration (C2 sizes) and the per-stage event brackets.
Developer tool (GPU box): python tools/rccl_single_rank_probe.py"""
import sys, os, time, numpy as np, contextlib, io
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from bench import build_planner as build
from mppi_numba_amd.mppi import comm_unique_id
with contextlib.redirect_stdout(io.StringIO()):
    w, cfg, lin, ang, planner, params = build("c2", 8192)
    planner.comm_init(comm_unique_id())
    planner.solve(); planner.iterate_async(50); planner.synchronize()
t0 = time.perf_counter(); planner.iterate_async(400); planner.synchronize(); dt = time.perf_counter() - t0
print("1-rank communicator: %.2f us per iteration" % (1e6 * dt / 400), planner.last_rollout_kernel()[:40])
planner.set_profiling(True)
acc = dict(noise=0, rollout=0, update=0, collective=0)
for _ in range(30):
    planner.iterate_async(3); planner.synchronize()
    for k, v in planner.stage_times_ms().items(): acc[k] += v / 30
print({k: round(v * 1e3, 2) for k, v in acc.items()})
