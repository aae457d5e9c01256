"""What a rank sees when its peer dies (the peer exchange): MPPI_ERR_COMM from the call that synchronises next, after
about two seconds -- not a hung device, not an aborted process.  Prints DEAD_PEER_OK ... on success (tests/test_gpu_p2p.py)."""
import os, sys, time, contextlib, io
sys.path.insert(0, os.getcwd())
from mppi_numba_amd import launch
if not launch.launched_by_a_launcher():
    sys.exit(launch.spawn_ranks(2, [sys.executable, os.path.abspath(__file__)], timeout=120))
rank, _, world = launch.rank_from_env()
hub = launch.Hub(rank, world)
import bench
with contextlib.redirect_stdout(io.StringIO()):
    _, _, lin, ang, peer, params = bench.build_planner("c2", 1024, rank=rank, world=world)
peer.p2p_connect(hub.all_gather(peer.p2p_export()))
hub.barrier()
assert peer.p2p_ping(7) == world
peer.solve()
hub.barrier()
if rank == 1:
    print("rank 1 leaves", file=sys.stderr); hub.close(); os._exit(0)
t0 = time.time()
try:
    peer.iterate_async(4); peer.synchronize()
    print("DEAD_PEER_NO_ERROR")
except Exception as e:
    ok = "did not arrive" in str(e) and time.time() - t0 < 30.0
    # the handle works again without the exchange (the device was never hung): stage-level calls, packets for a host-staged update
    try:
        peer.p2p_enable(False)
        peer.synchronize()
        peer.sample_noise(); peer.rollout()
        import numpy as np
        ok = ok and bool(np.isfinite(peer.update_local()).all()) and bool(np.isfinite(peer.costs_d.copy_to_host()).all())
    except Exception as e2:
        ok = False
        print("after the failure:", e2, file=sys.stderr)
    print("%s after %.2f s: %s" % ("DEAD_PEER_OK" if ok else "DEAD_PEER_OTHER_ERROR", time.time() - t0, str(e)[:160]))
sys.stdout.flush()
os._exit(0)
