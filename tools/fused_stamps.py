"""In-kernel clock stamps of k_rollout_fused at a bench workload (developer tool; `make -C mppi_numba_amd/csrc stamps`,
MPPI_HIP_LIB=build/libmppi_stamps.so): prologue | step loop | control-cost pass, cycles of s_memtime, workgroup 5 wave 0."""
import ctypes as C, sys, os, io, contextlib
sys.path.insert(0, os.getcwd())
import bench
from mppi_numba_amd import _lib
wl = sys.argv[1] if len(sys.argv) > 1 else "ns"
with contextlib.redirect_stdout(io.StringIO()):
    w, cfg, lin, ang, planner, params = bench.build_planner(wl)
    planner.solve(); planner.iterate_async(20); planner.synchronize()
buf = (C.c_ulonglong * 1024)()
_lib.call("mppi_debug_read_stamps", buf, 1024, 1)
for rep in range(3):
    planner.iterate_async(1); planner.synchronize()
    _lib.call("mppi_debug_read_stamps", buf, 1024, 1)
    st = [buf[710 + i] for i in range(4)]
    ex = [buf[714 + i] for i in range(3)]
    if all(ex): print('   pass detail: terminal %d | batches 0-6 %d | batches 7-11 %d | tail steps %d' % (ex[0]-st[2], ex[1]-ex[0], ex[2]-ex[1], st[3]-ex[2]))
    print(planner.last_rollout_kernel()[:60], "| prologue %d  step loop %d  terminal + control-cost pass %d  total %d cycles" %
          (st[1] - st[0], st[2] - st[1], st[3] - st[2], st[3] - st[0]))
