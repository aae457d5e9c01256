#!/bin/bash
# final verification of the round on the GPU box: the whole GPU suite (serial: the peer-exchange tests start several
# processes that must run side by side), smoke, the flag-synchronised kernels' tests against the bounded build, the
# default bench line (driver style) and the long one
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${TAG:-r06fin}
mkdir -p $OUT
cd $ROOT
timeout ${SUITE_LIMIT:-1500} python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=200 --timeout-method=thread > $OUT/pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?"; grep -E "passed|failed|FAILED|ERROR|Timeout" $OUT/pytest_gpu.log | tail -12 | cut -c1-230
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
# (make -C mppi_numba_amd/csrc bounded, before gpurun: the library travels with the snapshot)
if [ -f $ROOT/build/libmppi_bounded.so ]; then
  MPPI_HIP_LIB=$ROOT/build/libmppi_bounded.so timeout 900 python -m pytest tests/test_gpu_scan.py tests/test_gpu_fuzz.py tests/test_gpu_reduce_fold.py tests/test_gpu_semantic.py tests/test_gpu_soak.py tests/test_gpu_fold_failsoft.py -m gpu -q -p no:cacheprovider --timeout=200 > $OUT/pytest_bounded.log 2>&1
  echo "bounded build rc=$?"; tail -2 $OUT/pytest_bounded.log | cut -c1-200
fi
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_style.json 2> $OUT/bench_driver_style.err
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_c2.json 2> $OUT/bench_c2.err
for f in bench_driver_style bench_c2; do
python - <<PY
import json
try:
    d=json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1])
    print("$f", "us/step first %.2f cold %.2f median %.2f min %.2f" % (d["ms_per_step"]*1e3, d["ms_per_step_cold"]*1e3, d["ms_per_step_median"]*1e3, d["ms_per_step_min"]*1e3), d["kernel_us_in_loop"]["rollout"], d["kernel_us_in_loop"]["update"], "frac", round(d["roofline"]["frac"],4), "iter frac", round(d["roofline_iteration"]["frac"],4), "value %.3e" % d["value"], d.get("cpu_baseline",{}).get("parity_check"), d.get("cpu_baseline",{}).get("value"))
except Exception as e:
    print("$f no json", e); print(open("$OUT/$f.err").read()[-800:])
PY
done
