#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${TAG:-r03g}
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_scan.py -q -s -p no:cacheprovider > $OUT/pytest_scan.log 2>&1
echo "pytest scan rc=$?"; tail -3 $OUT/pytest_scan.log
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider --deselect tests/test_gpu_scan.py > $OUT/pytest_all.log 2>&1
echo "pytest all rc=$?"; tail -3 $OUT/pytest_all.log
for f in 0 256 64 128; do
  timeout 300 python bench.py --math fast --steps 200 --warmup 20 --no-cpu-baseline --debug-flags $f > $OUT/bench_fast_$f.json 2> $OUT/bench_fast_$f.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_fast_$f.json").read().strip().splitlines()[-1])
    print("fast flags $f", d["ms_per_step"], d["kernel_us_in_loop"]["rollout"], d["kernel_us_in_loop"]["update"], d["config"]["rollout_kernel"])
except Exception as e:
    print("no json", e)
PY
done
for w in c2 c3 c4 c5; do
  timeout 300 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_exact_$w.json 2> $OUT/bench_exact_$w.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_exact_$w.json").read().strip().splitlines()[-1])
    print("exact $w", d["ms_per_step"], d["kernel_us_in_loop"]["rollout"], d["kernel_us_in_loop"]["update"], d["config"]["rollout_kernel"][:40])
except Exception as e:
    print("no json", e)
PY
done
make -C mppi_numba_amd/csrc stamps > $OUT/stamps_build.log 2>&1
MPPI_HIP_LIB=$ROOT/build/libmppi_stamps.so timeout 300 python tools/scan_stamps.py --flags 0 > $OUT/stamps_0.txt 2>&1
cat $OUT/stamps_0.txt
