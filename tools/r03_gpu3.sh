#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${TAG:-r03k}
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_all.log 2>&1
echo "pytest all rc=$?"; tail -3 $OUT/pytest_all.log
timeout 900 bash tools/asan_abi.sh > $OUT/asan.log 2>&1
echo "asan rc=$?"; tail -3 $OUT/asan.log
for f in 0 64; do
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
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_exact_c2.json 2> $OUT/bench_exact_c2.err
tail -c 600 $OUT/bench_exact_c2.json | head -c 300; echo
timeout 300 python bench.py --math fast --steps 20 --warmup 5 > $OUT/bench_fast_full.json 2> $OUT/bench_fast_full.err
python - <<PY
import json
d=json.loads(open("$OUT/bench_fast_full.json").read().strip().splitlines()[-1])
print(json.dumps({k:d[k] for k in ("value","ms_per_step","roofline","roofline_iteration","cpu_baseline","cpu_baseline_reference")}, indent=1)[:3000])
PY
