#!/bin/bash
# throughput regimes under math=fast: tests, then exact vs fast bench lines of c3 / c4 / c5
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${TAG:-r03s}
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_fast_throughput.py tests/test_gpu_graph.py tests/test_gpu_batch.py -q -s -p no:cacheprovider > $OUT/pytest_fast.log 2>&1
echo "pytest rc=$?"; grep -E "cost f32|u_fast|passed|failed|FAILED|Error" $OUT/pytest_fast.log | cut -c1-300
for W in ${WL:-c4 c5 c3}; do
for M in exact fast; do
  timeout 300 python bench.py --workload $W --steps 50 --warmup 5 --no-cpu-baseline --math $M > $OUT/bench_${M}_$W.json 2> $OUT/bench_${M}_$W.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_${M}_$W.json").read().strip().splitlines()[-1])
    print("$W $M", d["ms_per_step"], d["kernel_us_in_loop"]["rollout"], d["kernel_us_in_loop"]["update"], d["config"]["rollout_kernel"][:60], d.get("parity_check"))
except Exception as e:
    print("no json", e); print(open("$OUT/bench_${M}_$W.err").read()[-1500:])
PY
done
done
