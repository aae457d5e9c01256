#!/bin/bash
# final verification of the round: the whole GPU suite (per-test and overall time limits), smoke, the two C2 bench lines
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${TAG:-r03fin}
mkdir -p $OUT
cd $ROOT
timeout 420 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=120 --timeout-method=thread > $OUT/pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?"; grep -E "passed|failed|FAILED|Timeout" $OUT/pytest_gpu.log | tail -8 | cut -c1-200
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
for M in exact fast; do
  timeout 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --math $M > $OUT/bench_${M}_c2.json 2> $OUT/bench_${M}_c2.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_${M}_c2.json").read().strip().splitlines()[-1])
    print("$M", d["ms_per_step"], d["kernel_us_in_loop"]["rollout"], d["kernel_us_in_loop"]["update"], d["roofline"]["frac"], d["value"])
except Exception as e:
    print("no json", e)
PY
done
