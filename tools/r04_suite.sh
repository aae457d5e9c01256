#!/bin/bash
# the whole GPU suite with per-test and overall limits (xdist over 4 workers), smoke, the C2 bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${TAG:-r04suite}
mkdir -p $OUT
cd $ROOT
timeout ${SUITE_LIMIT:-900} python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=150 --timeout-method=thread -n ${WORKERS:-4} ${PYTEST_ARGS:-} > $OUT/pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?"; grep -E "passed|failed|FAILED|ERROR|Timeout" $OUT/pytest_gpu.log | tail -${TAIL:-30} | cut -c1-230
if [ -z "${NO_SMOKE:-}" ]; then timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3; fi
if [ -z "${NO_BENCH:-}" ]; then
timeout 200 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
    print("bench", d["ms_per_step"], d["ms_per_step_median"], d["ms_per_step_min"], d["kernel_us_in_loop"], d["roofline"]["frac"], d["value"], d.get("cpu_baseline",{}).get("parity_check"))
except Exception as e:
    print("no json", e); print(open("$OUT/bench_default.err").read()[-800:])
PY
fi
