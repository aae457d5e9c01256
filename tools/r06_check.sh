#!/bin/bash
# Round 6 working check on the GPU box: (optionally) the GPU suite or a subset, then short bench lines.
#   TAG=r06a TESTS="tests/test_gpu_speedmap_scan.py tests/test_gpu_edges.py" BENCH="c2 ns c2m c2m1k bb" bash tools/r06_check.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${TAG:-r06a}
mkdir -p $OUT
cd $ROOT
if [ -n "${TESTS:-}" ]; then
  timeout ${SUITE_LIMIT:-1700} python -m pytest $TESTS -m gpu -q -p no:cacheprovider --timeout=200 --timeout-method=thread ${PYTEST_ARGS:-} > $OUT/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; grep -E "passed|failed|FAILED|ERROR|Timeout" $OUT/pytest_gpu.log | tail -25 | cut -c1-260
fi
for item in ${BENCH:-}; do
  wl=${item%%:*}; extra=""
  case $item in *:*) extra=$(echo ${item#*:} | tr '+' ' ');; esac
  timeout 300 python bench.py --workload $wl $extra --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --regions 5 > $OUT/bench_$(echo $item | tr ':+' '__').json 2> $OUT/bench_$(echo $item | tr ':+' '__').err
  python - <<PY
import json
f="$OUT/bench_$(echo $item | tr ':+' '__')"
try:
    d=json.loads(open(f+".json").read().strip().splitlines()[-1])
    k=d.get("kernel_us_in_loop") or {}
    print("$item", "us/step first %.2f median %.2f min %.2f" % (d["ms_per_step"]*1e3, d.get("ms_per_step_median",0)*1e3, d.get("ms_per_step_min",0)*1e3), "kern", k.get("rollout"), k.get("update"), "frac", round(d["roofline"]["frac"] or 0,4), "iter", round(d["roofline_iteration"]["frac"],4), "acct", (d.get("accounting") or {}).get("ok"), d["config"]["rollout_kernel"][:70], d.get("solve_ms_timeit_5x5"))
except Exception as e:
    print("$item no json", e); print(open(f+".err").read()[-1500:])
PY
done
