#!/bin/bash
# round 5 work-in-progress check on the GPU box: selected GPU tests with per-test limits, then bench lines of the
# workloads in BENCHES ("label:bench args" items separated by ';'), each with the library in LIB (default: in-tree)
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${TAG:-r05try}
mkdir -p $OUT
cd $ROOT
if [ -n "${TESTS:-}" ]; then
  timeout ${TEST_LIMIT:-600} python -m pytest $TESTS -m gpu -q -x -p no:cacheprovider --timeout=150 --timeout-method=thread ${PYTEST_ARGS:-} > $OUT/pytest.log 2>&1
  echo "pytest rc=$?"; grep -E "passed|failed|FAILED|Error|Timeout|max .du|bit-identical" $OUT/pytest.log | tail -${TAIL:-25} | cut -c1-220
fi
IFS=';' read -ra ITEMS <<< "${BENCHES:-}"
for rep in $(seq 1 ${REPS:-1}); do
for item in "${ITEMS[@]}"; do
  label=${item%%:*}; bargs=${item#*:}
  for v in ${VARIANTS:-default}; do
    lib=$ROOT/mppi_numba_amd/libmppi_hip.so
    [ "$v" != default ] && lib=$ROOT/build/libmppi_$v.so
    MPPI_HIP_LIB=$lib timeout 120 python bench.py --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --regions 5 $bargs 2>$OUT/bench_${label}_$v.err | tee $OUT/bench_${label}_$v.json | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label $v', 'us/step first %.2f median %.2f min %.2f' % (d['ms_per_step']*1e3, d['ms_per_step_median']*1e3, d['ms_per_step_min']*1e3), 'rollout %.2f update %.2f' % (d['kernel_us_in_loop']['rollout'], d['kernel_us_in_loop']['update']), d['config'].get('rollout_kernel','')[:80])
except Exception as e:
    print('$label $v no json', e); print(open('$OUT/bench_${label}_$v.err').read()[-600:])"
  done
done
done
