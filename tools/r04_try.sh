#!/bin/bash
# round 4 work-in-progress check on the GPU box: selected GPU tests with per-test limits, then A/B bench lines
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${TAG:-r04try}
mkdir -p $OUT
cd $ROOT
if [ -n "${TESTS:-}" ]; then
  timeout ${TEST_LIMIT:-400} python -m pytest $TESTS -m gpu -q -x -p no:cacheprovider --timeout=100 --timeout-method=thread -s > $OUT/pytest.log 2>&1
  echo "pytest rc=$?"; grep -E "passed|failed|FAILED|Error|Timeout|max .du|bit-identical" $OUT/pytest.log | tail -${TAIL:-25} | cut -c1-220
fi
for rep in 1 2; do
for v in ${VARIANTS:-}; do
  envs="X=1"; lib=$ROOT/mppi_numba_amd/libmppi_hip.so
  case $v in
    default) ;;
    nofold) envs="MPPI_NO_REDUCE_FOLD=1";;
    classes*) envs="MPPI_REDUCE_CLASSES=${v#classes}";;
    *) lib=$ROOT/build/libmppi_$v.so;;
  esac
  env $envs MPPI_HIP_LIB=$lib timeout 90 python bench.py --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline ${BENCH_ARGS:-} 2>$OUT/bench_$v.err | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step']*1e3,2), round(d['kernel_us_in_loop']['rollout'],2), round(d['kernel_us_in_loop']['update'],2), d['config'].get('rollout_kernel','')[:90])
except Exception as e:
    print('$v no json', e); print(open('$OUT/bench_$v.err').read()[-600:])"
done
done
