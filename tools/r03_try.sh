#!/bin/bash
# a change to the time-parallel exact kernel, tried safely: the tests on the build whose waits trap, then the bench lines
# (build both libraries BEFORE sending: make -C mppi_numba_amd/csrc all bounded -- built .so files travel with the snapshot)
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
MPPI_HIP_LIB=$ROOT/build/libmppi_bounded.so timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py tests/test_gpu_scan.py tests/test_gpu_batch.py -q -x -p no:cacheprovider --timeout=60 --timeout-method=thread 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | head -8
for rep in 1 2; do
  timeout 100 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('exact', round(d['ms_per_step']*1e3,2), round(d['kernel_us_in_loop']['rollout'],2), round(d['kernel_us_in_loop']['update'],2))"
done
