#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${TAG:-r03u}
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_gpu_multi.py tests/test_gpu_scan.py tests/test_gpu_graph.py -q -s -p no:cacheprovider -k "not eight_ranks and not self_launches and not sharding_the_traction" > $OUT/pytest_fold.log 2>&1
echo "pytest rc=$?"; grep -E "us per iteration|passed|failed|FAILED|Error|assert" $OUT/pytest_fold.log | cut -c1-400 | head -40
