#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${TAG:-r03s2}
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "default rc=$?"; grep -E "passed|failed|FAILED" $OUT/pytest_gpu.log | tail -8 | cut -c1-200
MPPI_NO_SCAN=1 timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu_noscan.log 2>&1
echo "MPPI_NO_SCAN rc=$?"; grep -E "passed|failed|FAILED" $OUT/pytest_gpu_noscan.log | tail -40 | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
