#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${TAG:-r03chk}
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py tests/test_gpu_batch.py tests/test_gpu_scan.py -q -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error" | tail -3
VARIANTS=default bash tools/r03_variants.sh
