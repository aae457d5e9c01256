#!/bin/bash
# after the exact time-parallel kernel became a pipeline of walks: its trace, byte counters, SQ counters, stamps
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
TAG=${TAG:-r03x}
cd $ROOT
TAG=$TAG bash tools/r03_profiles.sh c2:exact
TAG=$TAG COUNTERS="c2:exact" bash tools/r03_sq.sh | tail -5
make -C mppi_numba_amd/csrc stamps > /tmp/stamps_build.log 2>&1
MPPI_HIP_LIB=$ROOT/build/libmppi_stamps.so timeout 300 python tools/scan_stamps.py --flags 0 --math exact > $ROOT/gpurun_out/$TAG/${TAG}_stamps_scan_exact.txt 2>&1
cat $ROOT/gpurun_out/$TAG/${TAG}_stamps_scan_exact.txt | head -4
timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_scan.py tests/test_gpu_graph.py tests/test_gpu_batch.py tests/test_gpu_edges.py -q -p no:cacheprovider -k "not eight_ranks and not self_launches and not sharding_the_traction" 2>&1 | tail -3
