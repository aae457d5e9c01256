#!/bin/bash
# Runs on the GPU box (gpurun): kernel trace + HBM byte counters of the default bench.
# Usage: bash tools/profile_c2.sh <tag>    -> gpurun_out/prof_<tag>_{trace,fetch,write}/
set -u
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG}_trace -o trace -- $BENCH > $OUT/prof_${TAG}_trace.log 2>&1
# counters in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass)
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_${TAG}_fetch -o fetch -- $BENCH > $OUT/prof_${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_${TAG}_write -o write -- $BENCH > $OUT/prof_${TAG}_write.log 2>&1
grep -h '^{' $OUT/prof_${TAG}_trace.log | tail -1
ls $OUT/prof_${TAG}_*/
