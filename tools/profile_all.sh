#!/bin/bash
# Runs on the GPU box (gpurun): for every bench workload a kernel trace and the two HBM byte-counter
# passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; counters never share a run with other traces).
# Usage: bash tools/profile_all.sh <tag> [workloads...]   -> gpurun_out/prof_<tag>_<workload>_{trace,fetch,write}/
set -u
TAG=${1:-x}; shift
WL=${@:-c2 c3 c4 c5}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in $WL; do
  BENCH="python $ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --workload $w"
  rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG}_${w}_trace -o trace -- $BENCH > $OUT/prof_${TAG}_${w}_trace.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_${TAG}_${w}_fetch -o fetch -- $BENCH > $OUT/prof_${TAG}_${w}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_${TAG}_${w}_write -o write -- $BENCH > $OUT/prof_${TAG}_${w}_write.log 2>&1
  grep -h '^{' $OUT/prof_${TAG}_${w}_trace.log | tail -1 > $OUT/prof_${TAG}_${w}_bench.json
done
ls $OUT | grep prof_${TAG}_
