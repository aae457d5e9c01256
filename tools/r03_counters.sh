#!/bin/bash
# SQ counters of the scan kernel (dynamic instruction mix), one pass each group
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${TAG:-r03c}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --workload ${WORKLOAD:-c2} --math ${MATH:-fast} --steps 50 --warmup 5 --no-cpu-baseline --debug-flags ${FLAGS:-0}"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/sq_$$/pmc1 -o pmc -- $BENCH > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM -d /tmp/sq_$$/pmc2 -o pmc -- $BENCH > $OUT/pmc2.log 2>&1
for d in pmc1 pmc2; do
  db=$(find /tmp/sq_$$/$d -name "*_results.db" | head -1)
  python $ROOT/tools/rocpd_summary.py "$db" | grep -E "^# counters|k_rollout|k_combine|k_update" | grep -v "^void.* [0-9]+ +[0-9.]+ +[0-9.]+ +[0-9.]+ +[0-9.]+ +[0-9.]+ " | sed "s#/tmp/sq_$$/##"
done
