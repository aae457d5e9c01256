#!/bin/bash
# C5: the generator beside the ROLLOUT (16 waves per CU) instead of beside the slim update -- MPPI_SIDE_STREAM_MAX_WAVES=16
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/${TAG:-r06c5s}; mkdir -p $OUT; cd $ROOT
for mode in default side16; do
  if [ $mode = side16 ]; then export MPPI_SIDE_STREAM_MAX_WAVES=16; else unset MPPI_SIDE_STREAM_MAX_WAVES; fi
  timeout 300 python bench.py --workload c5 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/c5_$mode.json 2> $OUT/c5_$mode.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/c5_$mode.json").read().strip().splitlines()[-1]); print("$mode", d.get("ms_per_step"), d.get("ms_per_step_median"), d.get("kernel_us_in_loop"), d.get("error"))
except Exception as e: print("$mode unreadable", e)
PY
done
cd /tmp; export TMPDIR=/tmp; export MPPI_SIDE_STREAM_MAX_WAVES=16
timeout 200 rocprofv3 --kernel-trace -d /tmp/c5s -o trace -- python $ROOT/bench.py --workload c5 --steps 100 --warmup 20 --no-cpu-baseline --no-kernel-timing --regions 1 > /tmp/c5s.log 2>&1
db=$(find /tmp/c5s -name "*_results.db" | head -1)
python $ROOT/tools/rocpd_timeline.py "$db" 16 40 > $OUT/timeline_side16.txt 2>&1
python $ROOT/tools/rocpd_summary.py "$db" > $OUT/trace_side16.txt 2>&1
cut -c1-110 $OUT/timeline_side16.txt
