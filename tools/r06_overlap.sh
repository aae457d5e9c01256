#!/bin/bash
# Upper bound of "launch k+1 before launch k ends" at C2: the plain loop against the same loop with
# every other launch on a second stream and NOTHING ordering them (timing only; MPPI_EXPERIMENT_ALT_STREAMS, launch_plan.h).
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/${TAG:-r06ov}; mkdir -p $OUT; cd $ROOT
for rep in 1 2; do
  timeout 120 python bench.py --workload c2 --steps 2000 --warmup 50 --no-cpu-baseline --no-kernel-timing > $OUT/plain_$rep.json 2> $OUT/plain_$rep.err
  MPPI_EXPERIMENT_ALT_STREAMS=1 timeout 120 python bench.py --workload c2 --steps 2000 --warmup 50 --no-cpu-baseline --no-kernel-timing > $OUT/alt_$rep.json 2> $OUT/alt_$rep.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d.get("ms_per_step"), d.get("ms_per_step_median"), d.get("error"))
    except Exception as e: print(f, "unreadable", e)
PY
cd /tmp; export TMPDIR=/tmp
for mode in plain alt; do
  if [ $mode = alt ]; then export MPPI_EXPERIMENT_ALT_STREAMS=1; else unset MPPI_EXPERIMENT_ALT_STREAMS; fi
  timeout 200 rocprofv3 --kernel-trace -d /tmp/ov_$mode -o trace -- python $ROOT/bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline --no-kernel-timing --regions 1 > /tmp/ov_$mode.log 2>&1
  db=$(find /tmp/ov_$mode -name "*_results.db" | head -1)
  python $ROOT/tools/rocpd_timeline.py "$db" 16 40 > $OUT/timeline_$mode.txt 2>&1
  cut -c1-110 $OUT/timeline_$mode.txt
done
