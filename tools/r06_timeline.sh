#!/bin/bash
# kernel timeline (begin / end / gaps) of the plain loop of a bench workload:  WL=ns TAG=r06t bash tools/r06_timeline.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/${TAG:-r06t}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
WL=${WL:-ns}
timeout 200 rocprofv3 --kernel-trace -d /tmp/tl_$WL -o trace -- python $ROOT/bench.py --workload $WL ${BARGS:-} --steps 200 --warmup 20 --no-cpu-baseline --no-kernel-timing --regions 1 > /tmp/tl_$WL.log 2>&1
db=$(find /tmp/tl_$WL -name "*_results.db" | head -1)
python $ROOT/tools/rocpd_timeline.py "$db" 30 60 > $OUT/timeline_$WL.txt 2>&1
python $ROOT/tools/rocpd_summary.py "$db" > $OUT/trace_$WL.txt 2>&1
cat $OUT/timeline_$WL.txt | cut -c1-110
