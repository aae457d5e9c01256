#!/bin/bash
# ns (N=65536, T=100): does kernel_us_in_loop.rollout agree with rocprofv3's average for the same kernel?
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${TAG:-r06ns}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
WL=${WL:-ns}
for variant in default nosidestream; do
  case $variant in
    default) ENVV="";;
    nosidestream) ENVV="MPPI_NO_SIDE_STREAM=1";;
  esac
  env $ENVV timeout 200 python $ROOT/bench.py --workload $WL --steps 200 --warmup 20 --no-cpu-baseline --regions 3 > $OUT/bench_${WL}_$variant.json 2> $OUT/bench_${WL}_$variant.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_${WL}_$variant.json").read().strip().splitlines()[-1])
print("$variant", "step %.2f median %.2f" % (d["ms_per_step"]*1e3, d["ms_per_step_median"]*1e3), d["kernel_us_in_loop"]["rollout"], d["kernel_us_in_loop"]["update"], d["accounting"])
PY
done
for variant in default nosidestream; do
  case $variant in default) ENVV="";; nosidestream) ENVV="MPPI_NO_SIDE_STREAM=1";; esac
  env $ENVV timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_${WL}_$variant -o trace -- python $ROOT/bench.py --workload $WL --steps 200 --warmup 20 --no-cpu-baseline --no-kernel-timing --regions 3 > /tmp/prof_${WL}_$variant.log 2>&1
  db=$(find /tmp/prof_${WL}_$variant -name "*_results.db" | head -1)
  python $ROOT/tools/rocpd_summary.py "$db" > $OUT/trace_${WL}_$variant.txt 2>&1
  echo "--- trace $variant"; head -8 $OUT/trace_${WL}_$variant.txt | cut -c1-160
done
