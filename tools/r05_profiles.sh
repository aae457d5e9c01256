#!/bin/bash
# Runs on the GPU box (gpurun).  For every "<label>:<bench args>" the untraced bench line, a kernel trace and the two HBM
# byte-counter passes (one counter per pass, never together with other traces), reduced to the text tables that go under
# profiles/.   Usage: TAG=r05a bash tools/r05_profiles.sh "c2:--workload c2" "c4shard:--workload c4 --n 8192" ...
#   -> gpurun_out/<TAG>/<TAG>_<label>_{trace,fetch,write}.txt, <TAG>_bench_<label>.json
set -u
TAG=${TAG:-r05a}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
RAW=/tmp/prof_$TAG
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
for item in "$@"; do
  label=${item%%:*}; bargs=${item#*:}
  name=${TAG}_${label}
  BENCH="python $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --regions 5 $bargs"
  timeout 150 $BENCH > $OUT/${TAG}_bench_${label}.json 2> $RAW/${name}_bench.err
  for pass in ${PASSES:-trace fetch write}; do
    case $pass in
      trace) ARGS="--kernel-trace --stats";;
      fetch) ARGS="--kernel-trace --pmc FETCH_SIZE";;
      write) ARGS="--kernel-trace --pmc WRITE_SIZE";;
    esac
    timeout 150 rocprofv3 $ARGS -d $RAW/${name}_$pass -o $pass -- $BENCH > $RAW/${name}_$pass.log 2>&1
    db=$(find $RAW/${name}_$pass -name "*_results.db" | head -1)
    python $ROOT/tools/rocpd_summary.py "$db" | sed "s#$RAW/##" > $OUT/${name}_$pass.txt 2>&1
  done
  head -4 $OUT/${name}_trace.txt | cut -c1-150
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/${TAG}_bench_${label}.json").read().strip().splitlines()[-1])
    print("$label", "us/step first %.2f median %.2f min %.2f" % (d["ms_per_step"]*1e3, d["ms_per_step_median"]*1e3, d["ms_per_step_min"]*1e3), d["kernel_us_in_loop"]["rollout"], d["kernel_us_in_loop"]["update"], "frac", d["roofline"]["frac"], d["config"]["rollout_kernel"][:50])
except Exception as e:
    print("$label no json", e)
PY
done
ls $OUT | head -40
