#!/bin/bash
# Runs on the GPU box (gpurun).  For every "<label>:<bench args>": the untraced bench line (with its own in-loop kernel timing),
# then -- with --no-kernel-timing, so that the trace holds the plain loop's launches only -- a kernel trace and the two HBM
# byte-counter passes (one counter per pass, never together with other traces), reduced to the text tables under profiles/.
#   TAG=r06p bash tools/r06_profiles.sh "c2:--workload c2" "c4shard:--workload c4 --n 8192" ...
#   -> gpurun_out/<TAG>/<TAG>_<label>_{trace,fetch,write}.txt, <TAG>_bench_<label>.json
set -u
TAG=${TAG:-r06p}
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
    # (counter passes execute one kernel at a time: the two-stream loops then order their streams with events, as rounds
    #  1-5 did -- a generator that cannot run beside the launch that waits for its flag would time that launch out)
    case $pass in
      trace) ARGS="--kernel-trace --stats"; unset MPPI_NO_NOISE_FLAG;;
      fetch) ARGS="--kernel-trace --pmc FETCH_SIZE"; export MPPI_NO_NOISE_FLAG=1;;
      write) ARGS="--kernel-trace --pmc WRITE_SIZE"; export MPPI_NO_NOISE_FLAG=1;;
    esac
    timeout 150 rocprofv3 $ARGS -d $RAW/${name}_$pass -o $pass -- $BENCH --no-kernel-timing > $RAW/${name}_$pass.log 2>&1
    db=$(find $RAW/${name}_$pass -name "*_results.db" | head -1)
    python $ROOT/tools/rocpd_summary.py "$db" | sed "s#$RAW/##" > $OUT/${name}_$pass.txt 2>&1
    if [ $pass = trace ]; then python $ROOT/tools/rocpd_timeline.py "$db" 16 40 | sed "s#$RAW/##" > $OUT/${name}_timeline.txt 2>&1; fi
  done
  unset MPPI_NO_NOISE_FLAG
  head -4 $OUT/${name}_trace.txt | cut -c1-150
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/${TAG}_bench_${label}.json").read().strip().splitlines()[-1])
    k=d.get("kernel_us_in_loop") or {}
    print("$label", "us/step first %.2f median %.2f min %.2f" % (d["ms_per_step"]*1e3, d.get("ms_per_step_median",0)*1e3, d.get("ms_per_step_min",0)*1e3), k.get("rollout"), k.get("update"), "frac", d["roofline"]["frac"], "iter", d["roofline_iteration"]["frac"], d["config"]["rollout_kernel"][:50])
except Exception as e:
    print("$label no json", e)
PY
done
ls $OUT | wc -l
