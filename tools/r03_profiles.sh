#!/bin/bash
# Runs on the GPU box (gpurun).  For every "<workload>:<math>" a kernel trace and the two HBM byte-counter
# passes (one counter per pass, never together with other traces), reduced to the text tables that go
# under profiles/, plus the untraced bench line.
# Usage: TAG=r03x bash tools/r03_profiles.sh c2:exact c2:fast ...  -> gpurun_out/<TAG>/<TAG>_<w>[_fast]_{trace,fetch,write}.txt
set -u
TAG=${TAG:-r03x}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
RAW=/tmp/prof_$TAG
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
for item in "$@"; do
  w=${item%%:*}; m=${item##*:}
  name=${TAG}_${w}; [ "$m" = fast ] && name=${name}_fast
  BENCH="python $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --workload $w --math $m"
  timeout 150 $BENCH > $OUT/${name/_$w/_bench_$w}.json 2> $RAW/${name}_bench.err
  for pass in trace fetch write; do
    case $pass in
      trace) ARGS="--kernel-trace --stats";;
      fetch) ARGS="--kernel-trace --pmc FETCH_SIZE";;
      write) ARGS="--kernel-trace --pmc WRITE_SIZE";;
    esac
    timeout 150 rocprofv3 $ARGS -d $RAW/${name}_$pass -o $pass -- $BENCH > $RAW/${name}_$pass.log 2>&1
    db=$(find $RAW/${name}_$pass -name "*_results.db" | head -1)
    python $ROOT/tools/rocpd_summary.py "$db" | sed "s#$RAW/##" > $OUT/${name}_$pass.txt 2>&1
  done
  head -4 $OUT/${name}_trace.txt | cut -c1-140
done
ls $OUT
