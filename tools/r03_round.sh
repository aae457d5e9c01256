#!/bin/bash
# Round-3 evidence in one GPU call: profiles of every workload in both math modes, SQ counters of the
# kernels this round added or changed, the 8-rank host-exchange bench line.
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
TAG=${TAG:-r03w}
cd $ROOT
TAG=$TAG bash tools/r03_profiles.sh ${ITEMS:-c2:exact c2:fast c4:exact c4:fast c5:exact c5:fast c3:exact c3:fast}
for cfg in ${COUNTERS:-c2:exact c2:fast c3:exact c3:fast}; do
  w=${cfg%%:*}; m=${cfg##*:}
  TAG=$TAG/sq_${w}_$m WORKLOAD=$w MATH=$m bash tools/r03_counters.sh > $ROOT/gpurun_out/$TAG/${TAG}_sq_${w}_$m.txt 2>&1
done
cd $ROOT
timeout 600 python bench.py --gpus 8 --steps 20 --warmup 5 --n 1024 --no-cpu-baseline --exchange host > gpurun_out/$TAG/${TAG}_bench_c2_8ranks_host_exchange.json 2> gpurun_out/$TAG/8ranks.err
tail -c 600 gpurun_out/$TAG/${TAG}_bench_c2_8ranks_host_exchange.json
ls gpurun_out/$TAG | head -80
