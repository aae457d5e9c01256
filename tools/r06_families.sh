#!/bin/bash
# where launch_plan.h still selects k_rollout_deep / k_rollout_spec, what do they win -- on one box --
# against the exact pipeline (k_rollout_pipe) and the throughput kernel (k_rollout_fused)?   -> gpurun_out/<TAG>/families.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/${TAG:-r06fam}; mkdir -p $OUT; cd $ROOT
run() {  # label, env, bench args
  local label=$1 envv=$2; shift 2
  env $envv timeout 200 python bench.py "$@" --steps 200 --warmup 20 --no-cpu-baseline --regions 5 > $OUT/fam_$label.json 2> $OUT/fam_$label.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/fam_$label.json").read().strip().splitlines()[-1])
    k=d.get("kernel_us_in_loop") or {}
    print("%-28s median %7.2f us  first %7.2f  rollout %7.2f update %6.2f  %s" % ("$label", d["ms_per_step_median"]*1e3, d["ms_per_step"]*1e3, k.get("rollout") or 0, k.get("update") or 0, d["config"]["rollout_kernel"][:60]))
except Exception as e:
    print("$label failed", e)
PY
}
for cfg in "c2l:8192" "c2:16384" "c2l:16384" "c2:32768" "c2l:32768" "c2:49152"; do
  wl=${cfg%%:*}; n=${cfg#*:}
  run ${wl}_${n}_default "X=1" --workload $wl --n $n
  run ${wl}_${n}_nodeep "X=1" --workload $wl --n $n --debug-flags 4
  run ${wl}_${n}_pipe "X=1" --workload $wl --n $n --debug-flags 1
  run ${wl}_${n}_fused "MPPI_NO_PIPE=1" --workload $wl --n $n
done 2>&1 | tee $OUT/families.txt
