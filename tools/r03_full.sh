#!/bin/bash
# full GPU suite + the two C2 bench lines + the in-kernel timeline of the exact time-parallel kernel
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${TAG:-r03q}
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?"; tail -5 $OUT/pytest_gpu.log
for M in exact fast; do
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --math $M > $OUT/bench_${M}_c2.json 2> $OUT/bench_${M}_c2.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_${M}_c2.json").read().strip().splitlines()[-1])
    print("$M", d["ms_per_step"], d["kernel_us_in_loop"], d["config"]["rollout_kernel"][:70], d.get("parity"))
except Exception as e:
    print("no json", e); print(open("$OUT/bench_${M}_c2.err").read()[-1500:])
PY
done
make -C mppi_numba_amd/csrc stamps > $OUT/stamps_build.log 2>&1
MPPI_HIP_LIB=$ROOT/build/libmppi_stamps.so timeout 300 python tools/scan_stamps.py --flags 0 --math exact > $OUT/stamps_exact.txt 2>&1
head -16 $OUT/stamps_exact.txt
