#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${TAG:-r03n}
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -q -x -p no:cacheprovider > $OUT/pytest_a.log 2>&1
echo "pytest parity+scale rc=$?"; tail -4 $OUT/pytest_a.log
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_exact_c2.json 2> $OUT/bench_exact_c2.err
python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_exact_c2.json").read().strip().splitlines()[-1])
    print("exact", d["ms_per_step"], d["kernel_us_in_loop"]["rollout"], d["kernel_us_in_loop"]["update"], d["config"]["rollout_kernel"])
except Exception as e:
    print("no json", e); print(open("$OUT/bench_exact_c2.err").read()[-1500:])
PY
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --debug-flags 32 > $OUT/bench_exact_c2_noscan.json 2>/dev/null
python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_exact_c2_noscan.json").read().strip().splitlines()[-1])
    print("exact noscan", d["ms_per_step"], d["kernel_us_in_loop"]["rollout"], d["kernel_us_in_loop"]["update"], d["config"]["rollout_kernel"][:60])
except Exception as e:
    print("no json", e)
PY
make -C mppi_numba_amd/csrc stamps > $OUT/stamps_build.log 2>&1
MPPI_HIP_LIB=$ROOT/build/libmppi_stamps.so MPPI_MATH=exact timeout 300 python tools/scan_stamps.py --flags 0 --math exact > $OUT/stamps_exact.txt 2>&1
cat $OUT/stamps_exact.txt
