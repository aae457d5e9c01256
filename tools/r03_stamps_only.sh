#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${TAG:-r03st}
mkdir -p $OUT
cd $ROOT
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_exact_c2.json 2> $OUT/bench_exact_c2.err
python - <<PY
import json
d=json.loads(open("$OUT/bench_exact_c2.json").read().strip().splitlines()[-1])
print("exact", d["ms_per_step"], d["kernel_us_in_loop"]["rollout"], d["kernel_us_in_loop"]["update"], d["parity_check"] if "parity_check" in d else "")
PY
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -q -x -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tail -2
make -C mppi_numba_amd/csrc stamps > $OUT/stamps_build.log 2>&1
MPPI_HIP_LIB=$ROOT/build/libmppi_stamps.so timeout 300 python tools/scan_stamps.py --flags 0 --math exact > $OUT/stamps_exact.txt 2>&1
head -19 $OUT/stamps_exact.txt
