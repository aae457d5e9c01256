set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r03h
mkdir -p $OUT
cd $ROOT
make -C mppi_numba_amd/csrc stamps > $OUT/stamps_build.log 2>&1
MPPI_HIP_LIB=$ROOT/build/libmppi_stamps.so timeout 300 python tools/scan_stamps.py --flags 0 > $OUT/stamps_0.txt 2>&1
cat $OUT/stamps_0.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $ROOT/bench.py --math fast --steps 100 --warmup 10 --no-cpu-baseline > $OUT/trace.log 2>&1
python $ROOT/tools/rocpd_summary.py $(ls $OUT/trace/*.db | head -1) | head -12
python - <<PY
import sqlite3, glob
f=glob.glob("$OUT/trace/*.db")[0]
con=sqlite3.connect(f); cur=con.cursor()
rows=cur.execute("select name, start, end from kernels order by start").fetchall()
# print a window of consecutive dispatches in steady state
mid=len(rows)//2
prev=None
for name,st,en in rows[mid:mid+8]:
    print("%-40s dur %7.2f us  gap-from-prev-end %7.2f us" % (name[:40], (en-st)/1e3, (st-prev)/1e3 if prev else 0.0))
    prev=en
PY
