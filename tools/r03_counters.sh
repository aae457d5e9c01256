#!/bin/bash
# SQ counters of the scan kernel (dynamic instruction mix), one pass each group
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${TAG:-r03c}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --math fast --steps 50 --warmup 5 --no-cpu-baseline --debug-flags ${FLAGS:-0}"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc1 -o pmc -- $BENCH > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM -d $OUT/pmc2 -o pmc -- $BENCH > $OUT/pmc2.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("pmc1","pmc2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for row in csv.DictReader(open(f)):
            k=row["Kernel_Name"].split("(")[0][:60]
            acc[k][row["Counter_Name"]]+=float(row["Counter_Value"])
            cnt[(k,row["Counter_Name"])]+=1
        for k,v in acc.items():
            print(d,k)
            for c,x in sorted(v.items()):
                print("    %-24s %14.1f per launch" % (c, x/cnt[(k,c)]))
PY
