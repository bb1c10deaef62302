#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05_k}
mkdir -p $OUT
python -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1
grep -E "passed|failed" $OUT/pytest.txt
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PMC="bench.py --no-cpu-baseline --no-extras --no-converged --no-graph --steps 200 --warmup 200"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/sq_p1 -o p1 -- python3 $PMC > /dev/null 2> $OUT/sq_p1.err
python3 - <<P
import sqlite3, glob
for d in ("sq_p1",):
    for db in glob.glob("$OUT/%s/**/*_results.db"%d, recursive=True):
        c=sqlite3.connect(db)
        for kn,name,val,n in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%rti_block%' group by kernel_name, counter_name"):
            print(d, kn[22:60], name, val, n)
P
rm -rf $OUT/sq_p1
