#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05_e}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
export ALORE_NMPC_PERSIST=0
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES -d $OUT/ic -o p -- python3 bench.py --no-cpu-baseline --no-extras --no-converged --no-graph --steps 200 --warmup 200 > /dev/null 2> $OUT/ic.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAVES -d $OUT/in -o p -- python3 bench.py --no-cpu-baseline --no-extras --no-converged --no-graph --steps 200 --warmup 200 > /dev/null 2> $OUT/in.err
rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAVES -d $OUT/dc -o p -- python3 bench.py --no-cpu-baseline --no-extras --no-converged --no-graph --steps 200 --warmup 200 > /dev/null 2> $OUT/dc.err
python3 - <<P
import sqlite3, glob
for d in ("ic","in","dc"):
    for db in glob.glob("$OUT/%s/**/*_results.db"%d, recursive=True):
        c=sqlite3.connect(db)
        try:
            tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table' or type='view'")]
            for kn,name,val,n in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%rti_block%' group by kernel_name, counter_name"):
                print(d, kn[22:70], name, val, n)
        except Exception as e:
            print("ERR", e, tabs[:20])
P
rm -rf $OUT/ic $OUT/in $OUT/dc
tail -3 $OUT/ic.err
