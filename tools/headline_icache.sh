cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $R/gpurun_out/h17 -o p -- python3 bench.py --no-cpu-baseline --no-extras --no-converged --no-graph --steps 200 --warmup 200 > /dev/null 2>&1
python3 - <<'P'
import sqlite3, glob
for db in glob.glob("gpurun_out/h17/**/*_results.db", recursive=True):
    c=sqlite3.connect(db)
    for kn,name,val,n in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%rti_block%' group by kernel_name, counter_name"):
        print(kn[:60], name, val, n)
P
rm -rf gpurun_out/h17
