cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $R/gpurun_out/h11_ic -o p -- python3 $R/tools/be_profile.py > /dev/null 2> $R/gpurun_out/h11_ic.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM -d $R/gpurun_out/h11_sq -o p -- python3 $R/tools/be_profile.py > /dev/null 2>> $R/gpurun_out/h11_ic.err
cd $R
python3 - <<'P'
import sqlite3, glob
for d in ("gpurun_out/h11_ic","gpurun_out/h11_sq"):
    for db in glob.glob(d+"/**/*_results.db", recursive=True):
        c=sqlite3.connect(db)
        for name,val,n in c.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like '%backend_kernel%' group by counter_name"):
            print(name, val, n)
P
