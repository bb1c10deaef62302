cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ALORE_LTV_STAMPS=1 python3 $R/tools/ltv_profile.py 2>&1 | grep -a "stamps\|^B="
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS -d $R/gpurun_out/h14 -o p -- python3 $R/tools/ltv_profile.py > /dev/null 2>&1
cd $R
python3 - <<'P'
import sqlite3, glob
for db in glob.glob("gpurun_out/h14/**/*_results.db", recursive=True):
    c=sqlite3.connect(db)
    for name,val,n in c.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like '%get_cmd_lanes%' group by counter_name"):
        print(name, val/1024, n)
P
rm -rf gpurun_out/h14
